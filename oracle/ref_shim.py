"""ORACLE SUPPORT (test infrastructure only) — import the upstream reference's own Python modules on CPU.

Only usable in the authoring container where /root/reference exists (it does not exist on the GPU
box): used by oracle/make_golden.py to mint tests/golden/*.pt and by the optional
tests/test_oracle_vs_reference.py.  Recipe from SURVEY.md Appendix A:
  * bypass wan/__init__.py (easydict / xfuser imports) with stub packages whose __path__ points at the
    reference directories;
  * stub the diffusers mixins (base-class sugar only, causal_model.py:14-16);
  * replace sinusoidal_embedding_1d, which hard-codes torch.cuda.current_device() (model.py:22);
  * stub tokenizers / t5 / settings so utils/wan_wrapper.py imports.
Nothing here is copied from the reference; it only arranges for its modules to import.
"""
import importlib.machinery
import os
import sys
import types

import torch

REF = os.environ.get("RTV_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "wan", "modules"))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__spec__.submodule_search_locations = [path]
    sys.modules[name] = m
    return m


_loaded = {}


def load():
    """Returns a namespace with the reference modules: cm (causal_model), model, attention,
    scheduler, vae, vae_block3, wan_wrapper, v2v_schedule."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    _pkg("wan", REF + "/wan")
    _pkg("wan.modules", REF + "/wan/modules")
    for n in ("diffusers", "diffusers.configuration_utils", "diffusers.models",
              "diffusers.models.modeling_utils"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["diffusers.configuration_utils"].ConfigMixin = type("ConfigMixin", (), {})
    sys.modules["diffusers.configuration_utils"].register_to_config = lambda f: f
    sys.modules["diffusers.models.modeling_utils"].ModelMixin = type("ModelMixin", (torch.nn.Module,), {})
    # stubs for modules wan_wrapper imports but the hot path never touches
    tok = types.ModuleType("wan.modules.tokenizers")
    tok.HuggingfaceTokenizer = type("HuggingfaceTokenizer", (), {})
    sys.modules["wan.modules.tokenizers"] = tok
    t5 = types.ModuleType("wan.modules.t5")
    t5.umt5_xxl = lambda *a, **k: None
    sys.modules["wan.modules.t5"] = t5
    st = types.ModuleType("settings")
    st.MODEL_FOLDER = "/nonexistent"
    sys.modules["settings"] = st
    if REF not in sys.path:
        sys.path.insert(0, REF)

    import wan.modules.causal_model as cm
    import wan.modules.model as model
    import wan.modules.attention as attention
    import wan.modules.vae as vae
    from utils import scheduler
    import demo_utils.vae_block3 as vae_block3
    import utils.wan_wrapper as wan_wrapper

    def sinusoidal_embedding_1d(dim, position):  # device-agnostic restatement of model.py:15-24
        half = dim // 2
        position = position.type(torch.float64)
        sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half, dtype=torch.float64).div(half)))
        return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)

    cm.sinusoidal_embedding_1d = sinusoidal_embedding_1d
    model.sinusoidal_embedding_1d = sinusoidal_embedding_1d

    def get_denoising_schedule(timesteps, denoising_strength, steps=4):
        # v2v.py itself imports cv2/requests; execute only the 4-line function from its source
        src = open(os.path.join(REF, "v2v.py")).read()
        start = src.index("def get_denoising_schedule")
        end = src.index("def encode_video_latent")
        ns = {"torch": torch}
        exec(src[start:end], ns)
        return ns["get_denoising_schedule"](timesteps, denoising_strength, steps)

    _loaded.update(cm=cm, model=model, attention=attention, scheduler=scheduler, vae=vae,
                   vae_block3=vae_block3, wan_wrapper=wan_wrapper,
                   get_denoising_schedule=get_denoising_schedule)
    return types.SimpleNamespace(**_loaded)


def build_reference_model(ref, cfg, weights, text_dim):
    """Instantiate the reference CausalWanModel with our synthetic weights (bf16, eval)."""
    m = ref.cm.CausalWanModel(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                              num_layers=cfg["num_layers"], text_dim=text_dim,
                              freq_dim=cfg.get("freq_dim", 256),
                              local_attn_size=cfg.get("local_attn_size", -1),
                              sink_size=cfg.get("sink_size", 0)).eval()
    dtype = next(iter(weights.values())).dtype
    m = m.to(dtype)
    missing, unexpected = m.load_state_dict(weights, strict=False)
    assert not unexpected, unexpected
    assert not missing, missing
    m.num_frame_per_block = cfg.get("num_frame_per_block", 3)
    for blk in m.blocks:
        blk.self_attn.num_frame_per_block = m.num_frame_per_block
    return m


def build_reference_wrapper(ref, model, shift=5.0):
    """WanDiffusionWrapper without from_pretrained (utils/wan_wrapper.py:121-154)."""
    W = ref.wan_wrapper.WanDiffusionWrapper
    wr = W.__new__(W)
    torch.nn.Module.__init__(wr)
    wr.model = model
    wr.uniform_timestep = False
    wr.scheduler = ref.scheduler.FlowMatchScheduler(shift=shift, sigma_min=0.0, extra_one_step=True)
    wr.scheduler.set_timesteps(1000, training=True)
    wr.seq_len = 32760
    return wr


def load_t5():
    """The reference's real wan/modules/t5.py (load() stubs it because importing it evaluates
    `torch.cuda.current_device()` as a default argument, t5.py:476).  Imported under another module name inside the
    `wan.modules` package, with that call patched for the duration of the import; its `.tokenizers` import resolves to
    the stub (ftfy is not installed)."""
    import importlib.util
    load()
    name = "wan.modules.t5_reference"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "wan", "modules", "t5.py"))
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "wan.modules"
    sys.modules[name] = mod
    real = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0
    try:
        spec.loader.exec_module(mod)
    finally:
        torch.cuda.current_device = real
    return mod


class _NoCudaObject:
    """Stand-in for torch.cuda.Stream / torch.cuda.Event while the reference's serving module runs on the CPU."""

    def __init__(self, *a, **k):
        pass

    def record(self, *a, **k):
        pass

    def wait_event(self, *a, **k):
        pass

    def synchronize(self):
        pass


def load_release_server():
    """The reference's release_server.py (GenerationSession, GenerateParams, Models) and its CausalInferencePipeline,
    imported on the CPU: module-level CUDA calls (release_server.py:88-90, demo_utils/memory.py:9) are patched so that `gpu`
    is the CPU, third-party modules that are absent and unused on this path (omegaconf, torchvision, cv2, requests) are
    stubbed, `pipeline/__init__.py` (which imports the training pipelines) is bypassed.  The patches stay in place: this
    is a one-way switch for golden-minting processes only.  Returns (release_server module, CausalInferencePipeline)."""
    load()
    if "release_server" in sys.modules:
        return sys.modules["release_server"], sys.modules["pipeline.causal_inference"].CausalInferencePipeline
    for n in ("omegaconf", "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "cv2", "requests"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["omegaconf"].OmegaConf = type("OmegaConf", (), {})
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]

    def to_tensor(pic):     # torchvision.transforms.functional.to_tensor for 8-bit RGB PIL images (published semantics)
        import numpy as np
        return torch.from_numpy(np.asarray(pic).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    sys.modules["torchvision.transforms.functional"].to_tensor = to_tensor
    # the HTTP / WebSocket layer below the session class is control plane: its decorators become no-ops
    class _App:
        def __init__(self, *a, **k):
            self.state = types.SimpleNamespace()

        def __getattr__(self, name):
            return lambda *a, **k: (lambda f: f)

    fa = types.ModuleType("fastapi")
    fa.FastAPI = _App
    fa.File = lambda *a, **k: None
    for n in ("WebSocket", "WebSocketDisconnect", "UploadFile"):
        setattr(fa, n, type(n, (Exception,), {}))
    fam, fac, far = (types.ModuleType(n) for n in ("fastapi.middleware", "fastapi.middleware.cors", "fastapi.responses"))
    fac.CORSMiddleware = object
    far.HTMLResponse = far.JSONResponse = object
    sys.modules.update({"fastapi": fa, "fastapi.middleware": fam, "fastapi.middleware.cors": fac, "fastapi.responses": far})
    torch.cuda.current_device = lambda: 0
    import demo_utils.memory  # noqa: F401   (gpu = device(f"cuda:{current_device()}") at import)
    _pkg("pipeline", REF + "/pipeline")
    import pipeline.causal_inference as ci
    torch.cuda.current_device = lambda: torch.device("cpu")
    torch.cuda.Stream = _NoCudaObject
    torch.cuda.Event = _NoCudaObject
    torch.Tensor.cuda = lambda self, *a, **k: self          # v2v.encode_video_latent calls frames.cuda()
    import release_server as rs
    return rs, ci.CausalInferencePipeline
