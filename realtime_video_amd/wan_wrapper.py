"""Mirror of the reference's `WanDiffusionWrapper` (utils/wan_wrapper.py:121-323) for the causal
inference path: `forward(noisy_image_or_video[B,F,16,h,w], conditional_dict, timestep[B,F], kv_cache,
crossattn_cache, current_start)` -> `(flow_pred, pred_x0)`, flow->x0 conversion in float64
(:181-205), scheduler binding (:148-151, :303-323)."""
import torch

from . import ops
from .causal_model import CausalWanModel
from .scheduler import FlowMatchScheduler


# The two architectures the reference's server loads (release_server.py:162-165), dims from wan/configs/wan_t2v_14B.py:21-25 and
# wan_t2v_1_3B.py:21-25 (the reference reads them from <MODEL_FOLDER>/<model_name>/config.json through diffusers' from_pretrained,
# utils/wan_wrapper.py:135-139; a config.json found there is honoured here as well).
MODEL_ARCHS = {
    "Wan2.1-T2V-14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
    "Wan2.1-T2V-1.3B": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30),
}


def _arch_for(model_name):
    import json
    import os
    path = os.path.join(os.getenv("MODEL_FOLDER", "Wan-2.1"), model_name, "config.json")      # settings.py:5
    if os.path.isfile(path):
        with open(path) as f:
            cfg = json.load(f)
        keys = ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim", "num_heads",
                "num_layers", "qk_norm", "cross_attn_norm", "eps")
        return {k: cfg[k] for k in keys if k in cfg}
    if model_name not in MODEL_ARCHS:
        raise ValueError(f"unknown model_name {model_name!r} (known: {sorted(MODEL_ARCHS)}; or put a config.json under "
                         f"$MODEL_FOLDER/{model_name}/)")
    return dict(MODEL_ARCHS[model_name])


class WanDiffusionWrapper:
    """Two constructor forms:
      * the reference's (utils/wan_wrapper.py:121-151; release_server.py:167): `WanDiffusionWrapper(model_name="Wan2.1-T2V-14B",
        timestep_shift=..., is_causal=True[, local_attn_size, sink_size])` builds an EMPTY native CausalWanModel of that
        architecture on `device`; `load_state_dict(state_dict)` then takes the checkpoint with its `model.`-prefixed keys, and
        `.to(dtype=torch.bfloat16)`, `.eval()`, `.requires_grad_(False)`, `.to(device)` and `blocks[i].self_attn.fuse_projections()`
        (release_server.py:168-177) are accepted no-ops: weights are converted and q / k / v fused while loading;
      * `WanDiffusionWrapper(model, ...)` around an existing native CausalWanModel."""

    def __init__(self, model=None, timestep_shift=8.0, is_causal=True, local_attn_size=-1, sink_size=0, meta_init=False,
                 model_name=None, device="cuda"):
        if not is_causal:
            raise NotImplementedError("only the causal model is on the hot path")
        if isinstance(model, str):
            model, model_name = None, model
        if model is None:
            if model_name is None:
                raise ValueError("pass a CausalWanModel or model_name=")
            model = CausalWanModel(local_attn_size=local_attn_size, sink_size=sink_size,
                                   device="meta" if meta_init else device, **_arch_for(model_name))
        self.model = model
        self.uniform_timestep = not is_causal
        self.scheduler = FlowMatchScheduler(shift=timestep_shift, sigma_min=0.0, extra_one_step=True)
        self.scheduler.set_timesteps(1000, training=True)
        self.seq_len = 32760

    # ---- the nn.Module surface release_server.load_transformer (:150-187) touches
    def load_state_dict(self, state_dict, strict=True, assign=False):
        """The wrapper's state dict is its model's with `model.` in front of every key (the reference wrapper holds the
        CausalWanModel as `self.model` and has no parameters of its own)."""
        if strict:
            stray = [k for k in state_dict if not k.startswith("model.")]
            if stray and len(stray) != len(state_dict):      # (all keys unprefixed = the inner model's own state dict: accepted)
                raise RuntimeError("Error(s) in loading state_dict for WanDiffusionWrapper:\n\tUnexpected key(s) in state_dict: "
                                   + ", ".join(repr(k) for k in stray[:12]) + ".")
        return self.model.load_state_dict(state_dict, strict=strict)

    def to(self, *args, **kwargs):
        self.model.to(*args, **kwargs)
        return self

    def requires_grad_(self, requires_grad=False):
        if requires_grad:
            raise NotImplementedError("inference only")
        return self

    def eval(self):
        return self

    def get_scheduler(self):
        return self.scheduler

    @staticmethod
    def _kernel_ok(*tensors):
        """The one-launch scheduler kernel covers the inference case: bf16 [F, C, h, w] latents on the GPU."""
        return all(t.is_cuda and t.dtype == torch.bfloat16 and t.ndim == 4 for t in tensors)

    def _convert_flow_pred_to_x0(self, flow_pred, xt, timestep):
        sch = self.scheduler.to(flow_pred.device)
        if self._kernel_ok(flow_pred, xt) and timestep.dtype in ops._T_KIND:
            return ops.scheduler_step(sch.timesteps, sch.sigmas, flow=flow_pred, xt=xt,
                                      t=timestep.to(flow_pred.device).contiguous())[0]
        dt = flow_pred.dtype
        fp, x, sig, ts = flow_pred.double(), xt.double(), sch.sigmas.double(), sch.timesteps.double()
        idx = torch.argmin((ts.unsqueeze(0) - timestep.to(flow_pred.device).unsqueeze(1)).abs(), dim=1)
        return (x - sig[idx].reshape(-1, 1, 1, 1) * fp).to(dt)

    def forward(self, noisy_image_or_video, conditional_dict, timestep, kv_cache=None, crossattn_cache=None,
                current_start=None, cache_start=None, renoise=None, kv_cache_only=False, **unused):
        """`renoise=(noise[B*F,16,h,w], next_timestep[B*F])` (not in the reference signature): also return
        scheduler.add_noise(x0, noise, next_timestep) as a third value, computed in the same launch as x0 - the
        denoising loops of release_server.py:669-694 / causal_inference.py:187-212 call the two back to back.
        `kv_cache_only=True` (not in the reference signature either): the caller discards the output and only wants the K / V
        cache filled (the session's recompute pass); the model may stop behind the last layer's cache write."""
        if kv_cache is None:
            raise NotImplementedError("non-cached (training / bidirectional) forwards are out of scope")
        flow = self.model(noisy_image_or_video.permute(0, 2, 1, 3, 4), t=timestep,
                          context=conditional_dict["prompt_embeds"], seq_len=self.seq_len, kv_cache=kv_cache,
                          crossattn_cache=crossattn_cache, current_start=current_start,
                          cache_start=cache_start, kv_cache_only=kv_cache_only).permute(0, 2, 1, 3, 4)
        fl, xt, ts = flow.flatten(0, 1), noisy_image_or_video.flatten(0, 1), timestep.flatten(0, 1)
        if renoise is None:
            return flow, self._convert_flow_pred_to_x0(fl, xt, ts).unflatten(0, flow.shape[:2])
        noise, t_next = renoise
        sch = self.scheduler.to(flow.device)
        if self._kernel_ok(fl, xt, noise) and ts.dtype in ops._T_KIND and t_next.dtype == ts.dtype:
            x0, noisy = ops.scheduler_step(sch.timesteps, sch.sigmas, flow=fl, xt=xt, t=ts.contiguous(), noise=noise,
                                           t_next=t_next.to(flow.device).contiguous())
        else:
            x0 = self._convert_flow_pred_to_x0(fl, xt, ts)
            noisy = sch.add_noise(x0, noise, t_next)
        return flow, x0.unflatten(0, flow.shape[:2]), noisy.unflatten(0, flow.shape[:2])

    __call__ = forward
