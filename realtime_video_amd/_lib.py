"""ctypes binding of librtv_hip.so (the C-ABI HIP kernel library, see include/rtv_hip.h).

The product path has no CPU / eager fallback: if the library is missing or a kernel reports an
error, a RuntimeError is raised (the reference's attention()/pipeline API reports errors as Python
exceptions, wan/modules/attention.py:72-73,129).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTV_LIB_PATH: A/B measurements against another build of the same C ABI (scripts/ab_build.sh); never set in production
LIB_PATH = os.environ.get("RTV_LIB_PATH") or os.path.join(_HERE, "librtv_hip.so")
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 103   # include/rtv_hip.h RTV_ABI_VERSION this binding's structs / signatures mirror

_lib = None

c_int, c_i64, c_f32, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (restype is always int except where noted); mirrors include/rtv_hip.h
SIGNATURES = {
    "rtv_version": [],
    "rtv_quantize_fp8": [c_vp, c_i64, c_int, c_int, c_vp, c_i64, c_vp, c_vp, c_vp],
    "rtv_gemm_fp8": [c_vp, c_int, c_vp, c_int, c_vp, c_f32, c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int,
                     c_int, c_int, c_vp, c_int, c_vp],
    "rtv_prof_enable": [c_int],
    "rtv_prof_read": [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64),
                      ctypes.POINTER(ctypes.c_double)],
    "rtv_prof_reset": [],
    "rtv_prof_set_stride": [c_int, c_int],
    "rtv_prof_read_seen": [c_int, ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double)],
    "rtv_prof_bracket_overhead": [c_int, c_vp, ctypes.POINTER(ctypes.c_double)],
    "rtv_attn_fwd": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int,
                     c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                     c_f32, c_int, c_int, c_int, c_vp],
    "rtv_attn_fwd_win": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                         c_f32, c_int, c_int, c_int, c_vp],
    "rtv_attn_fwd_dup": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int,
                         c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                         c_f32, c_int, c_int, c_int, c_vp],
    "rtv_attn_fwd_split": [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                           c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                           c_f32, c_int, c_int, c_int, c_vp, ctypes.c_size_t, c_int, c_vp],
    "rtv_attn_set_waves": [c_int],             # include/rtv_hip_lab.h (variant switches for tests / scripts)
    "rtv_attn_set_skip_idle": [c_int],
    "rtv_gemm_set_half_tail": [c_int],
    "rtv_gemm_set_skip_idle": [c_int],
    "rtv_gemm_set_ragged_strips": [c_int],
    "rtv_lab_build": [],
    "rtv_rope_set_wave": [c_int],
    "rtv_gemm": [c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int,
                 c_vp, c_int, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_vp],
    "rtv_layernorm_modulate": [c_vp, c_vp, c_int, c_int, c_f32, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "rtv_rmsnorm": [c_vp, c_int, c_vp, c_int, c_int, c_int, c_f32, c_vp, c_vp],
    "rtv_qk_norm_rope_cache": [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_f32,
                               c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp],
    "rtv_qk_norm_rope_cache_ring": [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_f32,
                                    c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp],
    "rtv_modulation_table": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp],
    "rtv_sinusoidal_embedding": [c_vp, c_vp, c_int, c_int, c_vp],
    "rtv_patchify": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "rtv_unpatchify": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp],
    "rtv_scheduler_step": [c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_int, c_vp, c_vp, c_int,
                           c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "rtv_pixels_to_rgb8": [c_vp, c_vp, c_int, c_int, c_int, c_vp],
    "rtv_gemm_set_workspace": [c_vp, ctypes.c_size_t],
    "rtv_gemm_set_stream_workspace": [c_vp, c_vp, ctypes.c_size_t],
    "rtv_probe_mfma": [c_vp, c_vp, c_vp, c_vp],
    "rtv_probe_tr": [c_vp, c_vp, c_int, c_int, c_vp],
}
# later sections (DiT forward, VAE) register their signatures here as well
EXTRA_SIGNATURES = {}


def build(verbose=False):
    """Compile librtv_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(os.cpu_count() or 4)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise RuntimeError("building librtv_hip.so failed")
    return LIB_PATH


def declared_symbols(lab=True):
    """Every extern "C" function declared in include/rtv_hip.h - the drop-in boundary - and, with `lab`, in
    include/rtv_hip_lab.h (test / measurement hooks, not part of the boundary), parsed from the headers."""
    import re
    out = set()
    for name in ("rtv_hip.h",) + (("rtv_hip_lab.h",) if lab else ()):
        text = open(os.path.join(os.path.dirname(_HERE), "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out.update(re.findall(r"\b(rtv_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def load():
    """Load the library (no compute happens here; safe on a CPU-only box)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C realtime_video_amd/csrc`). There is no CPU fallback for the HIP path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.rtv_version.restype, lib.rtv_version.argtypes = c_int, []
    have = lib.rtv_version()
    if have != ABI_VERSION:
        # struct layouts (rtv_dit_config, rtv_dit_step, ...) are part of the ABI: a library of another revision would read
        # garbage for fields it does not know.  RTV_LIB_PATH (A/B against an older build) is no exception.
        raise RuntimeError(f"{LIB_PATH} reports ABI revision {have}, this binding is written against {ABI_VERSION} "
                           "(include/rtv_hip.h RTV_ABI_VERSION): rebuild the library (`make -C realtime_video_amd/csrc`)")
    lib.rtv_last_error.restype = ctypes.c_char_p
    lib.rtv_last_error.argtypes = []
    sigs = dict(SIGNATURES)
    sigs.update(EXTRA_SIGNATURES)
    for name, argtypes in sigs.items():
        fn = getattr(lib, name, None)
        if fn is None:
            # RTV_LIB_PATH = an older build of the same C ABI (A/B measurements): entry points added since are simply absent
            if name in SIGNATURES and not os.environ.get("RTV_LIB_PATH"):
                raise RuntimeError(f"librtv_hip.so does not export {name}")
            continue
        fn.argtypes = argtypes
        fn.restype = c_int
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().rtv_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {status}): {msg}")


def call(name, *args):
    lib = load()
    fn = getattr(lib, name)
    if fn.argtypes is None and name in EXTRA_SIGNATURES:   # registered by a module imported after load()
        fn.argtypes = EXTRA_SIGNATURES[name]
        fn.restype = c_int
    check(fn(*args), name)
