"""realtime_video_amd — MI355X (gfx950) native hot path for krea-ai/realtime-video.

(The repository-level name is `realtime-video_amd`; hyphens are not importable, so the Python
package is spelled with an underscore and `realtime-video_amd` is a symlink to it.)

Hot path = per-denoising-step causal Wan DiT forward with rolling KV cache + streaming VAE decode,
implemented as hand-written HIP kernels in csrc/ behind the C ABI of include/rtv_hip.h, with a thin
Python host side mirroring the reference's attention-backend / pipeline API.
"""
__version__ = "0.1.0"
