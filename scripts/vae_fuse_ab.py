"""Streaming VAE decode of one 3-latent-frame block at 832x480 with the conv + RMS_norm + SiLU fusion on / off (interleaved)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import _lib  # noqa: E402
from realtime_video_amd.vae_decoder import VAEDecoderWrapper  # noqa: E402

dev = "cuda"
dec = VAEDecoderWrapper(dev).init_random_weights(seed=1)
g = torch.Generator().manual_seed(0)
z = torch.randn(1, 3, 16, 60, 104, generator=g).half().to(dev)
cache = [None] * 55
_, cache = dec(z, *cache)
t = {1: [], 0: []}
for _ in range(7):
    for fuse in (1, 0):
        _lib.call("rtv_conv_set_fuse_norm", fuse)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _, cache = dec(z, *cache)
        e1.record()
        torch.cuda.synchronize()
        t[fuse].append(e0.elapsed_time(e1))
_lib.call("rtv_conv_set_fuse_norm", 1)
print(f"VAE decode of a 12-frame block: fused norm epilogue {statistics.median(t[1]):.2f} ms, separate pass {statistics.median(t[0]):.2f} ms")
