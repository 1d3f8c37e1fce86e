"""Time the first-frame re-encode (VAEEncoderWrapper on one 480x832 frame, fresh cache) in isolation."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402
from realtime_video_amd.vae_encoder import VAEEncoderWrapper, encode_video_latent  # noqa: E402

enc = VAEEncoderWrapper(device="cuda").init_random_weights()
frame = (torch.rand(1, 3, 480, 832, device="cuda") * 2 - 1).half()
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.prof_reset()
    ops.prof_enable(True)
    lat, _ = encode_video_latent(enc, [None] * 55, frames=frame, height=480, width=832)
    torch.cuda.synchronize()
    ops.prof_enable(False)
    dt = (time.perf_counter() - t0) * 1e3
    print(f"encode 1 frame: wall {dt:.2f} ms; conv {ops.prof_read('conv')['ms']:.2f} ms "
          f"({ops.prof_read('conv')['work'] / max(ops.prof_read('conv')['ms'], 1e-9) / 1e9:.0f} TF/s), "
          f"norm {ops.prof_read('layernorm')['ms']:.2f} ms", flush=True)
