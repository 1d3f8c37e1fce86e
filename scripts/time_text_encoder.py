"""Latency of the native UMT5-XXL encoder (random weights of the real architecture) per prompt length, and its error against
the float32 oracle at tiny dims.  usage: time_text_encoder.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd.text_encoder import WanTextEncoder  # noqa: E402

enc = WanTextEncoder(device="cuda").init_random_weights(seed=0)
print("weights GB:", round(sum(t.numel() * t.element_size() for t in enc._t.values()) / 1e9, 2))
for n in (20, 77, 256, 512):
    ids = torch.randint(2, 256384, (1, 512))
    mask = torch.zeros(1, 512, dtype=torch.long)
    mask[0, :n] = 1
    for _ in range(2):
        out = enc.encode_ids(ids, mask)["prompt_embeds"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = enc.encode_ids(ids, mask)["prompt_embeds"]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    rows = (n + 31) // 32 * 32
    flop = 24 * 2 * rows * (4 * 4096 * 4096 + 3 * 4096 * 10240)
    print(f"{n:4d} tokens: {ms:7.2f} ms  ({flop / ms / 1e9:6.0f} TF/s on the linears), finite={bool(torch.isfinite(out).all())}, "
          f"|out| mean {float(out[0, :n].abs().mean()):.3f}")
