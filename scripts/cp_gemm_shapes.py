"""GEMM time at the row counts of N-way context parallelism (4680 / N local rows) for the four projection shapes of the
14B layer, per tile config, with hipBLASLt (through torch) on the same operands as the library reference.  The variants
of one shape are timed INTERLEAVED over several rounds (boxes / clocks drift by 10 % between back-to-back measurements) and
the median is reported.  usage: cp_gemm_shapes.py [cfg ...]   (env CP_M=4680,2340,... CP_TORCH=0/1 CP_ROUNDS=5)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

cfgs = [int(c) for c in sys.argv[1:]] or [0, 1, 4]
rounds = int(os.environ.get("CP_ROUNDS", "5"))
ops.ensure_gemm_workspace(torch.device("cuda"))
shapes = [("qkv", 15360, 5120), ("o/cq/co", 5120, 5120), ("ffn0", 13824, 5120), ("ffn2", 5120, 13824)]


def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for m in [int(x) for x in os.environ.get("CP_M", "4680,2340,1170,585").split(",")]:
    for name, n, k in shapes:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device="cuda").to(torch.bfloat16)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        # cfg >= 1000: the same tile config with the idle-wave skipping of gemm8 switched off (A/B); cfg 3000 + c: tile config c with
        # the half-tile tail of gemm8 switched off (K-segment tail instead)
        lib = __import__("realtime_video_amd._lib", fromlist=["load"]).load()

        def variant(c):
            def run():
                lib.rtv_gemm_set_skip_idle(0 if 1000 <= c < 2000 else 1)
                lib.rtv_gemm_set_half_tail(0 if 3000 <= c < 4000 else 1)
                lib.rtv_gemm_set_ragged_strips(0 if c >= 4000 else 1)     # cfg 4000 + c: tile config c without the ragged-row strips
                ops.gemm(a, w, bias=b, out=out, tile_cfg=c % 1000)
                lib.rtv_gemm_set_half_tail(1)
            return run
        fns = {f"cfg{c}": variant(c) for c in cfgs}
        if os.environ.get("CP_TORCH", "1") == "1":
            fns["hipBLASLt"] = lambda: torch.nn.functional.linear(a, w, b)
        for f in fns.values():
            for _ in range(3):
                f()
        times = {key: [] for key in fns}
        for _ in range(rounds):
            for key, f in fns.items():
                times[key].append(timed(f))
        line = f"M={m:5d} {name:8s} N={n:5d} K={k:5d}:"
        for key, t in times.items():
            ms = statistics.median(t)
            line += f"  {key}: {ms * 1e3:6.0f} us {2.0 * m * n * k / ms / 1e9:5.0f} TF/s"
        print(line, flush=True)
