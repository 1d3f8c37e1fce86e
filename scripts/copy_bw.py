"""What does this chip sustain for a plain read-once / write-once stream of the size of a DiT row kernel?  torch's copy kernel over
[rows, 5120] bf16 tensors (rotating over > 256 MB so every launch goes to HBM), for 4680 rows (48 MB in + 48 MB out - the LayerNorm
kernel's traffic) and for 8x as many rows: the practical ceiling the row kernels' hbm_frac should be read against."""
import torch

for rows in (4680, 4 * 4680, 16 * 4680):
    nb = max(2, int(1.2e9 // (rows * 5120 * 2 * 2)))
    xs = [torch.randn(rows, 5120, device="cuda").to(torch.bfloat16) for _ in range(nb)]
    os_ = [torch.empty_like(x) for x in xs]
    for x, o in zip(xs, os_):
        o.copy_(x)
    torch.cuda.synchronize()
    iters = 40
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        os_[i % nb].copy_(xs[i % nb])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    byt = 2.0 * rows * 5120 * 2
    print(f"copy {rows:6d} x 5120 bf16 ({byt / 1e6:6.1f} MB moved): {us:7.1f} us  {byt / us / 1e6:5.2f} TB/s  frac of 8 TB/s {byt / us / 8e6:.2f}")
    del xs, os_
