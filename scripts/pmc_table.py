"""Condense scripts/pmc_hot_kernels.sh: one row per hot kernel with its MFMA utilisation and memory-side bandwidth.

Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the SIMDs that the sampled SQ
instance sees; GRBM_GUI_ACTIVE = kernel duration in shader cycles.  MFMA utilisation = sum over all instances of
MFMA_BUSY / (GRBM_GUI_ACTIVE averaged over its per-XCD instances x 256 CUs x 4 SIMDs).  FETCH_SIZE (KiB) x 2 = read bytes on gfx950, WRITE_SIZE (KiB) as is."""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
ROCM_KERNELS = {"gemm": "gemm%_kernel", "attn": "attn_fwd%_kernel", "conv": "conv_%_kernel", "layernorm": "layernorm_modulate_kernel",
                "rope": "qk_norm_rope_cache_kernel"}   # SQL LIKE patterns (conv: conv_halo_kernel / conv_igemm_kernel, attn: lockstep / four-phase)


def counters(k):
    out, dur = {}, None
    for db in sorted(glob.glob(os.path.join(root, f"{k}_s*", "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(db).cursor()
        like = f"%{ROCM_KERNELS[k]}%"
        n_disp = cur.execute("select count(*) from kernels where name like ?", (like,)).fetchone()[0]
        d = cur.execute("select avg(duration) from kernels where name like ?", (like,)).fetchone()[0]
        try:
            rows = cur.execute("select p.counter_name, sum(p.counter_value), count(*) from pmc_events p join kernels k on "
                               "k.dispatch_id = p.dispatch_id where k.name like ? group by p.counter_name", (like,)).fetchall()
        except sqlite3.Error:
            rows = []
        for name, s, cnt in rows:
            out[name] = s / max(n_disp, 1)          # per launch, summed over all counter instances
            out[name + ":instances"] = cnt / max(n_disp, 1)
        if d and "GRBM_GUI_ACTIVE" not in {r[0] for r in rows}:
            dur = d if dur is None else min(dur, d)
    return out, dur


print(f"{'kernel':10s} {'us/launch':>10s} {'clock MHz':>10s} {'MFMA util':>10s} {'wave stall':>11s} {'LDS busy':>9s} {'bank confl':>10s} "
      f"{'L2 hit':>7s} {'read GB/s':>10s} {'write GB/s':>10s}")
for k in ("gemm", "attn", "conv", "layernorm", "rope"):
    c, dur_ns = counters(k)
    if not c:
        print(f"{k:10s} (no data)")
        continue
    us = (dur_ns or 0) / 1e3
    n_gui = 1.0
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / max(c.get("GRBM_GUI_ACTIVE:instances", 1.0), 1.0)   # one value per XCD: average
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = mfma / (gui * 256 * 4) if gui else float("nan")
    wave = c.get("SQ_WAVE_CYCLES", 0.0)
    stall = c.get("SQ_WAIT_ANY", 0.0) / wave if wave else float("nan")
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / (gui * 256) if gui else float("nan")
    hit = c.get("TCC_HIT_sum", 0.0)
    miss = c.get("TCC_MISS_sum", 0.0)
    rd = 2 * 1024 * c.get("FETCH_SIZE", 0.0)
    wr = 1024 * c.get("WRITE_SIZE", 0.0)
    clock = gui / us if us else float("nan")
    print(f"{k:10s} {us:10.1f} {clock:10.0f} {100 * util:9.1f}% {100 * stall:10.1f}% {100 * lds:8.1f}% {c.get('SQ_LDS_BANK_CONFLICT', 0.0):10.0f} "
          f"{100 * hit / max(hit + miss, 1):6.1f}% {rd / us / 1e3 if us else 0:10.0f} {wr / us / 1e3 if us else 0:10.0f}")
print("(us/launch from the un-instrumented FETCH/WRITE passes; utilisations are relative to GRBM_GUI_ACTIVE of the SQ pass)")
