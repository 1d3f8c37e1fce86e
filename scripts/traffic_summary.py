"""Condense scripts/profile_bench.sh passes: kernel-time table + per-kernel-class HBM traffic.

FETCH_SIZE / WRITE_SIZE are rocprofv3 derived counters in KiB (L2 memory-side requests).  Following
MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
64 bytes, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is taken as reported (uncalibrated).  Infinity-cache hits are
counted as traffic too, so this is an upper bound on real HBM bytes.
usage: traffic_summary.py <raw dir> [summary dir]"""
import glob
import json
import os
import sqlite3
import sys

CLASSES = [
    ("gemm", ("gemm8_kernel<false", "gemm_kernel<false", "gemm5_kernel<false")),       # bf16 DiT projection GEMMs
    ("attn", ("attn_fwd_kernel", "attn_fwd_pp_kernel", "attn_fwd_w4_kernel")),
    ("conv", ("conv_igemm_kernel", "conv_halo", "gemm_kernel<true", "gemm8_kernel<true")),  # fp16: VAE
    ("layernorm", ("layernorm_modulate_kernel",)),
    ("rope", ("qk_norm_rope_cache_kernel",)),
]


def klass(name):
    for k, pats in CLASSES:
        if any(p in name for p in pats):
            return k
    return "other"


def db_of(root, sub):
    dbs = sorted(glob.glob(os.path.join(root, sub, "**", "*.db"), recursive=True))
    return sqlite3.connect(dbs[-1]).cursor() if dbs else None


def dbs_of(root, prefix):
    return [sqlite3.connect(d).cursor() for d in sorted(glob.glob(os.path.join(root, prefix + "*", "**", "*.db"), recursive=True))]


def counter_by_kernel(cur, counter):
    rows = cur.execute("select k.name, count(*), sum(p.counter_value) from pmc_events p join kernels k "
                       "on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by k.name", (counter,)).fetchall()
    return {n: (c, s) for n, c, s in rows}


def main():
    root = sys.argv[1]
    dest = sys.argv[2] if len(sys.argv) > 2 else root
    out = {"source": "rocprofv3 (scripts/profile_bench.sh)", "classes": {}, "kernels": []}
    cur = db_of(root, "trace")
    lines = []
    if cur:
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                           "from kernels group by name order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        lines.append(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
        for n, c, s, a, mn, mx in rows[:40]:
            lines.append(f"{n[:100]:100s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {100*s/total:6.2f}")
            out["kernels"].append({"name": n, "class": klass(n), "calls": c, "total_ms": s / 1e6, "avg_us": a / 1e3,
                                   "min_us": mn / 1e3, "max_us": mx / 1e3, "pct": 100 * s / total})
        for n, c, s, a, mn, mx in rows:
            k = out["classes"].setdefault(klass(n), {"launches": 0, "total_ms": 0.0})
            k["launches"] += c
            k["total_ms"] += s / 1e6
        for k in out["classes"].values():
            k["avg_launch_us"] = 1e3 * k["total_ms"] / max(k["launches"], 1)
    for sub, counter, key in (("fetch", "FETCH_SIZE", "fetch_kib"), ("write", "WRITE_SIZE", "write_kib")):
        for cur in dbs_of(root, sub):
            try:
                per = counter_by_kernel(cur, counter)
            except sqlite3.Error as e:
                lines.append(f"no {counter}: {e}")
                continue
            for n, (c, s) in per.items():
                k = out["classes"].setdefault(klass(n), {})
                k[key + "_sum"] = k.get(key + "_sum", 0.0) + s
                k[key + "_launches"] = k.get(key + "_launches", 0) + c
    for name, k in out["classes"].items():
        if "fetch_kib_sum" in k and "write_kib_sum" in k:
            rd = 2.0 * 1024 * k["fetch_kib_sum"] / max(k["fetch_kib_launches"], 1)   # x2: gfx950 correction
            wr = 1024.0 * k["write_kib_sum"] / max(k["write_kib_launches"], 1)
            k["hbm_read_bytes_per_launch"] = rd
            k["hbm_write_bytes_per_launch"] = wr
            k["hbm_bytes_per_launch"] = rd + wr
            k["launches_sampled"] = min(k["fetch_kib_launches"], k["write_kib_launches"])
            lines.append(f"traffic {name:10s} launches {k['fetch_kib_launches']:6d}  read {rd/1e6:10.2f} MB  write {wr/1e6:10.2f} MB per launch")
    txt = "\n".join(lines)
    print(txt)
    open(os.path.join(dest, "kernel_stats.txt"), "w").write(txt + "\n")
    json.dump(out, open(os.path.join(dest, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
