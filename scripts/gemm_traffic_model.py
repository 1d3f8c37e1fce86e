"""Where the L2-fabric traffic of the projection GEMMs goes ("976 MB per launch vs 262.6 MB algorithmic =
3.7x; 2.5x is the floor with eight private L2s - explain the other 1.2x").

MODEL (exact replay of gemm8's block id -> tile map on the host: xcd_remap, the 8-row supertile order, the ragged-row strips, the
half-tile / split-K tail of plan_split_k): the 256 workgroups of a round run on 8 XCDs x 32 CUs; the tiles of one XCD-round move
along K in lockstep and share operand panels through that XCD's 4 MiB L2, so an XCD-round reads every DISTINCT A panel
(256 rows x K) and W panel (256 columns x K) it touches exactly once from the fabric; nothing survives into the next round (a
round streams 12+ panels of 2.6 MB through a 4 MiB L2).  Output: one write per element (+ the residual read where fused), split-K
slabs written through and read once by the reducer.  "floor" = the same with the ideal 32-tile shape for r*c = 32 (r + c >= 11.3
panels) and no tile quantisation.

With a rocprofv3 directory (scripts/gemm_traffic_table.sh) the measured FETCH_SIZE x 2 / WRITE_SIZE / TCC hit rate per shape are
printed beside the model.   usage: gemm_traffic_model.py [pmc_dir]"""
import glob
import math
import os
import sqlite3
import sys

G, XCDS, BM, BN = 256, 8, 256, 256
SHAPES = [("qk", 10240, 5120, 0), ("v / o / cross-q / cross-o", 5120, 5120, 1), ("ffn-in", 13824, 5120, 0), ("ffn-out", 5120, 13824, 1)]
M = 4680


def xcd_remap(bid, nwg):
    xcd, slot = bid % XCDS, bid // XCDS
    q, r = nwg // XCDS, nwg % XCDS
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + slot


def tile_mn(tile_id, tiles_m, tiles_n, group_m=8):
    per_group = group_m * tiles_n
    g = tile_id // per_group
    first_m = g * group_m
    gm = min(tiles_m - first_m, group_m)
    ig = tile_id - g * per_group
    return first_m + ig % gm, ig // gm


def plan(N, K):
    """-> list of workgroups in block-id order: (rows as (m_tile, half or None), n tiles tuple, k fraction (lo, hi))."""
    tiles_m, tiles_n, nk = math.ceil(M / BM), math.ceil(N / BN), K // 64
    last_rows = M - (tiles_m - 1) * BM
    T0 = tiles_m * tiles_n
    wgs, pair_units, pair_pad = [], 0, 0
    if tiles_m >= 2 and last_rows <= 128 and T0 > G:
        pu, RA = (tiles_n + 1) // 2, T0 % G
        pp = (pu + 7) & ~7
        if 0 < RA <= tiles_n - pp:
            pair_units, pair_pad, tiles_m = pu, pp, tiles_m - 1
    T = tiles_m * tiles_n
    R = (T + pair_pad) % G if (T + pair_pad) % G <= T else 0
    S, half, split_first = 1, False, False
    if R > 0 and T > R and G // R == 2 and nk <= 128:
        half = True
    elif R > 0:
        S = min(8, G // R, nk // 4)
        if S < 2:
            S = 1
        split_first = S > 1 and R * S * 4 <= G
    first_unit = T - R if (half or S > 1) else T
    for u in range(pair_pad):                               # ragged-row strips: 128 x 512, ids [0, pair_units), then padding ids
        if u < pair_units:
            wgs.append(((tiles_m, 0), (2 * u, min(2 * u + 1, tiles_n - 1)), (0.0, 1.0)))
        else:
            wgs.append(None)
    body = []
    for bid in range(first_unit):
        m, n = tile_mn(xcd_remap(bid, first_unit), tiles_m, tiles_n)
        body.append(((m, None), (n,), (0.0, 1.0)))
    tail = []
    nu = R * (2 if half else S) if first_unit < T else 0
    for b in range(nu):
        v = xcd_remap(b, nu)
        if half:                                            # two halves of a tile side by side on one XCD
            tl, h = v // 2, v % 2
            m, n = tile_mn(first_unit + tl, tiles_m, tiles_n)
            tail.append(((m, h), (n,), (0.0, 1.0)))
        else:
            seg, tl = v // R, v % R
            m, n = tile_mn(first_unit + tl, tiles_m, tiles_n)
            tail.append(((m, None), (n,), (seg / S, (seg + 1) / S)))
    wgs += (tail + body) if split_first else (body + tail)
    return wgs, dict(tiles_m=tiles_m, tiles_n=tiles_n, T=T, R=R, S=S, half=half, strips=pair_units, split_units=nu if S > 1 else 0,
                     rounds=len(wgs) / G)


def model(N, K, res):
    wgs, info = plan(N, K)
    a_bytes = w_bytes = 0.0
    for r0 in range(0, len(wgs), G):
        per_xcd = [dict(a={}, w={}) for _ in range(XCDS)]
        for i, wg in enumerate(wgs[r0:r0 + G]):
            if wg is None:
                continue
            (m, h), ns, (k0, k1) = wg
            x = per_xcd[(r0 + i) % XCDS]
            rows = min(BM, M - m * BM)
            if h is not None:
                rows = max(0, min(128, rows - 128 * h))
            key = (m, h, k0)
            x["a"][key] = rows * K * (k1 - k0) * 2
            for n in ns:
                x["w"][(n, k0)] = min(BN, N - n * BN) * K * (k1 - k0) * 2
        a_bytes += sum(sum(x["a"].values()) for x in per_xcd)
        w_bytes += sum(sum(x["w"].values()) for x in per_xcd)
    c_bytes = M * N * 2
    slab = info["split_units"] * BM * BN * 4
    reads = a_bytes + w_bytes + res * c_bytes + slab
    writes = c_bytes + slab
    alg = 2 * (M * K + N * K + M * N + res * M * N)
    # floor: T real tiles' worth of work in 32-tile XCD-rounds of the ideal aspect (r + c = 2 sqrt(32)), no quantisation
    xr = (M / BM) * (N / BN) / 32
    floor = xr * 2 * math.sqrt(32) * BM * K * 2 + c_bytes * (1 + res) + c_bytes
    return dict(info, alg=alg, reads=reads, writes=writes, a=a_bytes, w=w_bytes, slab=slab, floor=floor)


def measured(root, tag):
    out = {}
    for sub, names in (("fetch", ("FETCH_SIZE",)), ("write", ("WRITE_SIZE",)), ("tcc", ("TCC_HIT_sum", "TCC_MISS_sum"))):
        for db in glob.glob(os.path.join(root, f"{sub}_{tag}", "**", "*.db"), recursive=True):
            cur = sqlite3.connect(db).cursor()
            n = cur.execute("select count(*) from kernels where name like '%gemm%_kernel%'").fetchone()[0]
            out["us"] = cur.execute("select avg(duration) from kernels where name like '%gemm%_kernel%'").fetchone()[0] / 1e3
            for c in names:
                s = cur.execute("select sum(p.counter_value) from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                                "where p.counter_name = ? and k.name like '%gemm%_kernel%'", (c,)).fetchone()[0]
                out[c] = (s or 0.0) / max(n, 1)
    return out


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else None
    print(f"M = {M} rows, bf16, default dispatch (gemm8_kernel); MB per launch; alg = operands + output once")
    print(f"{'shape':28s} {'N x K':>13s} {'rounds':>6s} {'tail':>16s} {'alg':>6s} {'floor':>6s} {'model rd':>8s} {'(A':>6s} {'W':>6s} {'slab)':>6s} "
          f"{'model wr':>8s} {'model/alg':>9s}" + (f" {'meas rd':>8s} {'meas wr':>8s} {'meas/alg':>8s} {'L2 hit':>7s} {'us':>6s}" if root else ""))
    tot = dict(alg=0.0, model=0.0, meas=0.0, floor=0.0, n=0)
    per_shape = []
    for i, (name, N, K, res) in enumerate(SHAPES):
        m = model(N, K, res)
        weight = 4 if i == 1 else 1           # v, o, cross-q, cross-o: four launches of that shape per layer (seven GEMM launches in all)
        tail = "half tiles" if m["half"] else (f"split-K {m['S']} x {m['R']}" if m["S"] > 1 else ("strips" if m["strips"] else "-"))
        line = (f"{name:28s} {N:6d}x{K:<6d} {m['rounds']:6.2f} {tail:>16s} {m['alg'] / 1e6:6.0f} {m['floor'] / 1e6:6.0f} {m['reads'] / 1e6:8.0f} "
                f"{m['a'] / 1e6:6.0f} {m['w'] / 1e6:6.0f} {m['slab'] / 1e6:6.0f} {m['writes'] / 1e6:8.0f} {(m['reads'] + m['writes']) / m['alg']:9.2f}")
        if root:
            z = measured(root, f"g{i + 1}")
            if z:
                rd, wr = 2 * 1024 * z.get("FETCH_SIZE", 0.0), 1024 * z.get("WRITE_SIZE", 0.0)
                hit = z.get("TCC_HIT_sum", 0.0) / max(z.get("TCC_HIT_sum", 0.0) + z.get("TCC_MISS_sum", 0.0), 1.0)
                line += f" {rd / 1e6:8.0f} {wr / 1e6:8.0f} {(rd + wr) / m['alg']:8.2f} {100 * hit:6.1f}% {z.get('us', 0):6.0f}"
                tot["meas"] += weight * (rd + wr)
                per_shape.append({"shape": name, "N": N, "K": K, "launches_per_layer": weight, "algorithmic_bytes": m["alg"],
                                  "model_bytes": m["reads"] + m["writes"], "read_bytes": rd, "write_bytes": wr, "l2_hit": hit,
                                  "us_under_profiler": z.get("us", 0.0)})
        tot["n"] += weight
        tot["alg"] += weight * m["alg"]
        tot["model"] += weight * (m["reads"] + m["writes"])
        tot["floor"] += weight * m["floor"]
        print(line)
    if root and per_shape and os.environ.get("TRAFFIC_JSON"):
        import json
        with open(os.environ["TRAFFIC_JSON"], "w") as f:      # the file bench.py's roofline.traffic reads (classes.gemm.*)
            json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, own passes, 30 launches per shape and "
                                 "pass (scripts/gemm_traffic_table.sh); read bytes = 2 x FETCH_SIZE (gfx950), write bytes = WRITE_SIZE",
                       "classes": {"gemm": {"hbm_bytes_per_launch": tot["meas"] / tot["n"], "launches_sampled": 30 * len(per_shape) * 2,
                                            "algorithmic_bytes_per_launch": tot["alg"] / tot["n"], "model_bytes_per_launch": tot["model"] / tot["n"],
                                            "floor_bytes_per_launch": tot["floor"] / tot["n"], "per_shape": per_shape,
                                            "note": "average over the seven projection launches of a layer (qk, v, o, cross-q, cross-o, ffn-in, "
                                                    "ffn-out); the 5120 x 5120 shape was measured with the fused residual and weighted x4"}}}, f, indent=1)
    print(f"layer average (seven launches: shape 2 weighted x4): alg {tot['alg'] / tot['n'] / 1e6:.0f} MB, floor {tot['floor'] / tot['alg']:.2f}x, model {tot['model'] / tot['alg']:.2f}x"
          + (f", measured {tot['meas'] / tot['alg']:.2f}x" if tot["meas"] else ""))


if __name__ == "__main__":
    main()
