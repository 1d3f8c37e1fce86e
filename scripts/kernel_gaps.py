"""GPU idle time BETWEEN kernels of the bench command, from a rocprofv3 --kernel-trace database (scripts/profile_bench.sh leaves it
in /tmp/prof_<tag>/trace): per block, the sum of (start of kernel i+1 - end of kernel i) over consecutive dispatches on the stream -
what launch latency / dependency stalls cost, i.e. the ceiling of what hipGraph replay or kernel fusion could recover.
usage: kernel_gaps.py <dir with the rocpd .db>"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
cur = sqlite3.connect(dbs[-1]).cursor()
cols = [r[1] for r in cur.execute("PRAGMA table_info(kernels)").fetchall()]
print("kernels columns:", cols)
sc = next((c for c in cols if c in ("start", "start_timestamp", "begin")), None)
ec = next((c for c in cols if c in ("end", "end_timestamp", "stop")), None)
if not sc or not ec:
    raise SystemExit("no start / end columns")
rows = cur.execute(f"select name, {sc}, {ec} from kernels order by {sc}").fetchall()
# the generation itself: from the first to the last kernel of the library (the random weight initialisation in front is torch's)
idx = [i for i, r in enumerate(rows) if "rtv::" in r[0]]
rows = rows[idx[0]:idx[-1] + 1]
nblocks = max(1, sum(1 for r in rows if "pixels_to_rgb8" in r[0]))
print("blocks in the window:", nblocks)
span = (rows[-1][2] - rows[0][1]) / 1e6
busy = sum(r[2] - r[1] for r in rows) / 1e6
gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print(f"{len(rows)} kernels over {span:.1f} ms ({span / nblocks:.1f} ms per block, idle {sum(g for g in gaps if g > 0) / 1e3 / nblocks:.2f} ms per block): busy {busy:.1f} ms, idle between kernels {sum(pos) / 1e3:.2f} ms ({100 * sum(pos) / 1e3 / span:.2f} %), "
      f"overlapped {-sum(g for g in gaps if g < 0) / 1e3:.2f} ms")
import statistics
print(f"gap per boundary: median {statistics.median(pos):.2f} us, mean {statistics.mean(pos):.2f} us, p99 {sorted(pos)[int(0.99 * len(pos))]:.1f} us; "
      f"gaps > 20 us: {sum(1 for g in pos if g > 20)} totalling {sum(g for g in pos if g > 20) / 1e3:.2f} ms")
big = sorted(((g, rows[i][0][:60], rows[i + 1][0][:60]) for i, g in enumerate(gaps) if g > 50), reverse=True)[:12]
for g, a, b in big:
    print(f"  {g:8.1f} us between {a} -> {b}")
