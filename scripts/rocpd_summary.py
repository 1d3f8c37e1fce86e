"""Summarise rocprofv3 rocpd SQLite results (kernel stats and, if present, PMC counters) as text/JSON.
Usage: python scripts/rocpd_summary.py <results.db> [out.json]
       python scripts/rocpd_summary.py --pmc-only <dir>      (all *.db below dir: per-kernel counter averages)"""
import glob
import json
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void rtv::", "").replace("rtv::", "")
    return name[:110]


def pmc_rows(cur, like="%"):
    return cur.execute("select k.name, p.counter_name, count(*), sum(p.counter_value), avg(p.counter_value) "
                       "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where k.name like ? "
                       "group by k.name, p.counter_name order by k.name", (like,)).fetchall()


def pmc_only(root):
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(db).cursor()
        tag = os.path.relpath(db, root).split(os.sep)[0]
        dur = dict(cur.execute("select name, avg(duration) from kernels group by name").fetchall())
        try:
            rows = pmc_rows(cur, "%rtv::%")
        except sqlite3.Error as e:
            print(tag, "no pmc", e)
            continue
        for n, cn, c, s, a in rows:
            print(f"{tag:12s} {short(n):60s} {cn:28s} n={c:3d} avg={a:16.1f}  kernel_avg_us={dur.get(n, 0)/1e3:9.1f}")


def main():
    if sys.argv[1] == "--pmc-only":
        return pmc_only(sys.argv[2])
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = {"kernels": [], "pmc": []}
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for n, c, s, a, mn, mx in rows[:40]:
        print(f"{short(n):110s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {100*s/total:6.2f}")
        out["kernels"].append({"name": n, "calls": c, "total_ms": s / 1e6, "avg_us": a / 1e3, "min_us": mn / 1e3,
                               "max_us": mx / 1e3, "pct": 100 * s / total})
    try:
        pm = pmc_rows(cur)
    except sqlite3.Error as e:
        pm = []
        print("no pmc:", e)
    pm.sort(key=lambda r: -r[3])
    for n, cn, c, s, a in pm[:40]:
        print(f"PMC {short(n):100s} {cn:12s} calls {c:6d} avg {a:14.1f}")
        out["pmc"].append({"name": n, "counter": cn, "calls": c, "avg": a, "sum": s})
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
