"""Summarise a rocprofv3 rocpd SQLite result (kernel stats and, if present, PMC counters) as text/JSON.
Usage: python scripts/rocpd_summary.py <results.db> [out.json]"""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void rtv::", "").replace("rtv::", "")
    return name[:110]


def main():
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
        pm = cur.execute("select k.name, p.counter_name, count(*), sum(p.counter_value), avg(p.counter_value) "
                         "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                         "group by k.name, p.counter_name order by sum(p.counter_value) desc").fetchall()
    except sqlite3.Error as e:
        pm = []
        print("no pmc:", e)
    for n, cn, c, s, a in pm[:40]:
        print(f"PMC {short(n):100s} {cn:12s} calls {c:6d} avg {a:14.1f}")
        out["pmc"].append({"name": n, "counter": cn, "calls": c, "avg": a, "sum": s})
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
