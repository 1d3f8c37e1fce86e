#!/bin/bash
# Samples rocm-smi (socket power, sclk) every 0.25 s while a command runs; prints min / median / max.
# usage: scripts/power_clock_sample.sh <label> <command...>
label=$1; shift
log=/tmp/smi_$label.log; : > $log
( while true; do rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power \(W\)|sclk clock|Max Graphics Package Power" >> $log; sleep 0.25; done ) &
smi=$!
"$@" > /tmp/cmd_$label.log 2>&1
kill $smi 2>/dev/null
python3 - "$label" "$log" <<'PY'
import re, statistics, sys
label, path = sys.argv[1], sys.argv[2]
pw, clk, cap = [], [], []
for line in open(path):
    m = re.search(r"([0-9.]+)\s*$", line.replace("Mhz)", ")").strip())
    if "Max Graphics Package Power" in line:
        v = re.findall(r"([0-9.]+)", line.split(":")[-1]); cap += [float(v[0])] if v else []
    elif "Power (W)" in line:
        v = re.findall(r"([0-9.]+)", line.split(":")[-1]); pw += [float(v[0])] if v else []
    elif "sclk" in line:
        v = re.findall(r"\(([0-9.]+)Mhz\)", line); clk += [float(v[0])] if v else []
def s(x): return f"min {min(x):.0f} median {statistics.median(x):.0f} max {max(x):.0f} (n={len(x)})" if x else "n/a"
print(f"[{label}] socket power W: {s(pw)}; sclk MHz: {s(clk)}; power cap W: {s(cap)}")
PY
