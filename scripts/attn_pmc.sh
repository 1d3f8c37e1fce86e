#!/bin/bash
# SQ counter passes over the attention kernel (bare launches, scripts/one_attn.py): LDS bank conflicts, wait / busy split.
# usage: scripts/attn_pmc.sh <waves: 81 lockstep | 82 four-phase>     (run on the GPU box from the repo root)
export TMPDIR=/tmp
R=$PWD
W=${1:-82}
OUT=/tmp/attn_pmc_$W
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
  --kernel-trace -d $OUT/p1 -o r -- python $R/scripts/one_attn.py 4680 9360 40 3 $W > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC \
  --kernel-trace -d $OUT/p2 -o r -- python $R/scripts/one_attn.py 4680 9360 40 3 $W > $OUT/p2.log 2>&1
tail -2 $OUT/p1.log $OUT/p2.log
cd $R
python - <<PY
import glob, sqlite3
for sub in ("p1", "p2"):
    for db in sorted(glob.glob("$OUT/%s/**/*.db" % sub, recursive=True)):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select k.name, p.counter_name, count(*), avg(p.counter_value) from pmc_events p join kernels k "
                               "on k.dispatch_id = p.dispatch_id where k.name like '%attn_fwd%' group by k.name, p.counter_name").fetchall()
        except Exception as e:
            print("query failed", e); continue
        for n, c, cnt, avg in rows:
            print(f"{n[:40]:40s} {c:32s} launches {cnt:3d}  avg/launch {avg:16.0f}")
PY
