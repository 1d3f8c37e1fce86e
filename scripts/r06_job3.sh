#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job3
mkdir -p $O
python -m pytest tests/test_context_parallel_gpu.py tests/test_dit_gpu.py -m gpu -x -q -k "context_parallel or diagnostics" 2>&1 | tail -5
timeout 300 python bench.py --cp-host-probe --hipgraph --steps 3 --warmup 3 --no-cpu-baseline > $O/probe_graph.json 2> $O/probe_graph.err; echo "graph rc=$?"
timeout 300 python bench.py --cp-host-probe --steps 3 --warmup 3 --no-cpu-baseline > $O/probe_eager.json 2> $O/probe_eager.err; echo "eager rc=$?"
grep -o '"cp_host_ms_per_block": [0-9.]*' $O/probe_graph.json $O/probe_eager.json
