#!/bin/bash
# usage: ab16.sh <lib>: bf16 parity tests on an experiment build, then interleaved config-3 A/B against scratch/lib_16base.so (a copy of the product library)
mkdir -p gpurun_out
export EDTTS_BENCH_SPAWNED=1  # (bench.py then skips its stale check + rebuild of the in-tree library: both sides run through EDTTS_LIB)
EDTTS_LIB=$PWD/scratch/lib_$1.so timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "bf16" 2>&1 | tail -3 || exit 1
for r in 1 2 3; do
  for l in 16base $1; do
    export EDTTS_LIB=$PWD/scratch/lib_$l.so
    timeout -k 10 200 python bench.py --config 3 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$l', d['ms_per_step'], d['roofline'].get('avg_launch_ms'))" || exit 1
  done
done
