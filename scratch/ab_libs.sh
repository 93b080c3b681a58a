#!/bin/bash
# usage: scratch/ab_libs.sh libA.so libB.so ...  -- bench prebuilt libraries interleaved, 3 rounds
for r in 1 2 3; do for L in "$@"; do
  EDTTS_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$L', 'layer_ms', round(d['roofline']['avg_launch_ms'],4), 'ms_per_step', round(d['ms_per_step'],3))"
done; done
