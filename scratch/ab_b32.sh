#!/bin/bash
# usage: ab_b32.sh libA libB: small-grid benches (B=32, B=1) of two experiment builds, interleaved
mkdir -p gpurun_out/ab3
EDTTS_LIB=$PWD/scratch/lib_$2.so timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "cooperative or generate_cfg1" 2>&1 | tail -2
for rep in 1 2 3; do for lib in "$@"; do
EDTTS_LIB=$PWD/scratch/lib_$lib.so python3 bench.py --batch 32 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/ab3/b32_$lib.json 2>/dev/null
EDTTS_LIB=$PWD/scratch/lib_$lib.so python3 bench.py --config 1 --steps 300 --warmup 20 --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/ab3/c1_$lib.json 2>/dev/null
python3 -c "
import json
r = json.load(open('gpurun_out/ab3/b32_$lib.json')); c = json.load(open('gpurun_out/ab3/c1_$lib.json'))
print('%-6s rep $rep B=32 k_layer %.4f ms frac %.4f call %.4f ms | B=1 call %.4f ms' % ('$lib', r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['ms_per_step'], c['ms_per_step']))"
done; done
