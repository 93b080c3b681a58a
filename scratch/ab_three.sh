#!/bin/bash
# usage: ab_three.sh libA libB ...: isolated k_layer time (roofline leg) AND the headline ms/step at substreams 2, interleaved
mkdir -p gpurun_out/ab3
for rep in 1 2 3; do for lib in "$@"; do
EDTTS_LIB=$PWD/scratch/lib_$lib.so python3 bench.py --steps 60 --warmup 10 --no-pmc --no-cpu-baseline > gpurun_out/ab3/$lib.json 2> gpurun_out/ab3/$lib.err
python3 -c "
import json
r = json.load(open('gpurun_out/ab3/$lib.json'))
print('%-10s rep $rep k_layer alone %.4f ms frac %.4f | call %.4f ms (substreams %d)' % ('$lib', r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['ms_per_step'], r['config']['substreams']))"
done; done
