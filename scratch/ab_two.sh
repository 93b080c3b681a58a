#!/bin/bash
# usage: ab_two.sh libA libB ...: interleaved isolated k_layer timing (roofline leg, substreams 1) of several experiment builds
mkdir -p gpurun_out/ab2
for rep in 1 2 3; do for lib in "$@"; do
EDTTS_LIB=$PWD/scratch/lib_$lib.so python3 bench.py --steps 30 --warmup 5 --no-pmc --no-cpu-baseline --substreams 1 > gpurun_out/ab2/$lib.json 2> gpurun_out/ab2/$lib.err
python3 -c "
import json
r = json.load(open('gpurun_out/ab2/$lib.json'))
print('%-10s rep $rep avg_launch %.4f ms frac %.4f' % ('$lib', r['roofline']['avg_launch_ms'], r['roofline']['frac']))"
done; done
