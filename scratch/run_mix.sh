#!/bin/bash
bash scratch/ab_three.sh cur lpt
EDTTS_LIB=$PWD/scratch/lib_ho.so timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "cooperative or generate_cfg1 or small_batch_instance" 2>&1 | tail -2
for rep in 1 2; do for lib in cur ho; do
EDTTS_LIB=$PWD/scratch/lib_$lib.so python3 bench.py --batch 32 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/ab3/b32_$lib.json 2>/dev/null
python3 -c "
import json
r = json.load(open('gpurun_out/ab3/b32_$lib.json'))
print('b32 %-4s rep $rep k_layer %.4f ms frac %.4f | call %.4f ms' % ('$lib', r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['ms_per_step']))"
done; done
