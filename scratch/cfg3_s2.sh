#!/bin/bash
mkdir -p gpurun_out/cfg3
EDTTS_LIB=$PWD/scratch/lib_16s2.so timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "bf16_cfg3_shape or bf16_full_size or bf16_sampler_cfg3" 2>&1 | tail -2
run() { name=$1; shift
env "$@" python3 bench.py --config 3 --steps 20 --warmup 4 --no-pmc --no-cpu-baseline > gpurun_out/cfg3/$name.json 2> gpurun_out/cfg3/$name.err
python3 -c "
import json
r = json.load(open('gpurun_out/cfg3/$name.json'))
print('%-12s ms/step %.4f roofline %.4f avg_launch %.4f substreams %d' % ('$name', r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms'], r['config']['substreams']))"
}
run base EDTTS_X=0
run s2 EDTTS_LIB=$PWD/scratch/lib_16s2.so
run base1 EDTTS_SUBSTREAMS=1
run s2_1 EDTTS_LIB=$PWD/scratch/lib_16s2.so EDTTS_SUBSTREAMS=1
