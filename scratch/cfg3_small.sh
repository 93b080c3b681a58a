#!/bin/bash
mkdir -p gpurun_out/cfg3
run() { name=$1; shift
env "$@" python3 bench.py --config 3 --steps 20 --warmup 4 --no-pmc --no-cpu-baseline --substreams 1 > gpurun_out/cfg3/$name.json 2> gpurun_out/cfg3/$name.err
python3 -c "
import json
r = json.load(open('gpurun_out/cfg3/$name.json'))
print('%-12s ms/step %.4f roofline %.4f avg_launch %.4f' % ('$name', r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms']))"
}
run a_base EDTTS_LIB=$PWD/scratch/lib_16a.so
run a_small EDTTS_LIB=$PWD/scratch/lib_16a.so EDTTS16_FORCE_SMALL=1
run b_base EDTTS_LIB=$PWD/scratch/lib_16b.so
run b_small EDTTS_LIB=$PWD/scratch/lib_16b.so EDTTS16_FORCE_SMALL=1
