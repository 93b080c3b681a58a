#!/bin/bash
# usage: BENCH_ARGS="--batch 32 --steps 50" scratch/ab_args.sh "<flags A>" "<flags B>" ...  -- like ab.sh with extra bench.py arguments
set -e
SRC=edge-diffusion-tts_amd/csrc/edtts_kernels.hip
i=0
for f in "$@"; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -DEDTTS_EXPERIMENTS $f $SRC -o /tmp/lib_ab_$i.so & i=$((i+1)); done; wait
for rep in 1 2; do i=0; for f in "$@"; do
  EDTTS_LIB=/tmp/lib_ab_$i.so python bench.py $BENCH_ARGS --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('[$f]', 'layer_avg_ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'ms_per_step', round(d['ms_per_step'],3), 'frames/s', round(d['value']))"
  i=$((i+1)); done; done
