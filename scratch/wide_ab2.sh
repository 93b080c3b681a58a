#!/bin/bash
# usage: scratch/wide_ab2.sh "<flags A>" "<flags B>" ... -- config-3 bench of the 64-frame bf16 instance (EDTTS16_WIDE=1) per flag set
set -e
SRC=edge-diffusion-tts_amd/csrc/edtts_kernels.hip
i=0
for f in "$@"; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -DEDTTS_EXPERIMENTS $f $SRC -o /tmp/lib_ab_$i.so & i=$((i+1)); done; wait
export EDTTS16_WIDE=${WIDE:-1}
for rep in 1 2; do i=0; for f in "$@"; do
  EDTTS_LIB=/tmp/lib_ab_$i.so timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('[$f]', 'layer_avg_ms', round(d['roofline']['avg_launch_ms'],4), 'ms_per_step', round(d['ms_per_step'],3), 'frames/s', round(d['value']))"
  i=$((i+1)); done; done
