#!/bin/bash
# A/B: register-ring weight stream (default) vs block-shared LDS ring (-DEDTTS_STREAM_LDS)
set -e
SRC=edge-diffusion-tts_amd/csrc/edtts_kernels.hip
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC $SRC -o /tmp/lib_ring.so
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -DEDTTS_STREAM_LDS $SRC -o /tmp/lib_lds.so
for rep in 1 2; do for v in ring lds; do
  EDTTS_LIB=/tmp/lib_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$v', 'layer_avg_ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'ms_per_step', round(d['ms_per_step'],3), 'frames/s', round(d['value']))"
done; done
