#!/bin/bash
# diagnostic only: time k_layer with phases skipped (outputs are wrong in these runs by construction)
set -e
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -DEDTTS_EXPERIMENTS -DEDTTS_DIAG edge-diffusion-tts_amd/csrc/edtts_kernels.hip -o /tmp/libedtts_diag.so
for skip in 0 1 2 4 8 3 7 15 14 13 11; do
  EDTTS_LIB=/tmp/libedtts_diag.so EDTTS_DIAG_SKIP=$skip python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('skip=$skip', 'layer_avg_ms', round(d['roofline']['avg_launch_ms'],4), 'ms_per_step', round(d['ms_per_step'],3))"
done
