#!/bin/bash
# A/B of the 64-frame bf16 instance (EDTTS16_WIDE=0/1, product library); parity of the wide path on the bf16 tests first
set -e
EDTTS16_WIDE=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "bf16" -x -q 2>&1 | tail -5
for rep in 1 2; do for w in 0 1; do
  EDTTS16_WIDE=$w timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline --no-pmc 2>gpurun_out/wide_ab.err | python -c "import sys,json; d=json.load(sys.stdin); print('[wide=$w]', 'layer_avg_ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'ms_per_step', round(d['ms_per_step'],3), 'frames/s', round(d['value']))"
done; done
