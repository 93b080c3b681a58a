#!/bin/bash
# usage: ab_lib.sh <lib.so> [extra env...]: interleaved A/B of the product library against an experiment build (same device)
set -e
LIBX=$1; shift
mkdir -p gpurun_out
run() { name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --no-pmc --no-cpu-baseline --substreams 1 > gpurun_out/r4_$name.json 2> gpurun_out/r4_$name.err || { tail -5 gpurun_out/r4_$name.err; return 1; }
  python - <<PY
import json
r = json.load(open("gpurun_out/r4_$name.json"))
print("$name ms/step %.4f value %.4g roofline %.4f avg_launch %.4f" % (r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]))
PY
}
EDTTS_LIB=$PWD/$LIBX "$@" python -m pytest tests -m gpu -x -q -k "generate_cfg1 or forward_cfg_dims or deterministic_and_batch or full_size_config2 or small_batch_instance" > gpurun_out/r4_x_tests.log 2>&1 || { tail -30 gpurun_out/r4_x_tests.log; echo TESTS FAILED; }
tail -2 gpurun_out/r4_x_tests.log
run base_a EDTTS_X=0
run x_a EDTTS_LIB=$PWD/$LIBX "$@"
run base_b EDTTS_X=0
run x_b EDTTS_LIB=$PWD/$LIBX "$@"
