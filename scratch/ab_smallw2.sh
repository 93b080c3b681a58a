#!/bin/bash
set -e
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 5 --no-pmc --no-cpu-baseline --substreams 1 > gpurun_out/r4_$name.json 2> gpurun_out/r4_$name.err || { tail -5 gpurun_out/r4_$name.err; return 1; }
  python - <<PY
import json
r = json.load(open("gpurun_out/r4_$name.json"))
print("$name ms/step %.4f value %.4g roofline %.4f avg_launch %.4f" % (r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]))
PY
}
run base_a EDTTS_X=0
run smallw2_a EDTTS_LIB=$PWD/scratch/lib_smallw2.so EDTTS_FORCE_SMALL=1
run smallw1_a EDTTS_LIB=$PWD/scratch/lib_smallw2.so EDTTS_FORCE_SMALL=0
run base_b EDTTS_X=0
run smallw2_b EDTTS_LIB=$PWD/scratch/lib_smallw2.so EDTTS_FORCE_SMALL=1
