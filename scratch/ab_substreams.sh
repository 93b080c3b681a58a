#!/bin/bash
# A/B of the two-stream cut on one device: the headline loop with edtts_set_substreams(1) and (2), interleaved
set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "substreams or full_size_config2 or graph_capturable or generate_cfg1 or library_is_loaded" > gpurun_out/r4_sub_tests.log 2>&1 || { tail -30 gpurun_out/r4_sub_tests.log; exit 1; }
tail -3 gpurun_out/r4_sub_tests.log
for rep in 1 2; do
  for n in 1 2; do
    python bench.py --steps 100 --warmup 5 --no-pmc --no-cpu-baseline --substreams $n > gpurun_out/r4_sub_${n}_${rep}.json 2> gpurun_out/r4_sub_${n}_${rep}.err
    python - <<PY
import json
r = json.load(open("gpurun_out/r4_sub_${n}_${rep}.json"))
print("substreams=$n rep=$rep ms/step %.4f median %.4f value %.4g roofline %.4f avg_launch %.4f" % (r["ms_per_step"], r["ms_per_step_median"], r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]))
PY
  done
done
