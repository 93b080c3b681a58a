#!/bin/bash
set -e
mkdir -p gpurun_out
for rep in 1 2; do
  for n in 1 2 3 4; do
    python bench.py --steps 100 --warmup 5 --no-pmc --no-cpu-baseline --no-roofline --substreams $n > gpurun_out/r4_sub_${n}_${rep}.json 2> gpurun_out/r4_sub_${n}_${rep}.err
    python - <<PY
import json
r = json.load(open("gpurun_out/r4_sub_${n}_${rep}.json"))
print("substreams=$n rep=$rep ms/step %.4f median %.4f value %.4g" % (r["ms_per_step"], r["ms_per_step_median"], r["value"]))
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for n in 2 4; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4_trace_sub$n -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-pmc --no-cpu-baseline --no-roofline --substreams $n > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r4_trace_sub$n.err
done
ls -R $GRAFT_REPO_ROOT/gpurun_out/r4_trace_sub2 | head
