#!/bin/bash
# usage: trace_cfg.sh <tag> <bench args...>: rocprofv3 kernel trace of a short bench run -> per-kernel mean durations of the LAST call
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-roofline "$@" > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.err
python3 - $GRAFT_REPO_ROOT/gpurun_out/trace_$tag/t_kernel_trace.csv <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if not r["Kernel_Name"].startswith(("void at::", "__amd"))]
# one call = from k_randn to the next k_randn: take the last complete one
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_randn")]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-60s start %8.1f us  dur %7.1f us  gap %5.1f us  grid %s block %s vgpr %s" % (r["Kernel_Name"][:60], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"]))
    prev_end = e
print("call span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
