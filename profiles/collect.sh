#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   profiles/collect.sh r01
# Pass 1: --kernel-trace --stats of the bench command.  Passes 2..n: PMC counters, each in its own run (never combined
# with trace domains other than the implicit kernel dispatch records).  Raw output -> gpurun_out/prof_<tag>/, summaries
# are written by profiles/summarise.py into profiles/<tag>_*.
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
# every profiled pass times k_layer ALONE on the device: the two-stream cut of large batches (edtts_set_substreams, round 4) would
# let two launches share the SIMDs and stretch each one's duration; the un-profiled bench at the end runs the default
export EDTTS_SUBSTREAMS=1
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/bench_under_trace.json" 2> "$OUT/trace.err"
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  tag=$(echo $P | cut -d" " -f1)
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc_$tag" -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > /dev/null 2> "$OUT/pmc_$tag.err" || echo "PMC pass $tag failed"
done
unset EDTTS_SUBSTREAMS
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 bench.py --substreams 1 --no-pmc --no-cpu-baseline > "$OUT/bench_substreams1.json" 2> "$OUT/bench_substreams1.err"
python3 profiles/summarise.py "$TAG" "$OUT"
