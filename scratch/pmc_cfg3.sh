#!/bin/bash
# PMC passes on the bf16 config-3 bench (gpurun from the repo root): scratch/pmc_cfg3.sh <tag>
TAG=${1:-cfg3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export EDTTS_SUBSTREAMS=1  # per-kernel numbers: launches must not share the device
CMD="python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/bench.json" 2> "$OUT/trace.err"
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" \
         "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $P | cut -d" " -f1)
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pmc_$tag" -o pmc -- $CMD > /dev/null 2> "$OUT/pmc_$tag.err" || echo "PMC pass $tag failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "k_layer16" in k or "k_prologue16" in k or "k_ctx" in k:
        print(k, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(cs.items())})
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for i, l in enumerate(open(f)):
        if i < 8: print(l.strip()[:150])
PY
