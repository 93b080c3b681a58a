#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_icache; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 -L 2>/dev/null | grep -iE "ICACHE|IFETCH|INST_LEVEL|SQ_INSTS_SMEM|SQ_WAIT_INST" | head -40 > "$OUT/counters.txt"
for cfg in 2 3; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/c$cfg" -o pmc -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-roofline > /dev/null 2> "$OUT/c$cfg.err" || echo "pass $cfg failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
print(open(out+"/counters.txt").read()[:1500])
for cfg in ("c2","c3"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out+"/"+cfg+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)): agg[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in agg.items():
        if "k_layer" in k: print(cfg, k, {c: f"{sum(v)/len(v):.4g}" for c,v in sorted(cs.items())})
    print(open(out+"/"+cfg+".err").read()[-300:] if not agg else "")
PY
