#!/bin/bash
# extra SQ counters of the shipped kernel: instruction fetch, VMEM / SALU issue cycles, VALU-MFMA co-execution, TA FIFO stalls
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in "SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM" \
         "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $P | cut -d" " -f2)
  rocprofv3 --pmc $P --output-format csv -d gpurun_out/diagpmc2/$tag -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/diagpmc2/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_layer" in r["Kernel_Name"] and ", 0, 0>" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
wc = m["SQ_WAVE_CYCLES"]
for k in sorted(m): print(f"{k:32s} {m[k]:16.0f}  per wave-cycle {m[k]/wc:8.4f}")
PY
