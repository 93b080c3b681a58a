#!/bin/bash
# diagnostic: SQ counters of k_layer with only one phase active (EDTTS_DIAG build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -DEDTTS_EXPERIMENTS -DEDTTS_DIAG edge-diffusion-tts_amd/csrc/edtts_kernels.hip -o /tmp/libedtts_diag.so
for skip in 14 13 11 7 0; do
  EDTTS_LIB=/tmp/libedtts_diag.so EDTTS_DIAG_SKIP=$skip rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/diagpmc/s$skip -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for skip in (14, 13, 11, 7, 0):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/diagpmc/s{skip}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_layer" in r["Kernel_Name"] and "Li0E" in r["Kernel_Name"] or ("k_layer" in r["Kernel_Name"] and ", 0>" in r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    if not m: print(skip, "no data"); continue
    wc = m["SQ_WAVE_CYCLES"]
    print(f"skip={skip:2d} wave_cyc {wc/1e6:8.1f}M  wait_any {m['SQ_WAIT_ANY']/wc:5.2f}  wait_inst_any {m['SQ_WAIT_INST_ANY']/wc:5.2f}  active_any {m['SQ_ACTIVE_INST_ANY']/wc:5.2f}  active_valu {m['SQ_ACTIVE_INST_VALU']/wc:5.2f}  mfma_busy/wave_cyc/4 {m['SQ_VALU_MFMA_BUSY_CYCLES']/(wc*4):5.2f}  valu/mfma {m['SQ_INSTS_VALU']/max(m['SQ_INSTS_MFMA'],1):5.2f}  mfma {m['SQ_INSTS_MFMA']/1e6:6.1f}M")
PY
