#!/bin/bash
# per-phase SQ counters of the layer kernel: -DEDTTS_EXPERIMENTS -DEDTTS_DIAG build, phases skipped by EDTTS_DIAG_SKIP (bit0 self-attn,
# bit1 q_proj + cross-attn, bit2 FFN, bit3 tail); results of the runs are wrong by construction -- counters and timing only
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form -DEDTTS_EXPERIMENTS -DEDTTS_FAST_BUILD -DEDTTS_DIAG edge-diffusion-tts_amd/csrc/edtts_kernels.hip -o /tmp/libedtts_diag.so
rm -rf gpurun_out/diagpmc; mkdir -p gpurun_out/diagpmc
for skip in 0 1 2 4 8 14 13 11 7; do
  EDTTS_LIB=/tmp/libedtts_diag.so EDTTS_DIAG_SKIP=$skip rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/diagpmc/s$skip -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-roofline --substreams 1 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
names = {0: "all phases", 1: "- self-attn", 2: "- q_proj+cross", 4: "- FFN", 8: "- tail", 14: "self-attn only", 13: "q_proj+cross only", 11: "FFN only", 7: "tail only"}
print("%-20s %12s %12s %12s %12s %10s" % ("run", "wave cyc/wave", "wait/wave", "valu/wave", "mfma/wave", "busy"))
for skip in (0, 1, 2, 4, 8, 14, 13, 11, 7):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/diagpmc/s{skip}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_layer" in r["Kernel_Name"] and ", 0, 0>" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    if not m:
        print(names[skip], "no data"); continue
    w = 4096.0
    print("%-20s %12.0f %12.0f %12.0f %12.0f %10.3f" % (names[skip], 4 * m["SQ_WAVE_CYCLES"] / w, 4 * m["SQ_WAIT_ANY"] / w, (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / w,
                                                        m["SQ_INSTS_MFMA"] / w, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_WAVE_CYCLES"])))
PY
