#!/bin/bash
# timing + SQ counters of experiment libraries against the product library (one device)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/w2diag
python3 -c "import __graft_entry__ as g; g.build()" 2> gpurun_out/w2diag/build.err
for lib in base "$@"; do
  if [ $lib = base ]; then unset EDTTS_LIB; else export EDTTS_LIB=$PWD/scratch/lib_$lib.so; fi
  python3 bench.py --steps 60 --warmup 10 --no-pmc --no-cpu-baseline --substreams ${SUBS:-1} > gpurun_out/w2diag/$lib.json 2> gpurun_out/w2diag/$lib.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/w2diag/pmcA_$lib -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-roofline --substreams 1 > /dev/null 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/w2diag/pmcB_$lib -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-roofline --substreams 1 > /dev/null 2>&1
done
python3 - "$@" <<'PY'
import csv, glob, collections, json, sys
for lib in ["base"] + sys.argv[1:]:
    try:
        r = json.load(open(f"gpurun_out/w2diag/{lib}.json"))
        line = "%-8s avg_launch %.4f ms frac %.4f ms/step %.3f |" % (lib, r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["ms_per_step"])
    except Exception as e:
        line = "%-8s bench failed %r |" % (lib, e)
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/w2diag/pmc?_{lib}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_layer" in row["Kernel_Name"] and ", 0, 0>" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    w = 4096.0
    if "SQ_WAVE_CYCLES" in m:
        line += " wave cyc %.0f wait %.0f wait_inst %.0f active %.0f valu/wave %.0f mfma/wave %.0f mfma_busy/wavecyc %.3f" % (
            4 * m["SQ_WAVE_CYCLES"] / w, 4 * m["SQ_WAIT_ANY"] / w, 4 * m["SQ_WAIT_INST_ANY"] / w, 4 * m["SQ_ACTIVE_INST_ANY"] / w,
            (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / w, m["SQ_INSTS_MFMA"] / w, m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * m["SQ_WAVE_CYCLES"]))
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in m:
        line += " | L1 acc %.3g -> L2 rd req %.3g (miss %.3f) vmem/wave %.0f salu/wave %.0f lds/wave %.0f busy_cyc %.4g gui %.4g" % (
            m["TCP_TOTAL_CACHE_ACCESSES_sum"], m["TCP_TCC_READ_REQ_sum"], m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"],
            m["SQ_INSTS_VMEM"] / w, m["SQ_INSTS_SALU"] / w, m["SQ_INSTS_LDS"] / w, m.get("SQ_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 0))
    print(line)
PY
