#!/usr/bin/env python3
"""PMC summary of the bf16 config-3 run collected by scratch/pmc_cfg3.sh:  profiles/summarise_cfg3.py <tag> gpurun_out/pmc_<tag>
-> profiles/<tag>_cfg3_pmc_summary.json (per-kernel counter means + derived fractions; FETCH_SIZE x2 on gfx950, KiB -> bytes)."""
import collections, csv, glob, json, os, sys
tag, raw = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(raw + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if not any(s in k for s in ("k_layer16", "k_prologue16", "k_ctx")):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"counters_mean_per_dispatch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        d["hbm_read_bytes"] = 2 * m["FETCH_SIZE"] * 1024
        d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    g = m.get
    if g("SQ_WAVE_CYCLES"):
        d["mfma_busy_frac_of_wave_lifetime"] = g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4 / g("SQ_WAVE_CYCLES")
        d["wait_any_frac_of_wave_cycles"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
        d["issue_stall_frac"] = g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
        d["active_frac"] = g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    if g("GRBM_GUI_ACTIVE"):
        d["mfma_busy_frac_of_chip"] = g("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g("GRBM_GUI_ACTIVE") * 128)  # (as profiles/summarise.py)
    if g("SQ_INSTS_MFMA"):
        d["non_mfma_valu_per_mfma"] = (g("SQ_INSTS_VALU", 0) - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    out[k] = d
json.dump(out, open(os.path.join(here, f"{tag}_cfg3_pmc_summary.json"), "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k[:70], {x: (round(v, 4) if isinstance(v, float) and v < 10 else v) for x, v in d.items() if x != "counters_mean_per_dispatch"})
