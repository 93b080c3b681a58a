#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of profiles/collect.sh into the small committed summaries profiles/<tag>_*."""
import collections, csv, glob, json, os, shutil, sys

tag, out = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))

# 1. kernel trace stats (copied verbatim) + bench lines
for src, dst in ((glob.glob(os.path.join(out, "trace", "*kernel_stats.csv")), f"{tag}_kernel_stats.csv"),):
    if src:
        shutil.copy(src[0], os.path.join(here, dst))
for name in ("bench.json", "bench_under_trace.json"):
    p = os.path.join(out, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(here, f"{tag}_{name}"))

# 2. PMC: mean per dispatch, per kernel
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
for k, cs in agg.items():
    if "k_" not in k:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["dispatches_sampled"] = max(len(v) for v in cs.values())
    d = {"counters_mean_per_dispatch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        # MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a
        # wide coalesced (16 B/lane) streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores.
        d["hbm_read_bytes"] = 2.0 * m["FETCH_SIZE"] * 1024
        d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
        cycles = m["GRBM_GUI_ACTIVE"] / 8.0
        d["mfma_busy_frac_of_chip"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024)
        d["mfma_busy_frac_of_wave_lifetime"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_WAVE_CYCLES"])
        d["wait_any_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in m:
        d["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in m:
        d["l1_read_miss_per_access"] = m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"]
    summary[k] = d
json.dump(summary, open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)

# 3. the number bench.py reports as roofline.traffic: HBM bytes per k_layer launch (mean over the layer-kernel variants)
layer = [v for k, v in summary.items() if "k_layer" in k and "hbm_bytes_per_launch" in v]
if layer:
    w = [v["counters_mean_per_dispatch"]["dispatches_sampled"] for v in layer]
    tr = sum(v["hbm_bytes_per_launch"] * n for v, n in zip(layer, w)) / sum(w)
    json.dump({"hbm_bytes_per_launch": tr, "source": f"profiles/{tag}_pmc_summary.json (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)"},
              open(os.path.join(here, f"{tag}_pmc_k_layer.json"), "w"))
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_mean_per_dispatch"} for k, v in summary.items()}, indent=1))
