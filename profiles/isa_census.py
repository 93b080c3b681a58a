#!/usr/bin/env python3
"""Instruction census of one kernel of a hipcc -save-temps .s file, per basic block (blocks end at labels AND at branches).

    python profiles/isa_census.py <file.s> <mangled-name-substring> [--min N]

Classes: mfma | acc (v_accvgpr_*) | trans (v_exp/rcp/rsq/log/sqrt) | valu (other VALU) | vld / vst (global, buffer, scratch) |
lds | wait (s_waitcnt) | nop | salu | smem | br.  Trip counts are not known to the tool: the per-phase totals of the shipped kernel
(profiles/r03_isa_census.md) multiply these rows by the loop structure of csrc/edtts_kernels.hip."""
import collections
import re
import sys


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")): return "vld"
    if op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store")): return "vst"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch")): return "br"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"


KEYS = ["mfma", "acc", "trans", "valu", "vld", "vst", "lds", "wait", "nop", "salu", "smem", "br"]


def census(path, name):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[_A-Za-z0-9]+:", l) and name in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # the function may hold several s_endpgm (early exits): take the LAST one before the next function symbol
    nxt = next((i for i in range(start + 1, len(lines)) if re.match(r"^[_A-Za-z0-9]+:\s", lines[i] + " ") and ".LBB" not in lines[i] and i > start + 5), len(lines))
    ends = [i for i in range(start, nxt) if lines[i].strip().startswith("s_endpgm")]
    end = ends[-1] if ends else end
    blocks, cur = [], None

    def new(label, i):
        nonlocal cur
        cur = {"label": label, "line": i - start, "n": collections.Counter(), "br": []}
        blocks.append(cur)

    new("entry", start)
    for i in range(start + 1, end + 1):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            new(m.group(1), i)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        c = cls(op)
        cur["n"][c] += 1
        if c == "br":
            cur["br"].append(t.split(";")[0].strip())
            new(cur["label"] + "+", i)
    return [b for b in blocks if sum(b["n"].values())]


if __name__ == "__main__":
    path, name = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 6
    bl = census(path, name)
    print("%-14s %6s " % ("block", "line") + " ".join("%5s" % k for k in KEYS))
    tot = collections.Counter()
    for b in bl:
        tot.update(b["n"])
        if sum(b["n"].values()) >= mn:
            print("%-14s %6d " % (b["label"][:14], b["line"]) + " ".join("%5d" % b["n"][k] for k in KEYS), " | ".join(b["br"]))
    print("%-14s %6s " % ("static total", "") + " ".join("%5d" % tot[k] for k in KEYS))
