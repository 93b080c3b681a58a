#!/bin/bash
# kernel-trace statistics of scratch/split_probe.py with a -DEDTTS16_SPLIT_BUILD=1 library (EDTTS_LIB): split layer at several
# attention occupancies (EDTTS16_ATT_LDS bytes of idle LDS per attention block: 0 -> 3 waves/SIMD, 80 KiB -> 2, 160 KiB -> 1)
# -> gpurun_out/prof_split/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_split
rm -rf "$OUT"; mkdir -p "$OUT"
export EDTTS16_SPLIT=1
for lds in 0 81920 163840; do
  export EDTTS16_ATT_LDS=$lds
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/lds$lds" -o t -- python3 scratch/split_probe.py > "$OUT/lds$lds.out" 2> "$OUT/lds$lds.err"
  f=$(find "$OUT/lds$lds" -name "*kernel_stats.csv" | head -1)
  echo "attention idle LDS $lds: $f"
  [ -n "$f" ] && grep "k_attn16" "$f" | cut -c1-140
done
