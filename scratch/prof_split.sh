#!/bin/bash
# kernel-trace statistics of scratch/split_probe.py in fused and split mode -> gpurun_out/prof_split/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_split
rm -rf "$OUT"; mkdir -p "$OUT"
for mode in 1 0; do
  export EDTTS16_SPLIT=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/m$mode" -o t -- python3 scratch/split_probe.py > "$OUT/m$mode.out" 2> "$OUT/m$mode.err"
  f=$(find "$OUT/m$mode" -name "*kernel_stats.csv" | head -1)
  echo "mode $mode: $f"
  [ -n "$f" ] && cut -c1-160 "$f" | head -9
done
