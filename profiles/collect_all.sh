#!/bin/bash
# Everything the round's evidence needs, in one gpurun call (from the repo root):  bash profiles/collect_all.sh r02
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/all_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
echo "== gpu tests"; timeout -k 10 600 python3 -m pytest tests -m gpu -q -s > "$OUT/gpu_tests.log" 2>&1; tail -2 "$OUT/gpu_tests.log"
echo "== headline profile (config 2, fp32)"; timeout -k 10 900 bash profiles/collect.sh "$TAG" > "$OUT/collect.log" 2>&1; tail -1 "$OUT/collect.log"
echo "== config 3 (bf16) PMC"; timeout -k 10 600 bash scratch/pmc_cfg3.sh "$TAG" > "$OUT/pmc_cfg3.log" 2>&1
echo "== bench lines"
timeout -k 10 300 python3 bench.py --config 3 > "$OUT/bench_cfg3_bf16.json" 2> "$OUT/bench_cfg3_bf16.err"
timeout -k 10 400 python3 bench.py --config 3 --dtype f32 --no-pmc > "$OUT/bench_cfg3_f32.json" 2> "$OUT/bench_cfg3_f32.err"
timeout -k 10 200 python3 bench.py --config 1 --no-pmc > "$OUT/bench_cfg1.json" 2> "$OUT/bench_cfg1.err"
timeout -k 10 200 python3 bench.py --batch 32 --no-pmc --no-cpu-baseline > "$OUT/bench_b32.json" 2> "$OUT/bench_b32.err"
timeout -k 10 300 python3 bench.py --config 5 --no-pmc > "$OUT/bench_cfg5.json" 2> "$OUT/bench_cfg5.err"
timeout -k 10 200 python3 bench.py --config 3 --batch 32 --steps 20 --no-pmc --no-cpu-baseline > "$OUT/bench_cfg3_b32.json" 2> "$OUT/bench_cfg3_b32.err"
timeout -k 10 200 python3 bench.py --config 3 --batch 16 --steps 30 --no-pmc --no-cpu-baseline > "$OUT/bench_cfg3_b16.json" 2> "$OUT/bench_cfg3_b16.err"
timeout -k 10 200 python3 bench.py --batch 8 --no-pmc --no-cpu-baseline > "$OUT/bench_b8.json" 2> "$OUT/bench_b8.err"
EDTTS_COOP=0 timeout -k 10 200 python3 bench.py --config 1 --no-pmc --no-cpu-baseline > "$OUT/bench_cfg1_coop0.json" 2> "$OUT/bench_cfg1_coop0.err"
EDTTS_COOP=0 timeout -k 10 200 python3 bench.py --batch 32 --no-pmc --no-cpu-baseline > "$OUT/bench_b32_coop0.json" 2> "$OUT/bench_b32_coop0.err"
echo "== kernel timeline of one B=1 call"; timeout -k 10 200 bash scratch/trace_cfg.sh cfg1 --config 1 > "$OUT/trace_cfg1.txt" 2>&1
echo "== elementwise"; timeout -k 10 200 python3 scratch/bench_elementwise.py 2>/dev/null > "$OUT/elementwise.json"
echo "== ring probe"; timeout -k 5 60 scratch/ring_probe > "$OUT/ring_probe.txt" 2>&1
ls -la "$OUT"
