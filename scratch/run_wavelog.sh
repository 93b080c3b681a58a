#!/bin/bash
mkdir -p gpurun_out
for lib in "$@"; do for s in 1 2; do
echo "=== $lib substreams=$s"
EDTTS_LIB=$PWD/scratch/lib_$lib.so timeout -k 10 200 python3 scratch/wavelog.py $s 2>&1 | tail -14
done; done > gpurun_out/r4_wavelog.txt 2>&1
cat gpurun_out/r4_wavelog.txt
