#!/bin/bash
# usage: ds_pre.sh lib1 lib2 ...: dsconv parity subset + graph timing of each experiment build, interleaved twice
for l in "$@"; do EDTTS_LIB=$PWD/scratch/lib_$l.so timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "dsconv" 2>&1 | tail -1; done
for rep in 1 2; do for l in "$@"; do echo -n "$l: "; EDTTS_LIB=$PWD/scratch/lib_$l.so python scratch/ds_time.py 2>/dev/null | tail -1; done; done
