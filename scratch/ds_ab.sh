#!/bin/bash
export EDTTS_LIB=$PWD/$1
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "dsconv" 2>&1 | tail -2
for rep in 1 2; do
EDTTS_DSCONV_WAVES8=1 python scratch/ds_time.py 2>&1 | tail -2
python scratch/ds_time.py 2>&1 | tail -2
done
