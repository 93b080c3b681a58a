#!/bin/bash
EDTTS_LIB=$PWD/scratch/lib_tr2.so timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "generate_cfg1 or forward_cfg_dims or full_size_config2 or deterministic" 2>&1 | tail -2
bash scratch/ab_three.sh head tr2
