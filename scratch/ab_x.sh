#!/bin/bash
# usage: ab_x.sh <lib>: parity subset of an experiment build (FAST build: default decoder only), then interleaved A/B against lib_head
EDTTS_LIB=$PWD/scratch/lib_$1.so timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "generate_cfg1 or forward_cfg_dims or full_size_config2 or deterministic or small_batch_instance or random_geometries" 2>&1 | tail -2
bash scratch/ab_three.sh head $1
