#!/bin/bash
# usage: ab_lib_subs.sh lib1 lib2 ...: headline loop (100 steps) of the product library and experiment builds at substreams 1, 2 (and 4)
mkdir -p gpurun_out/absubs
python3 -c "import __graft_entry__ as g; g.build()" 2> gpurun_out/absubs/build.err
for rep in 1 2; do
for lib in base "$@"; do
  if [ $lib = base ]; then unset EDTTS_LIB; else export EDTTS_LIB=$PWD/scratch/lib_$lib.so; fi
  for n in ${SUBLIST:-1 2}; do
    python3 bench.py --steps 100 --warmup 10 --no-pmc --no-cpu-baseline --no-roofline --substreams $n > gpurun_out/absubs/${lib}_$n.json 2> gpurun_out/absubs/${lib}_$n.err
    python3 -c "
import json
r = json.load(open('gpurun_out/absubs/${lib}_$n.json'))
print('%-8s substreams=$n rep=$rep ms/step %.4f median %.4f value %.4g whole-call frac %.4f' % ('$lib', r['ms_per_step'], r['ms_per_step_median'], r['value'], r['whole_call']['frac_of_mfma_peak']))"
  done
done
done
