#!/bin/bash
mkdir -p gpurun_out/cfg3
for rep in 1 2; do for n in 2 4; do
python3 bench.py --config 3 --steps 30 --warmup 5 --no-pmc --no-cpu-baseline --no-roofline --substreams $n > gpurun_out/cfg3/s${n}_$rep.json 2> gpurun_out/cfg3/s${n}_$rep.err
python3 -c "
import json
r = json.load(open('gpurun_out/cfg3/s${n}_$rep.json'))
print('cfg3 substreams=$n rep=$rep ms/step %.4f median %.4f' % (r['ms_per_step'], r['ms_per_step_median']))"
done; done
