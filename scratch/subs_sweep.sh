#!/bin/bash
mkdir -p gpurun_out/sweep
run() { tag=$1; shift
python3 bench.py "$@" --no-pmc --no-cpu-baseline --no-roofline > gpurun_out/sweep/$tag.json 2> gpurun_out/sweep/$tag.err
python3 -c "
import json
r = json.load(open('gpurun_out/sweep/$tag.json'))
print('%-22s ms/step %.4f median %.4f' % ('$tag', r['ms_per_step'], r['ms_per_step_median']))"
}
for rep in 1 2; do
for n in 4 6 8; do run cfg3_n${n}_$rep --config 3 --steps 30 --warmup 5 --substreams $n; done
for n in 1 2; do run b128_n${n}_$rep --batch 128 --steps 100 --warmup 10 --substreams $n; done
for n in 2 4; do run b512_n${n}_$rep --batch 512 --steps 50 --warmup 5 --substreams $n; done
done
