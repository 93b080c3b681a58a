#!/bin/bash
# usage: coop_check2.sh <lib.so>: cooperative-instance tests + small-grid benches of an experiment build against the product library
mkdir -p gpurun_out/coop
export X=$PWD/$1
EDTTS_LIB=$X timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "cooperative or generate_cfg1 or small_batch_instance or graph_capturable or dpm_solver or inpaint" > gpurun_out/coop/tests2.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/coop/tests2.log
for lib in base x; do
  if [ $lib = base ]; then export EDTTS_LIB=$PWD/scratch/lib_head.so; else export EDTTS_LIB=$X; fi
  python bench.py --config 1 --steps 300 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/coop/cfg1_$lib.json 2> gpurun_out/coop/cfg1_$lib.err
  python bench.py --config 2 --batch 32 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/coop/b32_$lib.json 2> gpurun_out/coop/b32_$lib.err
  python bench.py --config 2 --batch 8 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/coop/b8_$lib.json 2> gpurun_out/coop/b8_$lib.err
done
python - <<'PY'
import json
for n in ("cfg1", "b32", "b8"):
    for m in ("base", "x"):
        try:
            r = json.load(open(f"gpurun_out/coop/{n}_{m}.json"))
            print("%-5s %-5s ms/step %.4f value %.4g roofline %.4f avg_launch %.4f" % (n, m, r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]))
        except Exception as e:
            print(n, m, "failed", e)
PY
EDTTS_LIB=$X python scratch/graph_cfg1.py 2>&1 | tail -4
