#!/bin/bash
mkdir -p gpurun_out/coop
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "cooperative or small_batch_instance or generate_cfg1 or substreams" > gpurun_out/coop/tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/coop/tests.log
for mode in 0 -1; do
  EDTTS_COOP=$mode python bench.py --config 1 --steps 300 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/coop/cfg1_$mode.json 2> gpurun_out/coop/cfg1_$mode.err
  EDTTS_COOP=$mode python bench.py --config 2 --batch 32 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/coop/b32_$mode.json 2> gpurun_out/coop/b32_$mode.err
  EDTTS_COOP=$mode python bench.py --config 2 --batch 8 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline > gpurun_out/coop/b8_$mode.json 2> gpurun_out/coop/b8_$mode.err
done
python - <<'PY'
import json
for n in ("cfg1", "b32", "b8"):
    for m in ("0", "-1"):
        try:
            r = json.load(open(f"gpurun_out/coop/{n}_{m}.json"))
            print("%-5s coop=%-3s ms/step %.4f value %.4g roofline %.4f avg_launch %.4f" % (n, m, r["ms_per_step"], r["value"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"]))
        except Exception as e:
            print(n, m, "failed", e)
PY
