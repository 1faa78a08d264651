#!/bin/bash
# round 3, GPU call 6: whole GPU suite on the rebuilt library, then bench.py
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests_all.log 2>&1 || { tail -40 gpurun_out/r3_tests_all.log; exit 1; }
tail -3 gpurun_out/r3_tests_all.log
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err || { tail -20 gpurun_out/r3_bench_a.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_a.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], [round(t["value"],1) for t in d["temporal_blocking"]], round(d["fp16_storage"]["value"],1), d["cpu_baseline"]["value"])
PY
