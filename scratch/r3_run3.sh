#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ward_gpu.py tests/test_multi_gpu.py -x -q -m gpu -k "not 100k" 2>&1 | tail -3 || exit 1
for arg in "" "--no-overlap"; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline $arg > gpurun_out/ov.json 2> gpurun_out/ov.err || { tail -c 2000 gpurun_out/ov.err; exit 1; }
  python - <<PY
import json
j = json.load(open("gpurun_out/ov.json"))
print("100k [$arg]:", j["value"], j["ms_per_step"], j["stages_ms_last_step"])
PY
done
