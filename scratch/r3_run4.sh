#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_ward_gpu.py tests/test_multi_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu -k "not 100k" > gpurun_out/t4.log 2>&1 || { tail -40 gpurun_out/t4.log; exit 1; }
tail -3 gpurun_out/t4.log
ICL_WARD_STATS=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bb.json 2> gpurun_out/bb.err || { tail -c 2000 gpurun_out/bb.err; exit 1; }
python - <<PY
import json
j = json.load(open("gpurun_out/bb.json"))
print("100k bound:", j["value"], j["ms_per_step"], j["stages_ms_last_step"], j["roofline"]["avg_launch_us"])
PY
tail -2 gpurun_out/bb.err
