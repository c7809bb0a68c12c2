#!/bin/bash
# round-3 measurement pass: 100k / 250k bench lines on the recycled-storage Ward (3x3 halo kernel off), then the embed tests
# with the halo kernel and its A/B timing
set -o pipefail
mkdir -p gpurun_out
export ICL_WARD_STATS=1
ICL_CONV_MODE=2 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b100k.json 2> gpurun_out/b100k.err || { tail -c 2000 gpurun_out/b100k.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/b100k.json"))
print("100k:", j["value"], j["ms_per_step"], j["stages_ms_last_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"])
PY
ICL_CONV_MODE=2 timeout -k 10 400 python bench.py --total-images 250000 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/b250k.json 2> gpurun_out/b250k.err || { tail -c 2000 gpurun_out/b250k.err; exit 1; }
python - <<'PY'
import json
j = json.load(open("gpurun_out/b250k.json"))
print("250k:", j["value"], j["ms_per_step"], j["stages_ms_last_step"], j["roofline"]["avg_launch_us"], j["roofline"]["frac"])
PY
tail -n 3 gpurun_out/b250k.err
timeout -k 10 600 python -m pytest tests/test_embed_gpu.py -x -q -m gpu > gpurun_out/t_embed.log 2>&1 || { tail -30 gpurun_out/t_embed.log; exit 1; }
tail -3 gpurun_out/t_embed.log
for mode in 2 1 2 1; do
  ICL_CONV_MODE=$mode timeout -k 10 200 python bench.py --embed-only --total-images 51200 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/e_mode$mode.json 2>/dev/null || exit 1
  python - <<PY
import json
j = json.load(open("gpurun_out/e_mode$mode.json"))
print("embed-only mode $mode:", j["value"], j["ms_per_step"], j["roofline"]["achieved"], j["roofline"]["frac"])
PY
done
