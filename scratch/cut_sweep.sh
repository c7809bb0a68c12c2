#!/bin/bash
# margin sweep of the early cut at the metric's size
for m in 3 7 15; do
  ICL_WARD_CUT_MARGIN=$m ICL_WARD_STATS=1 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cs.json 2> gpurun_out/cs.err || { tail -c 1500 gpurun_out/cs.err; exit 1; }
  echo "== cut margin $m"; grep "distance bounds in" gpurun_out/cs.err | tail -1
  python3 -c "
import json; j=json.load(open('gpurun_out/cs.json')); print(j['value'], j['ms_per_step'], j['stages_ms_last_step'], j['roofline']['avg_launch_us'])"
done
