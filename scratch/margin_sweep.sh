#!/bin/bash
# in-loop margin sweep of the distance-bound path at the metric's size
for m in 0 0.5 2 8; do
  ICL_WARD_LOOP_MARGIN=$m ICL_WARD_STATS=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ms.json 2> gpurun_out/ms.err || { tail -c 1500 gpurun_out/ms.err; exit 1; }
  echo "== loop margin $m"; grep "distance bounds" gpurun_out/ms.err | tail -2
  python -c "
import json; j=json.load(open('gpurun_out/ms.json')); print(j['value'], j['ms_per_step'], j['stages_ms_last_step'], j['roofline']['avg_launch_us'])"
done
