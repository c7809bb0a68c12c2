#!/bin/bash
# A/B of two builds (scratch/so/lib_<tag>.so) at the metric's size, alternating
cp imageclust_amd/libimageclust_hip.so /tmp/lib_keep.so
for v in "$@"; do
  cp scratch/so/lib_$v.so imageclust_amd/libimageclust_hip.so
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ss.json 2> gpurun_out/ss.err || { tail -c 800 gpurun_out/ss.err; break; }
  python3 -c "
import json; j=json.load(open('gpurun_out/ss.json')); print('$v', j['value'], j['ms_per_step'], j['stages_ms_last_step'])"
done
cp /tmp/lib_keep.so imageclust_amd/libimageclust_hip.so
