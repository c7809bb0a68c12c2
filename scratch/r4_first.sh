#!/bin/bash
# round-4 GPU check: parity of the fused kernels, then embed-only A/B per ICL_FUSE mask, then a single-stream kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
timeout -k 10 420 python3 -m pytest tests/test_fused_gpu.py -x -q > $O/t_fused.log 2>&1; rc=$?; tail -15 $O/t_fused.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 -m pytest tests/test_embed_gpu.py -x -q > $O/t_embed.log 2>&1; rc=$?; tail -5 $O/t_embed.log; [ $rc -ne 0 ] && exit $rc
for m in ${MASKS:-0 7}; do
  ICL_FUSE=$m timeout -k 10 200 python3 bench.py --embed-only --total-images 51200 --steps 2 --warmup 1 --no-cpu-baseline > $O/e_fuse$m.json 2> $O/e_fuse$m.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/e_fuse$m.json')); print('fuse',$m, d['value'])"
done
bash scratch/r4_trace.sh ${TRACE:-7} 2>&1 | grep -v "^  " | head -30
