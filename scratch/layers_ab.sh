#!/bin/bash
# per-layer table of one forward pass (single stream) for each ICL_CONV_MODE given on the command line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "$@"; do
  out=$R/gpurun_out/lay_mode$mode
  rm -rf $out
  ICL_CONV_MODE=$mode ICL_EMBED_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --embed-only --total-images 2560 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $out -name '*kernel_trace.csv' | head -1)
  python3 $R/scratch/layer_report.py $f > $R/gpurun_out/layers_mode$mode.txt
  rm -rf $out
  echo "== mode $mode"; grep "k=3\|total conv\|batch span\|other" $R/gpurun_out/layers_mode$mode.txt
done
