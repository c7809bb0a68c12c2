#!/bin/bash
# kernel trace of single-stream forward passes for each ICL_FUSE mask given
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O
for m in "$@"; do
  out=$O/tr_$m; rm -rf $out
  ICL_FUSE=$m ICL_EMBED_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --embed-only --total-images 2560 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  f=$(find $out -name '*kernel_trace.csv' | head -1)
  python3 $R/scratch/kernel_times.py $f > $O/kt_fuse$m.txt
  rm -rf $out
  echo "== fuse $m"; cat $O/kt_fuse$m.txt
done
