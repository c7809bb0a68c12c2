#!/bin/bash
# per-kernel stats of the FAST-mode (Lance-Williams) merge loop at N=100k
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
out=$O/lwst; rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/bench.py --update lw --steps 1 --warmup 0 --no-cpu-baseline > $O/lwst.log 2>&1
f=$(find $out -name '*kernel_stats.csv' | head -1)
head -12 $f | cut -c1-200 > $O/lw_kernel_stats.txt
rm -rf $out
cat $O/lw_kernel_stats.txt
