#!/bin/bash
# per-layer table of one forward pass (single stream, fused graph) -> gpurun_out/r04c/embed_layers.txt ; packed-rate micro-benchmark
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c; mkdir -p $O
out=$O/lay; rm -rf $out
ICL_EMBED_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/bench.py --embed-only --total-images 2560 --steps 1 --warmup 1 --no-cpu-baseline > $O/lay.log 2>&1
f=$(find $out -name '*kernel_trace.csv' | head -1)
python3 $R/scratch/layer_report.py $f > $O/embed_layers.txt
rm -rf $out
cat $O/embed_layers.txt
hipcc -O3 --offload-arch=gfx950 -o /tmp/pk_rate $R/scratch/pk_rate_bench.hip && /tmp/pk_rate > $O/pk_rate.txt 2>&1
cat $O/pk_rate.txt
