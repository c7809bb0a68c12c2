#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2q; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; ( export $1; shift; timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $out/$name.out 2> $out/$name.err ); echo "$name rc=$?" | tee -a $out/log2; }
run e10k_s2 ICL_EMBED_STREAMS=2 --embed-only --total-images 10000
run e10k_s1 ICL_EMBED_STREAMS=1 --embed-only --total-images 10000
run f10k_s1 ICL_EMBED_STREAMS=1 --total-images 10000
run f100k_s1 ICL_EMBED_STREAMS=1 --total-images 100000
cat $out/log2
