#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2q; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --embed-only "$@" > $out/$name.out 2> $out/$name.err; echo "$name rc=$?" | tee -a $out/log5; wc -l $out/$name/runc/*_counter_collection.csv 2>/dev/null | tail -1; }
run e40k --total-images 40000
run e80k --total-images 80000
cat $out/log5
