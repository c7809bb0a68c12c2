#!/bin/bash
# round-3 profile / bench collection (one GPU box): outputs under gpurun_out/r03/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
bash scratch/collect_profiles.sh r03 || exit 1
run() { name=$1; shift; timeout -k 10 500 python3 bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/steps.log; }
run bench_n100000
run bench_n10000 --total-images 10000
run bench_n250000 --total-images 250000 --steps 2 --no-cpu-baseline
run bench_embed_1m --embed-only --total-images 1000000 --steps 1 --no-cpu-baseline
run bench_n100000_fp32 --prec fp32 --no-cpu-baseline
run ab_dist_exact --ward-dist exact --steps 2 --no-cpu-baseline
run ab_overlap --overlap --steps 2 --no-cpu-baseline
bash scratch/layers_ab.sh 1 > $O/layers.log 2>&1; cp gpurun_out/layers_mode1.txt $O/embed_layers.txt
cat $O/steps.log
