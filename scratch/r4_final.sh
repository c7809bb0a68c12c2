#!/bin/bash
# round-4 profile / bench collection (one GPU box): outputs under gpurun_out/r04/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O $R/gpurun_out/r04a
cd $R
bash scratch/collect_profiles.sh r04 || exit 1
run() { name=$1; shift; timeout -k 10 500 python3 bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?" | tee -a $O/steps.log; }
run bench_n100000
run bench_n10000 --total-images 10000 --no-cpu-baseline
run bench_n250000 --total-images 250000 --steps 2 --no-cpu-baseline
run bench_embed_1m --embed-only --total-images 1000000 --steps 1 --no-cpu-baseline
run bench_n100000_fp32 --prec fp32 --steps 2 --no-cpu-baseline
ICL_FUSE=0 timeout -k 10 300 python3 bench.py --embed-only --total-images 102400 --steps 2 --no-cpu-baseline > $O/ab_embed_unfused.json 2> $O/ab_embed_unfused.err; echo "ab_embed_unfused rc=$?" | tee -a $O/steps.log
timeout -k 10 300 python3 bench.py --embed-only --total-images 102400 --steps 2 --no-cpu-baseline > $O/ab_embed_fused.json 2> $O/ab_embed_fused.err; echo "ab_embed_fused rc=$?" | tee -a $O/steps.log
bash scratch/r4_trace.sh 15 0 > $O/layers.log 2>&1; cp gpurun_out/r04a/kt_fuse15.txt $O/embed_kernels_fused.txt; cp gpurun_out/r04a/kt_fuse0.txt $O/embed_kernels_unfused.txt
cat $O/steps.log
for f in bench_n100000 bench_n10000 bench_n250000 bench_embed_1m bench_n100000_fp32 ab_embed_unfused ab_embed_fused; do python3 -c "
import json; j=json.load(open('$O/$f.json')); print('$f', j['value'], j['ms_per_step'], j.get('stages_ms_last_step'), j.get('parity'))"; done
