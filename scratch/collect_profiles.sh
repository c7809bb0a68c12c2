#!/bin/bash
# Profiles behind bench.py's roofline object, taken THROUGH bench.py itself (the launch path of the timed region: hipGraph
# replay of the merge loop, two embed streams): one --kernel-trace --stats run and two separate --pmc passes
# (FETCH_SIZE / WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage (on the GPU box):
#   bash scratch/collect_profiles.sh <tag> [bench args...]      outputs -> gpurun_out/<tag>/
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
step() { # name, rocprof flags...
  name=$1; shift
  timeout -k 10 420 rocprofv3 "$@" --output-format csv -d $out/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "${BENCH_ARGS[@]}" > $out/$name.json 2> $out/$name.err
  rc=$?; echo "$name rc=$rc" | tee -a $out/steps.log
  [ $rc -eq 124 -o $rc -eq 137 ] && { echo "killed: stopping" | tee -a $out/steps.log; exit 1; }
  return 0
}
BENCH_ARGS=("$@")
step stats --kernel-trace --stats
step pmc_fetch --kernel-trace --pmc FETCH_SIZE
step pmc_write --kernel-trace --pmc WRITE_SIZE
python3 $GRAFT_REPO_ROOT/scratch/pmc_summary.py $out > $out/pmc_summary.json
# keep the summaries, drop the per-dispatch traces (tens of MB: gpurun copies back at most 64 MiB)
cp $out/stats/*/*_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
cat $out/steps.log
