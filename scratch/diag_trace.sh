#!/bin/bash
# usage (GPU box): bash scratch/diag_trace.sh <tag> <N>   with ICL_* switches in the environment: first update-launch durations
tag=$1; n=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dt_$tag
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/dt_$tag -- python3 $GRAFT_REPO_ROOT/scratch/scale_test.py $n > /tmp/dt_$tag.log 2>&1
f=$(ls /tmp/dt_$tag/*/*_kernel_trace.csv | head -1)
echo "$tag: $(python3 $GRAFT_REPO_ROOT/scratch/first_updates.py $f 5)" | tee -a $GRAFT_REPO_ROOT/gpurun_out/diag_trace.log
rm -rf /tmp/dt_$tag
