#!/bin/bash
# Which launch path does `rocprofv3 --pmc` fail on?  One run per switch.  A run that is killed by its
# timeout ends the chain (no further GPU step after a kill); a run that merely crashes does not.
out=$GRAFT_REPO_ROOT/gpurun_out/r2b; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() { # name, env assignments...
  name=$1; shift
  ( export "$@"; timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_$name -- python3 $GRAFT_REPO_ROOT/bench.py --total-images 2560 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_$name.out 2> $out/pmc_$name.err )
  rc=$?; echo "probe $name rc=$rc" | tee -a $out/probe.log
  [ $rc -eq 124 -o $rc -eq 137 ] && { echo "killed: stopping the chain" | tee -a $out/probe.log; exit 1; }
  return 0
}
run both_off ICL_WARD_GRAPH=0 ICL_EMBED_STREAMS=1
run graph_only ICL_WARD_GRAPH=1 ICL_EMBED_STREAMS=1
run streams_only ICL_WARD_GRAPH=0 ICL_EMBED_STREAMS=2
run both_on ICL_WARD_GRAPH=1 ICL_EMBED_STREAMS=2
