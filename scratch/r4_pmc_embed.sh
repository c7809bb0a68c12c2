#!/bin/bash
# HBM-side traffic of the embedding kernels (FETCH_SIZE / WRITE_SIZE in separate passes) on a short single-stream embed-only run
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r04pmc; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$out/pmc_$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
  rm -rf $d
  ICL_EMBED_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/bench.py --embed-only --total-images 2560 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/$c.err || { echo "$c failed"; tail -3 $out/$c.err; exit 1; }
done
python3 $R/scratch/pmc_summary.py $out > $out/pmc_summary.json
rm -rf $out/pmc_fetch $out/pmc_write
python3 - <<PY
import json
d=json.load(open("$out/pmc_summary.json"))
for k,v in d.items():
    if isinstance(v,dict): print("%-50s x%-4d fetch(x2) %8.1f MB  write %8.1f MB"%(k[:50],v["launches"],2*v["FETCH_SIZE_KB_mean"]/1024,v["WRITE_SIZE_KB_mean"]/1024))
PY
