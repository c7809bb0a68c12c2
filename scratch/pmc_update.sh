#!/bin/bash
# SQ counters of the batched update kernel (what do its chain waves wait for?)  usage: bash scratch/pmc_update.sh <N>
out=$GRAFT_REPO_ROOT/gpurun_out/r2s; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  ICL_WARD_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --kernel-include-regex ward_update_batch2 --output-format csv -d $out/$tag -- python3 $GRAFT_REPO_ROOT/scratch/scale_test.py $1 > $out/$tag.out 2> $out/$tag.err
  echo "$tag rc=$?"
  python3 - $out/$tag <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(float); n = defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print("  %-28s total %.4g over %d dispatches, mean %.4g" % (k, acc[k], n[k], acc[k] / max(n[k], 1)))
PY
  rm -rf $out/$tag
done
