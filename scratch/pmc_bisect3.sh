#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2q; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/$name -- python3 "$@" > $out/$name.out 2> $out/$name.err; echo "$name rc=$?" | tee -a $out/log3; tail -1 $out/$name.out | cut -c1-100; }
cat > /tmp/c.py <<'PY'
import sys, torch, numpy as np
sys.path.insert(0, sys.argv[1])
from imageclust_amd import _lib
ctx = _lib.Context(0)
n, d = int(sys.argv[2]), 64
print("ctx ok", flush=True)
E = torch.randn((n, d), device="cuda")
print("E ok", flush=True)
if len(sys.argv) > 3:
    ctx.ward_prepare(n, d); print("prepare ok", flush=True)
cid, r, nc = ctx.cluster_dev(E.data_ptr(), n, d, 5, 50); print("C ok", n, nc)
PY
run C60k /tmp/c.py $GRAFT_REPO_ROOT 60000 prep
run C100k /tmp/c.py $GRAFT_REPO_ROOT 100000 prep
cat $out/log3; cat $out/C100k.out
