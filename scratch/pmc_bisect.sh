#!/bin/bash
# Where does `rocprofv3 --pmc` die at large N?  Tiny scripts, one suspect each.
out=$GRAFT_REPO_ROOT/gpurun_out/r2q; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/$name -- python3 "$@" > $out/$name.out 2> $out/$name.err; echo "$name rc=$?" | tee -a $out/log; tail -2 $out/$name.out | cut -c1-200; }
cat > /tmp/a.py <<'PY'
import torch, sys
x = torch.empty(4 << 30, dtype=torch.uint8, device="cuda"); x.fill_(1); torch.cuda.synchronize(); print("A ok", int(x[123]))
PY
cat > /tmp/b.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from imageclust_amd import _lib
ctx = _lib.Context(0)
n = 20000
x = torch.empty(n * _lib.IMG_BYTES, dtype=torch.uint8, device="cuda")
ctx.synth_images_dev(20250217, 0, n, _lib.SYNTH_STRUCTURED, x.data_ptr()); ctx.sync(); print("B ok", int(x[5]))
PY
cat > /tmp/c.py <<'PY'
import sys, torch, numpy as np
sys.path.insert(0, sys.argv[1])
from imageclust_amd import _lib
ctx = _lib.Context(0)
n, d = int(sys.argv[2]), 64
E = torch.randn((n, d), device="cuda")
cid, r, nc = ctx.cluster_dev(E.data_ptr(), n, d, 5, 50); print("C ok", n, nc)
PY
run A /tmp/a.py
run B /tmp/b.py $GRAFT_REPO_ROOT
run C3k /tmp/c.py $GRAFT_REPO_ROOT 3000
run C20k /tmp/c.py $GRAFT_REPO_ROOT 20000
run Cgraph0 /tmp/c.py $GRAFT_REPO_ROOT 20000
cat $out/log
