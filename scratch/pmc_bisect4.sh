#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2q; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/$name -- python3 "$@" > $out/$name.out 2> $out/$name.err; echo "$name rc=$?" | tee -a $out/log4; tail -1 $out/$name.out | cut -c1-100; }
cat > /tmp/d.py <<'PY'
import sys, torch
sys.path.insert(0, sys.argv[1])
from imageclust_amd import _lib
ctx = _lib.Context(0)
print("ctx", flush=True)
ctx.load_synthetic(1); print("model", flush=True)
n = int(sys.argv[2])
x = torch.empty(n * _lib.IMG_BYTES, dtype=torch.uint8, device="cuda"); print("alloc", flush=True)
ctx.synth_images_dev(20250217, 0, n, _lib.SYNTH_STRUCTURED, x.data_ptr()); ctx.sync(); print("synth", flush=True)
E = torch.empty((n, 2048), dtype=torch.float32, device="cuda")
m = min(n, 2560)
ctx.embed_u8_dev(x.data_ptr(), m, E.data_ptr(), 2048, _lib.PREC_BF16); print("embed", m, flush=True)
PY
run D30k /tmp/d.py $GRAFT_REPO_ROOT 30000
run D100k /tmp/d.py $GRAFT_REPO_ROOT 100000
cat $out/log4; cat $out/D100k.out
