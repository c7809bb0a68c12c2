"""Embed throughput against the internal batch size and the number of forward passes in flight (one process per setting of
ICL_EMBED_STREAMS, which is read once).  usage: python scratch/embed_batch_sweep.py N batch [batch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclust_amd import _lib
import torch

n = int(sys.argv[1])
ctx = _lib.Context(0)
ctx.load_synthetic(1)
imgs = torch.empty(n * _lib.IMG_BYTES, dtype=torch.uint8, device="cuda")
ctx.synth_images_dev(20250217, 0, n, _lib.SYNTH_STRUCTURED, imgs.data_ptr())
E = torch.empty((n, 2048), dtype=torch.float32, device="cuda")
ref = None
for b in [int(x) for x in sys.argv[2:]]:
    ctx.set_batch(b)
    for rep in range(2):
        ctx.embed_u8_dev(imgs.data_ptr(), n, E.data_ptr(), 2048, _lib.PREC_BF16)
        ms = ctx.last_stage_ms()["embed_ms"]
    s = float(E.double().sum())
    print("streams", os.environ.get("ICL_EMBED_STREAMS", "2"), "batch", b, "embed_ms %.1f" % ms, "img/s %.0f" % (n / ms * 1e3), "sum", s, flush=True)
