"""PCIe-inclusive embed rate: icl_embed_u8 with HOST image and output buffers (pageable numpy memory, slabs of 4096 images)
against icl_embed_u8_dev on the same images resident in HBM.  usage: python scratch/pcie_rate.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from imageclust_amd import _lib
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = _lib.Context(0)
ctx.load_synthetic(1)
imgs = torch.empty(n * _lib.IMG_BYTES, dtype=torch.uint8, device="cuda")
ctx.synth_images_dev(20250217, 0, n, _lib.SYNTH_STRUCTURED, imgs.data_ptr())
ctx.sync()  # the engine's stream does not synchronise with torch's: the copy below must see finished images
E = torch.empty((n, 2048), dtype=torch.float32, device="cuda")
host = imgs.cpu().numpy()
for rep in range(2):
    ctx.embed_u8_dev(imgs.data_ptr(), n, E.data_ptr(), 2048, _lib.PREC_BF16)
    dev_ms = ctx.last_stage_ms()["embed_ms"]
for rep in range(2):
    t0 = time.perf_counter()
    out = ctx.embed_u8(host, _lib.HEAD_POOLED, _lib.PREC_BF16)
    host_ms = (time.perf_counter() - t0) * 1e3
ref = E.cpu().numpy()
if not np.array_equal(out, ref):
    bad = np.flatnonzero((out != ref).any(axis=1))
    print("MISMATCH rows", bad.size, "first", bad[:8], "last", bad[-8:], "max abs diff", float(np.abs(out - ref).max()), "nan", int(np.isnan(out).sum()), int(np.isnan(ref).sum()))
    out2 = ctx.embed_u8(host, _lib.HEAD_POOLED, _lib.PREC_BF16)
    print("host path repeatable:", np.array_equal(out, out2))
    ctx.embed_u8_dev(imgs.data_ptr(), n, E.data_ptr(), 2048, _lib.PREC_BF16)
    print("dev path repeatable:", np.array_equal(ref, E.cpu().numpy()))
print("n %d: resident %.1f ms (%.0f img/s), host buffers %.1f ms (%.0f img/s), %.1f MB in, %.1f MB out" %
      (n, dev_ms, n / dev_ms * 1e3, host_ms, n / host_ms * 1e3, host.nbytes / 1e6, out.nbytes / 1e6))
