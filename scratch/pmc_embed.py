"""One ResNet50 forward of a batch of 256 random u8 images (bf16) for rocprofv3 --pmc runs on the conv kernels."""
import sys
sys.path.insert(0, ".")
import torch
from imageclust_amd import _lib
ctx = _lib.Context(0)
ctx.load_synthetic(1)
x = torch.randint(0, 255, (256, 224, 224, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((256, 2048), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
ctx.embed_u8_dev(x.data_ptr(), 256, out.data_ptr(), _lib.HEAD_POOLED, _lib.PREC_BF16)
ctx.sync() if hasattr(ctx, "sync") else torch.cuda.synchronize()
print("done", float(out.abs().sum()))
