"""Can the exact distance build (vector fp32) hide behind the embedding (MFMA)?  Two contexts on one GPU, two host threads:
A embeds NE images, B computes distance rows [0, NR) of a random E -- alone, then both at once.
usage: python scratch/overlap_probe.py [NE] [NR]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageclust_amd import _lib
import torch

ne = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
A, B = _lib.Context(0), _lib.Context(0)
A.load_synthetic(1)
imgs = torch.empty(ne * _lib.IMG_BYTES, dtype=torch.uint8, device="cuda")
A.synth_images_dev(20250217, 0, ne, _lib.SYNTH_STRUCTURED, imgs.data_ptr())
A.sync()
EA = torch.empty((ne, 2048), dtype=torch.float32, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
EB = torch.randn((nr, 2048), generator=g, device="cuda")
off, cnt = _lib.ward_span(0, nr)
span = torch.empty(cnt, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
CH = 12800  # rows per distance call (100 tile rows)

def embed():
    A.embed_u8_dev(imgs.data_ptr(), ne, EA.data_ptr(), 2048, _lib.PREC_BF16)

def dist():
    for lo in range(0, nr, CH):
        hi = min(lo + CH, nr)
        o, c = _lib.ward_span(lo, hi)
        B.ward_distance_rows_dev(EB.data_ptr(), nr, 2048, lo, hi, span.data_ptr() + 4 * (o - off))
    B.sync()

def timed(fs):
    ts = [threading.Thread(target=f) for f in fs]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

embed(); dist()
te = min(timed([embed]) for _ in range(2))
td = min(timed([dist]) for _ in range(2))
tb = min(timed([embed, dist]) for _ in range(2))
print("embed %d images alone %.1f ms, distance rows [0,%d) alone %.1f ms, both at once %.1f ms (sum %.1f, max %.1f)" % (ne, te, nr, td, tb, te + td, max(te, td)), flush=True)
