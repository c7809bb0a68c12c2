"""BASELINE.json configs[3] and configs[4] at FULL size on one MI355X (the module name sorts last: these tests take ~260 GB
and ~160 GB of the 288 GB of HBM, so they run after every other module has released its context)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config4_250k_size_constrained_ward_full_size():
    """configs[4]: N = 250 000, D = 2048, minSize 5, maxSize 50 (k = 27 500, 222 500 merges) on ONE GPU: the 250 GB distance
    matrix with recycled rows / columns (ward.hip header) is resident in HBM.  No CPU restatement reaches this size: the
    size-independent properties of tests/ward_props.py, spot-checked merge values against the oracle's arithmetic, idempotence."""
    from imageclust_amd import _lib
    from tests.ward_props import check_full_size_run

    ctx = _lib.Context(0)
    try:
        merges, nc = check_full_size_run(ctx, 250000, 2048, 20250219, 5, 50)
        assert merges == 222500 and nc <= 27500
    finally:
        ctx.close()


def test_config3_one_million_images_embed_only():
    """configs[3]: 1 000 000 synthetic images (150 GB of u8 pixels generated on the device) -> 2048-d pooled embeddings, bf16
    batch 256, no clustering.  Every row finite; 512 sampled rows equal, bit for bit, a separate batch-256 embedding of the
    same images (the forward pass is batch-invariant and image i depends on nothing but its own pixels)."""
    import torch

    from imageclust_amd import _lib
    from imageclust_amd import distributed as D

    n, head = 1000000, _lib.HEAD_POOLED
    ctx = _lib.Context(0)
    d_img = d_E = d_s = d_o = None
    try:
        ctx.load_synthetic(1)
        d_img = ctx.malloc(n * _lib.IMG_BYTES)
        d_E = ctx.malloc(n * head * 4)
        ctx.synth_images_dev(20250217, 0, n, _lib.SYNTH_STRUCTURED, d_img)
        ctx.embed_u8_dev(d_img, n, d_E, head, _lib.PREC_BF16)
        ctx.sync()
        E = torch.as_tensor(D._DeviceSpan(d_E, n * head), device="cuda").view(n, head)
        assert bool(torch.isfinite(E).all())
        assert float(E.abs().max()) > 0
        rng = np.random.default_rng(3)
        idx = np.unique(np.r_[0, n - 1, 255, 256, n - 257, rng.integers(0, n, 512)])[:512]
        d_s = ctx.malloc(len(idx) * _lib.IMG_BYTES)
        d_o = ctx.malloc(len(idx) * head * 4)
        for q, i in enumerate(idx.tolist()):
            ctx.synth_images_dev(20250217, i, 1, _lib.SYNTH_STRUCTURED, d_s + q * _lib.IMG_BYTES)
        ctx.embed_u8_dev(d_s, len(idx), d_o, head, _lib.PREC_BF16)
        ctx.sync()
        got = E[torch.from_numpy(idx).cuda()].cpu().numpy()
        want = np.empty((len(idx), head), np.float32)
        ctx.d2h(want, d_o)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    finally:
        for p in (d_img, d_E, d_s, d_o):
            if p:
                ctx.free(p)
        ctx.close()
