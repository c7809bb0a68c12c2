"""GPU parity tests of the cross-layer fusions of the forward pass (imageclust_amd/csrc/resnet_fused.h) against the CPU
oracle, through the C-ABI: stem_pool_kernel (conv0 + BN + ReLU + maxpool in one launch) and bneck56_kernel (one whole
stage-1 bottleneck in one launch).  The reference fuses layers inside OpenCV-DNN's Net.Forward
(/root/reference/internal/embeddings/embeddings.go:141); the oracle restates the layers one by one.

Tolerances: fp32 <= 1e-4 * max(1, max|ref|) (north_star); bf16 kernels against the oracle run on bf16-rounded operands
with every intermediate tensor rounded to bf16 where the engine stores one (t1, t2): <= 1.2e-2 * max(1, max|ref|), i.e.
bf16's 2^-8 output rounding plus the rare one-ulp flip of a rounded intermediate."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from imageclust_amd import _lib

    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    c = L.Context(0)
    c.load_synthetic(1)
    yield c
    c.close()


def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32)


def ref_conv_hw(x_nhwc, w, scale, shift, stride, pad, relu):
    """conv + folded BN (+ ReLU) by the oracle's icl_ref_conv2d on [B][H][W][C] (rectangular images allowed)."""
    B, H, Wd, Cin = x_nhwc.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (Wd + 2 * pad - k) // stride + 1
    out = np.empty((B, Ho, Wo, Cout), np.float32)
    for b in range(B):
        x = np.ascontiguousarray(x_nhwc[b].transpose(2, 0, 1))
        y = np.zeros((Cout, Ho, Wo), np.float32)
        O.lib().icl_ref_conv2d(x, Cin, H, Wd, np.ascontiguousarray(w), None, Cout, k, stride, pad, y, Ho, Wo)
        y = (y * scale[:, None, None] + shift[:, None, None]).transpose(1, 2, 0)
        out[b] = np.maximum(y, 0) if relu else y
    return out


def conv0_of_blob(blob):
    """conv0's weights and folded BatchNorm out of the ICLW blob (include/icl_model_format.h: 80-byte header, then W, gamma,
    beta, mean, var; conv0 carries no bias)."""
    p = np.frombuffer(blob, np.float32, offset=80)
    eps = float(np.frombuffer(blob, np.float32, count=1, offset=8)[0])
    w = p[:64 * 3 * 49].reshape(64, 3, 7, 7)
    g, be, mu, var = (p[64 * 147 + i * 64:64 * 147 + (i + 1) * 64].astype(np.float64) for i in range(4))
    s = g / np.sqrt(var + eps)
    return w.copy(), s.astype(np.float32), (be - mu * s).astype(np.float32)


def ref_maxpool(y_nhwc):
    B, H, Wd, Cc = y_nhwc.shape
    out = np.empty((B, H // 2, Wd // 2, Cc), np.float32)
    for b in range(B):
        x = np.ascontiguousarray(y_nhwc[b].transpose(2, 0, 1))
        o = np.zeros((Cc, H // 2, Wd // 2), np.float32)
        O.lib().icl_ref_maxpool3x3s2(x, Cc, H, Wd, o, H // 2, Wd // 2)
        out[b] = o.transpose(1, 2, 0)
    return out


@pytest.mark.parametrize("B", [1, 3])
def test_stem_pool_fused_matches_oracle(ctx, L, B):
    """conv0 7x7/2 + BN + ReLU + maxpool 3x3/2 in one launch: fp32 at 1e-4 against the oracle's conv -> BN -> ReLU -> maxpool on
    the RGB/255 input (embeddings.go:96); bf16 against the oracle on bf16-rounded image and weights.  B = 3 makes a persistent
    workgroup walk more than one (image, strip) unit only on small grids; image borders (strip 0 / 7, tile 0 / 13) are in every
    image."""
    blob = L.synthetic_blob(1)
    w, sc, sh = conv0_of_blob(blob)
    imgs = np.concatenate([L.synth_images(20250217, 3, B - 1, L.SYNTH_STRUCTURED), L.synth_images(7, 11, 1, L.SYNTH_NOISE)]) if B > 1 \
        else L.synth_images(7, 11, 1, L.SYNTH_NOISE)
    x = imgs.astype(np.float32) * np.float32(1.0 / 255.0)
    r = ref_maxpool(ref_conv_hw(x, w, sc, sh, 2, 3, True))
    y = ctx.stem_pool(imgs, L.PREC_FP32)
    assert y.shape == (B, 56, 56, 64)
    assert np.abs(y - r).max() <= 1e-4 * max(1.0, np.abs(r).max())
    # the bf16 stem folds the BatchNorm scale into its weights before rounding them (icl_model_load_blob), the shift stays fp32
    rb = ref_maxpool(ref_conv_hw(bf16_round(x), bf16_round(w * sc[:, None, None, None]), np.ones(64, np.float32), sh, 2, 3, True))
    yb = ctx.stem_pool(imgs, L.PREC_BF16)
    assert np.abs(yb - rb).max() <= 1.2e-2 * max(1.0, np.abs(rb).max())


def _bneck_weights(rng, cin, ds):
    he = lambda co, ci, k: (rng.standard_normal((co, ci, k, k)) * np.sqrt(2.0 / (ci * k * k))).astype(np.float32)
    bn = lambda c: (rng.uniform(0.5, 1.5, c).astype(np.float32), (0.1 * rng.standard_normal(c)).astype(np.float32))
    p = {"w1": he(64, cin, 1), "bn1": bn(64), "w2": he(64, 64, 3), "bn2": bn(64), "w3": he(256, 64, 1), "bn3": bn(256)}
    if ds:
        p["wds"], p["bnds"] = he(256, cin, 1), bn(256)
    return p


def _bneck_ref(x, p, ds):
    """The oracle's layer-by-layer bottleneck on bf16-rounded operands.  As the engine does for stage 1 (icl_model_load_blob),
    every BatchNorm scale is folded into the weights BEFORE they are rounded to bf16 (w' = bf16(w * scale)), the shift stays in
    fp32; t1 and t2 are rounded to bf16, the form in which they feed the next matrix product."""
    r = bf16_round
    one = lambda c: np.ones(c, np.float32)
    fold = lambda w, bn: r(w * bn[0][:, None, None, None])
    xb = r(x)
    t1 = r(ref_conv_hw(xb, fold(p["w1"], p["bn1"]), one(64), p["bn1"][1], 1, 0, True))
    t2 = r(ref_conv_hw(t1, fold(p["w2"], p["bn2"]), one(64), p["bn2"][1], 1, 1, True))
    if ds:
        y = ref_conv_hw(t2, fold(p["w3"], p["bn3"]), one(256), p["bn3"][1] + p["bnds"][1], 1, 0, False) + \
            ref_conv_hw(xb, fold(p["wds"], p["bnds"]), one(256), np.zeros(256, np.float32), 1, 0, False)
    else:
        y = ref_conv_hw(t2, fold(p["w3"], p["bn3"]), one(256), p["bn3"][1], 1, 0, False) + xb
    return np.maximum(y, 0)


# (B, H, W): ResNet50's 56x56; widths that are no multiple of the 14-column strip (a masked last strip, a single strip);
# heights that leave ragged last steps; more images than image groups are exercised by B = 5 on the 56-wide case only when
# the device has < 20 CUs, so B x (H + 1) not a multiple of 8 is what the small shapes are for
BNECK_SHAPES = [(2, 56, 56), (3, 9, 20), (1, 7, 14), (5, 5, 33), (2, 16, 3)]


@pytest.mark.parametrize("ds", [False, True], ids=["identity", "downsample"])
@pytest.mark.parametrize("shape", BNECK_SHAPES, ids=lambda s: "b%d_h%d_w%d" % s)
def test_bottleneck56_fused_matches_oracle(ctx, L, shape, ds):
    B, H, Wd = shape
    cin = 64 if ds else 256
    rng = np.random.default_rng(B * 1000 + H * 10 + Wd + (1 if ds else 0))
    p = _bneck_weights(rng, cin, ds)
    x = rng.standard_normal((B, H, Wd, cin)).astype(np.float32)
    y = ctx.bottleneck56(x, p["w1"][:, :, 0, 0], p["bn1"], p["w2"], p["bn2"], p["w3"][:, :, 0, 0], p["bn3"],
                         p["wds"][:, :, 0, 0] if ds else None, p.get("bnds"))
    r = _bneck_ref(x, p, ds)
    assert y.shape == r.shape
    err = np.abs(y - r)
    assert err.max() <= 1.2e-2 * max(1.0, np.abs(r).max()), (err.max(), np.unravel_index(err.argmax(), err.shape))
    # the median error is that of one bf16 output rounding: a wrong tap, channel or neighbour would not hide in the maximum alone
    assert np.median(err) <= 2e-3 * max(1.0, np.abs(r).max())


def test_bottleneck56_rejects_other_widths(ctx, L):
    rng = np.random.default_rng(0)
    p = _bneck_weights(rng, 128, False)
    with pytest.raises(L.ICLError) as ei:
        ctx.bottleneck56(np.zeros((1, 4, 4, 128), np.float32), p["w1"][:, :, 0, 0], p["bn1"], p["w2"], p["bn2"], p["w3"][:, :, 0, 0], p["bn3"])
    assert ei.value.code == L.ICL_ERR_UNSUPPORTED


def test_fused_forward_equals_layer_by_layer(L):
    """The whole bf16 forward pass with the fusions (default) against the same pass launched layer by layer (ICL_FUSE=0 in a
    child process: the switch is read once per process).  Stem + maxpool in one launch (mask 1) is bit-identical by construction;
    the direct-patch bf16 stem and the fused bottlenecks (mask 15, the default) fold the BatchNorm scales into the bf16 weights
    and use another MFMA shape (16x16x32 instead of 32x32x16: another summation order inside a k-step), so there the embeddings
    agree to bf16 rounding noise, far inside the bf16 path's own distance from fp32 (3e-2 bound)."""
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from imageclust_amd import _lib as L; c = L.Context(0); c.load_synthetic(1); "
            "im = L.synth_images(20250217, 100, 6, L.SYNTH_STRUCTURED); "
            "np.savez(sys.argv[1], b=c.embed_u8(im, L.HEAD_POOLED, L.PREC_BF16), f=c.embed_u8(im, L.HEAD_POOLED, L.PREC_FP32))" % root)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for mask in ("0", "1", "15"):
            path = os.path.join(td, "e%s.npz" % mask)
            subprocess.check_call([sys.executable, "-c", code, path], env=dict(os.environ, ICL_FUSE=mask))
            out[mask] = dict(np.load(path))
    assert np.array_equal(out["0"]["f"], out["15"]["f"])  # fp32: the fused stem is bit-identical, nothing else changes
    assert np.array_equal(out["0"]["b"], out["1"]["b"])  # bf16 with only stem + maxpool fused: bit-identical
    f = out["0"]["f"]
    d_unf = np.linalg.norm(out["0"]["b"] - f, axis=1) / np.linalg.norm(f, axis=1)
    d_fus = np.linalg.norm(out["15"]["b"] - f, axis=1) / np.linalg.norm(f, axis=1)
    d_ab = np.linalg.norm(out["15"]["b"] - out["0"]["b"], axis=1) / np.linalg.norm(f, axis=1)
    print("bf16 vs fp32 rel L2: layer by layer", d_unf, "fused", d_fus, "fused vs layer by layer", d_ab)
    assert d_fus.max() < 3e-2 and d_ab.max() < 1e-2
