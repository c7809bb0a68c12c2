"""Checks the CPU oracle's ResNet50 restatement (oracle/resnet_ref.c) against torch-CPU operators -- an independent
implementation of the same published ONNX operator definitions.  The reference holds no fixture at this boundary
(parity unpinned), so this is the strongest pin available offline."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle as O


def test_conv_bn_pool_ops_match_torch():
    rng = np.random.default_rng(0)
    L = O.lib()
    for (cin, cout, k, s, p, h) in [(3, 8, 7, 2, 3, 30), (16, 8, 3, 1, 1, 9), (8, 16, 1, 2, 0, 10), (8, 8, 1, 1, 0, 7)]:
        x = rng.standard_normal((cin, h, h)).astype(np.float32)
        w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        ho = (h + 2 * p - k) // s + 1
        y = np.zeros((cout, ho, ho), np.float32)
        L.icl_ref_conv2d(x, cin, h, h, w, b.ctypes.data, cout, k, s, p, y, ho, ho)
        t = F.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), s, p)[0]
        assert np.abs(y - t.numpy()).max() < 1e-4 * max(1.0, np.abs(t.numpy()).max())
        g, be, mu = (rng.standard_normal(cout).astype(np.float32) for _ in range(3))
        var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        res = rng.standard_normal(y.shape).astype(np.float32)
        y2 = y.copy()
        L.icl_ref_bn_act(y2, cout, ho * ho, g, be, mu, var, 1e-5, res.ctypes.data, 1)
        tb = F.batch_norm(torch.from_numpy(y)[None], torch.from_numpy(mu), torch.from_numpy(var), torch.from_numpy(g),
                          torch.from_numpy(be), False, 0.0, 1e-5)[0]
        tb = torch.relu(tb + torch.from_numpy(res))
        assert np.abs(y2 - tb.numpy()).max() < 1e-5 * max(1.0, float(tb.abs().max()))
    x = rng.standard_normal((4, 12, 12)).astype(np.float32)
    y = np.zeros((4, 6, 6), np.float32)
    L.icl_ref_maxpool3x3s2(x, 4, 12, 12, y, 6, 6)
    assert np.array_equal(y, F.max_pool2d(torch.from_numpy(x)[None], 3, 2, 1)[0].numpy())


def torch_resnet50_v1(blob: np.ndarray, img_u8: np.ndarray):
    """Independent torch-CPU forward of the ICLW blob: Gluon resnet50_v1 (stride on the first 1x1)."""
    hdr = blob[:80]
    eps = float(np.frombuffer(hdr[8:12].tobytes(), np.float32)[0])
    has_bias = hdr[16:80]
    p = np.frombuffer(blob[80:].tobytes(), np.float32)
    pos = [0]

    def take(n):
        v = torch.from_numpy(p[pos[0]:pos[0] + n].copy()).double()
        pos[0] += n
        return v

    idx = [0]

    def conv_bn(x, cin, cout, k, s, pad, relu, res=None):
        w = take(cout * cin * k * k).reshape(cout, cin, k, k)
        b = take(cout) if has_bias[idx[0]] else None
        g, be, mu, var = take(cout), take(cout), take(cout), take(cout)
        idx[0] += 1
        y = F.conv2d(x, w, b, s, pad)
        y = F.batch_norm(y, mu, var, g, be, False, 0.0, eps)
        if res is not None:
            y = y + res
        return torch.relu(y) if relu else y

    x = torch.from_numpy(img_u8.astype(np.float32) * np.float32(1.0 / 255.0)).permute(2, 0, 1)[None].double()
    x = conv_bn(x, 3, 64, 7, 2, 3, True)
    x = F.max_pool2d(x, 3, 2, 1)
    cin = 64
    for s, nb in enumerate([3, 4, 6, 3]):
        cout, mid = 256 << s, (256 << s) // 4
        for b in range(nb):
            stride = 2 if (b == 0 and s > 0) else 1
            t = conv_bn(x, cin, mid, 1, stride, 0, True)
            t = conv_bn(t, mid, mid, 3, 1, 1, True)
            # canonical blob order: c1, c2, c3, then ds -> read c3's tensors before the downsample's
            w3 = take(cout * mid).reshape(cout, mid, 1, 1)
            b3 = take(cout) if has_bias[idx[0]] else None
            g3, be3, mu3, var3 = take(cout), take(cout), take(cout), take(cout)
            idx[0] += 1
            res = conv_bn(x, cin, cout, 1, stride, 0, False) if b == 0 else x
            y = F.batch_norm(F.conv2d(t, w3, b3), mu3, var3, g3, be3, False, 0.0, eps)
            x = torch.relu(y + res)
            cin = cout
    pooled = x.mean(dim=(2, 3))[0]
    fcw = take(1000 * 2048).reshape(1000, 2048)
    fcb = take(1000)
    return pooled.numpy(), (fcw @ pooled + fcb).numpy()


@pytest.fixture(scope="module")
def blob():
    from imageclust_amd import _lib

    return _lib.synthetic_blob(1)


def test_full_forward_matches_torch_fp64(blob):
    from imageclust_amd import _lib

    img = _lib.synth_images(20250217, 3, 1, _lib.SYNTH_STRUCTURED)[0]
    pooled, dense = O.resnet50_forward(blob, img)
    tp, td = torch_resnet50_v1(blob, img)
    assert np.isfinite(pooled).all() and pooled.std() > 1e-3
    assert np.abs(pooled - tp).max() <= 1e-4 * max(1.0, np.abs(tp).max())
    assert np.abs(dense - td).max() <= 1e-4 * max(1.0, np.abs(td).max())


def test_preprocess_is_rgb_over_255_nchw():
    from imageclust_amd import _lib

    img = _lib.synth_images(1, 0, 1)[0]
    out = np.zeros((3, 224, 224), np.float32)
    O.lib().icl_ref_preprocess_rgb_u8(img, out)
    assert np.array_equal(out, (img.astype(np.float32) * np.float32(1 / 255.0)).transpose(2, 0, 1))
    got = np.zeros((3, 224, 224), np.float32)
    assert _lib.load().icl_preprocess_u8(img.ctypes.data, got.ctypes.data) == 0
    assert np.array_equal(got, out)
