"""Host-side image ingest of PreprocessImage (embeddings.go:46-116): cv::resize INTER_LINEAR incl. OpenCV's INTER_AREA
switch at exact 2x2 decimation (:69), EXIF orientation as cv::imread applies it (:50), PreprocessImage end to end.
No GPU needed.  The resize vectors in tests/golden/resize_cv_linear.npz come from the published definition of OpenCV's
8-bit path (tests/golden/make_resize_golden.py), not from the engine; a few values are hand-derived literals."""
import io
import os

import numpy as np
import pytest
from PIL import Image, ImageOps

from imageclust_amd import _lib

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize_cv_linear.npz"))
CASES = sorted({k[:-4] for k in GOLD.files if k.endswith("_src")})


@pytest.mark.parametrize("name", CASES)
def test_resize_matches_opencv_definition(name):
    src, want = GOLD[name + "_src"], GOLD[name + "_dst"]
    got = _lib.resize_u8(src, want.shape[1], want.shape[0])
    assert np.array_equal(got, want), name


def test_resize_hand_derived_literals():
    # [0, 255] stretched to 3 pixels: 0 | f=0.5 -> (255*1024 >> 4) * 2048 >> 16 = 510, (510 + 2) >> 2 = 128 | clamp -> 255
    got = _lib.resize_u8(np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8), 3, 1)
    assert got[0, :, 0].tolist() == [0, 128, 255]
    # exact 2x2 decimation takes the INTER_AREA path: rounded mean of the block, (1+2+3+5+2)>>2 = 3 (bilinear would give 2 or 3 by position)
    src = np.array([[[1] * 3, [2] * 3], [[3] * 3, [5] * 3]], np.uint8)
    assert _lib.resize_u8(src, 1, 1)[0, 0].tolist() == [3, 3, 3]
    # constant images stay constant for every scale
    for (h, w) in [(5, 9), (448, 448), (1, 1), (223, 225)]:
        assert (_lib.resize_u8(np.full((h, w, 3), 77, np.uint8), 224, 224) == 77).all()


def _save(tmp_path, arr, name, **kw):
    p = tmp_path / name
    Image.fromarray(arr).save(p, **kw)
    return str(p)


@pytest.mark.parametrize("orient", [1, 2, 3, 4, 5, 6, 7, 8])
def test_exif_orientation_is_applied_like_imread(tmp_path, orient):
    """cv::imread rotates / mirrors by the EXIF orientation tag; Pillow's exif_transpose is the independent reference."""
    rng = np.random.default_rng(orient)
    arr = rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)
    im = Image.fromarray(arr)
    exif = Image.Exif()
    exif[0x0112] = orient
    buf = io.BytesIO()
    im.save(buf, "JPEG", quality=92, exif=exif.tobytes())
    p = tmp_path / "o.jpg"
    p.write_bytes(buf.getvalue())
    want = np.asarray(ImageOps.exif_transpose(Image.open(io.BytesIO(buf.getvalue()))).convert("RGB"))
    got = _lib.decode_image_file(str(p))
    assert got.shape == want.shape and np.array_equal(got, want)
    if orient >= 5:
        assert got.shape[:2] == (64, 40)  # transposing orientations swap width and height


def test_exif_garbage_is_ignored(tmp_path):
    arr = np.random.default_rng(0).integers(0, 256, (16, 24, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, "JPEG", quality=90)
    b = buf.getvalue()
    for payload in (b"Exif\x00\x00II*\x00\xff\xff\xff\x7f", b"Exif\x00\x00MM\x00*", b"Exif\x00\x00", b"http://ns.adobe.com/xap/1.0/\x00<x/>"):
        seg = b"\xff\xe1" + (len(payload) + 2).to_bytes(2, "big") + payload
        p = tmp_path / "g.jpg"
        p.write_bytes(b[:2] + seg + b[2:])
        got = _lib.decode_image_file(str(p))
        assert np.array_equal(got, np.asarray(Image.open(io.BytesIO(b)).convert("RGB")))


def test_preprocess_file_is_decode_resize_scale(tmp_path):
    """PreprocessImage end to end == decode -> resize -> RGB/255 NCHW, shape (1,3,224,224) (embeddings.go:96-108)."""
    rng = np.random.default_rng(3)
    arr = rng.integers(0, 256, (300, 260, 3), dtype=np.uint8)
    p = tmp_path / "x.ppm"
    with open(p, "wb") as f:
        f.write(b"P6\n260 300\n255\n" + arr.tobytes())
    blob = _lib.preprocess_file(str(p))
    img = _lib.load_image_224(str(p))
    assert blob.shape == (1, 3, 224, 224)
    assert np.array_equal(blob[0], (img.transpose(2, 0, 1).astype(np.float32) * np.float32(1.0 / 255.0)))
    with pytest.raises(_lib.ICLError):
        _lib.preprocess_file(str(tmp_path / "missing.jpg"))
