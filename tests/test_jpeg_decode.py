"""Image ingest (embeddings.go:50, gocv.IMRead -> libjpeg-turbo): the engine's baseline JPEG decoder must reproduce
libjpeg-turbo's pixels bit for bit (islow IDCT, fancy upsampling, fixed-point YCbCr->RGB).  Pillow bundles
libjpeg-turbo and is used here as the independent reference decoder.  Host-only code: runs without a GPU."""
import io

import numpy as np
import pytest
from PIL import Image, ImageFile

ImageFile.MAXBLOCK = 1 << 24  # Pillow's progressive / optimized encoder needs the whole file in one buffer

from imageclust_amd import _lib


def picture(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 17.0 + y / 31.0), 128 + 90 * np.cos(x / 11.0 - y / 23.0), (x * 255 // max(w - 1, 1) + y) % 256], -1)
    return np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.uint8)


def roundtrip(tmp_path, im, **save_kw):
    buf = io.BytesIO()
    im.save(buf, "JPEG", **save_kw)
    p = tmp_path / "t.jpg"
    p.write_bytes(buf.getvalue())
    ref = np.asarray(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))
    return _lib.decode_image_file(str(p)), ref


@pytest.mark.parametrize("size", [(64, 64), (224, 224), (37, 53), (300, 260), (17, 9), (8, 8), (1, 1), (481, 322)])
@pytest.mark.parametrize("sub", [0, 1, 2])
def test_matches_libjpeg_turbo(tmp_path, size, sub):
    for q in (35, 90, 100):
        got, ref = roundtrip(tmp_path, Image.fromarray(picture(*size, seed=q)), quality=q, subsampling=sub)
        assert got.shape == ref.shape and np.array_equal(got, ref), (size, sub, q)


def test_grayscale_optimized_tables_and_restart_markers(tmp_path):
    im = Image.fromarray(picture(150, 211, 1))
    got, ref = roundtrip(tmp_path, im.convert("L"), quality=80)
    assert np.array_equal(got, ref)
    got, ref = roundtrip(tmp_path, im, quality=75, optimize=True, subsampling=2)
    assert np.array_equal(got, ref)
    got, ref = roundtrip(tmp_path, im, quality=75, subsampling=2, restart_marker_blocks=3)
    assert np.array_equal(got, ref)
    got, ref = roundtrip(tmp_path, im, quality=75, subsampling=1, restart_marker_rows=1)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("size", [(64, 64), (224, 224), (37, 53), (300, 260), (17, 9), (1, 1), (481, 322)])
@pytest.mark.parametrize("sub", [0, 1, 2])
def test_progressive_matches_libjpeg_turbo(tmp_path, size, sub):
    """SOF2: DC first/refine, AC first/refine with EOB runs, one scan per component (what OpenCV's IMRead decodes for
    most product photos downloaded from the web)."""
    for q in (30, 85, 100):
        got, ref = roundtrip(tmp_path, Image.fromarray(picture(*size, seed=q + 1)), quality=q, subsampling=sub, progressive=True)
        assert got.shape == ref.shape and np.array_equal(got, ref), (size, sub, q)


def test_progressive_grayscale_optimized_and_restarts(tmp_path):
    im = Image.fromarray(picture(150, 211, 5))
    got, ref = roundtrip(tmp_path, im.convert("L"), quality=70, progressive=True)
    assert np.array_equal(got, ref)
    got, ref = roundtrip(tmp_path, im, quality=60, progressive=True, optimize=True, subsampling=2)
    assert np.array_equal(got, ref)
    got, ref = roundtrip(tmp_path, im, quality=75, progressive=True, subsampling=2, restart_marker_blocks=5)
    assert np.array_equal(got, ref)
    got, ref = roundtrip(tmp_path, im, quality=75, progressive=True, subsampling=1, restart_marker_rows=2)
    assert np.array_equal(got, ref)
    # smooth content: long EOB runs and many all-zero blocks
    flat = np.full((96, 160, 3), 200, np.uint8)
    flat[20:40, 30:90] = (10, 60, 250)
    got, ref = roundtrip(tmp_path, Image.fromarray(flat), quality=50, progressive=True)
    assert np.array_equal(got, ref)


def test_unsupported_and_corrupt_files(tmp_path):
    im = Image.fromarray(picture(64, 64, 2))
    buf = io.BytesIO()
    im.convert("CMYK").save(buf, "JPEG")
    p = tmp_path / "cmyk.jpg"
    p.write_bytes(buf.getvalue())
    with pytest.raises(_lib.ICLError) as ei:
        _lib.decode_image_file(str(p))
    assert ei.value.code == _lib.ICL_ERR_UNSUPPORTED and "failed to read image" in str(ei.value)
    buf = io.BytesIO()
    im.save(buf, "JPEG", progressive=True)
    pq = tmp_path / "prog_trunc.jpg"
    pq.write_bytes(buf.getvalue()[:150])
    with pytest.raises(_lib.ICLError):
        _lib.decode_image_file(str(pq))
    buf = io.BytesIO()
    im.save(buf, "JPEG")
    q = tmp_path / "trunc.jpg"
    q.write_bytes(buf.getvalue()[:200])
    with pytest.raises(_lib.ICLError) as ei:
        _lib.decode_image_file(str(q))
    assert ei.value.code == _lib.ICL_ERR_IO
    with pytest.raises(_lib.ICLError):
        _lib.decode_image_file(str(tmp_path / "missing.jpg"))


def test_load_image_224_identity_and_resize(tmp_path):
    im = picture(224, 224, 3)
    got, ref = roundtrip(tmp_path, Image.fromarray(im), quality=95, subsampling=0)
    assert np.array_equal(_lib.load_image_224(str(tmp_path / "t.jpg")), ref)  # 224x224: cv::resize is the identity
    big = picture(448, 448, 4)
    (tmp_path / "b.ppm").write_bytes(b"P6\n448 448\n255\n" + big.tobytes())
    out = _lib.load_image_224(str(tmp_path / "b.ppm"))
    # INTER_LINEAR at an exact 2:1 ratio samples the centre of each 2x2 block: the (rounded) mean of its 4 pixels
    box = big.reshape(224, 2, 224, 2, 3).astype(np.int32).sum((1, 3))
    assert np.abs(out.astype(np.int32) - (box + 2) // 4).max() <= 1
    const = np.full((50, 70, 3), 77, np.uint8)
    (tmp_path / "c.ppm").write_bytes(b"P6\n70 50\n255\n" + const.tobytes())
    assert (_lib.load_image_224(str(tmp_path / "c.ppm")) == 77).all()


# ---- hostile files (the decoder runs on uploaded / downloaded product images: embeddings.go:50) ----
def _jpeg_bytes(h=16, w=16, **kw):
    buf = io.BytesIO()
    Image.fromarray(picture(h, w, 3)).save(buf, "JPEG", **kw)
    return bytearray(buf.getvalue())


def _find_marker(b, m):
    i = 2
    while i + 4 <= len(b):
        assert b[i] == 0xFF
        if b[i + 1] == m:
            return i
        i += 2 + ((b[i + 2] << 8) | b[i + 3])
    raise AssertionError("marker %02x not found" % m)


@pytest.mark.parametrize("bits1", [3, 255])
def test_oversubscribed_huffman_table_is_rejected(tmp_path, bits1):
    """A DHT whose code-length counts are no prefix code (3 or 255 one-bit codes) must be refused, not indexed with."""
    b = _jpeg_bytes(quality=80)
    i = _find_marker(b, 0xC4)
    b[i + 5] = bits1  # bits[1] of the first table in the segment
    p = tmp_path / "bad_dht.jpg"
    p.write_bytes(bytes(b))
    with pytest.raises(_lib.ICLError, match="Bad Huffman table"):
        _lib.decode_image_file(str(p))


def test_truncated_scan_header_is_rejected(tmp_path):
    b = _jpeg_bytes(quality=80)
    i = _find_marker(b, 0xDA)
    cut = bytes(b[:i]) + b"\xff\xda\x00\x02"  # SOS with seglen == 2 as the last bytes of the file
    p = tmp_path / "cut_sos.jpg"
    p.write_bytes(cut)
    with pytest.raises(_lib.ICLError):
        _lib.decode_image_file(str(p))


def test_huge_frame_header_is_refused_without_allocating(tmp_path):
    b = _jpeg_bytes(quality=80)
    i = _find_marker(b, 0xC0)
    b[i + 5 : i + 9] = bytes([0x80, 0x00, 0x80, 0x00])  # H = W = 32768 in a 1 KB file: ~11 GB of buffers if believed
    p = tmp_path / "huge.jpg"
    p.write_bytes(bytes(b))
    with pytest.raises(_lib.ICLError, match="64 Mpixel"):
        _lib.decode_image_file(str(p))


def test_mutated_files_never_crash(tmp_path):
    """Byte-flip fuzz over headers and entropy data: every outcome is a decoded image or an ICLError."""
    rng = np.random.default_rng(7)
    base = [_jpeg_bytes(24, 40, quality=70), _jpeg_bytes(33, 17, quality=90, progressive=True, subsampling=2)]
    p = tmp_path / "m.jpg"
    for it in range(300):
        b = bytearray(base[it & 1])
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        if it % 7 == 0:
            b = b[: int(rng.integers(4, len(b)))]
        p.write_bytes(bytes(b))
        try:
            _lib.decode_image_file(str(p))
        except _lib.ICLError:
            pass
