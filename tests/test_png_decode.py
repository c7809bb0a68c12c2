"""PNG ingest (imageclust_amd/csrc/png_decode.hip: own inflate + PNG reader) against Pillow on a generated corpus, bit for bit,
plus hostile files.  gocv.IMRead(path, IMReadColor) (embeddings.go:50) reads PNG through OpenCV's libpng reader: 3 x 8-bit
colour, alpha stripped, 16-bit samples cut to their high byte, small greys scaled, palettes looked up.  Host-only code: no GPU."""
import io
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from imageclust_amd import _lib


def decode(tmp_path, data, name="x.png"):
    p = tmp_path / name
    p.write_bytes(data)
    return _lib.decode_image_file(str(p))


def png_bytes(img, **kw):
    b = io.BytesIO()
    img.save(b, format="PNG", **kw)
    return b.getvalue()


def chunk(t, body):
    return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)


def raw_png(w, h, depth, ctype, scanlines, level=6, extra=b"", interlace=0, split=1):
    """A PNG built by hand: scanlines = list of (filter byte, row bytes)."""
    raw = b"".join(bytes([f]) + r for f, r in scanlines)
    z = zlib.compress(raw, level)
    step = max(1, len(z) // split)
    idat = b"".join(chunk(b"IDAT", z[i:i + step]) for i in range(0, len(z), step))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra + idat + chunk(b"IEND", b""))


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1"])
@pytest.mark.parametrize("size", [(1, 1), (7, 5), (64, 33), (301, 203)])
def test_pillow_corpus_bit_exact(tmp_path, mode, size):
    rng = np.random.default_rng(hash((mode, size)) % (2 ** 32))
    w, h = size
    if mode == "P":
        img = Image.fromarray(rng.integers(0, 200, (h, w), dtype=np.uint8), "P")
        img.putpalette(rng.integers(0, 256, 768, dtype=np.uint8).tobytes())
    elif mode == "1":
        img = Image.fromarray((rng.random((h, w)) < 0.5), "1")
    else:
        nch = {"RGB": 3, "RGBA": 4, "L": 1, "LA": 2}[mode]
        a = rng.integers(0, 256, (h, w, nch), dtype=np.uint8)
        # smooth regions too, so that the encoder picks Sub / Up / Average / Paeth filters and long matches
        a[: h // 2] = (np.arange(w)[None, :, None] + np.arange(h // 2)[:, None, None]) % 256
        img = Image.fromarray(a[:, :, 0] if nch == 1 else a, mode)
    want = np.asarray(img.convert("RGB"))
    for kw in ({}, {"compress_level": 0}, {"compress_level": 9, "optimize": True}, {"compress_level": 1}):
        got = decode(tmp_path, png_bytes(img, **kw))
        assert got.shape == want.shape and np.array_equal(got, want), (mode, size, kw)


def test_every_filter_type_and_bit_depth_by_hand(tmp_path):
    rng = np.random.default_rng(4)
    w, h = 37, 11
    # RGB 8-bit, one filter type per row (encoded by hand so that all five are exercised whatever an encoder would pick)
    px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    lines, prev = [], np.zeros(w * 3, np.int32)
    for y in range(h):
        cur = px[y].reshape(-1).astype(np.int32)
        ft = y % 5
        left = np.r_[np.zeros(3, np.int32), cur[:-3]]
        ul = np.r_[np.zeros(3, np.int32), prev[:-3]]
        if ft == 0:
            enc = cur
        elif ft == 1:
            enc = cur - left
        elif ft == 2:
            enc = cur - prev
        elif ft == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            enc = cur - pred
        lines.append((ft, (enc & 255).astype(np.uint8).tobytes()))
        prev = cur
    assert np.array_equal(decode(tmp_path, raw_png(w, h, 8, 2, lines, split=5)), px)
    # 16-bit RGBA: the high byte of every sample, alpha dropped (png_set_strip_16 / png_set_strip_alpha)
    px16 = rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)
    lines = [(0, px16[y].astype(">u2").tobytes()) for y in range(h)]
    assert np.array_equal(decode(tmp_path, raw_png(w, h, 16, 6, lines)), (px16[:, :, :3] >> 8).astype(np.uint8))
    # 16-bit grey, 2- and 4-bit grey (scaled by 255 / (2^bits - 1)), 2-bit palette
    g16 = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    got = decode(tmp_path, raw_png(w, h, 16, 0, [(0, g16[y].astype(">u2").tobytes()) for y in range(h)]))
    assert np.array_equal(got, np.repeat((g16 >> 8).astype(np.uint8)[:, :, None], 3, 2))
    for depth in (2, 4):
        v = rng.integers(0, 1 << depth, (h, w), dtype=np.uint8)
        rows = []
        for y in range(h):
            bits = "".join(format(int(t), "0%db" % depth) for t in v[y])
            bits += "0" * (-len(bits) % 8)
            rows.append((0, int(bits, 2).to_bytes(len(bits) // 8, "big")))
        got = decode(tmp_path, raw_png(w, h, depth, 0, rows))
        assert np.array_equal(got[:, :, 0], v * (255 // ((1 << depth) - 1))) and np.array_equal(got[:, :, 0], got[:, :, 2])
        if depth == 2:
            pal = rng.integers(0, 256, (4, 3), dtype=np.uint8)
            got = decode(tmp_path, raw_png(w, h, 2, 3, rows, extra=chunk(b"PLTE", pal.tobytes()) + chunk(b"tRNS", b"\x00\x80")))
            assert np.array_equal(got, pal[v])


def test_load_image_224_and_embed_path_accept_png(tmp_path):
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (300, 260, 3), dtype=np.uint8)
    p = tmp_path / "a.png"
    Image.fromarray(a, "RGB").save(p)
    q = tmp_path / "a.ppm"
    with open(q, "wb") as f:
        f.write(b"P6\n260 300\n255\n" + a.tobytes())
    assert np.array_equal(_lib.load_image_224(str(p)), _lib.load_image_224(str(q)))  # same pixels, same resize


def test_hostile_png_files_are_rejected_not_crashed(tmp_path):
    rng = np.random.default_rng(0)
    good = png_bytes(Image.fromarray(rng.integers(0, 256, (40, 50, 3), dtype=np.uint8), "RGB"))
    assert decode(tmp_path, good).shape == (40, 50, 3)

    def bad(data, name):
        with pytest.raises(_lib.ICLError):
            decode(tmp_path, data, name)

    for cut in (8, 20, 33, 60, len(good) // 2, len(good) - 13, len(good) - 1):
        bad(good[:cut], "cut%d.png" % cut)
    for pos in (12, 29, 45, len(good) // 2, len(good) - 20):  # flipped bits: chunk type, IHDR CRC, compressed data
        b = bytearray(good)
        b[pos] ^= 0x5A
        bad(bytes(b), "flip%d.png" % pos)
    row = (0, bytes(30))
    bad(raw_png(10, 1, 8, 2, [row], interlace=1), "adam7.png")                      # interlaced, but the stream holds one progressive row: 31 bytes for 34
    bad(raw_png(10, 1, 8, 2, [row], interlace=2), "interlace2.png")                 # unknown interlace method
    bad(raw_png(10, 1, 8, 2, [(7, bytes(30))]), "filter7.png")                      # unknown filter type
    bad(raw_png(10, 1, 8, 2, [(0, bytes(29))]), "short.png")                        # too few bytes inflated
    bad(raw_png(10, 1, 8, 2, [row, row]), "long.png")                               # too many
    bad(raw_png(10, 1, 3, 2, [row]), "depth3.png")                                  # bit depth not in the standard
    bad(raw_png(100000, 100000, 8, 2, [row]), "huge.png")                           # 10^10 pixels
    bad(raw_png(10, 1, 8, 3, [(0, bytes(10))]), "nopal.png")                        # palette image without PLTE
    bad(raw_png(10, 1, 8, 3, [(0, bytes([5] * 10))], extra=chunk(b"PLTE", bytes(9))), "palidx.png")  # index 5 of a 3-entry palette
    sig, rest = good[:8], good[8:]
    bad(sig + chunk(b"XYZW", b"abc") + rest, "noihdr.png")                          # a chunk in front of IHDR
    bad(good[:33] + chunk(b"ABCD", b"critical") + good[33:], "critical.png")        # unknown CRITICAL chunk (upper-case first letter)
    assert decode(tmp_path, good[:33] + chunk(b"abCD", b"ancillary") + good[33:]).shape == (40, 50, 3)  # unknown ancillary chunk: skipped
    # corrupt DEFLATE: a stored block whose length check fails, a reserved block type, a distance in front of the output
    hdr = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 10, 1, 8, 2, 0, 0, 0))
    for z in (b"\x78\x01\x01\x1f\x00\xe0\x00" + bytes(31), b"\x78\x01\x07" + bytes(40), b"\x78\x01\x63\x00\x02\x00" + bytes(10)):
        bad(hdr + chunk(b"IDAT", z + b"\0\0\0\0") + chunk(b"IEND", b""), "defl%d.png" % len(z))
    # random garbage behind a valid signature never crashes
    for i in range(40):
        bad(b"\x89PNG\r\n\x1a\n" + rng.integers(0, 256, int(rng.integers(1, 400)), dtype=np.uint8).tobytes(), "rnd%d.png" % i)


def adam7_png(samples, depth, ctype, extra=b"", filters=(0, 1, 2, 3, 4)):
    """An Adam7-interlaced PNG built by hand from samples[h][w][channels] (ints below 2**depth): the seven reduced images, each
    with its own filtered scanlines (the filter types cycle through `filters`); empty passes are left out, as the standard says."""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    raw = b""
    k = 0
    for xs, ys, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
        sub = samples[ys::dy, xs::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for row in sub:
            vals = row.reshape(-1)
            if depth == 16:
                cur = np.stack([vals >> 8, vals & 255], 1).reshape(-1).astype(np.int32)
            elif depth == 8:
                cur = vals.astype(np.int32)
            else:
                bits = "".join(format(int(v), "0%db" % depth) for v in vals)
                bits += "0" * (-len(bits) % 8)
                cur = np.array([int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)], np.int32)
            if prev is None:
                prev = np.zeros_like(cur)
            left = np.r_[np.zeros(bpp, np.int32), cur[:-bpp]]
            ul = np.r_[np.zeros(bpp, np.int32), prev[:-bpp]]
            ft = filters[k % len(filters)]
            k += 1
            if ft == 0:
                enc = cur
            elif ft == 1:
                enc = cur - left
            elif ft == 2:
                enc = cur - prev
            elif ft == 3:
                enc = cur - ((left + prev) >> 1)
            else:
                pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
                enc = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            raw += bytes([ft]) + bytes((enc & 255).astype(np.uint8))
            prev = cur
    z = zlib.compress(raw, 6)
    idat = chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:])
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1)) + extra + idat + chunk(b"IEND", b"")


@pytest.mark.parametrize("size", [(1, 1), (3, 2), (8, 8), (9, 17), (37, 29), (130, 67)])
def test_adam7_interlaced_files(tmp_path, size):
    """cv::imread reads interlaced PNGs like any other: the seven passes are de-interleaved into the same image.  Hand-built files
    (Pillow cannot write them) in every colour type; Pillow READS them, and must agree with this decoder and with the source."""
    w, h = size
    rng = np.random.default_rng(w * 1000 + h)
    pal = rng.integers(0, 256, 48, dtype=np.uint8).tobytes()
    cases = [("rgb8", 8, 2, 3, b""), ("rgba16", 16, 6, 4, b""), ("grey2", 2, 0, 1, b""), ("pal4", 4, 3, 1, chunk(b"PLTE", pal)), ("ga8", 8, 4, 2, b""),
             ("grey16", 16, 0, 1, b""), ("grey1", 1, 0, 1, b"")]
    for name, depth, ctype, ch, extra in cases:
        s = rng.integers(0, 1 << depth, (h, w, ch), dtype=np.int64)
        data = adam7_png(s, depth, ctype, extra)
        got = decode(tmp_path, data, name + ".png")
        if ctype == 3:
            lut = np.frombuffer(pal, np.uint8).reshape(16, 3)
            want = lut[s[:, :, 0]]
        elif ctype in (0, 4):
            g = s[:, :, 0]
            g8 = (g >> 8) if depth == 16 else g * (255 // ((1 << depth) - 1))
            want = np.repeat(g8[:, :, None], 3, 2).astype(np.uint8)
        else:
            want = ((s[:, :, :3] >> 8) if depth == 16 else s[:, :, :3]).astype(np.uint8)
        assert got.shape == want.shape and np.array_equal(got, want), (name, size)
        if depth <= 8 and ctype in (2, 3):  # and Pillow's reader on the same file (its 16-bit / grey conversions differ from cv::imread's by design)
            assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert("RGB")), want), (name, size)
