"""Generates tests/golden/resize_cv_linear.npz: expected outputs of cv::resize(..., INTER_LINEAR) on 8-bit images
(embeddings.go:69), written from the published definition of OpenCV's 8-bit path -- NOT by calling the engine:

  for every destination index d along an axis with scale = src/dst:
      f = (d + 0.5) * scale - 0.5 (evaluated in float32 like OpenCV), s = floor(f), f -= s,
      s < 0 -> (s, f) = (0, 0);  s >= src-1 -> (s, f) = (src-1, 0)
      coefficients (short): a0 = cvRound((1-f) * 2048), a1 = cvRound(f * 2048)          (INTER_RESIZE_COEF_BITS = 11)
  horizontal pass (HResizeLinear<uchar,int,short>):  buf[x] = S[s] * a0 + S[s+1] * a1                     (int, scale 2048)
  vertical pass   (VResizeLinear<uchar,int,short>):  dst = (((b0 * (buf0 >> 4)) >> 16) + ((b1 * (buf1 >> 4)) >> 16) + 2) >> 2
  exact 2x2 decimation switches to INTER_AREA (ResizeAreaFast): dst = (a + b + c + d + 2) >> 2

Each case below also carries values worked out by hand in the comments; the test asserts those literally.
Run: python tests/golden/make_resize_golden.py"""
import os

import numpy as np


def cv_round(x):  # cvRound = lrint: round half to even
    return int(np.rint(np.float64(x)))


def axis(dn, sn):
    out = []
    scale = np.float64(sn) / np.float64(dn)
    for d in range(dn):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= sn - 1:
            s, f = sn - 1, np.float32(0)
        out.append((s, cv_round((np.float32(1) - f) * np.float32(2048)), cv_round(f * np.float32(2048))))
    return out


def resize_linear(src, dw, dh):
    sh, sw, cn = src.shape
    if sw == 2 * dw and sh == 2 * dh:
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xs, ys = axis(dw, sw), axis(dh, sh)
    dst = np.zeros((dh, dw, cn), np.uint8)
    for y, (sy, b0, b1) in enumerate(ys):
        sy1 = min(sy + 1, sh - 1)
        for x, (sx, a0, a1) in enumerate(xs):
            sx1 = min(sx + 1, sw - 1)
            for c in range(cn):
                r0 = int(src[sy, sx, c]) * a0 + int(src[sy, sx1, c]) * a1
                r1 = int(src[sy1, sx, c]) * a0 + int(src[sy1, sx1, c]) * a1
                dst[y, x, c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
    return dst


if __name__ == "__main__":
    rng = np.random.default_rng(20250217)
    cases = {}
    # (a) 1 x 2 -> 1 x 3 upscale of [0, 255]: by hand d=0: f<0 -> 0; d=1: f=0.5 -> (0*1024+255*1024)=261120, >>4 = 16320,
    #     *2048>>16 = 510, (510+0+2)>>2 = 128; d=2: s clamps to 1 -> 255.  Expected [0, 128, 255] in every channel.
    cases["up_1x2_to_1x3"] = (np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8), 3, 1)
    # (b) 3:2 downscale 6 x 6 -> 4 x 4, (c) 2:1 area path 8 x 6 -> 4 x 3, (d) upscale 5 x 7 -> 224 x 224 (the reference's target),
    # (e) one-pixel image, (f) one-row image, (g) large non-square photo-like downscale to 224 x 224
    cases["down_3to2_6x6"] = (rng.integers(0, 256, (6, 6, 3), dtype=np.uint8), 4, 4)
    cases["area_2to1_6x8"] = (rng.integers(0, 256, (6, 8, 3), dtype=np.uint8), 4, 3)
    cases["up_7x5_to_224"] = (rng.integers(0, 256, (7, 5, 3), dtype=np.uint8), 224, 224)
    cases["one_pixel"] = (np.array([[[7, 100, 250]]], np.uint8), 224, 224)
    cases["one_row"] = (rng.integers(0, 256, (1, 9, 3), dtype=np.uint8), 5, 3)
    y, x = np.mgrid[0:300, 0:260]
    photo = np.stack([128 + 100 * np.sin(x / 17.0 + y / 31.0), 128 + 90 * np.cos(x / 11.0 - y / 23.0), (x * 255 // 259 + y) % 256], -1)
    cases["photo_300x260_to_224"] = (np.clip(photo + rng.normal(0, 12, photo.shape), 0, 255).astype(np.uint8), 224, 224)
    cases["area_448_to_224"] = (rng.integers(0, 256, (448, 448, 3), dtype=np.uint8), 224, 224)
    out = {}
    for name, (src, dw, dh) in cases.items():
        out[name + "_src"] = src
        out[name + "_dst"] = resize_linear(src, dw, dh)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_cv_linear.npz"), **out)
    print({k: v.shape for k, v in out.items()})
