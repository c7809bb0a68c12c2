// jpeg_decode.hip -- host-only baseline JPEG decoder for the image-ingest step of the hot path.
//
// Replaces gocv.IMRead(imagePath, IMReadColor) in PreprocessImage
//   (/root/reference/internal/embeddings/embeddings.go:50), i.e. OpenCV imgcodecs -> libjpeg(-turbo) with its default
// settings: integer "islow" IDCT, "fancy" (triangle) chroma upsampling and the fixed-point YCbCr->RGB tables.  Those
// three algorithms are public (IJG libjpeg: jidctint.c, jdsample.c, jdcolor.c) and are restated here so that decoded
// pixels are bit-identical to libjpeg-turbo's (checked against Pillow's bundled libjpeg-turbo in
// tests/test_jpeg_decode.py).  Supported: 8-bit baseline / extended sequential (SOF0, SOF1) and PROGRESSIVE (SOF2:
// spectral selection + successive approximation, ITU T.81 annex G) Huffman streams, interleaved or one scan per
// component, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0 sampling, restart intervals, JFIF / Adobe-transform markers.
// Every scan decodes into per-component coefficient arrays; dequantisation + IDCT run once after the last scan.
// Arithmetic coding, lossless, 12-bit and CMYK return ICL_ERR_UNSUPPORTED.
#include "icl_common.h"

#include <cstring>
#include <new>
#include <vector>

namespace {

struct huff_table {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    uint16_t fast[512]; // 9-bit lookahead: (length << 8) | symbol, 0 = not resolvable in 9 bits
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memset(fast, 0, sizeof fast);
        code = 0;
        k = 0;
        for (int l = 1; l <= 9; ++l) {
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                const int lo = code << (9 - l), hi = lo + (1 << (9 - l));
                for (int c = lo; c < hi; ++c) fast[c] = (uint16_t)((l << 8) | vals[k]);
            }
            code <<= 1;
        }
    }
};

struct component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int wblocks = 0, hblocks = 0; // padded to whole MCUs
    int dw = 0, dh = 0;           // downsampled_width / _height (real samples)
    int pred = 0;
    std::vector<uint8_t> plane;   // wblocks*8 x hblocks*8
    std::vector<int16_t> coefs;   // wblocks*hblocks blocks of 64, natural order, NOT dequantised
};

struct bit_reader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void fill()
    {
        while (nbits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    if (p < end && *p == 0x00) ++p;                 // stuffed zero
                    else { hit_marker = true; --p; b = 0; }         // a real marker: feed zeros from here on
                }
            }
            acc |= (uint32_t)b << (24 - nbits);
            nbits += 8;
        }
    }
    int peek(int n) { fill(); return (int)(acc >> (32 - n)); }
    void skip(int n) { acc <<= n; nbits -= n; }
    int get(int n)
    {
        if (n == 0) return 0;
        const int v = peek(n);
        skip(n);
        return v;
    }
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
};

inline int huff_decode(bit_reader &br, const huff_table &t)
{
    const int look = br.peek(9);
    const uint16_t f = t.fast[look];
    if (f) {
        br.skip(f >> 8);
        return f & 0xff;
    }
    int code = br.peek(16), l = 10;
    for (; l <= 16; ++l) {
        const int c = code >> (16 - l);
        if (t.maxcode[l] >= 0 && c <= t.maxcode[l] && c >= t.mincode[l]) {
            br.skip(l);
            return t.vals[t.valptr[l] + c - t.mincode[l]];
        }
    }
    return -1;
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// IJG jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2.  coef is dequantized, natural order.
void idct_islow(const int *coef, uint8_t *out, int stride)
{
    constexpr int CB = 13, P1 = 2;
    constexpr int F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
                  F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    auto descale = [](long x, int n) { return (int)((x + (1L << (n - 1))) >> n); };
    int ws[64];
    for (int c = 0; c < 8; ++c) {
        const int *in = coef + c;
        int *w = ws + c;
        if (!(in[8] | in[16] | in[24] | in[32] | in[40] | in[48] | in[56])) {
            const int dc = in[0] * (1 << P1);
            for (int r = 0; r < 8; ++r) w[8 * r] = dc;
            continue;
        }
        long z2 = in[16], z3 = in[48];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        z2 = in[0];
        z3 = in[32];
        long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56];
        tmp1 = in[40];
        tmp2 = in[24];
        tmp3 = in[8];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298;
        tmp1 *= F2_053;
        tmp2 *= F3_072;
        tmp3 *= F1_501;
        z1 *= -F0_899;
        z2 *= -F2_562;
        z3 *= -F1_961;
        z4 *= -F0_390;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        w[0] = descale(tmp10 + tmp3, CB - P1);
        w[56] = descale(tmp10 - tmp3, CB - P1);
        w[8] = descale(tmp11 + tmp2, CB - P1);
        w[48] = descale(tmp11 - tmp2, CB - P1);
        w[16] = descale(tmp12 + tmp1, CB - P1);
        w[40] = descale(tmp12 - tmp1, CB - P1);
        w[24] = descale(tmp13 + tmp0, CB - P1);
        w[32] = descale(tmp13 - tmp0, CB - P1);
    }
    for (int r = 0; r < 8; ++r) {
        const int *w = ws + 8 * r;
        uint8_t *o = out + (size_t)r * stride;
        long z2 = w[2], z3 = w[6];
        long z1 = (z2 + z3) * F0_541;
        long tmp2 = z1 + z3 * (-F1_847), tmp3 = z1 + z2 * F0_765;
        long tmp0 = ((long)w[0] + w[4]) * (1L << CB), tmp1 = ((long)w[0] - w[4]) * (1L << CB);
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7];
        tmp1 = w[5];
        tmp2 = w[3];
        tmp3 = w[1];
        z1 = tmp0 + tmp3;
        z2 = tmp1 + tmp2;
        z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298;
        tmp1 *= F2_053;
        tmp2 *= F3_072;
        tmp3 *= F1_501;
        z1 *= -F0_899;
        z2 *= -F2_562;
        z3 *= -F1_961;
        z4 *= -F0_390;
        z3 += z5;
        z4 += z5;
        tmp0 += z1 + z3;
        tmp1 += z2 + z4;
        tmp2 += z2 + z3;
        tmp3 += z1 + z4;
        constexpr int S = CB + P1 + 3;
        // range_limit[(x) & RANGE_MASK] of libjpeg == clamp(x + 128) for every value a legal stream can produce
        o[0] = clamp8(descale(tmp10 + tmp3, S) + 128);
        o[7] = clamp8(descale(tmp10 - tmp3, S) + 128);
        o[1] = clamp8(descale(tmp11 + tmp2, S) + 128);
        o[6] = clamp8(descale(tmp11 - tmp2, S) + 128);
        o[2] = clamp8(descale(tmp12 + tmp1, S) + 128);
        o[5] = clamp8(descale(tmp12 - tmp1, S) + 128);
        o[3] = clamp8(descale(tmp13 + tmp0, S) + 128);
        o[4] = clamp8(descale(tmp13 - tmp0, S) + 128);
    }
}

} // namespace

#define ICL_JPEG_MAX_PIXELS (64LL << 20)
static int jpeg_decode_impl(icl_ctx *ctx, const uint8_t *data, size_t len, const char *path, std::vector<uint8_t> &rgb, int &W, int &H, int &orient);

// Decodes a JPEG file held in memory to interleaved RGB.  rgb is resized to w*h*3.  No C++ exception may cross the C ABI
// (cgo / ctypes would std::terminate the host process): allocation failures become status codes here.
// orient receives the EXIF orientation tag (1..8; 1 when absent): cv::imread applies it after decoding (embeddings.go:50
// passes IMReadColor without IMREAD_IGNORE_ORIENTATION), the caller does the same (resnet.hip apply_exif_orientation).
int icl_jpeg_decode(icl_ctx *ctx, const uint8_t *data, size_t len, const char *path, std::vector<uint8_t> &rgb, int &W, int &H, int &orient)
{
    orient = 1;
    try {
        return jpeg_decode_impl(ctx, data, len, path, rgb, W, H, orient);
    } catch (const std::bad_alloc &) {
        return icl_fail(ctx, ICL_ERR_NOMEM, "failed to read image: %s. Out of host memory while decoding", path);
    } catch (...) {
        return icl_fail(ctx, ICL_ERR_IO, "failed to read image: %s. Decoder error", path);
    }
}

// EXIF (APP1 "Exif\0\0" + TIFF header): orientation tag 0x0112 of IFD0, 1..8; anything malformed reads as 1.
static int exif_orientation(const uint8_t *s, size_t sl)
{
    if (sl < 14 || memcmp(s, "Exif\0\0", 6) != 0) return 1;
    const uint8_t *t = s + 6;
    const size_t tl = sl - 6;
    const bool le = t[0] == 'I' && t[1] == 'I', be = t[0] == 'M' && t[1] == 'M';
    if (!le && !be) return 1;
    auto u16 = [&](size_t o) -> unsigned { return le ? (unsigned)(t[o] | (t[o + 1] << 8)) : (unsigned)((t[o] << 8) | t[o + 1]); };
    auto u32 = [&](size_t o) -> unsigned {
        return le ? (unsigned)t[o] | ((unsigned)t[o + 1] << 8) | ((unsigned)t[o + 2] << 16) | ((unsigned)t[o + 3] << 24)
                  : ((unsigned)t[o] << 24) | ((unsigned)t[o + 1] << 16) | ((unsigned)t[o + 2] << 8) | (unsigned)t[o + 3];
    };
    if (u16(2) != 42) return 1;
    const size_t ifd = u32(4);
    if (ifd + 2 > tl) return 1;
    const unsigned nent = u16(ifd);
    for (unsigned e = 0; e < nent; ++e) {
        const size_t o = ifd + 2 + (size_t)e * 12;
        if (o + 12 > tl) return 1;
        if (u16(o) == 0x0112) {
            const unsigned v = u16(o + 8); // type SHORT, count 1: the value sits in the first two bytes of the value field
            return (u16(o + 2) == 3 && v >= 1 && v <= 8) ? (int)v : 1;
        }
    }
    return 1;
}

static int jpeg_decode_impl(icl_ctx *ctx, const uint8_t *data, size_t len, const char *path, std::vector<uint8_t> &rgb, int &W, int &H, int &orient)
{
    auto fail = [&](int code, const char *what) { return icl_fail(ctx, code, "failed to read image: %s. %s", path, what); };
    if (len < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail(ICL_ERR_IO, "Not a JPEG stream");
    uint16_t qt[4][64];
    bool qt_ok[4] = {false, false, false, false};
    huff_table dc[4], ac[4];
    component comp[3];
    int ncomp = 0, restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, nscans = 0;
    bool have_sof = false, adobe = false, progressive = false;
    int adobe_transform = -1;
    size_t pos = 2;
    W = H = 0;
    while (pos + 4 <= len) {
        if (data[pos] != 0xFF) { ++pos; continue; }
        const int m = data[pos + 1];
        if (m == 0xFF) { ++pos; continue; }
        pos += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > len) break;
        const size_t seglen = ((size_t)data[pos] << 8) | data[pos + 1];
        if (seglen < 2 || pos + seglen > len) return fail(ICL_ERR_IO, "The image file might be corrupt or unreadable");
        const uint8_t *s = data + pos + 2;
        const size_t sl = seglen - 2;
        if (m == 0xDB) { // DQT
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > sl) return fail(ICL_ERR_IO, "Bad quantization table");
                for (int k = 0; k < 64; ++k) {
                    qt[tq][zigzag[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
                    i += pq ? 2 : 1;
                }
                qt_ok[tq] = true;
            }
        } else if (m == 0xC4) { // DHT
            size_t i = 0;
            while (i + 17 <= sl) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                if (tc > 1 || th > 3) return fail(ICL_ERR_IO, "Bad Huffman table");
                huff_table &t = tc ? ac[th] : dc[th];
                int total = 0;
                t.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = s[i + l]; total += t.bits[l]; }
                i += 17;
                if (total > 256 || i + total > sl) return fail(ICL_ERR_IO, "Bad Huffman table");
                // the counts must form a prefix code (as IJG jdhuff.c checks): at every length the codes handed out so
                // far fit in l bits -- otherwise build()'s lookahead index runs past fast[512] (over-subscribed table)
                for (int l = 1, code = 0; l <= 16; ++l) {
                    code += t.bits[l];
                    if (code > (1 << l)) return fail(ICL_ERR_IO, "Bad Huffman table");
                    code <<= 1;
                }
                memcpy(t.vals, s + i, (size_t)total);
                i += total;
                t.present = true;
                t.build();
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { // SOF0 / SOF1 / SOF2
            if (have_sof) return fail(ICL_ERR_IO, "Second frame header");
            if (sl < 6 || s[0] != 8) return fail(ICL_ERR_UNSUPPORTED, "Only 8-bit JPEG is decoded");
            progressive = m == 0xC2;
            H = (s[1] << 8) | s[2];
            W = (s[3] << 8) | s[4];
            ncomp = s[5];
            if ((ncomp != 1 && ncomp != 3) || sl < (size_t)(6 + 3 * ncomp) || W <= 0 || H <= 0 || W > 32768 || H > 32768)
                return fail(ICL_ERR_UNSUPPORTED, "Only 1- or 3-component JPEG is decoded");
            // sizes come from the file: bound what they make us allocate (coefficients + planes + RGB, ~11 B per pixel)
            if ((int64_t)W * H > ICL_JPEG_MAX_PIXELS) return fail(ICL_ERR_UNSUPPORTED, "JPEG larger than 64 Mpixel is not decoded");
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = s[6 + 3 * c];
                comp[c].h = s[7 + 3 * c] >> 4;
                comp[c].v = s[7 + 3 * c] & 15;
                comp[c].tq = s[8 + 3 * c];
            }
            if (ncomp == 1) comp[0].h = comp[0].v = 1;
            if (ncomp == 3) {
                const bool ok = comp[1].h == 1 && comp[1].v == 1 && comp[2].h == 1 && comp[2].v == 1 && (comp[0].h == 1 || comp[0].h == 2) &&
                                (comp[0].v == 1 || comp[0].v == 2) && !(comp[0].h == 1 && comp[0].v == 2);
                if (!ok) return fail(ICL_ERR_UNSUPPORTED, "Only 4:4:4, 4:2:2 and 4:2:0 chroma sampling is decoded");
            }
            for (int c = 0; c < ncomp; ++c) {
                hmax = std::max(hmax, comp[c].h);
                vmax = std::max(vmax, comp[c].v);
            }
            mcux = (W + 8 * hmax - 1) / (8 * hmax);
            mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            for (int c = 0; c < ncomp; ++c) {
                component &k = comp[c];
                k.wblocks = mcux * k.h;
                k.hblocks = mcuy * k.v;
                k.dw = (W * k.h + hmax - 1) / hmax;
                k.dh = (H * k.v + vmax - 1) / vmax;
                k.coefs.assign((size_t)k.wblocks * k.hblocks * 64, 0);
            }
            have_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail(ICL_ERR_UNSUPPORTED, "Lossless / hierarchical / arithmetic-coded JPEG is not decoded by this build");
        } else if (m == 0xDD) {
            if (sl >= 2) restart = (s[0] << 8) | s[1];
        } else if (m == 0xE1) {
            if (orient == 1) orient = exif_orientation(s, sl); // the first APP1/Exif segment decides, as in OpenCV's ExifReader
        } else if (m == 0xEE) {
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; }
        } else if (m == 0xDA) { // SOS: one scan (a baseline file has one or ncomp of them, a progressive file many)
            if (!have_sof) return fail(ICL_ERR_IO, "Scan before frame header");
            if (sl < 1) return fail(ICL_ERR_IO, "Bad scan header");
            const int ns = s[0];
            if (ns < 1 || ns > ncomp || sl < (size_t)(1 + 2 * ns + 3)) return fail(ICL_ERR_IO, "Bad scan header");
            int sc[3];
            for (int i = 0; i < ns; ++i) {
                int ci = -1;
                for (int c = 0; c < ncomp; ++c)
                    if (comp[c].id == s[1 + 2 * i]) ci = c;
                if (ci < 0) return fail(ICL_ERR_IO, "Bad scan component");
                comp[ci].td = s[2 + 2 * i] >> 4;
                comp[ci].ta = s[2 + 2 * i] & 15;
                sc[i] = ci;
            }
            const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
            if (progressive) {
                const bool ok = Ss <= Se && Se <= 63 && Al <= 13 && (Ss == 0 ? Se == 0 : ns == 1) && (Ah == 0 || Ah == Al + 1);
                if (!ok) return fail(ICL_ERR_IO, "Bad progressive scan parameters");
            } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) {
                return fail(ICL_ERR_IO, "Bad sequential scan parameters");
            }
            for (int i = 0; i < ns; ++i) {
                const component &k = comp[sc[i]];
                const bool need_dc = Ss == 0 && Ah == 0, need_ac = Se > 0;
                if (k.td > 3 || k.ta > 3 || (need_dc && !dc[k.td].present) || (need_ac && !ac[k.ta].present)) return fail(ICL_ERR_IO, "Missing table");
                comp[sc[i]].pred = 0;
            }
            bit_reader br{data + pos + seglen, data + len};
            int eobrun = 0;
            // one block of one scan into the coefficient array (T.81 F.2.2 sequential, G.1.2 progressive; the
            // refinement pass follows the structure of IJG jdphuff.c decode_mcu_AC_refine)
            auto decode_block = [&](component &k, int16_t *cf) -> bool {
                if (!progressive) {
                    const int t = huff_decode(br, dc[k.td]);
                    if (t < 0 || t > 15) return false;
                    k.pred += t ? extend(br.get(t), t) : 0;
                    cf[0] = (int16_t)k.pred;
                    for (int i = 1; i < 64;) {
                        const int rs = huff_decode(br, ac[k.ta]);
                        if (rs < 0) return false;
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r == 15) { i += 16; continue; }
                            break;
                        }
                        i += r;
                        if (i > 63) return false;
                        cf[zigzag[i]] = (int16_t)extend(br.get(sz), sz);
                        ++i;
                    }
                    return true;
                }
                if (Ss == 0) {
                    if (Ah == 0) { // DC first
                        const int t = huff_decode(br, dc[k.td]);
                        if (t < 0 || t > 15) return false;
                        k.pred += t ? extend(br.get(t), t) : 0;
                        cf[0] = (int16_t)(k.pred * (1 << Al));
                    } else if (br.get(1)) { // DC refinement: one more bit
                        cf[0] = (int16_t)(cf[0] | (1 << Al));
                    }
                    return true;
                }
                if (Ah == 0) { // AC first
                    if (eobrun > 0) { --eobrun; return true; }
                    for (int i = Ss; i <= Se;) {
                        const int rs = huff_decode(br, ac[k.ta]);
                        if (rs < 0) return false;
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) {
                            if (r == 15) { i += 16; continue; }
                            eobrun = (1 << r) - 1;
                            if (r) eobrun += br.get(r);
                            break;
                        }
                        i += r;
                        if (i > Se) return false;
                        cf[zigzag[i]] = (int16_t)(extend(br.get(sz), sz) * (1 << Al));
                        ++i;
                    }
                    return true;
                }
                // AC refinement
                const int p1 = 1 << Al, m1 = -(1 << Al);
                int i = Ss;
                auto refine = [&](int16_t &c) {
                    if (br.get(1) && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
                };
                if (eobrun == 0) {
                    for (; i <= Se; ++i) {
                        const int rs = huff_decode(br, ac[k.ta]);
                        if (rs < 0) return false;
                        int r = rs >> 4, sv = rs & 15;
                        if (sv) {
                            if (sv != 1) return false;
                            sv = br.get(1) ? p1 : m1;
                        } else if (r != 15) { // EOBr: the rest of this block (and eobrun-1 more) only gets correction bits
                            eobrun = 1 << r;
                            if (r) eobrun += br.get(r);
                            break;
                        }
                        // skip r ZERO-history coefficients (ZRL: 16), refining the non-zero ones passed on the way
                        for (; i <= Se; ++i) {
                            int16_t &c = cf[zigzag[i]];
                            if (c != 0) refine(c);
                            else if (--r < 0) break;
                        }
                        if (sv) {
                            if (i > Se) return false;
                            cf[zigzag[i]] = (int16_t)sv;
                        }
                    }
                }
                if (eobrun > 0) {
                    for (; i <= Se; ++i) {
                        int16_t &c = cf[zigzag[i]];
                        if (c != 0) refine(c);
                    }
                    --eobrun;
                }
                return true;
            };
            auto do_restart = [&]() -> bool {
                const uint8_t *q = br.p; // byte-align, expect RSTn
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                if (q + 1 >= br.end) return false;
                br.p = q + 2;
                br.reset();
                for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                eobrun = 0;
                return true;
            };
            int rst_left = restart;
            if (ns == 1) { // non-interleaved: the component's own block raster, one block per "MCU"
                component &k = comp[sc[0]];
                const int bw = (k.dw + 7) / 8, bh = (k.dh + 7) / 8;
                for (int by = 0; by < bh; ++by)
                    for (int bx = 0; bx < bw; ++bx) {
                        if (restart && rst_left == 0) {
                            if (!do_restart()) return fail(ICL_ERR_IO, "Missing restart marker");
                            rst_left = restart;
                        }
                        if (!decode_block(k, k.coefs.data() + ((size_t)by * k.wblocks + bx) * 64)) return fail(ICL_ERR_IO, "Corrupt JPEG data");
                        if (restart) --rst_left;
                    }
            } else {
                for (int my = 0; my < mcuy; ++my)
                    for (int mx = 0; mx < mcux; ++mx) {
                        if (restart && rst_left == 0) {
                            if (!do_restart()) return fail(ICL_ERR_IO, "Missing restart marker");
                            rst_left = restart;
                        }
                        for (int i = 0; i < ns; ++i) {
                            component &k = comp[sc[i]];
                            for (int by = 0; by < k.v; ++by)
                                for (int bx = 0; bx < k.h; ++bx)
                                    if (!decode_block(k, k.coefs.data() + ((size_t)(my * k.v + by) * k.wblocks + (mx * k.h + bx)) * 64))
                                        return fail(ICL_ERR_IO, "Corrupt JPEG data");
                        }
                        if (restart) --rst_left;
                    }
            }
            ++nscans;
            // continue at the marker that ended the entropy-coded segment
            const uint8_t *q = br.p;
            while (q + 1 < data + len && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) ++q;
            pos = (size_t)(q - data);
            continue;
        }
        pos += seglen;
    }
    if (!have_sof || nscans == 0) return fail(ICL_ERR_IO, "The image file might be corrupt or unreadable");
    // dequantise + inverse DCT, once, after the last scan
    for (int c = 0; c < ncomp; ++c) {
        component &k = comp[c];
        if (k.tq > 3 || !qt_ok[k.tq]) return fail(ICL_ERR_IO, "Missing table");
        const size_t stride = (size_t)k.wblocks * 8;
        k.plane.assign(stride * k.hblocks * 8, 0);
        int coef[64];
        for (int by = 0; by < k.hblocks; ++by)
            for (int bx = 0; bx < k.wblocks; ++bx) {
                const int16_t *cf = k.coefs.data() + ((size_t)by * k.wblocks + bx) * 64;
                for (int i = 0; i < 64; ++i) coef[i] = cf[i] * qt[k.tq][i];
                idct_islow(coef, k.plane.data() + (size_t)by * 8 * stride + (size_t)bx * 8, (int)stride);
            }
    }
    rgb.assign((size_t)W * H * 3, 0);
    if (ncomp == 1) {
        const size_t stride = (size_t)comp[0].wblocks * 8;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const uint8_t g = comp[0].plane[y * stride + x];
                uint8_t *o = &rgb[((size_t)y * W + x) * 3];
                o[0] = o[1] = o[2] = g;
            }
        return ICL_OK;
    }
    // chroma upsampling (jdsample.c): h2v1 / h2v2 "fancy" triangle filters, or none
    const int hs = comp[0].h, vs = comp[0].v;
    std::vector<uint8_t> up[2];
    for (int c = 1; c <= 2; ++c) {
        const component &k = comp[c];
        const size_t stride = (size_t)k.wblocks * 8;
        std::vector<uint8_t> &o = up[c - 1];
        o.assign((size_t)W * H, 0);
        const int dw = k.dw, dh = k.dh;
        auto in = [&](int r) -> const uint8_t * { return k.plane.data() + (size_t)std::min(std::max(r, 0), dh - 1) * stride; };
        if (hs == 1 && vs == 1) {
            for (int y = 0; y < H; ++y) memcpy(&o[(size_t)y * W], in(y), (size_t)W);
        } else if (hs == 2 && vs == 1) {
            std::vector<uint8_t> row((size_t)dw * 2 + 2);
            for (int y = 0; y < H; ++y) {
                const uint8_t *p = in(y);
                if (dw == 1) { row[0] = row[1] = p[0]; }
                else {
                    row[0] = p[0];
                    row[1] = (uint8_t)((p[0] * 3 + p[1] + 2) >> 2);
                    for (int i = 1; i < dw - 1; ++i) {
                        row[2 * i] = (uint8_t)((p[i] * 3 + p[i - 1] + 1) >> 2);
                        row[2 * i + 1] = (uint8_t)((p[i] * 3 + p[i + 1] + 2) >> 2);
                    }
                    row[2 * (dw - 1)] = (uint8_t)((p[dw - 1] * 3 + p[dw - 2] + 1) >> 2);
                    row[2 * (dw - 1) + 1] = p[dw - 1];
                }
                memcpy(&o[(size_t)y * W], row.data(), (size_t)W);
            }
        } else { // h2v2
            std::vector<int> cs0((size_t)dw), cs1((size_t)dw);
            std::vector<uint8_t> row((size_t)dw * 2 + 2);
            for (int y = 0; y < H; ++y) {
                const int r = y >> 1;
                const uint8_t *p0 = in(r), *p1 = in((y & 1) ? r + 1 : r - 1); // nearer / further input row
                int *cs = cs0.data();
                for (int i = 0; i < dw; ++i) cs[i] = p0[i] * 3 + p1[i];
                if (dw == 1) {
                    row[0] = (uint8_t)((cs[0] * 4 + 8) >> 4);
                    row[1] = (uint8_t)((cs[0] * 4 + 7) >> 4);
                } else {
                    row[0] = (uint8_t)((cs[0] * 4 + 8) >> 4);
                    row[1] = (uint8_t)((cs[0] * 3 + cs[1] + 7) >> 4);
                    for (int i = 1; i < dw - 1; ++i) {
                        row[2 * i] = (uint8_t)((cs[i] * 3 + cs[i - 1] + 8) >> 4);
                        row[2 * i + 1] = (uint8_t)((cs[i] * 3 + cs[i + 1] + 7) >> 4);
                    }
                    row[2 * (dw - 1)] = (uint8_t)((cs[dw - 1] * 3 + cs[dw - 2] + 8) >> 4);
                    row[2 * (dw - 1) + 1] = (uint8_t)((cs[dw - 1] * 4 + 7) >> 4);
                }
                memcpy(&o[(size_t)y * W], row.data(), (size_t)W);
            }
        }
    }
    // colour conversion (jdcolor.c): fixed-point YCbCr -> RGB, or pass-through for Adobe transform 0
    const bool is_rgb = (adobe && adobe_transform == 0) || (!adobe && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
    int cr_r[256], cb_b[256], cr_g[256], cb_g[256];
    for (int i = 0; i < 256; ++i) {
        const int x = i - 128;
        cr_r[i] = (int)((91881L * x + 32768) >> 16);   // FIX(1.40200)
        cb_b[i] = (int)((116130L * x + 32768) >> 16);  // FIX(1.77200)
        cr_g[i] = (int)(-46802L * x);                  // FIX(0.71414)
        cb_g[i] = (int)(-22554L * x + 32768);          // FIX(0.34414)
    }
    const size_t ystride = (size_t)comp[0].wblocks * 8;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int Y = comp[0].plane[y * ystride + x], cb = up[0][(size_t)y * W + x], cr = up[1][(size_t)y * W + x];
            uint8_t *o = &rgb[((size_t)y * W + x) * 3];
            if (is_rgb) {
                o[0] = (uint8_t)Y; o[1] = (uint8_t)cb; o[2] = (uint8_t)cr;
            } else {
                o[0] = clamp8(Y + cr_r[cr]);
                o[1] = clamp8(Y + ((cb_g[cb] + cr_g[cr]) >> 16));
                o[2] = clamp8(Y + cb_b[cb]);
            }
        }
    return ICL_OK;
}
