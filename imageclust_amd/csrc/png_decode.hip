// png_decode.hip -- PNG ingest for the embed path (host code).
//
// gocv.IMRead(path, IMReadColor) (/root/reference/internal/embeddings/embeddings.go:50) decodes PNG through OpenCV's libpng
// reader, and the reference's upload form advertises "JPG, PNG, GIF" (frontend/src/components/ImageUploadForm.jsx:145).  No
// libpng / zlib headers exist in this image, so this file restates the two public formats it needs: DEFLATE / zlib (RFC 1950,
// RFC 1951) and PNG (ISO/IEC 15948): chunk framing with CRC-32, IHDR / PLTE / IDAT / IEND, the five scanline filters, colour
// types 0 (grey), 2 (RGB), 3 (palette), 4 (grey + alpha), 6 (RGBA) at bit depths 1-16.  What cv::imread(IMREAD_COLOR) makes of
// them: 3 x 8-bit colour, alpha STRIPPED (not blended: png_set_strip_alpha), 16-bit samples cut to their high byte
// (png_set_strip_16), grey 1 / 2 / 4 bits scaled to 0..255, palette looked up, tRNS / gAMA and every other ancillary chunk
// ignored.  Adam7-interlaced files are decoded pass by pass (PNG 8.2).
//
// Hostile input: every length comes from the file.  Chunk lengths are checked against the bytes that are there, CRCs and the
// Adler-32 are verified, dimensions are capped like the JPEG reader's (64 Mpx), the inflated size must equal exactly
// height x (1 + row bytes), Huffman codes must be neither over-subscribed nor (for the dynamic literal / distance codes of a
// block that uses them) incomplete, a match may not reach in front of the output.  No C++ exception crosses the C ABI: the
// callers wrap this in their no_throw guard.
#include "icl_common.h"

#include <cstring>
#include <vector>

namespace {

struct bits_in {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int n = 0;
    bool bad = false;
    uint32_t get(int k) // k <= 16 bits, LSB first (RFC 1951 3.1.1)
    {
        while (n < k) {
            if (p >= end) {
                bad = true;
                return 0;
            }
            acc |= (uint32_t)*p++ << n;
            n += 8;
        }
        const uint32_t v = acc & ((1u << k) - 1u);
        acc >>= k;
        n -= k;
        return v;
    }
    void align_byte()
    {
        acc = 0;
        n = 0;
    }
};

// canonical Huffman code: count[len] codes of each length, symbols ordered by (length, value)
struct huff {
    uint16_t count[16];
    uint16_t symbol[288];
};

// returns 0: complete code, > 0: incomplete (that many codes of the longest length unused), < 0: over-subscribed
static int huff_build(huff &h, const uint8_t *len, int n)
{
    memset(h.count, 0, sizeof h.count);
    for (int s = 0; s < n; ++s) ++h.count[len[s]];
    if (h.count[0] == n) return 0; // no codes at all: complete by convention, decoding any symbol fails
    int left = 1;
    for (int l = 1; l < 16; ++l) {
        left <<= 1;
        left -= h.count[l];
        if (left < 0) return left;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int s = 0; s < n; ++s)
        if (len[s]) h.symbol[offs[len[s]]++] = (uint16_t)s;
    return left;
}

static int huff_decode(bits_in &b, const huff &h)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; ++l) {
        code |= (int)b.get(1);
        if (b.bad) return -1;
        const int cnt = h.count[l];
        if (code - cnt < first) return h.symbol[index + (code - first)];
        index += cnt;
        first += cnt;
        first <<= 1;
        code <<= 1;
    }
    return -1; // ran out of code lengths
}

// DEFLATE into out, which must end up holding exactly `want` bytes
static bool inflate_exact(bits_in &b, std::vector<uint8_t> &out, size_t want)
{
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    out.clear();
    out.reserve(want);
    huff lit, dist;
    for (;;) {
        const uint32_t last = b.get(1), type = b.get(2);
        if (b.bad) return false;
        if (type == 0) { // stored
            b.align_byte();
            if (b.end - b.p < 4) return false;
            const uint32_t len = b.p[0] | (b.p[1] << 8), nlen = b.p[2] | (b.p[3] << 8);
            b.p += 4;
            if ((len ^ 0xffffu) != nlen || (size_t)(b.end - b.p) < len || out.size() + len > want) return false;
            out.insert(out.end(), b.p, b.p + len);
            b.p += len;
        } else if (type == 1 || type == 2) {
            uint8_t lens[320];
            if (type == 1) { // fixed codes (RFC 1951 3.2.6)
                int s = 0;
                for (; s < 144; ++s) lens[s] = 8;
                for (; s < 256; ++s) lens[s] = 9;
                for (; s < 280; ++s) lens[s] = 7;
                for (; s < 288; ++s) lens[s] = 8;
                huff_build(lit, lens, 288);
                for (s = 0; s < 30; ++s) lens[s] = 5;
                huff_build(dist, lens, 30);
            } else { // dynamic codes (3.2.7)
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
                if (b.bad || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19];
                memset(cl, 0, sizeof cl);
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)b.get(3);
                if (b.bad) return false;
                huff lencode;
                if (huff_build(lencode, cl, 19) != 0) return false; // the code-length code must be complete
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = huff_decode(b, lencode);
                    if (sym < 0) return false;
                    if (sym < 16) {
                        lens[idx++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return false;
                            val = lens[idx - 1];
                            rep = 3 + (int)b.get(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)b.get(3);
                        } else {
                            rep = 11 + (int)b.get(7);
                        }
                        if (b.bad || idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) return false; // no end-of-block code
                // over-subscribed codes are errors; an incomplete code is only legal as ONE code of length 1 (what zlib accepts)
                const int el = huff_build(lit, lens, nlen);
                if (el < 0 || (el > 0 && !(lit.count[1] == 1 && nlen - lit.count[0] == 1))) return false;
                const int ed = huff_build(dist, lens + nlen, ndist);
                if (ed < 0 || (ed > 0 && !(dist.count[1] == 1 && ndist - dist.count[0] == 1))) return false;
            }
            for (;;) {
                int sym = huff_decode(b, lit);
                if (sym < 0) return false;
                if (sym < 256) {
                    if (out.size() >= want) return false;
                    out.push_back((uint8_t)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    const size_t len = lbase[sym] + b.get(lext[sym]);
                    const int ds = huff_decode(b, dist);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + b.get(dext[ds]);
                    if (b.bad || d > out.size() || out.size() + len > want) return false;
                    const size_t from = out.size() - d;
                    for (size_t i = 0; i < len; ++i) out.push_back(out[from + i]); // (may overlap its own output: byte by byte)
                }
            }
        } else {
            return false;
        }
        if (last) break;
    }
    return out.size() == want;
}

static uint32_t crc32_of(const uint8_t *p, size_t n)
{
    // (a function-local static initialised by a lambda: thread-safe -- icl_embed_file decodes on its callers' threads concurrently)
    struct crc_table {
        uint32_t t[256];
    };
    static const crc_table tab = [] {
        crc_table x;
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            x.t[i] = c;
        }
        return x;
    }();
    const uint32_t *table = tab.t;
    uint32_t c = 0xffffffffu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

static uint32_t adler32_of(const uint8_t *p, size_t n)
{
    uint32_t a = 1, b = 0;
    while (n) {
        const size_t k = n < 5552 ? n : 5552; // the largest run that cannot overflow 32 bits
        for (size_t i = 0; i < k; ++i) {
            a += p[i];
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
        p += k;
        n -= k;
    }
    return (b << 16) | a;
}

static inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

} // namespace

bool icl_is_png(const uint8_t *data, size_t len)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    return len >= 8 && memcmp(data, sig, 8) == 0;
}

int icl_png_decode(icl_ctx *ctx, const uint8_t *data, size_t len, const char *path, std::vector<uint8_t> &rgb, int &W, int &H)
{
    auto fail = [&](const char *why) { return icl_fail(ctx, ICL_ERR_IO, "failed to read image: %s. PNG: %s", path, why); };
    if (!icl_is_png(data, len)) return fail("bad signature");
    size_t pos = 8;
    bool have_ihdr = false, have_iend = false;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    uint8_t pal[256][3];
    int npal = 0;
    std::vector<uint8_t> z;
    while (pos + 12 <= len && !have_iend) {
        const uint32_t clen = be32(data + pos);
        if (clen > 0x7fffffffu || (size_t)clen > len - pos - 12) return fail("chunk runs past the end of the file");
        const uint8_t *type = data + pos + 4, *body = data + pos + 8;
        if (crc32_of(type, 4 + (size_t)clen) != be32(body + clen)) return fail("chunk CRC mismatch");
        const bool is = !memcmp(type, "IHDR", 4), ip = !memcmp(type, "PLTE", 4), id = !memcmp(type, "IDAT", 4), ie = !memcmp(type, "IEND", 4);
        if (!have_ihdr && !is) return fail("IHDR is not the first chunk");
        if (is) {
            if (have_ihdr || clen != 13) return fail("bad IHDR");
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || interlace > 1) return fail("unknown compression / filter / interlace method");
            const bool ok_depth = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                                  (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                                  ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
            if (!ok_depth) return fail("colour type / bit depth combination is not in the standard");
            if (w == 0 || h == 0 || w > 65535u || h > 65535u || (uint64_t)w * h > (64ull << 20)) return fail("image dimensions out of range (limit: 64 Mpx)");
            have_ihdr = true;
        } else if (ip) {
            if (clen % 3 || clen > 768 || npal) return fail("bad PLTE");
            npal = (int)(clen / 3);
            memcpy(pal, body, clen);
        } else if (id) {
            z.insert(z.end(), body, body + clen);
        } else if (ie) {
            have_iend = true;
        } else if (!(type[0] & 0x20)) {
            return fail("unknown critical chunk");
        }
        pos += 12 + (size_t)clen;
    }
    if (!have_ihdr || !have_iend) return fail("truncated file (no IEND)");
    if (ctype == 3 && npal == 0) return fail("palette image without PLTE");
    if (z.size() < 6) return fail("no image data");
    // zlib wrapper (RFC 1950): CM = 8, window <= 32 KiB, no preset dictionary, header check
    if ((z[0] & 0x0f) != 8 || (z[0] >> 4) > 7 || (z[1] & 0x20) || (((unsigned)z[0] << 8) | z[1]) % 31 != 0) return fail("bad zlib header");
    const int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
    const size_t bits_px = (size_t)channels * depth, bpp = bits_px >= 8 ? bits_px / 8 : 1;
    // the reduced images of the stream (PNG 8.2): one for a progressive file, the seven Adam7 passes for an interlaced one -- each
    // its own sequence of filtered scanlines; empty passes are absent from the stream
    struct pass_t {
        uint32_t xs, ys, dx, dy, pw, ph;
        size_t rowb, off;
    } passes[7];
    int npass = 0;
    size_t want = 0;
    {
        static const uint8_t a7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
        const int np = interlace ? 7 : 1;
        for (int q = 0; q < np; ++q) {
            const uint32_t xs = interlace ? a7[q][0] : 0, ys = interlace ? a7[q][1] : 0, dx = interlace ? a7[q][2] : 1, dy = interlace ? a7[q][3] : 1;
            if (xs >= w || ys >= h) continue;
            const uint32_t pw = (w - xs + dx - 1) / dx, ph = (h - ys + dy - 1) / dy;
            const size_t rowb = ((size_t)pw * bits_px + 7) / 8;
            passes[npass++] = pass_t{xs, ys, dx, dy, pw, ph, rowb, want};
            want += (size_t)ph * (rowb + 1);
        }
    }
    std::vector<uint8_t> raw;
    bits_in b{z.data() + 2, z.data() + z.size() - 4};
    if (!inflate_exact(b, raw, want)) return fail("corrupt or truncated DEFLATE stream");
    if (adler32_of(raw.data(), raw.size()) != be32(z.data() + z.size() - 4)) return fail("Adler-32 mismatch");
    // -> 3 x 8-bit RGB as cv::imread(IMREAD_COLOR) delivers it (alpha stripped, 16 -> 8 by the high byte, small greys scaled)
    W = (int)w;
    H = (int)h;
    rgb.assign((size_t)w * h * 3, 0);
    const int step = depth == 16 ? 2 : 1; // bytes per sample for depth >= 8
    std::vector<uint8_t> zero(((size_t)w * bits_px + 7) / 8, 0);
    for (int q = 0; q < npass; ++q) {
        const pass_t &ps = passes[q];
        const size_t rowb = ps.rowb;
        // scanline filters (PNG 9.2): Sub, Up, Average, Paeth over bytes, bpp bytes to the left; the line above the first one of a pass is zero
        const uint8_t *prev = zero.data();
        for (uint32_t y = 0; y < ps.ph; ++y) {
            uint8_t *line = raw.data() + ps.off + (size_t)y * (rowb + 1);
            const uint8_t ft = line[0];
            uint8_t *cur = line + 1;
            switch (ft) {
            case 0:
                break;
            case 1:
                for (size_t i = bpp; i < rowb; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
                break;
            case 2:
                for (size_t i = 0; i < rowb; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
                break;
            case 3:
                for (size_t i = 0; i < rowb; ++i) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + prev[i]) >> 1));
                break;
            case 4:
                for (size_t i = 0; i < rowb; ++i) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, bb = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                    const int p = a + bb - c, pa = p > a ? p - a : a - p, pb = p > bb ? p - bb : bb - p, pc = p > c ? p - c : c - p;
                    cur[i] = (uint8_t)(cur[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c)));
                }
                break;
            default:
                return fail("unknown scanline filter");
            }
            prev = cur;
        }
        for (uint32_t y = 0; y < ps.ph; ++y) {
            const uint8_t *cur = raw.data() + ps.off + (size_t)y * (rowb + 1) + 1;
            uint8_t *orow = rgb.data() + (size_t)(ps.ys + y * ps.dy) * w * 3;
            for (uint32_t x = 0; x < ps.pw; ++x) {
                uint8_t *o = orow + (size_t)(ps.xs + x * ps.dx) * 3;
                if (ctype == 2 || ctype == 6) {
                    const uint8_t *sp = cur + (size_t)x * channels * step;
                    o[0] = sp[0];
                    o[1] = sp[step];
                    o[2] = sp[2 * step];
                } else if (ctype == 4 || (ctype == 0 && depth >= 8)) {
                    o[0] = o[1] = o[2] = cur[(size_t)x * channels * step];
                } else { // packed samples: grey 1 / 2 / 4 bits or palette indices 1 / 2 / 4 / 8 bits
                    unsigned v;
                    if (depth == 8) v = cur[x];
                    else {
                        const size_t bit = (size_t)x * depth;
                        v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                    }
                    if (ctype == 3) {
                        if ((int)v >= npal) return fail("palette index out of range");
                        o[0] = pal[v][0];
                        o[1] = pal[v][1];
                        o[2] = pal[v][2];
                    } else {
                        o[0] = o[1] = o[2] = (uint8_t)(v * (255u / ((1u << depth) - 1u)));
                    }
                }
            }
        }
    }
    return ICL_OK;
}
