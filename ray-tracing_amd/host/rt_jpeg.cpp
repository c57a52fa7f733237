// JPEG half of ImageUtils::load_image (src/utils/image.rs:13-29): baseline and progressive Huffman-coded files, 8 bits per
// sample, grey or YCbCr / RGB, chroma at full, half-width or half-width-half-height resolution.
//
// The reference decodes JPEG files with zune-jpeg 0.4.14 (through image 0.25.5, Cargo.lock); that crate is not part of the
// reference's tree, and its output is not a function of the file alone: like every JPEG decoder it chooses an inverse DCT,
// a chroma interpolation and a fixed-point colour conversion, and it chooses differently with and without AVX2.  So there
// is no byte sequence "the reference's texels" to reproduce for a JPEG file.  This decoder makes the choices of the JPEG
// group's library, which most tools share: the 13-bit integer inverse DCT, triangle-filter ("fancy") chroma interpolation
// with its alternating rounding, 16-bit fixed-point YCbCr -> RGB.  Its output is byte-equal to Pillow's for the files of
// the test suite (the reference's two JPEG textures among them); against zune-jpeg single texels can differ by a few units.
//
// Entropy decoding follows ITU-T T.81 (F.2.2 sequential, G.1.2 progressive with successive approximation).
#include "rt_host.h"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rt {
namespace {

const uint8_t kNatural[64 + 16] = {   // zigzag position -> row-major position (16 spare entries for runs past the end)
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huff {
    bool defined = false;
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
};

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;          // tables of the current scan
    uint32_t bw = 0, bh = 0;     // blocks per row / column, padded to whole MCUs
    uint32_t sw = 0, sh = 0;     // samples that belong to the image: ceil(width * h / hmax), ceil(height * v / vmax)
    std::vector<int16_t> coef;   // bw * bh blocks of 64, row-major positions
    int32_t pred = 0;
    std::vector<uint8_t> plane;  // bw * 8 x bh * 8 samples after the inverse DCT
};

struct Bits {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int cnt = 0;
    bool starved = false;   // a marker or the end of the file was reached: zero bits are supplied from there on
    void fill() {
        while (cnt <= 24) {
            uint32_t b = 0;
            if (!starved && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;   // stuffed zero
                    else { starved = true; b = 0; }             // a marker: stays where it is
                } else p++;
            } else starved = true;
            buf |= b << (24 - cnt);
            cnt += 8;
        }
    }
    int bit() { if (cnt < 1) fill(); const int r = (int)(buf >> 31); buf <<= 1; cnt--; return r; }
    int32_t bits(int n) {
        if (n == 0) return 0;
        if (cnt < n) fill();
        const int32_t r = (int32_t)(buf >> (32 - n)); buf <<= n; cnt -= n; return r;
    }
    void align() { buf = 0; cnt = 0; }
};

int32_t extend(int32_t v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }   // T.81 F.2.2.1

struct Decoder {
    const uint8_t *data; size_t size;
    std::string err;
    uint16_t quant[4][64]; bool have_quant[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Comp> comps;
    uint32_t width = 0, height = 0, restart = 0;
    int hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false, adobe = false; int adobe_transform = 0; bool jfif = false;

    bool fail(const char *m) { if (err.empty()) err = m; return false; }

    bool decode_symbol(Bits &b, const Huff &h, int &out) {   // T.81 F.2.2.3
        int32_t code = b.bit();
        for (int len = 1; len <= 16; len++) {
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) { out = h.vals[h.valptr[len] + code - h.mincode[len]]; return true; }
            code = (code << 1) | b.bit();
        }
        return fail("JPEG: code not in the Huffman table");
    }

    bool parse_dht(const uint8_t *p, size_t len) {
        while (len > 0) {
            if (len < 17) return fail("JPEG: short DHT segment");
            const int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) return fail("JPEG: bad Huffman table id");
            Huff &h = tc ? ac[th] : dc[th];
            int total = 0;
            for (int i = 1; i <= 16; i++) total += p[i];
            if (total > 256 || (size_t)(17 + total) > len) return fail("JPEG: bad Huffman table");
            std::memcpy(h.vals, p + 17, (size_t)total);
            int32_t code = 0; int k = 0;
            for (int l = 1; l <= 16; l++) {
                h.valptr[l] = k; h.mincode[l] = code;
                k += p[l]; code += p[l];
                h.maxcode[l] = p[l] ? code - 1 : -1;
                code <<= 1;
            }
            h.maxcode[17] = 0x7FFFFFFF;
            h.defined = true;
            p += 17 + total; len -= (size_t)(17 + total);
        }
        return true;
    }

    bool parse_dqt(const uint8_t *p, size_t len) {
        while (len > 0) {
            const int pq = p[0] >> 4, tq = p[0] & 15;
            if (tq > 3 || pq > 1) return fail("JPEG: bad quantisation table id");
            const size_t need = 1 + (pq ? 128 : 64);
            if (len < need) return fail("JPEG: short DQT segment");
            for (int i = 0; i < 64; i++) quant[tq][kNatural[i]] = pq ? (uint16_t)((p[1 + 2 * i] << 8) | p[2 + 2 * i]) : p[1 + i];
            have_quant[tq] = true;
            p += need; len -= need;
        }
        return true;
    }

    bool parse_sof(const uint8_t *p, size_t len) {
        if (have_frame) return fail("JPEG: more than one frame");
        if (len < 6) return fail("JPEG: short frame header");
        if (p[0] != 8) return fail("JPEG: only 8 bits per sample are supported");
        height = (uint32_t)((p[1] << 8) | p[2]); width = (uint32_t)((p[3] << 8) | p[4]);
        const int n = p[5];
        if (width == 0 || height == 0) return fail("JPEG: empty frame");
        if (n != 1 && n != 3) return fail("JPEG: only grey and three-component files are supported");
        if (len < (size_t)(6 + 3 * n)) return fail("JPEG: short frame header");
        comps.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            Comp &c = comps[(size_t)i];
            c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return fail("JPEG: unsupported sampling factors");
            hmax = std::max(hmax, c.h); vmax = std::max(vmax, c.v);
        }
        if (n == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }   // a single component is never interleaved
        if (n == 3) {
            if (comps[0].h != hmax || comps[0].v != vmax) return fail("JPEG: unsupported sampling factors");
            for (int i = 1; i < 3; i++) {
                const bool full = comps[(size_t)i].h == hmax && comps[(size_t)i].v == vmax;
                const bool half_w = hmax == 2 && comps[(size_t)i].h == 1 && comps[(size_t)i].v == vmax;
                const bool half_wh = hmax == 2 && vmax == 2 && comps[(size_t)i].h == 1 && comps[(size_t)i].v == 1;
                if (!full && !half_w && !half_wh) return fail("JPEG: unsupported chroma sampling");
            }
        }
        const uint32_t mcux = (width + 8u * (uint32_t)hmax - 1u) / (8u * (uint32_t)hmax), mcuy = (height + 8u * (uint32_t)vmax - 1u) / (8u * (uint32_t)vmax);
        uint64_t total = 0;
        for (Comp &c : comps) {
            c.bw = mcux * (uint32_t)c.h; c.bh = mcuy * (uint32_t)c.v;
            c.sw = (width * (uint32_t)c.h + (uint32_t)hmax - 1u) / (uint32_t)hmax; c.sh = (height * (uint32_t)c.v + (uint32_t)vmax - 1u) / (uint32_t)vmax;
            total += (uint64_t)c.bw * c.bh * 64u * 3u;
        }
        if (total > ((uint64_t)1 << 30)) return fail("JPEG larger than 1 GiB when decoded");
        for (Comp &c : comps) c.coef.assign((size_t)c.bw * c.bh * 64u, 0);
        have_frame = true;
        return true;
    }

    // one scan: header at p (after the length word), entropy-coded data behind it; `next` = where the following marker starts
    bool parse_scan(const uint8_t *p, size_t len, const uint8_t *&next) {
        if (!have_frame) return fail("JPEG: scan before the frame header");
        if (len < 1) return fail("JPEG: short scan header");
        const int ns = p[0];
        if (ns < 1 || ns > (int)comps.size() || len < (size_t)(4 + 2 * ns)) return fail("JPEG: bad scan header");
        Comp *sc[3];
        for (int i = 0; i < ns; i++) {
            sc[i] = nullptr;
            for (Comp &c : comps) if (c.id == p[1 + 2 * i]) sc[i] = &c;
            if (!sc[i]) return fail("JPEG: scan names an unknown component");
            for (int j = 0; j < i; j++) if (sc[j] == sc[i]) return fail("JPEG: scan names a component twice");
            sc[i]->td = p[2 + 2 * i] >> 4; sc[i]->ta = p[2 + 2 * i] & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("JPEG: bad table selector");
        }
        int ss = p[1 + 2 * ns], se = p[2 + 2 * ns];
        const int ah = p[3 + 2 * ns] >> 4, al = p[3 + 2 * ns] & 15;
        if (!progressive) { if (ss != 0 || se != 63 || ah != 0 || al != 0) { ss = 0; se = 63; } }
        else {
            if (ss > se || se > 63 || al > 13 || ah > 13) return fail("JPEG: bad progression parameters");
            if (ss == 0 && se != 0) return fail("JPEG: a progressive scan mixes DC and AC coefficients");
            if (ss != 0 && ns != 1) return fail("JPEG: an AC scan with several components");
            if (ah != 0 && ah != al + 1) return fail("JPEG: bad successive approximation");
        }
        for (int i = 0; i < ns; i++) {
            const bool need_dc = ss == 0 && ah == 0, need_ac = se > 0;
            if (need_dc && !dc[sc[i]->td].defined) return fail("JPEG: scan uses an undefined DC table");
            if (need_ac && !ac[sc[i]->ta].defined) return fail("JPEG: scan uses an undefined AC table");
        }
        Bits b; b.p = p + len; b.end = data + size;
        for (Comp &c : comps) c.pred = 0;
        uint32_t eobrun = 0;
        const bool interleaved = ns > 1;
        uint32_t units_x, units_y;   // MCUs of this scan
        if (interleaved) { units_x = comps[0].bw / (uint32_t)comps[0].h; units_y = comps[0].bh / (uint32_t)comps[0].v; }
        else { units_x = (sc[0]->sw + 7u) / 8u; units_y = (sc[0]->sh + 7u) / 8u; }
        uint32_t until_restart = restart, rst_index = 0;
        for (uint32_t my = 0; my < units_y; my++) for (uint32_t mx = 0; mx < units_x; mx++) {
            if (restart != 0 && until_restart == 0) {
                b.align();
                const uint8_t *q = b.p;
                while (q + 1 < b.end && q[0] == 0xFF && q[1] == 0xFF) q++;
                if (!(q + 1 < b.end && q[0] == 0xFF && q[1] == (uint8_t)(0xD0 + (rst_index & 7u)))) return fail("JPEG: restart marker missing");
                b.p = q + 2; b.starved = false;
                rst_index++; until_restart = restart;
                for (Comp &c : comps) c.pred = 0;
                eobrun = 0;
            }
            for (int i = 0; i < ns; i++) {
                Comp &c = *sc[i];
                const uint32_t nh = interleaved ? (uint32_t)c.h : 1u, nv = interleaved ? (uint32_t)c.v : 1u;
                for (uint32_t by = 0; by < nv; by++) for (uint32_t bx = 0; bx < nh; bx++) {
                    const uint32_t col = mx * nh + bx, row = my * nv + by;
                    int16_t *blk = &c.coef[((size_t)row * c.bw + col) * 64u];
                    if (!progressive) { if (!block_sequential(b, c, blk)) return false; }
                    else if (ss == 0) { if (!block_dc(b, c, blk, ah, al)) return false; }
                    else if (ah == 0) { if (!block_ac_first(b, c, blk, ss, se, al, eobrun)) return false; }
                    else { if (!block_ac_refine(b, c, blk, ss, se, al, eobrun)) return false; }
                }
            }
            if (restart != 0) until_restart--;
        }
        // the marker behind the scan's data
        const uint8_t *q = b.p;
        while (q + 1 < b.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
        next = q;
        return true;
    }

    bool block_sequential(Bits &b, Comp &c, int16_t *blk) {
        int s;
        if (!decode_symbol(b, dc[c.td], s)) return false;
        if (s > 11) return fail("JPEG: bad DC category");
        c.pred = (int32_t)((uint32_t)c.pred + (uint32_t)extend(b.bits(s), s));   // (wraps on hostile files instead of overflowing)
        blk[0] = (int16_t)c.pred;
        for (int k = 1; k < 64; k++) {
            int rs;
            if (!decode_symbol(b, ac[c.ta], rs)) return false;
            const int r = rs >> 4; s = rs & 15;
            if (s) { k += r; const int32_t v = extend(b.bits(s), s); if (k < 64) blk[kNatural[k]] = (int16_t)v; }
            else { if (r != 15) break; k += 15; }
        }
        return true;
    }
    bool block_dc(Bits &b, Comp &c, int16_t *blk, int ah, int al) {
        if (ah == 0) {
            int s;
            if (!decode_symbol(b, dc[c.td], s)) return false;
            if (s > 11) return fail("JPEG: bad DC category");
            c.pred = (int32_t)((uint32_t)c.pred + (uint32_t)extend(b.bits(s), s));   // (wraps on hostile files instead of overflowing)
            blk[0] = (int16_t)(c.pred * (1 << al));
        } else if (b.bit()) blk[0] = (int16_t)(blk[0] | (1 << al));
        return true;
    }
    bool block_ac_first(Bits &b, Comp &c, int16_t *blk, int ss, int se, int al, uint32_t &eobrun) {
        if (eobrun > 0) { eobrun--; return true; }
        for (int k = ss; k <= se; k++) {
            int rs;
            if (!decode_symbol(b, ac[c.ta], rs)) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s) { k += r; const int32_t v = extend(b.bits(s), s); if (k < 64) blk[kNatural[k]] = (int16_t)(v * (1 << al)); }
            else if (r == 15) k += 15;
            else { eobrun = 1u << r; if (r) eobrun += (uint32_t)b.bits(r); eobrun--; break; }
        }
        return true;
    }
    bool block_ac_refine(Bits &b, Comp &c, int16_t *blk, int ss, int se, int al, uint32_t &eobrun) {
        const int32_t p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        auto correct = [&](int16_t &coef) {   // one correction bit for a coefficient that is already non-zero
            if (b.bit() && (coef & p1) == 0) coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
        };
        if (eobrun == 0) {
            for (; k <= se; k++) {
                int rs;
                if (!decode_symbol(b, ac[c.ta], rs)) return false;
                int r = rs >> 4; int32_t s = rs & 15;
                if (s) { if (s != 1) return fail("JPEG: bad refinement code"); s = b.bit() ? p1 : m1; }
                else if (r != 15) { eobrun = 1u << r; if (r) eobrun += (uint32_t)b.bits(r); break; }
                do {   // pass the non-zero coefficients on the way (each takes a correction bit) and r zero ones
                    int16_t &coef = blk[kNatural[k]];
                    if (coef != 0) correct(coef);
                    else if (--r < 0) break;
                    k++;
                } while (k <= se);
                if (s && k < 64) blk[kNatural[k]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) { int16_t &coef = blk[kNatural[k]]; if (coef != 0) correct(coef); }
            eobrun--;
        }
        return true;
    }

    // inverse DCT of the JPEG group's library ("slow integer": 13-bit constants, two passes, 2 extra bits between them)
    static void idct(const int16_t *in, const uint16_t *q, uint8_t *out, size_t stride) {
        const int64_t c0_298 = 2446, c0_390 = 3196, c0_541 = 4433, c0_765 = 6270, c0_899 = 7373, c1_175 = 9633, c1_501 = 12299, c1_847 = 15137,
                      c1_961 = 16069, c2_053 = 16819, c2_562 = 20995, c3_072 = 25172;
        int64_t ws[64];
        auto descale = [](int64_t x, int n) { return (x + ((int64_t)1 << (n - 1))) >> n; };
        for (int pass = 0; pass < 2; pass++) {
            for (int i = 0; i < 8; i++) {
                int64_t v[8];
                for (int k = 0; k < 8; k++) v[k] = pass == 0 ? (int64_t)in[k * 8 + i] * (int64_t)q[k * 8 + i] : ws[i * 8 + k];
                int64_t z2 = v[2], z3 = v[6];
                int64_t z1 = (z2 + z3) * c0_541;
                int64_t tmp2 = z1 - z3 * c1_847, tmp3 = z1 + z2 * c0_765;
                int64_t tmp0 = (v[0] + v[4]) * 8192, tmp1 = (v[0] - v[4]) * 8192;
                const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
                tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
                z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
                int64_t z4 = tmp1 + tmp3;
                const int64_t z5 = (z3 + z4) * c1_175;
                tmp0 *= c0_298; tmp1 *= c2_053; tmp2 *= c3_072; tmp3 *= c1_501;
                z1 *= -c0_899; z2 *= -c2_562; z3 *= -c1_961; z4 *= -c0_390;
                z3 += z5; z4 += z5;
                tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
                const int64_t r[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
                if (pass == 0) for (int k = 0; k < 8; k++) ws[k * 8 + i] = descale(r[k], 11);
                else for (int k = 0; k < 8; k++) {
                    // the library's range table: sample + 128 clamped to 0..255, addressed by the low 10 bits
                    const uint32_t idx = (uint32_t)descale(r[k], 18) & 1023u;
                    out[(size_t)i * stride + (size_t)k] = (uint8_t)(idx < 128u ? idx + 128u : (idx < 512u ? 255u : (idx < 896u ? 0u : idx - 896u)));
                }
            }
        }
    }

    bool finish(Texture &t) {
        for (Comp &c : comps) {
            if (!have_quant[c.tq]) return fail("JPEG: component uses an undefined quantisation table");
            const size_t stride = (size_t)c.bw * 8u;
            c.plane.assign(stride * c.bh * 8u, 0);
            for (uint32_t by = 0; by < c.bh; by++) for (uint32_t bx = 0; bx < c.bw; bx++)
                idct(&c.coef[((size_t)by * c.bw + bx) * 64u], quant[c.tq], &c.plane[(size_t)by * 8u * stride + (size_t)bx * 8u], stride);
            std::vector<int16_t>().swap(c.coef);
        }
        t.width = width; t.height = height;
        t.bytes.assign((size_t)width * height * 3u, 0);
        if (comps.size() == 1) {
            const Comp &c = comps[0];
            for (uint32_t y = 0; y < height; y++) for (uint32_t x = 0; x < width; x++) {
                const uint8_t g = c.plane[(size_t)y * c.bw * 8u + x];
                uint8_t *o = &t.bytes[((size_t)y * width + x) * 3u]; o[0] = o[1] = o[2] = g;
            }
            return true;
        }
        // chroma at full resolution, one output row at a time
        std::vector<uint8_t> rows[3];
        std::vector<int32_t> sums;
        for (auto &r : rows) r.assign((size_t)width + 2u, 0);
        // YCbCr -> RGB in 16-bit fixed point (the library's tables: 1.402, 1.772 rounded per entry; 0.71414, 0.34414 rounded once per pixel)
        const bool ycc = jfif ? true : (adobe ? adobe_transform != 0 : !(comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B'));
        for (uint32_t y = 0; y < height; y++) {
            for (int ci = 0; ci < 3; ci++) {
                const Comp &c = comps[(size_t)ci];
                const size_t stride = (size_t)c.bw * 8u;
                uint8_t *o = rows[ci].data();
                if (c.h == hmax && c.v == vmax) { std::memcpy(o, &c.plane[(size_t)y * stride], width); continue; }
                const uint32_t dw = c.sw;
                if (c.v == vmax) {   // half width: 3/4 of the nearer sample, 1/4 of the farther one, roundings 1 and 2 alternating
                    const uint8_t *in = &c.plane[(size_t)y * stride];
                    if (dw <= 2u) { for (uint32_t x = 0; x < width; x++) o[x] = in[x / 2u]; continue; }
                    for (uint32_t x = 0; x < width; x++) {
                        const uint32_t i = x / 2u;
                        int v;
                        if ((x & 1u) == 0u) v = i == 0u ? in[0] : (3 * in[i] + in[i - 1u] + 1) >> 2;
                        else v = i == dw - 1u ? in[i] : (3 * in[i] + in[i + 1u] + 2) >> 2;
                        o[x] = (uint8_t)v;
                    }
                    continue;
                }
                // half width and half height: 9/16, 3/16, 3/16, 1/16 of the four nearest samples, roundings 8 and 7 alternating;
                // beyond the first and last row of the component the row itself stands in
                const uint32_t r0 = y / 2u;
                const uint32_t r1 = (y & 1u) == 0u ? (r0 == 0u ? 0u : r0 - 1u) : (r0 + 1u < c.sh ? r0 + 1u : c.sh - 1u);
                const uint8_t *in0 = &c.plane[(size_t)r0 * stride], *in1 = &c.plane[(size_t)r1 * stride];
                if (dw <= 2u) { for (uint32_t x = 0; x < width; x++) o[x] = in0[x / 2u]; continue; }   // (the library falls back to replication)
                sums.resize(dw);
                for (uint32_t i = 0; i < dw; i++) sums[i] = 3 * in0[i] + in1[i];
                for (uint32_t x = 0; x < width; x++) {
                    const uint32_t i = x / 2u;
                    int v;
                    if ((x & 1u) == 0u) v = i == 0u ? (sums[0] * 4 + 8) >> 4 : (sums[i] * 3 + sums[i - 1u] + 8) >> 4;
                    else v = i == dw - 1u ? (sums[i] * 4 + 7) >> 4 : (sums[i] * 3 + sums[i + 1u] + 7) >> 4;
                    o[x] = (uint8_t)v;
                }
            }
            uint8_t *o = &t.bytes[(size_t)y * width * 3u];
            for (uint32_t x = 0; x < width; x++) {
                const int Y = rows[0][x], cb = rows[1][x] - 128, cr = rows[2][x] - 128;
                if (!ycc) { o[3 * x] = rows[0][x]; o[3 * x + 1] = rows[1][x]; o[3 * x + 2] = rows[2][x]; continue; }
                const int r = Y + (int)((91881LL * cr + 32768) >> 16);
                const int g = Y + (int)((-22554LL * cb + 32768 - 46802LL * cr) >> 16);
                const int bl = Y + (int)((116130LL * cb + 32768) >> 16);
                o[3 * x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
                o[3 * x + 1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
                o[3 * x + 2] = (uint8_t)(bl < 0 ? 0 : (bl > 255 ? 255 : bl));
            }
        }
        return true;
    }

    bool run(Texture &t) {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG file");
        const uint8_t *p = data + 2, *end = data + size;
        bool saw_scan = false;
        for (;;) {
            while (p < end && *p != 0xFF) p++;            // (garbage between segments is skipped, as the library does)
            while (p < end && *p == 0xFF) p++;
            if (p >= end) break;
            const uint8_t m = *p++;
            if (m == 0xD9) break;                          // EOI
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (p + 2 > end) return fail("JPEG: truncated segment");
            const size_t len = (size_t)((p[0] << 8) | p[1]);
            if (len < 2 || p + len > end) return fail("JPEG: truncated segment");
            const uint8_t *body = p + 2; const size_t blen = len - 2;
            switch (m) {
                case 0xC0: case 0xC1: progressive = false; if (!parse_sof(body, blen)) return false; break;
                case 0xC2: progressive = true; if (!parse_sof(body, blen)) return false; break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    return fail("JPEG: lossless, hierarchical and arithmetic-coded files are not supported");
                case 0xC4: if (!parse_dht(body, blen)) return false; break;
                case 0xDB: if (!parse_dqt(body, blen)) return false; break;
                case 0xDD: if (blen < 2) return fail("JPEG: short DRI segment"); restart = (uint32_t)((body[0] << 8) | body[1]); break;
                case 0xE0: if (blen >= 5 && !std::memcmp(body, "JFIF", 5)) jfif = true; break;
                case 0xEE: if (blen >= 12 && !std::memcmp(body, "Adobe", 5)) { adobe = true; adobe_transform = body[11]; } break;
                case 0xDA: {
                    const uint8_t *next = nullptr;
                    if (!parse_scan(body, blen, next)) return false;
                    saw_scan = true;
                    p = next;
                    continue;
                }
                default: break;
            }
            p += len;
        }
        if (!have_frame || !saw_scan) return fail("JPEG: no image data");
        return finish(t);
    }
};

}  // namespace

bool decode_jpeg(const std::vector<uint8_t> &file, Texture &t, std::string &err) {
    Decoder d;
    d.data = file.data(); d.size = file.size();
    if (!d.run(t)) { err = d.err.empty() ? "JPEG: cannot decode" : d.err; return false; }
    return true;
}

}  // namespace rt
