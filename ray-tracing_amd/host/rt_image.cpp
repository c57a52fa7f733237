// Texture ingest for C++ callers: ImageUtils::load_image (src/utils/image.rs:13-29).
//
// The reference decodes any file the `image` crate understands and keeps `into_rgb8()` of it: tightly packed RGB8,
// alpha dropped, grey replicated.  PNG decoding is lossless, so every correct decoder yields the reference's bytes;
// that half is implemented here (zlib inflate + the five PNG row filters), together with binary PPM as a
// decoder-free interchange format.  JPEG files go to rt_jpeg.cpp, which says what can and cannot be matched there.
#include "rt_host.h"

#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rt {
bool decode_jpeg(const std::vector<uint8_t> &file, Texture &t, std::string &err);   // rt_jpeg.cpp
namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return false; }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); err = "cannot size " + path; return false; }
    out.resize((size_t)n);
    size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    if (got != (size_t)n) { err = "short read on " + path; return false; }
    return true;
}

int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Sample `i` of a row whose samples are `depth` bits wide (1, 2, 4 or 8), most significant bits first.
uint32_t sample(const uint8_t *row, uint32_t i, uint32_t depth) {
    if (depth == 8) return row[i];
    const uint32_t per = 8 / depth, shift = (per - 1 - i % per) * depth;
    return (row[i / per] >> shift) & ((1u << depth) - 1u);
}

bool decode_png(const std::vector<uint8_t> &file, Texture &t, std::string &err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    uint32_t w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool have_ihdr = false, have_end = false;
    size_t pos = 8;
    while (pos + 12 <= file.size() && !have_end) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t *tag = &file[pos + 4], *body = &file[pos + 8];
        if ((size_t)len > file.size() - pos - 12) { err = "truncated PNG chunk"; return false; }
        if (crc32(crc32(0L, Z_NULL, 0), tag, len + 4) != be32(body + len)) { err = "PNG chunk checksum mismatch"; return false; }
        if (!std::memcmp(tag, "IHDR", 4)) {
            if (len != 13) { err = "bad IHDR"; return false; }
            w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) { err = "unknown PNG compression or filter method"; return false; }
            have_ihdr = true;
        } else if (!std::memcmp(tag, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(tag, "IEND", 4)) {
            have_end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_end) { err = "PNG without IHDR or IEND"; return false; }
    if (w == 0 || h == 0 || w > 65536u || h > 65536u) { err = "unsupported PNG size"; return false; }
    if (interlace != 0) { err = "interlaced PNG files are not supported"; return false; }
    uint32_t channels = 0;
    switch (ctype) {
        case 0: channels = 1; break;   // grey
        case 2: channels = 3; break;   // RGB
        case 3: channels = 1; break;   // palette index
        case 4: channels = 2; break;   // grey + alpha
        case 6: channels = 4; break;   // RGB + alpha
        default: err = "unknown PNG colour type"; return false;
    }
    const bool sub_byte = depth == 1 || depth == 2 || depth == 4;
    if (!(depth == 8 || (sub_byte && (ctype == 0 || ctype == 3)))) { err = "unsupported PNG bit depth " + std::to_string(depth); return false; }
    if (ctype == 3 && palette.size() < 3) { err = "palette PNG without PLTE"; return false; }

    const size_t bits_per_pixel = (size_t)channels * depth, stride = ((size_t)w * bits_per_pixel + 7) / 8;
    const size_t bpp = bits_per_pixel >= 8 ? bits_per_pixel / 8 : 1;   // filter distance in bytes
    // the header alone decides this allocation: refuse images whose decoded rows exceed 1 GiB before allocating anything
    if ((stride + 1) > ((size_t)1 << 30) / h) { err = "PNG larger than 1 GiB when decoded"; return false; }
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    int z = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (z != Z_OK || raw_len != raw.size()) { err = "PNG pixel data does not inflate to the declared size"; return false; }

    std::vector<uint8_t> prev(stride, 0), cur(stride);
    t.width = w; t.height = h;
    t.bytes.assign((size_t)w * h * 3, 0);
    const uint32_t grey_scale = depth == 8 ? 1u : 255u / ((1u << depth) - 1u);   // 1-, 2- and 4-bit grey expand by replication
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *in = &raw[(size_t)y * (stride + 1)];
        const uint8_t filter = in[0];
        if (filter > 4) { err = "unknown PNG row filter"; return false; }
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) / 2; break;
                case 4: pred = paeth(a, b, c); break;
                default: break;
            }
            cur[i] = (uint8_t)(in[1 + i] + pred);
        }
        uint8_t *out = &t.bytes[(size_t)y * w * 3];
        for (uint32_t x = 0; x < w; x++) {
            uint8_t r, g, bl;
            switch (ctype) {
                case 0: r = g = bl = (uint8_t)(sample(cur.data(), x, depth) * grey_scale); break;
                case 2: r = cur[3 * x]; g = cur[3 * x + 1]; bl = cur[3 * x + 2]; break;
                case 3: {
                    const size_t k = 3 * (size_t)sample(cur.data(), x, depth);
                    if (k + 2 >= palette.size()) { err = "palette index out of range"; return false; }
                    r = palette[k]; g = palette[k + 1]; bl = palette[k + 2];
                    break;
                }
                case 4: r = g = bl = cur[2 * x]; break;
                default: r = cur[4 * x]; g = cur[4 * x + 1]; bl = cur[4 * x + 2]; break;
            }
            out[3 * x] = r; out[3 * x + 1] = g; out[3 * x + 2] = bl;
        }
        prev.swap(cur);
    }
    return true;
}

// Binary PPM ("P6", maxval 255): header tokens separated by whitespace, `#` comments, one whitespace byte before the data.
bool decode_ppm(const std::vector<uint8_t> &file, Texture &t, std::string &err) {
    size_t pos = 2;
    uint32_t vals[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
        for (;;) {
            while (pos < file.size() && std::isspace(file[pos])) pos++;
            if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') pos++; continue; }
            break;
        }
        size_t digits = 0;
        uint64_t v = 0;
        while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9' && digits < 9) { v = v * 10 + (file[pos] - '0'); pos++; digits++; }
        if (digits == 0) { err = "bad PPM header"; return false; }
        vals[k] = (uint32_t)v;
    }
    if (pos >= file.size() || !std::isspace(file[pos])) { err = "bad PPM header"; return false; }
    pos++;
    if (vals[2] != 255u) { err = "only PPM files with maxval 255 are supported"; return false; }
    const size_t n = (size_t)vals[0] * vals[1] * 3;
    if (vals[0] == 0 || vals[1] == 0 || file.size() - pos < n) { err = "PPM pixel data shorter than the header says"; return false; }
    t.width = vals[0]; t.height = vals[1];
    t.bytes.assign(file.begin() + (long)pos, file.begin() + (long)(pos + n));
    return true;
}

}  // namespace

std::string load_image(const std::string &path, Texture &out) {
    std::vector<uint8_t> file;
    std::string err;
    if (!read_file(path, file, err)) return err;
    Texture t;
    t.path = path;
    bool ok = false;
    if (file.size() >= 8 && file[0] == 0x89 && file[1] == 'P') ok = decode_png(file, t, err);
    else if (file.size() >= 2 && file[0] == 'P' && file[1] == '6') ok = decode_ppm(file, t, err);
    else if (file.size() >= 2 && file[0] == 0xFF && file[1] == 0xD8) ok = decode_jpeg(file, t, err);
    else err = "unknown image format";
    if (!ok) return path + ": " + err;
    out = std::move(t);
    return "";
}

}  // namespace rt

extern "C" {

int32_t rth_load_image(const char *path, uint32_t *width, uint32_t *height, uint8_t **rgb8, char *err, size_t err_len) {
    if (err && err_len) err[0] = 0;
    if (!path || !width || !height || !rgb8) return -1;
    rt::Texture t;
    std::string r = rt::load_image(path, t);
    if (!r.empty()) {
        if (err && err_len) { std::strncpy(err, r.c_str(), err_len - 1); err[err_len - 1] = 0; }
        return -1;
    }
    uint8_t *p = (uint8_t *)std::malloc(t.bytes.size());
    if (!p) return -1;
    std::memcpy(p, t.bytes.data(), t.bytes.size());
    *width = t.width; *height = t.height; *rgb8 = p;
    return 0;
}

void rth_free_image(uint8_t *rgb8) { std::free(rgb8); }

}  // extern "C"
