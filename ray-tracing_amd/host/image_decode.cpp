// image_decode.cpp — texture image loading for the host scene loader.
// ≙ `image::load(file, format)…into_rgba8()` at /root/reference/src/scene.rs:157-170.
// The reference accepts whatever the `image` crate decodes; this host decodes
// PNG (1-16-bit, grey / grey+alpha / RGB / RGBA / palette, tRNS, plain or Adam7-interlaced;
// inflate by zlib) and binary PPM (P6).  Anything else is reported as an error
// (the reference panics on a failed decode, src/scene.rs:165-168).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_host.h"

namespace rt_host {

namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool decode_png(const std::vector<uint8_t> &f, Image &img, std::string &err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (f.size() < 8 || memcmp(f.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    while (pos + 12 <= f.size()) {
        uint32_t n = be32(&f[pos]);
        const uint8_t *type = &f[pos + 4], *body = &f[pos + 8];
        if (pos + 12 + (size_t)n > f.size()) { err = "truncated PNG chunk"; return false; }
        if (!memcmp(type, "IHDR", 4) && n >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + n);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + n);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + n);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)n;
    }
    if (w == 0 || h == 0 || w > 65536 || h > 65536) { err = "bad PNG dimensions"; return false; }
    if (interlace > 1) { err = "unknown PNG interlace method"; return false; }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || !(depth == 8 || (depth == 16 && ctype != 3) || (ctype == 3 && (depth == 1 || depth == 2 || depth == 4)) ||
                       (ctype == 0 && (depth == 1 || depth == 2 || depth == 4)))) { err = "unsupported PNG colour type / bit depth"; return false; }
    const size_t bpp_bits = (size_t)channels * depth, bpp = (bpp_bits + 7) / 8;
    // Sub-images: the whole picture, or the seven passes of Adam7 (PNG spec 8.2), each filtered on its own
    struct Pass { uint32_t x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass *passes = interlace ? adam7 : &whole;
    const int n_passes = interlace ? 7 : 1;
    size_t total = 0;
    for (int k = 0; k < n_passes; k++) {
        const uint32_t pw = (w - passes[k].x0 + passes[k].dx - 1) / passes[k].dx, ph = (h - passes[k].y0 + passes[k].dy - 1) / passes[k].dy;
        if (passes[k].x0 < w && passes[k].y0 < h && pw && ph) total += ((pw * bpp_bits + 7) / 8 + 1) * (size_t)ph;
    }
    // deflate cannot expand more than ~1032:1: a header that promises more than the IDAT bytes can hold is refused before anything is allocated
    if (total > idat.size() * 1040 + 1024) { err = "PNG inflate failed"; return false; }
    std::vector<uint8_t> raw(total);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) { err = "PNG inflate failed"; return false; }
    img.width = w; img.height = h;
    img.rgba.assign((size_t)w * h * 4, 255);
    // 16-bit samples become 8-bit as the `image` crate's into_rgba8() makes them (image 0.25, FromPrimitive<u16> for u8:
    // round(c * 255 / 65535) = (c + 128) / 257; the crate is not vendored with the reference, so this is its published rule)
    auto to8 = [](uint32_t c16) -> uint8_t { return (uint8_t)((c16 + 128u) / 257u); };
    size_t at = 0;
    for (int k = 0; k < n_passes; k++) {
        const Pass &ps = passes[k];
        if (ps.x0 >= w || ps.y0 >= h) continue;
        const uint32_t pw = (w - ps.x0 + ps.dx - 1) / ps.dx, ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        const size_t stride = (pw * bpp_bits + 7) / 8;
        std::vector<uint8_t> prev(stride, 0);
        for (uint32_t py = 0; py < ph; py++) {
            uint8_t *row = &raw[at];
            at += stride + 1;
            const int filter = row[0];
            uint8_t *cur = row + 1;
            for (size_t i = 0; i < stride; i++) {   // undo the scanline filter in place
                int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = cur[i];
                switch (filter) {
                    case 0: break;
                    case 1: x += a; break;
                    case 2: x += b; break;
                    case 3: x += (a + b) >> 1; break;
                    case 4: { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); x += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                    default: err = "bad PNG filter"; return false;
                }
                cur[i] = (uint8_t)x;
            }
            memcpy(prev.data(), cur, stride);
            const uint32_t y = ps.y0 + py * ps.dy;
            for (uint32_t px = 0; px < pw; px++) {
                const uint32_t x = ps.x0 + px * ps.dx;
                uint8_t *o = &img.rgba[4 * ((size_t)y * w + x)];
                auto raw_sample = [&](int ch) -> uint32_t {   // the sample as stored: 1..16 bits
                    if (depth == 8) return cur[(size_t)px * channels + ch];
                    if (depth == 16) { const uint8_t *q = &cur[2 * ((size_t)px * channels + ch)]; return ((uint32_t)q[0] << 8) | q[1]; }
                    size_t bit = (size_t)px * depth;
                    return (uint32_t)(cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                };
                auto sample8 = [&](int ch) -> uint8_t { const uint32_t v = raw_sample(ch); return depth == 16 ? to8(v) : (uint8_t)v; };
                if (ctype == 3) {
                    const uint32_t idx = raw_sample(0);
                    if ((size_t)idx * 3 + 2 < plte.size()) { o[0] = plte[3 * idx]; o[1] = plte[3 * idx + 1]; o[2] = plte[3 * idx + 2]; }
                    else { o[0] = o[1] = o[2] = 0; }
                    o[3] = (size_t)idx < trns.size() ? trns[idx] : 255;
                } else if (ctype == 0 || ctype == 4) {
                    const uint32_t r = raw_sample(0);
                    uint32_t v = depth == 16 ? to8(r) : r;
                    if (depth < 8) v = v * 255 / ((1u << depth) - 1u);
                    o[0] = o[1] = o[2] = (uint8_t)v;
                    if (ctype == 4) o[3] = sample8(1);
                    else if (trns.size() >= 2 && r == (((uint32_t)trns[0] << 8) | trns[1])) o[3] = 0;       // tRNS colour key (PNG spec 11.3.2.1)
                } else {
                    o[0] = sample8(0); o[1] = sample8(1); o[2] = sample8(2);
                    if (ctype == 6) o[3] = sample8(3);
                    else if (trns.size() >= 6 && raw_sample(0) == (((uint32_t)trns[0] << 8) | trns[1]) && raw_sample(1) == (((uint32_t)trns[2] << 8) | trns[3]) &&
                             raw_sample(2) == (((uint32_t)trns[4] << 8) | trns[5])) o[3] = 0;
                }
            }
        }
    }
    return true;
}

bool decode_ppm(const std::vector<uint8_t> &f, Image &img, std::string &err) {
    size_t pos = 2;
    auto next_int = [&]() -> long {
        for (;;) {
            while (pos < f.size() && isspace(f[pos])) pos++;
            if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') pos++; continue; }
            break;
        }
        long v = 0; bool any = false;
        while (pos < f.size() && isdigit(f[pos])) { v = v * 10 + (f[pos++] - '0'); any = true; }
        return any ? v : -1;
    };
    long w = next_int(), h = next_int(), maxv = next_int();
    if (w <= 0 || h <= 0 || maxv != 255) { err = "unsupported PPM header"; return false; }
    pos++;   // single whitespace after maxval
    if (pos + (size_t)w * h * 3 > f.size()) { err = "truncated PPM"; return false; }
    img.width = (uint32_t)w; img.height = (uint32_t)h;
    img.rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        img.rgba[4 * i] = f[pos + 3 * i]; img.rgba[4 * i + 1] = f[pos + 3 * i + 1]; img.rgba[4 * i + 2] = f[pos + 3 * i + 2]; img.rgba[4 * i + 3] = 255;
    }
    return true;
}

}  // namespace

bool load_image_rgba8(const std::string &path, Image &img, std::string &err) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) { err = "cannot open texture " + path; return false; }
    std::vector<uint8_t> f;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + n);
    fclose(fp);
    bool ok;
    if (f.size() >= 2 && f[0] == 'P' && f[1] == '6') ok = decode_ppm(f, img, err);
    else ok = decode_png(f, img, err);
    if (!ok) err = path + ": " + err;
    return ok;
}

// src/scene.rs:210-223
std::string strip_flags(const std::string &path) {
    size_t i = 0;
    while (i < path.size() && path[i] == '-') {
        for (int k = 0; k < 2; k++) {
            while (i < path.size() && path[i] != ' ') i++;
            i++;
        }
    }
    return i < path.size() ? path.substr(i) : std::string();
}

}  // namespace rt_host
