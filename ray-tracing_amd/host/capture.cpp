// capture.cpp — image export and defaults of the reference's host.
//   capture_rgb8   ≙ /root/reference/src/ui.rs:84-88  `(x * 255.0).map(|x| x as u8)`
//   write_png_rgb8 ≙ src/ui.rs:83,90-92 (PngEncoder, ExtendedColorType::Rgb8)
//   uniform_default ≙ src/main.rs:73-84 + Camera::default (src/camera.rs:112-123)
#include "rt_host.h"
#include "../csrc/rt_math.h"

#include <cstdio>
#include <cstring>

namespace rt_host {

void capture_rgb8(const float *accum16, uint64_t n_pixels, uint8_t *rgb) {
    for (uint64_t i = 0; i < n_pixels; i++) {
        for (int k = 0; k < 3; k++) {
            rgb[3 * i + k] = rt_capture_u8(accum16[4 * i + k]);
        }
    }
}

namespace {
uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c;
}
void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
    put32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    uint32_t c = crc_update(0xFFFFFFFFu, &out[start], out.size() - start) ^ 0xFFFFFFFFu;
    put32(out, c);
}
}  // namespace

bool write_png_rgb8(const std::string &path, const uint8_t *rgb, uint32_t w, uint32_t h, std::string &err) {
    if (!crc_ready) crc_init();
    // raw scanlines with filter byte 0
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (3 * (size_t)w + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    // zlib stream of stored (uncompressed) deflate blocks
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    size_t off = 0;
    uint32_t a = 1, b = 0;
    do {
        size_t n = raw.size() - off;
        if (n > 65535) n = 65535;
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    } while (off < raw.size());
    put32(z, (b << 16) | a);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, w); put32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) err = "short write to " + path;
    return ok;
}

void uniform_default(RtUniform &u) {
    memset(&u, 0, sizeof u);
    u.camera.position[0] = u.camera.position[1] = u.camera.position[2] = 1.0f;
    u.camera.pitch = 0.0f;
    u.camera.yaw = 0.0f;
    u.camera.fov = 1.57079632679489661923f;  // FRAC_PI_2
    u.camera.aspect = 0.0f;
    u.frame = 0;
    u.accumulation_frame = 1;
    u.flags = 0;
    u.exposure = 1.0f;
    u.environment = 1.0f;
    u.max_bounces = 10;
    u.samples = 5;
}

}  // namespace rt_host
