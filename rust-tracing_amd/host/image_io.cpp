#include "image_io.hpp"
#include "jpeg_decoder.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <zlib.h>

namespace rt {

// ---- procedural stand-in for the earth map: integer arithmetic only, so every machine produces the same bytes
static inline uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// lattice value in [0, 65535]; x wraps with period `px` so that the u = 0 / u = 1 seam is continuous
static inline uint32_t lattice(int32_t ix, int32_t iy, int32_t px, uint32_t oct) {
    ix %= px; if (ix < 0) ix += px;
    return hash_u32((uint32_t)ix * 0x9E3779B1u ^ hash_u32((uint32_t)iy + 0x85EBCA6Bu * (oct + 1u))) & 0xFFFFu;
}
// bilinear value noise at fixed-point position (x, y) in 16.16 lattice units
static inline uint32_t value_noise(uint32_t x, uint32_t y, int32_t px, uint32_t oct) {
    int32_t ix = (int32_t)(x >> 16), iy = (int32_t)(y >> 16);
    uint64_t fx = x & 0xFFFFu, fy = y & 0xFFFFu;
    // smoothstep weights in 0..65536
    fx = (fx * fx >> 16) * (3u * 65536u - 2u * fx) >> 16;
    fy = (fy * fy >> 16) * (3u * 65536u - 2u * fy) >> 16;
    uint64_t a = lattice(ix, iy, px, oct), b = lattice(ix + 1, iy, px, oct);
    uint64_t c = lattice(ix, iy + 1, px, oct), d = lattice(ix + 1, iy + 1, px, oct);
    uint64_t top = a * (65536u - fx) + b * fx, bot = c * (65536u - fx) + d * fx; // <= 2^32
    return (uint32_t)(((top >> 16) * (65536u - fy) + (bot >> 16) * fy) >> 16);   // 0..65535
}

ImageRGB8 synthetic_earth(int32_t width, int32_t height) {
    if (width <= 0 || height <= 0) throw std::runtime_error("synthetic_earth: bad size");
    auto px = std::make_shared<std::vector<uint8_t>>((size_t)width * (size_t)height * 3u);
    const int32_t base_period = 8; // lattice cells around the globe at octave 0
    for (int32_t j = 0; j < height; ++j) {
        for (int32_t i = 0; i < width; ++i) {
            uint32_t e = 0, amp = 32768u;
            for (uint32_t o = 0; o < 6; ++o) {
                int32_t period = base_period << o;
                uint32_t x = (uint32_t)(((uint64_t)i * (uint64_t)period << 16) / (uint64_t)width);
                uint32_t y = (uint32_t)(((uint64_t)j * (uint64_t)(period / 2) << 16) / (uint64_t)height);
                e += (uint32_t)(((uint64_t)value_noise(x, y, period, o) * amp) >> 16);
                amp >>= 1;
            }
            // e in [0, 65535); latitude in 0..32768 from the equator to the poles
            int32_t lat = (int32_t)(((int64_t)std::abs(2 * j - height + 1) << 15) / height);
            uint32_t dither = hash_u32((uint32_t)(j * width + i)) & 7u;
            uint8_t r, g, b;
            if (lat * 2 + (int32_t)(e >> 3) > 66000) { // ice caps
                r = (uint8_t)(232 + dither); g = (uint8_t)(238 + dither); b = (uint8_t)(244 + dither);
            } else if (e < 33000u) { // ocean: deeper = darker
                uint32_t d = e >> 9; // 0..64
                r = (uint8_t)(6 + (d >> 2) + dither);
                g = (uint8_t)(22 + d + dither);
                b = (uint8_t)(70 + 2 * d + dither);
            } else { // land: green lowlands to brown highlands
                uint32_t hgt = (e - 33000u) >> 8; // 0..127
                if (hgt > 127) hgt = 127;
                r = (uint8_t)(40 + hgt + dither);
                g = (uint8_t)(110 - (hgt >> 1) + dither);
                b = (uint8_t)(30 + (hgt >> 2) + dither);
            }
            uint8_t *p = px->data() + ((size_t)j * (size_t)width + (size_t)i) * 3u;
            p[0] = r; p[1] = g; p[2] = b;
        }
    }
    ImageRGB8 img;
    img.width = width;
    img.height = height;
    img.pixels = px;
    return img;
}

static ImageRGB8 load_ppm(const std::string &path, FILE *f) {
    int w = 0, h = 0, maxv = 0;
    char magic[3] = {0, 0, 0};
    auto skip = [&]() {
        int c;
        for (;;) {
            c = fgetc(f);
            if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); }
            else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
            else { ungetc(c, f); break; }
        }
    };
    if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || magic[1] != '6')
        throw std::runtime_error("load_image_rgb8: " + path + ": not a binary PPM");
    skip(); if (fscanf(f, "%d", &w) != 1) throw std::runtime_error("load_image_rgb8: bad PPM header");
    skip(); if (fscanf(f, "%d", &h) != 1) throw std::runtime_error("load_image_rgb8: bad PPM header");
    skip(); if (fscanf(f, "%d", &maxv) != 1) throw std::runtime_error("load_image_rgb8: bad PPM header");
    fgetc(f); // single whitespace after maxval
    if (w <= 0 || h <= 0 || maxv != 255) throw std::runtime_error("load_image_rgb8: unsupported PPM");
    if ((uint64_t)w * (uint64_t)h > MAX_IMAGE_PIXELS) throw std::runtime_error("load_image_rgb8: " + path + ": image too large");
    auto px = std::make_shared<std::vector<uint8_t>>((size_t)w * (size_t)h * 3u);
    if (fread(px->data(), 1, px->size(), f) != px->size())
        throw std::runtime_error("load_image_rgb8: " + path + ": truncated PPM");
    ImageRGB8 img;
    img.width = w; img.height = h; img.pixels = px;
    return img;
}

static inline int paeth_predictor(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// PNG (ISO 15948): grey, grey + alpha, RGB, palette, RGBA at every bit depth the standard allows, plain or Adam7-interlaced.  The
// reference reads a texel with DynamicImage::get_pixel (src/texture.rs:89), i.e. as RGBA8 whatever the file holds: grey is replicated
// (1 / 2 / 4-bit grey scaled to 0..255), a palette looked up, alpha dropped ([r, g, b, _]), a 16-bit sample v becomes (v + 128) / 257 —
// round(v * 255 / 65535), the `image` crate's u16 -> u8 conversion.
static ImageRGB8 decode_png(const std::string &path, const std::vector<uint8_t> &file) {
    auto fail = [&](const char *what) -> ImageRGB8 { throw std::runtime_error("load_image_rgb8: " + path + ": PNG: " + what); };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) return fail("bad signature");
    auto be32 = [&](size_t at) { return ((uint32_t)file[at] << 24) | ((uint32_t)file[at + 1] << 16) | ((uint32_t)file[at + 2] << 8) | file[at + 3]; };
    uint32_t width = 0, height = 0;
    int depth = 0, colour = -1, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool seen_end = false;
    for (size_t at = 8; at + 12 <= file.size() && !seen_end;) {
        const uint32_t len = be32(at);
        if ((uint64_t)at + 12u + len > file.size()) return fail("truncated chunk");
        const char *type = reinterpret_cast<const char *>(file.data() + at + 4);
        const uint8_t *data = file.data() + at + 8;
        if ((uint32_t)crc32(0L, file.data() + at + 4, (uInt)(len + 4)) != be32(at + 8 + len)) return fail("chunk CRC mismatch");
        if (memcmp(type, "IHDR", 4) == 0) {
            if (len != 13) return fail("bad IHDR");
            width = be32(at + 8); height = be32(at + 12);
            depth = data[8]; colour = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return fail("unknown compression or filter method");
        } else if (memcmp(type, "PLTE", 4) == 0) {
            palette.assign(data, data + len);
        } else if (memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (memcmp(type, "IEND", 4) == 0) {
            seen_end = true;
        }
        at += 12u + (size_t)len;
    }
    if (colour < 0 || width == 0 || height == 0 || width > 65535u || height > 65535u) return fail("missing or bad IHDR");
    if ((uint64_t)width * (uint64_t)height > MAX_IMAGE_PIXELS) return fail("image too large");
    if (interlace != 0 && interlace != 1) return fail("unknown interlace method");
    int channels = 0;
    bool depth_ok = false;
    switch (colour) {
    case 0: channels = 1; depth_ok = depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16; break;
    case 2: channels = 3; depth_ok = depth == 8 || depth == 16; break;
    case 3:
        channels = 1; depth_ok = depth == 1 || depth == 2 || depth == 4 || depth == 8;
        if (palette.size() < 3 || palette.size() % 3 != 0) return fail("palette image without a PLTE chunk");
        break;
    case 4: channels = 2; depth_ok = depth == 8 || depth == 16; break;
    case 6: channels = 4; depth_ok = depth == 8 || depth == 16; break;
    default: return fail("unknown colour type");
    }
    if (!depth_ok) return fail("bit depth not allowed for the colour type");
    const size_t bits_pp = (size_t)channels * (size_t)depth;  // bits per pixel
    const size_t fbpp = bits_pp >= 8 ? bits_pp / 8 : 1;        // the filters' "corresponding byte of the pixel to the left"
    // the image as a sequence of reduced images (ISO 15948 section 8.2): one for a plain file, seven for Adam7
    struct Pass { uint32_t x0, y0, dx, dy, w, h; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, width, height});
    else {
        static const uint32_t A7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
        for (const auto &a : A7) {
            const uint32_t pw = width > a[0] ? (width - a[0] + a[2] - 1) / a[2] : 0, ph = height > a[1] ? (height - a[1] + a[3] - 1) / a[3] : 0;
            if (pw && ph) passes.push_back({a[0], a[1], a[2], a[3], pw, ph});
        }
    }
    size_t total = 0;
    for (const Pass &ps : passes) total += (((size_t)ps.w * bits_pp + 7) / 8 + 1u) * (size_t)ps.h;
    std::vector<uint8_t> raw(total);
    uLongf raw_len = (uLongf)raw.size();
    const int zrc = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || raw_len != raw.size()) return fail("image data does not inflate to the size IHDR announces");
    auto px = std::make_shared<std::vector<uint8_t>>((size_t)width * (size_t)height * 3u);
    auto to8 = [&](uint32_t v) -> uint8_t { // a sample of `depth` bits as the `image` crate hands it out in 8
        switch (depth) {
        case 1: return (uint8_t)(v * 255u);
        case 2: return (uint8_t)(v * 85u);
        case 4: return (uint8_t)(v * 17u);
        case 8: return (uint8_t)v;
        default: return (uint8_t)((v + 128u) / 257u);
        }
    };
    size_t at_raw = 0;
    for (const Pass &ps : passes) {
        const size_t stride = ((size_t)ps.w * bits_pp + 7) / 8;
        const std::vector<uint8_t> zero(stride, 0);
        for (uint32_t y = 0; y < ps.h; ++y) {
            // undo the row's filter in place (ISO 15948 section 9)
            uint8_t *row = raw.data() + at_raw + (size_t)y * (stride + 1u);
            const int filter = row[0];
            uint8_t *cur = row + 1;
            const uint8_t *up = y ? row - stride : zero.data();
            if (filter < 0 || filter > 4) return fail("unknown row filter");
            for (size_t x = 0; x < stride; ++x) {
                const int a = x >= fbpp ? cur[x - fbpp] : 0, b = up[x], c = x >= fbpp ? up[x - fbpp] : 0;
                int v = cur[x];
                switch (filter) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth_predictor(a, b, c); break;
                default: break;
                }
                cur[x] = (uint8_t)v;
            }
            auto sample = [&](uint32_t i, int ch) -> uint32_t { // channel ch of the row's pixel i, as stored
                if (depth == 8) return cur[(size_t)i * (size_t)channels + (size_t)ch];
                if (depth == 16) { const uint8_t *q = cur + ((size_t)i * (size_t)channels + (size_t)ch) * 2u; return ((uint32_t)q[0] << 8) | q[1]; }
                const size_t bit = (size_t)i * (size_t)depth; // (one channel below 8 bits) samples are packed from the high bit down
                return (cur[bit >> 3] >> (8u - (unsigned)depth - (bit & 7u))) & ((1u << depth) - 1u);
            };
            uint8_t *dst_row = px->data() + (size_t)(ps.y0 + y * ps.dy) * (size_t)width * 3u;
            for (uint32_t i = 0; i < ps.w; ++i) {
                uint8_t r, g, b;
                if (colour == 0 || colour == 4) { r = g = b = to8(sample(i, 0)); }
                else if (colour == 3) {
                    const size_t idx = sample(i, 0);
                    if (idx * 3u + 2u >= palette.size()) return fail("palette index out of range");
                    r = palette[idx * 3u]; g = palette[idx * 3u + 1u]; b = palette[idx * 3u + 2u];
                } else { r = to8(sample(i, 0)); g = to8(sample(i, 1)); b = to8(sample(i, 2)); }
                uint8_t *dst = dst_row + (size_t)(ps.x0 + i * ps.dx) * 3u;
                dst[0] = r; dst[1] = g; dst[2] = b;
            }
        }
        at_raw += (stride + 1u) * (size_t)ps.h;
    }
    ImageRGB8 img;
    img.width = (int32_t)width; img.height = (int32_t)height; img.pixels = px;
    return img;
}

ImageRGB8 load_image_rgb8(const std::string &path) {
    if (path.rfind("synthetic:", 0) == 0) {
        int w = 0, h = 0;
        if (sscanf(path.c_str() + 10, "%dx%d", &w, &h) != 2)
            throw std::runtime_error("load_image_rgb8: expected synthetic:WxH, got " + path);
        return synthetic_earth(w, h);
    }
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("load_image_rgb8: cannot open " + path);
    unsigned char head[2] = {0, 0};
    size_t n = fread(head, 1, 2, f);
    rewind(f);
    try {
        ImageRGB8 img;
        if (n == 2 && head[0] == 'P' && head[1] == '6') {
            img = load_ppm(path, f);
        } else if (n == 2 && ((head[0] == 0xFF && head[1] == 0xD8) || (head[0] == 0x89 && head[1] == 'P'))) {
            std::vector<uint8_t> bytes;
            fseek(f, 0, SEEK_END);
            long sz = ftell(f);
            rewind(f);
            bytes.resize((size_t)sz);
            if (fread(bytes.data(), 1, bytes.size(), f) != bytes.size())
                throw std::runtime_error("load_image_rgb8: short read on " + path);
            img = head[0] == 0xFF ? decode_jpeg(bytes.data(), bytes.size()) : decode_png(path, bytes);
        } else {
            throw std::runtime_error("load_image_rgb8: " + path + ": unsupported format (PPM P6, JPEG or PNG)");
        }
        fclose(f);
        return img;
    } catch (...) {
        fclose(f);
        throw;
    }
}

// ---- PNG ------------------------------------------------------------------------------------------------
static void put_u32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void put_chunk(std::vector<uint8_t> &out, const char type[4], const uint8_t *data, size_t len) {
    put_u32(out, (uint32_t)len);
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + start, (uInt)(len + 4));
    put_u32(out, crc);
}
static inline int paeth(int a, int b, int c) { return paeth_predictor(a, b, c); }

bool write_png_rgb8(const std::string &path, int32_t width, int32_t height, const uint8_t *rgb) {
    if (width <= 0 || height <= 0 || !rgb) return false;
    const size_t stride = (size_t)width * 3u;
    std::vector<uint8_t> raw((stride + 1u) * (size_t)height);
    std::vector<uint8_t> cand(stride), zero(stride, 0);
    for (int32_t y = 0; y < height; ++y) {
        const uint8_t *cur = rgb + (size_t)y * stride;
        const uint8_t *up = y ? rgb + (size_t)(y - 1) * stride : zero.data();
        // adaptive filter: minimum sum of absolute (signed) residuals
        int best_f = 0; uint64_t best_cost = ~0ull;
        for (int f = 0; f < 5; ++f) {
            uint64_t cost = 0;
            for (size_t x = 0; x < stride; ++x) {
                int a = x >= 3 ? cur[x - 3] : 0, b = up[x], c = x >= 3 ? up[x - 3] : 0, v = cur[x], r;
                switch (f) {
                case 0: r = v; break;
                case 1: r = v - a; break;
                case 2: r = v - b; break;
                case 3: r = v - ((a + b) >> 1); break;
                default: r = v - paeth(a, b, c); break;
                }
                cost += (uint64_t)std::abs((int)(int8_t)(uint8_t)r);
            }
            if (cost < best_cost) { best_cost = cost; best_f = f; }
        }
        uint8_t *dst = raw.data() + (size_t)y * (stride + 1u);
        dst[0] = (uint8_t)best_f;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= 3 ? cur[x - 3] : 0, b = up[x], c = x >= 3 ? up[x - 3] : 0, v = cur[x], r;
            switch (best_f) {
            case 0: r = v; break;
            case 1: r = v - a; break;
            case 2: r = v - b; break;
            case 3: r = v - ((a + b) >> 1); break;
            default: r = v - paeth(a, b, c); break;
            }
            dst[1 + x] = (uint8_t)r;
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), Z_DEFAULT_COMPRESSION) != Z_OK) return false;

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_u32(ihdr, (uint32_t)width); put_u32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(out, "IDAT", z.data(), (size_t)zlen);
    put_chunk(out, "IEND", nullptr, 0);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    ok = (fclose(f) == 0) && ok;
    return ok;
}

} // namespace rt
