// Image ingest for ImageTexture (reference: image::io::Reader::open(path).decode(), src/texture.rs:78) and
// PNG output for render() (reference: image::codecs::png::PngEncoder, src/renderer.rs:59-72).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace rt {

// The readers refuse a file that announces more pixels than this before allocating anything for it (2^28: a 16 384 x 16 384 texture)
constexpr uint64_t MAX_IMAGE_PIXELS = 1ull << 28;

struct ImageRGB8 {
    int32_t width = 0, height = 0;
    std::shared_ptr<const std::vector<uint8_t>> pixels; // row-major RGB8, row 0 = top
};

// Accepts: "synthetic:WxH" (deterministic integer-only procedural earth-like map, used where the reference's
// assets/earth-large.jpg is not available), binary PPM (P6, maxval 255), JPEG (sequential or progressive, 8-bit) and
// PNG (every bit depth, plain or Adam7-interlaced).
// Throws std::runtime_error on failure (the reference panics: `.unwrap()`, src/texture.rs:78).
ImageRGB8 load_image_rgb8(const std::string &path);

ImageRGB8 synthetic_earth(int32_t width, int32_t height);

// RGB8 PNG, zlib-deflated, adaptive per-row filter.  Returns false on I/O failure.
bool write_png_rgb8(const std::string &path, int32_t width, int32_t height, const uint8_t *rgb);

} // namespace rt
