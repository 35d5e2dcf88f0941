// JPEG decode (sequential and progressive DCT, Huffman, 8-bit) for ImageTexture — stands in for the `image` crate's decoder the reference
// calls at src/texture.rs:78.  Decoder output is not pinned by the reference (different decoders differ by
// +-1 level on chroma-subsampled files); see DESIGN.md "Image ingest".
#pragma once
#include "image_io.hpp"
#include <cstddef>
#include <cstdint>

namespace rt {
// Throws std::runtime_error on malformed or unsupported (arithmetic-coded, lossless, 12-bit, CMYK) streams.
ImageRGB8 decode_jpeg(const uint8_t *data, size_t size);
} // namespace rt
