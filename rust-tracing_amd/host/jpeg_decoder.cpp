#include "jpeg_decoder.hpp"
#include <stdexcept>

namespace rt {
ImageRGB8 decode_baseline_jpeg(const uint8_t *, size_t) {
    throw std::runtime_error("decode_baseline_jpeg: JPEG ingest is not built yet; convert the texture to "
                             "binary PPM (P6) or use synthetic:WxH");
}
} // namespace rt
