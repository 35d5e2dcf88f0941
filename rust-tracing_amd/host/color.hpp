// Output-stage conversion.  reference: src/color.rs:3-19
#pragma once
#include "vec3.hpp"
#include <array>

namespace rt {

inline FP linear_to_gamma(FP linear_component) { return std::pow(linear_component, 1.0 / 2.2); }

// `(256.0 * x.clamp(0.0, 0.999)) as u8`: f64::clamp keeps NaN, and `NaN as u8` is 0 (Rust saturating cast)
inline uint8_t quantise(FP gamma_component) {
    if (gamma_component != gamma_component) return 0;
    FP c = gamma_component < 0.0 ? 0.0 : (gamma_component > 0.999 ? 0.999 : gamma_component);
    return (uint8_t)(256.0 * c);
}

inline std::array<uint8_t, 3> color_to_rgb(const Color &rgb) {
    return {quantise(linear_to_gamma(rgb.x)), quantise(linear_to_gamma(rgb.y)), quantise(linear_to_gamma(rgb.z))};
}

} // namespace rt
