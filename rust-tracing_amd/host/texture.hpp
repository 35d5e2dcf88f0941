// The Texture trait and its four implementations, host side: construction + describe().  `value()` is
// evaluated on the GPU (csrc/rt_kernel.hip); the host objects only hold what the reference's structs hold.
//   reference: src/texture.rs:12-111
#pragma once
#include "describe.hpp"
#include "image_io.hpp"
#include "perlin.hpp"
#include "vec3.hpp"
#include <memory>
#include <string>

namespace rt {

class Texture {
  public:
    virtual ~Texture() = default;
    // appends this texture's rt_texture record (once) and returns its index
    int32_t describe(SceneDescriber &sd) const {
        auto it = sd.seen_textures.find(this);
        if (it != sd.seen_textures.end()) return it->second;
        rt_texture t = record(sd);
        int32_t idx = (int32_t)sd.textures.size();
        sd.textures.push_back(t);
        sd.seen_textures.emplace(this, idx);
        return idx;
    }

  protected:
    virtual rt_texture record(SceneDescriber &sd) const = 0;
    static rt_texture blank(int32_t kind) {
        rt_texture t{};
        t.kind = kind;
        t.even = t.odd = t.image = t.perlin = -1;
        return t;
    }
};

// src/texture.rs:16-37
class SolidColor : public Texture {
  public:
    SolidColor(FP red, FP green, FP blue) : color(red, green, blue) {}
    explicit SolidColor(const Color &c) : color(c) {}
    Color color;

  protected:
    rt_texture record(SceneDescriber &) const override {
        rt_texture t = blank(RT_TEXTURE_SOLID);
        t.color = color.pod();
        return t;
    }
};

// src/texture.rs:39-70
class CheckerTexture : public Texture {
  public:
    CheckerTexture(FP scale, std::shared_ptr<Texture> even_, std::shared_ptr<Texture> odd_)
        : inv_scale(1.0 / scale), even(std::move(even_)), odd(std::move(odd_)) {}
    static std::shared_ptr<CheckerTexture> new_from_colors(FP scale, const Color &even, const Color &odd) {
        return std::make_shared<CheckerTexture>(scale, std::make_shared<SolidColor>(even),
                                                std::make_shared<SolidColor>(odd));
    }
    FP inv_scale;
    std::shared_ptr<Texture> even, odd;

  protected:
    rt_texture record(SceneDescriber &sd) const override {
        rt_texture t = blank(RT_TEXTURE_CHECKER);
        t.inv_scale = inv_scale;
        t.even = even->describe(sd);
        t.odd = odd->describe(sd);
        return t;
    }
};

// src/texture.rs:72-93.  The reference decodes with the `image` crate (src/texture.rs:78); here the decode
// is image_io.hpp's (PPM, JPEG and PNG, or a procedural stand-in "synthetic:WxH").
class ImageTexture : public Texture {
  public:
    explicit ImageTexture(const std::string &path) : image(load_image_rgb8(path)) {}
    explicit ImageTexture(ImageRGB8 img) : image(std::move(img)) {}
    ImageRGB8 image;

  protected:
    rt_texture record(SceneDescriber &sd) const override {
        rt_texture t = blank(RT_TEXTURE_IMAGE);
        rt_image im{};
        im.width = image.width;
        im.height = image.height;
        im.rgb = image.pixels->data();
        sd.image_storage.push_back(image.pixels);
        t.image = (int32_t)sd.images.size();
        sd.images.push_back(im);
        return t;
    }
};

// src/texture.rs:95-111
class NoiseTexture : public Texture {
  public:
    explicit NoiseTexture(FP scale_) : noise(), scale(scale_) {}
    Perlin noise;
    FP scale;

  protected:
    rt_texture record(SceneDescriber &sd) const override {
        rt_texture t = blank(RT_TEXTURE_NOISE);
        t.scale = scale;
        t.perlin = (int32_t)sd.perlins.size();
        sd.perlins.push_back(noise.pod());
        return t;
    }
};

} // namespace rt
