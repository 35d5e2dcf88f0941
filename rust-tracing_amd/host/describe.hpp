// SceneDescriber: turns the host object graph (Hittable / Material / Texture objects) into the POD arrays of
// rt_scene_desc (include/rt_amd.h).  This is the one surface method the GPU path adds to the reference's
// three traits: `describe()` — each object appends its own record and returns its index; shared objects
// (Arc clones in the reference, shared_ptr here) are described once and referenced by index.
#pragma once
#include "rt_amd.h"
#include <cstdint>
#include <memory>
#include <unordered_map>
#include <vector>

namespace rt {

class SceneDescriber {
  public:
    std::vector<rt_sphere> spheres;
    std::vector<rt_quad> quads;
    std::vector<rt_list> lists;
    std::vector<rt_ref> list_items;
    std::vector<rt_translate> translates;
    std::vector<rt_rotate_y> rotates;
    std::vector<rt_bvh_node> bvh_nodes;
    std::vector<rt_bvh> bvhs;
    std::vector<rt_constant_medium> media;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_perlin> perlins;
    std::vector<rt_image> images;
    // keeps image pixels alive for as long as the description is
    std::vector<std::shared_ptr<const std::vector<uint8_t>>> image_storage;

    // identity maps (object address -> record) so that a shared object is described once
    std::unordered_map<const void *, rt_ref> seen_hittables;
    std::unordered_map<const void *, int32_t> seen_materials;
    std::unordered_map<const void *, int32_t> seen_textures;

    rt_scene_desc desc(rt_ref world) const {
        rt_scene_desc d{};
        d.abi_version = RT_ABI_VERSION;
        d.world = world;
        d.n_spheres = (int32_t)spheres.size();
        d.n_quads = (int32_t)quads.size();
        d.n_lists = (int32_t)lists.size();
        d.n_list_items = (int32_t)list_items.size();
        d.n_translates = (int32_t)translates.size();
        d.n_rotates = (int32_t)rotates.size();
        d.n_bvh_nodes = (int32_t)bvh_nodes.size();
        d.n_bvhs = (int32_t)bvhs.size();
        d.n_media = (int32_t)media.size();
        d.n_materials = (int32_t)materials.size();
        d.n_textures = (int32_t)textures.size();
        d.n_perlins = (int32_t)perlins.size();
        d.n_images = (int32_t)images.size();
        d.spheres = spheres.data();
        d.quads = quads.data();
        d.lists = lists.data();
        d.list_items = list_items.data();
        d.translates = translates.data();
        d.rotates = rotates.data();
        d.bvh_nodes = bvh_nodes.data();
        d.bvhs = bvhs.data();
        d.media = media.data();
        d.materials = materials.data();
        d.textures = textures.data();
        d.perlins = perlins.data();
        d.images = images.data();
        return d;
    }
};

} // namespace rt
