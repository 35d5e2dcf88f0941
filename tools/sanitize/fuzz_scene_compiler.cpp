// CPU.  Scene-description fuzzer: the nine reference scenes' rt_scene_desc (built by the host library), copied into mutable arrays and
// damaged — indices, counts, kinds; NaN, inf, 0, 1e300 in the doubles — then through the part of rt_scene_create that needs no GPU:
// compile_scene, build_ordered (two- and four-child records, random leaf sizes), qfilt_table.  Built with AddressSanitizer + UBSan
// (+ float-cast-overflow) by tools/fuzz_scene_compiler.sh: a damaged description must end in an exception (rt_scene_create turns it
// into RT_ERR_INVALID_ARGUMENT), never in a fault.  Images are left alone: a pointer and its extent are the caller's word.
// Round 4: 27 000 damaged descriptions, half refused, half compiled, no finding.
#include "rt_host.h"
#include "rt_ordered.hpp"
#include "rt_qfilt.hpp"
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>
using namespace rtd;

struct Owned {
    rt_scene_desc d;
    std::vector<std::vector<uint8_t>> bufs;
    template <class T> void own(const T *&p, int32_t n) {
        bufs.emplace_back((size_t)(n > 0 ? n : 0) * sizeof(T) + 64);
        if (n > 0 && p) memcpy(bufs.back().data(), p, (size_t)n * sizeof(T));
        p = reinterpret_cast<const T *>(bufs.back().data());
    }
    explicit Owned(const rt_scene_desc &src) : d(src) {
        own(d.spheres, d.n_spheres); own(d.quads, d.n_quads); own(d.lists, d.n_lists); own(d.list_items, d.n_list_items);
        own(d.translates, d.n_translates); own(d.rotates, d.n_rotates); own(d.bvh_nodes, d.n_bvh_nodes); own(d.bvhs, d.n_bvhs);
        own(d.media, d.n_media); own(d.materials, d.n_materials); own(d.textures, d.n_textures); own(d.perlins, d.n_perlins);
        own(d.images, d.n_images);
    }
};

int main(int argc, char **argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int per_scene = argc > 2 ? atoi(argv[2]) : 200;
    std::mt19937_64 rng(seed);
    long ok = 0, refused = 0;
    for (int scene = 0; scene <= 8; ++scene) {
        rth_scene_options o{};
        o.scene = scene; o.scene_seed = seed; o.image_width = 64; o.samples_per_pixel = 1; o.max_depth = 4; o.earth_image = "synthetic:64x32";
        rth_scene *hs = nullptr;
        if (rth_scene_build(&o, &hs) != 0) { fprintf(stderr, "scene %d: %s\n", scene, rth_last_error()); return 2; }
        const rt_scene_desc *base = rth_scene_desc(hs);
        for (int it = 0; it < per_scene; ++it) {
            Owned m(*base);
            const int n_mut = it == 0 ? 0 : 1 + (int)(rng() % 3);
            for (int k = 0; k < n_mut; ++k) {
                const int what = (int)(rng() % 10);
                if (what == 0) { // a count (never beyond what the arrays hold: the ABI says the caller's arrays have that many)
                    int32_t *counts = &m.d.n_spheres;
                    int32_t &c = counts[rng() % 13];
                    c = c > 0 ? (int32_t)(rng() % (uint64_t)c) : 0;
                } else if (what == 1) { m.d.world.kind = (int32_t)(rng() % 10) - 1; }
                else if (what == 2) { m.d.world.index = (int32_t)(rng() % 4000) - 8; }
                else { // a random 4-byte or 8-byte word of a random table
                    auto &b = m.bufs[rng() % (m.bufs.size() - 1)]; // (not the images: a pointer and its extent are the caller's word)
                    if (b.size() <= 72) continue;
                    const size_t at = (rng() % ((b.size() - 64) / 4)) * 4;
                    const int how = (int)(rng() % 8);
                    int32_t iv = 0; double dv = 0;
                    switch (how) {
                    case 0: iv = -1; memcpy(&b[at], &iv, 4); break;
                    case 1: iv = (int32_t)(rng() % 100000); memcpy(&b[at], &iv, 4); break;
                    case 2: iv = std::numeric_limits<int32_t>::max(); memcpy(&b[at], &iv, 4); break;
                    case 3: iv = std::numeric_limits<int32_t>::min(); memcpy(&b[at], &iv, 4); break;
                    case 4: dv = std::numeric_limits<double>::quiet_NaN(); if (at + 8 <= b.size() - 64) memcpy(&b[at & ~7ull], &dv, 8); break;
                    case 5: dv = std::numeric_limits<double>::infinity(); if (at + 8 <= b.size() - 64) memcpy(&b[at & ~7ull], &dv, 8); break;
                    case 6: dv = 0.0; if (at + 8 <= b.size() - 64) memcpy(&b[at & ~7ull], &dv, 8); break;
                    default: dv = 1e300; if (at + 8 <= b.size() - 64) memcpy(&b[at & ~7ull], &dv, 8); break;
                    }
                }
            }
            for (int wide = 0; wide < 2; ++wide) {
                try {
                    CompiledScene cs = compile_scene(m.d, (it & 1) != 0);
                    OrderedOptions oo; oo.wide = wide != 0; oo.leaf_max = 1 + (uint32_t)(rng() % 8); oo.flat_max = (uint32_t)(rng() % 9);
                    build_ordered(cs, oo);
                    const std::vector<QFiltPair> qf = qfilt_table(cs.quads);
                    ok += (long)(qf.size() >= 0);
                } catch (const std::exception &e) { ++refused; if (refused < 12 || it == 0) fprintf(stderr, "scene %d it %d wide %d: %s\n", scene, it, wide, e.what()); }
            }
        }
        rth_scene_destroy(hs);
    }
    printf("compiled %ld, refused %ld\n", ok, refused);
    return 0;
}
