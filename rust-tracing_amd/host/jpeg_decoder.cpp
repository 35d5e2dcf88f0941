// Baseline JPEG decoder (ITU T.81 sequential DCT, Huffman, 8-bit) for ImageTexture ingest.
//
// The reference decodes its texture with the `image` crate (src/texture.rs:78).  Decoder output is not part of any
// contract the reference states (decoders differ by +-1 level on chroma-subsampled files), so this one follows the
// most widely deployed arithmetic — the IJG/libjpeg "islow" integer IDCT, "fancy" (triangle) chroma upsampling and
// 16-bit fixed-point YCbCr->RGB — which lets tests/test_host.py compare it with Pillow's decode of the same file.
#include "jpeg_decoder.hpp"

#include <array>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt {
namespace {

[[noreturn]] void bad(const std::string &m) { throw std::runtime_error("decode_baseline_jpeg: " + m); }

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    // canonical decoding (T.81 F.2.2.3): per code length the smallest code, the largest code and the value index
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t values[256];
    // 9-bit fast lookup: (length << 8) | value, 0 = not resolved
    uint16_t fast[512];

    void build(const uint8_t counts[16], const uint8_t *vals, int nvals) {
        memcpy(values, vals, (size_t)nvals);
        int code = 0, k = 0;
        memset(fast, 0, sizeof fast);
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k;
            mincode[len] = code;
            for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
                if (len <= 9) {
                    const int shift = 9 - len;
                    for (int fill = 0; fill < (1 << shift); ++fill) fast[(code << shift) | fill] = (uint16_t)((len << 8) | values[k]);
                }
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0; // allocated size in 8x8 blocks (padded to whole MCUs)
    int width = 0, height = 0;      // real (downsampled) size in samples
    int pred = 0;
    std::vector<uint8_t> plane;     // blocks_w*8 x blocks_h*8 samples
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;

    void fill() {
        while (bits <= 24) {
            int byte = 0;
            if (!hit_marker && p < end) {
                byte = *p++;
                if (byte == 0xFF) {
                    if (p < end && *p == 0x00) { ++p; }             // stuffed zero
                    else { hit_marker = true; --p; byte = 0; }      // a marker: feed zeros from here on
                }
            }
            acc |= (uint32_t)byte << (24 - bits);
            bits += 8;
        }
    }
    int peek(int n) { if (bits < n) fill(); return (int)(acc >> (32 - n)); }
    void skip(int n) { acc <<= n; bits -= n; }
    int get(int n) { if (n == 0) return 0; int v = peek(n); skip(n); return v; }
    void reset() { acc = 0; bits = 0; hit_marker = false; }
};

int decode_symbol(BitReader &br, const HuffTable &t) {
    const int look = br.peek(9);
    const uint16_t f = t.fast[look];
    if (f) { br.skip(f >> 8); return f & 0xff; }
    int code = br.peek(16);
    for (int len = 10; len <= 16; ++len) {
        const int c = code >> (16 - len);
        if (t.maxcode[len] >= 0 && c <= t.maxcode[len] && c >= t.mincode[len]) {
            br.skip(len);
            return t.values[t.valptr[len] + c - t.mincode[len]];
        }
    }
    bad("corrupt Huffman code");
}

inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; } // T.81 F.2.2.1

// IJG jidctint.c ("islow"): 13-bit constants, 2 extra bits kept after the column pass
constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int32_t FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270,
                  FIX_0_899976223 = 7373, FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137,
                  FIX_1_961570560 = 16069, FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;
inline int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }
inline uint8_t clamp255(int32_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void idct_islow(const int32_t coef[64], uint8_t *out, int stride) {
    int32_t ws[64];
    for (int c = 0; c < 8; ++c) { // pass 1: columns
        const int32_t *in = coef + c;
        int32_t z2 = in[16], z3 = in[48];
        int32_t z1 = (z2 + z3) * FIX_0_541196100;
        int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        int32_t tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = in[0]; z3 = in[32];
        int32_t tmp0 = (z2 + z3) * (1 << CONST_BITS), tmp1 = (z2 - z3) * (1 << CONST_BITS);
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        int32_t *w = ws + c;
        w[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);  w[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        w[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);  w[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        w[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS); w[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        w[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS); w[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) { // pass 2: rows
        const int32_t *w = ws + r * 8;
        int32_t z2 = w[2], z3 = w[6];
        int32_t z1 = (z2 + z3) * FIX_0_541196100;
        int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        int32_t tmp3 = z1 + z2 * FIX_0_765366865;
        int32_t tmp0 = (w[0] + w[4]) * (1 << CONST_BITS), tmp1 = (w[0] - w[4]) * (1 << CONST_BITS);
        const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        const int32_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        uint8_t *o = out + r * stride;
        const int sh = CONST_BITS + PASS1_BITS + 3;
        o[0] = clamp255(descale(tmp10 + tmp3, sh) + 128); o[7] = clamp255(descale(tmp10 - tmp3, sh) + 128);
        o[1] = clamp255(descale(tmp11 + tmp2, sh) + 128); o[6] = clamp255(descale(tmp11 - tmp2, sh) + 128);
        o[2] = clamp255(descale(tmp12 + tmp1, sh) + 128); o[5] = clamp255(descale(tmp12 - tmp1, sh) + 128);
        o[3] = clamp255(descale(tmp13 + tmp0, sh) + 128); o[4] = clamp255(descale(tmp13 - tmp0, sh) + 128);
    }
}

struct Decoder {
    const uint8_t *data;
    size_t size, pos = 2;
    int width = 0, height = 0, max_h = 1, max_v = 1, restart_interval = 0;
    uint16_t quant[4][64];
    bool quant_present[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    std::vector<Component> comps;
    bool have_frame = false;
    int adobe_transform = -1;

    Decoder(const uint8_t *d, size_t n) : data(d), size(n) {}

    int u8() { if (pos >= size) bad("truncated"); return data[pos++]; }
    int u16() { int a = u8(); return (a << 8) | u8(); }

    void read_dqt(int len) {
        size_t stop = pos + (size_t)len;
        while (pos < stop) {
            int pq_tq = u8();
            int pq = pq_tq >> 4, tq = pq_tq & 15;
            if (tq > 3) bad("bad quantisation table id");
            for (int i = 0; i < 64; ++i) quant[tq][ZIGZAG[i]] = (uint16_t)(pq ? u16() : u8());
            quant_present[tq] = true;
        }
    }
    void read_dht(int len) {
        size_t stop = pos + (size_t)len;
        while (pos < stop) {
            int tc_th = u8();
            int tc = tc_th >> 4, th = tc_th & 15;
            if (tc > 1 || th > 3) bad("bad Huffman table id");
            uint8_t counts[16];
            int n = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)u8(); n += counts[i]; }
            if (n > 256 || pos + (size_t)n > size) bad("bad Huffman table");
            (tc ? ac[th] : dc[th]).build(counts, data + pos, n);
            pos += (size_t)n;
        }
    }
    void read_sof(int len) {
        (void)len;
        if (u8() != 8) bad("only 8-bit precision is supported");
        height = u16(); width = u16();
        int nc = u8();
        if (width <= 0 || height <= 0) bad("empty image");
        if (nc != 1 && nc != 3) bad("only grayscale and 3-component images are supported");
        comps.resize((size_t)nc);
        for (auto &c : comps) {
            c.id = u8();
            int hv = u8();
            c.h = hv >> 4; c.v = hv & 15; c.tq = u8();
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) bad("unsupported sampling factors");
            max_h = c.h > max_h ? c.h : max_h;
            max_v = c.v > max_v ? c.v : max_v;
        }
        if (nc == 1) { comps[0].h = comps[0].v = 1; max_h = max_v = 1; }
        const int mcus_x = (width + 8 * max_h - 1) / (8 * max_h), mcus_y = (height + 8 * max_v - 1) / (8 * max_v);
        for (auto &c : comps) {
            c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
            c.width = (width * c.h + max_h - 1) / max_h; c.height = (height * c.v + max_v - 1) / max_v;
            c.plane.assign((size_t)c.blocks_w * 8 * (size_t)c.blocks_h * 8, 0);
        }
        have_frame = true;
    }

    void decode_block(BitReader &br, Component &c, int bx, int by) {
        const HuffTable &hd = dc[c.td], &ha = ac[c.ta];
        if (!hd.present || !ha.present || !quant_present[c.tq]) bad("missing table");
        int32_t coef[64] = {0};
        int t = decode_symbol(br, hd);
        if (t > 11) bad("bad DC size");
        int diff = t ? extend(br.get(t), t) : 0;
        c.pred += diff;
        coef[0] = c.pred * quant[c.tq][0];
        for (int k = 1; k < 64;) {
            int rs = decode_symbol(br, ha);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break; // EOB
            }
            k += r;
            if (k > 63) bad("AC run overflows the block");
            const int z = ZIGZAG[k];
            coef[z] = extend(br.get(s), s) * quant[c.tq][z];
            ++k;
        }
        idct_islow(coef, c.plane.data() + ((size_t)by * 8 * c.blocks_w + (size_t)bx) * 8, c.blocks_w * 8);
    }

    void read_scan(int len) {
        (void)len;
        if (!have_frame) bad("SOS before SOF");
        int ns = u8();
        if (ns != (int)comps.size()) bad("only single-scan (interleaved) files are supported");
        for (int i = 0; i < ns; ++i) {
            int id = u8(), tdta = u8();
            Component *c = nullptr;
            for (auto &x : comps) if (x.id == id) c = &x;
            if (!c) bad("scan names an unknown component");
            c->td = tdta >> 4; c->ta = tdta & 15;
            if (c->td > 3 || c->ta > 3) bad("bad table selector");
        }
        int ss = u8(), se = u8(), ahal = u8();
        if (ss != 0 || se != 63 || ahal != 0) bad("progressive scans are not supported");
        BitReader br{data + pos, data + size};
        const int mcus_x = (width + 8 * max_h - 1) / (8 * max_h), mcus_y = (height + 8 * max_v - 1) / (8 * max_v);
        int until_restart = restart_interval;
        for (auto &c : comps) c.pred = 0;
        for (int my = 0; my < mcus_y; ++my)
            for (int mx = 0; mx < mcus_x; ++mx) {
                if (restart_interval && until_restart == 0) {
                    // byte-align, expect RSTn
                    const uint8_t *q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                    if (q + 1 >= br.end) bad("missing restart marker");
                    br.p = q + 2;
                    br.reset();
                    for (auto &c : comps) c.pred = 0;
                    until_restart = restart_interval;
                }
                for (auto &c : comps)
                    for (int v = 0; v < c.v; ++v)
                        for (int h = 0; h < c.h; ++h) decode_block(br, c, mx * c.h + h, my * c.v + v);
                if (restart_interval) --until_restart;
            }
        pos = (size_t)(br.p - data);
    }

    // IJG jdsample.c: h2v1 / h2v2 "fancy" (triangle filter) upsampling, plain replication otherwise
    std::vector<uint8_t> upsample(const Component &c) const {
        std::vector<uint8_t> out((size_t)width * (size_t)height);
        const int stride = c.blocks_w * 8;
        const int hs = max_h / c.h, vs = max_v / c.v;
        auto row = [&](int y) { return c.plane.data() + (size_t)(y < 0 ? 0 : (y >= c.height ? c.height - 1 : y)) * stride; };
        if (hs == 1 && vs == 1) {
            for (int y = 0; y < height; ++y) memcpy(&out[(size_t)y * width], row(y), (size_t)width);
            return out;
        }
        const int dw = c.width;
        std::vector<uint8_t> line((size_t)dw * 2 + 2);
        if (dw <= 2 || hs == 1) { // IJG uses the triangle filter only when the component is more than 2 samples wide
            for (int y = 0; y < height; ++y) {
                const uint8_t *in = row(vs == 2 ? y / 2 : y);
                for (int x = 0; x < width; ++x) out[(size_t)y * width + x] = in[hs == 2 ? x / 2 : x];
            }
            return out;
        }
        for (int y = 0; y < height; ++y) {
            const int iy = vs == 2 ? y / 2 : y;
            const uint8_t *in0 = row(iy);
            const uint8_t *in1 = vs == 2 ? ((y & 1) ? row(iy + 1) : row(iy - 1)) : nullptr;
            if (hs == 2) {
                // column sums: 3*near + far (h2v2) or 4*sample (h2v1 scaled to the same 1/16 grid)
                auto colsum = [&](int x) { return vs == 2 ? in0[x] * 3 + in1[x] : in0[x] * 4; };
                // h2v1 uses (3*this + neighbour + 1 or 2) >> 2 on raw samples; h2v2 (3*this + neighbour + 8 or 7) >> 4 on column sums
                if (vs == 2) {
                    if (dw == 1) {
                        const int t = colsum(0);
                        line[0] = (uint8_t)((t * 4 + 8) >> 4); line[1] = (uint8_t)((t * 4 + 7) >> 4);
                    } else {
                        int thisc = colsum(0), nextc = colsum(1), lastc;
                        line[0] = (uint8_t)((thisc * 4 + 8) >> 4);
                        line[1] = (uint8_t)((thisc * 3 + nextc + 7) >> 4);
                        lastc = thisc; thisc = nextc;
                        for (int x = 1; x < dw - 1; ++x) {
                            nextc = colsum(x + 1);
                            line[2 * x] = (uint8_t)((thisc * 3 + lastc + 8) >> 4);
                            line[2 * x + 1] = (uint8_t)((thisc * 3 + nextc + 7) >> 4);
                            lastc = thisc; thisc = nextc;
                        }
                        line[2 * (dw - 1)] = (uint8_t)((thisc * 3 + lastc + 8) >> 4);
                        line[2 * (dw - 1) + 1] = (uint8_t)((thisc * 4 + 7) >> 4);
                    }
                } else {
                    if (dw == 1) { line[0] = line[1] = in0[0]; }
                    else {
                        line[0] = in0[0];
                        line[1] = (uint8_t)((in0[0] * 3 + in0[1] + 2) >> 2);
                        for (int x = 1; x < dw - 1; ++x) {
                            line[2 * x] = (uint8_t)((in0[x] * 3 + in0[x - 1] + 1) >> 2);
                            line[2 * x + 1] = (uint8_t)((in0[x] * 3 + in0[x + 1] + 2) >> 2);
                        }
                        line[2 * (dw - 1)] = (uint8_t)((in0[dw - 1] * 3 + in0[dw - 2] + 1) >> 2);
                        line[2 * (dw - 1) + 1] = in0[dw - 1];
                    }
                }
                memcpy(&out[(size_t)y * width], line.data(), (size_t)width);
            } else { // hs == 1, vs == 2: IJG has no fancy h1v2 in classic releases: replicate rows
                memcpy(&out[(size_t)y * width], in0, (size_t)width);
            }
        }
        return out;
    }

    ImageRGB8 run() {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) bad("not a JPEG stream");
        bool done = false;
        while (!done) {
            int b = u8();
            if (b != 0xFF) continue;
            int m = u8();
            while (m == 0xFF) m = u8();
            if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
            if (m == 0xD9) break;
            int len = u16() - 2;
            if (len < 0 || pos + (size_t)len > size) bad("bad segment length");
            const size_t next = pos + (size_t)len;
            switch (m) {
            case 0xDB: read_dqt(len); break;
            case 0xC4: read_dht(len); break;
            case 0xC0: case 0xC1: read_sof(len); break;
            case 0xC2: bad("progressive JPEG is not supported (baseline only)");
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                bad("unsupported JPEG process");
            case 0xDD: restart_interval = u16(); break;
            case 0xEE: // Adobe: colour transform flag
                if (len >= 12 && memcmp(data + pos, "Adobe", 5) == 0) adobe_transform = data[pos + 11];
                break;
            case 0xDA: read_scan(len); done = true; break;
            default: break;
            }
            if (m != 0xDA) pos = next;
        }
        if (!have_frame || !done) bad("no image data");

        ImageRGB8 img;
        img.width = width; img.height = height;
        auto px = std::make_shared<std::vector<uint8_t>>((size_t)width * (size_t)height * 3u);
        if (comps.size() == 1) {
            const std::vector<uint8_t> y = upsample(comps[0]);
            for (size_t i = 0; i < y.size(); ++i) { (*px)[3 * i] = (*px)[3 * i + 1] = (*px)[3 * i + 2] = y[i]; }
        } else {
            const std::vector<uint8_t> y = upsample(comps[0]), cb = upsample(comps[1]), cr = upsample(comps[2]);
            const bool rgb = adobe_transform == 0 || (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
            // IJG jdcolor.c: 16-bit fixed point
            constexpr int SCALEBITS = 16;
            constexpr int32_t ONE_HALF = 1 << (SCALEBITS - 1);
            auto FIX = [](double x) { return (int32_t)(x * (1 << SCALEBITS) + 0.5); };
            int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
            for (int i = 0; i < 256; ++i) {
                const int x = i - 128;
                cr_r[i] = (FIX(1.40200) * x + ONE_HALF) >> SCALEBITS;
                cb_b[i] = (FIX(1.77200) * x + ONE_HALF) >> SCALEBITS;
                cr_g[i] = -FIX(0.71414) * x;
                cb_g[i] = -FIX(0.34414) * x + ONE_HALF;
            }
            for (size_t i = 0; i < y.size(); ++i) {
                if (rgb) { (*px)[3 * i] = y[i]; (*px)[3 * i + 1] = cb[i]; (*px)[3 * i + 2] = cr[i]; continue; }
                const int Y = y[i], B = cb[i], R = cr[i];
                (*px)[3 * i] = clamp255(Y + cr_r[R]);
                (*px)[3 * i + 1] = clamp255(Y + ((cb_g[B] + cr_g[R]) >> SCALEBITS));
                (*px)[3 * i + 2] = clamp255(Y + cb_b[B]);
            }
        }
        img.pixels = px;
        return img;
    }
};

} // namespace

ImageRGB8 decode_baseline_jpeg(const uint8_t *data, size_t size) {
    if (!data) bad("null input");
    return Decoder(data, size).run();
}

} // namespace rt
