// JPEG decoder (ITU T.81, Huffman, 8-bit: baseline / extended sequential DCT in one or several scans, and progressive DCT —
// spectral selection and successive approximation, Annex G) for ImageTexture ingest.
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

[[noreturn]] void bad(const std::string &m) { throw std::runtime_error("decode_jpeg: " + m); }

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
            if (code + counts[len - 1] > (1 << len)) bad("over-subscribed Huffman table"); // (more codes of this length than there are)
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
    std::vector<int16_t> coefs;     // blocks_w x blocks_h blocks of 64 coefficients (natural order), as the scans leave them: quantised
    uint16_t qt[64];                // the component's quantisation table, latched at its first scan (a file may redefine a slot later)
    bool qt_latched = false;
    std::vector<uint8_t> plane;     // blocks_w*8 x blocks_h*8 samples
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int bits = 0;
    bool hit_marker = false;
    int made_up = 0; // bits at the low end of the accumulator that no byte of the file stands behind (fed after a marker or the end)

    void fill() {
        while (bits <= 24) {
            int byte = 0;
            bool real = false;
            if (!hit_marker && p < end) {
                byte = *p++;
                real = true;
                if (byte == 0xFF) {
                    if (p < end && *p == 0x00) { ++p; }             // stuffed zero
                    else { hit_marker = true; --p; byte = 0; real = false; } // a marker: feed zeros from here on
                }
            }
            if (!real) made_up += 8;
            acc |= (uint32_t)byte << (24 - bits);
            bits += 8;
        }
    }
    bool overran() const { return bits < made_up; } // the decoder has consumed bits the file does not hold
    int peek(int n) { if (bits < n) fill(); return (int)(acc >> (32 - n)); }
    void skip(int n) { acc <<= n; bits -= n; }
    int get(int n) { if (n == 0) return 0; int v = peek(n); skip(n); return v; }
    void reset() { acc = 0; bits = 0; hit_marker = false; made_up = 0; }
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
// (64-bit intermediates: a valid file never leaves the 32-bit range IJG computes in — the results are IJG's —, a damaged one, whose
// coefficients can be anything, must not overflow a signed integer either)
using wide_t = int64_t;
inline wide_t descale(wide_t x, int n) { return (x + ((wide_t)1 << (n - 1))) >> n; }
inline uint8_t clamp255(wide_t x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void idct_islow(const int32_t coef[64], uint8_t *out, int stride) {
    wide_t ws[64];
    for (int c = 0; c < 8; ++c) { // pass 1: columns
        const int32_t *in = coef + c;
        wide_t z2 = in[16], z3 = in[48];
        wide_t z1 = (z2 + z3) * FIX_0_541196100;
        wide_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        wide_t tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = in[0]; z3 = in[32];
        wide_t tmp0 = (z2 + z3) * ((wide_t)1 << CONST_BITS), tmp1 = (z2 - z3) * ((wide_t)1 << CONST_BITS);
        const wide_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56]; tmp1 = in[40]; tmp2 = in[24]; tmp3 = in[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        wide_t z4 = tmp1 + tmp3;
        const wide_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        wide_t *w = ws + c;
        w[0] = descale(tmp10 + tmp3, CONST_BITS - PASS1_BITS);  w[56] = descale(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        w[8] = descale(tmp11 + tmp2, CONST_BITS - PASS1_BITS);  w[48] = descale(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        w[16] = descale(tmp12 + tmp1, CONST_BITS - PASS1_BITS); w[40] = descale(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        w[24] = descale(tmp13 + tmp0, CONST_BITS - PASS1_BITS); w[32] = descale(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int r = 0; r < 8; ++r) { // pass 2: rows
        const wide_t *w = ws + r * 8;
        wide_t z2 = w[2], z3 = w[6];
        wide_t z1 = (z2 + z3) * FIX_0_541196100;
        wide_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        wide_t tmp3 = z1 + z2 * FIX_0_765366865;
        wide_t tmp0 = (w[0] + w[4]) * ((wide_t)1 << CONST_BITS), tmp1 = (w[0] - w[4]) * ((wide_t)1 << CONST_BITS);
        const wide_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        wide_t z4 = tmp1 + tmp3;
        const wide_t z5 = (z3 + z4) * FIX_1_175875602;
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
    bool have_frame = false, progressive = false;
    int adobe_transform = -1;
    uint32_t eobrun = 0; // progressive AC scans: blocks still covered by the last end-of-band run

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
        if ((uint64_t)width * (uint64_t)height > MAX_IMAGE_PIXELS) bad("image too large"); // (before anything is allocated for it)
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
            c.coefs.assign((size_t)c.blocks_w * (size_t)c.blocks_h * 64, 0);
        }
        have_frame = true;
    }

    int16_t *block_of(Component &c, int bx, int by) { return c.coefs.data() + ((size_t)by * c.blocks_w + (size_t)bx) * 64; }

    // One block of a sequential scan (T.81 F.2.2): DC difference, then (run, size) pairs up to the end of block
    void block_sequential(BitReader &br, Component &c, int16_t *blk) {
        const HuffTable &hd = dc[c.td], &ha = ac[c.ta];
        if (!hd.present || !ha.present) bad("missing table");
        int t = decode_symbol(br, hd);
        if (t > 11) bad("bad DC size");
        c.pred = (int16_t)(c.pred + (t ? extend(br.get(t), t) : 0)); // (a DC value has 16 bits at most: a damaged stream wraps, it does not overflow)
        blk[0] = (int16_t)c.pred;
        for (int k = 1; k < 64;) {
            int rs = decode_symbol(br, ha);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) { k += 16; continue; }
                break; // EOB
            }
            k += r;
            if (k > 63) bad("AC run overflows the block");
            blk[ZIGZAG[k]] = (int16_t)extend(br.get(s), s);
            ++k;
        }
    }
    // Progressive scans (T.81 G.1.2).  DC: the first scan carries the difference of the value's high bits, a later one ONE more bit.
    void block_dc_first(BitReader &br, Component &c, int16_t *blk, int al) {
        const HuffTable &hd = dc[c.td];
        if (!hd.present) bad("missing table");
        int t = decode_symbol(br, hd);
        if (t > 11) bad("bad DC size");
        c.pred = (int16_t)(c.pred + (t ? extend(br.get(t), t) : 0)); // (a DC value has 16 bits at most: a damaged stream wraps, it does not overflow)
        blk[0] = (int16_t)(c.pred * (1 << al));
    }
    void block_dc_refine(BitReader &br, int16_t *blk, int al) {
        if (br.get(1)) blk[0] = (int16_t)(blk[0] | (1 << al));
    }
    // AC, first pass over a band [ss, se]: as sequential, but an end-of-band code may cover a RUN of blocks (G.1.2.2)
    void block_ac_first(BitReader &br, Component &c, int16_t *blk, int ss, int se, int al) {
        if (eobrun > 0) { --eobrun; return; }
        const HuffTable &ha = ac[c.ta];
        if (!ha.present) bad("missing table");
        for (int k = ss; k <= se;) {
            int rs = decode_symbol(br, ha);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) { eobrun = (1u << r) - 1u; if (r) eobrun += (uint32_t)br.get(r); break; }
                k += 16;
                continue;
            }
            k += r;
            if (k > se) bad("AC run overflows the band");
            blk[ZIGZAG[k]] = (int16_t)(extend(br.get(s), s) * (1 << al));
            ++k;
        }
    }
    // AC, refinement (G.1.2.3): every coefficient that is already non-zero gets one correction bit as the scan passes it; a
    // (run, 1) code places a NEW +-1 << al after `run` coefficients that are still zero
    void block_ac_refine(BitReader &br, Component &c, int16_t *blk, int ss, int se, int al) {
        const HuffTable &ha = ac[c.ta];
        if (!ha.present) bad("missing table");
        const int p1 = 1 << al, m1 = -(1 << al);
        auto correct = [&](int16_t &v) {
            if (br.get(1) && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
        };
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                int rs = decode_symbol(br, ha);
                int r = rs >> 4, s = rs & 15, value = 0;
                if (s) {
                    if (s != 1) bad("bad refinement code");
                    value = br.get(1) ? p1 : m1;
                } else if (r != 15) {
                    eobrun = 1u << r;
                    if (r) eobrun += (uint32_t)br.get(r);
                    break;
                }
                for (; k <= se; ++k) {
                    int16_t &v = blk[ZIGZAG[k]];
                    if (v != 0) correct(v);
                    else if (--r < 0) break;
                }
                if (value) {
                    if (k > se) bad("AC run overflows the band");
                    blk[ZIGZAG[k]] = (int16_t)value;
                }
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) {
                int16_t &v = blk[ZIGZAG[k]];
                if (v != 0) correct(v);
            }
            --eobrun;
        }
    }

    void read_scan(int len) {
        (void)len;
        if (!have_frame) bad("SOS before SOF");
        int ns = u8();
        if (ns < 1 || ns > (int)comps.size()) bad("bad component count in a scan");
        std::vector<Component *> sc;
        for (int i = 0; i < ns; ++i) {
            int id = u8(), tdta = u8();
            Component *c = nullptr;
            for (auto &x : comps) if (x.id == id) c = &x;
            if (!c) bad("scan names an unknown component");
            for (Component *o : sc) if (o == c) bad("scan names a component twice");
            c->td = tdta >> 4; c->ta = tdta & 15;
            if (c->td > 3 || c->ta > 3) bad("bad table selector");
            if (!c->qt_latched) {
                if (!quant_present[c->tq]) bad("missing table");
                memcpy(c->qt, quant[c->tq], sizeof c->qt);
                c->qt_latched = true;
            }
            sc.push_back(c);
        }
        const int ss = u8(), se = u8(), ahal = u8();
        const int ah = ahal >> 4, al = ahal & 15;
        if (!progressive) {
            if (ss != 0 || se != 63 || ahal != 0) bad("bad scan parameters for a sequential file");
        } else {
            if (ss > se || se > 63 || al > 13 || (ah != 0 && ah != al + 1)) bad("bad progressive scan parameters");
            if (ss == 0 ? se != 0 : ns != 1) bad("bad progressive scan parameters"); // DC scans carry DC only; AC scans one component
        }
        BitReader br{data + pos, data + size};
        eobrun = 0;
        auto one_block = [&](Component &c, int bx, int by) {
            int16_t *blk = block_of(c, bx, by);
            if (!progressive) block_sequential(br, c, blk);
            else if (ss == 0) { if (ah == 0) block_dc_first(br, c, blk, al); else block_dc_refine(br, blk, al); }
            else if (ah == 0) block_ac_first(br, c, blk, ss, se, al);
            else block_ac_refine(br, c, blk, ss, se, al);
        };
        // a scan of ONE component walks that component's own blocks, row by row, without the padding to whole MCUs (T.81 A.2.3)
        const bool single = ns == 1;
        const int mcus_x = single ? (sc[0]->width + 7) / 8 : (width + 8 * max_h - 1) / (8 * max_h);
        const int mcus_y = single ? (sc[0]->height + 7) / 8 : (height + 8 * max_v - 1) / (8 * max_v);
        int until_restart = restart_interval;
        for (Component *c : sc) c->pred = 0;
        for (int my = 0; my < mcus_y; ++my)
            for (int mx = 0; mx < mcus_x; ++mx) {
                if (restart_interval && until_restart == 0) {
                    // byte-align, expect RSTn
                    const uint8_t *q = br.p;
                    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                    if (q + 1 >= br.end) bad("missing restart marker");
                    if (br.overran()) bad("truncated entropy-coded segment");
                    br.p = q + 2;
                    br.reset();
                    for (Component *c : sc) c->pred = 0;
                    eobrun = 0;
                    until_restart = restart_interval;
                }
                if (single) one_block(*sc[0], mx, my);
                else
                    for (Component *c : sc)
                        for (int v = 0; v < c->v; ++v)
                            for (int h = 0; h < c->h; ++h) one_block(*c, mx * c->h + h, my * c->v + v);
                if (restart_interval) --until_restart;
            }
        if (br.overran()) bad("truncated entropy-coded segment");
        pos = (size_t)(br.p - data);
    }

    // after the last scan: dequantise and transform every block
    void reconstruct() {
        for (auto &c : comps) {
            if (!c.qt_latched) bad("a component has no scan");
            for (int by = 0; by < c.blocks_h; ++by)
                for (int bx = 0; bx < c.blocks_w; ++bx) {
                    const int16_t *blk = block_of(c, bx, by);
                    int32_t coef[64];
                    for (int i = 0; i < 64; ++i) coef[i] = (int32_t)blk[i] * c.qt[i];
                    idct_islow(coef, c.plane.data() + ((size_t)by * 8 * c.blocks_w + (size_t)bx) * 8, c.blocks_w * 8);
                }
        }
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
        bool done = false; // (a scan has been read)
        for (;;) {
            if (pos >= size) { if (done) break; bad("truncated"); } // (files without an EOI marker exist: what was decoded stands)
            int b = u8();
            if (b != 0xFF) continue;
            if (pos >= size) break;
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
            case 0xC0: case 0xC1: case 0xC2:
                if (have_frame) bad("more than one frame");
                progressive = m == 0xC2;
                read_sof(len);
                break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                bad("unsupported JPEG process");
            case 0xDD: restart_interval = u16(); break;
            case 0xEE: // Adobe: colour transform flag
                if (len >= 12 && memcmp(data + pos, "Adobe", 5) == 0) adobe_transform = data[pos + 11];
                break;
            case 0xDA: read_scan(len); done = true; break; // (pos now stands behind the scan's entropy-coded data)
            default: break;
            }
            if (m != 0xDA) pos = next;
        }
        if (!have_frame || !done) bad("no image data");
        reconstruct();

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

ImageRGB8 decode_jpeg(const uint8_t *data, size_t size) {
    if (!data) bad("null input");
    return Decoder(data, size).run();
}

} // namespace rt
