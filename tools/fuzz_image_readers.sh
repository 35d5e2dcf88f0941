#!/bin/bash
# CPU.  The host library's image readers (JPEG, PNG, PPM: rust-tracing_amd/host/image_io.cpp, jpeg_decoder.cpp) parse files a user hands
# them: this builds them with AddressSanitizer + UBSan into a small harness and feeds it mutated files (bytes overwritten, inserted, cut
# off; two-byte fields set to extremes; PNG chunk CRCs repaired so that the damage reaches the decoder).  Usage: tools/fuzz_image_readers.sh [seed] [per-file]
# Round 4: 23 000 mutants of 12 seed files; found and fixed an over-subscribed Huffman table indexing past the fast-lookup table and a
# signed overflow in the IDCT on absurd coefficients; clean since.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SEED="${1:-1}"; PER="${2:-400}"
W="$(mktemp -d)"
cat > "$W/harness.cpp" <<'CPP'
#include "image_io.hpp"
#include <cstdio>
#include <stdexcept>
int main(int argc, char **argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; ++i) {
        try { rt::ImageRGB8 im = rt::load_image_rgb8(argv[i]); ok += im.width > 0; }
        catch (const std::exception &) { ++bad; }
    }
    printf("decoded %d, refused %d\n", ok, bad);
    return 0;
}
CPP
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -I"$ROOT/rust-tracing_amd/host" -I"$ROOT/include" \
    "$W/harness.cpp" "$ROOT/rust-tracing_amd/host/image_io.cpp" "$ROOT/rust-tracing_amd/host/jpeg_decoder.cpp" -lz -o "$W/harness"
python3 - "$W" "$SEED" "$PER" <<'PY'
import os, random, struct, sys, zlib
import numpy as np
from PIL import Image
out, seed, per = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed); random.seed(seed)
img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8); img[5:20, 7:30] = (200, 30, 90)
seeds = []
def save(name, im, **kw):
    p = os.path.join(out, name); im.save(p, **kw); seeds.append(p)
save("a.jpg", Image.fromarray(img), quality=85); save("b.jpg", Image.fromarray(img), quality=85, progressive=True)
save("c.jpg", Image.fromarray(img), quality=60, progressive=True, subsampling=1); save("d.jpg", Image.fromarray(img[:, :, 0]), quality=70, progressive=True)
save("e.jpg", Image.fromarray(img), quality=85, subsampling=2)
save("a.png", Image.fromarray(img)); save("b.png", Image.fromarray(img).quantize(16)); save("c.png", Image.fromarray(img[:, :, 0]))
save("d.png", Image.fromarray(np.dstack([img, img[:, :, 0]])))
def chunk(k, b): return struct.pack(">I", len(b)) + k + b + struct.pack(">I", zlib.crc32(k + b))
def png(w, h, depth, colour, inter, raw, pal=None):
    d = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, inter))
    if pal: d += chunk(b"PLTE", pal)
    return d + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
def rows(n, width): return b"".join(b"\x04" + bytes(rng.integers(0, 256, width, dtype=np.uint8)) for _ in range(n))
extra = {"e.png": png(9, 9, 16, 2, 0, rows(9, 54)), "f.png": png(9, 9, 1, 3, 0, rows(9, 2), bytes(range(6)))}
a7 = b""
for x0, y0, dx, dy in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
    pw, ph = (9 - x0 + dx - 1) // dx if 9 > x0 else 0, (9 - y0 + dy - 1) // dy if 9 > y0 else 0
    if pw and ph: a7 += rows(ph, pw * 3)
extra["g.png"] = png(9, 9, 8, 2, 1, a7)
extra["h.ppm"] = b"P6\n5 4\n255\n" + bytes(rng.integers(0, 256, 60, dtype=np.uint8))
for name, data in extra.items():
    p = os.path.join(out, name); open(p, "wb").write(data); seeds.append(p)
n = 0
for s in seeds:
    data = bytearray(open(s, "rb").read())
    for _ in range(per):
        d = bytearray(data)
        for _ in range(random.randint(1, 3)):
            mode = random.random()
            if len(d) < 8: break
            if mode < 0.45: d[random.randrange(len(d))] = random.randrange(256)
            elif mode < 0.55: d = d[:random.randrange(1, len(d))]
            elif mode < 0.7:
                i = random.randrange(len(d)); d[i:i] = bytes(random.randrange(256) for _ in range(random.randint(1, 8)))
            elif mode < 0.85:
                i = random.randrange(len(d) - 4); d[i:i + 2] = struct.pack(">H", random.choice([0, 1, 2, 0xffff, 0x7fff, 64, 65, 63, 0x0101, 0x1111, 0x2222]))
            else:
                i = random.randrange(min(len(d) - 1, 200)); d[i] = random.choice([0, 1, 2, 3, 4, 8, 16, 0x11, 0x22, 0x41, 0xff, 0xc0, 0xc2, 0xc4, 0xda, 0xdd])
        if s.endswith(".png") and random.random() < 0.8:
            fixed, at = bytes(d[:8]), 8
            while at + 12 <= len(d):
                ln = struct.unpack(">I", d[at:at + 4])[0]
                if at + 12 + ln > len(d): break
                body = bytes(d[at + 4:at + 8 + ln]); fixed += bytes(d[at:at + 4]) + body + struct.pack(">I", zlib.crc32(body)); at += 12 + ln
            d = bytearray(fixed + bytes(d[at:]))
        open(os.path.join(out, f"m{n}{os.path.splitext(s)[1]}"), "wb").write(d); n += 1
print(n, "mutants of", len(seeds), "files")
PY
ASAN_OPTIONS=detect_leaks=0 "$W/harness" "$W"/*.jpg "$W"/*.png "$W"/*.ppm
rm -rf "$W"
