#!/bin/bash
# CPU.  Builds tools/sanitize/fuzz_scene_compiler.cpp with AddressSanitizer + UBSan and runs it.  Usage: tools/fuzz_scene_compiler.sh [seed] [per-scene]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
H="$ROOT/rust-tracing_amd/host"
W="$(mktemp -d)"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=all -I"$ROOT/include" -I"$ROOT/rust-tracing_amd/csrc" -I"$H" \
    "$ROOT/tools/sanitize/fuzz_scene_compiler.cpp" "$H/scenes.cpp" "$H/image_io.cpp" "$H/jpeg_decoder.cpp" "$H/capi.cpp" -lz -o "$W/harness"
ASAN_OPTIONS=detect_leaks=0 "$W/harness" "${1:-1}" "${2:-200}" 2>/dev/null
rm -rf "$W"
