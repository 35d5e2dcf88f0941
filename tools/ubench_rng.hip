// Micro-benchmark: SIMD-cycles per wave-instruction of integer / f64 instructions an RNG could be built from, and the
// whole-draw cost of candidate generators (4 waves per SIMD).  hipcc --offload-arch=gfx950 -O3 -o ubench_rng ubench_rng.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}
template <int OP> __global__ void k(uint64_t *out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7;
    double d0 = a0 * 1.0001 + 1.0, d1 = a1 * 0.5 + 1.0, d2 = a2 * 0.25 + 1.0, d3 = a3 * 0.125 + 1.0;
    uint64_t q0 = a0 | ((uint64_t)a1 << 32), q1 = a2 | ((uint64_t)a3 << 32);
    double acc = 0.0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { a0 = a0 * a1 + 1; a1 = a1 * a2 + 1; a2 = a2 * a3 + 1; a3 = a3 * a0 + 1; }   // 4 x (mul_lo + add)
            if (OP == 1) { a0 = __umulhi(a0, a1) | 1; a1 = __umulhi(a1, a2) | 1; a2 = __umulhi(a2, a3) | 3; a3 = __umulhi(a3, a0) | 5; }
            if (OP == 2) { a0 = a0 + a1; a1 = a1 ^ a2; a2 = a2 + a3; a3 = a3 ^ a0; }
            if (OP == 3) { a0 = mul24(a0, a1) + 1; a1 = mul24(a1, a2) + 1; a2 = mul24(a2, a3) + 1; a3 = mul24(a3, a0) + 1; } // mad_u32_u24
            if (OP == 4) { q0 = (uint64_t)a0 * a1 + q0; a0 = (uint32_t)(q0 >> 32); q1 = (uint64_t)a2 * a3 + q1; a2 = (uint32_t)(q1 >> 32);
                           q0 = (uint64_t)a1 * a2 + q0; a1 = (uint32_t)(q0 >> 13); q1 = (uint64_t)a3 * a0 + q1; a3 = (uint32_t)(q1 >> 7); } // 4 x mad_u64_u32
            if (OP == 5) { q0 = (q0 << (a0 & 31)) + 1; q1 = (q1 >> (a1 & 31)) + q0; q0 = (q0 << 3) ^ q1; q1 = (q1 >> 5) ^ q0; }
            if (OP == 6) { a0 = __builtin_amdgcn_alignbit(a0, a1, 7) ; a1 = __builtin_amdgcn_alignbit(a1, a2, 9); a2 = __builtin_amdgcn_alignbit(a2, a3, 11); a3 = __builtin_amdgcn_alignbit(a3, a0, 13); }
            if (OP == 7) { d0 = __builtin_fma(d0, d1, d2); d1 = __builtin_fma(d1, d2, d3); d2 = __builtin_fma(d2, d3, d0); d3 = __builtin_fma(d3, d0, d1); }
            if (OP == 8) { d0 = d0 * d1; d1 = d1 + d2; d2 = d2 * d3; d3 = d3 + d0; }
            if (OP == 9) { d0 = (double)a0; a1 += (uint32_t)__double2hiint(d0); d1 = (double)a1; a2 += (uint32_t)__double2loint(d1); d2 = (double)a2; a3 += (uint32_t)__double2hiint(d2); d3 = (double)a3; a0 += (uint32_t)__double2loint(d3); } // cvt_f64_u32
            // ---- whole draws (one random() each, accumulated) ----
            if (OP == 20) { q0 += 0x9E3779B97F4A7C15ull; acc += (double)(mix64(q0) >> 11) * 0x1p-53; }
            if (OP == 21) { // paired 32-bit hashes with a cross step (4 mul_lo)
                q0 += 0x9E3779B97F4A7C15ull; uint32_t x = (uint32_t)q0, y = (uint32_t)(q0 >> 32);
                x ^= x >> 16; x *= 0x7feb352du; y ^= y >> 15; y *= 0x846ca68bu; x ^= y >> 13; y ^= x >> 16; x *= 0x9e3779b1u; y *= 0x85ebca6bu; x ^= y >> 15;
                acc += __hiloint2double((int)((x >> 12) | 0x3ff00000u), (int)y) - 1.0; }
            if (OP == 22) { // 24-bit multiplies only
                q0 += 0x9E3779B97F4A7C15ull; uint32_t x = (uint32_t)q0, y = (uint32_t)(q0 >> 32);
                x = mul24(x, 0xB5297Au | 1) ^ (y >> 9); y = mul24(y, 0x68E31Du | 1) ^ (x >> 11); x = mul24(x ^ (x >> 13), 0x1B873Du) + y; y = mul24(y ^ (y >> 12), 0x93D765u) ^ (x >> 7);
                x ^= y >> 15;
                acc += __hiloint2double((int)((x >> 12) | 0x3ff00000u), (int)y) - 1.0; }
            if (OP == 23) { // f64 product error term
                d0 = d0 + 0.6180339887498949; if (d0 >= 2.0) d0 -= 1.0; const double p = d0 * d1; const double lo = __builtin_fma(d0, d1, -p);
                const uint64_t b = (uint64_t)__double_as_longlong(lo); acc += __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull)) - 1.0; d1 = d1 * 1.0000001; }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t sink = a0 + a1 + a2 + a3 + (uint64_t)d0 + (uint64_t)d1 + (uint64_t)d2 + (uint64_t)d3 + q0 + q1 + (uint64_t)acc;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = sink; }
}

template <int OP> void run(const char *name, int per_iter) {
    uint64_t *d;
    hipMalloc(&d, 4096 * 16);
    const int iters = 2000, threads = 1024;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, 1u, iters);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, 2u, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(512);
    hipMemcpy(h.data(), d, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < 256; ++b) cyc += (double)h[b * 2];
    cyc /= 256;
    // s_memtime ticks at 100 MHz; shader clock ~2.4 GHz: report both raw ticks and relative to add/xor
    printf("%-34s %9.3f memtime-ticks per (wave x unit) at 4 waves/SIMD  [units/iter %d]\n", name, cyc / (iters * 16.0 * per_iter) / 4.0, per_iter);
    hipFree(d);
}

int main() {
    run<2>("add/xor u32", 4);
    run<0>("mul_lo_u32 + add", 4);
    run<1>("mul_hi_u32 + or", 4);
    run<3>("mad_u32_u24", 4);
    run<4>("mad_u64_u32 (+shift)", 4);
    run<5>("64-bit shift (+op)", 4);
    run<6>("alignbit", 4);
    run<7>("fma f64", 4);
    run<8>("mul/add f64", 4);
    run<9>("cvt_f64_u32 + hi/lo + add", 4);
    run<20>("draw: splitmix64 (current)", 1);
    run<21>("draw: paired 32-bit, 4 mul_lo", 1);
    run<22>("draw: 24-bit multiplies", 1);
    run<23>("draw: f64 product error term", 1);
    return 0;
}
