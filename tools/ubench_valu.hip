// Micro-benchmark: issue cost of the VALU instructions the render kernel leans on (one wave per SIMD and four waves
// per SIMD), in shader cycles per wave-instruction.   hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP> __global__ void k(uint64_t *out, uint32_t seed, int iters) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7;
    double d0 = a0 * 1.0001, d1 = a1 * 0.5, d2 = a2 * 0.25, d3 = a3 * 0.125;
    float f0 = a0 * 1.0f, f1 = a1 * 0.5f, f2 = a2, f3 = a3;
    uint64_t q0 = a0 | ((uint64_t)a1 << 32), q1 = a2 | ((uint64_t)a3 << 32);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { a0 = a0 * a1 + 1; a1 = a1 * a2 + 1; a2 = a2 * a3 + 1; a3 = a3 * a0 + 1; }              // v_mul_lo_u32 (+add)
            if (OP == 1) { a0 = __umulhi(a0, a1); a1 = __umulhi(a1, a2) | 1; a2 = __umulhi(a2, a3) | 3; a3 = __umulhi(a3, a0) | 5; }
            if (OP == 2) { a0 = a0 + a1; a1 = a1 ^ a2; a2 = a2 + a3; a3 = a3 ^ a0; }                               // 32-bit add/xor
            if (OP == 3) { d0 = d0 * d1 + d2; d1 = d1 * d2 + d3; d2 = d2 * d3 + d0; d3 = d3 * d0 + d1; }            // f64 fma (contract on here)
            if (OP == 4) { f0 = f0 * f1 + f2; f1 = f1 * f2 + f3; f2 = f2 * f3 + f0; f3 = f3 * f0 + f1; }            // f32 fma
            if (OP == 5) { q0 = q0 * 0xBF58476D1CE4E5B9ull + 1; q1 = q1 * 0x94D049BB133111EBull + 1; q0 ^= q0 >> 30; q1 ^= q1 >> 27; } // 2 x (64-bit mul + xorshift)
            if (OP == 6) { d0 = d0 + d1; d1 = d1 * d2; d2 = d2 + d3; d3 = d3 * d0; }                                // f64 add / mul
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t sink = a0 + a1 + a2 + a3 + (uint64_t)d0 + (uint64_t)d1 + (uint64_t)d2 + (uint64_t)d3 + (uint64_t)f0 + (uint64_t)f1 + (uint64_t)f2 + (uint64_t)f3 + q0 + q1;
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = sink; }
}

template <int OP> void run(const char *name, int per_iter) {
    uint64_t *d;
    hipMalloc(&d, 4096 * 16);
    const int iters = 2000;
    for (int threads : {64, 256, 1024}) { // 1 wave per CU, 1 per SIMD, 4 per SIMD
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, 1u, iters);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, d, 2u, iters);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(512);
        hipMemcpy(h.data(), d, 512 * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (int b = 0; b < 256; ++b) cyc += (double)h[b * 2];
        cyc /= 256;
        const double waves_per_simd = threads >= 256 ? threads / 256.0 : 0.25;
        printf("%-28s threads/block %4d : %7.2f cycles per wave-instruction (wall), %6.2f SIMD-cycles per instruction\n", name,
               threads, cyc / (iters * 16.0 * per_iter), cyc / (iters * 16.0 * per_iter) / (waves_per_simd < 1 ? 1 : waves_per_simd));
    }
    hipFree(d);
}

int main() {
    run<2>("add/xor u32", 4);
    run<4>("fma f32", 4);
    run<3>("fma f64", 4);
    run<6>("add/mul f64", 4);
    run<0>("mul_lo u32 (+add)", 4);
    run<1>("mul_hi u32 (+or)", 4);
    run<5>("2x(64-bit mul+add+xorshift)", 2);
    return 0;
}
