// Does CDNA4 skip the 16-lane passes of a wave64 VALU instruction whose lanes are all inactive?  Times a dependent f64 fma
// chain with 64 / 32 / 16 active lanes, the active lanes either packed into the low lanes or spread over the whole wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(uint64_t *out, uint64_t active_mask, int iters) {
    const uint32_t lane = threadIdx.x & 63u;
    double a = 1.0 + lane * 1e-3, b = 0.999999, c = 1e-9, d = 0.5 + lane * 1e-4;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if ((active_mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a + d == 12345.678) out[0] = 0;
}
int main() {
    uint64_t *d;
    hipMalloc(&d, 256 * 16 * 8);
    struct { const char *name; uint64_t mask; } cases[] = {
        {"64 lanes", ~0ull}, {"32 lanes, low half", 0xffffffffull}, {"32 lanes, every other", 0x5555555555555555ull},
        {"16 lanes, low quarter", 0xffffull}, {"16 lanes, every fourth", 0x1111111111111111ull}, {"1 lane", 1ull}};
    for (auto &c : cases) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, c.mask, 2000);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(256 * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        printf("%-24s %10.1f memtime ticks per wave\n", c.name, s / h.size());
    }
    return 0;
}
