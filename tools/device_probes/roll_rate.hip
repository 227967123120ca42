// roll_rate.hip -- the rolling step of the fused consumers (run_kernel.hpp: shift_encoding / shift_first_encoding of the complement,
// canonical select, XOR fold) alone in registers: SIMD cycles per step at 1-8 wavefronts per SIMD, in the forms the kernel could use.
//   hipcc --offload-arch=gfx950 -O3 -o tools/roll_rate tools/roll_rate.hip && tools/roll_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Stamp { uint64_t cyc, real, r0, r1; };
constexpr int STEPS = 32, ITERS = 512;

// V = 0: the kernel's form (64-bit shifts, 64-bit compare, two selects)   1: forward kmers only (no compare / select)
// V = 2: 32-bit halves with v_alignbit, compare + select                      3: 32-bit halves, select-free canonical fold
//                                                                                (min(a, b) = b ^ ((a ^ b) & -(a < b)), borrow of a 64-bit subtraction)
template <int V, int SKEW = 0>
__global__ __launch_bounds__(256) void roll(uint64_t *out, Stamp *st, uint64_t seed, uint32_t k) {
    const uint32_t t = threadIdx.x + blockIdx.x * 256u;
    const uint64_t mask = (1ull << (2 * k)) - 1ull;
    const uint32_t top = 2u * (k - 1u);
    uint64_t fw = (seed * (t + 1)) & mask, rc = (~fw) & mask, S = seed ^ (t * 0x9E3779B97F4A7C15ull), acc = 0;
    uint32_t flo = (uint32_t)fw, fhi = (uint32_t)(fw >> 32), rlo = (uint32_t)rc, rhi = (uint32_t)(rc >> 32), alo = 0, ahi = 0;
    const uint32_t mhi = (uint32_t)(mask >> 32);
    if (SKEW) {  // desynchronise: workgroup b of a CU waits b x 37 sleeps, every wavefront a few more
        const uint32_t n = (blockIdx.x / 256u) * 37u + (threadIdx.x >> 6) * 5u;
        for (uint32_t i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(3);
    }
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        S = S * 0x9E3779B97F4A7C15ull + it;  // (new symbols for the next 32 steps: two instructions per 32 steps)
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const uint32_t sym = (uint32_t)(S >> (2 * j)) & 3u;
            if constexpr (V == 0 || V == 1) {
                fw = ((fw << 2) | sym) & mask;
                if constexpr (V == 0) {
                    rc = (rc >> 2) | ((uint64_t)(sym ^ 3u) << top);
                    acc ^= fw < rc ? fw : rc;
                } else {
                    acc ^= fw;
                }
            } else {
                const uint32_t nhi = __builtin_amdgcn_alignbit(fhi, flo, 30) & mhi;
                flo = (flo << 2) | sym;
                fhi = nhi;
                rlo = __builtin_amdgcn_alignbit(rhi, rlo, 2);
                rhi = (rhi >> 2) | ((sym ^ 3u) << (top - 32u));
                if constexpr (V == 2) {
                    const bool lt = (((uint64_t)fhi << 32) | flo) < (((uint64_t)rhi << 32) | rlo);
                    alo ^= lt ? flo : rlo;
                    ahi ^= lt ? fhi : rhi;
                } else {
                    const uint64_t d = (((uint64_t)fhi << 32) | flo) - (((uint64_t)rhi << 32) | rlo);
                    const uint32_t m = (uint32_t)((int32_t)(uint32_t)(d >> 32) >> 31);  // all ones iff fw < rc (both below 2^62)
                    alo ^= rlo ^ ((flo ^ rlo) & m);
                    ahi ^= rhi ^ ((fhi ^ rhi) & m);
                }
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if constexpr (V >= 2) acc = (((uint64_t)ahi << 32) | alo) ^ flo ^ rhi;
    out[t] = acc ^ fw ^ rc;
    if ((threadIdx.x & 63u) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0, r0, r1};
}

template <int V, int SKEW = 0>
int run(const char *name, uint64_t *out, Stamp *st, std::vector<Stamp> &h) {
    printf("%-70s", name);
    for (int w : {1, 2, 4, 8}) {
        const int grid = 256 * w;
        hipLaunchKernelGGL((roll<V, SKEW>), dim3(grid), dim3(256), 0, 0, out, st, 0x1234567ull, 31u);
        hipLaunchKernelGGL((roll<V, SKEW>), dim3(grid), dim3(256), 0, 0, out, st, 0x1234567ull, 31u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * grid * 4, hipMemcpyDeviceToHost));
        // instructions / SPAN of the launch (first start to last end), not / mean lifetime: the arbiter prefers the oldest
        // wavefront of a SIMD, the wavefronts finish one after the other (tools/valu_rates.hip)
        double cyc = 0, real = 0;
        uint64_t first = ~0ull, last = 0;
        for (int i = 0; i < grid * 4; ++i) {
            cyc += (double)h[i].cyc;
            real += (double)h[i].real;
            first = h[i].r0 < first ? h[i].r0 : first;
            last = h[i].r1 > last ? h[i].r1 : last;
        }
        const double span_cycles = (double)(last - first) * cyc / real;
        printf("  w%d %6.1f cyc/step", w, span_cycles / ((double)ITERS * STEPS) / w);
    }
    printf("\n");
    return 0;
}

int main() {
    uint64_t *out; Stamp *st;
    CHECK(hipMalloc(&out, 8ull * 256 * 2048)); CHECK(hipMalloc(&st, sizeof(Stamp) * 2048 * 4));
    std::vector<Stamp> h(2048 * 4);
    printf("SIMD cycles per rolling step (one kmer per lane), K = 31, at w wavefronts per SIMD\n");
    if (run<0>("canonical, 64-bit shifts + v_cmp_lt_u64 + two selects (run_kernel.hpp)", out, st, h)) return 1;
    if (run<0, 1>("the same, wavefronts started at different times (different places of the code)", out, st, h)) return 1;
    if (run<1>("forward only, 64-bit shift", out, st, h)) return 1;
    if (run<2>("canonical, 32-bit halves (v_alignbit) + compare + selects", out, st, h)) return 1;
    if (run<3>("canonical, 32-bit halves, select-free fold (borrow mask)", out, st, h)) return 1;
    return 0;
}
