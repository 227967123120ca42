// store_pacing.hip -- persistent workgroups write two arrays at 5.0-5.6 TB/s on MI355X, one-pass workgroups at 6.4-6.9
// (tools/store_regimes.hip).  Does pacing, desynchronising or splitting the streams of a persistent wavefront close the gap?
// Measurement tooling only.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/store_pacing tools/store_pacing.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void one_pass(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v, int per_tile) {
    size_t base = (size_t)blockIdx.x * per_tile;
    for (int r = threadIdx.x; r < per_tile && base + r < n; r += 256) {
        size_t i = base + r;
        a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i);
    }
}
// MODE 0 plain grid-stride; 1 s_sleep(SL) after every store pair; 2 random start phase; 3 even wavefronts write a, odd write b
// (two elements each); 4 four pairs then s_waitcnt vmcnt(0); 5 sleep only every 4th pair
template <int MODE, int SL>
__global__ __launch_bounds__(256) void persistent(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    const size_t stride = (size_t)gridDim.x * 256;
    if constexpr (MODE == 2) {
        unsigned h = (blockIdx.x * 2654435761u) >> 26;  // 0..63
        for (unsigned i = 0; i < h; ++i) __builtin_amdgcn_s_sleep(8);
    }
    if constexpr (MODE == 3) {
        const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        ulonglong2* dst = (wave & 1u) ? b : a;
        // the workgroup's 256 elements of each array per step: wavefront pair (0,1) takes the first 128, (2,3) the second
        for (size_t i0 = (size_t)blockIdx.x * 256; i0 < n; i0 += stride) {
            size_t i = i0 + (wave >> 1) * 128 + lane;
            if (i < n) dst[i] = make_ulonglong2(v + i, v ^ i);
            if (i + 64 < n) dst[i + 64] = make_ulonglong2(v * i, v - i);
        }
        return;
    }
    int cnt = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i);
        if constexpr (MODE == 1) __builtin_amdgcn_s_sleep(SL);
        if constexpr (MODE == 4) { if ((++cnt & 3) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if constexpr (MODE == 5) { if ((++cnt & 3) == 0) __builtin_amdgcn_s_sleep(SL); }
    }
}
// a persistent workgroup that does ALU work between its stores (the real kernels do): W dependent multiply-adds per pair
template <int W>
__global__ __launch_bounds__(256) void persistent_work(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned long long x = v + i;
#pragma unroll
        for (int w = 0; w < W; ++w) x = x * 0x9E3779B97F4A7C15ull + w;
        a[i] = make_ulonglong2(x, v ^ i); b[i] = make_ulonglong2(v * i, x - i);
    }
}
// the same with the stores made conditional on something that never happens: the arithmetic alone
template <int W>
__global__ __launch_bounds__(256) void persistent_work_only(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned long long x = v + i;
#pragma unroll 16
        for (int w = 0; w < W; ++w) x = x * 0x9E3779B97F4A7C15ull + w;
        if (x == 0x1234567ull) { a[i] = make_ulonglong2(x, v ^ i); b[i] = make_ulonglong2(v * i, x - i); }
    }
}
template <int W>
__global__ __launch_bounds__(256) void persistent_work_heavy(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        unsigned long long x = v + i;
#pragma unroll 16
        for (int w = 0; w < W; ++w) x = x * 0x9E3779B97F4A7C15ull + w;
        a[i] = make_ulonglong2(x, v ^ i); b[i] = make_ulonglong2(v * i, x - i);
    }
}
// The same arithmetic and the same stores with the two decoupled: wavefronts 0-2 of a workgroup compute into an LDS ring (two
// slots each), wavefront 3 does nothing but move full slots to HBM.  A full store queue then stalls the mover only.
template <int W>
__global__ __launch_bounds__(256) void persistent_split(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    __shared__ ulonglong2 ring_a[3][2][64], ring_b[3][2][64];
    __shared__ unsigned long long ring_i[3][2];   // first element index of the slot, ~0 = stop
    __shared__ volatile unsigned int full[3][2];
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (threadIdx.x < 6) full[threadIdx.x / 2][threadIdx.x % 2] = 0u;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * 192;
    if (wave < 3) {
        unsigned slot = 0;
        for (size_t i0 = (size_t)blockIdx.x * 192 + wave * 64; ; i0 += stride) {
            const bool last = i0 >= n;
            const size_t i = i0 + lane;
            unsigned long long x = v + i;
            if (!last) {
#pragma unroll 16
                for (int w = 0; w < W; ++w) x = x * 0x9E3779B97F4A7C15ull + w;
            }
            while (full[wave][slot]) __builtin_amdgcn_s_sleep(1);
            ring_a[wave][slot][lane] = make_ulonglong2(x, v ^ i);
            ring_b[wave][slot][lane] = make_ulonglong2(v * i, x - i);
            if (lane == 0) ring_i[wave][slot] = last ? ~0ull : (unsigned long long)i0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) full[wave][slot] = 1u;
            slot ^= 1u;
            if (last) break;
        }
    } else {
        unsigned slot[3] = {0, 0, 0};
        unsigned live = 7u;
        while (live) {
#pragma unroll
            for (unsigned p = 0; p < 3; ++p) {
                if (!(live & (1u << p))) continue;
                if (!full[p][slot[p]]) continue;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const unsigned long long i0 = ring_i[p][slot[p]];
                if (i0 == ~0ull) {
                    live &= ~(1u << p);
                } else {
                    const size_t i = (size_t)i0 + lane;
                    const ulonglong2 xa = ring_a[p][slot[p]][lane], xb = ring_b[p][slot[p]][lane];
                    if (i < n) { a[i] = xa; b[i] = xb; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) full[p][slot[p]] = 0u;
                slot[p] ^= 1u;
            }
        }
    }
}
template <int W>
__global__ __launch_bounds__(256) void one_pass_work(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v, int per_tile) {
    size_t base = (size_t)blockIdx.x * per_tile;
    for (int r = threadIdx.x; r < per_tile && base + r < n; r += 256) {
        size_t i = base + r;
        unsigned long long x = v + i;
#pragma unroll
        for (int w = 0; w < W; ++w) x = x * 0x9E3779B97F4A7C15ull + w;
        a[i] = make_ulonglong2(x, v ^ i); b[i] = make_ulonglong2(v * i, x - i);
    }
}
template <class F> float timeit(F f, int reps = 9) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}
int main() {
    size_t n = (size_t)500'000'000;
    ulonglong2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    double gb = 2.0 * n * 16 / 1e9;
#define RUN(label, ...) { float t = timeit([&] { __VA_ARGS__; }); printf("%-64s %.3f ms  %.1f GB/s\n", label, t, gb / t * 1e3); fflush(stdout); }
#define P(MODE, SL, grid) hipLaunchKernelGGL((persistent<MODE, SL>), dim3(grid), dim3(256), 0, 0, a, b, n, 1ull)
    RUN("one pass per workgroup (4 KiB per array)", hipLaunchKernelGGL(one_pass, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n, 1ull, 256));
    RUN("two passes per workgroup", hipLaunchKernelGGL(one_pass, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, a, b, n, 1ull, 512));
    for (int grid : {256, 512, 768, 1024, 2048, 4096}) {
        char l[128];
        snprintf(l, 128, "persistent grid-stride, grid %d", grid); RUN(l, P(0, 0, grid));
    }
    for (int grid : {768, 2048}) {
        char l[128];
        snprintf(l, 128, "persistent + s_sleep 1 per pair, grid %d", grid); RUN(l, P(1, 1, grid));
        snprintf(l, 128, "persistent + s_sleep 4 per pair, grid %d", grid); RUN(l, P(1, 4, grid));
        snprintf(l, 128, "persistent + s_sleep 16 per pair, grid %d", grid); RUN(l, P(1, 16, grid));
        snprintf(l, 128, "persistent + s_sleep 64 per pair, grid %d", grid); RUN(l, P(1, 64, grid));
        snprintf(l, 128, "persistent, random start phase, grid %d", grid); RUN(l, P(2, 0, grid));
        snprintf(l, 128, "persistent, one array per wavefront, grid %d", grid); RUN(l, P(3, 0, grid));
        snprintf(l, 128, "persistent, 4 pairs then vmcnt(0), grid %d", grid); RUN(l, P(4, 0, grid));
        snprintf(l, 128, "persistent, s_sleep 32 every 4th pair, grid %d", grid); RUN(l, P(5, 32, grid));
        snprintf(l, 128, "persistent + 16 mul-adds per pair, grid %d", grid); RUN(l, hipLaunchKernelGGL(persistent_work<16>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "persistent + 64 mul-adds per pair, grid %d", grid); RUN(l, hipLaunchKernelGGL(persistent_work<64>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
    }
    for (int grid : {768, 1536}) {
        char l[128];
#define HW(W) snprintf(l, 128, "persistent, %d mul-adds per pair, NO stores, grid %d", W, grid); RUN(l, hipLaunchKernelGGL(persistent_work_only<W>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull)); \
              snprintf(l, 128, "persistent, %d mul-adds per pair + stores, grid %d", W, grid); RUN(l, hipLaunchKernelGGL(persistent_work_heavy<W>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        HW(128) HW(192) HW(256) HW(384)
#define SP(W) snprintf(l, 128, "persistent, %d mul-adds per pair, 3 compute + 1 store wavefront, grid %d", W, grid); RUN(l, hipLaunchKernelGGL(persistent_split<W>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        SP(128) SP(192) SP(256)
    }
    RUN("one pass + 16 mul-adds per pair", hipLaunchKernelGGL(one_pass_work<16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n, 1ull, 256));
    RUN("one pass + 64 mul-adds per pair", hipLaunchKernelGGL(one_pass_work<64>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n, 1ull, 256));
    RUN("four passes + 64 mul-adds per pair", hipLaunchKernelGGL(one_pass_work<64>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n, 1ull, 1024));
    return 0;
}
