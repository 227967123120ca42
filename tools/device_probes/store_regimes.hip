// store_regimes.hip -- why do persistent / long-lived workgroups write slower than short ones?
// Measurement tooling only.  Variants of a two-array fill (2 x 8 GB, 16 B per lane per store):
//   A  one pass per workgroup (reference point)
//   B  persistent grid-stride, unthrottled
//   C  persistent, s_waitcnt vmcnt(k) after every store pair (k outstanding ops allowed)
//   D  persistent, each workgroup owns one contiguous region (sequential within the workgroup)
//   E  P passes per workgroup with vmcnt throttle
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int WAIT>
__device__ __forceinline__ void throttle() {
    if constexpr (WAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (WAIT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (WAIT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (WAIT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (WAIT == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

template <int WAIT>  // WAIT < 0: unthrottled
__global__ __launch_bounds__(256) void fill_gridstride(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i);
        throttle<WAIT>();
    }
}
template <int WAIT>
__global__ __launch_bounds__(256) void fill_region(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    size_t per = (n + gridDim.x - 1) / gridDim.x; per = (per + 255) / 256 * 256;
    size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i);
        throttle<WAIT>();
    }
}
template <int WAIT>
__global__ __launch_bounds__(256) void fill_tiled(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v, int per_tile) {
    size_t base = (size_t)blockIdx.x * per_tile;
    for (int r = threadIdx.x; r < per_tile && base + r < n; r += 256) {
        size_t i = base + r;
        a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i);
        throttle<WAIT>();
    }
}
// F: P chunks per workgroup, chunk j of workgroup b is chunk b + j*gridDim (P in-order sliding windows)
__global__ __launch_bounds__(256) void fill_windows(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v, int P) {
    for (int j = 0; j < P; ++j) {
        size_t i = ((size_t)j * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        if (i < n) { a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i); }
    }
}
// G: one chunk per workgroup, but workgroup b writes chunk (b % 8) * (nchunks / 8) + b / 8: with round-robin
// dispatch every XCD then owns one contiguous eighth of each array (XCD-aware remap, T1 of the guide)
__global__ __launch_bounds__(256) void fill_xcd(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v, int passes) {
    size_t nchunks = gridDim.x, per = nchunks / 8;
    size_t c = blockIdx.x < per * 8 ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    for (int p = 0; p < passes; ++p) {
        size_t i = (c * passes + p) * 256 + threadIdx.x;
        if (i < n) { a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i); }
    }
}
// H: each lane writes 32 contiguous bytes per array as two 16-byte stores (lane stride 32 B): the
// shape of a "4 kmers per lane" design; one workgroup covers 8 KiB per array in a single pass
__global__ __launch_bounds__(256) void fill_wide(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) {
        a[i] = make_ulonglong2(v + i, v ^ i); a[i + 1] = make_ulonglong2(v - i, v * i);
        b[i] = make_ulonglong2(v * i, v - i); b[i + 1] = make_ulonglong2(v ^ i, v + i);
    }
}
// I: like H but 64 contiguous bytes per lane (four 16-byte stores per array)
__global__ __launch_bounds__(256) void fill_wide4(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[i + j] = make_ulonglong2(v + i + j, v ^ i); b[i + j] = make_ulonglong2(v * i, v - i - j); }
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
#define RUN(label, ...) { float t = timeit([&] { __VA_ARGS__; }); printf("%-58s %.3f ms  %.1f GB/s\n", label, t, gb / t * 1e3); }
    unsigned nt256 = (unsigned)((n + 255) / 256);
    RUN("A  1 pass/workgroup", hipLaunchKernelGGL(fill_tiled<-1>, dim3(nt256), dim3(256), 0, 0, a, b, n, 1ull, 256));
    for (int grid : {1024, 2048, 4096}) {
        char l[128];
        snprintf(l, 128, "B  grid-stride unthrottled grid=%d", grid); RUN(l, hipLaunchKernelGGL(fill_gridstride<-1>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "C  grid-stride vmcnt(0)  grid=%d", grid);  RUN(l, hipLaunchKernelGGL(fill_gridstride<0>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "C  grid-stride vmcnt(2)  grid=%d", grid);  RUN(l, hipLaunchKernelGGL(fill_gridstride<2>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "C  grid-stride vmcnt(4)  grid=%d", grid);  RUN(l, hipLaunchKernelGGL(fill_gridstride<4>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "C  grid-stride vmcnt(8)  grid=%d", grid);  RUN(l, hipLaunchKernelGGL(fill_gridstride<8>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "C  grid-stride vmcnt(16) grid=%d", grid);  RUN(l, hipLaunchKernelGGL(fill_gridstride<16>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "D  contiguous region per workgroup, unthrottled grid=%d", grid); RUN(l, hipLaunchKernelGGL(fill_region<-1>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
        snprintf(l, 128, "D  contiguous region per workgroup, vmcnt(2) grid=%d", grid); RUN(l, hipLaunchKernelGGL(fill_region<2>, dim3(grid), dim3(256), 0, 0, a, b, n, 1ull));
    }
    for (int per_tile : {512, 1024, 2048, 4096}) {
        unsigned nt = (unsigned)((n + per_tile - 1) / per_tile);
        char l[128];
        snprintf(l, 128, "E  %d passes/workgroup unthrottled", per_tile / 256); RUN(l, hipLaunchKernelGGL(fill_tiled<-1>, dim3(nt), dim3(256), 0, 0, a, b, n, 1ull, per_tile));
        snprintf(l, 128, "E  %d passes/workgroup vmcnt(0)", per_tile / 256);    RUN(l, hipLaunchKernelGGL(fill_tiled<0>, dim3(nt), dim3(256), 0, 0, a, b, n, 1ull, per_tile));
        snprintf(l, 128, "E  %d passes/workgroup vmcnt(2)", per_tile / 256);    RUN(l, hipLaunchKernelGGL(fill_tiled<2>, dim3(nt), dim3(256), 0, 0, a, b, n, 1ull, per_tile));
    }
    for (int rep = 0; rep < 2; ++rep) {
        RUN("H  32 contiguous bytes per lane (2 x 16 B), 1 pass", hipLaunchKernelGGL(fill_wide, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, 0, a, b, n, 1ull));
        RUN("I  64 contiguous bytes per lane (4 x 16 B), 1 pass", hipLaunchKernelGGL(fill_wide4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, a, b, n, 1ull));
        RUN("   (reference: 2 passes/workgroup, 16 B per lane per pass)", hipLaunchKernelGGL(fill_tiled<-1>, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, a, b, n, 1ull, 512));
        RUN("   (reference: 1 pass/workgroup)", hipLaunchKernelGGL(fill_tiled<-1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n, 1ull, 256));
    }
    for (int passes : {1, 2, 4}) {
        unsigned g = (unsigned)((n + 256ull * passes - 1) / (256ull * passes));
        char l[128];
        snprintf(l, 128, "G  XCD-contiguous remap, %d passes/workgroup", passes);
        RUN(l, hipLaunchKernelGGL(fill_xcd, dim3(g), dim3(256), 0, 0, a, b, n, 1ull, passes));
        snprintf(l, 128, "   (reference: plain order, %d passes/workgroup)", passes);
        RUN(l, hipLaunchKernelGGL(fill_tiled<-1>, dim3(g), dim3(256), 0, 0, a, b, n, 1ull, 256 * passes));
    }
    for (int P : {1, 2, 3, 4, 8}) {
        unsigned g = (unsigned)((n + 256ull * P - 1) / (256ull * P));
        char l[128]; snprintf(l, 128, "F  %d far-apart chunks per workgroup (sliding windows)", P);
        RUN(l, hipLaunchKernelGGL(fill_windows, dim3(g), dim3(256), 0, 0, a, b, n, 1ull, P));
    }
    return 0;
}
