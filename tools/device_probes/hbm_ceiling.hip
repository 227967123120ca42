// hbm_ceiling.hip -- what a pure store stream / copy reaches on this MI355X, in the same shape
// as the canonical kernel's output (two 8 GB arrays, 16 B per lane per store, lanes consecutive).
// Measurement tooling only (not part of libkmers_hip.so).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void fill2(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        ulonglong2 x = make_ulonglong2(v + i, v ^ i), y = make_ulonglong2(v * i, v - i);
        if (NT) { __builtin_nontemporal_store(x.x, &a[i].x); __builtin_nontemporal_store(x.y, &a[i].y);
                  __builtin_nontemporal_store(y.x, &b[i].x); __builtin_nontemporal_store(y.y, &b[i].y); }
        else { a[i] = x; b[i] = y; }
    }
}
// tile-ordered variant: each block owns contiguous 32 KiB chunks of each array (like stream_kernel)
__global__ __launch_bounds__(256) void fill2_tiled(ulonglong2* __restrict__ a, ulonglong2* __restrict__ b, size_t n, unsigned long long v, int per_tile) {
    size_t ntiles = (n + per_tile - 1) / per_tile;
    for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        size_t base = t * per_tile;
        for (int r = threadIdx.x; r < per_tile && base + r < n; r += 256) {
            size_t i = base + r;
            a[i] = make_ulonglong2(v + i, v ^ i); b[i] = make_ulonglong2(v * i, v - i);
        }
    }
}
__global__ __launch_bounds__(256) void copy1(const ulonglong2* __restrict__ s, ulonglong2* __restrict__ d, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = s[i];
}

template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}


// one array only, tiled
__global__ __launch_bounds__(256) void fill1_tiled(ulonglong2* __restrict__ a, size_t n, unsigned long long v, int per_tile) {
    size_t base = (size_t)blockIdx.x * per_tile;
    for (int r = threadIdx.x; r < per_tile && base + r < n; r += 256) { size_t i = base + r; a[i] = make_ulonglong2(v + i, v ^ i); }
}

int main() {
    size_t n = (size_t)500'000'000;  // ulonglong2 elements per array = 8 GB each
    ulonglong2 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16 + (64 << 20)));
    double gb = 2.0 * n * 16 / 1e9;
    printf("== two arrays, one tile per block, 256 threads, passes = per_tile/256 ==\n");
    for (int per_tile : {256, 512, 768, 1024, 1536, 2048, 4096}) {
        size_t ntiles = (n + per_tile - 1) / per_tile;
        float t = timeit([&] { hipLaunchKernelGGL(fill2_tiled, dim3((unsigned)ntiles), dim3(256), 0, 0, a, b, n, 123ull, per_tile); });
        printf("fill2 tiled per_tile %5d passes %2d: %.3f ms  %.1f GB/s\n", per_tile, per_tile / 256, t, gb / t * 1e3);
    }
    printf("== second array shifted by off bytes (per_tile 512) ==\n");
    for (size_t off : {(size_t)0, (size_t)2048, (size_t)4096, (size_t)8192, (size_t)65536, (size_t)(1 << 20), (size_t)(2 << 20) + 4096, (size_t)(32 << 20)}) {
        int per_tile = 512; size_t ntiles = (n + per_tile - 1) / per_tile;
        ulonglong2* b2 = (ulonglong2*)((char*)b + off);
        float t = timeit([&] { hipLaunchKernelGGL(fill2_tiled, dim3((unsigned)ntiles), dim3(256), 0, 0, a, b2, n, 123ull, per_tile); });
        printf("fill2 tiled off %9zu: %.3f ms  %.1f GB/s\n", off, t, gb / t * 1e3);
    }
    printf("== one array only ==\n");
    for (int per_tile : {256, 512, 1024, 2048}) {
        size_t ntiles = (n + per_tile - 1) / per_tile;
        float t = timeit([&] { hipLaunchKernelGGL(fill1_tiled, dim3((unsigned)ntiles), dim3(256), 0, 0, a, n, 123ull, per_tile); });
        printf("fill1 tiled per_tile %5d: %.3f ms  %.1f GB/s\n", per_tile, t, gb / 2 / t * 1e3);
    }
    printf("== grid-stride fills / copy ==\n");
    for (int grid : {2048, 1 << 20}) {
        float t = timeit([&] { hipLaunchKernelGGL(fill2<false>, dim3(grid), dim3(256), 0, 0, a, b, n, 123ull); });
        printf("fill2 plain   grid %8d: %.3f ms  %.1f GB/s\n", grid, t, gb / t * 1e3);
    }
    for (int grid : {8192, 65536, 1 << 20}) {
        float t = timeit([&] { hipLaunchKernelGGL(copy1, dim3(grid), dim3(256), 0, 0, a, b, n); });
        printf("copy (8 GB -> 8 GB) grid %8d: %.3f ms  %.1f GB/s (read+write)\n", grid, t, gb / t * 1e3);
    }
    return 0;
}
