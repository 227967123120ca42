// copy_classes.hip -- the ceiling of the copy-like kernels (f4: fx_hash / reverse_complement over an array of kmers, 8 B read + 8 B
// written per element; f1: 1 B/base of text in, 16 B/kmer out) as a function of WHERE source and destination lie: one 200 GiB
// block, the source at its start, the destination at an offset (region classes of HBM, profiles/r03_alloc.md), short-lived
// workgroups of 256 threads, 16 bytes per lane per access.  Also: reads alone, writes alone, and the copy with its destination
// written through two windows half an array apart (the stream kernels' split order).
//   hipcc --offload-arch=gfx950 -O3 -o tools/copy_classes tools/copy_classes.hip && tools/copy_classes
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every workgroup moves PER consecutive 16-byte elements per thread-pass x 256 threads
template <int MODE>  // 0 copy, 1 read only (folded into a word nobody stores), 2 write only, 3 copy in split order
__global__ __launch_bounds__(256) void mover(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, size_t n, int passes, unsigned long long *sink) {
    size_t wg = blockIdx.x;
    if (MODE == 3) {  // even workgroups walk the first half of the array, odd ones the second
        const size_t half = (gridDim.x + 1) / 2;
        wg = (wg & 1) ? half + (wg >> 1) : (wg >> 1);
        if (wg >= gridDim.x) return;
    }
    const size_t base = wg * (size_t)passes * 256u;
    unsigned long long acc = 0;
    for (int p = 0; p < passes; ++p) {
        const size_t i = base + (size_t)p * 256u + threadIdx.x;
        if (i >= n) break;
        if (MODE == 2) {
            dst[i] = make_ulonglong2(i, wg);
        } else {
            const ulonglong2 v = src[i];
            if (MODE == 1) acc ^= v.x + v.y;
            else dst[i] = make_ulonglong2(v.x * 0x517cc1b727220a95ull, v.y * 0x517cc1b727220a95ull);  // (fx_hash of two one-word kmers)
        }
    }
    if (MODE == 1 && acc == 0x6b6d657273ull) *sink = acc;
}

template <class Fn> float timeit(Fn fn, int reps = 7) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn(); fn(); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}

int main() {
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;  // 8 GiB per array (1 G one-word kmers)
    char *block; unsigned long long *sink;
    CK(hipMalloc(&block, (size_t)200 << 30)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(block, 1, bytes));
    printf("8 GiB source at the start of a 200 GiB block, 8 GiB destination at an offset; TB/s of read + written bytes\n");
    for (int passes : {4, 8}) {  // 16 / 32 KiB of each array per workgroup
        const unsigned grid = (unsigned)((n + (size_t)passes * 256 - 1) / ((size_t)passes * 256));
        auto run = [&](int mode, size_t off) {
            const ulonglong2 *s = (const ulonglong2 *)block; ulonglong2 *d = (ulonglong2 *)(block + off);
            return timeit([&] {
                if (mode == 0) hipLaunchKernelGGL(mover<0>, dim3(grid), dim3(256), 0, 0, s, d, n, passes, sink);
                else if (mode == 1) hipLaunchKernelGGL(mover<1>, dim3(grid), dim3(256), 0, 0, s, d, n, passes, sink);
                else if (mode == 2) hipLaunchKernelGGL(mover<2>, dim3(grid), dim3(256), 0, 0, s, d, n, passes, sink);
                else hipLaunchKernelGGL(mover<3>, dim3(grid), dim3(256), 0, 0, s, d, n, passes, sink);
            });
        };
        const float tr = run(1, 0), tw = run(2, 0);
        printf("%2d KiB per workgroup and array: read only %.3f ms %.2f TB/s | write only %.3f ms %.2f TB/s\n", passes * 4, tr, bytes / 1e9 / tr, tw, bytes / 1e9 / tw);
        for (int g : {8, 16, 32, 48, 64, 96, 128, 160, 188}) {
            const float t0 = run(0, (size_t)g << 30), t3 = run(3, (size_t)g << 30);
            printf("   destination at +%3d GiB: copy %.3f ms %.2f TB/s | copy, destination in split order %.3f ms %.2f TB/s\n", g, t0, 2.0 * bytes / 1e9 / t0, t3, 2.0 * bytes / 1e9 / t3);
            fflush(stdout);
        }
    }
    return 0;
}
