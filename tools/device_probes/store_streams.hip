// store_streams.hip -- how a persistent grid should lay its output streams: the store pattern of unambiguous_kernel (every
// WAVEFRONT writes its own contiguous quarter of its workgroup's tile output, frame by frame, arithmetic between the frames)
// against the same bytes written by the WORKGROUP as one stream (frame f by wavefront f % 4), two arrays, 16 B per lane per store.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_streams tools/store_streams.hip && tools/store_streams
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one frame = 128 elements of 8 bytes per array = 64 lanes x 16 bytes.  A tile = 4 F frames per array.
// MODE 0: wavefront w writes frames [w F, (w + 1) F) of the tile;  MODE 1: wavefront w writes frames w, w + 4, w + 8, ...
// MODE 2: like 0 but the frames of a wavefront in bursts of B without arithmetic between them (arithmetic B-fold between bursts)
template <int MODE>
__global__ __launch_bounds__(256) void streams(ulonglong2 *__restrict__ a, ulonglong2 *__restrict__ b, size_t n_tiles, int F, int work, int burst,
                                                unsigned long long *ticket) {
    __shared__ unsigned long long s_tile;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long x = threadIdx.x + 1;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1ull);
        __syncthreads();
        const unsigned long long tile = s_tile;
        if (tile >= n_tiles) break;
        const size_t base = (size_t)tile * 4u * F * 64u;  // in 16-byte units
        for (int f = 0; f < F; ++f) {
            const int per = MODE == 2 ? ((f % burst) == 0 ? work * burst : 0) : work;
            for (int w = 0; w < per; ++w) x = x * 0x9E3779B97F4A7C15ull + w;  // (dependent: nothing to overlap inside the wavefront)
            const size_t frame = MODE == 1 ? (size_t)f * 4u + wave : (size_t)wave * F + f;
            const size_t i = base + frame * 64u + lane;
            a[i] = make_ulonglong2(x, tile);
            b[i] = make_ulonglong2(tile, x ^ f);
        }
    }
}

template <class Fn> float timeit(Fn fn, int reps = 7) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn(); CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t); }
    std::sort(ms.begin(), ms.end()); return ms[ms.size() / 2];
}

int main() {
    const size_t bytes_per_array = (size_t)4 << 30;  // 4 GiB per array
    ulonglong2 *a, *b; unsigned long long *ticket;
    CK(hipMalloc(&a, bytes_per_array)); CK(hipMalloc(&b, bytes_per_array)); CK(hipMalloc(&ticket, 8));
    printf("two arrays of 4 GiB, 16 B per lane per store; TB/s of both arrays together\n");
    for (int F : {14, 28}) {            // frames per wavefront and tile: the C5 lattice / K = 31 at p(N) = 0.04
        const size_t n_tiles = bytes_per_array / ((size_t)4 * F * 1024);
        for (int grid : {512, 768, 1024}) {
            for (int work : {0, 8}) {
                auto run = [&](int mode, int burst) {
                    return timeit([&] {
                        CK(hipMemsetAsync(ticket, 0, 8));
                        if (mode == 0) hipLaunchKernelGGL(streams<0>, dim3(grid), dim3(256), 0, 0, a, b, n_tiles, F, work, burst, ticket);
                        else if (mode == 1) hipLaunchKernelGGL(streams<1>, dim3(grid), dim3(256), 0, 0, a, b, n_tiles, F, work, burst, ticket);
                        else hipLaunchKernelGGL(streams<2>, dim3(grid), dim3(256), 0, 0, a, b, n_tiles, F, work, burst, ticket);
                    });
                };
                const double tb = 2.0 * (double)n_tiles * 4 * F * 1024 / 1e9;  // GB
                const float t0 = run(0, 1), t1 = run(1, 1), t2 = run(2, 7);
                printf("F %2d grid %4d work %3d: per-wavefront streams %.3f ms %.2f TB/s | one stream per workgroup %.3f ms %.2f TB/s | per-wavefront, bursts of 7 %.3f ms %.2f TB/s\n",
                       F, grid, work, t0, tb / t0, t1, tb / t1, t2, tb / t2);
                fflush(stdout);
            }
        }
    }
    // where the two arrays lie (region classes of HBM, profiles/r03_alloc.md): one 200 GiB block, a at its start, b at an offset
    CK(hipFree(a)); CK(hipFree(b));
    char *block;
    if (hipMalloc(&block, (size_t)200 << 30) == hipSuccess) {
        const int F = 28, grid = 1024;
        const size_t n_tiles = bytes_per_array / ((size_t)4 * F * 1024);
        const double tb = 2.0 * (double)n_tiles * 4 * F * 1024 / 1e9;
        for (int g : {4, 8, 16, 24, 32, 48, 64, 96, 128, 160, 192}) {
            ulonglong2 *pa = (ulonglong2 *)block, *pb = (ulonglong2 *)(block + ((size_t)g << 30));
            const float t0 = timeit([&] { CK(hipMemsetAsync(ticket, 0, 8)); hipLaunchKernelGGL(streams<0>, dim3(grid), dim3(256), 0, 0, pa, pb, n_tiles, F, 0, 1, ticket); });
            const float t1 = timeit([&] { CK(hipMemsetAsync(ticket, 0, 8)); hipLaunchKernelGGL(streams<0>, dim3(768), dim3(256), 0, 0, pa, pb, n_tiles, F, 0, 1, ticket); });
            printf("one block, b at +%3d GiB: persistent grid 1024 %.3f ms %.2f TB/s | grid 768 %.3f ms %.2f TB/s\n", g, t0, tb / t0, t1, tb / t1);
            fflush(stdout);
        }
    }
    return 0;
}
