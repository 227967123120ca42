// store_flavours.hip -- does the cache policy of the streaming stores matter?  Two-array fill,
// one 4 KiB chunk per array per workgroup (the best shape), stores issued as plain / nt / sc1 /
// sc0 sc1 via inline asm.  Measurement tooling only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int FL>
__device__ __forceinline__ void st16(void* p, u32x4 v) {
    if constexpr (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (FL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(p) = v;  // FL == 5: compiler-generated store
}
__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
// DATA: 0 = low entropy (counters / constants), 1 = high entropy (hash of the index), 2 = zeros
template <int FL, int DATA = 0>
__global__ __launch_bounds__(256) void fill(ulonglong2* a, ulonglong2* b, size_t n, unsigned v, int passes) {
    size_t base = (size_t)blockIdx.x * 256 * passes;
    for (int p = 0; p < passes; ++p) {
        size_t i = base + p * 256 + threadIdx.x;
        if (i < n) {
            u32x4 x = {v + (unsigned)i, v ^ (unsigned)i, v, (unsigned)(i >> 32)}, y = x;
            if constexpr (DATA == 1) {
                unsigned long long h0 = mix(i * 2 + v), h1 = mix(i * 2 + 1 + v), h2 = mix(~i), h3 = mix(i ^ 0x5555555555ull);
                x = u32x4{(unsigned)h0, (unsigned)(h0 >> 32), (unsigned)h1, (unsigned)(h1 >> 32)};
                y = u32x4{(unsigned)h2, (unsigned)(h2 >> 32), (unsigned)h3, (unsigned)(h3 >> 32)};
            } else if constexpr (DATA == 2) {
                x = u32x4{0, 0, 0, 0}; y = x;
            }
            st16<FL>(a + i, x); st16<FL>(b + i, y);
        }
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
    const char* names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0"};
    for (int passes : {1, 2, 4}) {
        unsigned g = (unsigned)((n + 256ull * passes - 1) / (256ull * passes));
        for (int round = 0; round < 2; ++round) {
#define RUN(FL) { float t = timeit([&] { hipLaunchKernelGGL(fill<FL>, dim3(g), dim3(256), 0, 0, a, b, n, 7u, passes); }); printf("passes %d  %-8s %.3f ms  %.1f GB/s\n", passes, names[FL], t, gb / t * 1e3); }
            RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
        }
    }
    printf("== inline-asm store vs compiler-generated store (1 pass) ==\n");
    {
        unsigned g = (unsigned)((n + 255) / 256);
        for (int round = 0; round < 4; ++round) {
            { float t = timeit([&] { hipLaunchKernelGGL((fill<0, 0>), dim3(g), dim3(256), 0, 0, a, b, n, 7u, 1); }); printf("asm plain store       %.3f ms  %.1f GB/s\n", t, gb / t * 1e3); }
            { float t = timeit([&] { hipLaunchKernelGGL((fill<5, 0>), dim3(g), dim3(256), 0, 0, a, b, n, 7u, 1); }); printf("compiler store        %.3f ms  %.1f GB/s\n", t, gb / t * 1e3); }
            { float t = timeit([&] { hipLaunchKernelGGL((fill<5, 1>), dim3(g), dim3(256), 0, 0, a, b, n, 7u, 1); }); printf("compiler store, hash  %.3f ms  %.1f GB/s\n", t, gb / t * 1e3); }
        }
    }
    printf("== data dependence (plain stores, 1 and 2 passes) ==\n");
    for (int passes : {1, 2}) {
        unsigned g = (unsigned)((n + 256ull * passes - 1) / (256ull * passes));
        for (int round = 0; round < 3; ++round) {
            { float t = timeit([&] { hipLaunchKernelGGL((fill<0, 0>), dim3(g), dim3(256), 0, 0, a, b, n, 7u, passes); }); printf("passes %d  low-entropy data   %.3f ms  %.1f GB/s\n", passes, t, gb / t * 1e3); }
            { float t = timeit([&] { hipLaunchKernelGGL((fill<0, 1>), dim3(g), dim3(256), 0, 0, a, b, n, 7u, passes); }); printf("passes %d  high-entropy data  %.3f ms  %.1f GB/s\n", passes, t, gb / t * 1e3); }
            { float t = timeit([&] { hipLaunchKernelGGL((fill<0, 2>), dim3(g), dim3(256), 0, 0, a, b, n, 7u, passes); }); printf("passes %d  zeros             %.3f ms  %.1f GB/s\n", passes, t, gb / t * 1e3); }
        }
    }
    return 0;
}
