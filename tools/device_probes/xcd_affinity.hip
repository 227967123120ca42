// xcd_affinity.hip -- does it matter WHICH XCD writes WHICH addresses?  (round 3: the C4 launch runs at 0.72-0.89 of 8 TB/s
// depending on where its two output arrays happen to lie; profiles/r03_alloc.md)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/xcd_affinity tools/xcd_affinity.hip && tools/xcd_affinity
//
// Experiment 1 (matrix): only the workgroups of ONE XCD (HW_REG_XCC_ID) store, and only into chunks of one residue class
// (chunk index % 8 == r, chunk = G bytes): 8 x 8 rates per granule G.  A fine-grained, uniform interleave of the address
// space over the HBM stacks gives a flat matrix; any structure in it is affinity between an XCD and a set of addresses.
// Experiment 2 (rotation): the whole device stores one array, one chunk per short-lived workgroup, chunk = a rotation of the
// workgroup's index inside its group of eight (workgroups b and b + 8 share an XCD: dealt round-robin) -- what a tile -> workgroup
// remap could buy, per granule and rotation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

// one XCD, one residue class: active workgroups draw chunk tickets
__global__ __launch_bounds__(256) void one_xcd(ulonglong2 *out, uint64_t n_tickets, uint32_t chunk_vec, uint32_t xcd, uint32_t residue,
                                               unsigned long long *ticket) {
    __shared__ unsigned long long s_t;
    if (xcc_id() != xcd) return;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1ull);
        __syncthreads();
        const uint64_t t = s_t;
        if (t >= n_tickets) return;
        ulonglong2 *p = out + (t * 8u + residue) * (uint64_t)chunk_vec;
        for (uint32_t i = threadIdx.x; i < chunk_vec; i += 256u) p[i] = make_ulonglong2(t, i);
    }
}

// whole device, one chunk per workgroup, rotated inside groups of eight consecutive workgroups
__global__ __launch_bounds__(256) void rotated(ulonglong2 *out, uint32_t chunk_vec, uint32_t rot, unsigned int *xcd_of_label) {
    const uint64_t b = blockIdx.x;
    const uint64_t chunk = (b & ~7ull) | ((b + rot) & 7ull);
    ulonglong2 *p = out + chunk * (uint64_t)chunk_vec;
    for (uint32_t i = threadIdx.x; i < chunk_vec; i += 256u) p[i] = make_ulonglong2(b, i);
    if (xcd_of_label && b < 8 && threadIdx.x == 0) xcd_of_label[b] = xcc_id();
}

// two arrays like the stream kernel's outputs: workgroup b writes chunk b of BOTH; the second array's base is shifted
__global__ __launch_bounds__(256) void two_arrays(ulonglong2 *a, ulonglong2 *bb, uint32_t chunk_vec) {
    const uint64_t b = blockIdx.x;
    ulonglong2 *p = a + b * (uint64_t)chunk_vec, *q = bb + b * (uint64_t)chunk_vec;
    for (uint32_t i = threadIdx.x; i < chunk_vec; i += 256u) {
        p[i] = make_ulonglong2(b, i);
        q[i] = make_ulonglong2(i, b);
    }
}

// ONE output per workgroup, alternating between two places: even workgroups fill chunk b/2 of `a`, odd ones chunk b/2 of `bb` --
// what a stream kernel with a single output array does if it visits the two halves of its tile range alternately
__global__ __launch_bounds__(256) void alternating(ulonglong2 *a, ulonglong2 *bb, uint32_t chunk_vec) {
    const uint64_t b = blockIdx.x;
    ulonglong2 *p = ((b & 1u) ? bb : a) + (b >> 1) * (uint64_t)chunk_vec;
    for (uint32_t i = threadIdx.x; i < chunk_vec; i += 256u) p[i] = make_ulonglong2(b, i);
}

__global__ __launch_bounds__(256) void copy16(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, uint32_t chunk_vec) {
    const uint64_t b = blockIdx.x;
    const ulonglong2 *p = src + b * (uint64_t)chunk_vec;
    ulonglong2 *q = dst + b * (uint64_t)chunk_vec;
    for (uint32_t i = threadIdx.x; i < chunk_vec; i += 256u) q[i] = p[i];
}
__global__ __launch_bounds__(256) void read16(const ulonglong2 *__restrict__ a, const ulonglong2 *__restrict__ bsrc, uint32_t chunk_vec, unsigned long long *sink) {
    const uint64_t b = blockIdx.x;
    const ulonglong2 *p = a + b * (uint64_t)chunk_vec, *q = bsrc + b * (uint64_t)chunk_vec;
    unsigned long long acc = 0;
    for (uint32_t i = threadIdx.x; i < chunk_vec; i += 256u) acc += p[i].x ^ q[i].y;
    if (acc == 0x123456789ull) *sink = acc;
}

struct Ptrs { ulonglong2 *p[4]; };
// k arrays, 16 KiB per workgroup in all (16 / k KiB of each)
__global__ __launch_bounds__(256) void multi(Ptrs ps, uint32_t k, uint32_t chunk_vec) {
    const uint64_t b = blockIdx.x;
    for (uint32_t j = 0; j < k; ++j) {
        ulonglong2 *p = ps.p[j] + b * (uint64_t)chunk_vec;
        for (uint32_t i = threadIdx.x; i < chunk_vec; i += 256u) p[i] = make_ulonglong2(b, i + j);
    }
}

int main(int argc, char **argv) {
    const size_t GB = argc > 1 ? (size_t)std::atoll(argv[1]) : 8;  // bytes written per timed launch of experiment 2
    const size_t bytes = GB << 30;
    char *buf = nullptr;
    CHECK(hipMalloc(&buf, 2 * bytes + (64u << 20)));
    unsigned long long *ticket = nullptr;
    unsigned int *labels = nullptr;
    CHECK(hipMalloc(&ticket, 8));
    CHECK(hipMalloc(&labels, 32));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("buffer at %p\n", (void *)buf);
    // warm up
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rotated, dim3((unsigned)(bytes / 8192)), dim3(256), 0, 0, (ulonglong2 *)buf, 512u, 0u, labels);
    CHECK(hipDeviceSynchronize());
    unsigned int lab[8];
    CHECK(hipMemcpy(lab, labels, 32, hipMemcpyDeviceToHost));
    std::printf("XCC_ID of workgroups 0..7:");
    for (int i = 0; i < 8; ++i) std::printf(" %u", lab[i]);
    std::printf("\n");

    std::printf("== experiment 2: whole device, one chunk per workgroup, chunk = rotation of the workgroup index in its group of 8 (TB/s) ==\n");
    std::printf("%10s", "G \\ rot");
    for (int r = 0; r < 8; ++r) std::printf(" %6d", r);
    std::printf("\n");
    for (uint32_t G : {4096u, 8192u, 16384u, 32768u, 65536u, 131072u}) {
        std::printf("%10u", G);
        for (uint32_t rot = 0; rot < 8; ++rot) {
            const unsigned grid = (unsigned)(bytes / G);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(rotated, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)buf, G / 16u, rot, (unsigned int *)nullptr);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = std::min(best, ms);
            }
            std::printf(" %6.2f", (double)bytes / best / 1e9);
        }
        std::printf("\n");
    }

    std::printf("== experiment 3: two arrays (8 KiB of each per workgroup), second array's base shifted by S bytes (TB/s) ==\n");
    for (size_t S : {(size_t)0, (size_t)4096, (size_t)8192, (size_t)16384, (size_t)32768, (size_t)65536, (size_t)(1u << 20), (size_t)(2u << 20), (size_t)(3u << 20),
                     (size_t)(32u << 20)}) {
        const unsigned grid = (unsigned)(bytes / 8192);
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(two_arrays, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)buf, (ulonglong2 *)(buf + bytes + S), 512u);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = std::min(best, ms);
        }
        std::printf("  shift %9zu: %6.2f TB/s\n", S, 2.0 * (double)bytes / best / 1e9);
    }

    {
        // experiment 4: does the write rate depend on WHERE in a large block the arrays lie?  One block of `big` GiB; the
        // two-array fill of experiment 3 (2 x 4 GiB) at every 8 GiB step of the block.
        const size_t big = argc > 2 ? (size_t)std::atoll(argv[2]) : 200;
        char *blk = nullptr;
        if (hipMalloc(&blk, big << 30) == hipSuccess) {
            std::printf("== experiment 4: two 4 GiB arrays (8 KiB of each per workgroup) at offset X GiB of one %zu GiB block at %p (TB/s) ==\n", big, (void *)blk);
            const size_t half = (size_t)4 << 30;
            for (size_t x = 0; x + 8 <= big; x += 8) {
                const unsigned grid = (unsigned)(half / 8192);
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(two_arrays, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)(blk + (x << 30)), (ulonglong2 *)(blk + (x << 30) + half), 512u);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) best = std::min(best, ms);
                }
                std::printf(" %zu:%.2f", x, 2.0 * (double)half / best / 1e9);
            }
            std::printf("\n");
            // experiment 5: the same fill, first array at X GiB, second array D GiB behind it: rows X, columns D
            std::printf("== experiment 5: two 2 GiB arrays, first at X GiB (rows), second at X + D GiB (columns), TB/s ==\n      D:");
            const size_t half2 = (size_t)2 << 30;
            const size_t dmax = big > 80 ? 72 : big / 2;
            for (size_t d = 2; d <= dmax; d += 2) std::printf(" %4zu", d);
            std::printf("\n");
            for (size_t x = 0; x + dmax + 2 <= big && x <= 96; x += 8) {
                std::printf("X = %3zu:", x);
                for (size_t d = 2; d <= dmax; d += 2) {
                    const unsigned grid = (unsigned)(half2 / 8192);
                    float best = 1e9f;
                    for (int rep = 0; rep < 3; ++rep) {
                        CHECK(hipEventRecord(e0, 0));
                        hipLaunchKernelGGL(two_arrays, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)(blk + (x << 30)), (ulonglong2 *)(blk + ((x + d) << 30)), 512u);
                        CHECK(hipEventRecord(e1, 0));
                        CHECK(hipEventSynchronize(e1));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep) best = std::min(best, ms);
                    }
                    std::printf(" %4.2f", 2.0 * (double)half2 / best / 1e9);
                }
                std::printf("\n");
            }
            // experiment 6: ONE stream of chunks split over two places (alternating workgroups), 16 KiB per workgroup, 4 GiB in all
            std::printf("== experiment 6: one output per workgroup (16 KiB), even workgroups at X GiB, odd ones at X + D GiB (TB/s) ==\n      D:");
            for (size_t d = 2; d <= dmax; d += 2) std::printf(" %4zu", d);
            std::printf("\n");
            for (size_t x = 0; x + dmax + 2 <= big && x <= 64; x += 16) {
                std::printf("X = %3zu:", x);
                for (size_t d = 2; d <= dmax; d += 2) {
                    const unsigned grid = (unsigned)(2 * half2 / 16384);
                    float best = 1e9f;
                    for (int rep = 0; rep < 3; ++rep) {
                        CHECK(hipEventRecord(e0, 0));
                        hipLaunchKernelGGL(alternating, dim3(grid), dim3(256), 0, 0, (ulonglong2 *)(blk + (x << 30)), (ulonglong2 *)(blk + ((x + d) << 30)), 1024u);
                        CHECK(hipEventRecord(e1, 0));
                        CHECK(hipEventSynchronize(e1));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep) best = std::min(best, ms);
                    }
                    std::printf(" %4.2f", 2.0 * (double)half2 / best / 1e9);
                }
                std::printf("\n");
            }
            // experiment 7: k output arrays per workgroup at the given GiB offsets, 16 KiB per workgroup in all, 4 GiB in all
            std::printf("== experiment 7: k arrays at the listed GiB offsets, 16 KiB per workgroup in all (TB/s) ==\n");
            const size_t sets[][5] = {{1, 0, 0, 0, 0}, {2, 0, 8, 0, 0}, {2, 0, 64, 0, 0}, {2, 0, 128, 0, 0}, {3, 0, 8, 16, 0}, {3, 0, 64, 128, 0}, {4, 0, 8, 16, 24}, {4, 0, 64, 128, 176},
                                      {4, 0, 32, 64, 96}, {2, 64, 128, 0, 0}, {2, 64, 176, 0, 0}, {2, 128, 176, 0, 0}, {2, 128, 144, 0, 0}};
            for (const auto &st : sets) {
                const uint32_t k = (uint32_t)st[0];
                Ptrs ps{};
                bool fits = true;
                for (uint32_t j = 0; j < k; ++j) {
                    ps.p[j] = (ulonglong2 *)(blk + (st[1 + j] << 30));
                    fits &= st[1 + j] + 5 <= big;
                }
                if (!fits) continue;
                const size_t total = (size_t)4 << 30;
                const unsigned grid = (unsigned)(total / 16384);
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(multi, dim3(grid), dim3(256), 0, 0, ps, k, 1024u / k);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) best = std::min(best, ms);
                }
                std::printf("  k = %u at", k);
                for (uint32_t j = 0; j < k; ++j) std::printf(" %zu", st[1 + j]);
                std::printf(": %.2f TB/s\n", (double)total / best / 1e9);
            }
            std::printf("== experiment 8: copy (16 B per lane, 8 KiB read + 8 KiB written per workgroup, 4 GiB each way) and two read streams ==\n");
            for (size_t d : {(size_t)8, (size_t)64, (size_t)128}) {
                const size_t each = (size_t)4 << 30;
                const unsigned grid = (unsigned)(each / 8192);
                float bc = 1e9f, br = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    float ms;
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)blk, (ulonglong2 *)(blk + (d << 30)), 512u);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) bc = std::min(bc, ms);
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const ulonglong2 *)blk, (const ulonglong2 *)(blk + (d << 30)), 512u, ticket);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep) br = std::min(br, ms);
                }
                std::printf("  D = %3zu GiB: copy %.2f TB/s (read + written), two read streams %.2f TB/s\n", d, 2.0 * (double)each / bc / 1e9, 2.0 * (double)each / br / 1e9);
            }
            CHECK(hipFree(blk));
        } else {
            std::printf("experiment 4: no %zu GiB block\n", big);
        }
    }

    std::printf("== experiment 1: ONE XCD stores into chunks of ONE residue class (chunk %% 8), GB/s ==\n");
    const uint64_t per_cell = 1ull << 30;  // bytes per cell
    for (uint32_t G : {4096u, 65536u, 1048576u}) {
        std::printf("G = %u: rows = XCC_ID, columns = residue\n", G);
        const uint64_t n_tickets = per_cell / G;
        for (uint32_t x = 0; x < 8; ++x) {
            std::printf("  xcd %u:", x);
            for (uint32_t r = 0; r < 8; ++r) {
                float best = 1e9f;
                for (int rep = 0; rep < 2; ++rep) {
                    CHECK(hipMemsetAsync(ticket, 0, 8, 0));
                    CHECK(hipEventRecord(e0, 0));
                    hipLaunchKernelGGL(one_xcd, dim3(2048), dim3(256), 0, 0, (ulonglong2 *)buf, n_tickets, G / 16u, x, r, ticket);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                std::printf(" %6.0f", (double)per_cell / best / 1e6);
            }
            std::printf("\n");
        }
    }
    return 0;
}
