// vmm_life.hip -- the life cycle of HIP virtual-memory handles on this stack (ROCm 7.2, gfx950), as far as an allocator built on
// them must know it (round 5; tools/vmm_va.hip found that a range unmapped and mapped again keeps its OLD translations):
//   1. cost of hipMemCreate by handle size, first time and again after a release
//   2. remap at the same address: which sequence (if any) makes the new mapping visible -- per-handle or whole-range calls,
//      hipDeviceSynchronize, a hipMalloc + hipFree in between (the legacy path unmaps through KFD, which flushes the TLB)
//   3. does unmap + release give the memory back (hipMemGetInfo, and a hipMalloc of what should be free)?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/vmm_life tools/vmm_life.hip && tools/vmm_life
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

constexpr size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_access;

__global__ void touch(uint64_t *p, size_t stride_words, size_t n, uint64_t v) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i * stride_words] = v + i;
}
__global__ void peek(const uint64_t *p, size_t stride_words, size_t n, uint64_t *out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = p[i * stride_words];
}

static double free_gib() {
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    return (double)f / GiB;
}

int main() {
    CHECK(hipSetDevice(0));
    g_prop = {};
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = 0;
    g_access = {};
    g_access.location = g_prop.location;
    g_access.flags = hipMemAccessFlagsProtReadWrite;
    std::printf("free at start %.1f GiB\n", free_gib());

    // ---- 3. does the memory come back?
    for (int variant = 0; variant < 5; ++variant) {
        const size_t big = 32 * MiB, m = 60 * GiB / big;
        std::vector<hipMemGenericAllocationHandle_t> h(m);
        const double f0 = free_gib();
        for (size_t i = 0; i < m; ++i) CHECK(hipMemCreate(&h[i], big, &g_prop, 0));
        void *r = nullptr;
        CHECK(hipMemAddressReserve(&r, m * big, 0, nullptr, 0));
        for (size_t i = 0; i < m; ++i) CHECK(hipMemMap((char *)r + i * big, big, 0, h[i], 0));
        if (variant == 3)  // the handle released right after the map: the mapping keeps the memory until it is unmapped
            for (size_t i = 0; i < m; ++i) CHECK(hipMemRelease(h[i]));
        CHECK(hipMemSetAccess(r, m * big, &g_access, 1));
        hipLaunchKernelGGL(touch, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, 0, (uint64_t *)r, big / 8, m, 7ull);
        CHECK(hipDeviceSynchronize());
        const double f1 = free_gib();
        for (size_t i = 0; i < m; ++i) CHECK(hipMemUnmap((char *)r + i * big, big));
        const double f2 = free_gib();
        if (variant != 3)
            for (size_t i = 0; i < m; ++i) CHECK(hipMemRelease(h[i]));
        if (variant == 4) CHECK(hipDeviceReset());
        const double f3 = free_gib();
        if (variant >= 1 && variant != 4) CHECK(hipMemAddressFree(r, m * big));
        const double f4 = free_gib();
        if (variant >= 2) {
            void *p = nullptr;
            CHECK(hipMalloc(&p, 64 * MiB));
            CHECK(hipFree(p));
        }
        const double f5 = free_gib();
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, 270 * GiB);  // fits only if the 100 GiB came back
        if (e == hipSuccess) e = hipMemset(q, 0, 4096);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        std::printf("60 GiB of handles, variant %d: free %.1f -> mapped %.1f -> unmapped %.1f -> released %.1f -> %s %.1f -> %s %.1f; hipMalloc of 270 GiB then: %s\n", variant, f0, f1,
                    f2, f3, variant >= 1 ? "address range freed" : "(range kept)", f4, variant >= 2 ? "hipMalloc + hipFree" : "(nothing)", f5, hipGetErrorString(e));
        (void)hipGetLastError();
        if (q) (void)hipFree(q);
        std::fflush(stdout);
    }
    // ---- 1. creation cost by size, 16 GiB of each, twice
    for (int round = 0; round < 1; ++round)
        for (size_t sz : {2 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 256 * MiB, GiB}) {
            const size_t n = 16 * GiB / sz;
            std::vector<hipMemGenericAllocationHandle_t> h(n);
            void *r = nullptr;
            CHECK(hipMemAddressReserve(&r, n * sz, 0, nullptr, 0));
            const double t0 = now_s();
            for (size_t i = 0; i < n; ++i) CHECK(hipMemCreate(&h[i], sz, &g_prop, 0));
            const double t1 = now_s();
            for (size_t i = 0; i < n; ++i) CHECK(hipMemMap((char *)r + i * sz, sz, 0, h[i], 0));
            const double t2 = now_s();
            CHECK(hipMemSetAccess(r, n * sz, &g_access, 1));
            const double t3 = now_s();
            hipLaunchKernelGGL(touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint64_t *)r, sz / 8, n, 7ull);
            CHECK(hipDeviceSynchronize());
            const double t4 = now_s();
            for (size_t i = 0; i < n; ++i) CHECK(hipMemUnmap((char *)r + i * sz, sz));
            const double t5 = now_s();
            for (size_t i = 0; i < n; ++i) CHECK(hipMemRelease(h[i]));
            const double t6 = now_s();
            CHECK(hipMemAddressFree(r, n * sz));
            std::printf("round %d, 16 GiB as %5zu handles of %4zu MiB: create %7.1f ms/GiB, map %6.2f, set access %6.2f, first touch %6.2f, unmap %6.2f, release %6.2f ms/GiB; free now %.1f GiB\n",
                        round, n, sz / MiB, (t1 - t0) / 16 * 1e3, (t2 - t1) / 16 * 1e3, (t3 - t2) / 16 * 1e3, (t4 - t3) / 16 * 1e3, (t5 - t4) / 16 * 1e3, (t6 - t5) / 16 * 1e3, free_gib());
            std::fflush(stdout);
        }

    // ---- 2. remap at the same address
    const size_t sz = 32 * MiB, n = 64;
    uint64_t *out = nullptr;
    CHECK(hipMalloc(&out, n * 8));
    std::vector<uint64_t> host(n);
    for (int variant = 0; variant < 6; ++variant) {
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (size_t i = 0; i < n; ++i) CHECK(hipMemCreate(&h[i], sz, &g_prop, 0));
        void *r = nullptr;
        CHECK(hipMemAddressReserve(&r, n * sz, 0, nullptr, 0));
        char *R = (char *)r;
        for (size_t i = 0; i < n; ++i) CHECK(hipMemMap(R + i * sz, sz, 0, h[i], 0));
        if (variant == 1 || variant == 5)
            for (size_t i = 0; i < n; ++i) CHECK(hipMemSetAccess(R + i * sz, sz, &g_access, 1));
        else CHECK(hipMemSetAccess(r, n * sz, &g_access, 1));
        hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, 0, (uint64_t *)r, sz / 8, n, 1000ull);  // handle i holds 1000 + i
        CHECK(hipDeviceSynchronize());
        // unmap
        if (variant == 2) CHECK(hipMemUnmap(r, n * sz));
        else
            for (size_t i = 0; i < n; ++i) CHECK(hipMemUnmap(R + i * sz, sz));
        if (variant == 3) CHECK(hipDeviceSynchronize());
        if (variant == 4 || variant == 5) {
            void *p = nullptr;
            CHECK(hipMalloc(&p, 64 * MiB));
            CHECK(hipMemset(p, 0, 64 * MiB));
            CHECK(hipDeviceSynchronize());
            CHECK(hipFree(p));
        }
        // map reversed
        for (size_t i = 0; i < n; ++i) CHECK(hipMemMap(R + i * sz, sz, 0, h[n - 1 - i], 0));
        if (variant == 1 || variant == 5)
            for (size_t i = 0; i < n; ++i) CHECK(hipMemSetAccess(R + i * sz, sz, &g_access, 1));
        else CHECK(hipMemSetAccess(r, n * sz, &g_access, 1));
        hipLaunchKernelGGL(peek, dim3(1), dim3(64), 0, 0, (const uint64_t *)r, sz / 8, n, out);
        CHECK(hipMemcpy(host.data(), out, n * 8, hipMemcpyDeviceToHost));
        size_t fresh = 0, stale = 0;
        for (size_t i = 0; i < n; ++i) {
            fresh += host[i] == 1000 + (n - 1 - i);
            stale += host[i] == 1000 + i;
        }
        static const char *names[] = {"unmap per handle, map, one set-access over the range", "set-access per handle", "ONE unmap over the whole range",
                                      "hipDeviceSynchronize after the unmaps", "hipMalloc + memset + hipFree after the unmaps", "per-handle set-access and hipMalloc + hipFree"};
        std::printf("remap variant %d (%s): %zu of %zu slots show the NEW mapping, %zu the old one\n", variant, names[variant], fresh, n, stale);
        // after a later flush?
        void *p = nullptr;
        CHECK(hipMalloc(&p, 64 * MiB));
        CHECK(hipFree(p));
        hipLaunchKernelGGL(peek, dim3(1), dim3(64), 0, 0, (const uint64_t *)r, sz / 8, n, out);
        CHECK(hipMemcpy(host.data(), out, n * 8, hipMemcpyDeviceToHost));
        fresh = 0;
        for (size_t i = 0; i < n; ++i) fresh += host[i] == 1000 + (n - 1 - i);
        std::printf("      ... and after one more hipMalloc + hipFree: %zu of %zu new\n", fresh, n);
        for (size_t i = 0; i < n; ++i) (void)hipMemUnmap(R + i * sz, sz);
        for (size_t i = 0; i < n; ++i) (void)hipMemRelease(h[i]);
    }

    return 0;
}
