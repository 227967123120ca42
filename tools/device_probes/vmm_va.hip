// vmm_va.hip -- is the "region class" of HBM (profiles/r03_alloc.md) a property of the PHYSICAL memory or of the VIRTUAL address?
// (round 5: tools/vmm_stripes.hip mapped the same 1 GiB handles in creation order and in reverse and measured the SAME class map)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/vmm_va tools/vmm_va.hip && tools/vmm_va [n_GiB]
//
//   1. n handles of 1 GiB mapped in creation order: class map M1 (two-stream probe, 2 GiB granules); each handle tagged.
//   2. the same handles mapped in reverse (tags read back through the new mapping to prove it IS reversed): class map M2.
//   3. ALIASES: two handles only, H[a] under every even GiB of the range and H[b] under every odd one (a handle may be mapped at
//      many addresses): the physical memory of every probe is the same; what varies along the range is the address alone.
//   4. FIXED ADDRESSES: two slots of 1 GiB; handles from the different runs of M1 mapped under them pair by pair.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

constexpr size_t GiB = (size_t)1 << 30;

__global__ __launch_bounds__(256) void fill2(ulonglong2 *a, ulonglong2 *b) {  // 8 KiB of each per workgroup
    const uint64_t w = blockIdx.x;
    ulonglong2 *p = a + w * 512u, *q = b + w * 512u;
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        p[i] = make_ulonglong2(w, i);
        q[i] = make_ulonglong2(i, w);
    }
}
static hipEvent_t g_e0, g_e1;
static float pair_ms(char *a, char *b) {
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(g_e0, 0);
        hipLaunchKernelGGL(fill2, dim3((unsigned)(GiB / 8192)), dim3(256), 0, 0, (ulonglong2 *)a, (ulonglong2 *)b);
        (void)hipEventRecord(g_e1, 0);
        (void)hipEventSynchronize(g_e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, g_e0, g_e1);
        best = std::min(best, ms);
    }
    return best;
}
static double tbps(float ms) { return 2.0 * (double)GiB / 1e9 / (double)ms; }

static std::string class_map(char *base, size_t bytes, std::vector<int> *cls_out) {
    const size_t G = 2 * GiB, n = bytes / G;
    std::vector<float> same(n);
    for (size_t g = 0; g < n; ++g) same[g] = pair_ms(base + g * G, base + g * G + GiB);
    std::vector<float> sorted = same;
    std::sort(sorted.begin(), sorted.end());
    const float slow = sorted[n / 2], threshold = 0.93f * slow;
    std::vector<int> cls(n, -1);
    std::vector<size_t> refs;
    float fastest = slow;
    for (size_t g = 0; g < n; ++g) {
        int c = -1;
        for (size_t r = 0; r < refs.size() && c < 0; ++r) {
            if (refs[r] == g) {
                c = (int)r;
                break;
            }
            const float t = pair_ms(base + g * G, base + refs[r] * G + GiB);
            if (t >= threshold) c = (int)r;
            else fastest = std::min(fastest, t);
        }
        if (c < 0) {
            if (refs.size() >= 8) break;
            refs.push_back(g);
            c = (int)refs.size() - 1;
        }
        cls[g] = c;
    }
    std::string s;
    for (size_t g = 0; g < n;) {
        size_t e = g;
        while (e < n && cls[e] == cls[g]) ++e;
        char buf[32];
        std::snprintf(buf, sizeof buf, "%c%zu ", cls[g] < 0 ? '?' : 'A' + cls[g], (e - g) * 2);
        s += buf;
        g = e;
    }
    char buf[96];
    std::snprintf(buf, sizeof buf, "(one class %.2f TB/s, two %.2f)", tbps(slow), tbps(fastest));
    s += buf;
    if (cls_out) *cls_out = cls;
    return s;
}

int main(int argc, char **argv) {
    const size_t n = argc > 1 ? (size_t)std::atol(argv[1]) : 160;
    CHECK(hipSetDevice(0));
    CHECK(hipEventCreate(&g_e0));
    CHECK(hipEventCreate(&g_e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc access = {};
    access.location = prop.location;
    access.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> H(n);
    for (size_t i = 0; i < n; ++i) CHECK(hipMemCreate(&H[i], GiB, &prop, 0));
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, n * GiB, 0, nullptr, 0));
    char *V = (char *)va;
    std::printf("%zu handles of 1 GiB; range at %p\n", n, va);

    // 1. creation order
    for (size_t i = 0; i < n; ++i) CHECK(hipMemMap(V + i * GiB, GiB, 0, H[i], 0));
    CHECK(hipMemSetAccess(va, n * GiB, &access, 1));
    std::vector<int> cls;
    std::printf("1. creation order : %s\n", class_map(V, n * GiB, &cls).c_str());
    for (size_t i = 0; i < n; ++i) {
        uint64_t tag = 0xabc000 + i;
        CHECK(hipMemcpy(V + i * GiB, &tag, 8, hipMemcpyHostToDevice));
    }
    CHECK(hipDeviceSynchronize());
    // 2a. reversed IN PLACE (unmap, map again at the same addresses)
    {
        for (size_t i = 0; i < n; ++i) CHECK(hipMemUnmap(V + i * GiB, GiB));
        for (size_t i = 0; i < n; ++i) CHECK(hipMemMap(V + i * GiB, GiB, 0, H[n - 1 - i], 0));
        CHECK(hipMemSetAccess(va, n * GiB, &access, 1));
        size_t rev = 0, stale = 0;
        for (size_t i = 0; i < n; ++i) {
            uint64_t tag = 0;
            CHECK(hipMemcpy(&tag, V + i * GiB, 8, hipMemcpyDeviceToHost));
            rev += tag == 0xabc000 + (n - 1 - i);
            stale += tag == 0xabc000 + i;
        }
        std::printf("2a. remapped in reverse at the SAME addresses: %zu of %zu tags where the new mapping puts them, %zu where the OLD mapping had them\n", rev, n, stale);
        for (size_t i = 0; i < n; ++i) CHECK(hipMemUnmap(V + i * GiB, GiB));
    }
    // 2b. reversed under a FRESH range
    void *va2 = nullptr;
    CHECK(hipMemAddressReserve(&va2, n * GiB, 0, nullptr, 0));
    char *W = (char *)va2;
    {
        for (size_t i = 0; i < n; ++i) CHECK(hipMemMap(W + i * GiB, GiB, 0, H[n - 1 - i], 0));
        CHECK(hipMemSetAccess(va2, n * GiB, &access, 1));
        size_t rev = 0;
        for (size_t i = 0; i < n; ++i) {
            uint64_t tag = 0;
            CHECK(hipMemcpy(&tag, W + i * GiB, 8, hipMemcpyDeviceToHost));
            rev += tag == 0xabc000 + (n - 1 - i);
        }
        std::printf("2b. reversed under a fresh range at %p (%zu of %zu tags where the mapping puts them): %s\n", va2, rev, n, class_map(W, n * GiB, nullptr).c_str());
        for (size_t i = 0; i < n; ++i) CHECK(hipMemUnmap(W + i * GiB, GiB));
    }
    // representatives: the second granule of every run of M1 that is at least 3 granules long
    std::vector<size_t> rep;
    for (size_t g = 0; g < cls.size();) {
        size_t e = g;
        while (e < cls.size() && cls[e] == cls[g]) ++e;
        if (e - g >= 3) rep.push_back(g + 1);
        g = e;
    }
    // 3. aliases, each under a fresh range
    for (size_t t = 0; t < rep.size() && t < 3; ++t) {
        const size_t a = rep[0] * 2, b = rep[t] * 2 + 1;  // handle indices (creation order): GiB a and GiB b of map 1
        void *r = nullptr;
        CHECK(hipMemAddressReserve(&r, n * GiB, 0, nullptr, 0));
        char *R = (char *)r;
        for (size_t i = 0; i < n; ++i) CHECK(hipMemMap(R + i * GiB, GiB, 0, H[i % 2 ? b : a], 0));
        CHECK(hipMemSetAccess(r, n * GiB, &access, 1));
        std::printf("3. range at %p: aliases of handle %3zu (class %c in map 1) under even GiB, handle %3zu (class %c) under odd GiB: %s\n", r, a, 'A' + cls[a / 2], b,
                    'A' + cls[b / 2], class_map(R, n * GiB, nullptr).c_str());
        for (size_t i = 0; i < n; ++i) CHECK(hipMemUnmap(R + i * GiB, GiB));
    }
    // 4. pairs of handles, each pair under a fresh range of 2 GiB
    std::printf("4. pairs of handles from the runs of map 1, each pair under a fresh 2 GiB range; TB/s\n      ");
    for (size_t j = 0; j < rep.size(); ++j) std::printf("  %c@%-3zu", 'A' + cls[rep[j]], rep[j] * 2 + 1);
    std::printf("\n");
    for (size_t i = 0; i < rep.size(); ++i) {
        std::printf("%c@%-3zu ", 'A' + cls[rep[i]], rep[i] * 2);
        for (size_t j = 0; j < rep.size(); ++j) {
            void *r = nullptr;
            CHECK(hipMemAddressReserve(&r, 2 * GiB, 0, nullptr, 0));
            char *R = (char *)r;
            CHECK(hipMemMap(R, GiB, 0, H[rep[i] * 2], 0));
            CHECK(hipMemMap(R + GiB, GiB, 0, H[rep[j] * 2 + 1], 0));
            CHECK(hipMemSetAccess(r, 2 * GiB, &access, 1));
            std::printf("  %5.2f", tbps(pair_ms(R, R + GiB)));
            CHECK(hipMemUnmap(R, GiB));
            CHECK(hipMemUnmap(R + GiB, GiB));
        }
        std::printf("\n");
    }
    for (size_t i = 0; i < n; ++i) (void)hipMemRelease(H[i]);
    return 0;
}
