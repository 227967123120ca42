// vmm_stripes.hip -- can the region classes of HBM (profiles/r03_alloc.md) be made a property of every ARRAY instead of a
// property of where a 230 GB reservation happens to put it?  
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/vmm_stripes tools/vmm_stripes.hip && tools/vmm_stripes [total_GiB]
//
// HIP's virtual-memory management gives physical memory in handles (hipMemCreate) that are mapped wherever one likes
// (hipMemAddressReserve + hipMemMap).  This program
//   0. reports what the API does here (granularity, map with an offset, one handle at two addresses, cost per call);
//   1. creates `total` GiB as 1 GiB handles, maps them in creation order and measures their class map with the arena's own
//      two-stream probe (memory_api.hip::calibrate): is creation order physical order, is a 1 GiB handle of one class?
//   2. re-creates the same sequence with two ZONES of 2 MiB handles (one in the first class, one in the first chunk of another
//      class), checks the zones' classes, and builds arrays out of them: pure A, pure B, and A/B stripes of 2 MiB ... 1 GiB;
//   3. fills them in the output shapes of the stream kernels: one array (16 KiB per workgroup), two arrays (8 KiB of each per
//      workgroup) in the same and in opposite stripe phase, a copy.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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

constexpr size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;

__global__ __launch_bounds__(256) void fill2(ulonglong2 *a, ulonglong2 *b) {  // 8 KiB of each per workgroup
    const uint64_t w = blockIdx.x;
    ulonglong2 *p = a + w * 512u, *q = b + w * 512u;
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        p[i] = make_ulonglong2(w, i);
        q[i] = make_ulonglong2(i, w);
    }
}
__global__ __launch_bounds__(256) void fill1(ulonglong2 *a) {  // 16 KiB per workgroup
    const uint64_t w = blockIdx.x;
    ulonglong2 *p = a + w * 1024u;
    for (uint32_t i = threadIdx.x; i < 1024u; i += 256u) p[i] = make_ulonglong2(w, i);
}
__global__ __launch_bounds__(256) void copy1(const ulonglong2 *__restrict__ a, ulonglong2 *__restrict__ b) {  // 8 KiB read + written
    const uint64_t w = blockIdx.x;
    const ulonglong2 *p = a + w * 512u;
    ulonglong2 *q = b + w * 512u;
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) q[i] = p[i];
}

static hipEvent_t g_e0, g_e1;

template <class F>
static float best_ms(F launch, int reps = 4) {
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(g_e0, 0);
        launch();
        (void)hipEventRecord(g_e1, 0);
        (void)hipEventSynchronize(g_e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, g_e0, g_e1);
        if (r && ms < best) best = ms;
    }
    return best;
}
static double rate_fill2(char *a, char *b, size_t bytes_each) {
    const float ms = best_ms([&] { hipLaunchKernelGGL(fill2, dim3((unsigned)(bytes_each / 8192)), dim3(256), 0, 0, (ulonglong2 *)a, (ulonglong2 *)b); });
    return 2.0 * (double)bytes_each / 1e9 / (double)ms;  // TB/s
}
static double rate_fill1(char *a, size_t bytes) {
    const float ms = best_ms([&] { hipLaunchKernelGGL(fill1, dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, (ulonglong2 *)a); });
    return (double)bytes / 1e9 / (double)ms;
}
static double rate_copy(char *a, char *b, size_t bytes) {
    const float ms = best_ms([&] { hipLaunchKernelGGL(copy1, dim3((unsigned)(bytes / 8192)), dim3(256), 0, 0, (const ulonglong2 *)a, (ulonglong2 *)b); });
    return 2.0 * (double)bytes / 1e9 / (double)ms;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Phys {
    hipMemGenericAllocationHandle_t h;
    size_t bytes;
};

static hipMemAllocationProp g_prop;
static hipMemAccessDesc g_access;

// class of every 2 GiB granule of a mapped range (two 1 GiB streams: the granule beside one representative per class found so far)
static std::string class_map(char *base, size_t bytes, std::vector<int> *cls_out, double *one_class, double *two_class) {
    const size_t G = 2 * GiB, n = bytes / G;
    std::vector<float> same(n);
    for (size_t g = 0; g < n; ++g)
        same[g] = best_ms([&] { hipLaunchKernelGGL(fill2, dim3((unsigned)(GiB / 8192)), dim3(256), 0, 0, (ulonglong2 *)(base + g * G), (ulonglong2 *)(base + g * G + GiB)); }, 3);
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
            const float t = best_ms([&] { hipLaunchKernelGGL(fill2, dim3((unsigned)(GiB / 8192)), dim3(256), 0, 0, (ulonglong2 *)(base + g * G), (ulonglong2 *)(base + refs[r] * G + GiB)); }, 3);
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
    if (cls_out) *cls_out = cls;
    if (one_class) *one_class = 2.0 * (double)GiB / 1e9 / (double)slow;
    if (two_class) *two_class = 2.0 * (double)GiB / 1e9 / (double)fastest;
    return s;
}

int main(int argc, char **argv) {
    size_t total = (argc > 1 ? (size_t)std::atol(argv[1]) : 192) * GiB;
    CHECK(hipSetDevice(0));
    CHECK(hipEventCreate(&g_e0));
    CHECK(hipEventCreate(&g_e1));
    int vmm = 0;
    CHECK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, 0));
    g_prop = {};
    g_prop.type = hipMemAllocationTypePinned;
    g_prop.location.type = hipMemLocationTypeDevice;
    g_prop.location.id = 0;
    g_access = {};
    g_access.location = g_prop.location;
    g_access.flags = hipMemAccessFlagsProtReadWrite;
    size_t gmin = 0, grec = 0, free_b = 0, total_b = 0;
    CHECK(hipMemGetAllocationGranularity(&gmin, &g_prop, hipMemAllocationGranularityMinimum));
    CHECK(hipMemGetAllocationGranularity(&grec, &g_prop, hipMemAllocationGranularityRecommended));
    CHECK(hipMemGetInfo(&free_b, &total_b));
    std::printf("vmm supported %d, granularity min %zu recommended %zu, free %.1f GiB of %.1f\n", vmm, gmin, grec, (double)free_b / GiB, (double)total_b / GiB);
    if (!vmm) return 2;
    if (total + 8 * GiB > free_b) total = (free_b - 8 * GiB) / (4 * GiB) * (4 * GiB);

    // ---- 0. what the API does
    {
        hipMemGenericAllocationHandle_t h;
        CHECK(hipMemCreate(&h, 4 * MiB, &g_prop, 0));
        void *va = nullptr;
        CHECK(hipMemAddressReserve(&va, 16 * MiB, 0, nullptr, 0));
        hipError_t e = hipMemMap(va, 2 * MiB, 2 * MiB, h, 0);
        std::printf("hipMemMap with offset 2 MiB of a 4 MiB handle: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) (void)hipMemUnmap(va, 2 * MiB);
        (void)hipGetLastError();
        e = hipMemMap(va, 4 * MiB, 0, h, 0);
        std::printf("hipMemMap whole handle: %s\n", hipGetErrorString(e));
        hipError_t e2 = hipMemMap((char *)va + 8 * MiB, 4 * MiB, 0, h, 0);
        std::printf("the same handle at a second address: %s\n", hipGetErrorString(e2));
        (void)hipGetLastError();
        if (e == hipSuccess) (void)hipMemUnmap(va, 4 * MiB);
        if (e2 == hipSuccess) (void)hipMemUnmap((char *)va + 8 * MiB, 4 * MiB);
        (void)hipMemRelease(h);
        for (size_t sz : {2 * MiB, 64 * MiB, GiB}) {  // cost per call
            const int n = sz == GiB ? 8 : 256;
            std::vector<hipMemGenericAllocationHandle_t> hs(n);
            void *r = nullptr;
            CHECK(hipMemAddressReserve(&r, sz * n, 0, nullptr, 0));
            double t0 = now_s();
            for (int i = 0; i < n; ++i) CHECK(hipMemCreate(&hs[i], sz, &g_prop, 0));
            double t1 = now_s();
            for (int i = 0; i < n; ++i) CHECK(hipMemMap((char *)r + sz * i, sz, 0, hs[i], 0));
            double t2 = now_s();
            CHECK(hipMemSetAccess(r, sz * n, &g_access, 1));
            double t3 = now_s();
            CHECK(hipMemset(r, 1, std::min(sz * n, (size_t)64 * MiB)));
            CHECK(hipDeviceSynchronize());
            double t4 = now_s();
            for (int i = 0; i < n; ++i) CHECK(hipMemUnmap((char *)r + sz * i, sz));
            double t5 = now_s();
            for (int i = 0; i < n; ++i) CHECK(hipMemRelease(hs[i]));
            double t6 = now_s();
            std::printf("handles of %5zu MiB x %3d: create %7.1f us, map %7.1f us, set access (whole range, per handle) %7.1f us, unmap %7.1f us, release %7.1f us each\n",
                        sz / MiB, n, (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6, (t3 - t2) / n * 1e6, (t5 - t4) / n * 1e6, (t6 - t5) / n * 1e6);
        }
        std::fflush(stdout);
    }

    // ---- 0b. the map of one hipMalloc block of the same size (what the arena of rounds 3-4 sees)
    {
        void *p = nullptr;
        CHECK(hipMalloc(&p, total));
        double one = 0, two = 0;
        std::string m = class_map((char *)p, total, nullptr, &one, &two);
        std::printf("hipMalloc %zu GiB          : %s (one class %.2f TB/s, two %.2f)\n", total / GiB, m.c_str(), one, two);
        CHECK(hipFree(p));
        std::fflush(stdout);
    }

    // ---- 1. zones of 2 MiB handles between ballast of 1 GiB handles, all mapped in creation order under ONE range that is never
    //         unmapped (tools/vmm_va.hip: the class belongs to the PHYSICAL memory -- the same handles mapped in reverse under a fresh
    //         range give the mirrored map -- and a range that was mapped once keeps its OLD translations when it is unmapped and
    //         mapped again, even after hipMemAddressFree + a new reservation at the same address: every layout below gets a fresh
    //         range and no reservation is ever freed)
    const size_t H = 2 * MiB, ZONE = 8 * GiB, nz = ZONE / H, n_zones = 3;
    const size_t ballast_each = (total - n_zones * ZONE) / (n_zones - 1) / GiB;
    struct Piece {
        hipMemGenericAllocationHandle_t h;
        size_t bytes, at;  // offset under the creation-order range
    };
    std::vector<Piece> pieces;
    size_t at = 0;
    {
        double t0 = now_s();
        for (size_t z = 0; z < n_zones; ++z) {
            for (size_t i = 0; i < nz; ++i) {
                Piece pc{nullptr, H, at};
                CHECK(hipMemCreate(&pc.h, H, &g_prop, 0));
                pieces.push_back(pc);
                at += H;
            }
            for (size_t i = 0; z + 1 < n_zones && i < ballast_each; ++i) {
                Piece pc{nullptr, GiB, at};
                CHECK(hipMemCreate(&pc.h, GiB, &g_prop, 0));
                pieces.push_back(pc);
                at += GiB;
            }
        }
        std::printf("%zu zones of %zu handles of 2 MiB with %zu GiB of 1 GiB handles between them: created in %.2f s\n", n_zones, nz, ballast_each, now_s() - t0);
    }
    const size_t span = at;
    void *r1 = nullptr;
    CHECK(hipMemAddressReserve(&r1, span, 0, nullptr, 0));
    char *R1 = (char *)r1;
    {
        double t0 = now_s();
        for (const Piece &pc : pieces) CHECK(hipMemMap(R1 + pc.at, pc.bytes, 0, pc.h, 0));
        CHECK(hipMemSetAccess(r1, span, &g_access, 1));
        std::printf("mapped in creation order at %p in %.2f s\n", r1, now_s() - t0);
    }
    std::vector<int> cls;
    {
        double one = 0, two = 0;
        std::string m = class_map(R1, span, &cls, &one, &two);
        std::printf("creation order: %s (one class %.2f TB/s, two %.2f)\n", m.c_str(), one, two);
    }
    // granules (2 GiB) of the zones whose class is certain: the same as both neighbours'
    std::vector<std::vector<size_t>> by_class(8);  // class -> piece indices (2 MiB handles), granule by granule
    {
        size_t idx = 0;
        for (size_t z = 0; z < n_zones; ++z) {
            for (size_t i = 0; i < nz; ++i, ++idx) {
                const size_t g = pieces[idx].at / (2 * GiB);
                const bool inner = (g == 0 || cls[g - 1] == cls[g]) && (g + 1 >= cls.size() || cls[g + 1] == cls[g]);
                if (inner && cls[g] >= 0) by_class[(size_t)cls[g]].push_back(idx);
            }
            idx += z + 1 < n_zones ? ballast_each : 0;
        }
    }
    size_t c1 = 0, c2 = 1;
    {
        std::vector<size_t> order(8);
        for (size_t i = 0; i < 8; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return by_class[a].size() > by_class[b].size(); });
        c1 = order[0];
        c2 = order[1];
        for (size_t i = 0; i < 8; ++i)
            if (!by_class[i].empty()) std::printf("class %c: %.1f GiB of 2 MiB handles in granules of certain class\n", (char)('A' + i), (double)(by_class[i].size() * H) / GiB);
    }
    const size_t per_class = std::min(by_class[c1].size(), by_class[c2].size()) / 1024 * 1024;  // handles of each class that are used: a multiple of 2 GiB
    if (per_class == 0) {
        std::printf("the zones do not hold 2 GiB of two classes: nothing to stripe on this box\n");
        return 0;
    }
    const size_t ARR = per_class * H;  // bytes of X and of Y: half of each class
    std::printf("arrays X and Y of %zu GiB each, built from %zu GiB of class %c and %zu GiB of class %c\n", ARR / GiB, ARR / GiB, (char)('A' + c1), ARR / GiB, (char)('A' + c2));
    for (size_t i = 0; i < pieces.size(); ++i)
        if (pieces[i].bytes == H) CHECK(hipMemcpy(R1 + pieces[i].at, &i, 8, hipMemcpyHostToDevice));  // tag: the piece's index
    CHECK(hipDeviceSynchronize());
    char *X = nullptr, *Y = nullptr;
    // stripe 0: X = class c1, Y = class c2; else both alternate c1, c2 in runs of `stripe` bytes, in the same phase.  Fresh range each time.
    auto layout = [&](size_t stripe) -> int {
        void *r = nullptr;
        CHECK(hipMemAddressReserve(&r, 2 * ARR, 0, nullptr, 0));
        X = (char *)r;
        Y = X + ARR;
        size_t i1 = 0, i2 = 0, bad = 0;
        std::vector<size_t> which(2 * per_class);
        for (size_t i = 0; i < 2 * per_class; ++i) {
            bool first = stripe == 0 ? i < per_class : ((i * H) / stripe) % 2 == 0;
            if (first && i1 == per_class) first = false;
            if (!first && i2 == per_class) first = true;
            which[i] = first ? by_class[c1][i1++] : by_class[c2][i2++];
            CHECK(hipMemMap(X + i * H, H, 0, pieces[which[i]].h, 0));
        }
        CHECK(hipMemSetAccess(r, 2 * ARR, &g_access, 1));
        for (size_t i = 0; i < 2 * per_class; i += 61) {  // the mapping is what was asked for
            size_t tag = ~(size_t)0;
            CHECK(hipMemcpy(&tag, X + i * H, 8, hipMemcpyDeviceToHost));
            bad += tag != which[i];
        }
        if (bad) std::printf("  (!! %zu sampled handles are NOT where they were mapped, range %p)\n", bad, r);
        return 0;
    };
    auto retag = [&]() -> int {
        for (size_t i = 0; i < pieces.size(); ++i)
            if (pieces[i].bytes == H) CHECK(hipMemcpy(R1 + pieces[i].at, &i, 8, hipMemcpyHostToDevice));
        return 0;
    };
    if (layout(0)) return 1;
    const size_t T = std::min(ARR, 2 * GiB) ;  // bytes of each array a test writes
    std::printf("\nTB/s                           one array   two arrays (8K+8K per workgroup)                      copy 8K->8K\n");
    {
        const double xa = rate_fill1(X, T), xb = rate_fill1(Y, T);
        const double aa = rate_fill2(X, X + ARR / 2, ARR / 2), bb = rate_fill2(Y, Y + ARR / 2, ARR / 2), ab = rate_fill2(X, Y, T);
        const double caa = rate_copy(X, X + ARR / 2, ARR / 2), cab = rate_copy(X, Y, T);
        std::printf("pure: X = class %c, Y = class %c  X %.2f Y %.2f  halves of X %.2f halves of Y %.2f (X,Y) %.2f           X->X %.2f X->Y %.2f\n", (char)('A' + c1), (char)('A' + c2), xa, xb, aa,
                    bb, ab, caa, cab);
    }
    for (size_t s : {2 * MiB, 4 * MiB, 8 * MiB, 16 * MiB, 32 * MiB, 64 * MiB, 256 * MiB, GiB}) {
        if (s > ARR / 2) break;
        if (retag()) return 1;
        double t0 = now_s();
        if (layout(s)) return 1;
        const double tl = now_s() - t0;
        const double x = rate_fill1(X, T);
        const double same = rate_fill2(X, Y, T);
        const double opp = rate_fill2(X, Y + s, std::min(T, ARR - s));
        const double within = rate_fill2(X, X + ARR / 2, ARR / 2);
        const double c_same = rate_copy(X, Y, T), c_opp = rate_copy(X, Y + s, std::min(T, ARR - s));
        std::printf("stripes of %4zu MiB             X %.2f         (X,Y) same phase %.2f opposite phase %.2f halves of X %.2f     X->Y same %.2f opposite %.2f   (layout %.2f s)\n", s / MiB,
                    x, same, opp, within, c_same, c_opp, tl);
        std::fflush(stdout);
    }
    // ---- 4. does the memory come back?  (unmap + release of everything; the reservations stay)
    {
        size_t f0 = 0, f1 = 0, tb = 0;
        CHECK(hipMemGetInfo(&f0, &tb));
        for (const Piece &pc : pieces) (void)hipMemUnmap(R1 + pc.at, pc.bytes);
        for (const Piece &pc : pieces) (void)hipMemRelease(pc.h);
        CHECK(hipMemGetInfo(&f1, &tb));
        std::printf("\nfree before release %.1f GiB, after unmapping the creation-order range and releasing every handle %.1f GiB (the layouts' ranges are still mapped)\n",
                    (double)f0 / GiB, (double)f1 / GiB);
    }
    return 0;
}
