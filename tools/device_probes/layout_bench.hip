// layout_bench.hip -- the layout pass of kmers_batch (csrc/scan_kernels.hpp) on its own: spans of N reads in, offsets out, timed
// between events and checked against a host scan, for every segment length the kernel is built with.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/layout_bench tools/layout_bench.hip && tools/layout_bench [reads] [len]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../kmers.jl_amd/csrc/scan_kernels.hpp"

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

using namespace kmers;

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8000000ull;
    const uint64_t len = argc > 2 ? strtoull(argv[2], nullptr, 10) : 125ull;
    const uint32_t k = 31;
    std::vector<RaggedSpan> spans(n);
    std::vector<uint64_t> want(n + 1);
    uint64_t acc = 0, x = 88172645463325252ull;
    for (uint64_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const uint64_t l = argc > 3 ? len / 2 + x % len : len;  // a third argument: ragged lengths
        spans[i] = {i * 2 * len, l};
        want[i] = acc;
        acc += l < k ? 0 : l - k + 1;
    }
    want[n] = acc;
    RaggedSpan *d_spans;
    uint64_t *d_off;
    unsigned long long *d_layout, *d_header;
    const uint64_t n_seg = (n + LAYOUT_CHUNK - 1) / LAYOUT_CHUNK;  // (room for the shortest segments)
    CHECK(hipMalloc(&d_spans, n * 16));
    CHECK(hipMalloc(&d_off, (n + 1) * 8));
    CHECK(hipMalloc(&d_layout, (4 + 2 * n_seg) * 8));
    CHECK(hipMalloc(&d_header, 64 * 8));
    CHECK(hipMemset(d_layout, 0, (4 + 2 * n_seg) * 8));
    CHECK(hipMemset(d_header, 0, 64 * 8));
    CHECK(hipMemcpy(d_spans, spans.data(), n * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    uint64_t tickets = 0;
    uint32_t epoch = 0;
    void *d_big = nullptr;
    if (getenv("LAYOUT_AFTER_WRITES")) CHECK(hipMalloc(&d_big, (size_t)4 << 30));
    for (int chunks : {1, 1, 2, 4, 8, LAYOUT_CHUNKS}) {
        const uint64_t segs = (n + (uint64_t)LAYOUT_CHUNK * chunks - 1) / ((uint64_t)LAYOUT_CHUNK * chunks);
        float best = 1e30f, sum = 0;
        for (int rep = 0; rep < 20; ++rep) {
            LayoutArgs la{};
            la.spans = d_spans;
            la.n = n;
            la.pool_bases = n * 2 * len + len;
            la.desc = d_layout + 4;
            la.ticket = d_layout;
            la.ticket_base = tickets;
            la.offsets = d_off;
            la.header = d_header;
            la.epoch = ++epoch;
            la.k = k;
            la.step = 1;
            CHECK(hipMemsetAsync(d_off, 0xEE, (n + 1) * 8, 0));
            if (d_big) CHECK(hipMemsetAsync(d_big, rep, (size_t)4 << 30, 0));  // (a kernel that has just written 4 GB in front of it)
            CHECK(hipEventRecord(e0, 0));
            const dim3 g((unsigned)segs), b(256);
            if (chunks == 1) hipLaunchKernelGGL(ragged_layout_kernel<1>, g, b, 0, 0, la);
            else if (chunks == 2) hipLaunchKernelGGL(ragged_layout_kernel<2>, g, b, 0, 0, la);
            else if (chunks == 4) hipLaunchKernelGGL(ragged_layout_kernel<4>, g, b, 0, 0, la);
            else hipLaunchKernelGGL(ragged_layout_kernel<8>, g, b, 0, 0, la);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            tickets += segs;
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
            sum += ms;
        }
        std::vector<uint64_t> got(n + 1);
        CHECK(hipMemcpy(got.data(), d_off, (n + 1) * 8, hipMemcpyDeviceToHost));
        uint64_t bad = 0;
        for (uint64_t i = 0; i <= n; ++i) bad += got[i] != want[i];
        std::printf("%llu records, segments of %u x %d: best %.1f us, mean %.1f us  (%llu offsets differ from the host scan)\n",
                    (unsigned long long)n, LAYOUT_CHUNK, chunks, best * 1e3f, sum / 20 * 1e3f, (unsigned long long)bad);
    }
    return 0;
}
