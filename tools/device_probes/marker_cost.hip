// marker_cost.hip -- what an event record costs the STREAM it is recorded on (round 6: bench.py's three records per step cost its
// timed region 0.5 %; the pool records one per kmers_dev_free).  A series of launches back to back, with nothing between them and with
// one hipEventRecord behind each, by the flags the event was created with.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/marker_cost tools/device_probes/marker_cost.hip && /tmp/marker_cost [kernel bytes = 256 MiB]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void fill(ulonglong2 *p, size_t n, unsigned long long v) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) p[i] = make_ulonglong2(v, i);
}

int main(int argc, char **argv) {
    const size_t bytes = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (size_t)256 << 20;
    const int launches = 200;
    CK(hipSetDevice(0));
    void *buf;
    CK(hipMalloc(&buf, bytes));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct Mode {
        const char *name;
        int flags;  // -1: no events
    } modes[] = {{"no event", -1},
                 {"hipEventDefault (timing)", (int)hipEventDefault},
                 {"hipEventDisableTiming", (int)hipEventDisableTiming},
                 {"hipEventDisableTiming | hipEventDisableSystemFence", (int)(hipEventDisableTiming | hipEventDisableSystemFence)},
                 {"hipEventDisableSystemFence (timing)", (int)hipEventDisableSystemFence}};
    double base_us = 0;
    for (int rep = 0; rep < 2; ++rep)  // (the first round warms up)
        for (const Mode &m : modes) {
            std::vector<hipEvent_t> evs;
            if (m.flags >= 0)
                for (int i = 0; i < launches; ++i) {
                    hipEvent_t e;
                    CK(hipEventCreateWithFlags(&e, (unsigned)m.flags));
                    evs.push_back(e);
                }
            CK(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < launches; ++i) {
                hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, st, static_cast<ulonglong2 *>(buf), bytes / 16, (unsigned long long)i);
                if (m.flags >= 0) CK(hipEventRecord(evs[(size_t)i], st));
            }
            CK(hipStreamSynchronize(st));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / launches;
            if (m.flags < 0) base_us = us;
            if (rep) std::printf("%-55s %8.2f us per launch%s\n", m.name, us, m.flags < 0 ? "" : "");
            if (rep && m.flags >= 0) std::printf("%-55s %8.2f us per record\n", "", us - base_us);
            for (hipEvent_t e : evs) CK(hipEventDestroy(e));
        }
    CK(hipFree(buf));
    return 0;
}
