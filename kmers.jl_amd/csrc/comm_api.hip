// comm_api.hip -- the communication of the sharded path behind the C ABI, on RCCL (include/kmers_hip.h,
// "the communication of the sharded path").  The reference (BioJulia/Kmers.jl) has no distributed code; the
// dependency that makes its iterators shardable is that iterate() carries only the previous K-1 symbols
// (src/iterators/FwKmers.jl:57-66, CanonicalKmers.jl:94-105).  Everything here is enqueued on the context's
// stream, so the halo words are ordered before the next kernel of the same context without a host wait.
//
// RCCL is bound at the FIRST kmers_comm_* call (dlopen), not at load time: a single-GPU consumer of libkmers_hip.so needs
// no librccl on its machine, and a process that already holds an RCCL (PyTorch brings its own librccl.so.1) shares it.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only; every call below goes through the table

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "context.hpp"

using namespace kmers;

namespace {

struct Rccl {
    void *handle = nullptr;
    std::string error;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclCommUserRank) CommUserRank = nullptr;
    decltype(&::ncclCommCount) CommCount = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclAllReduce) AllReduce = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
};

Rccl load_rccl() {
    Rccl r;
    std::vector<std::string> names;
    if (const char *env = std::getenv("KMERS_RCCL_LIB")) {
        if (std::string(env) == "none") {  // a host that wants this library never to load RCCL: the kmers_comm_* entry points answer KMERS_E_UNSUPPORTED
            r.error = "RCCL disabled by KMERS_RCCL_LIB=none: the kmers_comm_* entry points need RCCL";
            return r;
        }
        names.push_back(env);
    }
    names.push_back("librccl.so.1");  // the soname: an RCCL the process already holds, else the loader's search path + our RUNPATH
    if (const char *rocm = std::getenv("ROCM_PATH")) names.push_back(std::string(rocm) + "/lib/librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("librccl.so");
    for (const auto &n : names) {
        r.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
        if (const char *e = dlerror()) r.error = e;
    }
    if (!r.handle) {
        r.error = "librccl.so.1 cannot be loaded (" + r.error + "): the kmers_comm_* entry points need RCCL";
        return r;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(r.handle, name);
        if (!p) {
            ok = false;
            r.error = std::string("RCCL lacks ") + name;
        }
        return p;
    };
#define BIND(f) r.f = reinterpret_cast<decltype(r.f)>(sym("nccl" #f))
    BIND(GetErrorString); BIND(GetUniqueId); BIND(CommInitRank); BIND(CommDestroy); BIND(CommUserRank); BIND(CommCount);
    BIND(GroupStart); BIND(GroupEnd); BIND(Send); BIND(Recv); BIND(AllReduce); BIND(AllGather);
#undef BIND
    if (!ok) {
        dlclose(r.handle);
        r.handle = nullptr;
    }
    return r;
}

const Rccl &rccl() {
    static const Rccl r = load_rccl();  // thread-safe once
    return r;
}

int nccl_fail(kmers_ctx *ctx, const char *what, ncclResult_t r) {
    if (ctx) {
        ctx->last_error = what;
        ctx->last_error += ": ";
        ctx->last_error += rccl().GetErrorString(r);
    }
    return KMERS_E_NCCL;
}

// every entry point starts with this: without an RCCL on the machine the communication is KMERS_E_UNSUPPORTED, never an abort
#define NEED_RCCL(ctx)                                                                  \
    do {                                                                                \
        if (!rccl().handle) return fail(ctx, KMERS_E_UNSUPPORTED, rccl().error.c_str()); \
    } while (0)

#define NCCL_TRY(ctx, call)                                    \
    do {                                                       \
        ncclResult_t r_ = (call);                              \
        if (r_ != ncclSuccess) return nccl_fail(ctx, #call, r_); \
    } while (0)

static_assert(sizeof(ncclUniqueId) == KMERS_COMM_ID_BYTES, "KMERS_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

constexpr uint64_t NO_ERROR_KEY = ~0ull;

}  // namespace

extern "C" {

int kmers_comm_id(void *out_id) {
    if (!out_id) return KMERS_E_BADARG;
    if (!rccl().handle) return KMERS_E_UNSUPPORTED;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return KMERS_E_NCCL;
    std::memcpy(out_id, &id, sizeof id);
    return KMERS_OK;
}

int kmers_comm_create(kmers_ctx *ctx, const void *id_bytes, int n_ranks, int rank, void **out_comm) {
    if (!ctx) return KMERS_E_BADARG;
    if (!id_bytes || !out_comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ctx, KMERS_E_BADARG, "kmers_comm_create: bad arguments");
    *out_comm = nullptr;
    NEED_RCCL(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    ncclComm_t comm = nullptr;
    NCCL_TRY(ctx, rccl().CommInitRank(&comm, n_ranks, id, rank));
    *out_comm = comm;
    return KMERS_OK;
}

int kmers_comm_destroy(kmers_ctx *ctx, void *nccl_comm) {
    if (!ctx) return KMERS_E_BADARG;
    if (!nccl_comm) return KMERS_OK;
    NEED_RCCL(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    NCCL_TRY(ctx, rccl().CommDestroy(static_cast<ncclComm_t>(nccl_comm)));
    return KMERS_OK;
}

int kmers_comm_rank(kmers_ctx *ctx, void *nccl_comm, int *out_rank, int *out_n_ranks) {
    if (!ctx) return KMERS_E_BADARG;
    if (!nccl_comm) return fail(ctx, KMERS_E_BADARG, "nccl_comm is NULL");
    NEED_RCCL(ctx);
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    int r = 0, n = 0;
    NCCL_TRY(ctx, rccl().CommUserRank(comm, &r));
    NCCL_TRY(ctx, rccl().CommCount(comm, &n));
    if (out_rank) *out_rank = r;
    if (out_n_ranks) *out_n_ranks = n;
    return KMERS_OK;
}

int kmers_comm_sendrecv(kmers_ctx *ctx, void *nccl_comm, const uint64_t *send_dev, uint64_t send_words, int send_peer,
                        uint64_t *recv_dev, uint64_t recv_words, int recv_peer) {
    if (!ctx) return KMERS_E_BADARG;
    if (!nccl_comm) return fail(ctx, KMERS_E_BADARG, "nccl_comm is NULL");
    NEED_RCCL(ctx);
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    const bool do_send = send_peer >= 0 && send_words > 0, do_recv = recv_peer >= 0 && recv_words > 0;
    if ((do_send && !send_dev) || (do_recv && !recv_dev)) return fail(ctx, KMERS_E_BADARG, "kmers_comm_sendrecv: NULL buffer");
    if (!do_send && !do_recv) return KMERS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // one group: the send and the receive progress together (neighbours send to each other's left at the same time)
    NCCL_TRY(ctx, rccl().GroupStart());
    ncclResult_t rs = ncclSuccess, rr = ncclSuccess;
    if (do_send) rs = rccl().Send(send_dev, (size_t)send_words, ncclUint64, send_peer, comm, ctx->stream);
    if (do_recv) rr = rccl().Recv(recv_dev, (size_t)recv_words, ncclUint64, recv_peer, comm, ctx->stream);
    ncclResult_t re = rccl().GroupEnd();
    if (rs != ncclSuccess) return nccl_fail(ctx, "ncclSend", rs);
    if (rr != ncclSuccess) return nccl_fail(ctx, "ncclRecv", rr);
    if (re != ncclSuccess) return nccl_fail(ctx, "ncclGroupEnd", re);
    return KMERS_OK;
}

int kmers_halo_exchange(kmers_ctx *ctx, void *nccl_comm, const kmers_shard *shard, uint64_t *words_dev) {
    if (!ctx) return KMERS_E_BADARG;
    if (!nccl_comm || !shard) return fail(ctx, KMERS_E_BADARG, "kmers_halo_exchange: NULL communicator or shard");
    int rank = 0, n = 0;
    if (int rc = kmers_comm_rank(ctx, nccl_comm, &rank, &n)) return rc;
    const bool sends = rank > 0 && shard->send_words > 0, recvs = rank < n - 1 && shard->halo_words > 0;
    if ((sends || recvs) && !words_dev) return fail(ctx, KMERS_E_BADARG, "kmers_halo_exchange: words_dev is NULL");
    if (sends && shard->send_words > shard->n_own_words) return fail(ctx, KMERS_E_BADARG, "kmers_halo_exchange: shard owns fewer words than it must send");
    return kmers_comm_sendrecv(ctx, nccl_comm, words_dev, sends ? shard->send_words : 0, sends ? rank - 1 : -1,
                               words_dev ? words_dev + shard->n_own_words : nullptr, recvs ? shard->halo_words : 0, recvs ? rank + 1 : -1);
}

int kmers_first_error_allreduce(kmers_ctx *ctx, void *nccl_comm, kmers_result *res) {
    if (!ctx) return KMERS_E_BADARG;
    if (!nccl_comm || !res) return fail(ctx, KMERS_E_BADARG, "kmers_first_error_allreduce: NULL communicator or result");
    if (res->status != KMERS_OK && res->status != KMERS_E_ENCODE) return fail(ctx, KMERS_E_BADARG, "kmers_first_error_allreduce: status must be KMERS_OK or KMERS_E_ENCODE");
    if (res->status == KMERS_E_ENCODE && (res->err_pos >> 56)) return fail(ctx, KMERS_E_BADARG, "kmers_first_error_allreduce: err_pos out of range");
    NEED_RCCL(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // position in the high bits, raw symbol below it: the minimum is the first offending symbol in sequence order
    uint64_t key = res->status == KMERS_E_ENCODE ? ((res->err_pos << 8) | (uint64_t)(res->err_enc & 0xFFu)) : NO_ERROR_KEY;
    if (int rc = ensure_stage(ctx, 7, 16)) return rc;
    uint64_t *d = static_cast<uint64_t *>(ctx->stage[7]);
    HIP_TRY(ctx, hipMemcpyAsync(d, &key, 8, hipMemcpyHostToDevice, ctx->stream));
    NCCL_TRY(ctx, rccl().AllReduce(d, d + 1, 1, ncclUint64, ncclMin, static_cast<ncclComm_t>(nccl_comm), ctx->stream));
    uint64_t first = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&first, d + 1, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (first == NO_ERROR_KEY) {
        res->status = KMERS_OK;
        res->err_pos = 0;
        res->err_enc = 0;
        return KMERS_OK;
    }
    res->status = KMERS_E_ENCODE;
    res->err_pos = first >> 8;
    res->err_enc = (uint32_t)(first & 0xFFu);
    ctx->last_error = "EncodeError: symbol cannot be encoded in the kmer alphabet (first over all shards)";
    return KMERS_E_ENCODE;
}

int kmers_offsets_allgather(kmers_ctx *ctx, void *nccl_comm, uint64_t n_local, uint64_t *out_offset, uint64_t *out_total) {
    if (!ctx) return KMERS_E_BADARG;
    if (!nccl_comm) return fail(ctx, KMERS_E_BADARG, "nccl_comm is NULL");
    int rank = 0, n = 0;
    if (int rc = kmers_comm_rank(ctx, nccl_comm, &rank, &n)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure_stage(ctx, 7, ((size_t)n + 1) * 8)) return rc;
    uint64_t *d = static_cast<uint64_t *>(ctx->stage[7]);
    HIP_TRY(ctx, hipMemcpyAsync(d, &n_local, 8, hipMemcpyHostToDevice, ctx->stream));
    NCCL_TRY(ctx, rccl().AllGather(d, d + 1, 1, ncclUint64, static_cast<ncclComm_t>(nccl_comm), ctx->stream));
    std::vector<uint64_t> counts((size_t)n);
    HIP_TRY(ctx, hipMemcpyAsync(counts.data(), d + 1, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t before = 0, total = 0;
    for (int g = 0; g < n; ++g) {
        if (g < rank) before += counts[(size_t)g];
        total += counts[(size_t)g];
    }
    if (out_offset) *out_offset = before;
    if (out_total) *out_total = total;
    return KMERS_OK;
}

}  // extern "C"
