/*
 * kmers_hip.h -- C ABI of libkmers_hip.so, the MI355X (gfx950) implementation of the
 * k-mer iteration hot path of BioJulia/Kmers.jl v1.2.0.
 *
 * The reference has no FFI: its boundary is Julia's iteration protocol over
 * `FwKmers` / `FwRvIterator` / `CanonicalKmers` / `UnambiguousKmers` / `SpacedKmers`
 * plus `fx_hash`, `canonical`, `reverse_complement`.  Each entry point below is the
 * bulk form ("run iterate() to the end and collect") of one of those, with plain
 * pointers and sizes only; the Julia `@ccall` shim that binds them is shown in
 * INTEGRATION.md and julia/KmersHIP.jl.  file:line citations are relative to the
 * reference checkout.
 *
 * Data formats (bit-identical to the reference):
 *   input  = LongSequence.data :: Vector{UInt64}, symbol i (1-based) at bits
 *            [((i-1)*bps) mod 64, +bps) of word ((i-1)*bps) div 64   (bps = 2 or 4),
 *            or bytes (src_bits = 8, `words` then points at the bytes, one per symbol): ASCII text (String /
 *            Vector{UInt8} sources; AsciiEncode, src/construction.jl:94-95) or symbol values (Vector{DNA} and the
 *            like; GenericRecoding, :90-98) -- see kmers_seq.alphabet
 *   output = Vector{Kmer{A,K,N}} memory: N = cld(K*bps_dst, 64) UInt64 per element,
 *            data[1] first, first symbol in the most significant used bits, unused
 *            bits = top bits of data[1] = 0                        (src/kmer.jl:32-44)
 *
 * Errors never cross the ABI as exceptions: every call returns a status and fills
 * a kmers_result.  KMERS_E_ENCODE carries what the Julia shim needs to
 * `throw(BioSequences.EncodeError(A(), reinterpret(DNA, enc)))` exactly like
 * src/construction.jl:108-110: the 1-based position of the FIRST offending symbol
 * in sequence order and its raw source encoding (4-bit code, or the byte of an ASCII
 * source).  On E_ENCODE the output buffers
 * are unspecified (the bulk form cannot "yield some, then throw").
 *
 * There is NO CPU fallback in this library: without a usable HIP device every
 * compute entry point returns KMERS_E_HIP.
 */
#ifndef KMERS_HIP_H
#define KMERS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMERS_ABI_VERSION 1

/* status codes */
#define KMERS_OK 0
#define KMERS_E_ENCODE 1      /* BioSequences.EncodeError (src/construction.jl:108-110) */
#define KMERS_E_BADARG 2      /* error("K must be at least 1") etc. (FwKmers.jl:31-35, SpacedKmers.jl:26-32) */
#define KMERS_E_HIP 3         /* HIP runtime failure / no device */
#define KMERS_E_NOMEM 4
#define KMERS_E_UNSUPPORTED 5 /* geometry outside what the kernels cover (see kmers_supported) */
#define KMERS_E_CAPACITY 6    /* kmers_unambiguous / kmers_batch: output capacity too small (res->n_out = needed) */
#define KMERS_E_NCCL 7        /* an RCCL call failed (kmers_last_error has ncclGetErrorString) */

/* flags */
#define KMERS_MEM_HOST 0x0   /* sequence/output pointers are host memory (staged through HBM) */
#define KMERS_MEM_DEVICE 0x1 /* sequence/output pointers are device (HBM) memory              */
#define KMERS_ASYNC 0x2      /* device memory only: enqueue and return; collect status (kmers_unambiguous: and the count) with kmers_sync */
#define KMERS_OUT_TUPLES 0x4 /* array-of-structs output = the eltype of the tuple-yielding iterators, written to the
                              * FIRST output pointer (second must be NULL): kmers_fw -> Tuple{Kmer,Kmer} (fw, rc; eltype of
                              * FwRvIterator, CanonicalKmers.jl:44-45), kmers_canonical -> Tuple{Kmer,UInt64} (kmer, fx_hash),
                              * kmers_unambiguous -> Tuple{Kmer,Int} (kmer, start; UnambiguousKmers.jl:39-41) */

typedef struct kmers_ctx kmers_ctx; /* one per host thread / HIP stream; not thread-safe */

typedef struct {
    int32_t status;   /* KMERS_OK or KMERS_E_* */
    uint32_t err_enc; /* E_ENCODE: raw source encoding of the offending symbol */
    uint64_t err_pos; /* E_ENCODE: 1-based index (index_origin added) of the offending symbol */
    uint64_t n_out;   /* elements written (kmers_batch / kmers_minhash_batch on E_ENCODE: index of the failing record) */
} kmers_result;

/* A borrowed view of a LongSequence (never mutated, never retained past the call:
 * FwKmers.jl:28-30).  first_base lets a LongSubSeq or a halo shard start inside
 * words[0]; index_origin is added to every reported 1-based position (shards of one
 * long sequence report global positions). */
typedef struct {
    const uint64_t *words; /* LongSequence.data */
    uint64_t n_bases;      /* LongSequence.len */
    uint64_t first_base;   /* 0-based symbol offset of the view inside words[] */
    uint64_t index_origin; /* 0 for a whole sequence */
    int32_t src_bits;      /* 2 (DNA/RNAAlphabet{2}), 4 (DNA/RNAAlphabet{4}) or 8 (ASCII bytes: String, Vector{UInt8}) */
    int32_t alphabet;      /* byte sources only (src_bits = 8): KMERS_ALPHABET_DNA / _RNA = ASCII text for a DNA (T valid) / RNA
                            * (U valid) kmer alphabet; KMERS_ALPHABET_SYMBOLS = not text: one BioSymbols value per byte, the
                            * memory of a Vector{DNA} / Vector{RNA} -- the reference's GenericRecoding sources
                            * (src/construction.jl:90-98, FwKmers.jl:80-86, CanonicalKmers.jl:81-91): each symbol goes
                            * through BioSequences.encode of the kmer alphabet (2-bit: one-hot values only, anything else is
                            * the reference's EncodeError; 4-bit: every value below 16) */
} kmers_seq;
#define KMERS_ALPHABET_DNA 0
#define KMERS_ALPHABET_RNA 1
#define KMERS_ALPHABET_SYMBOLS 2

/* ---- library / context ------------------------------------------------------- */
int kmers_abi_version(void);
/* hip_stream: NULL -> the context creates and owns a stream; else a hipStream_t to borrow */
int kmers_ctx_create(int device, void *hip_stream, kmers_ctx **out);
void kmers_ctx_destroy(kmers_ctx *ctx);
void *kmers_ctx_stream(kmers_ctx *ctx);
const char *kmers_last_error(kmers_ctx *ctx);
/* Wait for KMERS_ASYNC work; reports the first EncodeError seen since the last sync.  The kernels record the offending
 * symbol itself next to its position (index_origin + position, 1-based) at fault time, so nothing is read back from a
 * sequence here: if several asynchronous launches (shards of one sequence, or unrelated sequences) ran since the last
 * sync, the error with the smallest reported position wins.  Sequence and output buffers of an asynchronous call must
 * stay valid until the work has run (this call, or later work on the same stream, has completed). */
int kmers_sync(kmers_ctx *ctx, kmers_result *res);

/* launch tunables (bench / tuning; 0 restores the default) */
#define KMERS_PARAM_TILE_KMERS 1 /* kmers per workgroup tile (multiple of 512) */
#define KMERS_PARAM_MAX_GRID 2   /* cap on workgroups per launch (persistent grid-stride above it) */
#define KMERS_PARAM_STAMPS_PTR 3 /* diagnostic builds (-DKMERS_STAMPS) only: device buffer for in-kernel stamps */
#define KMERS_PARAM_SKETCH_HOST_ONLY 4 /* kmers_minhash: 1 = use the host-feedback path even for small sketches (tests) */
#define KMERS_PARAM_BATCH_PASSES 5     /* kmers_batch: elements per workgroup tile = 1024 * value (1..16); 0 = chosen from the batch size */
#define KMERS_PARAM_SKETCH_BATCH_LDS 6 /* kmers_minhash_batch: candidate values per workgroup (2048 / 4096 / 8192); 0 = chosen per call */
#define KMERS_PARAM_SPLIT_ORDER 9      /* the tile kernels visit the two halves of their tile range alternately (two write windows per output
                                         * array).  0: when the launch has ONE output array whose halves lie in two region classes (a block
                                         * taken with kmers_dev_alloc_role(KMERS_ALLOC_LONE_OUTPUT)): +6 % there; 1: always; -1: never.
                                         * Results are identical either way. */
#define KMERS_PARAM_BLOCK_THREADS 10   /* threads per workgroup of the tile kernels: 64, 128 or 256 (0: chosen per output shape) */
#define KMERS_PARAM_WIDE_NO_TILES 11   /* A/B, tests.  1: kmers of more than four words always on the one-lane-per-kmer kernel; 2: the
                                        * run-time-width tile form also for kmers of one to four words (it loses there: profiles/r03_wide.md) */
#define KMERS_PARAM_HOST_CHUNKS 12      /* -1: a host-pointer call (KMERS_MEM_HOST) is one launch + one copy whatever its size; 0 (default): outputs of
                                        * 96 MiB or more travel in chunks, the kernel of the next chunk beside the copy of the current one */
#define KMERS_PARAM_POOL 14             /* 1 (default): kmers_dev_alloc of KMERS_POOL_MIN_BYTES or more comes from the device's class pool (below); 0: plain hipMalloc */
#define KMERS_PARAM_POOL_SEARCH_GIB 15  /* how far past a request the pool may grow in search of memory of the classes it wants (default: 128, at most half
                                         * of what the device has free; held for the SEARCH only: what it walked past goes back to the driver once the block
                                         * is made); 0: never */
#define KMERS_PARAM_POOL_CACHE 18       /* 1 (default): a freed block of the pool stays mapped for the next request of its shape; 0: taken apart at once (the
                                         * free then waits for the work queued on the device's contexts) */
#define KMERS_PARAM_BATCH_DENSE 17       /* A/B, tests.  -1: kmers_batch never takes its dense tile path (csrc/ragged_kernels.hpp); 0 (default): wherever a tile allows */
#define KMERS_PARAM_POOL_MAX_GIB 16     /* cap on the physical memory the pool holds (0, default: what the device has) */
#define KMERS_PARAM_SUBTILES 7         /* strided tile kernels (kmers_spaced, kmers_minimizers): consecutive tiles per workgroup, the next one's source words in flight */
int kmers_ctx_set_param(kmers_ctx *ctx, int param, int64_t value);
/* The launch shape the library chose for the most recent launch of its tile kernel in this context (kmers_fw / kmers_canonical /
 * kmers_spaced and the fused consumers that run on it): threads per workgroup, kmers per tile, and whether every output array
 * was written through two windows.  The table behind the choice (csrc/stream_launch.hpp) depends on the kmer width, the number
 * of output arrays and on where the pool says they lie; this is how a host, a test or bench.py sees what it came to, and
 * times the alternatives against it (KMERS_PARAM_TILE_KMERS / _BLOCK_THREADS / _SPLIT_ORDER).  All zero before any launch. */
int kmers_last_launch_shape(kmers_ctx *ctx, int *threads, int *tile_kmers, int *split_order);

/* ---- device memory ---------------------------------------------------------------------------
 * For hosts without a HIP binding of their own (the reference allocates its outputs itself: `collect` makes one Vector per
 * call, Base.collect over src/iterators/CanonicalKmers.jl:199-225).
 *
 * WHERE an array lies matters on MI355X: HBM behaves as three REGION CLASSES of physical memory; store streams that run side by
 * side inside one class share about 6.0-6.4 TB/s, streams in different classes reach 7.1-7.2 (profiles/r03_alloc.md,
 * profiles/r05_vmm.md).  A plain hipMalloc lies wherever the driver puts it -- usually inside one class.  So a block of
 * KMERS_POOL_MIN_BYTES or more comes from the device's CLASS POOL: physical memory in 1 GiB handles of HIP's virtual-memory
 * management, the class of each handle measured once when the pool takes it (about 1 ms of probes), each block made of handles
 * chosen by class: a block differs from the block allocated before it (the most recent one that is still out) at every relative
 * position -- allocate the arrays of one launch one after the other and they are written at the two-class rate;
 * KMERS_ALLOC_LONE_OUTPUT gives a block whose second half differs from its first, for launches with one output (written through
 * two windows half an array apart).  No reservation is needed and it does not matter how finely the classes are interleaved in a
 * box's physical memory.  Blocks are whole handles (sizes round up to 1 GiB; an array of 128 MiB to 1 GiB is one handle).
 *
 * What a block costs (profiles/r06_pool.md).  kmers_dev_free of a pool block does NOT wait and calls nothing in the driver: the
 * block stays mapped in the pool's cache, and the next kmers_dev_alloc of its shape (same number of handles, same role, classes
 * that suit its new partner) returns it as it is -- a host loop {alloc, alloc, launch, free, free} (what `collect` per sequence
 * amounts to) runs within a few microseconds per call of the same loop over resident arrays.  ORDER: the block's next user comes
 * after everything that was queued, at the time of the free, on the streams of the contexts of this device that use the pool (the
 * next user's stream waits for events recorded then; a user on the freeing context's own stream waits for nothing).  Work on any
 * OTHER stream that still touches the block (a host framework's own streams) must have finished before kmers_dev_free -- as with
 * any stream-ordered allocator.  A cached block nobody asked for over 32 pool calls is taken apart.
 *
 * What the pool holds: its blocks (out or cached) plus at most max(4 GiB, a quarter of that) of handles outside blocks -- one
 * yardstick handle per class it has found, and free handles; what it walked past in search of a class
 * (KMERS_PARAM_POOL_SEARCH_GIB) goes back to the driver as soon as the block is made.  kmers_pool_trim returns everything that is
 * not out; so does any allocation of the library that would otherwise fail.  One pool per device and process, shared by its
 * contexts, thread-safe.  Smaller blocks, a device without virtual-memory management, a pool that cannot get its handles (the
 * cap KMERS_PARAM_POOL_MAX_GIB, a device too full), or KMERS_PARAM_POOL = 0: plain hipMalloc (kmers_dev_free of those waits for
 * the context's stream and is a hipFree).  KMERS_E_NOMEM only when that fails too.  Alignment: what hipMalloc gives
 * (256 bytes at least) for plain blocks; a block of the pool to 4096 bytes at least (to 2 MiB unless it was asked for as a lone
 * output) and never to a GiB boundary -- on ROCm 7.2 a 1 GiB handle mapped on one faults now and then while handles come and
 * go, profiles/r06_pool.md section 5. */
int kmers_dev_alloc(kmers_ctx *ctx, size_t bytes, void **out);
/* The same with a word about what the block is for:
 *   KMERS_ALLOC_DEFAULT      anything: inputs, the arrays of a launch with two outputs (placed in different region classes)
 *   KMERS_ALLOC_LONE_OUTPUT  the ONLY output array of the launches that fill it (collect(CanonicalKmers) without hashes,
 *                            FwKmers, SpacedKmers, a tuple array): its second half in another class than its first, and a
 *                            launch that finds its one output there writes it through two windows (C3: 0.81 -> 0.87 of 8 TB/s). */
#define KMERS_ALLOC_DEFAULT 0
#define KMERS_ALLOC_LONE_OUTPUT 1
int kmers_dev_alloc_role(kmers_ctx *ctx, size_t bytes, int role, void **out);
int kmers_dev_free(kmers_ctx *ctx, void *p);
int kmers_memcpy_h2d(kmers_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int kmers_memcpy_d2h(kmers_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* For a chunk-buffered iterate() that computes chunk c + 1 while the host loops over chunk c (src/iterators/FwKmers.jl:57-66 is
 * the protocol; julia/KmersHIP.jl GPUIterator and kmers.jl_amd/host.py are the callers): page-locked host memory, and copies that
 * are only ENQUEUED on the context's stream, behind the launches before them -- complete after kmers_sync.  (From pageable memory
 * such a copy is staged by the runtime and blocks the caller: use kmers_host_alloc for the buffers.) */
int kmers_host_alloc(kmers_ctx *ctx, size_t bytes, void **out_host);
int kmers_host_free(kmers_ctx *ctx, void *p_host);
int kmers_memcpy_h2d_async(kmers_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int kmers_memcpy_d2h_async(kmers_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

#define KMERS_POOL_MIN_BYTES ((size_t)128 << 20)
#define KMERS_POOL_CLASSES 4
/* The device's class pool: physical bytes it holds, bytes in blocks that are out, region classes found so far, bytes held per
 * class (KMERS_POOL_CLASSES entries: classes 0..2, then memory the pool could not classify), and what its probes measured in GB/s of two store streams side by side (1 GiB each, the
 * shape of the stream kernels' outputs): the fastest probe (two classes; about 7200; 0 while only one class is known) and the
 * slowest pair of the calibration (one class; about 6200).  The first is the write ceiling bench.py prices the materialising kernels against.  Any
 * output may be NULL; all zero before the first block. */
int kmers_pool_info(kmers_ctx *ctx, size_t *held, size_t *in_use, int *n_classes, size_t *class_bytes, double *two_class_gbps, double *one_class_gbps);
/* Counters of the pool (at most `capacity` of the KMERS_POOL_STATS entries are written): [0] bytes held, [1] bytes in blocks that
 * are out, [2] bytes in cached (freed, still mapped) blocks, [3] bytes in free handles, [4] allocations served from the cache,
 * [5] allocations that assembled a block, [6] cached blocks taken apart, [7] handles created, [8] handles returned to the driver,
 * [9] class probes run, [10] cached blocks, [11] blocks out. */
#define KMERS_POOL_STATS 12
int kmers_pool_stats(kmers_ctx *ctx, uint64_t *out, size_t capacity);
/* Return to the driver every handle of the pool that is not part of a block that is out: the cache of freed blocks, the free
 * handles (and the yardsticks too, if no block is out). */
int kmers_pool_trim(kmers_ctx *ctx, size_t *released);
/* The handles of the pool block that holds `block`: *n_chunks chunks of *chunk_bytes, chunk i in class classes[i]
 * (0..2, or KMERS_POOL_CLASSES - 1 = not classified; at most `capacity` entries are written).  *n_chunks = 0: not a block of the pool. */
int kmers_pool_layout(kmers_ctx *ctx, const void *block, size_t *chunk_bytes, unsigned char *classes, size_t capacity, size_t *n_chunks);
/* The virtual-memory calls the pool relies on, checked on this box (csrc/pool_api.hip): KMERS_OK, or KMERS_E_HIP with the reason in
 * kmers_last_error.  *stale_without_flush = 1: a re-used address range showed the memory of its PREVIOUS mapping until the pool's
 * flush ran -- the behaviour of this stack (ROCm 7.2) that every unmap of the pool guards against. */
int kmers_pool_selftest(kmers_ctx *ctx, int *stale_without_flush);
/* The pool's own measurement for any two device buffers: the write rate, in GB/s, of two store streams side by side into them
 * (the shape of the stream kernels' outputs).  DESTRUCTIVE: the first min(bytes, 2 GiB) of both buffers are overwritten.  About
 * 7000 = the buffers lie in different region classes, about 6000 = in one. */
int kmers_placement_probe(kmers_ctx *ctx, void *a_dev, void *b_dev, size_t bytes, double *gbps);

/* ---- geometry (src/kmer.jl:117-137; iterator length()) ----------------------- */
int kmers_words_per_kmer(int k, int dst_bits);                      /* n_coding_elements, kmer.jl:123-125 */
uint64_t kmers_count(uint64_t n_bases, int k, int stride);          /* FwKmers.jl:40-43; SpacedKmers.jl:38-42 */
/* 1 if kmers_fw / kmers_canonical / kmers_spaced cover the geometry: every K >= 1 (Kmer{A,K,N} has no bound on N,
 * src/kmer.jl:97-111; kmers of one to four words run on the tile kernels, wider ones on a run-time-width kernel).
 * kmers_unambiguous (and the fused reducer over it) takes K <= 30720 (its one-pass kernel stages a tile together with its
 * K-1 symbols of overlap).  The fused consumers, the element-wise operations and the batch entry points take every width
 * as well (run-time-width kernels beyond four words); what stays refused (KMERS_E_UNSUPPORTED) is a resource bound, not a
 * width: kmers_composition above K = 16 (4^K counters), sketches per record above 2048 values, 2^32 records per batch. */
int kmers_supported(int src_bits, int dst_bits, int k, int stride);

/* ---- iterators --------------------------------------------------------------- */
/* FwKmers{A,K}(seq) collected (FwKmers.jl:57-115); with out_rc != NULL also
 * reverse_complement of every element = the (fw, rv) pairs of FwRvIterator
 * (CanonicalKmers.jl:54-144) as two arrays.  n = kmers_count(n_bases, k, 1). */
int kmers_fw(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, uint64_t *out_fw,
             uint64_t *out_rc, int flags, kmers_result *res);

/* CanonicalKmers{A,K}(seq) collected (CanonicalKmers.jl:199-225: fw < rv ? fw : rv);
 * with out_hashes != NULL also fx_hash(kmer, seed) of every element (kmer.jl:255-261).
 * Either output may be NULL. */
int kmers_canonical(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits,
                    uint64_t *out_kmers, uint64_t *out_hashes, uint64_t seed, int flags,
                    kmers_result *res);

/* SpacedKmers{A,K,J}(seq) collected (SpacedKmers.jl:83-139), strict semantics: an
 * ambiguous symbol inside any inspected position is E_ENCODE; with J >= K the gaps
 * are never inspected (SpacedKmers.jl:133-134). */
int kmers_spaced(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, int dst_bits,
                 uint64_t *out_kmers, int flags, kmers_result *res);

/* UnambiguousKmers{A,K}(seq) collected (UnambiguousKmers.jl:59-148): every window of
 * K unambiguous symbols with its 1-based start.  Output count is data dependent
 * (SizeUnknown, :33): capacity = elements the buffers can hold; res->n_out = count.
 * stride > 1 keeps only windows whose 0-based start is a multiple of stride (the
 * "Spaced with ambiguous-base skip" composition of BASELINE.json config 5,
 * docs/src/faq.md:28-33); stride = 1 is the reference iterator.  Sources: 2-bit and
 * 4-bit sequences (in a 4-bit sequence the gap is skipped like an ambiguity code,
 * :134-148), text (ASCII_SKIPPING_LUT: ambiguity letters are skipped, any other byte
 * is E_ENCODE, :109-132) and collections of symbols (KMERS_ALPHABET_SYMBOLS, the
 * generic method :88-106: ambiguous symbols are skipped, the gap is E_ENCODE).
 * KMERS_MEM_DEVICE | KMERS_ASYNC enqueues the one pass and returns: the count is not known yet (res->n_out = 0, except for a
 * 2-bit source, where nothing can be dropped); the next kmers_sync reports it in its res->n_out, with KMERS_E_CAPACITY if it
 * exceeds `capacity` (nothing was stored at or beyond it).  If several asynchronous calls ran since the last sync, the sync
 * reports the count of the LAST one. */
int kmers_unambiguous(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride,
                      uint64_t *out_kmers, int64_t *out_starts, uint64_t capacity, int flags,
                      kmers_result *res);

/* Fused consumer of test/benchmark.jl:9-15: XOR of data[1] over all (canonical or
 * forward) kmers, nothing materialised.  *out_value is host memory. */
int kmers_reduce_xor(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, int canonical,
                     uint64_t *out_value, int flags, kmers_result *res);
/* The same reducer over the other iterators of test/benchmark.jl:35-94 (`y ⊻= kmer.data[1]`, for
 * UnambiguousKmers `first(x).data[1]`): iter = KMERS_ITER_FW / _CANONICAL (stride ignored),
 * KMERS_ITER_SPACED (SpacedKmers{A,K,stride}, strict; stride * dst_bits <= 64) or
 * KMERS_ITER_UNAMBIGUOUS (2-bit kmers; stride = 1 is the reference iterator). */
#define KMERS_ITER_FW 0
#define KMERS_ITER_CANONICAL 1
#define KMERS_ITER_SPACED 2
#define KMERS_ITER_UNAMBIGUOUS 3
int kmers_reduce_xor_iter(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, int iter, int stride,
                          uint64_t *out_value, int flags, kmers_result *res);

/* Fused consumer of docs/src/minhash.md:31-35, `sketch(fx_hash, CanonicalKmers{A,K}(seq), s)`:
 * the s smallest DISTINCT values of fx_hash(canonical kmer, seed), ascending, written to
 * out_hashes (HOST memory, room for s values); res->n_out = values written (< s when the
 * sequence has fewer distinct hashes).  No per-kmer output is materialised.  (MinHash.jl is an
 * absent third-party package; "bottom-s distinct, sorted" is its published sketch definition.) */
int kmers_minhash(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, uint64_t seed, uint64_t s,
                  uint64_t *out_hashes, int flags, kmers_result *res);

/* Minimizers, the kmer replacement the reference builds from its public primitives in
 * docs/src/replacements.md:33-51 and test/benchmark.jl:96-110: element j is the kmer with the
 * smallest fx_hash among the W consecutive kmers starting at symbol 1 + j*stride (windows that do
 * not fit are not produced).  mode 0 reproduces the published `unsafe_extract_minimizer`
 * literally (each new symbol is shifted into the current minimum), mode 1 is the true
 * sliding-window minimum (leftmost on ties).  out_kmers: n * N words. */
int kmers_minimizers(kmers_ctx *ctx, const kmers_seq *seq, int k, int w, int stride, int dst_bits, int mode,
                     uint64_t *out_kmers, int flags, kmers_result *res);

/* Fused consumer of docs/src/composition.md:28-39: out_counts[as_integer(kmer)] += 1 for every
 * kmer of FwKmers{DNA/RNAAlphabet{2},K}(seq); out_counts has 4^K uint32 entries (K <= 16: 16 GiB of counters,
 * which a host array stages through HBM; K <= 10 counts in LDS, above that with memory-side atomics). */
int kmers_composition(kmers_ctx *ctx, const kmers_seq *seq, int k, uint32_t *out_counts, int flags,
                      kmers_result *res);

/* ---- element-wise operations on arrays of kmers ------------------------------ */
/* fx_hash(x::Kmer, h::UInt) (kmer.jl:255-261) over n kmers of n_words words each */
int kmers_fx_hash(kmers_ctx *ctx, const uint64_t *kmers, int n_words, uint64_t n, uint64_t seed,
                  uint64_t *out_hashes, int flags);

#define KMERS_OP_REVERSE 0     /* transformations.jl:1-10  */
#define KMERS_OP_COMPLEMENT 1  /* transformations.jl:14-25 */
#define KMERS_OP_REVCOMP 2     /* transformations.jl:32-34 */
#define KMERS_OP_CANONICAL 3   /* transformations.jl:36-39 */
#define KMERS_OP_ISCANONICAL 4 /* transformations.jl:41 ; out = one uint64 0/1 per kmer */
#define KMERS_OP_TO_LONGSEQ 5  /* LongSequence{A}(kmer).data, construction.jl:289-324; out = N words per kmer */
#define KMERS_OP_COUNT_GC 6    /* count(isGC, kmer), counting.jl:1-8 (2-bit); out = one uint64 per kmer */
#define KMERS_OP_AS_INTEGER 7   /* as_integer, kmer.jl:305-326: u64 (<= 64 coding bits) or little-endian u128 per kmer */
#define KMERS_OP_FROM_INTEGER 8 /* from_integer, kmer.jl:361-384: the inverse; only the lowest K*bits bits are used */
int kmers_transform(kmers_ctx *ctx, int op, const uint64_t *kmers, int k, int bits, uint64_t n,
                    uint64_t *out, int flags);

/* ---- batches of records (reads, contigs, FASTA records) ------------------------------------ */
/* The reference iterates one sequence at a time (`for record in reader ... CanonicalDNAMers{K}(seq)`,
 * docs/src/minhash.md:31-35); a GPU call has a fixed cost of ~16 us, so short records go many at a
 * time.  `pool` holds the symbols of all records (any layout: concatenated LongSequence data words,
 * a FASTA buffer, ...); record i is the view [spans[i].first_base, +n_bases) of the pool, exactly a
 * LongSubSeq.  One call produces, concatenated in record order, what the per-record iterator
 * would: mode KMERS_BATCH_FW = FwKmers (out_a) and their reverse complements (out_b, nullable:
 * FwRvIterator); mode KMERS_BATCH_CANONICAL = CanonicalKmers (out_a) and fx_hash(kmer, seed)
 * (out_b, nullable).  out_offsets (host memory, n_spans + 1 entries, nullable) receives the element
 * offset of every record (records shorter than k yield nothing and are never inspected,
 * FwKmers.jl:63).  capacity = elements out_a / out_b can hold; if the batch needs more the call
 * returns KMERS_E_CAPACITY with res->n_out = the number required (capacity 0 + NULL outputs = a size
 * query).  KMERS_MEM_DEVICE applies to pool->words, out_a and out_b; out_offsets is always host memory;
 * spans is host memory unless KMERS_SPANS_DEVICE is set (tens of millions of reads: keep them
 * resident).  Kmers of any width (more than four words: one lane per element, not tuned).
 * EncodeError: the first failing record in batch order wins, res->err_pos = 1-based position inside
 * THAT record, res->err_enc = the raw symbol, res->n_out = the record's index in spans[]. */
typedef struct {
    uint64_t first_base; /* 0-based symbol offset of the record inside the pool view */
    uint64_t n_bases;
} kmers_span;
#define KMERS_SPANS_DEVICE 8 /* flag of kmers_batch: `spans` points to HBM */
#define KMERS_BATCH_SKIP 16  /* flag of kmers_batch: an element whose window holds a symbol the kmer alphabet cannot encode
                              * (N in a read, say) does not fail the call; it is written as all-ones in every output
                              * array -- never a canonical kmer, and never a (kmer, reverse complement) pair -- so that
                              * the kept elements are those of UnambiguousKmers{A,K}(record) at the same indices (the
                              * composition docs/src/faq.md:28-33 describes); element offsets stay those of the strict call.
                              * That equivalence holds for 2- and 4-bit pools.  An 8-bit (ASCII) pool is read through the kmer
                              * alphabet's ascii_encode table, like FwKmers over a String: every byte outside it is marked --
                              * including U for DNA kmers, T for RNA kmers and non-nucleotide bytes, which UnambiguousKmers
                              * would keep as 3 / throw on (its ASCII_SKIPPING_LUT, src/iterators/common.jl:22-32) */
#define KMERS_BATCH_FW 0
#define KMERS_BATCH_CANONICAL 1
int kmers_batch(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int mode, int k,
                int dst_bits, uint64_t *out_a, uint64_t *out_b, uint64_t seed, uint64_t *out_offsets,
                uint64_t capacity, int flags, kmers_result *res);

/* SpacedKmers{A,K,J}(record) for every record of a batch (each_codon over the coding sequences of a genome: K = J = 3;
 * src/iterators/SpacedKmers.jl:23-42,77-139): record i yields (n_bases - k) / stride + 1 forward kmers, at symbols 0, J, 2J, ...
 * of the record (none if it is shorter than k).  Strict like the reference's iterator: only the symbols inside a window are
 * inspected (J > K leaves gaps that may hold anything, test/runtests.jl:866), and a window over a symbol the kmer alphabet
 * cannot encode fails the call (EncodeError reported as by kmers_batch) unless KMERS_BATCH_SKIP marks it instead.  Everything
 * else -- pool, spans, offsets, capacity / size query, flags -- as kmers_batch. */
int kmers_batch_spaced(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int k, uint64_t stride,
                       int dst_bits, uint64_t *out_kmers, uint64_t *out_offsets, uint64_t capacity, int flags, kmers_result *res);

/* One MinHash sketch per record (MinHash.jl is used on collections: one sketch per genome / FASTA record,
 * docs/src/minhash.md:31-41): record i of the batch gets the s smallest distinct values of
 * fx_hash(canonical kmer, seed) over CanonicalKmers{A,K}(record i), ascending, in
 * out_hashes[i * s .. i * s + out_counts[i]) (out_counts[i] <= s; fewer when the record has fewer distinct
 * kmers).  pool / spans / flags as for kmers_batch (KMERS_MEM_DEVICE covers pool->words, out_hashes and
 * out_counts); s <= 2048.  EncodeError: as kmers_batch (res->n_out = the failing record); with KMERS_BATCH_SKIP
 * windows over symbols that cannot be encoded are left out of the sketches instead.  A pool of 128 MiB and more in
 * HOST memory whose records lie in pool order (first_base ascending: the records of a FASTA file) is brought up, recoded and
 * sketched in pieces of 32 MiB, the copy of one piece beside the kernels of the piece before: the call takes about as long as the
 * copy of its pool alone (0.92-0.94 of it from pinned memory, bench.py's e2e legs). */
int kmers_minhash_batch(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int k,
                        int dst_bits, uint64_t seed, uint64_t s, uint64_t *out_hashes, uint64_t *out_counts, int flags,
                        kmers_result *res);
/* In how many pieces the most recent kmers_minhash_batch of this context brought its pool up (1: one copy, one launch). */
int kmers_last_batch_pieces(kmers_ctx *ctx, uint64_t *pieces);

/* ---- sharding one long sequence over the GPUs of a node ------- */
/* The reference has no distributed code; kmer i depends only on symbols [i*stride, i*stride + k),
 * so shard g owns a contiguous range of kmers (in iteration order) whose first symbol sits on a
 * source-word boundary and on the stride lattice, and needs the first halo_words words of shard
 * g+1 appended to its own words: one neighbour step, the only communication on the path.
 * Outputs stay with the shard (concatenation in shard order == the reference's iteration order).
 * Run a shard with kmers_seq{words = own words + halo, n_bases, first_base = 0,
 * index_origin = first_base}; UnambiguousKmers shards additionally exchange their element
 * counts (exclusive scan) to place outputs, and the first EncodeError is the minimum err_pos
 * over shards. */
typedef struct {
    uint64_t first_kmer;  /* global 0-based ordinal of the first kmer owned */
    uint64_t n_kmers;     /* kmers owned */
    uint64_t first_base;  /* global 0-based symbol index of the view (= first_kmer * stride; word aligned) */
    uint64_t n_bases;     /* symbols in the view: (n_kmers - 1) * stride + k, 0 if the shard is empty */
    uint64_t first_word;  /* global index of the first source word owned */
    uint64_t n_own_words; /* source words owned (the last shard keeps the tail) */
    uint32_t halo_words;  /* words to receive from shard g+1 */
    uint32_t send_words;  /* words to send to shard g-1 (its halo_words) */
} kmers_shard;
/* Pure host arithmetic (no context, no GPU).  src_bits 2, 4 or 8.  Sequences too short to give
 * every shard a halo's worth of words are handled whole by shard 0 (the rest are empty). */
int kmers_shard_plan(uint64_t n_bases, int k, uint64_t stride, int src_bits, int n_shards, int shard_id,
                     kmers_shard *out);

/* ---- the communication of the sharded path, on RCCL -----------------
 * The reference has no distributed code; what makes its iterators shardable is that iterate()
 * carries nothing from one kmer to the next but the previous K-1 symbols (FwKmers.jl:57-66,
 * CanonicalKmers.jl:94-105).  Three exchanges complete the semantics across shards, all on the
 * context's stream through an ncclComm_t (RCCL: xGMI between the GPUs of a node):
 *   kmers_halo_exchange          the first halo_words words of shard g+1 -> the end of shard g's words
 *                                (grouped ncclSend / ncclRecv between neighbours, <= 32 B per pair)
 *   kmers_first_error_allreduce  the reference throws at the FIRST offending symbol in sequence
 *                                order (FwKmers.jl:112, CanonicalKmers.jl:139): ncclAllReduce(min)
 *   kmers_offsets_allgather      UnambiguousKmers is SizeUnknown (UnambiguousKmers.jl:33): shard
 *                                element counts -> this shard's offset in the global output
 * `nccl_comm` is an ncclComm_t: one made by kmers_comm_create, or any communicator of the host
 * program's own RCCL binding whose rank r runs shard r (one rank per GPU, rank order = shard order). */
#define KMERS_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* ncclGetUniqueId: ONE rank calls it and the host program hands the bytes to the others over any
 * channel it has (MPI, Distributed.jl, a torch.distributed store, a file). */
int kmers_comm_id(void *out_id);
/* ncclCommInitRank on the context's device; collective over the n_ranks callers. */
int kmers_comm_create(kmers_ctx *ctx, const void *id, int n_ranks, int rank, void **out_nccl_comm);
int kmers_comm_destroy(kmers_ctx *ctx, void *nccl_comm);
int kmers_comm_rank(kmers_ctx *ctx, void *nccl_comm, int *out_rank, int *out_n_ranks);
/* One grouped ncclSend + ncclRecv of 64-bit words on the context's stream (enqueue only; ordered
 * with the context's kernels).  A peer < 0 or a zero count skips that half; peer == own rank is a
 * device copy through RCCL.  The building block of kmers_halo_exchange. */
int kmers_comm_sendrecv(kmers_ctx *ctx, void *nccl_comm, const uint64_t *send_dev, uint64_t send_words, int send_peer,
                        uint64_t *recv_dev, uint64_t recv_words, int recv_peer);
/* words_dev = this shard's buffer in HBM: n_own_words own words followed by room for halo_words.
 * Sends words_dev[0, send_words) to rank - 1 and receives words_dev[n_own_words, +halo_words) from
 * rank + 1 (shard = kmers_shard_plan(..., n_ranks, rank)).  Enqueue only: the next kernel of this
 * context sees the halo.  Every rank of the communicator must call it (ranks with nothing to send
 * or receive enqueue nothing). */
int kmers_halo_exchange(kmers_ctx *ctx, void *nccl_comm, const kmers_shard *shard, uint64_t *words_dev);
/* In: this shard's result (status KMERS_OK or KMERS_E_ENCODE with GLOBAL err_pos, i.e. the shard ran
 * with index_origin = shard.first_base).  Out, identical on every rank: the first EncodeError of the
 * whole sequence (smallest err_pos) or KMERS_OK.  n_out is left alone.  Synchronous. */
int kmers_first_error_allreduce(kmers_ctx *ctx, void *nccl_comm, kmers_result *res);
/* Exclusive scan of the shards' element counts: *out_offset = elements of shards 0..rank-1,
 * *out_total = elements of all shards.  Synchronous. */
int kmers_offsets_allgather(kmers_ctx *ctx, void *nccl_comm, uint64_t n_local, uint64_t *out_offset, uint64_t *out_total);

/* ---- synthetic input (bench / tests) -------------------------- */
/* Fills out_dev (DEVICE memory) with words [first_word, first_word + n_words) of the
 * seeded uniform A/C/G/T sequence in LongSequence layout; bits = 2 or 4.  With
 * ambig_per_65536 > 0 (4-bit only) each base becomes N with that probability. */
int kmers_synth_dna(kmers_ctx *ctx, uint64_t seed, uint64_t first_word, uint64_t n_words, int bits,
                    uint32_t ambig_per_65536, uint64_t *out_dev);

#ifdef __cplusplus
}
#endif
#endif /* KMERS_HIP_H */
