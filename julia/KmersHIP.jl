# KmersHIP.jl -- the Julia side of the drop-in boundary: thin `@ccall` bindings of
# include/kmers_hip.h plus the methods that route Kmers.jl's own iterator API to them.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: neither `julia` nor BioSequences.jl exist in the build
# image (see DESIGN.md).  It is the binding a Kmers.jl maintainer would add (INTEGRATION.md); the same mapping is exercised
# through ctypes by kmers.jl_amd/host.py and tests/.
#
# DISPATCH POLICY (round 5).  The GPU path is EXPLICIT: `KmersHIP.gpu_collect(it)` always runs on the device.  `Base.collect`
# of Kmers.jl's iterator types is overloaded too, but GATED: a source shorter than `KmersHIP.MIN_BASES[]` symbols (default
# 100 000; ENV["KMERS_HIP_MIN_BASES"]; `typemax(Int)` switches the overload off) keeps Kmers.jl's own method, as does every
# geometry the library does not take (`invoke(collect, Tuple{Any}, it)`).  A call costs ~16-25 us plus two PCIe hops; the reference
# iterates at ~1 ns per symbol (docs/src/kmers.md:133), so `collect(FwDNAMers{3}("AGCGTATA"))` (src/iterators/FwKmers.jl:14-22)
# must not become a device call: the measured crossover is 25-40 k symbols.
#
# Usage:
#     using Kmers, BioSequences, KmersHIP
#     seq  = randdnaseq(10^9)                               # LongDNA{4}
#     v    = collect(CanonicalDNAMers{31}(seq))             # Vector{DNAKmer{31,1}}, computed on the MI355X (length(seq) >= MIN_BASES[])
#     v    = KmersHIP.gpu_collect(CanonicalDNAMers{31}(seq)) # the same, whatever the length
#     v, h = KmersHIP.collect_with_hashes(CanonicalDNAMers{31}(seq))   # + fx_hash of every element
#     for kmer in KmersHIP.gpu(CanonicalDNAMers{31}(seq)) ... end      # chunk-buffered iterate(): all five iterator types; chunk c + 1 is
#                                                                     # computed and copied while the loop consumes chunk c
#     collect(FwDNAMers{21}(view(seq, 2:length(seq))))                  # a LongSubSeq source: kmers_seq.first_base
module KmersHIP

using Kmers
using BioSequences

const LIB = get(ENV, "KMERS_HIP_LIB",
                joinpath(@__DIR__, "..", "kmers.jl_amd", "csrc", "libkmers_hip.so"))

# ---- mirror of the C structs (include/kmers_hip.h) ------------------------------------------
struct CSeq
    words::Ptr{UInt64}
    n_bases::UInt64
    first_base::UInt64
    index_origin::UInt64
    src_bits::Int32
    alphabet::Int32
end

mutable struct CResult
    status::Int32
    err_enc::UInt32
    err_pos::UInt64
    n_out::UInt64
    CResult() = new(0, 0, 0, 0)
end

const OK, E_ENCODE, E_BADARG, E_HIP, E_NOMEM, E_UNSUPPORTED, E_CAPACITY, E_NCCL = Int32.(0:7)
const ALPHABET_DNA, ALPHABET_RNA, ALPHABET_SYMBOLS = Int32(0), Int32(1), Int32(2)
const MEM_HOST, MEM_DEVICE, ASYNC, OUT_TUPLES = Int32(0), Int32(1), Int32(2), Int32(4)

# ---- context (one per Julia thread) --------------------------------------------------------
mutable struct Context
    handle::Ptr{Cvoid}
    function Context(device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = @ccall LIB.kmers_ctx_create(device::Cint, C_NULL::Ptr{Cvoid}, h::Ptr{Ptr{Cvoid}})::Cint
        rc == OK || error("kmers_ctx_create failed ($rc): no usable MI355X / HIP device")
        ctx = new(h[])
        finalizer(c -> (@ccall LIB.kmers_ctx_destroy(c.handle::Ptr{Cvoid})::Cvoid), ctx)
        return ctx
    end
end

# One context (= one HIP stream) per TASK: tasks migrate between threads, so a table keyed by threadid() would hand one
# context to two running tasks; task-local storage cannot.  (A kmers_ctx is not thread-safe; any number of them may run
# concurrently.)
context() = get!(() -> Context(0), task_local_storage(), :kmers_hip_context)::Context
last_error(ctx::Context) = unsafe_string(@ccall LIB.kmers_last_error(ctx.handle::Ptr{Cvoid})::Cstring)

# ---- helpers -------------------------------------------------------------------------------
const NucAlphabet24 = Union{DNAAlphabet{2}, DNAAlphabet{4}, RNAAlphabet{2}, RNAAlphabet{4}}
const NucSeq24 = Union{LongSequence{<:NucAlphabet24}, LongSubSeq{<:NucAlphabet24}}   # RecodingScheme takes any BioSequence (src/construction.jl:75-100); views: test/runtests.jl:162
const ByteSource = Union{String, SubString{String}, Vector{UInt8}, Base.CodeUnits{UInt8, String}}
const SymbolVector = Union{Vector{DNA}, Vector{RNA}}   # GenericRecoding sources whose memory is one BioSymbols value per byte
const Source = Union{NucSeq24, ByteSource, SymbolVector}
const TwoBitAlphabet = Union{DNAAlphabet{2}, RNAAlphabet{2}}

dst_bits(::Type{A}) where {A} = Int32(BioSequences.bits_per_symbol(A()))
isrna(::Type{A}) where {A} = Int32(A <: RNAAlphabet)

# kmers_seq of a source for kmer alphabet A (the ASCII validity table is A's: construction.jl:94-95)
cseq(s::LongSequence, ::Type{A}) where {A} =
    CSeq(pointer(s.data), length(s) % UInt64, 0, 0, Int32(BioSequences.bits_per_symbol(Alphabet(s))), 0)
# a view of a LongSequence (BioSequences' LongSubSeq: the parent's `data` and the range `part`): the same words, the view's first
# symbol `first(part) - 1` symbols in (kmers_seq.first_base, include/kmers_hip.h:75-90); positions are reported relative to the view
cseq(s::LongSubSeq, ::Type{A}) where {A} =
    CSeq(pointer(s.data), length(s) % UInt64, (first(s.part) - 1) % UInt64, 0, Int32(BioSequences.bits_per_symbol(Alphabet(s))), 0)
cseq(s::ByteSource, ::Type{A}) where {A} =
    CSeq(Ptr{UInt64}(pointer(s)), ncodeunits_or_length(s) % UInt64, 0, 0, Int32(8), isrna(A))
# a Vector{DNA} / Vector{RNA}: GenericRecoding in the reference (src/construction.jl:90-98); its memory is what
# KMERS_ALPHABET_SYMBOLS reads.  (Other generic sources keep Kmers.jl's own path: they are not `Source`s here.)
cseq(s::SymbolVector, ::Type{A}) where {A} =
    CSeq(Ptr{UInt64}(pointer(s)), length(s) % UInt64, 0, 0, Int32(8), ALPHABET_SYMBOLS)
ncodeunits_or_length(s::AbstractString) = ncodeunits(s)
ncodeunits_or_length(s) = length(s)

# Reproduce the reference's exception (src/construction.jl:108-110) from the C result.
function check(ctx::Context, rc::Integer, res::CResult, ::Type{A}, s) where {A}
    rc == OK && return nothing
    if rc == E_ENCODE
        # LongSequence sources: the offending symbol; byte sources: repr(byte) (FwKmers.jl:124-126)
        sym = (s isa BioSequence || s isa SymbolVector) ? reinterpret(eltype(s), res.err_enc % UInt8) : repr(res.err_enc % UInt8)
        throw(BioSequences.EncodeError(A(), sym))
    end
    error("libkmers_hip: status $rc: $(last_error(ctx))")
end

# ---- dispatch policy -----------------------------------------------------------------------
"Sources shorter than this many symbols keep Kmers.jl's own `collect` (header comment: DISPATCH POLICY)."
const MIN_BASES = Ref{Int}(parse(Int, get(ENV, "KMERS_HIP_MIN_BASES", "100000")))
source(it::Union{FwKmers, FwRvIterator, SpacedKmers}) = it.seq
source(it::Union{CanonicalKmers, UnambiguousKmers}) = it.it.seq
"true iff `Base.collect(it)` goes to the device"
gpu_dispatch(it) = ncodeunits_or_length(source(it)) >= MIN_BASES[]

const GpuIterator{A, S} = Union{FwKmers{A, <:Any, S}, FwRvIterator{A, <:Any, S}, CanonicalKmers{A, <:Any, S},
                                SpacedKmers{A, <:Any, <:Any, S}, UnambiguousKmers{A, <:Any, S}}
"`collect` of Kmers.jl's iterators: on the device from `MIN_BASES[]` symbols on, Kmers.jl's own method below"
Base.collect(it::GpuIterator{A, S}) where {A <: NucAlphabet24, S <: Source} =
    gpu_dispatch(it) && applicable(gpu_collect, it) ? gpu_collect(it) : invoke(collect, Tuple{Any}, it)

# ---- bulk forms of the iterators -----------------------------------------------------------
"collect(FwKmers{A,K}(seq)) on the GPU (src/iterators/FwKmers.jl:57-115)"
function gpu_collect(it::FwKmers{A, K, S}) where {A <: NucAlphabet24, K, S <: Source}
    ctx, s = context(), it.seq
    out = Vector{eltype(it)}(undef, length(it))
    res = CResult()
    GC.@preserve s out begin
        rc = @ccall LIB.kmers_fw(ctx.handle::Ptr{Cvoid}, Ref(cseq(s, A))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                                 pointer(out)::Ptr{UInt64}, C_NULL::Ptr{UInt64}, MEM_HOST::Cint,
                                 res::Ref{CResult})::Cint
    end
    check(ctx, rc, res, A, s)
    return out
end

"collect(FwRvIterator{A,K}(seq)): (forward, reverse_complement) pairs (CanonicalKmers.jl:54-144).
The library writes `Vector{Tuple{T,T}}` memory directly (KMERS_OUT_TUPLES)."
function gpu_collect(it::FwRvIterator{A, K, S}) where {A <: NucAlphabet24, K, S <: Source}
    ctx, s = context(), it.seq
    T = Kmers.derive_type(Kmer{A, K})
    out = Vector{Tuple{T, T}}(undef, length(it))
    res = CResult()
    GC.@preserve s out begin
        rc = @ccall LIB.kmers_fw(ctx.handle::Ptr{Cvoid}, Ref(cseq(s, A))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                                 pointer(out)::Ptr{UInt64}, C_NULL::Ptr{UInt64}, (MEM_HOST | OUT_TUPLES)::Cint,
                                 res::Ref{CResult})::Cint
    end
    check(ctx, rc, res, A, s)
    return out
end

"collect(CanonicalKmers{A,K}(seq)) (CanonicalKmers.jl:199-225)"
function gpu_collect(it::CanonicalKmers{A, K, S}) where {A <: NucAlphabet24, K, S <: Source}
    return first(collect_with_hashes(it; hashes = false))
end

"Canonical kmers and fx_hash.(kmers, seed) from one fused kernel (kmer.jl:255-261)."
function collect_with_hashes(it::CanonicalKmers{A, K, S}; seed::UInt = zero(UInt),
                             hashes::Bool = true) where {A <: NucAlphabet24, K, S <: Source}
    ctx, s = context(), it.it.seq
    n = length(it)
    out = Vector{eltype(it)}(undef, n)
    h = hashes ? Vector{UInt64}(undef, n) : UInt64[]
    res = CResult()
    GC.@preserve s out h begin
        rc = @ccall LIB.kmers_canonical(ctx.handle::Ptr{Cvoid}, Ref(cseq(s, A))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                                        pointer(out)::Ptr{UInt64},
                                        (hashes ? pointer(h) : Ptr{UInt64}(C_NULL))::Ptr{UInt64},
                                        seed::UInt64, MEM_HOST::Cint, res::Ref{CResult})::Cint
    end
    check(ctx, rc, res, A, s)
    return out, h
end

"collect(SpacedKmers{A,K,J}(seq)) (SpacedKmers.jl:83-139), strict semantics incl. EncodeError"
function gpu_collect(it::SpacedKmers{A, K, J, S}) where {A <: NucAlphabet24, K, J, S <: Source}
    ctx, s = context(), it.seq
    out = Vector{eltype(it)}(undef, length(it))
    res = CResult()
    GC.@preserve s out begin
        rc = @ccall LIB.kmers_spaced(ctx.handle::Ptr{Cvoid}, Ref(cseq(s, A))::Ptr{CSeq}, K::Cint, J::Cint, dst_bits(A)::Cint,
                                     pointer(out)::Ptr{UInt64}, MEM_HOST::Cint, res::Ref{CResult})::Cint
    end
    check(ctx, rc, res, A, s)
    return out
end

"collect(UnambiguousKmers{A,K}(seq)): (kmer, start) tuples (UnambiguousKmers.jl:59-148)"
function gpu_collect(it::UnambiguousKmers{A, K, S}) where {A <: TwoBitAlphabet, K, S <: Source}
    # a geometry this entry point does not take keeps Kmers.jl's own path (K above 30720).  A Vector{DNA} source is the
    # reference's generic method (UnambiguousKmers.jl:88-106): ambiguous symbols skipped, the gap an EncodeError.
    K > 30720 && return invoke(collect, Tuple{Any}, it)
    ctx, s = context(), it.it.seq
    T = Kmers.derive_type(Kmer{A, K})
    res = CResult()
    GC.@preserve s begin   # count first (SizeUnknown, :33), then fill
        rc = @ccall LIB.kmers_unambiguous(ctx.handle::Ptr{Cvoid}, Ref(cseq(s, A))::Ptr{CSeq}, K::Cint, 1::Cint,
                                          C_NULL::Ptr{UInt64}, C_NULL::Ptr{Int64}, 0::UInt64, MEM_HOST::Cint,
                                          res::Ref{CResult})::Cint
        check(ctx, rc, res, A, s)
        n = Int(res.n_out)
        out = Vector{Tuple{T, Int}}(undef, n)   # eltype of UnambiguousKmers (:39-41), written in place
        GC.@preserve out begin
            rc = @ccall LIB.kmers_unambiguous(ctx.handle::Ptr{Cvoid}, Ref(cseq(s, A))::Ptr{CSeq}, K::Cint, 1::Cint,
                                              pointer(out)::Ptr{UInt64}, C_NULL::Ptr{Int64},
                                              n::UInt64, (MEM_HOST | OUT_TUPLES)::Cint, res::Ref{CResult})::Cint
        end
        check(ctx, rc, res, A, s)
        return out
    end
end

# ---- fused consumers (nothing materialised per kmer) -------------------------------------------
"""
    sketch(fx_hash, CanonicalKmers{A,K}(seq), s)

The bottom-`s` MinHash of `docs/src/minhash.md:31-35` in one fused GPU pass: the `s` smallest
distinct `fx_hash` values of the canonical kmers, ascending.
"""
function sketch(::typeof(fx_hash), it::CanonicalKmers{A, K, S}, s::Integer; seed::UInt = zero(UInt)) where {A <: NucAlphabet24, K, S <: Source}
    ctx, src = context(), it.it.seq
    out = Vector{UInt64}(undef, s)
    res = CResult()
    GC.@preserve src out begin
        rc = @ccall LIB.kmers_minhash(ctx.handle::Ptr{Cvoid}, Ref(cseq(src, A))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                                      seed::UInt64, s::UInt64, pointer(out)::Ptr{UInt64}, MEM_HOST::Cint,
                                      res::Ref{CResult})::Cint
    end
    check(ctx, rc, res, A, src)
    return resize!(out, res.n_out)
end

"Kmer composition counts of `docs/src/composition.md:28-39`: `counts[as_integer(kmer) + 1]` over `FwKmers{A,K}(seq)`."
function composition(it::FwKmers{A, K, S}) where {A <: TwoBitAlphabet, K, S <: Source}
    ctx, src = context(), it.seq
    counts = Vector{UInt32}(undef, 4^K)
    res = CResult()
    GC.@preserve src counts begin
        rc = @ccall LIB.kmers_composition(ctx.handle::Ptr{Cvoid}, Ref(cseq(src, A))::Ptr{CSeq}, K::Cint,
                                          pointer(counts)::Ptr{UInt32}, MEM_HOST::Cint, res::Ref{CResult})::Cint
    end
    check(ctx, rc, res, A, src)
    return counts
end

"Minimizers (`docs/src/replacements.md:33-51`): `mode = 0` is the published `unsafe_extract_minimizer` literally, `1` the true sliding-window minimum."
function minimizers(it::FwKmers{A, K, S}, W::Integer; stride::Integer = 1, mode::Integer = 0) where {A <: NucAlphabet24, K, S <: Source}
    ctx, src = context(), it.seq
    span = K + W - 1
    n = length(src) < span ? 0 : (length(src) - span) ÷ stride + 1
    out = Vector{eltype(it)}(undef, n)
    res = CResult()
    GC.@preserve src out begin
        rc = @ccall LIB.kmers_minimizers(ctx.handle::Ptr{Cvoid}, Ref(cseq(src, A))::Ptr{CSeq}, K::Cint, W::Cint, stride::Cint,
                                         dst_bits(A)::Cint, mode::Cint, pointer(out)::Ptr{UInt64}, MEM_HOST::Cint,
                                         res::Ref{CResult})::Cint
    end
    check(ctx, rc, res, A, src)
    return out
end

# ---- element-wise functions on vectors of kmers ----------------------------------------------
"fx_hash.(v, h) (src/kmer.jl:255-261)"
function Kmers.fx_hash(v::Vector{Kmer{A, K, N}}, h::UInt = zero(UInt)) where {A, K, N}
    ctx = context()
    out = Vector{UInt64}(undef, length(v))
    GC.@preserve v out begin
        rc = @ccall LIB.kmers_fx_hash(ctx.handle::Ptr{Cvoid}, pointer(v)::Ptr{UInt64}, N::Cint,
                                      length(v)::UInt64, h::UInt64, pointer(out)::Ptr{UInt64}, MEM_HOST::Cint)::Cint
    end
    rc == OK || error("kmers_fx_hash: status $rc: $(last_error(ctx))")
    return out
end

# One launch for `f.(v)`, f in reverse / complement / reverse_complement / canonical (src/transformations.jl:1-41).  The three
# BioSequences functions get a method for vectors of kmers (they have none: nothing changes meaning); `reverse` of a Vector
# already MEANS the vector back to front, so the element-wise form is `KmersHIP.reverse_each(v)` == `reverse.(v)`.
function transform_each(op::Integer, v::Vector{Kmer{A, K, N}}) where {A <: NucleicAcidAlphabet, K, N}
    ctx = context()
    out = similar(v)
    bits = BioSequences.bits_per_symbol(A())
    GC.@preserve v out begin
        rc = @ccall LIB.kmers_transform(ctx.handle::Ptr{Cvoid}, op::Cint, pointer(v)::Ptr{UInt64}, K::Cint,
                                        bits::Cint, length(v)::UInt64, pointer(out)::Ptr{UInt64},
                                        MEM_HOST::Cint)::Cint
    end
    rc == OK || error("kmers_transform: status $rc: $(last_error(ctx))")
    return out
end
"`reverse.(v)` (src/transformations.jl:1-10) in one launch"
reverse_each(v::Vector{<:Kmer}) = transform_each(0, v)
BioSequences.complement(v::Vector{Kmer{A, K, N}}) where {A <: NucleicAcidAlphabet, K, N} = transform_each(1, v)
BioSequences.reverse_complement(v::Vector{Kmer{A, K, N}}) where {A <: NucleicAcidAlphabet, K, N} = transform_each(2, v)
BioSequences.canonical(v::Vector{Kmer{A, K, N}}) where {A <: NucleicAcidAlphabet, K, N} = transform_each(3, v)

"as_integer.(v) (src/kmer.jl:305-326) for kmers of up to 128 coding bits: Vector{UInt64} or Vector{UInt128}"
function Kmers.as_integer(v::Vector{Kmer{A, K, N}}) where {A <: NucleicAcidAlphabet, K, N}
    N <= 2 || throw(ArgumentError("Must have at most 128 bits in encoding"))
    ctx = context()
    out = Vector{N == 1 ? UInt64 : UInt128}(undef, length(v))   # little-endian u128 == two UInt64 (low, high)
    bits = BioSequences.bits_per_symbol(A())
    GC.@preserve v out begin
        rc = @ccall LIB.kmers_transform(ctx.handle::Ptr{Cvoid}, 7::Cint, pointer(v)::Ptr{UInt64}, K::Cint, bits::Cint,
                                        length(v)::UInt64, pointer(out)::Ptr{UInt64}, MEM_HOST::Cint)::Cint
    end
    rc == OK || error("kmers_transform: status $rc: $(last_error(ctx))")
    return out
end

# ---- batches of records: one launch for many short sequences ------------------------------------
struct CSpan
    first_base::UInt64
    n_bases::UInt64
end

"""
    collect_batch(CanonicalKmers{A,K}, seqs; hashes = false, seed = 0x0)
    collect_batch(FwKmers{A,K}, seqs)

`vcat((collect(It(s)) for s in seqs)...)` from one GPU launch (`kmers_batch`), plus the element offset of
every record.  `seqs`: LongSequences of one alphabet.  The records' data words are concatenated into one
pool (each record starts on a word boundary; a record is a view `(first_base, n_bases)` of the pool).
Returns `(kmers, hashes_or_nothing, offsets)`; record `i` owns `kmers[offsets[i]+1 : offsets[i+1]]`.
"""
function collect_batch(::Type{It}, seqs::Vector{<:LongSequence}; hashes::Bool = false, seed::UInt = zero(UInt)) where {A, K, It <: Union{FwKmers{A, K}, CanonicalKmers{A, K}}}
    ctx = context()
    T = Kmers.derive_type(Kmer{A, K})
    isempty(seqs) && return (T[], hashes ? UInt64[] : nothing, UInt64[0])
    sbits = BioSequences.bits_per_symbol(Alphabet(first(seqs)))
    per = 64 ÷ sbits
    pool = UInt64[]
    spans = Vector{CSpan}(undef, length(seqs))
    for (i, s) in enumerate(seqs)
        spans[i] = CSpan(length(pool) * per, length(s))
        append!(pool, s.data)
    end
    push!(pool, zero(UInt64))
    seq = CSeq(pointer(pool), (length(pool) - 1) * per, 0, 0, Int32(sbits), 0)
    mode = It <: CanonicalKmers ? Int32(1) : Int32(0)
    offsets = Vector{UInt64}(undef, length(seqs) + 1)
    res = CResult()
    call(out_a, out_b, cap) = GC.@preserve pool spans offsets begin
        @ccall LIB.kmers_batch(ctx.handle::Ptr{Cvoid}, Ref(seq)::Ptr{CSeq}, pointer(spans)::Ptr{CSpan}, length(seqs)::UInt64,
                               mode::Cint, K::Cint, dst_bits(A)::Cint, out_a::Ptr{Cvoid}, out_b::Ptr{Cvoid}, seed::UInt64,
                               pointer(offsets)::Ptr{UInt64}, cap::UInt64, MEM_HOST::Cint, Ref(res)::Ptr{CResult})::Cint
    end
    rc = call(C_NULL, C_NULL, 0)                      # size query
    rc == OK || error("kmers_batch: status $rc: $(last_error(ctx))")
    total = Int(res.n_out)
    kmers = Vector{T}(undef, total)
    hs = hashes && mode == 1 ? Vector{UInt64}(undef, total) : nothing
    GC.@preserve kmers hs begin
        rc = call(pointer(kmers), hs === nothing ? C_NULL : pointer(hs), total)
    end
    if rc == E_ENCODE    # res.n_out = index of the failing record; the reference would have thrown there
        bad = seqs[Int(res.n_out) + 1]
        throw(BioSequences.EncodeError(A(), reinterpret(eltype(bad), res.err_enc % UInt8)))
    end
    rc == OK || error("kmers_batch: status $rc: $(last_error(ctx))")
    return (kmers, hs, offsets)
end

"""
    collect_batch(SpacedKmers{A,K,J}, seqs)

`vcat((collect(SpacedKmers{A,K,J}(s)) for s in seqs)...)` from one GPU launch (`kmers_batch_spaced`), e.g.
`each_codon` of every coding sequence of a genome.  Returns `(kmers, offsets)`.
"""
function collect_batch(::Type{SpacedKmers{A, K, J}}, seqs::Vector{<:LongSequence}) where {A, K, J}
    ctx = context()
    T = Kmers.derive_type(Kmer{A, K})
    isempty(seqs) && return (T[], UInt64[0])
    sbits = BioSequences.bits_per_symbol(Alphabet(first(seqs)))
    per = 64 ÷ sbits
    pool = UInt64[]
    spans = Vector{CSpan}(undef, length(seqs))
    for (i, s) in enumerate(seqs)
        spans[i] = CSpan(length(pool) * per, length(s))
        append!(pool, s.data)
    end
    push!(pool, zero(UInt64))
    seq = CSeq(pointer(pool), (length(pool) - 1) * per, 0, 0, Int32(sbits), 0)
    offsets = Vector{UInt64}(undef, length(seqs) + 1)
    res = CResult()
    call(out, cap) = GC.@preserve pool spans offsets begin
        @ccall LIB.kmers_batch_spaced(ctx.handle::Ptr{Cvoid}, Ref(seq)::Ptr{CSeq}, pointer(spans)::Ptr{CSpan}, length(seqs)::UInt64,
                                      K::Cint, J::UInt64, dst_bits(A)::Cint, out::Ptr{Cvoid}, pointer(offsets)::Ptr{UInt64},
                                      cap::UInt64, MEM_HOST::Cint, Ref(res)::Ptr{CResult})::Cint
    end
    rc = call(C_NULL, 0)                              # size query
    rc == OK || error("kmers_batch_spaced: status $rc: $(last_error(ctx))")
    total = Int(res.n_out)
    kmers = Vector{T}(undef, total)
    GC.@preserve kmers begin
        rc = call(pointer(kmers), total)
    end
    if rc == E_ENCODE
        bad = seqs[Int(res.n_out) + 1]
        throw(BioSequences.EncodeError(A(), reinterpret(eltype(bad), res.err_enc % UInt8)))
    end
    rc == OK || error("kmers_batch_spaced: status $rc: $(last_error(ctx))")
    return (kmers, offsets)
end

"""
    sketch_batch(fx_hash, CanonicalKmers{A,K}, seqs, s; seed = 0x0)

`[MinHash.sketch(fx_hash, CanonicalKmers{A,K}(x), s) for x in seqs]` from one launch sequence
(`kmers_minhash_batch`): a vector of ascending `Vector{UInt64}`, one per record.
"""
function sketch_batch(::typeof(fx_hash), ::Type{CanonicalKmers{A, K}}, seqs::Vector{<:LongSequence}, s::Integer; seed::UInt = zero(UInt)) where {A, K}
    ctx = context()
    isempty(seqs) && return Vector{UInt64}[]
    sbits = BioSequences.bits_per_symbol(Alphabet(first(seqs)))
    per = 64 ÷ sbits
    pool = UInt64[]
    spans = Vector{CSpan}(undef, length(seqs))
    for (i, x) in enumerate(seqs)
        spans[i] = CSpan(length(pool) * per, length(x))
        append!(pool, x.data)
    end
    push!(pool, zero(UInt64))
    seq = CSeq(pointer(pool), (length(pool) - 1) * per, 0, 0, Int32(sbits), 0)
    out = Matrix{UInt64}(undef, s, length(seqs))      # column i = record i (column-major == the C layout)
    counts = Vector{UInt64}(undef, length(seqs))
    res = CResult()
    rc = GC.@preserve pool spans out counts begin
        @ccall LIB.kmers_minhash_batch(ctx.handle::Ptr{Cvoid}, Ref(seq)::Ptr{CSeq}, pointer(spans)::Ptr{CSpan}, length(seqs)::UInt64,
                                       K::Cint, dst_bits(A)::Cint, seed::UInt64, s::UInt64, pointer(out)::Ptr{UInt64},
                                       pointer(counts)::Ptr{UInt64}, MEM_HOST::Cint, Ref(res)::Ptr{CResult})::Cint
    end
    if rc == E_ENCODE
        bad = seqs[Int(res.n_out) + 1]
        throw(BioSequences.EncodeError(A(), reinterpret(eltype(bad), res.err_enc % UInt8)))
    end
    rc == OK || error("kmers_minhash_batch: status $rc: $(last_error(ctx))")
    return [out[1:Int(counts[i]), i] for i in eachindex(seqs)]
end

# ---- sharding one long sequence over several GPUs / processes ---------------------------------
struct CShard
    first_kmer::UInt64
    n_kmers::UInt64
    first_base::UInt64
    n_bases::UInt64
    first_word::UInt64
    n_own_words::UInt64
    halo_words::UInt32
    send_words::UInt32
end

"""
    shard_plan(seq_length, K, n_shards, shard_id; stride = 1, src_bits = 4)

The contiguous, word- and stride-aligned range of kmers shard `shard_id` (0-based) owns, and the
number of words it needs from its right neighbour (`kmers_shard_plan`, include/kmers_hip.h).
Run the shard with `CSeq(pointer(own_words_plus_halo), n_bases, 0, first_base, ...)`.
"""
function shard_plan(len::Integer, K::Integer, n_shards::Integer, shard_id::Integer; stride::Integer = 1, src_bits::Integer = 4)
    out = Ref{CShard}()
    rc = @ccall LIB.kmers_shard_plan(len::UInt64, K::Cint, stride::UInt64, src_bits::Cint, n_shards::Cint,
                                     shard_id::Cint, out::Ptr{CShard})::Cint
    rc == OK || error("kmers_shard_plan: bad arguments")
    return out[]
end

# ---- the sharded path: one Julia process per GPU, RCCL behind the C ABI ---------------------------
# (include/kmers_hip.h, "the communication of the sharded path").  The 128-byte id made by ONE rank travels over whatever
# the host program has (MPI.jl `MPI.bcast`, Distributed.jl `remotecall`, a shared file).
"ncclGetUniqueId through the library: call on one rank, hand the bytes to all."
function comm_id()
    id = Vector{UInt8}(undef, 128)
    rc = @ccall LIB.kmers_comm_id(pointer(id)::Ptr{UInt8})::Cint
    rc == OK || error("kmers_comm_id: status $rc")
    return id
end

mutable struct Comm
    ctx::Context
    handle::Ptr{Cvoid}
    rank::Int
    n_ranks::Int
end

"ncclCommInitRank on the context's device; collective over all `n_ranks` callers."
function Comm(id::Vector{UInt8}, n_ranks::Integer, rank::Integer; ctx::Context = context())
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = GC.@preserve id @ccall LIB.kmers_comm_create(ctx.handle::Ptr{Cvoid}, pointer(id)::Ptr{UInt8}, n_ranks::Cint, rank::Cint,
                                                      h::Ptr{Ptr{Cvoid}})::Cint
    rc == OK || error("kmers_comm_create: status $rc: $(last_error(ctx))")
    c = Comm(ctx, h[], rank, n_ranks)
    # ncclCommDestroy synchronises the stream and is collective in effect: it must not run from the GC (arbitrary time, arbitrary
    # order across ranks, and finalizers must not block).  Call close(comm) on every rank; the finalizer only reports a leak.
    finalizer(x -> (x.handle == C_NULL || @async @warn "KmersHIP.Comm was never closed: call close(comm) on every rank"), c)
    return c
end

"ncclCommDestroy (after a stream synchronize): call on every rank when the sharded work is done."
function Base.close(c::Comm)
    c.handle == C_NULL && return nothing
    rc = @ccall LIB.kmers_comm_destroy(c.ctx.handle::Ptr{Cvoid}, c.handle::Ptr{Cvoid})::Cint
    c.handle = C_NULL
    rc == OK || error("kmers_comm_destroy: status $rc: $(last_error(c.ctx))")
    return nothing
end

# ---- device memory (include/kmers_hip.h, "device memory") ------------------------------------------------------------------
# `device_alloc` of 128 MiB or more comes from the device's CLASS POOL: physical memory whose HBM region class the library has
# measured, every block assembled so that the arrays of one launch differ in class at every position (no reservation; `pool_info`,
# `pool_trim!`).  `device_free` of such a block does not wait: the block stays mapped in the pool's cache for the next request of
# its shape, whose stream comes after the work that was queued at the free.
"What the device's class pool holds: (bytes held, bytes in blocks, classes found, GB/s of two store streams in two classes / in one)."
function pool_info(ctx::Context = context())
    held, used, ncls = Ref{Csize_t}(0), Ref{Csize_t}(0), Ref{Cint}(0)
    two, one = Ref{Cdouble}(0), Ref{Cdouble}(0)
    rc = @ccall LIB.kmers_pool_info(ctx.handle::Ptr{Cvoid}, held::Ptr{Csize_t}, used::Ptr{Csize_t}, ncls::Ptr{Cint}, C_NULL::Ptr{Csize_t},
                                    two::Ptr{Cdouble}, one::Ptr{Cdouble})::Cint
    rc == OK || error("kmers_pool_info: status $rc: $(last_error(ctx))")
    return (held = Int(held[]), in_use = Int(used[]), classes = Int(ncls[]), two_class_gbps = two[], one_class_gbps = one[])
end
"Return the pool's free handles to the driver; the bytes released."
function pool_trim!(ctx::Context = context())
    r = Ref{Csize_t}(0)
    rc = @ccall LIB.kmers_pool_trim(ctx.handle::Ptr{Cvoid}, r::Ptr{Csize_t})::Cint
    rc == OK || error("kmers_pool_trim: status $rc: $(last_error(ctx))")
    return Int(r[])
end
"""
`n` elements of `T` in HBM: from 128 MiB on a block of the device's class pool, else a plain allocation.  `lone_output = true`: the ONLY output array of the launches that fill it (`collect` of an iterator
without hashes / reverse complements): its second half lies in another region class of HBM than its first and such a launch
writes it through two windows (`kmers_dev_alloc_role`, `KMERS_ALLOC_LONE_OUTPUT`).
"""
function device_alloc(ctx::Context, ::Type{T}, n::Integer; lone_output::Bool = false) where {T}
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = @ccall LIB.kmers_dev_alloc_role(ctx.handle::Ptr{Cvoid}, (n * sizeof(T))::Csize_t, Cint(lone_output)::Cint, p::Ptr{Ptr{Cvoid}})::Cint
    rc == OK || error("kmers_dev_alloc_role: status $rc: $(last_error(ctx))")
    return Ptr{T}(p[])
end
device_free(ctx::Context, p::Ptr) = (@ccall LIB.kmers_dev_free(ctx.handle::Ptr{Cvoid}, p::Ptr{Cvoid})::Cint; nothing)

"""
    halo_exchange!(comm, shard, words_dev)

The one neighbour step of the path (grouped ncclSend/ncclRecv on the context's stream, enqueue only): `words_dev` (HBM:
the shard's own words followed by room for `shard.halo_words`) receives the first words of shard `rank + 1` behind its own
and sends its first `shard.send_words` words to `rank - 1`.
"""
function halo_exchange!(c::Comm, shard::CShard, words_dev::Ptr{UInt64})
    rc = @ccall LIB.kmers_halo_exchange(c.ctx.handle::Ptr{Cvoid}, c.handle::Ptr{Cvoid}, Ref(shard)::Ptr{CShard},
                                        words_dev::Ptr{UInt64})::Cint
    rc == OK || error("kmers_halo_exchange: status $rc: $(last_error(c.ctx))")
    return nothing
end

"The first EncodeError of the whole sequence (`res` holds this shard's, positions global): ncclAllReduce(min)."
function first_error!(c::Comm, res::CResult)
    rc = @ccall LIB.kmers_first_error_allreduce(c.ctx.handle::Ptr{Cvoid}, c.handle::Ptr{Cvoid}, res::Ref{CResult})::Cint
    (rc == OK || rc == E_ENCODE) || error("kmers_first_error_allreduce: status $rc: $(last_error(c.ctx))")
    return res
end

"(offset of this shard's elements in the global output, total) for SizeUnknown iterators: ncclAllGather + scan."
function output_offsets(c::Comm, n_local::Integer)
    off, tot = Ref{UInt64}(0), Ref{UInt64}(0)
    rc = @ccall LIB.kmers_offsets_allgather(c.ctx.handle::Ptr{Cvoid}, c.handle::Ptr{Cvoid}, n_local::UInt64,
                                            off::Ptr{UInt64}, tot::Ptr{UInt64})::Cint
    rc == OK || error("kmers_offsets_allgather: status $rc: $(last_error(c.ctx))")
    return Int(off[]), Int(tot[])
end

# ---- chunk-buffered iterate(): `for x in gpu(it)` ------------------------------------------------
# The reference's protocol (src/iterators/FwKmers.jl:57-66, CanonicalKmers.jl:54-66, SpacedKmers.jl:121-139,
# UnambiguousKmers.jl:59-62) yields one element per `iterate`; here `iterate` hands out elements of a chunk the device computed
# while the loop was busy with the chunk before: the source goes to HBM once, two chunk buffers in HBM and two in page-locked host
# memory take turns -- chunk c + 1 is launched (`KMERS_MEM_DEVICE | KMERS_ASYNC`) and its copy enqueued
# (`kmers_memcpy_d2h_async`) BEFORE the loop gets chunk c, and `kmers_sync` is only called when the loop has used chunk c up.
# A chunk is a view of the sequence (`first_base`, `index_origin` = the chunk's offset: EncodeError positions and UnambiguousKmers
# starts stay those of the whole sequence), K - J symbols of overlap -- exactly how a resumed reference iterator would restart.
# kmers.jl_amd/host.py (`_ChunkPipe`) is the same pipeline through ctypes, tested on hardware (tests/test_gpu_mirror.py).
struct GPUIterator{I}
    it::I
    chunk::Int
end
gpu(it::GpuIterator; chunk::Int = 1 << 22) = GPUIterator(it, chunk)
Base.IteratorSize(::Type{GPUIterator{I}}) where {I} = Base.IteratorSize(I)
Base.length(g::GPUIterator) = length(g.it)
Base.eltype(::Type{GPUIterator{I}}) where {I} = eltype(I)

kmer_alphabet(::Union{FwKmers{A}, FwRvIterator{A}, CanonicalKmers{A}, SpacedKmers{A}, UnambiguousKmers{A}}) where {A} = A
ksize_of(::Union{FwKmers{A, K}, FwRvIterator{A, K}, CanonicalKmers{A, K}, SpacedKmers{A, K}, UnambiguousKmers{A, K}}) where {A, K} = K
stride_of(::SpacedKmers{A, K, J}) where {A, K, J} = J
stride_of(::Union{FwKmers, FwRvIterator, CanonicalKmers, UnambiguousKmers}) = 1
# what a chunk is counted in: elements -- for UnambiguousKmers (SizeUnknown, UnambiguousKmers.jl:33) candidate windows
units(it::Union{FwKmers, FwRvIterator, CanonicalKmers, SpacedKmers}) = length(it)
units(it::UnambiguousKmers) = max(0, ncodeunits_or_length(source(it)) - ksize_of(it) + 1)
# units whose windows end before the 0-based symbol `bad0`: what iterate() yields before it throws there (FwKmers.jl:112,
# CanonicalKmers.jl:139, SpacedKmers.jl:133-134); host.py `_yielded_before`
function units_before(it, bad0::Int)
    K, J = ksize_of(it), stride_of(it)
    J >= K && return bad0 ÷ J
    bad0 < K && return 0
    return (bad0 - K) ÷ J + 1
end

# units [u0, u0 + n) of the iteration as a view of `base` (a kmers_seq over host or device words)
function chunk_view(base::CSeq, K::Int, J::Int, u0::Int, n::Int)
    off = u0 * J
    return CSeq(base.words, ((n - 1) * J + K) % UInt64, base.first_base + off % UInt64, off % UInt64, base.src_bits, base.alphabet)
end

# One chunk, ENQUEUED: units [u0, u0 + n) into `out` (HBM, room for n elements of eltype(it)); one method per iterator type.
function fill_chunk(it::FwKmers{A, K}, ctx::Ptr{Cvoid}, base::CSeq, out::Ptr{Cvoid}, u0::Int, n::Int, res::CResult) where {A, K}
    @ccall LIB.kmers_fw(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                        out::Ptr{UInt64}, C_NULL::Ptr{UInt64}, (MEM_DEVICE | ASYNC)::Cint, res::Ref{CResult})::Cint
end
function fill_chunk(it::FwRvIterator{A, K}, ctx::Ptr{Cvoid}, base::CSeq, out::Ptr{Cvoid}, u0::Int, n::Int, res::CResult) where {A, K}
    @ccall LIB.kmers_fw(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                        out::Ptr{UInt64}, C_NULL::Ptr{UInt64}, (MEM_DEVICE | ASYNC | OUT_TUPLES)::Cint, res::Ref{CResult})::Cint
end
function fill_chunk(it::CanonicalKmers{A, K}, ctx::Ptr{Cvoid}, base::CSeq, out::Ptr{Cvoid}, u0::Int, n::Int, res::CResult) where {A, K}
    @ccall LIB.kmers_canonical(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                               out::Ptr{UInt64}, C_NULL::Ptr{UInt64}, 0::UInt64, (MEM_DEVICE | ASYNC)::Cint, res::Ref{CResult})::Cint
end
function fill_chunk(it::SpacedKmers{A, K, J}, ctx::Ptr{Cvoid}, base::CSeq, out::Ptr{Cvoid}, u0::Int, n::Int, res::CResult) where {A, K, J}
    @ccall LIB.kmers_spaced(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, J, u0, n))::Ptr{CSeq}, K::Cint, J::Cint, dst_bits(A)::Cint,
                            out::Ptr{UInt64}, (MEM_DEVICE | ASYNC)::Cint, res::Ref{CResult})::Cint
end
# (kmer, start) tuples; the number kept is known at the next kmers_sync (its res.n_out); capacity = the chunk's candidate windows
function fill_chunk(it::UnambiguousKmers{A, K}, ctx::Ptr{Cvoid}, base::CSeq, out::Ptr{Cvoid}, u0::Int, n::Int, res::CResult) where {A, K}
    @ccall LIB.kmers_unambiguous(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, 1::Cint,
                                 out::Ptr{UInt64}, C_NULL::Ptr{Int64}, n::UInt64, (MEM_DEVICE | ASYNC | OUT_TUPLES)::Cint,
                                 res::Ref{CResult})::Cint
end

# The same units computed NOW, from host memory into a fresh Vector (the clean prefix of a chunk that met an EncodeError).
function fill_sync(it, ctx::Context, u0::Int, n::Int)
    A, s = kmer_alphabet(it), source(it)
    buf = Vector{eltype(it)}(undef, n)
    res = CResult()
    GC.@preserve s buf begin
        # (the device-pointer form of fill_chunk with host pointers: the flags are replaced, nothing else differs)
        rc = fill_chunk_host(it, ctx.handle, cseq(s, A), Ptr{Cvoid}(pointer(buf)), u0, n, res)
    end
    check(ctx, rc, res, A, s)
    it isa UnambiguousKmers && resize!(buf, Int(res.n_out))
    return buf
end
fill_chunk_host(it::FwKmers{A, K}, ctx, base, out, u0, n, res) where {A, K} =
    @ccall LIB.kmers_fw(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                        out::Ptr{UInt64}, C_NULL::Ptr{UInt64}, MEM_HOST::Cint, res::Ref{CResult})::Cint
fill_chunk_host(it::FwRvIterator{A, K}, ctx, base, out, u0, n, res) where {A, K} =
    @ccall LIB.kmers_fw(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                        out::Ptr{UInt64}, C_NULL::Ptr{UInt64}, (MEM_HOST | OUT_TUPLES)::Cint, res::Ref{CResult})::Cint
fill_chunk_host(it::CanonicalKmers{A, K}, ctx, base, out, u0, n, res) where {A, K} =
    @ccall LIB.kmers_canonical(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, dst_bits(A)::Cint,
                               out::Ptr{UInt64}, C_NULL::Ptr{UInt64}, 0::UInt64, MEM_HOST::Cint, res::Ref{CResult})::Cint
fill_chunk_host(it::SpacedKmers{A, K, J}, ctx, base, out, u0, n, res) where {A, K, J} =
    @ccall LIB.kmers_spaced(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, J, u0, n))::Ptr{CSeq}, K::Cint, J::Cint, dst_bits(A)::Cint,
                            out::Ptr{UInt64}, MEM_HOST::Cint, res::Ref{CResult})::Cint
fill_chunk_host(it::UnambiguousKmers{A, K}, ctx, base, out, u0, n, res) where {A, K} =
    @ccall LIB.kmers_unambiguous(ctx::Ptr{Cvoid}, Ref(chunk_view(base, K, 1, u0, n))::Ptr{CSeq}, K::Cint, 1::Cint,
                                 out::Ptr{UInt64}, C_NULL::Ptr{Int64}, n::UInt64, (MEM_HOST | OUT_TUPLES)::Cint,
                                 res::Ref{CResult})::Cint

# The pipeline's buffers.  It owns a context (= a stream) of its own, so that its finalizer -- a loop left with `break` never
# reaches the end of the iteration -- touches nothing another task may be using.
mutable struct ChunkPipe{E}
    ctx::Ptr{Cvoid}
    base::CSeq                       # the source in HBM, as a kmers_seq
    d_src::Ptr{Cvoid}
    d_out::NTuple{2, Ptr{Cvoid}}
    h_ptr::NTuple{2, Ptr{Cvoid}}
    h_out::NTuple{2, Vector{E}}      # the page-locked buffers as Julia arrays (unsafe_wrap, not owned)
    cap::Int
    res::CResult
end

function pipe_call(ctx::Ptr{Cvoid}, rc::Integer, what)
    rc == OK || error("$what: status $rc: $(unsafe_string(@ccall LIB.kmers_last_error(ctx::Ptr{Cvoid})::Cstring))")
    return nothing
end

function ChunkPipe(g::GPUIterator)
    it = g.it
    E, A, s = eltype(it), kmer_alphabet(it), source(it)
    cap = max(1, min(g.chunk, units(it)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = @ccall LIB.kmers_ctx_create(0::Cint, C_NULL::Ptr{Cvoid}, h::Ptr{Ptr{Cvoid}})::Cint
    rc == OK || error("kmers_ctx_create failed ($rc): no usable MI355X / HIP device")
    ctx = h[]
    alloc(bytes, lone) = (p = Ref{Ptr{Cvoid}}(C_NULL);
                          pipe_call(ctx, (@ccall LIB.kmers_dev_alloc_role(ctx::Ptr{Cvoid}, bytes::Csize_t, Cint(lone)::Cint, p::Ptr{Ptr{Cvoid}})::Cint), "kmers_dev_alloc_role"); p[])
    pinned(bytes) = (p = Ref{Ptr{Cvoid}}(C_NULL);
                     pipe_call(ctx, (@ccall LIB.kmers_host_alloc(ctx::Ptr{Cvoid}, bytes::Csize_t, p::Ptr{Ptr{Cvoid}})::Cint), "kmers_host_alloc"); p[])
    # the words (bytes) that hold the view go to HBM once
    host = cseq(s, A)
    per = 64 ÷ Int(host.src_bits)                         # symbols per 8-byte word
    w0 = Int(host.first_base) ÷ per
    src_bytes = host.src_bits == 8 ? Int(host.first_base + host.n_bases) - 8 * w0 :
                                     8 * (cld(Int(host.first_base + host.n_bases), per) - w0)
    d_src = alloc(src_bytes + 16, false)
    GC.@preserve s pipe_call(ctx, (@ccall LIB.kmers_memcpy_h2d(ctx::Ptr{Cvoid}, d_src::Ptr{Cvoid}, (Ptr{UInt8}(host.words) + 8 * w0)::Ptr{Cvoid},
                                                             src_bytes::Csize_t)::Cint), "kmers_memcpy_h2d")
    base = CSeq(Ptr{UInt64}(d_src), host.n_bases, host.first_base - (w0 * per) % UInt64, 0, host.src_bits, host.alphabet)
    d_out = (alloc(cap * sizeof(E), true), alloc(cap * sizeof(E), true))
    h_ptr = (pinned(cap * sizeof(E)), pinned(cap * sizeof(E)))
    h_out = (unsafe_wrap(Array, Ptr{E}(h_ptr[1]), cap), unsafe_wrap(Array, Ptr{E}(h_ptr[2]), cap))
    p = ChunkPipe{E}(ctx, base, d_src, d_out, h_ptr, h_out, cap, CResult())
    finalizer(close!, p)
    return p
end

"Free the pipeline's buffers and its context (idempotent; the finalizer of a pipeline whose loop was left early)."
function close!(p::ChunkPipe)
    p.ctx == C_NULL && return nothing
    ctx = p.ctx
    p.ctx = C_NULL
    @ccall LIB.kmers_sync(ctx::Ptr{Cvoid}, C_NULL::Ptr{CResult})::Cint
    for q in (p.d_src, p.d_out...)
        @ccall LIB.kmers_dev_free(ctx::Ptr{Cvoid}, q::Ptr{Cvoid})::Cint
    end
    for q in p.h_ptr
        @ccall LIB.kmers_host_free(ctx::Ptr{Cvoid}, q::Ptr{Cvoid})::Cint
    end
    @ccall LIB.kmers_ctx_destroy(ctx::Ptr{Cvoid})::Cvoid
    return nothing
end

# enqueue units [u0, u0 + n): the kernel, then the copy of its output into the slot's page-locked buffer
function enqueue!(g::GPUIterator, p::ChunkPipe{E}, slot::Int, u0::Int, n::Int) where {E}
    rc = fill_chunk(g.it, p.ctx, p.base, p.d_out[slot], u0, n, p.res)
    pipe_call(p.ctx, rc, "fill_chunk")
    pipe_call(p.ctx, (@ccall LIB.kmers_memcpy_d2h_async(p.ctx::Ptr{Cvoid}, p.h_ptr[slot]::Ptr{Cvoid}, p.d_out[slot]::Ptr{Cvoid},
                                                        (n * sizeof(E))::Csize_t)::Cint), "kmers_memcpy_d2h_async")
    return nothing
end

mutable struct PipeState{E}
    pipe::ChunkPipe{E}
    buf::Vector{E}          # what the loop is consuming: one of the page-locked buffers, or the clean prefix before an error
    i::Int                  # next element of buf
    n::Int                  # elements of buf that count
    slot::Int               # the slot in flight (its chunk: units [u0, u0 + m))
    u0::Int
    m::Int                  # 0: nothing in flight
    err::Union{Nothing, Exception}
end

function Base.iterate(g::GPUIterator)
    units(g.it) == 0 && return nothing
    p = ChunkPipe(g)
    m = min(p.cap, units(g.it))
    enqueue!(g, p, 1, 0, m)
    return iterate(g, PipeState{eltype(g)}(p, p.h_out[1], 1, 0, 1, 0, m, nothing))
end

function Base.iterate(g::GPUIterator, st::PipeState)
    p, it = st.pipe, g.it
    while st.i > st.n                                   # the buffer is used up
        if st.err !== nothing
            close!(p)
            throw(st.err)
        elseif st.m == 0
            close!(p)
            return nothing
        end
        rc = @ccall LIB.kmers_sync(p.ctx::Ptr{Cvoid}, p.res::Ref{CResult})::Cint      # the chunk in flight (and only it) is done
        u0, m, slot = st.u0, st.m, st.slot
        if rc == E_ENCODE
            # the reference yields every element whose window ends before the offending symbol, then throws there
            A, s = kmer_alphabet(it), source(it)
            sym = (s isa BioSequence || s isa SymbolVector) ? reinterpret(eltype(s), p.res.err_enc % UInt8) : repr(p.res.err_enc % UInt8)
            st.err = BioSequences.EncodeError(A(), sym)
            st.m = 0
            good = units_before(it, Int(p.res.err_pos) - 1) - u0
            if good > 0
                st.buf = fill_sync(it, context(), u0, good)
                st.i, st.n = 1, length(st.buf)
            end
            continue
        end
        rc == OK || (close!(p); error("libkmers_hip: status $rc"))
        st.buf, st.i = p.h_out[slot], 1
        st.n = (it isa UnambiguousKmers && p.base.src_bits != 2) ? Int(p.res.n_out) : m     # (a 2-bit source drops nothing: the count is the chunk's)
        u1 = u0 + m
        if u1 < units(it)                               # chunk c + 1 goes to the device before the loop gets chunk c
            st.slot, st.u0, st.m = 3 - slot, u1, min(p.cap, units(it) - u1)
            enqueue!(g, p, st.slot, st.u0, st.m)
        else
            st.m = 0
        end
    end
    x = @inbounds st.buf[st.i]
    st.i += 1
    return (x, st)
end

end # module
