"""julia/KmersHIP.jl cannot run here (no Julia in the image).  What CAN be pinned mechanically is that its text
agrees with the C ABI it binds: every `@ccall LIB.kmers_*` against the prototype in include/kmers_hip.h (arity, scalar kinds, pointer
vs scalar, pointee types), the four mirrored structs against the C structs (field order and types, and -- through a compiled
`offsetof` probe -- offsets and sizes), the status / flag constants against the #defines, and that every iterator type the binding
routes to the device has BOTH its bulk form (`gpu_collect`) and its chunk form (`fill_chunk`, the iterate() protocol:
src/iterators/FwKmers.jl:57-66, CanonicalKmers.jl:54-66, SpacedKmers.jl:121-139, UnambiguousKmers.jl:59-62).  The checker is itself
checked: a binding with two arguments swapped, a field moved, a flag changed must fail."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "kmers_hip.h")
JULIA = os.path.join(ROOT, "julia", "KmersHIP.jl")


# ---- the header ---------------------------------------------------------------------------------------------------------------
def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_type(decl):
    """('scalar', 'uint64_t') / ('ptr', 'kmers_seq') from a C parameter or field declaration (name dropped, const dropped)."""
    decl = re.sub(r"\bconst\b", " ", decl).strip()
    stars = decl.count("*")
    words = decl.replace("*", " ").split()
    if len(words) > 1 and words[-1] not in ("int", "char", "double", "void", "size_t") and not words[-1].endswith("_t"):
        words = words[:-1]  # the parameter's name
    base = " ".join(words)
    base = {"unsigned char": "uint8_t", "char": "char"}.get(base, base)
    if stars == 0:
        return ("scalar", base)
    if stars == 1:
        return ("ptr", base)
    return ("ptr", "ptr")  # pointer to pointer


def header_prototypes():
    text = strip_comments(open(HEADER).read())
    protos = {}
    for m in re.finditer(r"(?m)^((?:const\s+)?(?:int|void|uint64_t|char)\s*\*?)\s*(kmers_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else [c_type(a) for a in args.split(",")]
        protos[name] = (c_type(ret + " r") if "*" in ret else ("scalar", ret.strip()), params)
    return protos


def header_structs():
    text = strip_comments(open(HEADER).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(kmers_\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if decl:
                name = decl.replace("*", " ").split()[-1]
                fields.append((name, c_type(decl)))
        structs[m.group(2)] = fields
    return structs


def header_defines():
    text = open(HEADER).read()
    return {m.group(1): int(m.group(2), 0) for m in re.finditer(r"(?m)^#define\s+(KMERS_\w+)\s+(0x[0-9a-fA-F]+|\d+)\b", text)}


# ---- the binding --------------------------------------------------------------------------------------------------------------
JL_SCALARS = {"Cint": "int", "Csize_t": "size_t", "UInt64": "uint64_t", "Int64": "int64_t", "UInt32": "uint32_t", "Int32": "int32_t",
              "Cdouble": "double", "Cvoid": "void", "UInt8": "uint8_t"}
JL_STRUCTS = {"CSeq": "kmers_seq", "CResult": "kmers_result", "CSpan": "kmers_span", "CShard": "kmers_shard"}


def jl_type(t):
    t = t.strip()
    if t == "Cstring":
        return ("ptr", "char")
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.*)\}", t)
    if m:
        inner = m.group(1).strip()
        if inner.startswith(("Ptr{", "Ref{")):
            return ("ptr", "ptr")
        return ("ptr", JL_STRUCTS.get(inner, JL_SCALARS.get(inner, inner)))
    return ("scalar", JL_SCALARS.get(t, t))


def split_top_level(text, sep=","):
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def arg_type(arg):
    """the text after the LAST top-level `::` of a @ccall argument"""
    depth, cut = 0, -1
    for i, ch in enumerate(arg):
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        elif ch == ":" and depth == 0 and arg[i:i + 2] == "::":
            cut = i
    assert cut >= 0, f"a @ccall argument without a type: {arg!r}"
    return arg[cut + 2:].strip()


def julia_ccalls(text):
    """[(function name, [argument types], return type, line number)] of every `@ccall LIB.kmers_*`"""
    calls = []
    for m in re.finditer(r"@ccall\s+LIB\.(kmers_\w+)\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        args = text[m.end():i - 1]
        ret = re.match(r"::\s*([\w{}]+)", text[i:])
        assert ret, f"@ccall {m.group(1)} without a return type"
        calls.append((m.group(1), [arg_type(a) for a in split_top_level(args)], ret.group(1), text.count("\n", 0, m.start()) + 1))
    return calls


def julia_structs(text):
    out = {}
    for m in re.finditer(r"(?m)^(?:mutable\s+)?struct\s+(C\w+)\n(.*?)^end", text, flags=re.S):
        if m.group(1) not in JL_STRUCTS:
            continue
        fields = [(f.group(1), f.group(2)) for f in re.finditer(r"(?m)^\s+(\w+)::(\w+(?:\{\w+\})?)\s*$", m.group(2))]
        out[m.group(1)] = fields
    return out


def compatible(c, j):
    """a C parameter (kind, base) against a Julia @ccall argument type (kind, base)"""
    if c[0] != j[0]:
        return False
    if c[0] == "scalar":
        return c[1] == j[1]
    if j[1] == "void" or c[1] == "void":                      # Ptr{Cvoid} passes for any pointer; a void * takes any pointer
        return not (c[1] == "ptr") or j[1] in ("ptr", "void")
    if c[1] == "kmers_ctx":
        return j[1] == "void"
    if c[1] == "char":
        return j[1] in ("char", "uint8_t")
    return c[1] == j[1]


def mismatches(julia_text):
    """every disagreement between the binding's text and the header, as strings"""
    protos, problems = header_prototypes(), []
    for name, args, ret, line in julia_ccalls(julia_text):
        if name not in protos:
            problems.append(f"line {line}: {name} is not declared in include/kmers_hip.h")
            continue
        c_ret, c_args = protos[name]
        if len(args) != len(c_args):
            problems.append(f"line {line}: {name} takes {len(c_args)} arguments, the binding passes {len(args)}")
            continue
        for k, (ca, ja) in enumerate(zip(c_args, args)):
            if not compatible(ca, jl_type(ja)):
                problems.append(f"line {line}: {name} argument {k + 1}: C {ca} against Julia {ja}")
        if not compatible(c_ret, jl_type(ret)):
            problems.append(f"line {line}: {name} returns C {c_ret}, the binding says {ret}")
    c_structs = header_structs()
    for jname, fields in julia_structs(julia_text).items():
        c_fields = c_structs[JL_STRUCTS[jname]]
        if [f for f, _ in fields] != [f for f, _ in c_fields]:
            problems.append(f"struct {jname}: fields {[f for f, _ in fields]} against {[f for f, _ in c_fields]} of {JL_STRUCTS[jname]}")
            continue
        for (f, jt), (_, ct) in zip(fields, c_fields):
            if not compatible(ct, jl_type(jt)):
                problems.append(f"struct {jname}.{f}: C {ct} against Julia {jt}")
    defs = header_defines()
    m = re.search(r"const OK, E_ENCODE, E_BADARG, E_HIP, E_NOMEM, E_UNSUPPORTED, E_CAPACITY, E_NCCL = Int32\.\((\d+):(\d+)\)", julia_text)
    names = ["KMERS_OK", "KMERS_E_ENCODE", "KMERS_E_BADARG", "KMERS_E_HIP", "KMERS_E_NOMEM", "KMERS_E_UNSUPPORTED", "KMERS_E_CAPACITY", "KMERS_E_NCCL"]
    if not m or [defs[n] for n in names] != list(range(int(m.group(1)), int(m.group(2)) + 1)):
        problems.append("the status codes of the binding are not those of the header")
    m = re.search(r"const MEM_HOST, MEM_DEVICE, ASYNC, OUT_TUPLES = Int32\((\d+)\), Int32\((\d+)\), Int32\((\d+)\), Int32\((\d+)\)", julia_text)
    if not m or [int(x) for x in m.groups()] != [defs["KMERS_MEM_HOST"], defs["KMERS_MEM_DEVICE"], defs["KMERS_ASYNC"], defs["KMERS_OUT_TUPLES"]]:
        problems.append("the flags of the binding are not those of the header")
    m = re.search(r"const ALPHABET_DNA, ALPHABET_RNA, ALPHABET_SYMBOLS = Int32\((\d+)\), Int32\((\d+)\), Int32\((\d+)\)", julia_text)
    if not m or [int(x) for x in m.groups()] != [defs["KMERS_ALPHABET_DNA"], defs["KMERS_ALPHABET_RNA"], defs["KMERS_ALPHABET_SYMBOLS"]]:
        problems.append("the alphabet constants of the binding are not those of the header")
    return problems


# ---- tests --------------------------------------------------------------------------------------------------------------------
def test_every_ccall_and_struct_of_the_binding_agrees_with_the_header():
    text = open(JULIA).read()
    calls = julia_ccalls(text)
    assert len(calls) >= 45 and len({c[0] for c in calls}) >= 30, len(calls)       # (the parser found them)
    assert set(julia_structs(text)) == set(JL_STRUCTS)
    assert mismatches(text) == []
    # what the binding is there for: the five iterators, their chunk forms, the batches, the sharded path, the pool
    bound = {c[0] for c in calls}
    for name in ("kmers_fw", "kmers_canonical", "kmers_spaced", "kmers_unambiguous", "kmers_batch", "kmers_batch_spaced", "kmers_minhash_batch",
                 "kmers_halo_exchange", "kmers_dev_alloc_role", "kmers_dev_free", "kmers_host_alloc", "kmers_memcpy_d2h_async", "kmers_sync"):
        assert name in bound, name


def test_the_checker_fails_on_a_binding_that_is_wrong():
    text = open(JULIA).read()
    # two arguments of kmers_canonical swapped (seed and flags: UInt64 and Cint)
    swapped = text.replace("seed::UInt64, MEM_HOST::Cint, res::Ref{CResult})::Cint", "MEM_HOST::Cint, seed::UInt64, res::Ref{CResult})::Cint", 1)
    assert swapped != text and any("kmers_canonical argument" in p for p in mismatches(swapped))
    # K and dst_bits are both Cint: a swap the types cannot see -- but an argument dropped is seen
    dropped = text.replace("K::Cint, J::Cint, dst_bits(A)::Cint,", "K::Cint, dst_bits(A)::Cint,", 1)
    assert dropped != text and any("kmers_spaced takes" in p for p in mismatches(dropped))
    # a pointer where a scalar belongs
    ptr = text.replace("n::UInt64, (MEM_DEVICE | ASYNC | OUT_TUPLES)::Cint", "n::Ptr{UInt64}, (MEM_DEVICE | ASYNC | OUT_TUPLES)::Cint", 1)
    assert ptr != text and any("kmers_unambiguous argument 7" in p for p in mismatches(ptr))
    # the wrong pointee
    pointee = text.replace("C_NULL::Ptr{Int64}, 0::UInt64, MEM_HOST::Cint", "C_NULL::Ptr{UInt32}, 0::UInt64, MEM_HOST::Cint", 1)
    assert pointee != text and any("kmers_unambiguous argument 6" in p for p in mismatches(pointee))
    # a struct field moved, a field's type changed
    moved = text.replace("    first_base::UInt64\n    index_origin::UInt64\n", "    index_origin::UInt64\n    first_base::UInt64\n", 1)
    assert moved != text and any("struct CSeq" in p for p in mismatches(moved))
    retyped = text.replace("    halo_words::UInt32\n", "    halo_words::UInt64\n", 1)
    assert retyped != text and any("CShard.halo_words" in p for p in mismatches(retyped))
    # a flag with another value, a function the header does not have
    flag = text.replace("const MEM_HOST, MEM_DEVICE, ASYNC, OUT_TUPLES = Int32(0), Int32(1), Int32(2), Int32(4)",
                        "const MEM_HOST, MEM_DEVICE, ASYNC, OUT_TUPLES = Int32(0), Int32(1), Int32(4), Int32(2)", 1)
    assert flag != text and any("flags" in p for p in mismatches(flag))
    gone = text.replace("LIB.kmers_pool_trim(", "LIB.kmers_arena_reserve(", 1)
    assert gone != text and any("kmers_arena_reserve is not declared" in p for p in mismatches(gone))


JL_SIZES = {"UInt64": 8, "Int64": 8, "UInt32": 4, "Int32": 4, "Ptr{UInt64}": 8}


def test_struct_layouts_against_a_compiled_offsetof_probe(tmp_path):
    """The isbits structs of the binding are laid out like C structs (natural alignment): field offsets and sizes computed from the
    Julia field types must be the ones the C compiler gives include/kmers_hip.h."""
    c_structs, jl = header_structs(), julia_structs(open(JULIA).read())
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "kmers_hip.h"', "int main(void) {"]
    for cname, fields in c_structs.items():
        lines.append(f'    printf("{cname} size %zu\\n", sizeof({cname}));')
        for f, _ in fields:
            lines.append(f'    printf("{cname} {f} %zu\\n", offsetof({cname}, {f}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    measured = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        s, f, v = line.split()
        measured[(s, f)] = int(v)
    assert set(JL_STRUCTS.values()) <= set(c_structs)
    for jname, cname in JL_STRUCTS.items():
        off, align = 0, 1
        for f, t in jl[jname]:
            size = JL_SIZES[t]
            off = -(-off // size) * size
            assert measured[(cname, f)] == off, (jname, f, off, measured[(cname, f)])
            off += size
            align = max(align, size)
        assert measured[(cname, "size")] == -(-off // align) * align, jname


def test_every_device_iterator_has_its_bulk_and_its_chunk_form():
    text = open(JULIA).read()
    m = re.search(r"const GpuIterator\{A, S\} = Union\{(.*?)\}\n", text, flags=re.S)
    assert m
    names = re.findall(r"(\w+)\{A,", m.group(1))
    assert names == ["FwKmers", "FwRvIterator", "CanonicalKmers", "SpacedKmers", "UnambiguousKmers"], names
    for n in names:
        assert re.search(rf"function gpu_collect\(it::{n}\{{", text), f"gpu_collect(::{n})"
        assert re.search(rf"function fill_chunk\(it::{n}\{{", text), f"fill_chunk(::{n})"          # enqueued chunk: iterate()
        assert re.search(rf"fill_chunk_host\(it::{n}\{{", text), f"fill_chunk_host(::{n})"          # the clean prefix before a throw
    # iterate() is written once, over the wrapper, and asks for chunk c + 1 before it hands out chunk c
    assert len(re.findall(r"function Base\.iterate\(g::GPUIterator", text)) == 2
    body = text[text.index("function Base.iterate(g::GPUIterator, st::PipeState)"):]
    assert body.index("kmers_sync") < body.index("enqueue!(g, p, st.slot, st.u0, st.m)") < body.index("x = @inbounds st.buf[st.i]")
    # views of sequences reach the device through kmers_seq.first_base (test/runtests.jl:162)
    assert re.search(r"cseq\(s::LongSubSeq, ::Type\{A\}\) where \{A\} =\s*\n\s*CSeq\(pointer\(s\.data\), length\(s\) % UInt64, \(first\(s\.part\) - 1\) % UInt64", text)
    assert "LongSubSeq{<:NucAlphabet24}" in text


def test_julia_binding_gates_base_collect():
    """ONE Base.collect method, gated by length: below KmersHIP.MIN_BASES[] symbols Kmers.jl's own method runs (the docstring case
    `collect(FwDNAMers{3}("AGCGTATA"))`, src/iterators/FwKmers.jl:14-22, must not become a device call)."""
    text = open(JULIA).read()
    assert len(re.findall(r"(?m)^Base\.collect\(", text)) == 1
    assert "gpu_dispatch(it) && applicable(gpu_collect, it) ? gpu_collect(it) : invoke(collect, Tuple{Any}, it)" in text
    assert 'const MIN_BASES = Ref{Int}(parse(Int, get(ENV, "KMERS_HIP_MIN_BASES", "100000")))' in text


def test_the_maintainers_parity_script_uses_names_the_binding_defines():
    """julia/runtests.jl (KmersHIP against Kmers.jl's own iteration, for a host that has Julia and an MI355X) cannot run here; what can
    be checked: every `KmersHIP.<name>` it uses is defined in julia/KmersHIP.jl with the keywords it is called with, every iterator
    type of GpuIterator is exercised through all three device routes, and its brackets balance."""
    text = open(JULIA).read()
    script = open(os.path.join(ROOT, "julia", "runtests.jl")).read()
    used = set(re.findall(r"KmersHIP\.([A-Za-z_][A-Za-z_0-9!]*)", script)) - {"jl"}     # (the file name in comments and the include)
    assert {"gpu_collect", "gpu", "collect_with_hashes", "sketch", "composition", "minimizers", "collect_batch", "sketch_batch",
            "MIN_BASES", "gpu_dispatch"} <= used, used
    for name in used:
        defined = re.search(rf"(?m)^(function |const |struct |mutable struct )?{re.escape(name)}\b[({{ =]", text)
        assert defined, f"julia/runtests.jl uses KmersHIP.{name}, which julia/KmersHIP.jl does not define"
    # keywords as called
    for call, kw in (("gpu", "chunk"), ("collect_with_hashes", "seed"), ("collect_batch", "hashes"), ("minimizers", "mode")):
        assert re.search(rf"KmersHIP\.{call}\([^\n]*;\s*{kw}\s*=", script), (call, kw)
        sig = re.search(rf"(?ms)^(?:function )?{call}\((.*?)\)(?: where|\s*=)", text)
        assert sig and re.search(rf"(?s);.*\b{kw}\b", sig.group(1)), (call, kw, sig and sig.group(1))
    for it in ("FwDNAMers", "FwRvIterator", "CanonicalDNAMers", "SpacedDNAMers", "UnambiguousDNAMers"):
        assert re.search(rf"same\({it}\{{", script), it       # same() = gpu_collect, Base.collect above the gate, iterate() in chunks
    assert "KmersHIP.gpu_collect(it), collect(it), collect(KmersHIP.gpu(it; chunk = chunk))" in script
    for a, b in ("()", "[]", "{}"):
        assert script.count(a) == script.count(b), (a, script.count(a), script.count(b))
    assert len(re.findall(r"(?m)^@testset |^    @testset |^@testset\b", script)) >= 9
    # (begin / for / function / try / if ... end balance is Julia's to check; the reference's style of one testset per iterator is kept)


if __name__ == "__main__":
    raise SystemExit(pytest.main([__file__, "-q"]))
