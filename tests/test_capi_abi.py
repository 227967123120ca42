"""CPU-side checks of the drop-in boundary: libkmers_hip.so builds for gfx950, loads without a
GPU, exports every symbol include/kmers_hip.h declares, and refuses to compute without a device
(there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def km():
    import kmers_jl_amd
    kmers_jl_amd.build.build()
    return kmers_jl_amd


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "kmers_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kmers_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(km):
    decl = declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(km._capi.SYMBOLS) == decl


def test_library_exports_every_declared_symbol(km):
    lib = C.CDLL(km._capi.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert km._capi.load().kmers_abi_version() == 1


def test_geometry_helpers(km):
    lib = km._capi.load()
    # n_coding_elements (src/kmer.jl:123-125) and iterator length (FwKmers.jl:40-43, SpacedKmers.jl:38-42)
    assert [lib.kmers_words_per_kmer(k, 2) for k in (1, 31, 32, 33, 63, 64, 65)] == [1, 1, 1, 2, 2, 2, 3]
    assert [lib.kmers_words_per_kmer(k, 4) for k in (16, 17, 21, 31, 32, 33)] == [1, 2, 2, 2, 2, 3]
    assert lib.kmers_count(10, 3, 1) == 8 and lib.kmers_count(2, 3, 1) == 0
    assert lib.kmers_count(8, 3, 2) == 3 and lib.kmers_count(11, 3, 3) == 3
    assert lib.kmers_count(10**10, 31, 1) == 10**10 - 30
    assert lib.kmers_supported(4, 2, 31, 1) == 1 and lib.kmers_supported(4, 2, 128, 1) == 1
    # the iterators take kmers of any width (src/kmer.jl:97-111 has no bound on N)
    assert lib.kmers_supported(4, 2, 129, 1) == 1 and lib.kmers_supported(2, 4, 64, 1) == 1 and lib.kmers_supported(2, 4, 65, 1) == 1
    assert lib.kmers_supported(3, 2, 5, 1) == 0 and lib.kmers_supported(4, 3, 5, 1) == 0 and lib.kmers_supported(4, 2, 0, 1) == 0


def test_no_cpu_fallback(km):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(km.KmersError):
        km.Context(0)
    with pytest.raises(km.KmersError):
        km.collect(km.CanonicalDNAMers[5](km.LongDNA[4]("ACGTACGTAC")))


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "kmers.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "kmers_oracle" not in text, f


def test_host_mirror_value_types(km):
    """Kmer / LongSequence packing of the host mirror (layout only, no compute)."""
    assert km.mer("TAGCTAG").data == (0x3272,)                       # test/runtests.jl:906 word
    assert km.mer("UGAUGCA", "r").data == (0x38E4,)
    assert km.as_integer(km.mer("AACT")) == 0x07                     # src/kmer.jl:288-289
    assert km.from_integer(km.DNAAlphabet[2], 3, 0xFF) == km.mer("TTT")  # test/runtests.jl:270
    assert str(km.DNAKmer[36]("TTGCTAGGGATTCGAGGATCCTCTAGAGCGCGGCAC")) == "TTGCTAGGGATTCGAGGATCCTCTAGAGCGCGGCAC"
    s = km.LongDNA[4]("TGAGCWKCATC")
    assert len(s) == 11 and str(s) == "TGAGCWKCATC"
    assert km.mer("AAC") < km.mer("AAG") and not (km.mer("T") < km.mer("A"))


def test_header_is_plain_c_and_example_links(km, tmp_path):
    """include/kmers_hip.h compiles as C99 and plain-C clients link against libkmers_hip.so."""
    import subprocess
    import torch
    csrc = os.path.join(ROOT, "kmers.jl_amd", "csrc")
    for name in ("canonical_hashes", "batch_reads", "sharded_canonical", "resident_pipeline"):
        exe = tmp_path / name
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", name + ".c"), "-L", csrc, "-lkmers_hip",
                        f"-Wl,-rpath,{csrc}", "-o", str(exe)], check=True)
        if not torch.cuda.is_available():
            out = subprocess.run([str(exe)], capture_output=True, text=True)
            assert out.returncode == 2 and "no usable HIP device" in out.stderr  # fails loudly, no fallback


def test_class_pool_logic_on_the_cpu(tmp_path):
    """csrc/class_pool.hpp -- which region classes the chunks of a new block of the device's pool get, pure host code -- under
    AddressSanitizer + UBSan: blocks as pure as the stock allows, the second array of a launch in another class than the first at
    every relative position (an exact small transport problem, not a greedy walk), a lone output's halves in different classes,
    the questions the launchers ask about arrays inside blocks, the books under take / give, 3000 random stocks and partners; the
    cache of freed blocks (which cached block a request takes) and the bound on what the pool holds outside blocks."""
    import subprocess
    exe = tmp_path / "class_pool_check"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "kmers.jl_amd", "csrc"), "-o", str(exe), os.path.join(ROOT, "tests", "c", "class_pool_check.cpp")], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "class_pool_check: ok" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


