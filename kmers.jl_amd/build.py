"""Builds libkmers_hip.so (gfx950) in-tree with hipcc.  No other backend exists.

One translation unit per entry-point family of include/kmers_hip.h (csrc/api_common.hpp says which is which), compiled in
parallel and incrementally: an object is rebuilt when its source or any header it includes (transitively) is newer.
`build_variant` links the same objects with some units recompiled under extra -D flags: tuning / diagnostic builds, and the
test build whose one-pass UnambiguousKmers kernel withholds a tile's aggregate (tests/test_gpu_parity.py).
"""
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libkmers_hip.so")
SOURCES = ["iterators_api.hip", "consumers_api.hip", "unambiguous_api.hip", "batch_api.hip", "elementwise_api.hip", "context_api.hip",
           "memory_api.hip", "pool_api.hip", "comm_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-array-bounds"]
# the look-backs of unambiguous_kernel.hpp and scan_kernels.hpp give up instead of hanging the device; this build provokes it (one
# tile / segment never publishes its count) so that the abort / drain path and the host's KMERS_E_HIP are exercised once on hardware
TEST_ABORT_LIB = os.path.join(CSRC, "libkmers_hip_testabort.so")
TEST_ABORT = ("testabort", ["-DKMERS_TEST_ABORT"], ["unambiguous_api.hip", "batch_api.hip"])


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libkmers_hip.so cannot be built")


_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def dependencies(source):
    """The source file and every quoted header it includes, transitively (absolute paths)."""
    seen, todo = set(), [os.path.join(CSRC, source)]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen or not os.path.exists(f):
            continue
        seen.add(f)
        with open(f) as fh:
            for inc in _INCLUDE.findall(fh.read()):
                todo.append(os.path.join(os.path.dirname(f), inc))
    return seen


def _object(source, tag=""):
    return os.path.join(OBJ, os.path.splitext(source)[0] + (f".{tag}" if tag else "") + ".o")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stale(lib=LIB):
    return _newer(lib, set().union(*(dependencies(s) for s in SOURCES)))


def _compile(jobs, verbose):
    """jobs: [(source, object, extra flags)] -- the stale ones, in parallel."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)

    def one(job):
        src, obj, extra = job
        tmp = f"{obj}.{os.getpid()}.tmp"
        cmd = [hipcc()] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", tmp]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.run(cmd, check=True, cwd=CSRC)
            os.replace(tmp, obj)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    if jobs:
        with ThreadPoolExecutor(min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(one, jobs))


def _link(objs, lib, verbose):
    # RCCL is NOT linked: comm_api.hip binds librccl.so.1 with dlopen at the first kmers_comm_* call; the RUNPATH hipcc
    # writes (the ROCm lib directory) lets that dlopen find /opt/rocm's copy in a process that does not already hold one
    tmp = f"{lib}.{os.getpid()}.tmp"  # atomic replace: concurrent builders / loaders never see a partial file
    cmd = [hipcc()] + FLAGS + ["-shared", "-o", tmp] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def build(force=False, verbose=False):
    """Compile the HIP kernels + C ABI for gfx950 and link them with the HIP runtime (RCCL is bound lazily, comm_api.hip).
    Returns the path of the shared library."""
    if not force and not stale():
        # a tree that received the library without its objects (the GPU box: .gpurunignore drops csrc/build/): up to date as it is
        return LIB
    jobs = [(s, _object(s), []) for s in SOURCES if force or _newer(_object(s), dependencies(s))]
    _compile(jobs, verbose)
    objs = [_object(s) for s in SOURCES]
    if force or jobs or _newer(LIB, objs):
        _link(objs, LIB, verbose)
    return LIB


def build_variant(name, defines, units=None, out=None, force=False, verbose=False):
    """libkmers_hip_<name>.so: the product's objects with `units` (default: all) recompiled under the extra flags `defines`."""
    units = list(units or SOURCES)
    out = out or os.path.join(CSRC, f"libkmers_hip_{name}.so")
    if not force and not stale(out):
        return out
    build(verbose=verbose)
    if any(not os.path.exists(_object(s)) for s in SOURCES):  # (the product came without its objects: compile them to link against)
        build(force=True, verbose=verbose)
    jobs = [(s, _object(s, name), list(defines)) for s in units if force or _newer(_object(s, name), dependencies(s))]
    _compile(jobs, verbose)
    objs = [_object(s, name) if s in units else _object(s) for s in SOURCES]
    if force or jobs or _newer(out, objs):
        _link(objs, out, verbose)
    return out


def build_test_abort(force=False, verbose=False):
    name, defines, units = TEST_ABORT
    return build_variant(name, defines, units, TEST_ABORT_LIB, force, verbose)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1] == "variant":  # python -m kmers_jl_amd.build variant <name> -DX ... [unit.hip ...]
        nm = sys.argv[2]
        print(build_variant(nm, [a for a in sys.argv[3:] if a.startswith("-")], [a for a in sys.argv[3:] if a.endswith(".hip")] or None,
                            verbose=True))
    else:
        print(build(force=True, verbose=True))
        print(build_test_abort(verbose=True))
