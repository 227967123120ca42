"""Builds libkmers_hip.so (gfx950) in-tree with hipcc.  No other backend exists."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libkmers_hip.so")
SOURCES = ["kmers_api.hip", "comm_api.hip"]  # iterators + consumers; RCCL communication
HEADERS = ["context.hpp", "device_bits.hpp", "stream_kernel.hpp", "unambiguous_kernel.hpp", "wide_kernel.hpp", "composition_kernel.hpp", "run_kernel.hpp", "ragged_kernels.hpp", "record_sketch_kernel.hpp", "batch_kernels.hpp", "ascii_tables.hpp",
           os.path.join("..", "..", "include", "kmers_hip.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libkmers_hip.so cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile the HIP kernels + C ABI for gfx950 and link them with the HIP runtime and RCCL.
    Returns the path of the shared library."""
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    tag = f"{os.getpid()}.tmp"
    common = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-array-bounds"]
    objs = [os.path.join(CSRC, f"{os.path.splitext(s)[0]}.{tag}.o") for s in SOURCES]

    def compile_one(i):
        cmd = common + ["-c", os.path.join(CSRC, SOURCES[i]), "-o", objs[i]]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    tmp = f"{LIB}.{tag}"  # atomic replace: concurrent builders / loaders never see a partial file
    try:
        with ThreadPoolExecutor(len(SOURCES)) as ex:
            list(ex.map(compile_one, range(len(SOURCES))))
        rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc()))), "lib")
        cmd = common + ["-shared", "-o", tmp] + objs + [f"-L{rocm_lib}", "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
        os.replace(tmp, LIB)
    finally:
        for o in objs + [tmp]:
            if os.path.exists(o):
                os.remove(o)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
