"""Builds libisaac_gpu.so with hipcc for gfx950, in-tree: every csrc/*.hip is a translation unit of its own (kernel families + the
host side), compiled in parallel and linked into one shared library; and bin/isaac-align, the command-line host (host/*.cpp, g++)
that uses nothing but include/isaac_gpu.h and links against that library."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ISAAC_GPU_BUILD_TAG / ISAAC_GPU_BUILD_FLAGS: a variant of the library for A/B measurements (libisaac_gpu_<tag>.so, extra -D flags);
# gpu.load_library() takes it through ISAAC_GPU_LIBRARY
TAG = os.environ.get("ISAAC_GPU_BUILD_TAG", "")
OBJ = os.path.join(HERE, "build" + ("_" + TAG if TAG else ""))
LIB = os.path.join(HERE, "libisaac_gpu%s.so" % ("_" + TAG if TAG else ""))
# -amdgpu-function-calls=false: everything is inlined into the kernels, so that their occupancy targets
# (amdgpu_waves_per_eu in kernels.h) bind the whole call tree and not just the kernel body
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-function-calls=false", "-I", CSRC] + os.environ.get("ISAAC_GPU_BUILD_FLAGS", "").split()


def units():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]


def headers():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "isaac_gpu.h")]


def sources():
    return units() + headers()


def is_stale():
    return not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in sources())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in headers())

    def compile_unit(src):
        obj = os.path.join(OBJ, os.path.splitext(os.path.basename(src))[0] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        cmd = (["hipcc"] + FLAGS if src.endswith(".hip") else ["g++", "-O2", "-std=c++17", "-fPIC", "-Wall"]) + ["-c", src, "-o", obj]      # .cpp: host code on the C ABI
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_unit, units()))
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


HOST = os.path.join(HERE, "host")
HOST_EXE = os.path.join(HERE, "bin", "isaac-align")
SORT_EXE = os.path.join(HERE, "bin", "isaac-sort-reference")
# the programs of host/: plain C++17 on include/isaac_gpu.h; they find the library next to them through their rpath
HOST_PROGRAMS = {HOST_EXE: ["isaac_align.cpp", "align_options.cpp", "fastq_flowcell.cpp"], SORT_EXE: ["isaac_sort_reference.cpp"]}


def build_host(force=False, verbose=False):
    """bin/isaac-align and bin/isaac-sort-reference; returns the path of isaac-align"""
    build(verbose=verbose)
    headers_ = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "isaac_gpu.h"), LIB]
    for exe, names in HOST_PROGRAMS.items():
        srcs = [os.path.join(HOST, n) for n in names]
        if not force and os.path.exists(exe) and all(os.path.getmtime(d) <= os.path.getmtime(exe) for d in srcs + headers_):
            continue
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(HERE, "..", "include"), "-I", HOST] + srcs + \
              ["-L", HERE, "-l" + os.path.basename(LIB)[3:-3], "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-lz", "-o", exe]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST_EXE


if __name__ == "__main__":
    build(force=True, verbose=True)
    if not TAG:
        build_host(force=True, verbose=True)
