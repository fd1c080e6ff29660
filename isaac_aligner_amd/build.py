"""Builds libisaac_gpu.so (the only native artefact of the product) with hipcc for gfx950, in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libisaac_gpu.so")


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(HERE, "..", "include", "isaac_gpu.h")]


def is_stale():
    return not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in sources())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    # -amdgpu-function-calls=false: everything is inlined into the kernels, so that their occupancy targets
    # (amdgpu_waves_per_eu in isaac_gpu.hip) bind the whole call tree and not just the kernel body
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-value",
           "-mllvm", "-amdgpu-function-calls=false", "-I", CSRC, "-o", LIB, os.path.join(CSRC, "isaac_gpu.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
