"""tests/gpucheck: brute-force GPU witnesses used by the at-scale tests (test infrastructure, never loaded by the product).
build() is called by __graft_entry__.build(); the .so travels to the GPU box with the snapshot."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpucheck")
LIB = os.path.join(HERE, "libgpucheck.so")
SRC = os.path.join(HERE, "kmer_scan.hip")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-unused-result", SRC, "-o", LIB])
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  -- one HIP runtime in the process
        _lib = C.CDLL(build())
    return _lib


def kmer_scan(bases_dev, contig_offsets, queries):
    """bases_dev: uint8 device tensor of the concatenated contigs; queries: uint64 array of packed 32-mers.
    Returns (forward occurrences, sum of their ReferencePosition values, "a forward 32-mer within 1..4 mismatches exists") per query."""
    lib = load()
    offsets = np.ascontiguousarray(contig_offsets, np.uint64)
    q = np.ascontiguousarray(queries, np.uint64)
    count, possum, near = np.zeros(len(q), np.uint64), np.zeros(len(q), np.uint64), np.zeros(len(q), np.uint8)
    rc = lib.gpucheck_kmer_scan(C.c_void_p(bases_dev.data_ptr()), offsets.ctypes.data_as(C.c_void_p), C.c_uint32(len(offsets) - 1), q.ctypes.data_as(C.c_void_p), C.c_uint32(len(q)),
                                count.ctypes.data_as(C.c_void_p), possum.ctypes.data_as(C.c_void_p), near.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("gpucheck_kmer_scan: %d" % rc)
    return count, possum, near.astype(bool)


def reverse_complement(kmers):
    """of packed 32-mers (uint64 array)"""
    k = np.asarray(kmers, np.uint64)
    out = np.zeros_like(k)
    for i in range(32):
        out = (out << np.uint64(2)) | ((k >> np.uint64(2 * i)) & np.uint64(3))
    return ~out
