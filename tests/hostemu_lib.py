"""ctypes binding of tests/hostemu/libhostemu.so: the product's thread-serial device headers compiled for the CPU.
TEST INFRASTRUCTURE ONLY (a debugging harness for the GPU-less build container)."""
import ctypes as C
import os
import subprocess

import numpy as np

from isaac_aligner_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIR = os.path.join(ROOT, "tests", "hostemu")


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def load():
    so = os.path.join(DIR, "libhostemu.so")
    srcs = [os.path.join(DIR, "hostemu.cpp")] + [os.path.join(ROOT, "isaac_aligner_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "isaac_aligner_amd", "csrc")) if f.endswith(".h")]
    srcs.append(os.path.join(ROOT, "include", "isaac_gpu.h"))
    tables = os.path.join(ROOT, "isaac_aligner_amd", "csrc", "deflate_tables.cpp")      # host code of the product: the Huffman tables of the device deflate
    srcs.append(tables)
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(DIR, "hostemu.cpp"), tables])
    lib = C.CDLL(so)
    lib.emu_last_error.restype = C.c_char_p
    lib.emu_create.restype = C.c_void_p
    return lib


class Emu:
    def __init__(self, lib, params, contigs, loaded=None):
        self.lib = lib
        self.contigs = [bytes(c) for c in contigs]
        self.bases = np.frombuffer(b"".join(self.contigs), np.uint8).copy()
        self.offsets = np.zeros(len(contigs) + 1, np.uint64)
        self.offsets[1:] = np.cumsum([len(c) for c in self.contigs])
        self.loaded = None if loaded is None else np.ascontiguousarray(loaded, np.uint8)
        self.h = C.c_void_p(lib.emu_create(C.byref(params), ptr(self.bases), ptr(self.offsets), C.c_uint32(len(contigs)), ptr(self.loaded)))
        if not self.h:
            raise RuntimeError(lib.emu_last_error().decode())

    def check(self, rc):
        if rc:
            raise RuntimeError(self.lib.emu_last_error().decode())

    def set_matches(self, matches, n_clusters):
        m = np.ascontiguousarray(matches, abi.MATCH_DTYPE)
        self.check(self.lib.emu_set_matches(self.h, ptr(m), C.c_uint64(len(m)), C.c_uint32(n_clusters)))

    def build_fragments(self, bcl, n_clusters, with_gaps=True, trim=True):
        cap = n_clusters * 64 + 1024
        out = np.zeros(cap, abi.CANDIDATE_DTYPE)
        cig = np.zeros(cap * 8, np.uint32)
        n, nc = C.c_uint64(), C.c_uint64()
        self.check(self.lib.emu_build_fragments(self.h, ptr(bcl), C.c_uint32(n_clusters), int(with_gaps), int(trim), ptr(out), C.c_uint64(cap), C.byref(n), ptr(cig), C.c_uint64(len(cig)), C.byref(nc)))
        return out[:n.value].copy(), cig[:nc.value].copy()

    def determine_tls(self, bcl, n_clusters):
        t = abi.Tls()
        self.check(self.lib.emu_determine_tls(self.h, ptr(bcl), C.c_uint32(n_clusters), C.byref(t)))
        return t

    def select(self, bcl, n_clusters, tls, tile=0, n_reads=2):
        rec = np.zeros(n_clusters * n_reads, abi.FRAGMENT_DTYPE)
        cig = np.zeros(n_clusters * n_reads * abi.MAX_CIGAR_OPS, np.uint32)
        self.check(self.lib.emu_select(self.h, ptr(bcl), C.c_uint32(n_clusters), C.c_uint32(tile), C.byref(tls), ptr(rec), ptr(cig)))
        return rec, cig

    def counters(self):
        c = abi.Counters()
        self.lib.emu_get_counters(self.h, C.byref(c))
        return c.asdict()
