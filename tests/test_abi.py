"""CPU-side checks of the product boundary: the library builds, loads and exports every symbol include/isaac_gpu.h declares;
record layouts of the Python mirror, the oracle C API and the device headers agree."""
import ctypes as C
import os
import re

import pytest

import hostemu_lib
import oracle_lib
from isaac_aligner_amd import abi, build, gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    build.build()
    lib = gpu.load_library()
    header = open(os.path.join(ROOT, "include", "isaac_gpu.h")).read()
    declared = set(re.findall(r"\b(isaac_gpu_\w+)\s*\(", header))
    declared.discard("isaac_gpu_ctx")
    assert declared == set(gpu.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_record_layouts_agree():
    lib = hostemu_lib.load()
    assert lib.emu_sizeof(4) == abi.FRAGMENT_DTYPE.itemsize == oracle_lib.RECORD_DTYPE.itemsize == 64
    assert lib.emu_sizeof(6) == C.sizeof(abi.Params) == C.sizeof(oracle_lib.Params)
    assert abi.CANDIDATE_DTYPE == oracle_lib.CANDIDATE_DTYPE
    assert [f for f in abi.FRAGMENT_DTYPE.names] == [f for f in oracle_lib.RECORD_DTYPE.names]
    assert lib.emu_sizeof(0) == 64


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from isaac_aligner_amd import options
    with pytest.raises(gpu.IsaacGpuError):
        gpu.Aligner(options.default_params(150, 150), 0)


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "isaac_aligner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "liboracle" not in text and "oracle/" not in text.replace("oracle/_ref", ""), f


def test_header_is_plain_c_and_a_c_host_links():
    """include/isaac_gpu.h compiles as C99; a C translation unit that references every entry point links against the library"""
    import subprocess
    import tempfile
    build.build()
    lib_dir = os.path.join(ROOT, "isaac_aligner_amd")
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, "abi_c.o")
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fPIC", "-I", os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "tests", "host_example", "abi_c.c"), "-o", obj])
        so = os.path.join(tmp, "libabi_c.so")
        subprocess.check_call(["gcc", "-shared", "-o", so, obj, "-L", lib_dir, "-lisaac_gpu", "-Wl,-rpath," + lib_dir])
        gpu.load_library()                       # the HIP runtime torch ships is resident before the probe library pulls the product in
        probe = C.CDLL(so)
        probe.isaac_gpu_entry_point_count.restype = C.c_size_t
        assert probe.isaac_gpu_entry_point_count() == len(gpu.EXPORTS)
        assert probe.isaac_gpu_struct_sizes_ok() == 1


def test_cpp_host_example_builds():
    """tests/host_example/align_tile.cpp (a host on the C ABI alone) compiles and links; test_gpu_parity runs it on the GPU box"""
    import subprocess
    exe = hostexample_build()
    assert os.path.exists(exe)
    # without a GPU the example must fail loudly in isaac_gpu_create, not fall back to anything
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "100"], capture_output=True, text=True)
        assert r.returncode != 0 and "isaac_gpu_create" in r.stderr


def hostexample_build():
    import subprocess
    build.build()
    lib_dir = os.path.join(ROOT, "isaac_aligner_amd")
    src = os.path.join(ROOT, "tests", "host_example", "align_tile.cpp")
    exe = os.path.join(ROOT, "tests", "host_example", "align_tile")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(build.LIB)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-L", lib_dir, "-lisaac_gpu", "-Wl,-rpath," + lib_dir,
                               "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe])
    return exe
