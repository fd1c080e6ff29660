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
