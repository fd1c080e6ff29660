"""The product's thread-serial device logic (isaac_aligner_amd/csrc/*.h compiled for the CPU by tests/hostemu) against the
oracle, on seeded synthetic inputs.  This validates the host logic and the kernels' per-cluster functions without a GPU;
the same comparisons run on the real kernels in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import hostemu_lib
from isaac_aligner_amd import options
from parity_util import compare_candidates, compare_records, make_inputs


@pytest.fixture(scope="module")
def emulib():
    return hostemu_lib.load()


def test_exact_sort_is_std_sort(emulib):
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 15, 16, 17, 33, 100, 257, 1000, 5000]:
        for _ in range(10):
            keys = rng.integers(0, max(2, n // 4 + 1), n).astype(np.uint32)
            a, b = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
            emulib.emu_exact_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(a))
            emulib.emu_std_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(b))
            assert (a == b).all(), n
    # adversarial: organ-pipe / sorted / constant inputs drive introsort into its heap-sort fallback
    for n in [64, 300, 2000]:
        for keys in (np.arange(n), np.arange(n)[::-1], np.zeros(n), np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]])):
            keys = np.ascontiguousarray(keys, np.uint32)
            a, b = np.zeros(n, np.uint16), np.zeros(n, np.uint16)
            emulib.emu_exact_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(a))
            emulib.emu_std_sort(hostemu_lib.ptr(keys), n, hostemu_lib.ptr(b))
            assert (a == b).all(), n


def test_default_params_match_oracle(oracle):
    for lens in [(100, 100), (150, 150), (250, 250), (150, 0), (36, 36), (75, 101)]:
        n_reads = 2 if lens[1] else 1
        assert bytes(oracle.default_params(n_reads, lens[0], lens[1])) == bytes(options.default_params(*lens)), lens


@pytest.mark.parametrize("cfg", [
    dict(read_length=150, n_pairs=1500, seed=1),
    dict(read_length=100, n_pairs=1200, seed=5),
    dict(read_length=250, n_pairs=600, seed=9, indel_read_fraction=0.2, indel_max=10),
    dict(read_length=150, read_length2=100, n_pairs=800, seed=13, subst_rate=0.02, n_rate=0.01),
])
def test_fragments_tls_and_records(oracle, emulib, cfg):
    contigs, bcl, _ = make_inputs(**cfg)
    n = len(bcl)
    p = options.default_params(cfg["read_length"], cfg.get("read_length2") or cfg["read_length"])
    ref = oracle.reference(contigs)
    ref.build_index()
    matches, hits = ref.find_matches(p, bcl, n)
    emu = hostemu_lib.Emu(emulib, p, contigs, hits)
    emu.set_matches(matches, n)
    for with_gaps, trim in ((True, True), (False, False)):
        oc, ocig = ref.build_fragments(p, bcl, matches, hits, with_gaps=with_gaps, trim=trim)
        ec, ecig = emu.build_fragments(bcl, n, with_gaps=with_gaps, trim=trim)
        assert not compare_candidates(oc, ocig, ec, ecig)
    otls = ref.determine_tls(p, bcl, matches, hits)
    etls = emu.determine_tls(bcl, n)
    assert otls.astuple() == etls.astuple()
    orec, ocig, _ = ref.select(p, bcl, matches, otls, hits, n_clusters_hint=n)
    erec, ecig = emu.select(bcl, n, etls)
    assert not (erec["reserved"] & 5).any()
    assert not compare_records(orec, ocig, erec, ecig)
    assert emu.counters()["mapq_near_integer"] == 0
