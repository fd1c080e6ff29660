"""Pins the CPU restatement (oracle/) against the reference's own cppunit known-answer vectors (tests/golden/*.json,
generated from /root/reference by tests/golden/make_golden.py)."""
import json
import os

import pytest

from oracle_lib import cigar_string

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def test_bsw_known_answers(oracle):
    g = load("bsw.json")
    bad = []
    for c in g["cases"]:
        cigar, _ = oracle.bsw_align(g["scores"], g["max_read_length"], c["query"], c["database"])
        if list(cigar) != c["cigar"]:
            bad.append((c["name"], c["genome"], cigar_string(cigar), cigar_string(c["cigar"])))
    assert not bad, bad[:10]


def test_bsw_overflow_rule(oracle):
    for m, mm, go, ge, length, throws in load("bsw.json")["overflow"]:
        assert oracle.bsw_check(m, mm, go, ge, length) == throws


def test_bsw_survey_smoke_vectors(oracle):
    # the unmodified reference object gives these on the strings of the (disabled) testCustom
    q = "CTAAGACCCCACACTCTGGGACACCAAGGTGGGAGGATCGCTGGAGCTCAGGAGTTTGAGACCAGCCTGGACAACATGGTGTGACCCTGTCTACAGAAAA"
    d = "AATGCCTCTGGCCTGGGCGTGGGAGTTCATGCTTGTAATCGCATATCGCTAGAGCCCAGGAGTTTGAGACCAGCCTGGACAACATGGTGAAAACCCTCGTTGCTACTAAAAATAC"
    cigar, off = oracle.bsw_align([2, -1, 15, 3], 300, q, d)
    assert (cigar_string(cigar), off) == ("100M", 8)
    cigar, off = oracle.bsw_align([0, -3, 11, 4], 300, q, d)
    assert (cigar_string(cigar), off) == ("27M3I70M", 11)


def test_seed_id(oracle):
    g = load("seed_id.json")
    shifts = {"reverse": 0, "seed": 1, "cluster": 9, "barcode": 40, "tile": 52}
    for v in g["valid"]:
        value = oracle.seed_id(*v)
        assert value == sum(x << shifts[k] for x, k in zip(v, g["order"]))
    for v in g["throws"]:
        with pytest.raises(RuntimeError):
            oracle.seed_id(*v)


def test_simple_indel_aligner(oracle):
    g = load("simple_indel.json")
    for i, c in enumerate(g["cases"]):
        out, cig = oracle.simple_indel_literal(c["read"], c["reference"], c["seeds"], c["left_clip0"], c["right_clip1"])
        f = out[0]
        e = c["expect"]
        got = cigar_string(cig[f["cigar_offset"]:f["cigar_offset"] + f["cigar_length"]])
        assert got == e["getCigarString"], (i, got, e)
        if "getMismatchCount" in e:
            assert f["mismatch_count"] == e["getMismatchCount"], (i, e)
        if "getEditDistance" in e:
            assert f["edit_distance"] == e["getEditDistance"], (i, e)
        if "getFStrandReferencePosition" in e:
            assert f["position"] == e["getFStrandReferencePosition"], (i, e)
        if "leftClipped" in e:
            assert f["low_clipped"] == e["leftClipped"] and f["high_clipped"] == e["rightClipped"], (i, e)


def test_fragment_builder2(oracle):
    g = load("fragment_builder2.json")
    for c in g["cases"]:
        f, cig, cyc = oracle.fragment_builder2_literal(c["read"], c["reference"], c["reverse"], c["position"], c["gapped"])
        e = c["expect"]
        assert cigar_string(cig) == e["getCigarString"], (c["name"], cigar_string(cig))
        assert f["mismatch_count"] == e["getMismatchCount"], c["name"]
        assert f["edit_distance"] == e["getEditDistance"], c["name"]
        assert f["observed_length"] == e["getObservedLength"], c["name"]
        if "getFStrandReferencePosition" in e:
            assert [f["contig_id"], f["position"]] == e["getFStrandReferencePosition"], c["name"]
        if "getStrandReferencePosition" in e:
            assert [f["contig_id"], f["position"]] == e["getStrandReferencePosition"], c["name"]
        if "getMismatchCyclesBegin" in e:
            assert cyc == e["getMismatchCyclesBegin"], c["name"]


def test_threaded_seed_lookup_equals_single_thread(oracle):
    """oracle_find_matches_mt (the CPU-baseline form: cluster ranges on host threads) returns the single-thread result"""
    from parity_util import make_inputs
    from isaac_aligner_amd import options
    contigs, bcl, _ = make_inputs(genome_bases=200000, n_pairs=3001, seed=9)
    ref = oracle.reference(contigs)
    ref.build_index()
    p = options.default_params(150, 150)
    m1, h1 = ref.find_matches(p, bcl, len(bcl))
    m7, h7 = ref.find_matches(p, bcl, len(bcl), n_threads=7)
    assert len(m1) == len(m7) and (m1 == m7).all() and (h1 == h7).all()
