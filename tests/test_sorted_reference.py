"""sorted-reference.xml reader / writer (isaac_gpu_sorted_reference_parse / _format) pinned by the document and the asserted values
of the reference's own test, reference/cppunit/testSortedReferenceXml.cpp (tests/golden/sorted_reference.json); and, on the GPU,
the round trip of a built table through mask files and the XML (isaac_gpu_save_sorted_reference / isaac_gpu_load_sorted_reference)."""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from isaac_aligner_amd import abi, options, sorted_reference as sr

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sorted_reference.json")))
TEXT = ("name", "file", "bam_sq_as", "bam_sq_ur", "bam_m5")


def check_content(contigs, masks, with_masks=True):
    assert len(contigs) == len(GOLDEN["contigs"])                      # checkContigs
    for c, want in zip(contigs, GOLDEN["contigs"]):
        for k, v in want.items():
            got = getattr(c, k)
            assert (got.decode() if k in TEXT else got) == v, (k, got, v)
    if with_masks:                                                     # checkMasks
        m32 = [m for m in masks if m.seed_length == 32]
        assert len(m32) == GOLDEN["masks"]["count"]
        assert m32[-1].file.decode() == GOLDEN["masks"]["last_file"]
        assert m32[-1].mask_width == GOLDEN["masks"]["mask_width"] == m32[0].mask_width


def test_reader_reproduces_the_reference_test():
    contigs, masks, version = sr.parse(GOLDEN["xml"])
    check_content(contigs, masks)
    assert version == 3                                               # bumped to CURRENT_REFERENCE_FORMAT_VERSION on load
    assert [m.kmers for m in masks] == [107990454, 40727835, 56150179] and [m.mask for m in masks] == [0, 1, 2]
    # an independent XML parser sees the same contigs
    root = ET.fromstring(GOLDEN["xml"])
    assert [c.find("Name").text for c in root.find("Contigs")] == [c.name.decode() for c in contigs]


def test_writer_round_trip():
    """testWriter and testContigsOnly of the reference's suite"""
    contigs, masks, _ = sr.parse(GOLDEN["xml"])
    text = sr.format(contigs, masks)
    ET.fromstring(text)                                               # well formed
    check_content(*sr.parse(text)[:2])
    only = sr.format(contigs, [])
    c2, m2, _ = sr.parse(only)
    check_content(c2, m2, with_masks=False)
    assert not m2 and "Permutations" not in only


def test_reader_errors_as_the_reference_raises_them():
    bad_version = GOLDEN["xml"].replace("<FormatVersion>2</FormatVersion>", "<FormatVersion>7</FormatVersion>")
    with pytest.raises(sr.FormatError, match="Unexpected sorted reference FormatVersion: 7. FormatVersion must be in range \\[2,3\\]"):
        sr.parse(bad_version)
    with pytest.raises(sr.FormatError, match="Only ABCD permutation masks are supported"):
        sr.parse(GOLDEN["xml"].replace('Name="ABCD"', 'Name="BCDA"'))
    twice = GOLDEN["xml"].replace("</Masks>", '</Masks><Masks Width="1"></Masks>')
    with pytest.raises(sr.FormatError, match="Multiple Masks elements with same SeedLength are not allowed"):
        sr.parse(twice)
    with pytest.raises(sr.FormatError):
        sr.parse(GOLDEN["xml"][:len(GOLDEN["xml"]) // 2])
    with pytest.raises(sr.FormatError):
        sr.parse(GOLDEN["xml"].replace("<Total>107990454</Total>", "<Total>many</Total>"))
    # numbers in attributes are validated like numbers in elements (the reference's reader throws on lexical_cast)
    first_position = GOLDEN["xml"].index('Position="')
    with pytest.raises(sr.FormatError, match="Position"):
        sr.parse(GOLDEN["xml"][:first_position] + 'Position="abc' + GOLDEN["xml"][first_position + len('Position="'):])
    with pytest.raises(sr.FormatError, match="Mask"):
        sr.parse(GOLDEN["xml"].replace('Mask="1"', 'Mask="x1"', 1))


def test_reader_takes_cdata_sections():
    name = GOLDEN["contigs"][0]["name"]
    text = GOLDEN["xml"].replace("<Name>%s</Name>" % name, "<Name><![CDATA[%s]]></Name>" % name, 1)
    assert text != GOLDEN["xml"]
    check_content(*sr.parse(text)[:2])


@pytest.mark.gpu
def test_table_round_trip_through_mask_files(torch, tmp_path):
    """build -> isaac_gpu_save_sorted_reference (64 *.dat + sorted-reference.xml) -> a fresh context loads it with
    isaac_gpu_load_sorted_reference, contigs in a permuted karyotype order: same table, same cuts, translated matches"""
    from isaac_aligner_amd import gpu
    from parity_util import make_inputs, sort_matches
    contigs, bcl, _ = make_inputs(genome_bases=500000, n_pairs=2000, read_length=150, seed=31, n_contigs=3)
    p = options.default_params(150, 150)
    a = gpu.Aligner(p, 0, contigs)
    n = a.build_index()
    index, cuts = a.get_index(), a.mask_offsets()
    karyotype = [2, 0, 1]
    meta = []
    position = 0
    for i, c in enumerate(contigs):
        m = sr.Contig()
        m.genomic_position, m.index, m.karyotype_index, m.name, m.file = position, i, karyotype[i], b"contig%d" % i, b"genome.fa"
        m.offset, m.size, m.total_bases, m.acgt_bases = position + 9 * (i + 1), len(c) + len(c) // 70, len(c), sum(c.count(b) for b in b"ACGT")
        position += len(c)
        meta.append(m)
    a.save_sorted_reference(str(tmp_path), "genome.fa", meta)
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".dat"))
    assert len(files) == 64 and files[5] == "genome.fa-32mer-6bit-ABCD-05.dat"
    assert sum(os.path.getsize(tmp_path / f) for f in files) == 16 * n
    on_disk = np.concatenate([np.fromfile(tmp_path / f, abi.REFERENCE_KMER_DTYPE) for f in files])
    assert on_disk.tobytes() == index.tobytes()
    xml_contigs, xml_masks, _ = sr.parse(open(tmp_path / "sorted-reference.xml").read())
    assert [m.kmers for m in xml_masks] == list(np.diff(cuts.astype(np.int64))) and all(m.mask_width == 6 and m.seed_length == 32 for m in xml_masks)
    assert [(c.index, c.karyotype_index, c.name) for c in xml_contigs] == [(i, karyotype[i], b"contig%d" % i) for i in range(3)]
    ordered = [None] * 3
    for stored, k in enumerate(karyotype):
        ordered[k] = contigs[stored]
    b = gpu.Aligner(p, 0, ordered)
    b.load_sorted_reference(str(tmp_path / "sorted-reference.xml"))
    assert b.get_index().tobytes() == index.tobytes() and (b.mask_offsets() == cuts).all()
    c = gpu.Aligner(p, 0, ordered)
    c.load_index([index[int(cuts[m]):int(cuts[m + 1])] for m in range(64)], karyotype)
    dev = torch.from_numpy(bcl).to(b.device)
    mb = b.find_matches(dev)[0].cpu().numpy()
    mc = c.find_matches(dev)[0].cpu().numpy()
    assert mb.shape == mc.shape and (np.sort(mb.view(np.uint64).reshape(-1, 2), axis=0) == np.sort(mc.view(np.uint64).reshape(-1, 2), axis=0)).all()
    with pytest.raises(gpu.IsaacGpuError):
        b.load_sorted_reference(str(tmp_path / "missing.xml"))
    # <Index> / <KaryotypeIndex> that are not permutations are refused instead of silently translating a contig to 0
    text = open(tmp_path / "sorted-reference.xml").read()
    assert "<KaryotypeIndex>2</KaryotypeIndex>" in text
    open(tmp_path / "duplicate.xml", "w").write(text.replace("<KaryotypeIndex>2</KaryotypeIndex>", "<KaryotypeIndex>1</KaryotypeIndex>"))
    with pytest.raises(gpu.IsaacGpuError, match="permutation"):
        b.load_sorted_reference(str(tmp_path / "duplicate.xml"))
