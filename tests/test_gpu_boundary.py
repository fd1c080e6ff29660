"""The drop-in seams of the C ABI on the GPU: the sorted-reference index handed in as mask files with a karyotype translation,
a host that uses nothing but include/isaac_gpu.h, and the geometry bench.py runs (one launch per kernel over 1 M clusters)."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from isaac_aligner_amd import abi, options
from parity_util import compare_records, count_record_diffs, make_inputs, sort_matches

pytestmark = pytest.mark.gpu


def gpu_matches_numpy(matches):
    m = matches.cpu().numpy().view(np.uint64).reshape(-1, 2)
    return np.rec.fromarrays([m[:, 0], m[:, 1]], dtype=oracle_lib.MATCH_DTYPE)


def test_load_index_from_mask_files_with_karyotype(torch, oracle):
    """isaac_gpu_load_index: the 64 masks of a table built for contigs (a, b, c) are handed to a context whose contigs are loaded
    in karyotype order (c, a, b).  Every stored position is translated as ReferenceKmer::getTranslatedPosition does
    (MatchFinder.cpp:51-66), checked against the oracle's translateContig, down to the final records."""
    from isaac_aligner_amd import gpu
    contigs, bcl, _ = make_inputs(genome_bases=600000, n_pairs=3000, read_length=150, seed=21, n_contigs=3)
    p = options.default_params(150, 150)
    builder = gpu.Aligner(p, 0, contigs)
    n = builder.build_index()
    index = builder.get_index()
    cut = builder.mask_offsets()
    assert cut[0] == 0 and cut[-1] == n and (np.diff(cut.astype(np.int64)) >= 0).all()
    masks = [index[int(cut[m]):int(cut[m + 1])] for m in range(64)]
    assert all(((mk["kmer"] >> np.uint64(58)) == m).all() for m, mk in enumerate(masks))        # a mask = the k-mer's top 6 bits
    builder.close()
    karyotype = np.array([1, 2, 0], np.uint32)           # stored contig i sits at karyotype position karyotype[i]
    ordered = [None] * 3
    for stored, k in enumerate(karyotype):
        ordered[k] = contigs[stored]
    # reads drawn from the karyotype-ordered genome, so that positions and contig ids of the records refer to that order
    from isaac_aligner_amd import synth
    bcl = synth.make_read_pairs([torch.frombuffer(bytearray(c), dtype=torch.uint8) for c in ordered], 3000, 150, seed=22)[0].numpy()
    al = gpu.Aligner(p, 0, ordered)
    al.load_index(masks, karyotype)
    assert (al.mask_offsets() == cut).all()
    dev_bcl = torch.from_numpy(bcl).to(al.device)
    matches, offsets, hits = al.find_matches(dev_bcl)
    ref = oracle.reference(ordered)
    ref.set_index(index)
    ref.set_karyotype(karyotype)
    om, ohits = ref.find_matches(p, bcl, len(bcl))
    a, b = sort_matches(om), sort_matches(gpu_matches_numpy(matches))
    assert len(a) == len(b) and (a["seed_id"] == b["seed_id"]).all() and (a["location"] == b["location"]).all()
    assert (hits == ohits).all() and hits.all()
    # a translation that is not the identity must have changed something
    plain = gpu.Aligner(p, 0, ordered)
    plain.load_index(masks)
    pm = sort_matches(gpu_matches_numpy(plain.find_matches(dev_bcl)[0]))
    assert len(pm) != len(b) or (pm["location"] != b["location"]).any()
    plain.close()
    al.set_loaded_contigs(hits)
    tls = al.determine_tls(dev_bcl, matches, offsets)
    otls = ref.determine_tls(p, bcl, om, ohits)
    assert otls.astuple() == tls.astuple()
    rec, cig = al.records_to_numpy(*al.select(dev_bcl, matches, offsets, tls))
    orec, ocig, _ = ref.select(p, bcl, om, otls, ohits, n_clusters_hint=len(bcl))
    assert not compare_records(orec, ocig, rec, cig)
    # masks out of global k-mer order are refused (ExactMaskMatcher relies on the order)
    with pytest.raises(gpu.IsaacGpuError):
        al.load_index(list(reversed(masks)))


def test_empty_tile(torch):
    from isaac_aligner_amd import gpu
    contigs, _, _ = make_inputs(genome_bases=100000, n_pairs=10, seed=3)
    al = gpu.Aligner(options.default_params(150, 150), 0, contigs)
    al.build_index()
    bcl = torch.zeros((0, 300), dtype=torch.uint8, device=al.device)
    matches, offsets, hits = al.find_matches(bcl)
    assert matches.shape[0] == 0 and int(offsets[0]) == 0 and not hits.any()


def test_host_without_python_runtime(torch):
    """tests/host_example/align_tile.cpp: find -> template length statistics -> select through the C ABI from a plain C++ program"""
    from test_abi import hostexample_build
    exe = hostexample_build()
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host example: ok" in r.stdout, r.stdout + r.stderr


def test_bench_geometry(torch, oracle):
    """what bench.py launches: 1 048 576 clusters in one chunk (one launch per kernel), on a human-like reference with repeat
    families.  Every record against the oracle, idempotence of the call, and the records of a strided 50 k-cluster sub-batch run
    on their own (chunk-size independence)."""
    from isaac_aligner_amd import gpu, synth
    n = 1 << 20
    genome = synth.make_human_like_genome(100_000_000, seed=5, device="cuda")
    bcl = synth.make_read_pairs(genome, n, 150, seed=6, device="cuda", avoid_gaps=True)[0]
    p = options.default_params(150, 150)
    al = gpu.Aligner(p, 0, genome)
    al.build_index()
    matches, offsets, hits = al.find_matches(bcl)
    al.set_loaded_contigs(hits)
    tls = al.determine_tls(bcl, matches, offsets)
    rec_t, cig_t = al.select(bcl, matches, offsets, tls)
    rec, cig = al.records_to_numpy(rec_t, cig_t)
    counters = al.counters()
    assert counters["large_sums"] > 0 and counters["overflow_clusters"] == 0
    rec2, cig2 = al.records_to_numpy(*al.select(bcl, matches, offsets, tls))
    assert rec.tobytes() == rec2.tobytes() and count_record_diffs(rec, cig, rec2, cig2)[0] == 0
    # the oracle on all of it (all host cores)
    host = bcl.cpu().numpy()
    ref = oracle.reference([c.cpu().numpy().tobytes() for c in genome.contigs])
    ref.set_index(al.get_index())
    om, ohits = ref.find_matches(p, host, n, n_threads=min(64, os.cpu_count() or 1))
    assert (ohits == hits).all()
    otls = ref.determine_tls(p, host, om, ohits)
    assert otls.astuple() == tls.astuple()
    orec, ocig, _ = ref.select(p, host, om, otls, ohits, n_threads=os.cpu_count() or 1, n_clusters_hint=n)
    n_diff, text = count_record_diffs(orec, ocig, rec, cig)
    assert n_diff == 0, "\n".join(text)
    # every 21st cluster as a tile of its own
    pick = torch.arange(0, n, 21, device=bcl.device)
    sub = bcl[pick].contiguous()
    sm, so, _ = al.find_matches(sub)
    srec, scig = al.records_to_numpy(*al.select(sub, sm, so, tls))
    whole = rec.reshape(n, 2)[pick.cpu().numpy()].reshape(-1)
    for f in rec.dtype.names:
        if f not in ("cigar_offset", "cluster_id", "reserved"):
            assert (srec[f] == whole[f]).all(), f


def test_build_fragments_after_smaller_selects(torch):
    """A context that has served an even number of small isaac_gpu_select calls (its second ClusterFragments buffer was the last one
    in use, sized for those calls) is then asked for isaac_gpu_build_fragments on a far larger tile: the candidates must be what a
    fresh context produces (the call must not write through the small buffer)"""
    from isaac_aligner_amd import gpu, synth
    contigs = synth.make_genome(600000, seed=41, n_contigs=2)
    host_contigs = [bytes(c.numpy()) for c in contigs]
    p = options.default_params(150, 150)
    small = [synth.make_read_pairs(contigs, 3000, 150, seed=50 + i)[0].cuda() for i in range(2)]
    large = synth.make_read_pairs(contigs, 200000, 150, seed=60)[0].cuda()

    def candidates(al):
        m, o, _ = al.find_matches(large)
        return al.build_fragments(large, m, o)

    used = gpu.Aligner(p, 0, host_contigs)
    used.build_index()
    m, o, hits = used.find_matches(small[0])
    tls = used.determine_tls(small[0], m, o)
    for b in small:                                        # two chunks: the buffers alternate, the second one ends up current
        m, o, _ = used.find_matches(b)
        used.select(b, m, o, tls)
    got_c, got_g = candidates(used)
    used.close()
    fresh = gpu.Aligner(p, 0, host_contigs)
    fresh.build_index()
    want_c, want_g = candidates(fresh)
    fresh.close()
    assert len(got_c) == len(want_c) > 300000 and got_c.tobytes() == want_c.tobytes() and (got_g == want_g).all()
