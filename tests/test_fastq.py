"""FASTQ text -> BCL bytes (the data format on the input side of the path).

CPU part: the oracle's restatement of io::FastqReader on hand-made records (the reference has no unit test for its reader, so
these pin the restatement against the parser's documented behaviour: lib/io/FastqReader.cpp:120-283, FastqReader.hh:144-210).
GPU part (-m gpu): isaac_gpu_fastq_to_bcl against the oracle, byte for byte, including piecewise feeding and the errors."""
import numpy as np
import pytest

from isaac_aligner_amd import options, synth
from parity_util import make_inputs

RECORDS = b"@r1\nACGTN\n+\nIIII#\n@r2\r\nacgtn\r\n+r2\r\n!!!!!\r\n\n\n@r3\nAAAAA\n+\n+++++"


def test_oracle_reader_semantics(oracle):
    rc, bcl, n, consumed, _ = oracle.fastq_to_bcl(RECORDS, 5)
    assert rc == 0 and n == 3 and consumed == len(RECORDS)
    q40 = 40 << 2
    assert bcl[0].tolist() == [q40 | 0, q40 | 1, q40 | 2, q40 | 3, 0]          # N -> 0 whatever its quality
    assert bcl[1].tolist() == [0, 1, 2, 3, 0]                                  # lower case, quality 0 ('!'), CRLF, '+' with the name
    assert bcl[2].tolist() == [10 << 2] * 5                                    # a quality line may start with '+'; no final newline
    # a piece of a longer file: the unterminated last record waits for the next piece
    rc, _, n, consumed, _ = oracle.fastq_to_bcl(RECORDS, 5, final=False)
    assert rc == 0 and n == 2 and RECORDS[consumed:].lstrip(b"\r\n").startswith(b"@r3")
    # max_clusters stops after the record
    rc, _, n, consumed, _ = oracle.fastq_to_bcl(RECORDS, 5, max_clusters=1)
    assert rc == 0 and n == 1 and RECORDS[:consumed].endswith(b"IIII#")
    # reads longer than the configured length are cut, shorter ones are an error unless variable length is allowed (N padding)
    rc, bcl, n, _, _ = oracle.fastq_to_bcl(RECORDS, 3)
    assert rc == 0 and n == 3 and bcl[0, :3].tolist() == [q40, q40 | 1, q40 | 2]
    rc, _, n, _, err = oracle.fastq_to_bcl(RECORDS, 7)
    assert rc == 2 and n == 0 and err == 0
    rc, bcl, n, _, _ = oracle.fastq_to_bcl(RECORDS, 7, allow_variable_length=True)
    assert rc == 0 and n == 3 and bcl[2].tolist() == [10 << 2] * 5 + [0, 0]


def test_oracle_reader_errors(oracle):
    bad_base = b"@a\nACXT\n+\nIIII\n"
    rc, _, n, _, err = oracle.fastq_to_bcl(bad_base, 4)
    assert rc == 1 and n == 0 and err == bad_base.index(b"X")
    bad_quality = b"@a\nACGT\n+\nIIII\n@b\nACGT\n+\nII~I\n"          # '~' = 93 > 63
    rc, _, n, _, err = oracle.fastq_to_bcl(bad_quality, 4)
    assert rc == 1 and n == 1 and err == bad_quality.rindex(b"ACGT") + 2
    no_plus = b"@a\nACGT\nIIII\n@b\nACGT\n+\nIIII\n"
    rc, _, n, _, err = oracle.fastq_to_bcl(no_plus, 4)
    assert rc == 1 and n == 0 and err == no_plus.index(b"IIII")
    truncated = b"@a\nACGT\n+\nIIII\n@b\nAC"
    assert oracle.fastq_to_bcl(truncated, 4)[0] == 1                                  # end of file inside a record
    rc, _, n, consumed, _ = oracle.fastq_to_bcl(truncated, 4, final=False)            # ... or just the end of this piece
    assert rc == 0 and n == 1 and truncated[consumed:].lstrip(b"\n").startswith(b"@b")
    zero_length = b"@a\n+\n@b\nACGT\n+\nIIII\n"                                        # "special case for zero-length reads"
    rc, bcl, n, _, _ = oracle.fastq_to_bcl(zero_length, 4, allow_variable_length=True)
    assert rc == 0 and n == 2 and bcl[0].tolist() == [0, 0, 0, 0] and bcl[1, 0] == 40 << 2
    assert oracle.fastq_to_bcl(b"\n\r\n\n", 4)[2:4] == (0, 4)                         # nothing but newlines


def _gpu_vs_oracle(al, oracle, text, read_index, read_length, stride, offset, **kw):
    from isaac_aligner_amd.gpu import IsaacGpuError
    rc, obcl, on, oconsumed, oerr = oracle.fastq_to_bcl(text, read_length, cluster_stride=stride, offset=offset, **kw)
    try:
        gbcl, gn, gconsumed = al.fastq_to_bcl(text, read_index, **kw)
        grc, gerr = 0, 0
    except IsaacGpuError as e:
        grc, gn, gerr, gbcl, gconsumed = e.code, e.n_clusters, e.error_offset, None, None
    assert (grc == 0) == (rc == 0) and gn == on, (grc, rc, gn, on)
    if rc == 0:
        assert gconsumed == oconsumed
        assert (gbcl[:gn].cpu().numpy()[:, offset:offset + read_length] == obcl[:on, offset:offset + read_length]).all()
    else:
        assert gerr == oerr and grc == {1: 6, 2: 7}[rc]
    return gn


@pytest.mark.gpu
def test_gpu_fastq_parity(torch, oracle):
    from isaac_aligner_amd import gpu
    contigs, bcl, _ = make_inputs(genome_bases=200000, n_pairs=3000, read_length=150, read_length2=100, seed=21)
    p = options.default_params(150, 100)
    al = gpu.Aligner(p, 0, contigs)
    # round trip of a synthetic tile, both reads, Unix and DOS line ends, '+' line with and without the name
    for read_index, (off, length) in enumerate(((0, 150), (150, 100))):
        for newline, plus_header in ((b"\n", False), (b"\r\n", True)):
            text = synth.bcl_to_fastq(bcl, off, length, newline=newline, plus_header=plus_header)
            n = _gpu_vs_oracle(al, oracle, text, read_index, length, 250, off)
            assert n == len(bcl)
            # piecewise: cut anywhere, feed the rest from where the converter stopped
            cut = len(text) // 3 + 17
            g1, n1, consumed = al.fastq_to_bcl(text[:cut], read_index, final=False)
            g2, n2, _ = al.fastq_to_bcl(text[consumed:], read_index)
            whole, nw, _ = al.fastq_to_bcl(text, read_index)
            assert n1 + n2 == nw == len(bcl)
            joined = torch.cat([g1[:n1], g2[:n2]])[:, off:off + length]
            assert (joined == whole[:nw, off:off + length]).all()
    # hand-made records at read length 36 (the shortest the seed layout takes): every reader rule and every error
    L = 36
    seq, qual = b"ACGTN" * 7 + b"A", b"IIII#" * 7 + b"I"
    good = b"@r1\n" + seq + b"\n+\n" + qual + b"\n@r2\r\n" + seq.lower() + b"\r\n+r2\r\n" + b"!" * L + b"\r\n\n\n@r3\n" + b"A" * L + b"\n+\n" + b"+" * L
    pl = options.default_params(L, L)
    all_ = gpu.Aligner(pl, 0, contigs)
    for kw in ({}, {"final": False}, {"max_clusters": 1}, {"max_clusters": 2, "final": False}):
        _gpu_vs_oracle(all_, oracle, good, 0, L, 2 * L, 0, **kw)
    rec = lambda name, s_, q_: b"@" + name + b"\n" + s_ + b"\n+\n" + q_ + b"\n"
    ok = rec(b"a", seq, qual)
    cases = [ok + rec(b"b", seq[:10] + b"X" + seq[11:], qual),                       # not a base
             ok + rec(b"b", seq, qual[:20] + b"~" + qual[21:]),                      # quality 93
             b"@a\n" + seq + b"\n" + qual + b"\n" + ok,                              # '+' line missing
             ok + b"@b\n" + seq[:7],                                                 # the text ends inside a record
             b"@a\n+\n" + ok,                                                        # zero-length read
             b"\n\r\n\n",                                                            # nothing but newlines
             rec(b"a", seq[:30], qual[:30]) + ok,                                    # short read
             rec(b"a", seq + b"ACGT", qual + b"IIII") + ok,                          # long read: cut
             ok + b"@b\n" + seq + b"\n+", ok + b"@b\n" + seq + b"\n+\n", ok + b"@b",  # the text ends at every stage of a record
             ok + b"@b\n" + seq + b"\n+\n" + qual[:12]]                               # qualities shorter than the read, no newline
    for text in cases:
        for kw in ({}, {"final": False}, {"allow_variable_length": True}):
            _gpu_vs_oracle(all_, oracle, text, 1, L, 2 * L, L, **kw)


def test_fastq_tile_rule():
    """FastqSeedSource's tile breakdown (FastqDataSource.cpp:82-84,153-173): product (host code of the C ABI) against the oracle's restatement
    and against the numbers the rule gives by hand"""
    import oracle_lib
    from isaac_aligner_amd import gpu
    o = oracle_lib.load()
    # 2x150 with --seeds auto: 8 seeds -> 5 000 000 clusters per tile; a load of 12 M clusters is tiles 1, 2 (full) and 3 (2 M)
    assert gpu.fastq_tiles(12_000_000, 8) == ([(1, 5_000_000), (2, 5_000_000), (3, 2_000_000)], 4)
    # a whole number of tiles: no empty tile at the end; numbering goes on in the lane's next load
    assert gpu.fastq_tiles(10_000_000, 8, first_tile=4) == ([(4, 5_000_000), (5, 5_000_000)], 6)
    # --clusters-at-a-time below the seed bound caps the tile size
    assert gpu.fastq_tiles(2_500_000, 8, clusters_at_a_time=1_000_000) == ([(1, 1_000_000), (2, 1_000_000), (3, 500_000)], 4)
    assert gpu.fastq_tiles(0, 8) == ([], 1)
    rng = np.random.default_rng(5)
    for _ in range(300):
        loaded = int(rng.integers(0, 60_000_000)); seeds = int(rng.integers(1, 17)); at = int(rng.choice([0, 200_000, 3_000_000, 50_000_000])); first = int(rng.integers(1, 100))
        assert gpu.fastq_tiles(loaded, seeds, at, first) == o.fastq_tiles(loaded, seeds, at, first)

