"""Synthetic references and read pairs: there is no network and no genome in the image, so every
configuration runs on data synthesised here.  torch is used for the array arithmetic only, so the same code makes small
inputs on the CPU for the parity tests and full-size inputs directly in HBM for bench.py.

Reads are emitted in the reference's in-memory input encoding, "BCL bytes" (include/io/FastqReader.hh:144-210 as converted
from FASTQ: byte = base | quality << 2, 0 for N), cluster-major: [n_clusters, len(read1) + len(read2)].
"""
import torch

_ASCII = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8)  # A C G T N


def make_genome(n_bases, seed=1, device="cpu", n_contigs=1, repeat_families=True):
    """Random ACGT contigs with the features the hot path reacts to:
       * interspersed repeat families with per-copy divergence (creates k-mers with neighbours and 2..9-copy seeds),
       * a family with 12..40 exact copies (>= repeat threshold 10 at lookup time -> TooManyMatch records),
       * a short element present > 1000 times (stored as a single TooManyMatch index entry, ReferenceSorter.cpp:201-222),
       * runs of N.
    Returns a list of uint8 tensors (ASCII)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    per = n_bases // n_contigs
    contigs = []
    for c in range(n_contigs):
        n = per if c + 1 < n_contigs else n_bases - per * (n_contigs - 1)
        codes = torch.randint(0, 4, (n,), generator=g, dtype=torch.uint8)
        if repeat_families and n >= 20000:
            def paste(src, copies, length, divergence):
                for _ in range(copies):
                    dst = int(torch.randint(0, n - length, (1,), generator=g))
                    seg = src.clone()
                    if divergence > 0:
                        mut = torch.rand(length, generator=g) < divergence
                        seg[mut] = (seg[mut] + torch.randint(1, 4, (int(mut.sum()),), generator=g, dtype=torch.uint8)) % 4
                    codes[dst:dst + length] = seg
            n_fam = max(2, n // 100000)
            for f in range(n_fam):
                length = int(torch.randint(200, 800, (1,), generator=g))
                src = torch.randint(0, 4, (length,), generator=g, dtype=torch.uint8)
                paste(src, int(torch.randint(2, 9, (1,), generator=g)), length, 0.02 if f % 2 else 0.0)
            src = torch.randint(0, 4, (300,), generator=g, dtype=torch.uint8)
            paste(src, int(torch.randint(12, 40, (1,), generator=g)), 300, 0.0)
            if n >= 200000:
                src = torch.randint(0, 4, (40,), generator=g, dtype=torch.uint8)
                paste(src, 1100, 40, 0.0)
            for _ in range(max(1, n // 500000)):
                length = int(torch.randint(50, 2000, (1,), generator=g))
                dst = int(torch.randint(0, n - length, (1,), generator=g))
                codes[dst:dst + length] = 4
        contigs.append(_ASCII[codes.long()].to(device))
    return contigs


def make_read_pairs(contigs, n_pairs, read_length=150, seed=2, device=None, insert_mean=350.0, insert_sd=50.0,
                    subst_rate=0.003, indel_read_fraction=0.03, indel_max=5, n_rate=0.001, low_quality_tail_fraction=0.1,
                    random_pair_fraction=0.002, read_length2=None, avoid_gaps=False):
    """FR-oriented pairs drawn uniformly from the contigs; returns (bcl uint8 [n_pairs, L1+L2], truth dict).
    indel_read_fraction = fraction of reads carrying one indel of 1..indel_max bases (0.02 %/base at 150 bp ~ 3 %).
    contigs: list of uint8 tensors or a Genome (used in place).  avoid_gaps: fragments that touch a run of N are drawn again
    (a sequencer produces no reads from assembly gaps)."""
    device = device or contigs[0].device
    L1 = read_length
    L2 = read_length2 or read_length
    g = torch.Generator(device=device).manual_seed(seed)
    lens = torch.tensor([c.numel() for c in contigs], dtype=torch.float64)
    genome = contigs.bases.to(device) if isinstance(contigs, Genome) else torch.cat([c.to(device) for c in contigs])
    starts = torch.zeros(len(contigs) + 1, dtype=torch.long)
    starts[1:] = torch.cumsum(lens, 0).long()
    starts = starts.to(device)

    def draw(n):
        # contig choice proportional to length, fragment inside the contig
        contig = torch.multinomial(lens.float().to(device), n, replacement=True, generator=g)
        clen = lens.long().to(device)[contig]
        insert = (torch.randn(n, generator=g, device=device) * insert_sd + insert_mean).round().long()
        insert = insert.clamp(min=max(L1, L2) + 10)
        insert = torch.minimum(insert, clen - 2 * indel_max - 2)
        frag_start = (torch.rand(n, generator=g, device=device, dtype=torch.float64) * (clen - insert - 2 * indel_max).double()).long()
        return contig, clen, insert, frag_start

    contig, clen, insert, frag_start = draw(n_pairs)
    if avoid_gaps:
        for _ in range(8):
            probe = torch.linspace(0.0, 1.0, 12, device=device).unsqueeze(0)
            at = (starts[contig] + frag_start).unsqueeze(1) + (probe * (insert + indel_max).unsqueeze(1).float()).long()
            bad = (genome[at.clamp(max=genome.numel() - 1)] == 78).any(1)
            n_bad = int(bad.sum())
            if not n_bad:
                break
            c2, l2, i2, f2 = draw(n_bad)
            contig[bad], clen[bad], insert[bad], frag_start[bad] = c2, l2, i2, f2
    flip = torch.rand(n_pairs, generator=g, device=device) < 0.5          # fragment taken from the reverse strand
    comp = torch.full((256,), 78, dtype=torch.uint8, device=device)
    comp[65], comp[67], comp[71], comp[84] = 84, 71, 67, 65

    def one_read(L, left_pos, reverse):
        """read bases (ASCII) of the forward-strand window starting at left_pos, with substitutions and <= 1 indel, then
        reverse-complemented where `reverse`"""
        i = torch.arange(L, device=device).unsqueeze(0).expand(n_pairs, L)
        has_indel = torch.rand(n_pairs, generator=g, device=device) < indel_read_fraction
        is_ins = torch.rand(n_pairs, generator=g, device=device) < 0.5
        ilen = torch.randint(1, indel_max + 1, (n_pairs,), generator=g, device=device)
        ipos = torch.randint(20, L - 20 - indel_max, (n_pairs,), generator=g, device=device)
        ilen = torch.where(has_indel, ilen, torch.zeros_like(ilen))
        ip, il = ipos.unsqueeze(1), ilen.unsqueeze(1)
        ins = (is_ins & has_indel).unsqueeze(1)
        dele = (~is_ins & has_indel).unsqueeze(1)
        shift = torch.where(dele & (i >= ip), il, torch.zeros_like(i)) - torch.where(ins & (i >= ip + il), il, torch.zeros_like(i))
        shift = shift - torch.where(ins & (i >= ip) & (i < ip + il), i - ip, torch.zeros_like(i))  # inserted bases re-read one position (then randomised)
        idx = (starts[contig] + left_pos).unsqueeze(1) + i + shift
        bases = genome[idx]
        inserted = ins & (i >= ip) & (i < ip + il)
        rnd = _ASCII.to(device)[torch.randint(0, 4, (n_pairs, L), generator=g, device=device)]
        bases = torch.where(inserted, rnd, bases)
        sub = torch.rand(n_pairs, L, generator=g, device=device) < subst_rate
        code = torch.zeros_like(bases)
        code[bases == 67], code[bases == 71], code[bases == 84] = 1, 2, 3
        sub_base = _ASCII.to(device)[((code + torch.randint(1, 4, (n_pairs, L), generator=g, device=device, dtype=torch.uint8)) % 4).long()]
        bases = torch.where(sub & (bases != 78), sub_base, bases)
        rc = comp[torch.flip(bases, dims=[1]).long()]
        bases = torch.where(reverse.unsqueeze(1), rc, bases)
        return bases

    # FR: the read on the forward strand starts the fragment, its mate is the reverse complement of the fragment end
    r1_left = torch.where(flip, frag_start + insert - L1, frag_start)
    r2_left = torch.where(flip, frag_start, frag_start + insert - L2)
    r1 = one_read(L1, r1_left, flip)
    r2 = one_read(L2, r2_left, ~flip)
    bases = torch.cat([r1, r2], dim=1)
    L = L1 + L2
    rnd_pair = torch.rand(n_pairs, generator=g, device=device) < random_pair_fraction        # unalignable clusters
    rnd = _ASCII.to(device)[torch.randint(0, 4, (n_pairs, L), generator=g, device=device)]
    bases = torch.where(rnd_pair.unsqueeze(1), rnd, bases)
    qual = torch.randint(30, 41, (n_pairs, L), generator=g, device=device, dtype=torch.uint8)
    # low-quality 3' tails on a fraction of reads (exercises --base-quality-cutoff trimming, Quality.cpp:72-105)
    for off, Lr in ((0, L1), (L1, L2)):
        tail = torch.rand(n_pairs, generator=g, device=device) < low_quality_tail_fraction
        tlen = torch.randint(1, 40, (n_pairs,), generator=g, device=device)
        pos = torch.arange(Lr, device=device).unsqueeze(0)
        in_tail = tail.unsqueeze(1) & (pos >= (Lr - tlen).unsqueeze(1))
        q = qual[:, off:off + Lr]
        q[in_tail] = torch.randint(2, 15, (int(in_tail.sum()),), generator=g, device=device, dtype=torch.uint8)
    is_n = (torch.rand(n_pairs, L, generator=g, device=device) < n_rate) | (bases == 78)
    code = torch.zeros_like(bases)
    code[bases == 67], code[bases == 71], code[bases == 84] = 1, 2, 3
    bcl = torch.where(is_n, torch.zeros_like(code), code | (qual << 2))
    truth = {"contig": contig, "r1_left": r1_left, "r2_left": r2_left, "flip": flip, "random": rnd_pair}
    return bcl.contiguous(), truth


def make_sample_with_indels(contigs, rng, spacing=(250, 550), max_indel=8):
    """A sample's chromosomes: the reference's contigs (ASCII, anything np.frombuffer / np.asarray takes) with an indel of
    1..max_indel bases every spacing[0]..spacing[1] bases, half deletions and half insertions; reads drawn from it show the indels as
    gaps, and the ones that cross an indel near one of their ends are the gap realigner's work.  rng is a numpy Generator."""
    import numpy as np
    sample = []
    for c in contigs:
        seq = np.frombuffer(c, np.uint8) if isinstance(c, (bytes, bytearray)) else np.asarray(c.cpu() if hasattr(c, "cpu") else c, dtype=np.uint8)
        pieces, at = [], 0
        while at < len(seq):
            step = int(rng.integers(spacing[0], spacing[1]))
            pieces.append(seq[at:at + step]); at += step
            if at >= len(seq):
                break
            n = int(rng.integers(1, max_indel + 1))
            if rng.random() < 0.5:
                at += n                                                  # deletion from the reference
            else:
                pieces.append(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)])     # insertion
        sample.append(torch.from_numpy(np.concatenate(pieces)))
    return sample


def bcl_to_fastq(bcl, read_offset, read_length, name="r", newline=b"\n", plus_header=False):
    """FASTQ text of one read of a BCL tile (numpy uint8 [n, cluster_length]); N for quality-0 bytes.  The inverse of
    isaac_gpu_fastq_to_bcl for bytes whose quality is not 0."""
    import numpy as np
    b = np.asarray(bcl)[:, read_offset:read_offset + read_length]
    is_n = (b & 0xFC) == 0
    bases = np.where(is_n, ord("N"), np.frombuffer(b"ACGT", np.uint8)[b & 3]).astype(np.uint8)
    quals = np.where(is_n, 33 + 2, 33 + (b >> 2)).astype(np.uint8)
    out = bytearray()
    for i in range(len(b)):
        header = b"@%s:%d" % (name.encode(), i)
        out += header + newline + bases[i].tobytes() + newline + b"+" + (header[1:] if plus_header else b"") + newline + quals[i].tobytes() + newline
    return bytes(out)


def write_fastq(files, bcl, read_length, name_prefix=b"M1:7:FCSYNTH:1:1101:", first_index=0):
    """appends the FASTQ records of a BCL tile (numpy uint8 [n, 2 * read_length] or [n, read_length]) to the open binary files `files` (one per read), built as
    one array per read: '@' + name_prefix + a nine-digit number, bases (N for quality-0 bytes), '+', qualities.  The flowcell id the reference's Casava name
    parser finds is the third ':'-separated field of the name."""
    import numpy as np
    bcl = np.asarray(bcl)
    n = len(bcl)
    header = np.frombuffer(b"@" + name_prefix, np.uint8)
    digits = np.char.zfill(np.arange(first_index, first_index + n).astype(str), 9).astype("S9").view(np.uint8).reshape(n, 9)
    nl, plus = np.full((n, 1), 10, np.uint8), np.full((n, 1), ord("+"), np.uint8)
    for r, f in enumerate(files):
        b = bcl[:, r * read_length:(r + 1) * read_length]
        is_n = (b & 0xFC) == 0
        bases = np.where(is_n, ord("N"), np.frombuffer(b"ACGT", np.uint8)[b & 3]).astype(np.uint8)
        quals = np.where(is_n, 35, 33 + (b >> 2)).astype(np.uint8)
        np.concatenate([np.tile(header, (n, 1)), digits, nl, bases, nl, plus, nl, quals, nl], axis=1).tofile(f)


def write_fasta(path, contigs, names=None):
    """the contigs (uint8 tensors / arrays of ASCII bases) as a FASTA file with 60-base lines; returns [(byte offset, bytes in the file, bases, ACGT bases)] per contig,
    what isaac_reference_contig wants to know"""
    import numpy as np
    meta = []
    with open(path, "wb") as f:
        for i, c in enumerate(contigs):
            seq = c.cpu().numpy() if hasattr(c, "cpu") else np.asarray(c)
            f.write(b">" + (names[i] if names else b"chr%d" % (i + 1)) + b" synthetic\n")
            begin = f.tell()
            full = len(seq) // 60 * 60
            if full:
                np.concatenate([seq[:full].reshape(-1, 60), np.full((full // 60, 1), 10, np.uint8)], axis=1).tofile(f)
            if len(seq) > full:
                f.write(seq[full:].tobytes() + b"\n")
            meta.append((begin, f.tell() - begin, len(seq), int(sum(int((seq == b).sum()) for b in b"ACGT"))))
    return meta


# ---- a human-like reference (BASELINE.json configs 2-4: "GRCh38") ---------------------------------------------------------
# Relative lengths of GRCh38's chr1..22, X, Y, M (Mbp): the contig list of the synthetic genome follows them.
_GRCH38_MBP = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28, 114.36, 107.04, 101.99,
               90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82, 156.04, 57.23, 0.0166]


class Genome:
    """contigs concatenated in one uint8 tensor (ASCII ACGTN) + their offsets; `contigs` are views of it"""

    def __init__(self, bases, offsets, padded=None):
        self.bases = bases
        self.padded = padded            # the same storage followed by >= 64 bytes of 'N' (what isaac_gpu_load_contigs_dev wants), or None
        self.offsets = [int(o) for o in offsets]
        self.contigs = [bases[self.offsets[i]:self.offsets[i + 1]] for i in range(len(self.offsets) - 1)]

    def __len__(self):
        return len(self.contigs)

    def __iter__(self):
        return iter(self.contigs)

    def __getitem__(self, i):
        return self.contigs[i]


def _expand(starts, lengths):
    """for segments (starts[i], lengths[i]): flat tensors (segment index, offset inside the segment) of all their elements"""
    seg = torch.repeat_interleave(torch.arange(len(lengths), device=lengths.device), lengths)
    first = torch.cumsum(lengths, 0) - lengths
    off = torch.arange(int(lengths.sum()), device=lengths.device) - first[seg]
    return seg, off


def _mutate(codes, rate, g):
    """substitutions at per-element probability `rate` (tensor or float); codes 0..3"""
    mut = torch.rand(codes.shape, generator=g, device=codes.device) < rate
    shift = torch.randint(1, 4, codes.shape, generator=g, device=codes.device, dtype=torch.uint8)
    return torch.where(mut, (codes + shift) % 4, codes)


def make_human_like_genome(n_bases, seed=3, device="cpu", n_contigs=25, chunk=1 << 26):
    """A synthetic stand-in for GRCh38 (no genome ships with the image): `n_contigs` contigs with the relative lengths of the
    human chromosomes and the repeat spectrum the seed-and-extend path reacts to, as fractions of the genome:
       * Alu-like SINEs: ~300 bp, one copy per 3 kbp (10 %), six subfamilies, 2-16 % divergence from their consensus, A-rich tails;
       * L1-like LINEs: 6 kbp consensus, 5'-truncated copies (mean ~1 kbp), one per 6 kbp (16 %), 3-20 % divergence;
       * segmental duplications: 5-50 kbp copies of other places at 0.5-3 % divergence (~4 %);
       * centromeric satellites: per contig an array of a 171-bp monomer in higher-order repeats (~1 %), 1-2 % between copies;
       * microsatellites and homopolymer runs; a short element present > 1000 times;
       * N: telomeres, one centromeric gap per contig and scattered assembly gaps.
    Everything is vectorised torch on `device` (3.1 Gbp take seconds on the GPU).  Returns a Genome."""
    g = torch.Generator(device=device).manual_seed(seed)
    rel = _GRCH38_MBP[:n_contigs] if n_contigs <= len(_GRCH38_MBP) else _GRCH38_MBP + [1.0] * (n_contigs - len(_GRCH38_MBP))
    lens = [max(200, int(n_bases * r / sum(rel))) for r in rel]
    lens[0] += n_bases - sum(lens) if n_bases > sum(lens) else 0
    offsets = [0]
    for n in lens:
        offsets.append(offsets[-1] + n)
    total = offsets[-1]
    codes = torch.empty(total, dtype=torch.uint8, device=device)
    for a in range(0, total, chunk):
        codes[a:min(total, a + chunk)] = torch.randint(0, 4, (min(total, a + chunk) - a,), generator=g, device=device, dtype=torch.uint8)
    off_t = torch.tensor(offsets, dtype=torch.long, device=device)

    def rand_int(lo, hi, n):
        return torch.randint(int(lo), int(hi), (n,), generator=g, device=device)

    def place(lengths):
        """a random place for every segment that does not cross a contig end"""
        n = len(lengths)
        contig = torch.multinomial(torch.tensor(lens, dtype=torch.float, device=device), n, replacement=True, generator=g)
        room = (off_t[contig + 1] - off_t[contig] - lengths).clamp(min=1)
        return off_t[contig] + (torch.rand(n, generator=g, device=device, dtype=torch.float64) * room.double()).long()

    def paste(src_of, dst, lengths, divergence):
        """codes[dst[i] + j] = mutate(src_of(i, j)) for j < lengths[i], in pieces of at most `chunk` elements"""
        n = len(lengths)
        csum = torch.cumsum(lengths, 0)
        a = 0
        while a < n:
            base = int(csum[a - 1]) if a else 0
            b = int(torch.searchsorted(csum, torch.tensor(base + chunk, device=device), right=True))
            b = max(a + 1, min(n, b))
            seg, off = _expand(dst[a:b], lengths[a:b])
            vals = _mutate(src_of(seg + a, off), divergence[a:b][seg] if torch.is_tensor(divergence) else divergence, g)
            idx = dst[a:b][seg] + off
            ok = idx < total
            idx, vals = idx[ok], vals[ok]
            # copies overlap: the later one wins, whatever the device (a scatter with repeated indices is not ordered on a GPU)
            idx, perm = torch.sort(idx, stable=True)
            last = torch.ones_like(idx, dtype=torch.bool)
            last[:-1] = idx[1:] != idx[:-1]
            codes[idx[last]] = vals[perm][last]
            a = b

    # segmental duplications first (they copy whatever is there; later elements land inside them as in a real genome)
    n_sd = total // 600_000
    if n_sd:
        length = rand_int(5_000, 50_000, n_sd).clamp(max=max(200, min(lens) // 4))
        src, dst = place(length), place(length)
        snapshot_free = True  # sources are read while destinations are written: harmless for a synthetic genome
        paste(lambda i, j: codes[(src[i] + j).clamp(max=total - 1)], dst, length, torch.rand(n_sd, generator=g, device=device) * 0.025 + 0.005)
    # L1-like
    n_l1 = total // 6_000
    if n_l1:
        consensus = torch.randint(0, 4, (6_000,), generator=g, device=device, dtype=torch.uint8)
        u = torch.rand(n_l1, generator=g, device=device)
        length = (6_000 * u * u * 0.5 + 100).long().clamp(max=6_000)            # 5'-truncated: the 3' end is what is left
        length = torch.where(torch.rand(n_l1, generator=g, device=device) < 0.03, torch.full_like(length, 6_000), length).clamp(max=max(100, min(lens) // 4))
        start = 6_000 - length
        paste(lambda i, j: consensus[start[i] + j], place(length), length, torch.rand(n_l1, generator=g, device=device) * 0.17 + 0.03)
    # Alu-like
    n_alu = total // 3_000
    if n_alu:
        master = torch.randint(0, 4, (282,), generator=g, device=device, dtype=torch.uint8)
        sub = torch.stack([_mutate(master, 0.03, g) for _ in range(6)])
        tail = torch.zeros((6, 30), dtype=torch.uint8, device=device)                 # A-rich tail
        sub = torch.cat([sub, tail], 1)                                                # 312 columns
        fam = rand_int(0, 6, n_alu)
        length = rand_int(280, 313, n_alu).clamp(max=max(50, min(lens) // 4))
        paste(lambda i, j: sub[fam[i], j], place(length), length, torch.rand(n_alu, generator=g, device=device) * 0.14 + 0.02)
    # a short element present more than 1000 times (a single TooManyMatch entry of the index, ReferenceSorter.cpp:201-222)
    if total >= 200_000:
        element = torch.randint(0, 4, (40,), generator=g, device=device, dtype=torch.uint8)
        n_el = max(1100, total // 2_000_000)
        length = torch.full((n_el,), 40, dtype=torch.long, device=device)
        paste(lambda i, j: element[j], place(length), length, 0.0)
    # microsatellites / homopolymers
    n_ms = total // 20_000
    if n_ms:
        unit_len = rand_int(1, 5, n_ms)
        unit = torch.randint(0, 4, (n_ms, 4), generator=g, device=device, dtype=torch.uint8)
        length = rand_int(15, 120, n_ms)
        paste(lambda i, j: unit[i, j % unit_len[i]], place(length), length, 0.01)
    # centromeric satellite array + gap per contig; telomeres; scattered gaps
    is_n = torch.zeros(0, dtype=torch.bool, device=device)
    n_starts, n_lengths = [], []
    for c, n in enumerate(lens):
        if n < 20_000:
            continue
        mono = torch.randint(0, 4, (171,), generator=g, device=device, dtype=torch.uint8)
        k = int(torch.randint(4, 13, (1,), generator=g, device=device))
        hor = torch.cat([_mutate(mono, 0.2, g) for _ in range(k)])                     # higher-order repeat of k diverged monomers
        array_len = max(2 * len(hor), n // 100)
        centre = offsets[c] + int(n * (0.35 + 0.3 * float(torch.rand(1, generator=g, device=device))))
        a0 = centre - array_len // 2
        for a in range(a0, a0 + array_len, chunk):
            m = min(a0 + array_len, a + chunk) - a
            j = torch.arange(a - a0, a - a0 + m, device=device)
            codes[a:a + m] = _mutate(hor[j % len(hor)], 0.015, g)
        n_starts += [offsets[c], offsets[c + 1] - max(10, n // 20_000), centre + array_len // 2]
        n_lengths += [max(10, n // 20_000), max(10, n // 20_000), max(50, n // 30)]
    n_gaps = total // 5_000_000
    if n_gaps:
        gl = rand_int(50, 2_000, n_gaps)
        gs = place(gl)
        n_starts += gs.tolist(); n_lengths += gl.tolist()
    # to ASCII, then the N runs
    bases = torch.empty(total + 64, dtype=torch.uint8, device=device)
    lut = _ASCII.to(device)
    for a in range(0, total, chunk):
        b = min(total, a + chunk)
        bases[a:b] = lut[codes[a:b].long()]
    bases[total:] = 78
    del codes
    for s, n in zip(n_starts, n_lengths):
        bases[max(0, s):min(total, s + n)] = 78
    return Genome(bases[:total], offsets, padded=bases)
