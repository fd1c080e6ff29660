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
                    random_pair_fraction=0.002, read_length2=None):
    """FR-oriented pairs drawn uniformly from the contigs; returns (bcl uint8 [n_pairs, L1+L2], truth dict).
    indel_read_fraction = fraction of reads carrying one indel of 1..indel_max bases (0.02 %/base at 150 bp ~ 3 %)."""
    device = device or contigs[0].device
    L1 = read_length
    L2 = read_length2 or read_length
    g = torch.Generator(device=device).manual_seed(seed)
    lens = torch.tensor([c.numel() for c in contigs], dtype=torch.float64)
    genome = torch.cat([c.to(device) for c in contigs])
    starts = torch.zeros(len(contigs) + 1, dtype=torch.long)
    starts[1:] = torch.cumsum(lens, 0).long()
    starts = starts.to(device)
    # contig choice proportional to length, fragment inside the contig
    contig = torch.multinomial(lens.float().to(device), n_pairs, replacement=True, generator=g)
    clen = lens.long().to(device)[contig]
    insert = (torch.randn(n_pairs, generator=g, device=device) * insert_sd + insert_mean).round().long()
    insert = insert.clamp(min=max(L1, L2) + 10)
    insert = torch.minimum(insert, clen - 2 * indel_max - 2)
    frag_start = (torch.rand(n_pairs, generator=g, device=device, dtype=torch.float64) * (clen - insert - 2 * indel_max).double()).long()
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
