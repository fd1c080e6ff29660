"""Read-sharded multi-GPU execution of the path: one process per GPU, every rank aligns its own clusters.

The reference shards the same way inside one process (FindMatchesTransition / SelectMatchesTransition hand tiles to worker
threads; MatchSelector::parallelSelect, lib/alignment/MatchSelector.cpp:372-460): clusters are independent once three pieces
of run-wide state are fixed, and those are the only things the ranks exchange:

  * the set of contigs that received any seed match (MatchSelector.cpp:85-90 loads only those; the rest-of-genome
    correction depends on it): OR over the ranks                      -> all_reduce(MAX)
  * the template length statistics, learnt from the first tile only (MatchSelector.cpp:402-417)  -> broadcast from rank 0
  * the FragmentHeader records, collected once at the end             -> gather to rank 0, rank order = cluster order

No collective sits on the per-cluster data path.  `dist` is torch.distributed (backend "nccl" = RCCL on the GPUs, "gloo" in
the CPU tests) or None for a single process.
"""
import numpy as np
import torch


def shard_bounds(n_items, rank, world):
    """contiguous static shard [begin, end) of n_items for this rank (sizes differ by at most one)"""
    base, extra = divmod(int(n_items), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def reduce_contig_hits(hits, dist, device="cpu"):
    """OR of the per-contig hit flags over all ranks"""
    h = np.ascontiguousarray(hits, np.uint8)
    if dist is None:
        return h
    t = torch.from_numpy(h.astype(np.int32)).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.cpu().numpy().astype(np.uint8)


TLS_FIELDS = ("min", "max", "median", "low_std_dev", "high_std_dev")


def broadcast_tls(tls, dist, device="cpu", src=0):
    """rank `src`'s template length statistics become everybody's (in place); tls: abi.Tls-like ctypes structure"""
    if dist is None:
        return tls
    t = torch.tensor([int(v) for v in tls.astuple()], dtype=torch.int64, device=device)
    dist.broadcast(t, src)
    v = t.cpu().tolist()
    tls.min, tls.max, tls.median, tls.low_std_dev, tls.high_std_dev = v[0:5]
    tls.best_model[0], tls.best_model[1], tls.stable, tls.mate_min, tls.mate_max = v[5:10]
    return tls


def broadcast_table(entries, dist, rank, device="cpu", src=0, chunk=1 << 27):
    """The resident table of rank `src` (an [n, 2] int64 array: isaac_gpu_index_dev) becomes every rank's: one build and N - 1 transfers over
    xGMI instead of N builds.  Ranks other than `src` pass None and get a fresh tensor to hand to isaac_gpu_set_index_dev.  The entry
    count goes first; the table follows in pieces of `chunk` entries (2 GB), so that no single collective carries a count beyond 2^31."""
    if dist is None:
        return entries
    n = torch.tensor([int(entries.shape[0]) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    n = int(n.item())
    if rank != src:
        entries = torch.empty((n, 2), dtype=torch.int64, device=device)
    for begin in range(0, n, chunk):
        dist.broadcast(entries[begin:begin + chunk], src)
    return entries


def gather_records(records, dist, rank, world, dst=0):
    """records: (n, record_bytes) uint8 tensor of this rank; returns the list of all ranks' tensors on `dst`, else None.
    Shards may differ in size by one cluster: sizes are exchanged first and the payload is padded for the gather."""
    if dist is None:
        return [records]
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    width = max(sizes)
    mine = records
    if records.shape[0] < width:
        mine = torch.zeros((width,) + tuple(records.shape[1:]), dtype=records.dtype, device=records.device)
        mine[:records.shape[0]] = records
    out = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, out, dst=dst)
    if rank != dst:
        return None
    return [t[:s] for t, s in zip(out, sizes)]


class StepGather:
    """Collects every step's records and packed CIGARs on rank `dst` while the later steps compute: the payload of a step is handed to
    asynchronous gathers as soon as its kernels are queued, and only finish() waits.  Per rank and step about 140 B per pair cross one
    xGMI link (about 48 GB/s): left to the end of a run that is 8-9 % of the run's time on its own, spread over the steps it hides
    behind them.  Nothing in add() waits for the GPU or for another rank: the record count of every rank is known beforehand
    (`record_counts`, e.g. 2 x shard_bounds), the CIGAR pool travels at its fixed capacity and its fill level as a third, tiny gather
    of the device word isaac_gpu_compact_cigars_async wrote."""

    def __init__(self, dist, rank, world, dst=0, record_counts=None):
        self.dist, self.rank, self.world, self.dst = dist, rank, world, dst
        self.record_counts = None if record_counts is None else [int(c) for c in record_counts]
        self.pending, self.steps = [], []

    def _pad(self, t, n):
        if t.shape[0] == n:
            return t.contiguous()
        out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        out[:t.shape[0]] = t
        return out

    def add(self, records, cigars, n_words=None, record_counts=None):
        """records: (n, record_bytes) uint8 of this rank for one step; cigars: its CIGAR pool (1-D int32, at most 4 words per record of the
        largest shard), n_words: 1-element int64 tensor holding the number of pool words in use (may still be in flight on the device),
        or None when all of `cigars` is meant; record_counts: this step's record count of every rank when it differs from the constructor's"""
        if n_words is None:
            n_words = torch.tensor([cigars.shape[0]], dtype=torch.int64, device=cigars.device)
        if self.dist is None:
            self.pending.append((None, None, None, records, cigars, n_words, None, None, None))
            return
        dist = self.dist
        counts = [int(c) for c in record_counts] if record_counts is not None else self.record_counts if self.record_counts is not None else [records.shape[0]] * self.world
        assert counts[self.rank] == records.shape[0], "record_counts disagrees with this rank's records"
        assert cigars.shape[0] <= 4 * max(counts), "the CIGAR pool is larger than 4 words per record"
        width = max(counts)
        rec = self._pad(records, width)
        cig = self._pad(cigars, 4 * width)                # the pool's capacity is 4 words per record everywhere (bench.py: buffers())
        on_dst = self.rank == self.dst
        out_r = [torch.empty_like(rec) for _ in range(self.world)] if on_dst else None
        out_c = [torch.empty_like(cig) for _ in range(self.world)] if on_dst else None
        out_n = [torch.empty_like(n_words) for _ in range(self.world)] if on_dst else None
        h_r = dist.gather(rec, out_r, dst=self.dst, async_op=True)
        h_c = dist.gather(cig, out_c, dst=self.dst, async_op=True)
        h_n = dist.gather(n_words, out_n, dst=self.dst, async_op=True)
        self.pending.append((h_r, h_c, h_n, rec, cig, n_words, out_r, out_c, (out_n, counts)))     # the tensors stay referenced until finish()

    def finish(self):
        """waits for every gather; on `dst`: [(list of record tensors by rank, list of CIGAR tensors by rank)] per step, else None"""
        for h_r, h_c, h_n, rec, cig, n_words, out_r, out_c, out_n in self.pending:
            if self.dist is None:
                self.steps.append(([rec], [cig[:int(n_words.item())]]))
                continue
            h_r.wait(); h_c.wait(); h_n.wait()
            if self.rank == self.dst:
                out_n, counts = out_n
                for r, (t, n) in enumerate(zip(out_c, out_n)):      # a pool that overflowed on its rank would come back clamped by the slice below
                    assert int(n.item()) <= t.shape[0], "rank %d packed %d CIGAR words into a pool of %d" % (r, int(n.item()), t.shape[0])
                self.steps.append(([t[:c] for t, c in zip(out_r, counts)], [t[:int(n.item())] for t, n in zip(out_c, out_n)]))
        self.pending = []
        if self.dist is not None and self.rank != self.dst:
            return None
        return self.steps
