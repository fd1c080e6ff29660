#!/usr/bin/env python3
"""Experiment: C contexts on one GPU, each on a stream of its own, sharing the contigs and one resident table (isaac_gpu_set_index_dev), the
select calls of consecutive 1 M-pair steps handed to them in turn without host waits (deferred completion).  Kernels of different steps then
share the GPU: tails and memory-bound kernels of one overlap the arithmetic of another.  Prints reads/s for each C on the bench's workload.
    EXP_GENOME_BASES (3.1e9), EXP_STEPS (12), EXP_CONTEXTS ("1,2,3")"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isaac_aligner_amd import abi, gpu, options, synth

def main():
    pairs, steps, L = 1_000_000, int(os.environ.get("EXP_STEPS", "12")), 150
    dev = torch.device("cuda", 0)
    params = options.default_params(L, L)
    genome = synth.make_human_like_genome(int(float(os.environ.get("EXP_GENOME_BASES", "3.1e9"))), seed=3, device=dev)
    main_al = gpu.Aligner(params, 0, genome, deferred_completion=True)
    main_al.build_index(repeat_threshold=1000)
    table = main_al.index_tensors()
    batches = [synth.make_read_pairs(genome, pairs, L, seed=1000 + b, device=dev, avoid_gaps=True)[0] for b in range(steps + 1)]
    m, o, hits = main_al.find_matches(batches[0]); main_al.set_loaded_contigs(np.ones_like(hits))
    tls = main_al.determine_tls(batches[0], m, o)
    outs = [(torch.empty((pairs * 2, abi.FRAGMENT_DTYPE.itemsize), dtype=torch.uint8, device=dev), torch.empty(pairs * 2 * abi.MAX_CIGAR_OPS, dtype=torch.int32, device=dev)) for _ in range(steps)]
    found = [main_al.find_matches(batches[1 + s], tile=1 + s)[:2] for s in range(steps)]
    torch.cuda.synchronize()
    reference = None
    for C in [int(x) for x in os.environ.get("EXP_CONTEXTS", "1,2,3").split(",")]:
        als = [main_al]
        for _ in range(C - 1):
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                al = gpu.Aligner(params, 0, genome, deferred_completion=True)
                al.set_index_tensors(table)
                al.set_loaded_contigs(np.ones_like(hits))
                als.append(al)
        for k, al in enumerate(als):                       # warm-up: every context grows its buffers
            al.select(batches[1], found[0][0], found[0][1], tls, tile=1, out=outs[k]); al.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            als[s % C].select(batches[1 + s], found[s][0], found[s][1], tls, tile=1 + s, out=outs[s])
        for al in als:
            al.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        digest = [int(o_[0].view(torch.int64).sum().item()) for o_ in outs]
        if reference is None:
            reference = digest
        print("contexts %d: %.1f M reads/s, %.2f ms per 1 M-pair select step, records %s, %.0f GB in use" % (
            C, steps * pairs * 2 / dt / 1e6, dt / steps * 1e3, "identical" if digest == reference else "DIFFERENT", (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
        for al in als[1:]:
            al.close()
        del als
        torch.cuda.empty_cache()

main()
