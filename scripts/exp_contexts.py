#!/usr/bin/env python3
"""Experiment: one context per stream, C contexts on one GPU fed from C host threads (the C ABI allows different contexts to run
concurrently).  Prints reads/s for C = 1 and C = 2 on the same batches."""
import os, sys, threading, time
os.environ.setdefault("ISAAC_GPU_DEFERRED_COMPLETION", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isaac_aligner_amd import abi, gpu, options, synth

def main():
    pairs, steps, L = 1_000_000, int(os.environ.get("EXP_STEPS", "24")), 150
    dev = torch.device("cuda", 0)
    params = options.default_params(L, L)
    contigs = synth.make_genome(46_700_000, seed=2, device=dev, n_contigs=1)
    batches = [synth.make_read_pairs(contigs, pairs, L, seed=1000 + b, device=dev)[0] for b in range(steps + 1)]
    for C in [int(x) for x in os.environ.get("EXP_CONTEXTS", "1,2,3").split(",")]:
        streams = [torch.cuda.Stream(dev) for _ in range(C)]
        als = []
        for s in streams:
            with torch.cuda.stream(s):
                al = gpu.Aligner(params, 0, contigs)
                al.build_index(repeat_threshold=1000)
                als.append(al)
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[0]):
            m, o, hits = als[0].find_matches(batches[0]); als[0].set_loaded_contigs(hits)
            tls = als[0].determine_tls(batches[0], m, o)
        for k, al in enumerate(als):
            with torch.cuda.stream(streams[k]):
                al.set_loaded_contigs(hits)
                m, o, _ = al.find_matches(batches[0]); al.select(batches[0], m, o, tls); al.synchronize()
        torch.cuda.synchronize()
        def work(k):
            al = als[k]
            with torch.cuda.stream(streams[k]):
                outs = [(torch.empty((pairs * 2, abi.FRAGMENT_DTYPE.itemsize), dtype=torch.uint8, device=dev), torch.empty(pairs * 2 * abi.MAX_CIGAR_OPS, dtype=torch.int32, device=dev)) for _ in range(2)]
                mine = [b for b in range(1, steps + 1) if b % C == k]
                found = [al.find_matches(batches[b]) for b in mine]
                for i, b in enumerate(mine):
                    al.select(batches[b], found[i][0], found[i][1], tls, out=outs[i & 1])
                al.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(C)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("contexts %d: %.1f M reads/s (%.1f ms per 1M-pair step)" % (C, steps * pairs * 2 / dt / 1e6, dt / steps * 1e3), flush=True)
        for al in als: al.close()
        del als
        torch.cuda.empty_cache()

main()
