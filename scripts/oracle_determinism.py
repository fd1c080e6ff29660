#!/usr/bin/env python3
"""the oracle's multi-threaded match selection against itself on the bench's first timed batch (a development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from isaac_aligner_amd import abi, gpu, options, synth
dev = torch.device("cuda", 0)
g = synth.make_human_like_genome(3_100_000_000, seed=3, device=dev)
p = options.default_params(150, 150)
al = gpu.Aligner(p, 0, g)
al.build_index()
orc = oracle_lib.load()
ref = orc.reference([c.cpu().numpy().tobytes() for c in g.contigs])
ref.set_index(al.get_index())
op = orc.default_params(2, 150, 150)
n = 1_000_000
bcl = synth.make_read_pairs(g, n, 150, seed=1001, device=dev, avoid_gaps=True)[0]
host = bcl.cpu().numpy()
m, o, hits = al.find_matches(bcl, tile=1)
al.set_loaded_contigs(np.ones_like(hits))
tls = al.determine_tls(bcl, m, o)
otls = oracle_lib.Tls()
for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
    setattr(otls, name, getattr(tls, name))
otls.best_model[0], otls.best_model[1] = tls.best_model[0], tls.best_model[1]
om, ohits = ref.find_matches(op, host, n, tile=1, n_threads=64)
cluster_of = ((om["seed_id"] >> np.uint64(9)) & np.uint64(0x7fffffff)).astype(np.int64)
runs = []
for nt in (256, 256, 256, 64, 64, 200):
    rec, cig, _ = ref.select(op, host, om, otls, np.ones_like(hits), tile=1, n_threads=nt, n_clusters_hint=n)
    runs.append((nt, rec))
    if len(runs) > 1:
        base = runs[0][1]
        same = np.ones(len(rec), bool)
        for f in rec.dtype.names:
            if f != "cigar_offset":
                same &= base[f] == rec[f]
        bad = np.nonzero(~same)[0]
        # first cluster of every thread's share (oracle_select cuts the match list evenly and moves the cut to a cluster boundary)
        starts = set()
        for i in range(1, nt):
            b = len(om) * i // nt
            while b < len(om) and b and cluster_of[b] == cluster_of[b - 1]:
                b += 1
            if b < len(om):
                starts.add(int(cluster_of[b]))
        starts0 = set()
        for i in range(1, runs[0][0]):
            b = len(om) * i // runs[0][0]
            while b < len(om) and b and cluster_of[b] == cluster_of[b - 1]:
                b += 1
            if b < len(om):
                starts0.add(int(cluster_of[b]))
        print("threads", nt, "records differing from the first run:", len(bad), "clusters", sorted(set(int(i) // 2 for i in bad))[:10],
              "of which first-of-a-share (this run / first run):", sum(1 for c in set(int(i) // 2 for i in bad) if c in starts), sum(1 for c in set(int(i) // 2 for i in bad) if c in starts0), flush=True)
        for i in bad[:4]:
            print("   ", i, base[i], "\n        ", rec[i])
