#!/usr/bin/env python3
"""GPU records against the oracle's over several 1 M-pair batches of the GRCh38-sized workload (a development aid: rare cluster shapes)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from parity_util import count_record_diffs
from isaac_aligner_amd import abi, gpu, options, synth
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 3_100_000_000
n_pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
first_seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
dev = torch.device("cuda", 0)
g = synth.make_human_like_genome(bases, seed=3, device=dev)
p = options.default_params(150, 150)
al = gpu.Aligner(p, 0, g)
al.build_index()
orc = oracle_lib.load()
ref = orc.reference([c.cpu().numpy().tobytes() for c in g.contigs])
ref.set_index(al.get_index())
op = orc.default_params(2, 150, 150)
tls = None
total = 0
for b in range(n_batches):
    bcl = synth.make_read_pairs(g, n_pairs, 150, seed=first_seed + b, device=dev, avoid_gaps=True)[0]
    m, o, hits = al.find_matches(bcl, tile=1 + b)
    al.set_loaded_contigs(np.ones_like(hits))
    if tls is None:
        tls = al.determine_tls(bcl, m, o)
    rec, cig = al.select(bcl, m, o, tls, tile=1 + b)
    packed, _ = al.compact_cigars(rec, cig)
    grec = rec.cpu().numpy().view(abi.FRAGMENT_DTYPE).reshape(-1); gcig = packed.cpu().numpy().view(np.uint32)
    host = bcl.cpu().numpy()
    om, ohits = ref.find_matches(op, host, n_pairs, tile=1 + b, n_threads=64)
    otls = oracle_lib.Tls()
    for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
        setattr(otls, name, getattr(tls, name))
    otls.best_model[0], otls.best_model[1] = tls.best_model[0], tls.best_model[1]
    orec, ocig, _ = ref.select(op, host, om, otls, np.ones_like(hits), tile=1 + b, n_threads=os.cpu_count(), n_clusters_hint=n_pairs)
    n, text = count_record_diffs(orec, ocig, grec, gcig, limit=6)
    total += n
    print("batch", b, "seed", first_seed + b, "diffs", n, flush=True)
    for t in text:
        print(t)
print("total diffs", total, al.counters())
