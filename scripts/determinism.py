#!/usr/bin/env python3
"""select on the same batch several times: any record that differs between the runs points at a race (a development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaac_aligner_amd import abi, gpu, options, synth
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
bases = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000
dev = torch.device("cuda", 0)
g = synth.make_human_like_genome(bases, seed=3, device=dev)
al = gpu.Aligner(options.default_params(150, 150), 0, g)
al.build_index()
bcl = synth.make_read_pairs(g, n_pairs, 150, seed=1001, device=dev, avoid_gaps=True)[0]
m, o, hits = al.find_matches(bcl)
al.set_loaded_contigs(np.ones_like(hits))
tls = al.determine_tls(bcl, m, o)
first = None
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    rec, cig = al.records_to_numpy(*al.select(bcl, m, o, tls))
    if first is None:
        first = (rec, cig); continue
    same = np.ones(len(rec), bool)
    for f in rec.dtype.names:
        same &= rec[f] == first[0][f]
    cs = (cig.reshape(len(rec), -1) == first[1].reshape(len(rec), -1)).all(1) | (rec["cigar_length"] == 0)
    bad = np.nonzero(~same)[0]
    print("run", it, "records differing from run 0:", len(bad), "cigar slots differing:", int((~cs).sum()))
    for i in bad[:4]:
        print("  ", i, first[0][i], "\n      ", rec[i])
print(al.counters())
