#!/usr/bin/env python3
"""the bench's flow (find for all batches, then back-to-back deferred select calls) twice over the same batches: records that
differ between the two passes point at a hazard between calls (a development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from isaac_aligner_amd import abi, gpu, options, synth
n_pairs, bases, n_batches = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
g = synth.make_human_like_genome(bases, seed=3, device=dev)
al = gpu.Aligner(options.default_params(150, 150), 0, g, deferred_completion=True)
al.build_index()
batches = [synth.make_read_pairs(g, n_pairs, 150, seed=1001 + b, device=dev, avoid_gaps=True)[0] for b in range(n_batches)]
tls = None
def flow():
    global tls
    found = [al.find_matches(b, tile=1 + i) for i, b in enumerate(batches)]
    al.set_loaded_contigs(np.ones_like(found[0][2]))
    if tls is None:
        tls = al.determine_tls(batches[0], found[0][0], found[0][1])
    outs = [al.select(b, m, o, tls, tile=1 + i) for i, (b, (m, o, _)) in enumerate(zip(batches, found))]
    al.synchronize()
    return [(r.cpu().numpy().view(abi.FRAGMENT_DTYPE).reshape(-1), c.cpu().numpy()) for r, c in outs]
ref = flow()
other = synth.make_read_pairs(g, 700_000, 150, seed=77, device=dev, avoid_gaps=True)[0]
def perturb():
    # different data through every chunk buffer, so that anything read without having been written shows up as a difference
    m, o, _ = al.find_matches(other)
    al.select(other, m, o, tls)
    al.synchronize()
for it in range(int(sys.argv[4])):
    perturb()
    cur = flow()
    for b, ((r0, c0), (r1, c1)) in enumerate(zip(ref, cur)):
        same = np.ones(len(r0), bool)
        for f in r0.dtype.names:
            same &= r0[f] == r1[f]
        bad = np.nonzero(~same)[0]
        if len(bad):
            print("pass", it, "batch", b, "records differing:", len(bad))
            for i in bad[:3]:
                print("  ", i, r0[i], "\n      ", r1[i])
    print("pass", it, "compared")
print(al.counters())
