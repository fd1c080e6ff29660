#!/usr/bin/env python3
"""one batch of the bench (seed 1001, tile 1) several times on the GPU and on the oracle: which side moves? (a development aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oracle_lib
from parity_util import count_record_diffs, sort_matches
from isaac_aligner_amd import abi, gpu, options, synth
dev = torch.device("cuda", 0)
g = synth.make_human_like_genome(3_100_000_000, seed=3, device=dev)
p = options.default_params(150, 150)
al = gpu.Aligner(p, 0, g)
al.build_index()
orc = oracle_lib.load()
ref = orc.reference([c.cpu().numpy().tobytes() for c in g.contigs])
index = al.get_index()
index2 = al.get_index()
print("two downloads of the table identical:", index.tobytes() == index2.tobytes(), flush=True)
del index2
ref.set_index(index)
op = orc.default_params(2, 150, 150)
n = 1_000_000
warm = synth.make_read_pairs(g, n, 150, seed=1000, device=dev, avoid_gaps=True)[0]
bcl = synth.make_read_pairs(g, n, 150, seed=1001, device=dev, avoid_gaps=True)[0]
host = bcl.cpu().numpy()
wm, wo, wh = al.find_matches(warm)
al.set_loaded_contigs(np.ones_like(wh))
tls = al.determine_tls(warm, wm, wo)
otls = oracle_lib.Tls()
for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
    setattr(otls, name, getattr(tls, name))
otls.best_model[0], otls.best_model[1] = tls.best_model[0], tls.best_model[1]
prev = None
for it in range(2):
    m, o, hits = al.find_matches(bcl, tile=1)
    gm = m.cpu().numpy().view(np.uint64).reshape(-1, 2)
    gm = np.rec.fromarrays([gm[:, 0], gm[:, 1]], dtype=oracle_lib.MATCH_DTYPE)
    rec, cig = al.select(bcl, m, o, tls, tile=1)
    packed, _ = al.compact_cigars(rec, cig)
    grec = rec.cpu().numpy().view(abi.FRAGMENT_DTYPE).reshape(-1).copy(); gcig = packed.cpu().numpy().view(np.uint32).copy()
    om, ohits = ref.find_matches(op, host, n, tile=1, n_threads=64)
    a, b = sort_matches(om), sort_matches(gm)
    same_matches = len(a) == len(b) and (a["seed_id"] == b["seed_id"]).all() and (a["location"] == b["location"]).all()
    orec, ocig, _ = ref.select(op, host, om, otls, np.ones_like(hits), tile=1, n_threads=os.cpu_count(), n_clusters_hint=n)
    nd, text = count_record_diffs(orec, ocig, grec, gcig, limit=4)
    line = "iteration %d: matches identical %s, record diffs %d" % (it, same_matches, nd)
    if prev is not None:
        line += "; GPU records same as previous iteration %s; oracle records same %s; oracle matches same %s" % (
            prev[0].tobytes() == grec.tobytes(), all((prev[1][f] == orec[f]).all() for f in orec.dtype.names), prev[2].tobytes() == om.tobytes())
    print(line, flush=True)
    print('   gpu', grec[7670], grec[7671]); print('   orc', orec[7670], orec[7671])
    for t in text:
        print(t)
    if not same_matches:
        k = min(len(a), len(b)); bad = np.nonzero((a["seed_id"][:k] != b["seed_id"][:k]) | (a["location"][:k] != b["location"][:k]))[0]
        print("first differing match", bad[:3], a[bad[:3]], b[bad[:3]], len(a), len(b))
    prev = (grec, orec, om)
