"""Run with ISAAC_GPU_LIBRARY pointing at the -DISAAC_TEST_MAPQ_SKEW build: that library calls a twentieth of all MAPQ arguments "near an integer" and gets every
one of them wrong by one on the device.  isaac_gpu_resolve_flagged must put every such cluster right: it redoes the flagged clusters on the host with glibc --
FragmentBuilder::build from the seed matches, the template with its mate rescues, clippers, records -- and replaces what differs.  Before the call the device's
records differ from the oracle's in the flagged clusters and nowhere else; after it they are the oracle's.  (A process of its own: the product library is loaded once
per process.)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib                                                        # noqa: E402
from isaac_aligner_amd import gpu, options, synth                        # noqa: E402
from parity_util import compare_records                                  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
g = synth.make_human_like_genome(6_000_000, seed=21)
contigs = [bytes(c.numpy()) for c in g.contigs]
bcl = synth.make_read_pairs(g, n_pairs, 150, seed=23, avoid_gaps=True)[0]
p = options.default_params(150, 150)
al = gpu.Aligner(p, 0, contigs)
al.build_index()
dev_bcl = bcl.cuda()
m, o, hits = al.find_matches(dev_bcl, tile=3)
al.set_loaded_contigs(hits)
tls = al.determine_tls(dev_bcl, m, o, tile=3)
records, cigars = al.select(dev_bcl, m, o, tls, tile=3)
orc = oracle_lib.load()
ref = orc.reference(contigs)
ref.set_index(al.get_index())
host = bcl.numpy()
om, ohits = ref.find_matches(p, host, n_pairs, tile=3)
otls = oracle_lib.Tls()
for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
    setattr(otls, name, getattr(tls, name))
otls.best_model[0], otls.best_model[1] = tls.best_model[0], tls.best_model[1]
orec, ocig, _ = ref.select(p, host, om, otls, ohits, tile=3, n_threads=8, n_clusters_hint=n_pairs)

rec, cig = al.records_to_numpy(records, cigars)
flagged = (rec["reserved"][0::2] & 8) != 0
wrong = np.zeros(n_pairs, bool)
for f in rec.dtype.names:
    if f not in ("cigar_offset", "reserved"):
        wrong |= (rec[f] != orec[f]).reshape(-1, 2).any(axis=1)
print("before: %d clusters flagged, %d differ from the oracle, %d of those not flagged" % (flagged.sum(), wrong.sum(), (wrong & ~flagged).sum()))
ok = flagged.sum() > n_pairs // 100 and wrong.sum() > flagged.sum() // 4 and not (wrong & ~flagged).any()
n_flagged, n_changed = al.resolve_flagged(dev_bcl, m, o, tls, records, cigars, tile=3)
rec, cig = al.records_to_numpy(records, cigars)
diffs = compare_records(orec, ocig, rec, cig)
print("resolve_flagged: %d looked at, %d replaced; differences from the oracle afterwards: %d" % (n_flagged, n_changed, len(diffs)))
for d in diffs[:3]:
    print(d)
ok = ok and n_flagged == flagged.sum() and n_changed >= wrong.sum() and not diffs

# Other read lengths against the same resident genome (isaac_gpu_set_params), then the call again: what it keeps between calls -- host-side parameters, tables,
# the contigs' copy -- has to follow the context (it did not; the host then redid 2x100 clusters with 2x150 offsets and overwrote correct records)
n2 = n_pairs // 2
bcl2 = synth.make_read_pairs(g, n2, 100, seed=29, avoid_gaps=True)[0]
p2 = options.default_params(100, 100)
al.set_params(p2)
dev2 = bcl2.cuda()
m2, o2, hits2 = al.find_matches(dev2, tile=4)
al.set_loaded_contigs(hits2)
tls2 = al.determine_tls(dev2, m2, o2, tile=4)
records2, cigars2 = al.select(dev2, m2, o2, tls2, tile=4)
host2 = bcl2.numpy()
om2, ohits2 = ref.find_matches(p2, host2, n2, tile=4)
otls2 = oracle_lib.Tls()
for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
    setattr(otls2, name, getattr(tls2, name))
otls2.best_model[0], otls2.best_model[1] = tls2.best_model[0], tls2.best_model[1]
orec2, ocig2, _ = ref.select(p2, host2, om2, otls2, ohits2, tile=4, n_threads=8, n_clusters_hint=n2)
n_flagged2, n_changed2 = al.resolve_flagged(dev2, m2, o2, tls2, records2, cigars2, tile=4)
rec2, cig2 = al.records_to_numpy(records2, cigars2)
diffs2 = compare_records(orec2, ocig2, rec2, cig2)
print("after set_params(2x100): %d looked at, %d replaced; differences from the oracle afterwards: %d" % (n_flagged2, n_changed2, len(diffs2)))
for d in diffs2[:3]:
    print(d)
ok = ok and n_flagged2 > n2 // 100 and n_changed2 > 0 and not diffs2

# ... and with sequencing adapters: the host form (csrc/resolve_host.cpp, the product's own headers under g++) decides the adapter ranges itself, per read and per
# rescue, and has to clip exactly as the kernels did
from parity_util import add_adapters                                      # noqa: E402
bcl3, inserts = add_adapters(synth.make_read_pairs(g, n2, 150, seed=31, avoid_gaps=True)[0].numpy(), 150, adapter="AGATCGGAAGAGC", fraction=0.35, seed=32)
p3 = options.set_adapters(options.default_params(150, 150), "Standard")
al.set_params(p3)
dev3 = torch.from_numpy(bcl3).cuda()
m3, o3, hits3 = al.find_matches(dev3, tile=5)
al.set_loaded_contigs(hits3)
tls3 = al.determine_tls(dev3, m3, o3, tile=5)
records3, cigars3 = al.select(dev3, m3, o3, tls3, tile=5)
om3, ohits3 = ref.find_matches(p3, bcl3, n2, tile=5)
otls3 = oracle_lib.Tls()
for name in ("min", "max", "median", "low_std_dev", "high_std_dev", "stable", "mate_min", "mate_max"):
    setattr(otls3, name, getattr(tls3, name))
otls3.best_model[0], otls3.best_model[1] = tls3.best_model[0], tls3.best_model[1]
orec3, ocig3, _ = ref.select(p3, bcl3, om3, otls3, ohits3, tile=5, n_threads=8, n_clusters_hint=n2)
n_flagged3, n_changed3 = al.resolve_flagged(dev3, m3, o3, tls3, records3, cigars3, tile=5)
rec3, cig3 = al.records_to_numpy(records3, cigars3)
diffs3 = compare_records(orec3, ocig3, rec3, cig3)
clipped = int(((orec3["observed_length"].reshape(-1, 2)[:, 0] == inserts) & (inserts > 0)).sum())
print("with adapters: %d looked at, %d replaced, %d reads clipped where the insert ends; differences from the oracle afterwards: %d" % (n_flagged3, n_changed3, clipped, len(diffs3)))
for d in diffs3[:3]:
    print(d)
ok = ok and n_flagged3 > n2 // 100 and n_changed3 > 0 and clipped > n2 // 20 and not diffs3
sys.exit(0 if ok else 1)
