"""Experiment helper: host emulation (fast probability sort on / off) against the oracle on a sample with repeat-family clusters."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib, hostemu_lib, parity_util
from isaac_aligner_amd import options, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
contigs = synth.make_genome(4_000_000, seed=11, device="cpu", n_contigs=2)
bcl, truth = synth.make_read_pairs(contigs, N, 150, seed=12, device="cpu")
hb = bcl.numpy()
o = oracle_lib.load()
cb = [bytes(c.numpy()) for c in contigs]
t0 = time.time()
ref = o.reference(cb); ref.build_index()
p = options.default_params(150, 150)
om, hits = ref.find_matches(p, hb, N)
print("index+find %.1fs" % (time.time() - t0))
lib = hostemu_lib.load()
emu = hostemu_lib.Emu(lib, p, cb, hits)
emu.set_matches(om, N)
tls = emu.determine_tls(hb, N)
t0 = time.time()
otls = ref.determine_tls(p, hb, om, hits)
orec, ocig, _ = ref.select(p, hb, om, otls, hits, n_threads=8, n_clusters_hint=N)
print("oracle select %.1fs" % (time.time() - t0))
for fast in (1, 0):
    lib.emu_set_fast_sort(emu.h, fast)
    t0 = time.time()
    rec, cig = emu.select(hb, N, tls)
    d = parity_util.compare_records(orec, ocig, rec, cig)
    print("fast", fast, "emu select %.1fs" % (time.time() - t0), "diffs", len(d), emu.counters()['heavy_clusters'])
    for x in d[:3]: print(x)
