"""Experiment helper: per-cluster cost of the select stage on the CPU emulation (which clusters form the tail)."""
import sys, os, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import oracle_lib, hostemu_lib
from isaac_aligner_amd import options, synth, abi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
contigs = synth.make_genome(G, seed=11, device="cpu", n_contigs=2)
bcl, truth = synth.make_read_pairs(contigs, N, 150, seed=12, device="cpu")
hb = bcl.numpy()
o = oracle_lib.load()
cb = [bytes(c.numpy()) for c in contigs]
ref = o.reference(cb)
ref.build_index()
ep = options.default_params(150, 150)
om, hits = ref.find_matches(ep, hb, N)
print('matches', len(om))
lib = hostemu_lib.load()
emu = hostemu_lib.Emu(lib, ep, cb, hits)
emu.set_matches(om, N)
tls = emu.determine_tls(hb, N)
times = np.zeros(N)
lib.emu_set_cluster_times(emu.h, times.ctypes.data_as(C.c_void_p))
rec, cig = emu.select(hb, N, tls)
order = np.argsort(-times)
print("total %.2f s, top-20 share %.2f, top-100 share %.2f, median %.1f us" % (times.sum(), times[order[:20]].sum() / times.sum(), times[order[:100]].sum() / times.sum(), np.median(times) * 1e6))
for c in order[:20]:
    r = rec[2 * c: 2 * c + 2]
    print(c, "%.1f ms" % (times[c] * 1e3), "flags", r['flags'], "mapq/score", r['alignment_score'], "reserved", r['reserved'], 'pos', r['f_strand_position'] if 'f_strand_position' in r.dtype.names else '')
print(emu.counters())
cands, ccig = emu.build_fragments(hb, N)
cl = cands['cluster']
for c in order[:8]:
    k = cands[cl == c]
    st = np.zeros(7, np.uint64); lib.emu_cluster_job_stats(emu.h, C.c_uint32(int(c)), st.ctypes.data_as(C.c_void_p))
    print("   jobs/valid/cands/maxcands/gapped/fallback/windowbases", list(map(int, st)))
    print(c, "%.2f ms" % (times[c] * 1e3), "n0", int((k['read_index'] == 0).sum()), "n1", int((k['read_index'] == 1).sum()),
          "mm", list(k['mismatch_count']), "pos", list(k['position']), 'rev', list(k['reverse']))
