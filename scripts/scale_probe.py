#!/usr/bin/env python3
"""One pass of the path over a human-sized synthetic reference, stage by stage, with timings (a development aid)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from isaac_aligner_amd import gpu, options, synth

ap = argparse.ArgumentParser()
ap.add_argument("--genome-bases", type=int, default=3_100_000_000)
ap.add_argument("--pairs", type=int, default=1_000_000)
ap.add_argument("--read-length", type=int, default=150)
ap.add_argument("--no-neighbors", action="store_true")
ap.add_argument("--batches", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda", 0)
def mem():
    f, t = torch.cuda.mem_get_info(dev); return round((t - f) / 1e9, 1)
t = time.time(); g = synth.make_human_like_genome(a.genome_bases, seed=3, device=dev); torch.cuda.synchronize(); print("genome %.1f s, %d contigs, hbm %.1f GB" % (time.time() - t, len(g), mem()), flush=True)
torch.cuda.empty_cache()
L = a.read_length
p = options.default_params(L, L)
al = gpu.Aligner(p, 0, g, deferred_completion=True)
print("contigs loaded, hbm", mem(), flush=True)
t = time.time(); n = al.build_index(annotate_neighbors=not a.no_neighbors); print("index %.1f s, %d entries, hbm %.1f GB" % (time.time() - t, n, mem()), flush=True)
print("mask offsets", al.mask_offsets()[[0, 1, 2, 32, 63, 64]])
batches = []
t = time.time()
for b in range(a.batches):
    batches.append(synth.make_read_pairs(g, a.pairs, L, seed=100 + b, device=dev, avoid_gaps=True)[0])
torch.cuda.synchronize(); print("reads %.1f s" % (time.time() - t), flush=True)
tls = None
for it in range(2):
    al.reset_timers()
    t = time.time()
    found = [al.find_matches(b) for b in batches]
    torch.cuda.synchronize(); t_find = time.time() - t
    hits = found[0][2]
    al.set_loaded_contigs(np.ones_like(hits))
    if tls is None:
        t = time.time(); tls = al.determine_tls(batches[0], found[0][0], found[0][1]); print("tls %.2f s" % (time.time() - t), tls.astuple(), flush=True)
    t = time.time()
    outs = [al.select(b, m, o, tls) for b, (m, o, _) in zip(batches, found)]
    al.synchronize(); t_sel = time.time() - t
    print("iteration %d: find %.1f ms/batch, select %.1f ms/batch, %.2f M reads/s, hbm %.1f GB" % (it, 1e3 * t_find / len(batches), 1e3 * t_sel / len(batches), 2e-6 * a.pairs * len(batches) / (t_find + t_sel), mem()), flush=True)
c = al.counters()
print(json.dumps(c))
names = ("find_matches", "compact_matches", "build_fragments", "align_candidates", "finish_candidates", "indel_fragments", "gapped_fragments", "finish_fragments", "plan_rescue", "rescue_windows", "rescue_align",
         "rescue_gapped_plan", "gapped_rescue", "select_order", "select", "select_heavy", "select_residual")
print({k: round(al.kernel_time_ms(k)[0], 3) for k in names})
rec = outs[0][0].cpu().numpy().view(gpu.abi.FRAGMENT_DTYPE).reshape(-1)
print("unmapped %.4f, mapq0 %.4f, mapq>=30 %.4f, proper %.4f, overflow %d" % ((rec["flags"] & 2).astype(bool).mean(), (rec["mapq"] == 0).mean(), (rec["mapq"] >= 30).mean(), (rec["flags"] & 256).astype(bool).mean(), int((rec["reserved"] & 5).astype(bool).sum())))
