"""times isaac_gpu_bam_records on a mid-sized tile set (no index build at GRCh38 scale): python scripts/bam_probe.py [pairs] [tiles]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from isaac_aligner_amd import gpu, options, synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L = 150
g = synth.make_human_like_genome(200_000_000, seed=3, device="cuda")
a = gpu.Aligner(options.default_params(L, L), 0, g)
a.build_index()
tiles = []
for t in range(n_tiles):
    bcl = synth.make_read_pairs(g, pairs, L, seed=5 + t, avoid_gaps=True)[0]
    rec, cig = a.align_tile(bcl, tile=1 + t)
    packed, _ = a.compact_cigars(rec, cig)
    tiles.append((bcl, rec, packed.clone(), "SYNTH:1:%d:" % (1 + t)))
    del cig
buf = torch.empty(n_tiles * pairs * 2 * 340, dtype=torch.uint8, device="cuda")
a.bam_records(tiles, out=buf)
a.reset_timers()
torch.cuda.synchronize(); t0 = time.perf_counter()
s, n, un = a.bam_records(tiles, out=buf)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("records", n, "bytes", s.numel(), "ms %.2f" % (dt * 1e3), "order %.2f encode %.2f" % (a.kernel_time_ms("bam_order")[0], a.kernel_time_ms("bam_encode")[0]), "GB/s %.1f" % (s.numel() / dt / 1e9))
if "--check" in sys.argv:
    import oracle_lib
    o = oracle_lib.load()
    host = [(b.cpu().numpy(),) + a.records_to_numpy(r, c) + (p,) for b, r, c, p in tiles]
    want = o.bam_records(host, [L, L])[0]
    print("identical to the oracle:", s.cpu().numpy().tobytes() == want)
