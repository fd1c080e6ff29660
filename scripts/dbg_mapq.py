import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
from isaac_aligner_amd import gpu, options, synth
gpu.load_library()   # ISAAC_GPU_LIBRARY selects a debug build (-DISAAC_DEBUG_MAPQ, -DISAAC_PROFILE_HEAVY, -DISAAC_KERNEL_STAMPS)
contigs = synth.make_genome(4_000_000, seed=11, device="cuda", n_contigs=2)
bcl, truth = synth.make_read_pairs(contigs, 300_000, 150, seed=12, device="cuda")
p = options.default_params(150, 150)
al = gpu.Aligner(p, 0, contigs)
al.build_index(annotate_neighbors=False)
m,o,h = al.find_matches(bcl); al.set_loaded_contigs(h)
tls = al.determine_tls(bcl,m,o)
al.select(bcl,m,o,tls); al.synchronize()
print(al.counters())
