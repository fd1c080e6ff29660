import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import oracle_lib
from isaac_aligner_amd import gpu, options
from parity_util import make_inputs
contigs, bcl, truth = make_inputs(read_length=150, n_pairs=100, seed=1, genome_bases=400000)
p = options.default_params(150, 150)
al = gpu.Aligner(p, 0, contigs)
al.build_index(annotate_neighbors=True)
o = oracle_lib.load()
ref = o.reference(contigs)
a = np.sort(ref.build_index(), order=["kmer", "position"])
b = np.sort(al.get_index().view(oracle_lib.INDEX_DTYPE), order=["kmer", "position"])
print(len(a), len(b))
bad = np.nonzero(a["position"] != b["position"])[0]
print("differing", len(bad), "flag-only", int(((a["position"][bad] ^ b["position"][bad]) == 1).sum()))
print("oracle flagged", int((a["position"] & 1).sum()), "gpu flagged", int((b["position"] & 1).sum()))
for i in bad[:10]:
    print(hex(a["kmer"][i]), hex(a["position"][i]), hex(b["position"][i]))
