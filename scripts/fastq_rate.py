"""Experiment helper: device FASTQ -> BCL conversion rate (text resident in HBM)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
from isaac_aligner_amd import gpu, options, synth
contigs = synth.make_genome(2_000_000, seed=3, device="cuda", n_contigs=1)
bcl, _ = synth.make_read_pairs(contigs, 500_000, 150, seed=4, device="cuda")
text = synth.bcl_to_fastq(bcl.cpu().numpy(), 0, 150)
al = gpu.Aligner(options.default_params(150, 150), 0, contigs)
dev = torch.frombuffer(bytearray(text), dtype=torch.uint8).to("cuda")
out = torch.zeros_like(bcl)
al.fastq_to_bcl(dev, 0, bcl=out)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    _, n, _ = al.fastq_to_bcl(dev, 0, bcl=out)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print("fastq_to_bcl: %d reads, %.1f MB text, %.2f ms per call, %.1f GB/s text, %.1f M reads/s; kernel k_fq_records %.3f ms; identical to source: %s"
      % (n, len(text) / 1e6, dt * 1e3, len(text) / dt / 1e9, n / dt / 1e6, al.kernel_time_ms("fastq_to_bcl")[0], bool((out[:, :150] == bcl[:, :150]).all())))
