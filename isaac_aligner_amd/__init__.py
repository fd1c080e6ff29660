"""isaac_aligner_amd -- MI355X (gfx950) implementation of Isaac's seed-and-extend hot path.

The compute lives in csrc/ (hand-written HIP kernels behind the C ABI of include/isaac_gpu.h, built into
libisaac_gpu.so); this package is the thin host-side mirror of the reference's interface for that path:

    options.default_params   isaac-align option defaults and `--seeds auto`
    gpu.Aligner              FindMatchesTransition / MatchSelector stand-ins calling the C ABI with torch device buffers
    synth                    synthetic references and read pairs (no genomes ship with the image)

There is no CPU implementation in this package: gpu.load_library() raises if the HIP library is missing or no GPU is usable.
"""
from . import abi, options  # noqa: F401
