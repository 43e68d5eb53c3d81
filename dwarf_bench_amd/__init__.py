"""dwarf_bench_amd — MI355X (gfx950 / CDNA4) backend for dwarf_bench's data-parallel dwarf kernels.

  csrc/     hand-written HIP kernels + the C ABI of include/dbhip.h (libdbhip.so)
  host/     C++ host layer mirroring the reference's Dwarf / Meter / Registry / bench API (libdbench.so, CLI)
  ops.py    tensor-level front door (torch is plumbing: device memory, streams, torch.distributed)
  build.py  in-tree build (hipcc --offload-arch=gfx950, g++)

There is no CPU path: every op fails loudly if libdbhip.so is missing.
"""
__version__ = "0.1.0"
