"""MI355X-native Speculative Jacobi Decoding engine (hot path of tyshiwo1/Accelerating-T2I-AR-with-SJD).

Import as ``sjd_amd`` (alias package at the repo root).  Layout:
  csrc/        hand-written HIP kernels (gfx950) + the C-ABI (include/sjd_hip.h) -> libsjd_hip.so
  _lib.py      ctypes loader (fails loudly when the .so is missing)
  ops.py       torch-tensor wrappers over the C-ABI
  grammar.py   host-side integer grammar state -> per-row rules (mirrors the reference's 3-dim processors)
  engine.py    static-shape SJD iteration driver
  backbones.py PyTorch-ROCm transformer definitions (attention = kernel K1)
  scheduler/   mirror of the reference's scheduler/ entry points (renew_sampler, ...)
"""
__version__ = "0.1.0"
