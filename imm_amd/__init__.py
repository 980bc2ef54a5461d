"""imm_amd — MI355X-native implementation of the IMM conditional-generation training step.

Host code (this package) is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
all arithmetic on the hot path runs in hand-written gfx950 HIP kernels behind the C-ABI of
libimm_hip.so (include/imm_hip.h).  There is no CPU fallback.
"""
__version__ = '0.1.0'
