"""gdkvm_amd -- MI355X-native (gfx950) GDKVM memory path: LKVA read, GDR write, KPFF, argmax+Dice.

The compute lives in hand-written HIP kernels behind the C ABI of include/gdkvm.h
(gdkvm_amd/libgdkvm_hip.so, loaded with ctypes), the inference build's encoder/decoder convolutions included;
PyTorch is used for device memory, streams, torch.distributed and -- in the training build only -- the strided / 1x1 /
stem convolutions (MIOpen).  There is no CPU or eager fallback: every op raises if the native library is missing or the
tensors are not on a gfx950 device.
"""
from . import ops  # noqa: F401

__all__ = ["ops"]
__version__ = "0.1.0"
