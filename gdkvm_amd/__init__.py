"""gdkvm_amd -- MI355X-native (gfx950) GDKVM memory path: LKVA read, GDR write, KPFF, argmax+Dice.

The compute lives in hand-written HIP kernels behind the C ABI of include/gdkvm.h
(gdkvm_amd/libgdkvm_hip.so, loaded with ctypes) -- the memory path, and the encoder/decoder convolutions of the default
model in BOTH builds (inference: fuse_for_inference(); training: forward, data and weight gradients, deterministic).
PyTorch is used for device memory, streams, autograd plumbing and torch.distributed.  A convolution whose shape the kernels
do not serve (other `widths`, odd sizes, an fp32 inference module) runs on the framework's library with a one-time
RuntimeWarning naming the layer -- an error under GDKVM_STRICT=1.  The memory path has no CPU or eager fallback: every op
raises if the native library is missing or the tensors are not on a gfx950 device.
"""
from . import ops  # noqa: F401

__all__ = ["ops"]
__version__ = "0.1.0"
