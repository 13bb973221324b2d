"""pyrapose_amd -- MI355X-native (gfx950) implementation of the PyraPose hot path.

Host side = Python on PyTorch-ROCm tensors (device memory, streams, torch.distributed only);
compute = hand-written HIP kernels behind the C ABI of include/pyrapose_hip.h
(libpyrapose_hip.so, loaded by ``pyrapose_amd._lib``).  No CPU fallback exists.
"""
__version__ = "0.1.0"
