"""animal2vec_amd -- MI355X-native (gfx950) implementation of animal2vec's raw-waveform pretraining
hot path.  Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed only);
device side: hand-written HIP kernels behind the C ABI in include/a2v.h (liba2v_hip.so)."""
__version__ = "0.1.0"
