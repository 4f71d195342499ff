"""MI355X-native engine for the cooperative-training hot path (FTN/STN encoder-decoders + latent-space hard-example
generator).  Host side: Python on PyTorch-ROCm (memory, streams, torch.distributed); compute: hand-written HIP kernels
in csrc/ behind the C-ABI of include/ctl_hip.h."""
__version__ = "0.1.0"
