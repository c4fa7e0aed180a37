"""diffsep_amd — MI355X-native reverse-diffusion separation engine (host side).

Mirrors the inference API surface of fakufaku/diffusion-separation for ONE hot path:
  separate / evaluate entry points, DiffSepModel.get_pc_sampler, sdes.{predictors,correctors}, ScoreModelNCSNpp.
All arithmetic runs in libdiffsep_hip.so (hand-written HIP for gfx950) through the C-ABI in
include/diffsep_hip.h; PyTorch provides device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
