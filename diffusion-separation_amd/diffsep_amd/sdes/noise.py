"""The N(0, 1) draws of the step-by-step sampler loop.

The reference calls `torch.randn_like` at these sites (prior: sdes/sdes.py:345; correctors: sdes/correctors.py:47,82,117;
predictors: sdes/predictors.py:47,62 — quirk Q7 fixes their order).  Here a draw is ONE launch of the library's own generator
(`diffsep_randn`: Philox4x32-10 + Box-Muller, csrc/sde.hip), seeded with one draw of torch's HOST generator — so
`torch.manual_seed` governs reproducibility exactly as it does in the reference, and PyTorch computes nothing on the device.
`set_source(fn)` installs another source (tests inject the golden vectors' noise through it); predictors / correctors a
user writes against the reference API call `torch.randn_like` themselves and are not affected.
"""
import torch

from .. import ops

_source = None


def set_source(fn):
    """fn(shape, like) -> tensor, or None for the device generator.  Returns the previous source."""
    global _source
    prev, _source = _source, fn
    return prev


def randn(shape, like):
    if _source is not None:
        return _source(tuple(shape), like)
    n = 1
    for s in shape:
        n *= int(s)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return ops.randn(n, seed, 0, device=like.device).view(*shape).to(like.dtype)


def randn_like(x):
    return randn(tuple(x.shape), x)
