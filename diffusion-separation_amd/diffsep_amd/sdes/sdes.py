"""SDE objects of the hot path.  Mirrors the constructor / attribute surface of the reference's
sdes/sdes.py (MixSDE :180-349) that the sampler touches; all tensor arithmetic is delegated to the
HIP kernels (csrc/sde.hip) through the C-ABI."""
import math

import torch

from .. import _lib, ops
from ..registry import Registry

SDERegistry = Registry("SDE")


class SDE:
    """Base: holds N (number of discretisation steps) — reference sdes/sdes.py:43-52."""

    def __init__(self, N):
        self.N = N

    @property
    def T(self):
        raise NotImplementedError

    def copy(self):
        raise NotImplementedError


@SDERegistry.register("mix")
class MixSDE(SDE):
    """dx = -lambda P x dt + g(t) dw with the source-mixing structure of sdes/sdes.py:217-349."""

    def __init__(self, ndim, d_lambda, sigma_min, sigma_max, N=1000):
        super().__init__(N)
        self.ndim, self.d_lambda, self.sigma_min, self.sigma_max = ndim, d_lambda, sigma_min, sigma_max
        self.ratiosig = sigma_max / sigma_min
        self.logsig = math.log(self.ratiosig)

    @property
    def T(self):
        return 1.0

    def copy(self):
        return MixSDE(self.ndim, self.d_lambda, self.sigma_min, self.sigma_max, N=self.N)

    def engine_config(self):
        return dict(kind=_lib.SDE_MIX, ndim=self.ndim, d_lambda=self.d_lambda, sigma_min=self.sigma_min,
                    sigma_max=self.sigma_max)

    def _cov_eigval(self, t):
        """sdes/sdes.py:296-309 (tiny [B] host-side tensors; the kernels recompute these in registers)."""
        mult = self.sigma_min ** 2
        srp = self.ratiosig ** (2 * t)
        ev1 = mult * (srp - 1)
        ev2 = mult * (srp - torch.exp(-2.0 * self.d_lambda * t)) / (1.0 + self.d_lambda / self.logsig)
        return ev1, ev2

    def sigma_mix(self, y):
        """Per-sample noise scale (None for MixSDE: the perturbation kernel does not depend on the mixture)."""
        return None

    def prior_sampling(self, shape, y):
        """x_T = 0.5 y + L(T) z   (sdes/sdes.py:334-346); z from torch's generator like the reference."""
        B, _, T = y.shape
        z = torch.randn((B, self.ndim, T), dtype=y.dtype, device=y.device)
        return ops.sde_prior(self.engine_config(), y.contiguous(), z, self.sigma_mix(y))


@SDERegistry.register("priormix")
class PriorMixSDE(MixSDE):
    """MixSDE whose noise level follows the local energy of the mixture (speech enhancement, config/model/nr.yaml):
    sigma_mix = 0.5 sqrt(clamp(avg_pool1d(mix^2, avg_len), 1e-4)) scales L(t) and g(t) per sample
    (reference sdes/sdes.py:352-590)."""

    def __init__(self, ndim, d_lambda, sigma_min, sigma_max, N=1000, avg_len=510):
        super().__init__(ndim, d_lambda, sigma_min, sigma_max, N=N)
        self.avg_len = avg_len

    def copy(self):
        return PriorMixSDE(self.ndim, self.d_lambda, self.sigma_min, self.sigma_max, N=self.N, avg_len=self.avg_len)

    def engine_config(self):
        c = super().engine_config()
        c.update(kind=_lib.SDE_PRIORMIX, avg_len=self.avg_len)
        return c

    def sigma_mix(self, y):
        if y.shape[1] != 1:
            raise ValueError("PriorMixSDE expects a single-channel mixture [B,1,T]")
        return ops.sde_sigma_mix(y.contiguous(), self.avg_len)
