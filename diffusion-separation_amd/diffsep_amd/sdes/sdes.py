"""SDE objects of the hot path: the reference's plug-point surface (sdes/sdes.py) over HIP kernels.

A Predictor / Corrector written against the reference API — `self.rsde.discretize(x, t, *args)`,
`self.sde.marginal_prob(x, t, *args)[1]`, `self.sde.mult_std(L, g)`, `self.sde.sde(x, t, mix)`
(sdes/predictors.py:60-66, sdes/correctors.py:109-128) — runs unchanged on these classes: every method that
touches a [B,S,T] tensor is one call into libdiffsep_hip.so (csrc/sde.hip, "SDE object surface"); torch only
holds the buffers.  The fused sampler (Engine.pc_sample) does not go through here.
"""
import math
import warnings

import torch

from .. import _lib, ops
from . import noise
from ..registry import Registry

SDERegistry = Registry("SDE")


def _t32(t, like):
    """time steps as a contiguous float32 device vector [B]"""
    return torch.as_tensor(t, dtype=torch.float32, device=like.device).reshape(-1).contiguous()


class ReverseSDE:
    """What `SDE.reverse(score_fn, probability_flow)` returns (the reference builds a subclass on the fly,
    sdes/sdes.py:109-173): sde() = reverse-time drift / diffusion, discretize() = the reverse-diffusion step
    coefficients; N, T and every other attribute of the forward SDE are forwarded."""

    def __init__(self, forward, score_fn, probability_flow=False):
        self._fwd, self._score_fn, self.probability_flow = forward, score_fn, probability_flow
        self.N = forward.N

    def __getattr__(self, name):  # (only reached for attributes not set above)
        return getattr(self._fwd, name)

    @property
    def T(self):
        return self._fwd.T

    def rsde_parts(self, x, t, *args):
        """sdes/sdes.py:142-161"""
        drift, diffusion = self._fwd.sde(x, t, *args)
        score = self._score_fn(x, t, *args)
        # total drift = drift - g^2 score (x 0.5 for the ODE): the reverse-drift kernel with f = drift, G = diffusion
        total = ops.sde_reverse_drift(drift, diffusion, score.contiguous(), self.probability_flow)
        return {"total_drift": total, "diffusion": torch.zeros_like(diffusion) if self.probability_flow else diffusion,
                "sde_drift": drift, "sde_diffusion": diffusion, "score_drift": total - drift, "score": score}

    def sde(self, x, t, *args):
        """sdes/sdes.py:132-140"""
        parts = self.rsde_parts(x, t, *args)
        return parts["total_drift"], parts["diffusion"]

    def discretize(self, x, t, *args, **kwargs):
        """rev_f = f - G^2 score (x 0.5 if probability_flow), rev_G = G (0 if probability_flow)
        (sdes/sdes.py:163-171)."""
        f, G = self._fwd.discretize(x, t, *args, **kwargs)
        score = self._score_fn(x, t, *args)
        rev_f = ops.sde_reverse_drift(f, G, score.contiguous(), self.probability_flow)
        return rev_f, (torch.zeros_like(G) if self.probability_flow else G)


class SDE:
    """Base: holds N (number of discretisation steps) — reference sdes/sdes.py:43-52."""

    def __init__(self, N):
        self.N = N

    @property
    def T(self):
        raise NotImplementedError

    def copy(self):
        raise NotImplementedError

    def sde(self, x, t, *args):
        raise NotImplementedError

    def marginal_prob(self, x, t, *args):
        raise NotImplementedError

    def prior_sampling(self, shape, *args):
        raise NotImplementedError

    def prior_logp(self, z):
        raise NotImplementedError("prior_logp for OU SDE not yet implemented!")  # as the reference (sdes.py:348-349)

    def reverse(self, score_model, probability_flow=False):
        """sdes/sdes.py:109-173"""
        return ReverseSDE(self, score_model, probability_flow)


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

    # ---- scalar tables (tiny [B] host-side tensors; the kernels recompute them in registers)
    def _cov_eigval(self, t):
        """sdes/sdes.py:296-309"""
        mult = self.sigma_min ** 2
        srp = self.ratiosig ** (2 * t)
        ev1 = mult * (srp - 1)
        ev2 = mult * (srp - torch.exp(-2.0 * self.d_lambda * t)) / (1.0 + self.d_lambda / self.logsig)
        return ev1, ev2

    def _var(self, t):
        """sdes/sdes.py:311-313"""
        ev1, ev2 = self._cov_eigval(t)
        return 0.5 * (ev1 + ev2)

    def sigma_mix(self, y):
        """Per-sample noise scale (None for MixSDE: the perturbation kernel does not depend on the mixture)."""
        return None

    # ---- the tensor-valued surface
    def sde(self, x, t, mix=None):
        """(drift = -lambda P x [B,S,T], diffusion = sigma_min r^t sqrt(2 ln r) [B])   sdes/sdes.py:275-284"""
        return ops.sde_coefficients(self.engine_config(), x.contiguous(), _t32(t, x), self.sigma_mix(mix))

    def discretize(self, x, t, *args, **kwargs):
        """f = drift dt, G = diffusion sqrt(dt).  The reference reads dt with getattr() on the kwargs DICT
        (sdes/sdes.py:103), which never finds it: the step is 1/N whatever the caller passes (quirk Q1), here too."""
        dt = 1.0 / self.N
        return ops.sde_coefficients(self.engine_config(), x.contiguous(), _t32(t, x),
                                    self.sigma_mix(args[0]) if args else None, f_scale=dt, g_scale=math.sqrt(dt))

    def _mean(self, x0, t):
        """(A + exp(-lambda t) P) x0   sdes/sdes.py:286-294"""
        return ops.sde_mean(self.engine_config(), x0.contiguous(), _t32(t, x0))

    def _std(self, t, *args):
        """L = sqrt(ev1) A + sqrt(ev2) P, dense [B,S,S]   sdes/sdes.py:315-320"""
        return ops.sde_std(self.engine_config(), _t32(t, t), self.ndim)

    def marginal_prob(self, x0, t, *args):
        """(mean, std) of p_t(x | x0)   sdes/sdes.py:322-324"""
        return self._mean(x0, t), self._std(_t32(t, x0), *args)

    @staticmethod
    def mult_std(std, x):
        """std @ x   sdes/sdes.py:326-328 (PriorMixSDE: einsum "bcdt,bdt->bct", :534-537)"""
        return ops.sde_mult_std(std, x.contiguous())

    def prior_sampling(self, shape, y):
        """x_T = 0.5 y + L(T) z   (sdes/sdes.py:334-346); z: sdes/noise.py (device generator, seeded by torch's)."""
        if tuple(shape) != tuple(y.shape):
            warnings.warn(f"Target shape {tuple(shape)} does not match shape of y {tuple(y.shape)}! Ignoring target shape.")
        B, _, T = y.shape
        z = noise.randn((B, self.ndim, T), y)
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
        if y is None or y.shape[1] != 1:
            raise ValueError("PriorMixSDE expects a single-channel mixture [B,1,T]")
        return ops.sde_sigma_mix(y.contiguous(), self.avg_len)

    def _std_sigma_mix(self, mix):
        """[B,1,T] like the reference (sdes/sdes.py:477-489)"""
        return self.sigma_mix(mix).unsqueeze(1)

    def sde(self, x, t, mix):
        """diffusion is per sample here: g(t) sigma_mix broadcast to [B,S,T]   sdes/sdes.py:451-470"""
        return super().sde(x, t, mix)

    def _std(self, t, mix):
        """L sigma_mix, dense [B,S,S,T]   sdes/sdes.py:515-532"""
        sm = self.sigma_mix(mix)
        return ops.sde_std(self.engine_config(), _t32(t, mix), self.ndim, T=mix.shape[-1], sigma_mix=sm)

    def marginal_prob(self, x0, t, mix):
        return self._mean(x0, t), self._std(_t32(t, x0), mix)
