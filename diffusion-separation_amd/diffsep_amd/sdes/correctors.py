"""Correctors (reference sdes/correctors.py): registry names 'ald2' and 'none' on the accelerated path."""
import abc

from .. import ops
from . import noise
from ..registry import Registry
from .sdes import MixSDE

CorrectorRegistry = Registry("Corrector")


class Corrector(abc.ABC):
    def __init__(self, sde, score_fn, snr, n_steps):
        self.sde, self.score_fn, self.snr, self.n_steps = sde, score_fn, snr, n_steps
        self.rsde = sde.reverse(score_fn)  # (sdes/correctors.py:14-19)

    @abc.abstractmethod
    def update_fn(self, x, t, *args, **kwargs):
        ...


@CorrectorRegistry.register("ald2")
class AnnealedLangevinDynamics2(Corrector):
    """n_steps x { x_mean = x + 2 snr^2 L L score ; x = x_mean + 2 snr L z }  (sdes/correctors.py:109-128)."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if not isinstance(sde, MixSDE):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t, *args, **kwargs):
        x_mean = x
        smix = self.sde.sigma_mix(args[0]) if args else None
        for _ in range(self.n_steps):
            score = self.score_fn(x, t, *args)
            z = noise.randn_like(x)
            x, x_mean = ops.sde_corrector_update(self.sde.engine_config(), self.snr, x.contiguous(), t.contiguous(),
                                                 score, z, smix)
        return x, x_mean


@CorrectorRegistry.register("ald")
class AnnealedLangevinDynamics(Corrector):
    """The scalar-std annealed Langevin corrector (sdes/correctors.py:58-91): std = sqrt(sum_j (L L)[0, j]) = sqrt(ev1);
    step = 2 (snr std)^2; x_mean = x + step g; x = x_mean + sqrt(2 step) z.  MixSDE only, like the reference."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        if type(sde) is not MixSDE:
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")

    def update_fn(self, x, t, *args, **kwargs):
        x_mean = x
        for _ in range(self.n_steps):
            score = self.score_fn(x, t, *args)
            z = noise.randn_like(x)
            x, x_mean = ops.sde_corrector_update(self.sde.engine_config(), self.snr, x.contiguous(), t.contiguous(),
                                                 score, z, None, variant=1)
        return x, x_mean


@CorrectorRegistry.register("langevin")
class LangevinCorrector(Corrector):
    """step = 2 (snr <||z_b||> / <||g_b||>)^2 from batch-mean norms (sdes/correctors.py:35-55)."""

    def update_fn(self, x, t, *args, **kwargs):
        x_mean = x
        for _ in range(self.n_steps):
            score = self.score_fn(x, t, *args)
            z = noise.randn_like(x)
            x, x_mean = ops.sde_langevin_update(self.snr, x.contiguous(), score, z)
        return x, x_mean


@CorrectorRegistry.register("none")
class NoneCorrector(Corrector):
    """sdes/correctors.py:131-141.  The reference's update_fn returns the 1-tuple `(x,)`, which its own sampler loop
    (`xt, xt_mean = corrector.update_fn(...)`, sdes/__init__.py:179) cannot unpack: corrector "none" raises there.  The mirror returns
    the pair the loop needs."""

    def __init__(self, *args, **kwargs):
        self.snr, self.n_steps = 0, 0

    def update_fn(self, x, t, *args, **kwargs):
        return x, x
