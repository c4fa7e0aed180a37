"""Sampler factories with the reference signatures (sdes/__init__.py:46-190).

get_pc_sampler(...) returns a closure `pc_sampler() -> (x, nfe[, intermediates])`.  When score_fn is an
engine-backed model and the request is the default path (reverse_diffusion + ald2/none, no
intermediates) the whole loop runs as ONE engine call (hipGraph-replayed network evaluations, fused
update kernels, on-device Philox noise seeded from torch's generator); otherwise the generic
step-by-step loop below runs the same kernels through the predictor / corrector objects.
"""
import math

import torch

from .correctors import Corrector, CorrectorRegistry
from .predictors import Predictor, PredictorRegistry, ReverseDiffusionPredictor
from .sdes import MixSDE, PriorMixSDE, SDERegistry

__all__ = ["PredictorRegistry", "CorrectorRegistry", "SDERegistry", "Predictor", "Corrector", "MixSDE", "PriorMixSDE",
           "get_pc_sampler", "get_pc_scheduled_sampler"]


def _timesteps(sde, eps, schedule, device):
    """None -> linspace(T, eps, N) (sdes/__init__.py:175); scheduled samplers use N+1 points
    (sdes/__init__.py:91-111) but still step with dt = 1/N (reference quirk Q1)."""
    if schedule is None:
        return torch.linspace(sde.T, eps, sde.N, device=device)
    if schedule == "linear":
        return torch.linspace(sde.T, eps, sde.N + 1, device=device)
    if schedule == "log":
        return torch.logspace(math.log10(sde.T), math.log10(eps), sde.N + 1, base=10, device=device)
    if schedule == "revlog":
        return torch.logspace(math.log10(eps), math.log10(sde.T), sde.N + 1, base=10, device=device).flip(dims=(0,))
    raise NotImplementedError(f"Schedule '{schedule}' does not exist")


def _engine_of(score_fn):
    """The engine behind a score function: an engine-backed model (diffsep_amd's DiffSepModel / ScoreModelNCSNpp), or the
    REFERENCE's DiffSepModel holding a diffsep_amd ScoreModelNCSNpp as .score_model — its forward is nothing but
    `self.score_model(xt, time, mix)` (pl_model.py:407-409), so the fused sampler computes the same thing."""
    eng = getattr(score_fn, "engine", None)
    if eng is None:
        eng = getattr(getattr(score_fn, "score_model", None), "engine", None)
    return eng() if callable(eng) else eng


class _HybridScore:
    """score_fn of the step-by-step loop on a dtype="hybrid" model: the split-precision head model while .head is set (the
    first head_steps reverse steps), the 16-bit model after — the schedule the fused sampler runs inside one engine call."""

    def __init__(self, model):
        self.model, self.head, self.head_steps = model, True, int(getattr(model, "head_steps", 0))

    def __call__(self, xt, t, mix):
        return (self.model.tail_model if self.head else self.model.score_model)(xt, t, mix)


def _make_sampler(predictor_name, corrector_name, sde, score_fn, y, true_mean, denoise, eps, snr, corrector_steps,
                  probability_flow, intermediate, schedule, seed=None, lengths=None, seeds=None):
    predictor = PredictorRegistry.get_by_name(predictor_name)(sde, score_fn, probability_flow=probability_flow)
    corrector = CorrectorRegistry.get_by_name(corrector_name)(sde, score_fn, snr=snr, n_steps=corrector_steps)
    eng = _engine_of(score_fn)
    fused = (eng is not None and not intermediate and true_mean is None
             and predictor_name in ("reverse_diffusion", "euler_maruyama", "none")
             and corrector_name in ("ald2", "ald", "langevin", "none") and isinstance(sde, MixSDE)
             and not (corrector_name == "ald" and type(sde) is not MixSDE))

    if not fused and (seed is not None or lengths is not None or seeds is not None):
        # the step-by-step loop draws from torch's global generator like the reference (torch.manual_seed governs it)
        # and has no notion of a zero-padded batch
        raise ValueError("seed= / seeds= / lengths= are extensions of the fused engine sampler; this request "
                         "(intermediate, true_mean or a user predictor / corrector) runs the generic loop")

    # dtype="hybrid" outside the fused sampler (intermediate=True, true_mean, a user-written predictor / corrector — calls the
    # reference supports on any model): the step-by-step loop keeps the schedule — the score of the first head_steps reverse
    # steps comes from the model's split-precision engine, the rest from its 16-bit engine — through a score function that
    # the loop switches per step.
    hybrid = None
    if not fused and getattr(score_fn, "tail_engine", lambda: None)() is not None:
        hybrid = _HybridScore(score_fn)
        predictor = PredictorRegistry.get_by_name(predictor_name)(sde, hybrid, probability_flow=probability_flow)
        corrector = CorrectorRegistry.get_by_name(corrector_name)(sde, hybrid, snr=snr, n_steps=corrector_steps)

    def pc_sampler():
        with torch.no_grad():
            ns = sde.N * (corrector.n_steps + 1)
            if fused:
                # torch.manual_seed() governs reproducibility; seed=... (an extension) fixes the device RNG seed of
                # this sampler explicitly, e.g. when several samplers are driven from different threads
                s_ = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
                ts = None if schedule is None else _timesteps(sde, eps, schedule, "cpu").numpy()
                tail = getattr(score_fn, "tail_engine", lambda: None)()
                x, _ = eng.pc_sample(y, sde.engine_config(), N=sde.N, corrector_steps=corrector.n_steps, snr=snr,
                                     eps=eps, denoise=denoise, predictor=predictor_name, corrector=corrector_name,
                                     seed=s_, timesteps=ts, lengths=lengths, seeds=seeds, tail=tail,
                                     head_steps=getattr(score_fn, "head_steps", 0) if tail is not None else 0)
                return x, ns
            im = []
            xt = sde.prior_sampling((true_mean if true_mean is not None else y).shape,
                                    true_mean if true_mean is not None else y)
            ts = _timesteps(sde, eps, schedule, y.device)
            xt_mean = xt
            for i in range(sde.N):
                if hybrid is not None:
                    hybrid.head = i < hybrid.head_steps
                vec_t = torch.ones(y.shape[0], device=y.device) * ts[i]
                xt, xt_mean = corrector.update_fn(xt, vec_t, y)
                if intermediate:
                    im.append((xt, xt_mean))
                xt, xt_mean = predictor.update_fn(xt, vec_t, y)
            res = xt_mean if denoise else xt
            return (res, ns, im) if intermediate else (res, ns)

    return pc_sampler


def get_pc_sampler(predictor_name, corrector_name, sde, score_fn, y, true_mean=None, denoise=True, eps=3e-2, snr=0.1,
                   corrector_steps=1, probability_flow=False, intermediate=False, **kwargs):
    """Reference: sdes/__init__.py:132-190.  Extensions through **kwargs (fused engine path only): seed, and for a
    zero-padded batch of utterances of different lengths lengths=[B] (+ seeds=[B], per-utterance RNG seeds)."""
    return _make_sampler(predictor_name, corrector_name, sde, score_fn, y, true_mean, denoise, eps, snr,
                         corrector_steps, probability_flow, intermediate, None, seed=kwargs.get("seed"),
                         lengths=kwargs.get("lengths"), seeds=kwargs.get("seeds"))


def get_pc_scheduled_sampler(predictor_name, corrector_name, sde, score_fn, y, denoise=True, true_mean=None, eps=3e-2,
                             snr=0.1, corrector_steps=1, probability_flow=False, intermediate=False,
                             schedule="linear", **kwargs):
    """Reference: sdes/__init__.py:46-129."""
    return _make_sampler(predictor_name, corrector_name, sde, score_fn, y, true_mean, denoise, eps, snr,
                         corrector_steps, probability_flow, intermediate, schedule, seed=kwargs.get("seed"),
                         lengths=kwargs.get("lengths"), seeds=kwargs.get("seeds"))
