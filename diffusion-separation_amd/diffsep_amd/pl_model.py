"""DiffSepModel — inference members of the reference LightningModule (pl_model.py:95-759) re-hosted on
the HIP engine: checkpoint loading incl. the EMA swap (pl_model.py:642-670), normalize_batch (:81-92),
forward = score_fn (:407-409), get_pc_sampler (:687-759), separate (:148-164).  Training members are out
of scope."""
import math
import warnings

import torch

from . import ops, sdes
from .engine import param_table
from .score_models import ScoreModelNCSNpp
from .sdes import MixSDE, PriorMixSDE


def cfg_get(cfg, path, default=None):
    """Read 'a.b.c' from dicts / attribute objects / OmegaConf nodes alike."""
    cur = cfg
    for k in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(k, None)
        else:
            try:
                cur = cur[k]
            except Exception:
                cur = getattr(cur, k, None)
    return default if cur is None else cur


def normalize_batch(batch):
    """pl_model.py:81-88: mix [B,1,T] -> zero mean / unit (unbiased) std per utterance, std clamped at 1e-5.
    The reduction + scaling of the mixture run in the HIP kernel; the optional target only reuses mean/std."""
    mix, tgt = batch
    if mix.shape[1] != 1:
        raise ValueError("normalize_batch expects a single-channel mixture [B,1,T]")
    mix_n, mean, std = ops.normalize_batch(mix.contiguous().float())
    if tgt is not None:
        tgt = (tgt - mean) / std
    return (mix_n, tgt), mean, std


def denormalize_batch(x, mean, std):
    return x * std + mean


def enhancement_config(nf=128):
    """config/model/nr.yaml restated (values only): 16 kHz, nf 128, spec_factor 0.15, PriorMixSDE."""
    cfg = default_config(nf=nf, n_speakers=2, fs=16000, spec_factor=0.15)
    cfg["model"]["sde"] = {"_target_": "sdes.sdes.PriorMixSDE", "ndim": 2, "d_lambda": 2.0, "sigma_min": 0.05,
                           "sigma_max": 0.5, "N": 30}
    return cfg


def default_config(nf=64, n_speakers=2, fs=8000, spec_factor=0.33):
    """config/model/default.yaml restated as a plain dict (values only)."""
    return {"model": {
        "n_speakers": n_speakers, "fs": fs, "t_eps": 0.03,
        "score_model": {"num_sources": n_speakers,
                        "stft_args": {"n_fft": 510, "hop_length": 128, "center": True, "pad_mode": "constant"},
                        "backbone_args": {"nf": nf}, "transform": "exponent", "spec_abs_exponent": 0.5,
                        "spec_factor": spec_factor, "spec_trans_learnable": False},
        "sde": {"ndim": n_speakers, "d_lambda": 2.0, "sigma_min": 0.05, "sigma_max": 0.5, "N": 30},
        "sampler": {"N": 30, "snr": 0.5, "corrector_steps": 1}}}


# dtype="split": fp32 tensors, every matrix product as three bf16 MFMAs on the hi / lo bf16 halves of both operands
# (DIFFSEP_F32_SPLIT): 4e-5 relative RMS / >= 79 dB from the exact fp32 engine after 60 NFE at 1.9x its speed
# (tools/probes/split_probe.py) — the fast mode that meets the parity bar.
# dtype="hybrid": such a split-fp32 engine evaluates the score of the FIRST HYBRID_HEAD_STEPS reverse steps, the f16 engine
# the rest.  A score error enters the state scaled by the step size G(t)^2, which is ~100x larger at t = 1 than at
# t = 0.03, so the rounding of the EARLY steps is what separates a 16-bit trajectory from the fp32 one (measured on bf16 in
# round 2, tools/hybrid_probe.py: fp32 for the LAST 5 / 15 / 25 steps buys nothing, for the FIRST 5 / 10 / 15 it buys 9 / 14 /
# 17 dB).  With the f16 engine (tools/probes/hybrid_f16_probe.py; SI-SDR against the exact fp32 engine, mean / min dB at
# nf = 64 | nf = 128; time of one batch relative to f16 alone):
#   K = 0:  51.1 / 44.4 | 38.9 / 32.7   1.00        K = 5:  60.1 / 56.3 | 56.9 / 55.1   1.23
#   K = 2:  55.9 / 51.8 | 50.9 / 47.2   1.09        K = 7:  61.9 / 58.9 | 59.9 / 58.5   1.32
#   K = 3:  57.5 / 54.6 | 53.3 / 49.7   1.14        K = 10: 63.9 / 60.8 | 62.6 / 61.3   1.46
# (round-4 kernels.)  The criterion K = 5 was picked by — and is gated on, tests/test_fullsize_gpu.py — is >= 50 dB on EVERY
# utterance at both widths: at most 0.004 dB of SI-SDR at a 20 dB operating point, and a factor 3 inside the 1e-2 relative RMS bar
# against the CPU oracle (measured 3.2e-3 at nf = 128, T = 32000).  With the round-5 / 6 kernels (packed half-precision GroupNorm +
# SiLU in the streamed-weight convolution) K = 5 measures 59 / 54 dB at nf = 64 and 57 / 54 dB at nf = 128 (bench.py `precision`,
# `nf128.hybrid`): 1 - 2 dB under the table, same side of the criterion; K = 3 would not be (49.7 dB minimum at nf = 128 already with
# the round-4 kernels).
HYBRID_HEAD_STEPS = 5


class DiffSepModel:
    def __init__(self, config, dtype="auto", device=None, init_seed=0, head_steps=None):
        """dtype: "auto" (default) = "f16" for backbones up to nf = 64 and "hybrid" for wider ones — the rounding of a
        16-bit engine grows with the width (f16 alone: 50 dB from the fp32 result at nf = 64, 36 - 39 dB at the published
        nf = 128; hybrid: 60 / 57 dB at 1.23x the time); "f16" ( 16-bit tensors in IEEE half precision — the half-precision build of the library — 50 dB
        mean / 46 dB min from the fp32 result after 60 network evaluations, inside the 1e-3 RMS parity bar, at the speed of
        "bf16"), "bf16" (the same kernels on bfloat16 tensors: 32 dB / 25 dB), "f32" (exact fp32 MFMAs: parity with the
        reference to 1e-7), "split" (fp32 tensors, bf16x3 matrix products: parity to 4e-5 at twice the speed of "f32") or
        "hybrid": a "split" engine for the first head_steps reverse steps, f16 for the rest (extensions: the reference has
        one precision)."""
        self.config = config
        if dtype == "auto":
            dtype = "f16" if int(cfg_get(config, "model.score_model.backbone_args.nf", 128)) <= 64 else "hybrid"
        sm = dict(cfg_get(config, "model.score_model"))
        sm.pop("_target_", None)
        sm["stft_args"] = dict(sm["stft_args"])
        sm["backbone_args"] = {k: v for k, v in dict(sm["backbone_args"]).items()}
        self.dtype = dtype
        self.score_model = ScoreModelNCSNpp(dtype="f16" if dtype == "hybrid" else dtype, device=device,
                                            init_seed=init_seed, **sm)
        self.tail_model, self.head_steps = None, 0
        if dtype == "hybrid":
            # the split-precision head engine: the SAME parameters as score_model in another mode (a twin holds no weights)
            self.tail_model = self.score_model.twin("split", lib_kind="f16")
            self.head_steps = HYBRID_HEAD_STEPS if head_steps is None else int(head_steps)
        sd = dict(cfg_get(config, "model.sde"))
        target = str(sd.pop("_target_", "sdes.sdes.MixSDE"))
        if target.endswith("PriorMixSDE"):
            self.sde = PriorMixSDE(**sd)  # speech enhancement (config/model/nr.yaml:30-37)
        elif target.endswith("MixSDE"):
            self.sde = MixSDE(**sd)
        else:
            raise NotImplementedError(f"SDE '{target}' is not on the accelerated path (MixSDE / PriorMixSDE)")
        self.t_eps = cfg_get(config, "model.t_eps", 0.03)
        self.t_max = self.sde.T
        self.normalize_batch = normalize_batch
        self.denormalize_batch = denormalize_batch
        self.fallback_batches = 0  # sampler calls repeated on fallback_model() after non-finite samples

    # ---- checkpoint ----------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, path, dtype="auto", device=None, use_ema=True, head_steps=None):
        """Lightning .ckpt / HF checkpoint.pt: {'state_dict', 'hyper_parameters': {'config'}, 'ema'}
        (pl_model.py:100,642-673).  Inference runs on the EMA shadow weights (pl_model.py:655-660)."""
        ckpt = torch.load(str(path), map_location="cpu", weights_only=False)
        config = ckpt["hyper_parameters"]["config"]
        model = cls(config, dtype=dtype, device=device, head_steps=head_steps)
        state = {k[len("score_model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("score_model.")}
        ema = ckpt.get("ema", None) if use_ema else None
        if ema is not None:
            shadow = ema["shadow_params"]
            names = ["backbone." + n for n, _, _ in param_table(model.score_model.cfg)]
            if len(shadow) == len(names) - 1:  # torch_ema tracks requires_grad params only: frozen Fourier W skipped
                names = [n for n in names if not n.endswith("all_modules.0.W")]
            if len(shadow) != len(names):
                raise ValueError(f"EMA has {len(shadow)} shadow params, model has {len(names)} parameters")
            for n, v in zip(names, shadow):
                if tuple(v.shape) != tuple(state[n].shape):
                    raise ValueError(f"EMA shadow parameter for '{n}' has shape {tuple(v.shape)}")
                state[n] = v
        model.load_state_dict(state)
        return model

    def to(self, device):
        """The device the engines live on.  The parameters stay host tensors (the engine holds the only device copy of the
        weights, repacked): this is not nn.Module.to — score_model.to(device) is, and the engines follow that as well."""
        if device != self.score_model.device:
            self.score_model.device = device
            self.score_model.weights_changed()
        # (tail_model and the fallback's score model are twins: they follow score_model's device and weights)
        return self

    def replica(self):
        """Another DiffSepModel over the SAME parameters with engines of its own (twins of score_model): what a caller that
        keeps several batches in flight on one GPU uses — one engine (weights repacked on the device + workspace) per HIP
        stream, one set of host parameters.  load_state_dict() / to() on this model reach every replica."""
        r = object.__new__(DiffSepModel)
        r.__dict__.update(self.__dict__)
        r.score_model = self.score_model.twin(self.score_model.cfg.dtype, self.score_model.lib_kind)
        r.tail_model = self.score_model.twin("split", lib_kind="f16") if self.tail_model is not None else None
        r._fallback, r.fallback_batches = None, 0
        return r

    def set_throughput_mode(self, on=True):
        """For callers with SEVERAL batches in flight on one GPU (evaluate / separate --streams K > 1, bench.py): engine option
        rw_quarter — a register-weight convolution whose blocks would get <= 4 tiles runs on a quarter of the CUs with four
        times the tiles per block, and the other batches' kernels take the rest of the chip (+2.5 % throughput at K = 4;
        one batch alone is 15 % slower in this mode, which is why it is not the engine's default).  Same arithmetic; a block's fp32
        partial sums of the GroupNorm statistics cover other tiles, so a 16-bit result moves at its own rounding level, as it does
        with the batch size (tests/test_round6_gpu.py); the fp32 / split engines are not touched."""
        self.score_model.set_engine_option("rw_quarter", int(bool(on)))
        return self

    def has_fallback(self):
        """Whether this mode has an overflow fallback at all — from the mode alone, without constructing it."""
        return self.dtype in ("f16", "fp16", "hybrid")

    def load_state_dict(self, state, strict=True):
        """Weights in the reference's score_model key layout ('backbone.all_modules....').  Every engine of the model follows:
        the hybrid head engine and the overflow fallback are twins of score_model (ScoreModelNCSNpp.twin), rebuilt from its
        parameters at their next use; the cached fallback object is dropped."""
        self.score_model.load_state_dict(state, strict=strict)
        self._fallback = None
        return self

    def fallback_model(self):
        """The parity-grade twin an f16 / hybrid run is repeated on when it returns non-finite samples: the same weights on a
        "split" engine (fp32 tensors — the range of fp32, 4e-5 from the fp32 result; a hybrid model's own head engine).  IEEE
        half precision ends at 65504 where bfloat16 and fp32 do not; an overflow anywhere in the network reaches the output as
        inf / NaN (residual trunk), which is what rerun_if_nonfinite looks for.  None for the modes without a range limit
        (bf16, f32, split).  (Round 3 fell back to bf16: finite, but 13 - 19 dB from the fp32 result at nf = 128.)"""
        if not self.has_fallback():
            return None
        if getattr(self, "_fallback", None) is None:
            fb = object.__new__(DiffSepModel)
            fb.__dict__.update(self.__dict__)
            fb.dtype, fb.tail_model, fb.head_steps, fb._fallback, fb.fallback_batches = "split", None, 0, None, 0
            fb.score_model = self.tail_model if self.tail_model is not None else self.score_model.twin("split")
            self._fallback = fb
        return self._fallback

    def rerun_if_nonfinite(self, result, rerun, what="a batch"):
        """THE overflow net of the 16-bit modes, in one place: get_pc_sampler wraps every sampler it returns in it, the
        asynchronous CLIs (evaluate, separate: several batches in flight) call it when they collect a batch.
        result = (x, nfe, ...) of a sampler of this model; rerun(fallback_model) -> the same request on the fallback.
        Finite: returned as is.  Otherwise the request is repeated on fallback_model() (counted in .fallback_batches, with
        a warning) — or FloatingPointError when the mode has no fallback or the repeat is non-finite too."""
        if bool(torch.isfinite(result[0]).all()):
            return result
        fb = self.fallback_model()
        if fb is None:
            raise FloatingPointError(f"non-finite samples for {what} (dtype {self.dtype})")
        warnings.warn(f"non-finite samples for {what} with dtype {self.dtype} (half precision overflows at 65504): "
                      f"repeating on the split-precision engine", RuntimeWarning, stacklevel=2)
        self.fallback_batches = getattr(self, "fallback_batches", 0) + 1
        again = rerun(fb)
        if not bool(torch.isfinite(again[0]).all()):
            raise FloatingPointError(f"non-finite samples for {what} on the split-precision engine too")
        return again

    def tail_engine(self):
        """the (split-)fp32 engine of dtype="hybrid" (None otherwise)"""
        return self.tail_model.engine() if self.tail_model is not None and self.head_steps > 0 else None

    def eval(self, no_ema=False):
        return self

    def engine(self):
        return self.score_model.engine()

    # ---- score function --------------------------------------------------------------------
    def forward(self, xt, time, mix):
        return self.score_model(xt, time, mix)

    __call__ = forward

    # ---- sampler factory (pl_model.py:687-759) ---------------------------------------------
    def get_pc_sampler(self, predictor_name, corrector_name, y, N=None, minibatch=None, schedule=None, check_finite=True,
                       **kwargs):
        """check_finite (extension): the returned sampler looks at its samples and repeats the request on fallback_model()
        when an f16 / hybrid run overflowed (rerun_if_nonfinite).  That look synchronises with the sampler's stream: a caller
        that keeps several batches in flight passes check_finite=False and calls rerun_if_nonfinite when it collects one."""
        sde = self.sde.copy()
        sde.N = self.sde.N if N is None else N
        kwargs = {"eps": self.t_eps, **kwargs}

        def make_on(model, y_part):
            if schedule is None:
                return sdes.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=model, y=y_part, **kwargs)
            return sdes.get_pc_scheduled_sampler(predictor_name, corrector_name, sde=sde, score_fn=model, y=y_part,
                                                 schedule=schedule, **kwargs)

        def make(y_part):
            inner = make_on(self, y_part)
            if not check_finite or not self.has_fallback():  # (decided from the mode: the fallback is BUILT at the first overflow)
                return inner
            return lambda: self.rerun_if_nonfinite(inner(), lambda fb: make_on(fb, y_part)())

        if minibatch is None:
            return make(y)

        seed0 = kwargs.pop("seed", None)
        if kwargs.get("lengths") is not None or kwargs.get("seeds") is not None:
            raise ValueError("lengths= / seeds= describe one engine batch; they cannot be combined with minibatch=")

        def batched_sampling_fn():
            samples, ns, inter = [], [], []
            for i in range(int(math.ceil(y.shape[0] / minibatch))):
                # an explicit seed is advanced per minibatch: the same seed would give every minibatch of the same
                # shape the same device noise
                if seed0 is not None:
                    kwargs["seed"] = (int(seed0) + 0x9E3779B97F4A7C15 * i) % (1 << 64)
                sample, n, *other = make(y[i * minibatch:(i + 1) * minibatch])()
                samples.append(sample)
                ns.append(n)
                if other:
                    inter.append(other[0])
            samples = torch.cat(samples, dim=0)
            return (samples, ns, inter) if inter else (samples, ns)

        return batched_sampling_fn

    def separate(self, mix, **kwargs):
        (mix_n, _), mean, std = self.normalize_batch((mix, None))
        skw = dict(cfg_get(self.config, "model.sampler", {}))
        skw.update(kwargs)
        est, *others = self.get_pc_sampler("reverse_diffusion", "ald2", mix_n, **skw)()
        return self.denormalize_batch(est, mean, std)
