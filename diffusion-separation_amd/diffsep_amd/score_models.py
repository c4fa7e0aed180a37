"""ScoreModelNCSNpp with the reference constructor and forward contract
(models/score_models.py:10-138), backed by the HIP engine: forward(xt, time_cond, mix) runs
STFT -> NCSN++ -> iSTFT entirely in libdiffsep_hip.so.

A drop-in for hydra configs: `_target_: diffsep_amd.score_models.ScoreModelNCSNpp` accepts the same
kwargs as `models.score_models.ScoreModelNCSNpp`; weights arrive through load_state_dict() with the
reference's key layout (backbone.all_modules.{i}...., backbone.output_layer.*).
"""
import numpy as np
import torch

from . import _lib, synth
from .engine import Engine, pack_state_dict, param_table


class ScoreModelNCSNpp:
    def __init__(self, num_sources, stft_args, backbone_args, transform="exponent", spec_abs_exponent=0.5,
                 spec_factor=3.0, spec_trans_learnable=False, dtype="f16", device=None, init_seed=0, lib_kind=None):
        if transform != "exponent":
            raise NotImplementedError("only transform='exponent' runs on the accelerated path")
        if spec_trans_learnable:
            raise NotImplementedError("spec_trans_learnable is a training feature")
        if not stft_args.get("center", True) or stft_args.get("pad_mode", "constant") != "constant":
            raise NotImplementedError("STFT must be center=True, pad_mode='constant' (config/model/default.yaml:21-22)")
        ba = {k: v for k, v in dict(backbone_args).items() if k != "_target_"}
        # Every backbone argument that changes NCSNpp.forward and is not a parameter of the engine must hold the value
        # the engine implements (models/ncsnpp.py:45-70 constructor defaults): a checkpoint trained otherwise would
        # load and silently produce wrong scores.
        # (dropout does nothing at inference: any value loads)
        supported = dict(scale_by_sigma=True, nonlinearity="swish", resamp_with_conv=True,
                         conditional=True, fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True,
                         resblock_type="biggan", progressive="output_skip", progressive_input="input_skip",
                         progressive_combine="sum", init_scale=0.0, fourier_scale=16, image_size=256,
                         embedding_type="fourier", centered=False)
        for k, want in supported.items():
            if k in ba and (list(ba[k]) if isinstance(want, list) else ba[k]) != want and k != "init_scale":
                raise NotImplementedError(f"backbone_args.{k}={ba[k]!r}: the engine implements {want!r} only")
        # (arguments NCSNpp does not know are swallowed by its **unused_kwargs, ncsnpp.py:68: ignored here too)
        if len(tuple(ba.get("attn_resolutions", (16,)))) != 1:
            raise NotImplementedError("the engine implements exactly one attention resolution")
        # num_channels_in / num_channels_out of the config are overwritten with 2 S + 2 / 2 S exactly as the reference does
        # (models/score_models.py:24-26): whatever a checkpoint's config says there loads
        if stft_args["n_fft"] != 510 or stft_args["n_fft"] // 2 + 1 != 256:
            raise NotImplementedError("n_fft must be 510 (image height 256 = NCSNpp image_size; the DFT tables hold 510 taps)")
        self.num_sources = num_sources
        self.stft_args = dict(stft_args)
        self.spec_abs_exponent, self.spec_factor = spec_abs_exponent, spec_factor
        self.cfg = _lib.model_config(
            nf=ba.get("nf", 128), num_sources=num_sources, ch_mult=tuple(ba.get("ch_mult", (1, 1, 2, 2, 2, 2, 2))),
            num_res_blocks=ba.get("num_res_blocks", 2), attn_resolution=tuple(ba.get("attn_resolutions", (16,)))[0],
            n_fft=stft_args["n_fft"], hop=stft_args["hop_length"], spec_abs_exponent=abs(spec_abs_exponent),
            spec_factor=spec_factor, dtype={"bf16": _lib.BF16, "f16": _lib.F16, "fp16": _lib.F16, "f32": _lib.F32, "fp32": _lib.F32, "split": _lib.F32_SPLIT}[dtype])
        self.device = device
        self.lib_kind = lib_kind  # (Engine: which build of the library; None = by dtype)
        self._engine = None
        self._parent, self._version, self._built_version = None, 0, 0  # (twin(): weights / device follow the parent)
        # random init like the reference constructor (no checkpoint yet): synthetic variance-scaling weights
        self._state = synth.synth_state_dict([(n, s) for n, s, _ in param_table(self.cfg)], init_seed)

    def twin(self, dtype, lib_kind=None):
        """The same model in another precision mode; its engine is created on first use.  The twin FOLLOWS this model: it
        reads the weights and the device of its parent whenever it (re)creates its engine, and a later load_state_dict() /
        to() on the parent drops the twin's engine (a twin built once from a copy of __dict__ kept the weights and device
        of the moment it was made: an overflow fallback on stale weights returns finite but wrong samples)."""
        t = object.__new__(ScoreModelNCSNpp)
        t.__dict__.update(self.__dict__)
        t.cfg = _lib.ModelConfig.from_buffer_copy(self.cfg)
        t.cfg.dtype = {"bf16": _lib.BF16, "f16": _lib.F16, "f32": _lib.F32, "split": _lib.F32_SPLIT}[dtype]
        t.lib_kind, t._engine, t._parent, t._built_version = lib_kind, None, self, -1
        return t

    def _sync_with_parent(self):
        """twin only: adopt the parent's current weights / device; drop an engine built from older ones"""
        par = self._parent
        if par is None or self._built_version == par._version:
            return
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        self._state, self.device, self._built_version = par._state, par.device, par._version

    # ---- weights ---------------------------------------------------------------------------
    def param_names(self):
        return ["backbone." + n for n, _, _ in param_table(self.cfg)]

    def load_state_dict(self, state, strict=True):
        """Keys as in the reference ('backbone.all_modules.3.weight', ...); STFT window buffers are ignored."""
        new = {}
        for n, shape, _ in param_table(self.cfg):
            k = "backbone." + n
            if k not in state:
                if strict:
                    raise KeyError(f"missing key '{k}'")
                new[n] = self._state[n]
                continue
            v = state[k]
            v = v.detach().cpu().float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)
            if tuple(v.shape) != shape:
                raise ValueError(f"size mismatch for '{k}': {tuple(v.shape)} vs {shape}")
            new[n] = v
        self._state = new
        self._version += 1
        self._parent = None  # (a twin that gets its own weights stops following its parent)
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        return self

    def state_dict(self):
        return {"backbone." + n: torch.from_numpy(np.array(v)) for n, v in self._state.items()}

    def to(self, device):
        if device != self.device:
            self.device = device
            self._version += 1
            if self._engine is not None:  # (the engine is bound to one device: re-created there on next use)
                self._engine.close()
                self._engine = None
        return self

    def eval(self):
        return self

    def engine(self):
        """The device-resident engine (created lazily on the current / configured device)."""
        self._sync_with_parent()
        if self._engine is None:
            self._engine = Engine(self.cfg, pack_state_dict(self.cfg, self._state), device=self.device, lib_kind=self.lib_kind)
        return self._engine

    # ---- reference forward ---------------------------------------------------------------
    def forward(self, xt, time_cond, mix):
        return self.engine().score(xt, time_cond, mix)

    __call__ = forward
