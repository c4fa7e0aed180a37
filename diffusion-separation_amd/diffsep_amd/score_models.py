"""ScoreModelNCSNpp with the reference constructor and forward contract (models/score_models.py:10-138), backed by the HIP
engine: forward(xt, time_cond, mix) runs STFT -> NCSN++ -> iSTFT entirely in libdiffsep_hip.so.

It is a torch.nn.Module with the reference's parameter tree — `backbone.all_modules.{i}....`, `backbone.output_layer.*`
as nn.Parameters in the reference's registration order, `stft.window` / `stft_inv.window` as buffers — so that the
reference's own LightningModule can hold it: `_target_: diffsep_amd.score_models.ScoreModelNCSNpp` in
config/model/default.yaml:15 makes `DiffSepModel.__init__` (pl_model.py:105) build this class, `self.parameters()` feeds
torch_ema (pl_model.py:142), `load_from_checkpoint` loads a reference checkpoint strictly (separate.py:44), and the EMA
swap of `.eval()` (pl_model.py:655-660) writes into these tensors.  The parameters are the ONLY copy of the weights on the
Python side; an engine holds its own repacked copy on the device, and every use of an engine first asks whether the
tensors still hold what that copy was packed from (_poll_weights): if not, the engine is rebuilt from the new weights.
"""
import zlib

import numpy as np
import torch
from torch import nn

from . import _lib, synth
from .engine import Engine, param_table

_DTYPES = {"bf16": _lib.BF16, "f16": _lib.F16, "fp16": _lib.F16, "f32": _lib.F32, "fp32": _lib.F32, "split": _lib.F32_SPLIT}

try:  # (64-bit, ~10 GB/s; the image has it — zlib's crc32 otherwise)
    import xxhash

    def _digest(buf):
        return xxhash.xxh3_64_intdigest(buf)
except Exception:  # pragma: no cover
    def _digest(buf):
        return zlib.crc32(buf)


def _loaded(module, incompatible_keys):
    module.weights_changed()


class _Node(nn.Module):
    """One level of the reference's module tree (NCSNpp, its all_modules list, a ResnetBlockBigGANpp, a Conv2d ...): holds
    parameters and child nodes under the reference's names, nothing else."""


class _StftBuffers(nn.Module):
    """`stft.window` / `stft_inv.window`: torchaudio's Spectrogram / InverseSpectrogram keep their Hann window as a
    persistent buffer (models/score_models.py:29-30), so a reference checkpoint carries both keys.  They load into this
    buffer; a state dict WITHOUT them loads too (the window is a constant of the configuration: periodic Hann(n_fft), which
    is what the engine's STFT kernels use)."""

    def __init__(self, n_fft):
        super().__init__()
        self.register_buffer("window", torch.hann_window(n_fft))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        n = len(missing_keys)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        del missing_keys[n:]


class _EngineSlot:
    """One engine (a precision mode in one build of the library) over the weights of a ScoreModelNCSNpp.  engine() returns
    it, (re)built whenever the owner's weights or device are not the ones the current engine was packed from."""

    def __init__(self, owner, cfg, lib_kind):
        self.owner, self.cfg, self.lib_kind = owner, cfg, lib_kind
        self._engine, self._key = None, None
        self.builds = 0
        self.options = {}  # engine options (Engine.set_option) that every engine of this slot gets, rebuilt ones included

    def engine(self):
        own = self.owner
        blob = own._poll_weights()
        key = (own._fingerprint, str(own.engine_device()))
        if self._engine is None or key != self._key:
            self.close()
            self._engine = own._engine_factory(self.cfg, own.packed_blob() if blob is None else blob,
                                               device=own.engine_device(), lib_kind=self.lib_kind)
            self._key = key
            self.builds += 1
            for name, value in self.options.items():
                self._engine.set_option(name, value)
        return self._engine

    def set_option(self, name, value):
        self.options[name] = value
        if self._engine is not None:
            self._engine.set_option(name, value)

    # copy.deepcopy / pickle of the owning module (torch.save(model), Lightning utilities): an engine is a device object of THIS
    # process — the copy gets an empty slot and builds its own engine from its own parameters on first use
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"], st["_key"], st["builds"] = None, None, 0
        return st

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None


class ScoreModelTwin:
    """The same weights in another precision mode (ScoreModelNCSNpp.twin): a score function and an engine of its own, no
    parameters of its own — weights, device, load_state_dict() and to() are the owner's."""

    def __init__(self, owner, dtype, lib_kind=None):
        cfg = _lib.ModelConfig.from_buffer_copy(owner.cfg)
        cfg.dtype = dtype if isinstance(dtype, int) else _DTYPES[dtype]
        self.owner, self.cfg, self.lib_kind = owner, cfg, lib_kind
        self._slot = _EngineSlot(owner, cfg, lib_kind)
        self.num_sources = owner.num_sources

    def engine(self):
        return self._slot.engine()

    def set_engine_option(self, name, value):
        self._slot.set_option(name, value)

    def twin(self, dtype, lib_kind=None):
        return self.owner.twin(dtype, lib_kind)

    def forward(self, xt, time_cond, mix):
        return self.engine().score(xt, time_cond, mix)

    __call__ = forward

    def load_state_dict(self, state, strict=True):
        return self.owner.load_state_dict(state, strict=strict)

    def state_dict(self):
        return self.owner.state_dict()

    def to(self, *args, **kwargs):
        self.owner.to(*args, **kwargs)
        return self

    def eval(self):
        return self


class ScoreModelNCSNpp(nn.Module):
    def __init__(self, num_sources, stft_args, backbone_args, transform="exponent", spec_abs_exponent=0.5,
                 spec_factor=3.0, spec_trans_learnable=False, dtype="f16", device=None, init_seed=0, lib_kind=None):
        super().__init__()
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
            spec_factor=spec_factor, dtype=_DTYPES[dtype])
        self.device = device      # where the engine lives while the parameters are host tensors (None: the current device)
        self.lib_kind = lib_kind  # (Engine: which build of the library; None = by dtype)

        # ---- the reference's parameter tree, random init like the reference constructor: no
        # checkpoint yet, synthetic variance-scaling weights.  requires_grad as in the reference — everything but the frozen
        # Fourier projection (layerspp.py:35-37) — because torch_ema shadows exactly the parameters that require a gradient
        # (a reference checkpoint's `ema` holds one tensor fewer than there are parameters) and refuses a state of another
        # length.  The forward is not differentiable: this is an inference engine.
        self._table = param_table(self.cfg)
        off = 0
        for _, shape, o in self._table:  # (packed_blob concatenates: the blob is the parameters in table order, no gaps)
            assert o == off, "parameter table offsets are not contiguous"
            off += int(np.prod(shape))
        state = synth.synth_state_dict([(n, s) for n, s, _ in self._table], init_seed)
        self.backbone = _Node()
        plist = []
        for name, _, _ in self._table:
            node, parts = self.backbone, name.split(".")
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, _Node())
                node = getattr(node, p)
            par = nn.Parameter(torch.from_numpy(state[name]), requires_grad=not name.endswith("all_modules.0.W"))
            node.register_parameter(parts[-1], par)
            plist.append((name, parts))
        self.stft = _StftBuffers(stft_args["n_fft"])
        self.stft_inv = _StftBuffers(stft_args["n_fft"])
        self._paths = plist

        # ---- engine bookkeeping (plain attributes: none of this is module state)
        self._maybe_changed, self._versions, self._fingerprint, self._plist = True, None, None, None
        self._engine_factory = Engine
        self._slot = _EngineSlot(self, self.cfg, lib_kind)
        self.register_load_state_dict_post_hook(_loaded)  # (fires when a PARENT module's load_state_dict reaches this one, too)

    # ---- weights ---------------------------------------------------------------------------
    def param_names(self):
        return ["backbone." + n for n, _, _ in self._table]

    def _params_in_table_order(self):
        out = []
        for _, parts in self._paths:
            node = self.backbone
            for p in parts:
                node = getattr(node, p)
            out.append(node)
        return out

    def weights_changed(self):
        """Tell the model that its parameters MAY hold new values: the next use of an engine compares their content with
        what that engine was packed from.  Called by everything that can change them behind torch's version counters —
        load_state_dict (post hook: also when a PARENT module loads), train() / eval() (the reference swaps the EMA
        weights in right after, through `param.data.copy_`, pl_model.py:655-666), to() / half() / ... (_apply).  In-place
        writes on the parameters themselves (optimiser steps, `p.copy_`) are seen through `Tensor._version`; call this
        after writing through `.data` by hand."""
        self._maybe_changed, self._plist = True, None

    def packed_blob(self):
        """float32 numpy array: the parameters in the engine's order (diffsep_param_info), as they are NOW"""
        ps = self._plist = self._params_in_table_order()
        for (name, shape, _), p in zip(self._table, ps):
            if tuple(p.shape) != shape:
                raise ValueError(f"parameter 'backbone.{name}' has shape {tuple(p.shape)}, expected {shape}")
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
        return np.ascontiguousarray(flat.cpu().numpy())

    def _poll_weights(self):
        """None when nothing can have changed since the last poll; otherwise re-reads the parameters, updates
        ._fingerprint and returns the packed blob (for the caller that has to rebuild an engine from it)."""
        if self._plist is None:
            self._plist = self._params_in_table_order()
        versions = sum(p._version for p in self._plist)
        if not self._maybe_changed and versions == self._versions and self._fingerprint is not None:
            return None
        w = self.stft.window
        tol = 1e-6 if w.dtype in (torch.float32, torch.float64) else 4e-3  # (model.half() / .bfloat16() round the buffer too)
        if w.shape != (self.cfg.n_fft,) or not torch.allclose(w.float().cpu(), torch.hann_window(self.cfg.n_fft), atol=tol):
            raise NotImplementedError("stft.window is not the periodic Hann window the engine's STFT kernels implement")
        blob = self.packed_blob()
        self._fingerprint = _digest(blob.view(np.uint8))
        self._maybe_changed, self._versions = False, versions
        return blob

    def load_state_dict(self, state, strict=True, assign=False):
        """nn.Module.load_state_dict with the reference's keys ('backbone.all_modules.3.weight', ..., 'stft.window');
        numpy arrays are accepted as values, the two window buffers may be absent.  Strict like torch: a missing /
        unexpected key or a shape mismatch raises RuntimeError."""
        state = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in state.items()}
        return super().load_state_dict(state, strict=strict)

    def train(self, mode=True):
        super().train(mode)
        self.weights_changed()
        return self

    def _apply(self, fn, *args, **kwargs):
        res = super()._apply(fn, *args, **kwargs)
        self.weights_changed()
        return res

    def engine_device(self):
        """The device of the parameters when they live on a GPU (the reference moves the model with .to(device),
        separate.py:47); for host parameters the `device` constructor argument, else the current device."""
        p = self.backbone.output_layer.weight
        if p.is_cuda:
            return p.device
        return self.device

    # ---- engines ---------------------------------------------------------------------------
    def twin(self, dtype, lib_kind=None):
        """The same model in another precision mode: a ScoreModelTwin whose engine is created on first use and FOLLOWS this
        model — it is rebuilt when the weights or the device of this model are no longer the ones it was built from (a twin
        on stale weights would return finite but wrong samples as an overflow fallback)."""
        return ScoreModelTwin(self, dtype, lib_kind)

    def engine(self):
        """The device-resident engine: created lazily, rebuilt when the parameters or their device changed."""
        return self._slot.engine()

    def set_engine_option(self, name, value):
        """Engine.set_option on this model's engine, now and after every rebuild"""
        self._slot.set_option(name, value)

    # ---- reference forward ---------------------------------------------------------------
    def forward(self, xt, time_cond, mix):
        return self.engine().score(xt, time_cond, mix)
