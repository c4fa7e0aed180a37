"""Engine: the Python handle on one device-resident NCSN++ score model + PC sampler.

PyTorch is used for device memory (tensor.data_ptr()), streams and nothing else: every FLOP of
the path runs in libdiffsep_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import F32, BF16, F16, check, lib


def _stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _c_cfg(cfg):
    """the C-side view of a model config: the Python-only dtype code F16 is code 1 (16-bit storage) of the f16 build"""
    if cfg.dtype != F16:
        return cfg
    c2 = _lib.ModelConfig()
    C.memmove(C.byref(c2), C.byref(cfg), C.sizeof(cfg))
    c2.dtype = BF16
    return c2


def param_table(cfg):
    """[(name, shape, offset)] in the canonical (reference state_dict) order."""
    l = lib()
    cfg = _c_cfg(cfg)
    n = l.diffsep_param_count(C.byref(cfg))
    if n < 0:
        check(1)
    out = []
    name = C.create_string_buffer(256)
    shape = (C.c_int64 * 4)()
    ndim = C.c_int32()
    off = C.c_int64()
    for i in range(n):
        check(l.diffsep_param_info(C.byref(cfg), i, name, 256, shape, C.byref(ndim), C.byref(off)))
        out.append((name.value.decode(), tuple(int(shape[k]) for k in range(ndim.value)), int(off.value)))
    return out


def pack_state_dict(cfg, state, prefix=""):
    """Flatten {name: array/tensor} into the float32 blob the engine expects. Missing keys raise."""
    total = lib().diffsep_param_total(C.byref(_c_cfg(cfg)))
    blob = np.empty(total, dtype=np.float32)
    for name, shape, off in param_table(cfg):
        key = prefix + name
        if key not in state:
            raise KeyError(f"checkpoint is missing parameter '{key}'")
        v = state[key]
        v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        if tuple(v.shape) != shape:
            raise ValueError(f"parameter '{key}' has shape {tuple(v.shape)}, expected {shape}")
        blob[off:off + v.size] = v.astype(np.float32, copy=False).reshape(-1)
    return blob


class Engine:
    """Owns the repacked weights and workspace on the current CUDA(HIP) device."""

    def __init__(self, cfg, weights_blob, device=None, lib_kind=None):
        """lib_kind: which build of the library creates the engine ("bf16" / "f16"); default = by dtype (F16 lives in the
        half-precision build, everything else in the default one).  The fp32 engines are the same in both builds: pass
        lib_kind="f16" for a split / fp32 engine that serves as the head engine of an F16 one (pc_sample(tail=...))."""
        if not torch.cuda.is_available():
            raise _lib.DiffsepError("no GPU visible: the separation engine has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # dtype F16 = the 16-bit engine (code 1) of the half-precision build of the library
        self.kind = _lib.half_kind(cfg.dtype) if lib_kind is None else lib_kind
        if cfg.dtype in (F16, BF16) and self.kind != _lib.half_kind(cfg.dtype):
            raise _lib.DiffsepError("a 16-bit engine lives in the library build of its storage format")
        self._L = lib(self.kind)
        cfg = _c_cfg(cfg)
        self.cfg = cfg
        blob = np.ascontiguousarray(weights_blob, dtype=np.float32)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self._L.diffsep_engine_create(C.byref(cfg), blob.ctypes.data_as(C.c_void_p), blob.size,
                                                C.byref(self._h)), self._L)
        self.S = cfg.num_sources

    def close(self):
        if getattr(self, "_h", None):
            self._L.diffsep_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dtype(self):
        return self.cfg.dtype

    def device_bytes(self):
        return int(self._L.diffsep_engine_device_bytes(self._h))

    def set_graph(self, enable):
        check(self._L.diffsep_engine_set_graph(self._h, int(bool(enable))), self._L)

    def set_option(self, name, value):
        """diffsep_engine_set_option: "no_rw" / "no_rw128" / "rw_small" / "no_rw_res" (kernel dispatch A/B), "graph_cache"
        (captured graphs kept, LRU), "ablate" (measurement aid).  Synchronises the device, drops the captured graphs."""
        check(self._L.diffsep_engine_set_option(self._h, str(name).encode(), int(value)), self._L)

    def get_option(self, name):
        return int(self._L.diffsep_engine_get_option(self._h, str(name).encode()))

    CONV_CLASSES = ("conv3x3_8x32xN64", "conv3x3_8x32xN32", "conv3x3_8x8xN64", "gemm1x1_256xN64", "gemm1x1_256xN32",
                    "gemm1x1_64xN64", "conv3x3_ws_64to64", "conv3x3_small_16couts", "conv3x3_rw_regweights", "attention_fused",
                    "conv3x3_sw_streamed", "conv3x3_sws_split")

    def profile_begin(self):
        check(self._L.diffsep_engine_profile_begin(self._h), self._L)

    def profile_end(self):
        """{class: (algorithmic flops, milliseconds, launches, algorithmic bytes)} of the MFMA kernels since
        profile_begin."""
        nc = len(self.CONV_CLASSES)
        fl, ms, n, by = (C.c_double * nc)(), (C.c_double * nc)(), (C.c_int64 * nc)(), (C.c_double * nc)()
        nw = C.c_int32(0)
        check(self._L.diffsep_engine_profile_end_n(self._h, nc, fl, ms, n, by, C.byref(nw)), self._L)
        return {k: (fl[i], ms[i], int(n[i]), by[i]) for i, k in enumerate(self.CONV_CLASSES[:nw.value])}

    def profile_records(self):
        """The launches of the last profile_begin .. profile_end span, one dict per launch: kernel instantiation (real
        template arguments), shape, algorithmic flops / bytes, milliseconds."""
        from ._lib import ProfRecord
        n = C.c_int32(0)
        check(self._L.diffsep_engine_profile_records(self._h, None, 0, C.byref(n)), self._L)
        buf = (ProfRecord * max(1, n.value))()
        check(self._L.diffsep_engine_profile_records(self._h, buf, n.value, C.byref(n)), self._L)
        return [dict(kernel=r.kernel.decode(), B=r.B, H=r.H, W=r.W, Cin=r.Cin, Cout=r.Cout, taps=r.taps,
                     skip_cin=r.skip_cin, has_res=bool(r.has_res), cls=r.cls, flops=r.flops, bytes=r.bytes, ms=r.ms)
                for r in buf[:n.value]]

    def padded_frames(self, T):
        return int(self._L.diffsep_padded_frames(C.byref(self.cfg), T))

    def bucket_length(self, W):
        """The longest signal whose padded frame count is W = 64 k: F = 1 + (T + n_fft - hop) // hop <= W.  Mixed-length
        batches padded to this length share ONE workspace plan / captured graph per (B, W), whatever their members'
        lengths (the results do not depend on the padding: diffsep_sampler_ext.lengths_host)."""
        return int(self.cfg.hop) * int(W) - (int(self.cfg.n_fft) - int(self.cfg.hop)) - 1

    def _f32(self, t):
        assert t.is_cuda and t.dtype == torch.float32, "device float32 tensor expected"
        return t.contiguous()

    def score(self, xt, t, mix):
        """score_fn(x, t, mix) -> [B,S,T]   (DiffSepModel.forward / ScoreModelNCSNpp.forward)."""
        xt, t, mix = self._f32(xt), self._f32(t), self._f32(mix)
        B, S, T = xt.shape
        assert S == self.S and mix.shape == (B, 1, T) and t.shape == (B,)
        out = torch.empty_like(xt)
        with torch.cuda.device(self.device):
            check(self._L.diffsep_score_forward(self._h, _ptr(xt), _ptr(t), _ptr(mix), _ptr(out), B, T,
                                              _stream_ptr(self.device)), self._L)
        return out

    def reserve(self, B, T):
        """Size the workspace for batches of up to B x T samples now (later, smaller plans never reallocate)."""
        with torch.cuda.device(self.device):
            check(self._L.diffsep_engine_reserve(self._h, int(B), int(T), _stream_ptr(self.device)), self._L)

    def debug_absmax(self):
        """[(max finite |value|, non-finite count, rows H, channels C)] of every activation tensor of the last eager forward
        (set_option("track_tensors", 1) first; score() is an eager forward).  Debug / test aid."""
        n = C.c_int32(0)
        check(self._L.diffsep_engine_debug_absmax(self._h, None, 0, C.byref(n)), self._L)
        buf = (C.c_double * (4 * max(1, n.value)))()
        check(self._L.diffsep_engine_debug_absmax(self._h, buf, n.value, C.byref(n)), self._L)
        return [(buf[4 * i], int(buf[4 * i + 1]), int(buf[4 * i + 2]), int(buf[4 * i + 3])) for i in range(n.value)]

    def debug_arena(self):
        """(uint8 view of the workspace arena, offset of the forward region): every intermediate tensor of the last
        forward, in launch order.  Debug / test aid."""
        base, nb, fb = C.c_void_p(), C.c_int64(), C.c_int64()
        check(self._L.diffsep_engine_debug_arena(self._h, C.byref(base), C.byref(nb), C.byref(fb)), self._L)

        class _Raw:
            __cuda_array_interface__ = {"shape": (nb.value,), "typestr": "|u1", "data": (base.value, False), "version": 2}
        return torch.as_tensor(_Raw(), device=f"cuda:{self.device.index or 0}" if hasattr(self.device, "index") else "cuda"), fb.value

    def backbone(self, x_nhwc, t):
        """NCSNpp.forward on a packed NHWC input [B,256,W,Cpad] (engine dtype storage)."""
        B, H, W, Cp = x_nhwc.shape
        t = self._f32(t)
        cout = ((2 * self.S + 7) // 8) * 8
        y = torch.zeros((B, H, W, cout), dtype=x_nhwc.dtype, device=x_nhwc.device)
        with torch.cuda.device(self.device):
            check(self._L.diffsep_backbone_forward(self._h, _ptr(x_nhwc.contiguous()), _ptr(t), _ptr(y), B, W,
                                                 _stream_ptr(self.device)), self._L)
        return y

    def pc_sample(self, mix_norm, sde, N=30, corrector_steps=1, snr=0.5, eps=0.03, denoise=True,
                  predictor="reverse_diffusion", corrector="ald2", noise=None, seed=0, timesteps=None, lengths=None,
                  seeds=None, tail=None, tail_steps=0, head_steps=0):
        """The whole sampler (sdes.get_pc_sampler(...)()).  sde: dict(kind, ndim, d_lambda, sigma_min, sigma_max).
        Extensions (diffsep_sampler_ext): lengths [B] = a zero-padded batch of utterances of different lengths that share
        one padded frame count; seeds [B] = per-utterance device-RNG seeds; tail / head_steps / tail_steps = evaluate the
        score of the FIRST head_steps and / or the last tail_steps reverse steps on another Engine (dtype "hybrid" of
        pl_model: a split-fp32 engine for the first HYBRID_HEAD_STEPS steps of a 16-bit one — the early steps carry
        the rounding error)."""
        mix_norm = self._f32(mix_norm)
        B, one, T = mix_norm.shape
        assert one == 1
        sc = _lib.SdeConfig(sde.get("kind", _lib.SDE_MIX), sde["ndim"], sde["d_lambda"], sde["sigma_min"],
                            sde["sigma_max"], sde.get("avg_len", 0))
        pred = {"reverse_diffusion": _lib.PRED_REVERSE_DIFFUSION, "euler_maruyama": _lib.PRED_EULER_MARUYAMA,
                "none": _lib.PRED_NONE}[predictor]
        corr = {"ald2": _lib.CORR_ALD2, "none": _lib.CORR_NONE, "ald": _lib.CORR_ALD,
                "langevin": _lib.CORR_LANGEVIN}[corrector]
        sm = _lib.SamplerConfig(N, corrector_steps, snr, eps, int(bool(denoise)), pred, corr)
        out = torch.empty((B, self.S, T), dtype=torch.float32, device=mix_norm.device)
        if noise is not None:
            noise = self._f32(noise)
            ncs = corrector_steps if corrector != "none" else 0
            npred = 1 if predictor != "none" else 0
            assert noise.shape == (1 + N * (ncs + npred), B, self.S, T), "noise must be [draws,B,S,T]"
        ts = None
        if timesteps is not None:
            ts = np.ascontiguousarray(timesteps, dtype=np.float32)
            assert ts.size >= N
        nfe = C.c_int32()
        ext, keep = None, []
        if lengths is not None or seeds is not None or (tail is not None and (tail_steps > 0 or head_steps > 0)):
            ext = _lib.SamplerExt()
            if lengths is not None:
                la = np.ascontiguousarray(lengths, dtype=np.int64)
                assert la.shape == (B,), "lengths must be [B]"
                ext.lengths_host = la.ctypes.data_as(C.POINTER(C.c_int64))
                keep.append(la)
            if seeds is not None:
                sa = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
                assert sa.shape == (B,), "seeds must be [B]"
                ext.seeds_host = sa.ctypes.data_as(C.POINTER(C.c_uint64))
                keep.append(sa)
            if tail is not None and (tail_steps > 0 or head_steps > 0):
                if tail.kind != self.kind:  # an engine handle means something to the library that created it only
                    raise _lib.DiffsepError(f"the other engine was created by the '{tail.kind}' build of the library, this one "
                                            f"by the '{self.kind}' build: create it with Engine(..., lib_kind='{self.kind}')")
                ext.tail_engine, ext.tail_steps, ext.head_steps = tail._h, int(tail_steps), int(head_steps)
        with torch.cuda.device(self.device):
            check(self._L.diffsep_pc_sample_ex(self._h, C.byref(sc), C.byref(sm), C.byref(ext) if ext is not None else None,
                                             _ptr(mix_norm), _ptr(out), B, T, _ptr(noise), seed,
                                             ts.ctypes.data_as(C.c_void_p) if ts is not None else None, C.byref(nfe),
                                             _stream_ptr(self.device)), self._L)
        return out, int(nfe.value)
