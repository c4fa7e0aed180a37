"""Unit-op wrappers over the C-ABI (parity tests call the kernels through these).
Activations are NHWC torch tensors on the GPU; bf16 is torch.bfloat16 storage."""
import ctypes as C

import numpy as np
import torch

from ._lib import F32_SPLIT, F32, BF16, SdeConfig, SDE_MIX, check, lib
from .engine import _ptr, _stream_ptr


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype in (torch.bfloat16, torch.float16):  # the 16-bit storage format of the library _L(t) picks
        return BF16
    raise TypeError("float32, bfloat16 or float16 expected")


def _L(t):
    """the library build whose 16-bit storage format is t's: half-precision tensors go to libdiffsep_hip_f16.so"""
    return lib("f16" if (t is not None and t.dtype == torch.float16) else "bf16")


def _dts(t, split):
    """dtype code of a matrix-product launch: split=True asks for bf16x3 products on fp32 tensors."""
    if split and t.dtype != torch.float32:
        raise TypeError("split products are a mode of fp32 tensors")
    return F32_SPLIT if split else _dt(t)


def to_nhwc(x, cpad=None):
    """NCHW torch tensor -> contiguous NHWC (optionally zero-padded to cpad channels)."""
    y = x.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad > y.shape[-1]:
        y = torch.nn.functional.pad(y, (0, cpad - y.shape[-1]))
    return y.contiguous()


def to_nchw(y, c=None):
    y = y.permute(0, 3, 1, 2)
    return (y if c is None else y[:, :c]).contiguous()


def pack_conv_weight(w, dtype, chunk=0):
    """OIHW float32 -> [O][taps][Ipad] in `dtype` (Ipad = roundup(I, 8)); chunk = kc > 0: chunk-major
    [Ipad/kc][taps][O][kc] (kc = conv2d_chunk(ksize, dtype); needs Ipad % kc == 0)."""
    O, I, kh, kw = w.shape
    wp = w.permute(0, 2, 3, 1).reshape(O, kh * kw, I)
    ipad = (I + 7) // 8 * 8
    if ipad != I:
        wp = torch.nn.functional.pad(wp, (0, ipad - I))
    if chunk:
        assert ipad % chunk == 0
        wp = wp.reshape(O, kh * kw, ipad // chunk, chunk).permute(2, 1, 0, 3)
    return wp.contiguous().to(dtype)


def pack_frag_weight(w, dtype):
    """OIHW float32 -> the fragment-major order of the register-weight / streamed-weight kernels (include/diffsep_hip.h,
    diffsep_frag_index): element (o, tap, i) at ((((i // 64 * taps + tap) * 4 + i % 64 // 16) * (O // 32) + o // 32) * 64
    + (i % 16 // 8) * 32 + o % 32) * 8 + i % 8.  I % 64 == 0, O % 32 == 0."""
    O, I, kh, kw = w.shape
    taps = kh * kw
    assert I % 64 == 0 and O % 32 == 0
    o, t, i = torch.meshgrid(torch.arange(O), torch.arange(taps), torch.arange(I), indexing="ij")
    idx = ((((i // 64 * taps + t) * 4 + (i % 64) // 16) * (O // 32) + o // 32) * 64 + ((i % 16) // 8) * 32 + o % 32) * 8 + i % 8
    out = torch.empty(O * taps * I, dtype=torch.float32)
    out[idx.reshape(-1)] = w.permute(0, 2, 3, 1).reshape(-1).float()
    return out.to(dtype)


def pack_frag_weight_split(w):
    """OIHW float32 -> the split mode's fragment copy (diffsep_frag_index_split): bfloat16 planes hi = bf16(w), lo = bf16(w - hi);
    element (o, tap, i, plane) at ((((i // 32 * taps + tap) * 2 + i % 32 // 16) * 2 + plane) * (O // 32) + o // 32) * 64
    + (i % 16 // 8) * 32 + o % 32) * 8 + i % 8.  I % 32 == 0, O % 32 == 0.  Returns a bfloat16 tensor of 2 O taps I elements."""
    O, I, kh, kw = w.shape
    taps = kh * kw
    assert I % 32 == 0 and O % 32 == 0
    o, t, i = torch.meshgrid(torch.arange(O), torch.arange(taps), torch.arange(I), indexing="ij")
    base = (((i // 32 * taps + t) * 2 + (i % 32) // 16) * 2) * (O // 32)
    tail = (o // 32) * 64 + ((i % 16) // 8) * 32 + o % 32
    wf = w.permute(0, 2, 3, 1).reshape(-1).float()
    hi = wf.to(torch.bfloat16)
    lo = (wf - hi.float()).to(torch.bfloat16)
    out = torch.empty(2 * O * taps * I, dtype=torch.bfloat16)
    out[((base * 64 + tail) * 8 + i % 8).reshape(-1)] = hi
    out[(((base + O // 32) * 64 + tail) * 8 + i % 8).reshape(-1)] = lo
    return out


def conv3x3_streamed(x, w_frag, cout, x2=None, gn=None, bias=None, bias_b=None, skip=None, out_scale=1.0, stats=False, out=None,
                     res=None, ident_frag=None):
    """The streamed-weight 3x3 kernels as units (diffsep_conv3x3_streamed).  x [B,H,W,C1] (+ x2 [B,H,W,C2]) dense; 16-bit tensors:
    conv3x3_sw.hip with pack_frag_weight copies; float32 tensors: the split mode's conv3x3_sws.hip with pack_frag_weight_split
    copies (res + ident_frag: residual against the identity copy).  gn = (scale, shift) [B,Cin] f32 -> SiLU(GroupNorm(.)) on the
    fly; skip = (sx, sx2 | None, sw_frag): folded 1x1 on raw channels."""
    B, H, W, C1 = x.shape
    Cin = C1 + (x2.shape[-1] if x2 is not None else 0)
    y = torch.zeros((B, H, W, cout), dtype=x.dtype, device=x.device) if out is None else out
    sc, sh = gn if gn is not None else (None, None)
    st = None
    if stats is True:
        st = torch.zeros((B, cout, 2), dtype=torch.int64, device=x.device)
    elif stats is not False:
        st = stats  # (a tensor: the launch adds into it)
    sx, sx2, swf = skip if skip is not None else (None, None, None)
    sC1 = sx.shape[-1] if sx is not None else 0
    sCin = sC1 + (sx2.shape[-1] if sx2 is not None else 0)
    split = x.dtype == torch.float32
    L = lib("bf16") if split else _L(x)
    check(L.diffsep_conv3x3_streamed(_ptr(x), _ptr(x2), C1, _ptr(sc), _ptr(sh), _ptr(w_frag), _ptr(bias), _ptr(bias_b),
                                     _ptr(sx), _ptr(sx2), sC1, sCin, _ptr(swf), _ptr(y), B, H, W, Cin, cout, out_scale,
                                     F32_SPLIT if split else _dt(x), _ptr(st), _ptr(res), _ptr(ident_frag), _stream_ptr()), L)
    return (y, st) if stats is not False else y


def conv2d_chunk(ksize, dtype):
    return lib().diffsep_conv2d_chunk(ksize, F32 if dtype == torch.float32 else BF16)  # (the same in both builds)


def upfirdn2d(x, up):
    B, H, W, Cc = x.shape
    Ho, Wo = (2 * H, 2 * W) if up else (H // 2, W // 2)
    y = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    check(_L(x).diffsep_upfirdn2d(_ptr(x), _ptr(y), B, H, W, Cc, Cc, Cc, int(up), _dt(x), _stream_ptr()), _L(x))
    return y


def groupnorm_act(x, gamma, beta, groups, eps=1e-6, act=1, resample=0, want_xr=False):
    B, H, W, Cc = x.shape
    Ho, Wo = {0: (H, W), 1: (2 * H, 2 * W), 2: (H // 2, W // 2)}[resample]
    y = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    xr = torch.empty_like(y) if want_xr else None
    ws = torch.empty(B * 64 * Cc * 16 + 2 * B * Cc * 4 + 4096, dtype=torch.uint8, device=x.device)
    check(_L(x).diffsep_groupnorm_act(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(xr), B, H, W, Cc, Cc, Cc, Cc,
                                      groups, eps, act, resample, _dt(x), _ptr(ws), ws.numel(), _stream_ptr()), _L(x))
    return (y, xr) if want_xr else y


def conv2d(x, wpacked, bias, cout, ksize, bias_b=None, res=None, out_scale=1.0, cout_pad=None):
    B, H, W, Cin = x.shape
    cp = cout if cout_pad is None else cout_pad
    y = torch.zeros((B, H, W, cp), dtype=x.dtype, device=x.device)
    check(_L(x).diffsep_conv2d(_ptr(x), _ptr(wpacked), _ptr(bias), _ptr(bias_b), _ptr(res), _ptr(y), B, H, W, Cin,
                               cout, ksize, Cin, res.shape[-1] if res is not None else 0, cp, out_scale, _dt(x),
                               _stream_ptr()), _L(x))
    return y


def groupnorm_stats(x, gamma, beta, groups, eps=1e-6, x2=None):
    """scale, shift [B,C] fp32 of GroupNorm over x (or over cat([x, x2], channel) read in place)."""
    B, H, W, C1 = x.shape
    Cc = C1 + (x2.shape[-1] if x2 is not None else 0)
    scale = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    shift = torch.empty_like(scale)
    ws = torch.empty(B * 64 * Cc * 16 + 4096, dtype=torch.uint8, device=x.device)
    check(_L(x).diffsep_groupnorm_stats(_ptr(x), _ptr(x2), C1, _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift), B, H,
                                        W, Cc, C1, x2.shape[-1] if x2 is not None else 0, groups, eps, _dt(x), _ptr(ws),
                                        ws.numel(), _stream_ptr()), _L(x))
    return scale, shift


STAT_SUM_SCALE, STAT_SQ_SCALE = 2.0 ** 24, 2.0 ** 16  # fixed-point scales of the GroupNorm accumulators


def conv2d_fused(x, wpacked, bias, cout, ksize, x2=None, gn=None, gn_act=1, bias_b=None, res=None, out_scale=1.0,
                 cout_pad=None, out=None, stats=False, w_chunk=0, gn_acc=None, split=False):
    """stats=True additionally returns the int64 channel-sum accumulators [B,cout,2] of the output (sum * 2^24,
    sum of squares * 2^16); stats=<tensor> adds into it.  gn_acc=(acc1, acc2|None, gamma, beta, groups): GroupNorm of
    the input from such accumulators instead of gn=(scale, shift)."""
    B, H, W, C1 = x.shape
    Cin = C1 + (x2.shape[-1] if x2 is not None else 0)
    cp = cout if cout_pad is None else cout_pad
    y = torch.zeros((B, H, W, cp), dtype=x.dtype, device=x.device) if out is None else out
    sc, sh = gn if gn is not None else (None, None)
    st = None
    if stats is True:
        st = torch.zeros((B, cout, 2), dtype=torch.int64, device=x.device)
    elif stats is not False:
        st = stats
    a1, a2, gam, bet, grp = gn_acc if gn_acc is not None else (None, None, None, None, 0)
    check(_L(x).diffsep_conv2d_fused(_ptr(x), _ptr(x2), C1, _ptr(sc), _ptr(sh), gn_act, _ptr(wpacked), _ptr(bias),
                                     _ptr(bias_b), _ptr(res), _ptr(y), B, H, W, Cin, cout, ksize, C1,
                                     x2.shape[-1] if x2 is not None else 0, res.shape[-1] if res is not None else 0, cp,
                                     out_scale, _dts(x, split), _ptr(st), w_chunk, _ptr(a1), _ptr(a2), _ptr(gam), _ptr(bet), grp,
                                     _stream_ptr()), _L(x))
    return (y, st) if stats is not False else y


def stats_to_float(st):
    """int64 accumulators [B,C,2] -> float64 (sum, sum of squares)."""
    out = st.double()
    out[..., 0] /= STAT_SUM_SCALE
    out[..., 1] /= STAT_SQ_SCALE
    return out


def attention(q, k, vt, split=False):
    """split=True (fp32 tensors only): both GEMMs with bf16x3 products (DIFFSEP_F32_SPLIT)."""
    B, L, Cc = q.shape
    Lp = (L + 7) // 8 * 8
    assert vt.shape == (B, Cc, Lp)
    o = torch.empty_like(q)
    ws = torch.empty(2 * (B * L * Lp * q.element_size() + 256), dtype=torch.uint8, device=q.device)
    check(_L(q).diffsep_attention(_ptr(q), _ptr(k), _ptr(vt), _ptr(o), B, L, Cc, Cc, _dts(q, split), _ptr(ws), ws.numel(),
                                  _stream_ptr()), _L(q))
    return o


def resblock_forward(params, x, temb, out_ch, up=False, down=False):
    """ResnetBlockBigGANpp through the engine's block code.  params: list of float32 arrays in reference state_dict
    order; x [B,H,W,Cin] NHWC, temb [B,D] f32 -> [B,H',W',out_ch]."""
    B, H, W, cin = x.shape
    blob = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1) for p in params]))
    Ho, Wo = (2 * H, 2 * W) if up else ((H // 2, W // 2) if down else (H, W))
    y = torch.empty((B, Ho, Wo, out_ch), dtype=x.dtype, device=x.device)
    check(_L(x).diffsep_resblock_forward(cin, out_ch, int(up), int(down), temb.shape[1], _dt(x),
                                         blob.ctypes.data_as(C.c_void_p), blob.size, _ptr(x.contiguous()),
                                         _ptr(temb.contiguous()), _ptr(y), B, H, W, _stream_ptr()), _L(x))
    return y


def attnblock_forward(params, x):
    """AttnBlockpp through the engine's block code.  params: GroupNorm_0.{weight,bias}, NIN_0..3.{W,b}."""
    B, H, W, Cc = x.shape
    blob = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1) for p in params]))
    y = torch.empty_like(x)
    check(_L(x).diffsep_attnblock_forward(Cc, _dt(x), blob.ctypes.data_as(C.c_void_p), blob.size, _ptr(x.contiguous()),
                                          _ptr(y), B, H, W, _stream_ptr()), _L(x))
    return y


def stft_pack(xt, mix, W, cpad, n_fft=510, hop=128, exponent=0.5, factor=0.33, shift=False, dtype=torch.float32):
    B, S, T = xt.shape
    y = torch.empty((B, n_fft // 2 + 1, W, cpad), dtype=dtype, device=xt.device)
    F_ = 1 + (T + n_fft - hop) // hop
    ws = torch.empty(2 * ((B * (S + 1) * F_ + 8) * 512 + 64), dtype=torch.float32, device=xt.device)
    check(_L(y).diffsep_stft_pack(_ptr(xt), _ptr(mix), _ptr(y), B, S, T, n_fft, hop, exponent, factor, W, cpad,
                                  int(shift), F32 if dtype == torch.float32 else BF16, _ptr(ws), ws.numel() * 4,
                                  _stream_ptr()), _L(y))
    return y


def istft_unpack(x, S, T, n_fft=510, hop=128, exponent=0.5, factor=0.33):
    B, H, W, cpad = x.shape
    F_ = 1 + (T + n_fft - hop) // hop
    out = torch.empty((B, S, T), dtype=torch.float32, device=x.device)
    ws = torch.empty(2 * (B * S * F_ * 512 + 64), dtype=torch.float32, device=x.device)
    check(_L(x).diffsep_istft_unpack(_ptr(x), _ptr(out), B, S, T, n_fft, hop, exponent, factor, W, cpad, _dt(x),
                                     _ptr(ws), ws.numel() * 4, _stream_ptr()), _L(x))
    return out


def _sde(s):
    return SdeConfig(s.get("kind", SDE_MIX), s["ndim"], s["d_lambda"], s["sigma_min"], s["sigma_max"],
                     s.get("avg_len", 0))


def sde_sigma_mix(mix, avg_len=510):
    """PriorMixSDE per-sample noise scale: mix [B,1,T] -> [B,T]."""
    B, _, T = mix.shape
    out = torch.empty((B, T), dtype=torch.float32, device=mix.device)
    check(lib().diffsep_sde_sigma_mix(_ptr(mix), _ptr(out), B, T, avg_len, _stream_ptr()))
    return out


def sde_prior(sde, y, z, sigma_mix=None):
    B, S, T = z.shape
    x = torch.empty_like(z)
    sc = _sde(sde)
    check(lib().diffsep_sde_prior(C.byref(sc), _ptr(y), _ptr(z), _ptr(x), B, S, T, _ptr(sigma_mix), _stream_ptr()))
    return x


def sde_corrector_update(sde, snr, x, t, score, z, sigma_mix=None, variant=0):
    """variant 0: ald2, 1: ald."""
    B, S, T = x.shape
    xo, xm = torch.empty_like(x), torch.empty_like(x)
    sc = _sde(sde)
    check(lib().diffsep_sde_corrector_update(C.byref(sc), snr, _ptr(x), _ptr(t), _ptr(score), _ptr(z), _ptr(xo),
                                             _ptr(xm), B, S, T, _ptr(sigma_mix), variant, _stream_ptr()))
    return xo, xm


def sde_predictor_update(sde, N, x, t, score, z, sigma_mix=None, probability_flow=False):
    B, S, T = x.shape
    xo, xm = torch.empty_like(x), torch.empty_like(x)
    sc = _sde(sde)
    check(lib().diffsep_sde_predictor_update(C.byref(sc), N, _ptr(x), _ptr(t), _ptr(score), _ptr(z), _ptr(xo),
                                             _ptr(xm), B, S, T, _ptr(sigma_mix), int(bool(probability_flow)),
                                             _stream_ptr()))
    return xo, xm


def sde_coefficients(sde, x, t, sigma_mix=None, f_scale=1.0, g_scale=1.0):
    """(f_scale * drift [B,S,T], g_scale * diffusion): diffusion is [B] for MixSDE, [B,S,T] for PriorMixSDE."""
    B, S, T = x.shape
    drift = torch.empty_like(x)
    diff = torch.empty_like(x) if sigma_mix is not None else torch.empty(B, dtype=torch.float32, device=x.device)
    sc = _sde(sde)
    check(lib().diffsep_sde_coefficients(C.byref(sc), _ptr(x), _ptr(t), _ptr(sigma_mix), _ptr(drift), _ptr(diff), B, S, T,
                                         f_scale, g_scale, _stream_ptr()))
    return drift, diff


def sde_mean(sde, x0, t):
    B, S, T = x0.shape
    out = torch.empty_like(x0)
    sc = _sde(sde)
    check(lib().diffsep_sde_mean(C.byref(sc), _ptr(x0), _ptr(t), _ptr(out), B, S, T, _stream_ptr()))
    return out


def sde_std(sde, t, S, T=1, sigma_mix=None):
    """Dense matrix square root of the perturbation covariance: [B,S,S], or [B,S,S,T] with sigma_mix [B,T]."""
    B = t.shape[0]
    shape = (B, S, S, T) if sigma_mix is not None else (B, S, S)
    out = torch.empty(shape, dtype=torch.float32, device=t.device)
    sc = _sde(sde)
    check(lib().diffsep_sde_std(C.byref(sc), _ptr(t), _ptr(sigma_mix), _ptr(out), B, S, T, _stream_ptr()))
    return out


def sde_mult_std(std, x):
    """std [B,S,S] or [B,S,S,T] applied to x [B,S,T]."""
    B, S, T = x.shape
    per = std.dim() == 4
    assert std.shape == ((B, S, S, T) if per else (B, S, S)), "std must be [B,S,S] or [B,S,S,T]"
    out = torch.empty_like(x)
    check(lib().diffsep_sde_mult_std(_ptr(std.contiguous()), _ptr(x), _ptr(out), B, S, T, int(per), _stream_ptr()))
    return out


def sde_reverse_drift(f, G, score, probability_flow=False):
    """rev_f = f - G^2 score (x 0.5 for the probability-flow ODE); G [B] or the shape of f."""
    B = f.shape[0]
    n = f.numel() // B
    full = G.numel() != B
    assert (not full) or G.shape == f.shape
    out = torch.empty_like(f)
    check(lib().diffsep_sde_reverse_drift(_ptr(f), _ptr(G.contiguous()), _ptr(score), _ptr(out), B, n, int(full),
                                          int(bool(probability_flow)), _stream_ptr()))
    return out


def sde_langevin_update(snr, x, score, z):
    B = x.shape[0]
    n = x.numel() // B
    xo, xm = torch.empty_like(x), torch.empty_like(x)
    ws = torch.empty(16 * B + 64, dtype=torch.uint8, device=x.device)
    check(lib().diffsep_sde_langevin_update(snr, _ptr(x), _ptr(score), _ptr(z), _ptr(xo), _ptr(xm), B, n, _ptr(ws),
                                            ws.numel(), _stream_ptr()))
    return xo, xm


def normalize_batch(mix):
    B, _, T = mix.shape
    out = torch.empty_like(mix)
    mean = torch.empty(B, dtype=torch.float32, device=mix.device)
    std = torch.empty(B, dtype=torch.float32, device=mix.device)
    check(lib().diffsep_normalize_batch(_ptr(mix), _ptr(out), _ptr(mean), _ptr(std), B, T, _stream_ptr()))
    return out, mean.view(B, 1, 1), std.view(B, 1, 1)


def scale_output(mix, sep):
    B, S, T = sep.shape
    out = sep.clone()
    check(lib().diffsep_scale_output(_ptr(mix), _ptr(out), B, S, T, _stream_ptr()))
    return out


def time_embedding(t, fourier_w, w1, b1, w2, b2):
    """NCSNpp's time embedding (ncsnpp.py:324-343): t [B] -> temb [B, 4 nf]; all float32 device tensors."""
    B, nf = t.shape[0], fourier_w.shape[0]
    ts = [x.contiguous().float() for x in (t, fourier_w, w1, b1, w2, b2)]
    out = torch.empty((B, 4 * nf), dtype=torch.float32, device=t.device)
    ws = torch.empty(B * 6 * nf, dtype=torch.float32, device=t.device)
    check(lib().diffsep_time_embedding(*[_ptr(x) for x in ts], _ptr(out), B, nf, _ptr(ws), ws.numel() * 4, _stream_ptr()))
    return out


def randn(n, seed, stream_id, device="cuda"):
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(lib().diffsep_randn(_ptr(out), n, seed, stream_id, _stream_ptr()))
    return out


def randn_batch(B, S, T, seeds, lengths, stream_id, device="cuda"):
    """[B,S,T] draws: row (b, s) = values s*len_b .. of randn(S*len_b, seeds[b], stream_id), zero beyond len_b."""
    out = torch.empty((B, S, T), dtype=torch.float32, device=device)
    sd = torch.as_tensor(np.asarray(seeds, dtype=np.uint64).view(np.int64), device=device)
    ln = torch.as_tensor(np.asarray(lengths, dtype=np.int32), device=device)
    check(lib().diffsep_randn_batch(_ptr(out), B, S, T, _ptr(sd), _ptr(ln), stream_id, _stream_ptr()))
    return out


def gram(ref, est):
    """[B,S,T] x2 -> float64 [B,3,S,S] = (ref ref^T, ref est^T, est est^T)."""
    B, S, T = ref.shape
    out = torch.empty((B, 3, S, S), dtype=torch.float64, device=ref.device)
    check(lib().diffsep_gram(_ptr(ref.contiguous()), _ptr(est.contiguous()), _ptr(out), B, S, T, _stream_ptr()))
    return out
