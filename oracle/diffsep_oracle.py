"""CPU ORACLE — test infrastructure only.

A torch-CPU float32 (optionally float64) restatement of the reference's reverse-diffusion hot path
(fakufaku/diffusion-separation), written from the algorithm, not copied: every function cites the
reference file:line it follows (paths relative to the reference repository root).

ONLY tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module, and
only as the checker / the timed CPU baseline.  The product path (diffsep_amd + libdiffsep_hip.so)
never imports it and fails loudly when the HIP library is missing.

Parity status: PINNED — except si_bss_eval_sources (third-party fast_bss_eval, source absent: unpinned, see there).  The reference has no tests or golden vectors of its own, so
this oracle is pinned against outputs of the reference itself, imported in the build container
from /root/reference with stubs for its missing third-party packages
(tests/golden/gen_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).

Weights are a plain dict {reference state_dict key (without the "score_model.backbone." prefix):
torch tensor in the reference layout}.  Activations are NCHW like the reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- config
def default_config(nf=64, num_sources=2, spec_factor=0.33, spec_abs_exponent=0.5):
    """config/model/default.yaml:14-37 + models/ncsnpp.py:45-70 constructor defaults."""
    return dict(
        nf=nf, num_sources=num_sources, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolution=16,
        n_fft=510, hop=128, spec_abs_exponent=spec_abs_exponent, spec_factor=spec_factor,
        d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, N=30, snr=0.5, corrector_steps=1, t_eps=0.03,
    )


def param_table(cfg):
    """(name, shape) in reference state_dict order: output_layer first (registered at
    models/ncsnpp.py:104-105, before all_modules at :308), then all_modules.{i}.* following the
    constructor walk models/ncsnpp.py:106-306."""
    nf, S = cfg["nf"], cfg["num_sources"]
    ch = 2 * S + 2
    out = [("output_layer.weight", (2 * S, ch, 1, 1)), ("output_layer.bias", (2 * S,))]
    idx = [0]

    def pfx():
        return f"all_modules.{idx[0]}."

    def nxt():
        idx[0] += 1

    def res(i, o, resample=False):
        p = pfx()
        out.extend([(p + "GroupNorm_0.weight", (i,)), (p + "GroupNorm_0.bias", (i,)),
                    (p + "Conv_0.weight", (o, i, 3, 3)), (p + "Conv_0.bias", (o,)),
                    (p + "Dense_0.weight", (o, 4 * nf)), (p + "Dense_0.bias", (o,)),
                    (p + "GroupNorm_1.weight", (o,)), (p + "GroupNorm_1.bias", (o,)),
                    (p + "Conv_1.weight", (o, o, 3, 3)), (p + "Conv_1.bias", (o,))])
        if i != o or resample:
            out.extend([(p + "Conv_2.weight", (o, i, 1, 1)), (p + "Conv_2.bias", (o,))])
        nxt()

    def attn(c):
        p = pfx()
        out.extend([(p + "GroupNorm_0.weight", (c,)), (p + "GroupNorm_0.bias", (c,))])
        for k in range(4):
            out.extend([(p + f"NIN_{k}.W", (c, c)), (p + f"NIN_{k}.b", (c,))])
        nxt()

    out.append((pfx() + "W", (nf,))); nxt()
    out.extend([(pfx() + "weight", (4 * nf, 2 * nf)), (pfx() + "bias", (4 * nf,))]); nxt()
    out.extend([(pfx() + "weight", (4 * nf, 4 * nf)), (pfx() + "bias", (4 * nf,))]); nxt()
    out.extend([(pfx() + "weight", (nf, ch, 3, 3)), (pfx() + "bias", (nf,))]); nxt()
    image = cfg["n_fft"] // 2 + 1
    L = len(cfg["ch_mult"])
    hs_c, in_ch = [nf], nf
    for i in range(L):
        for _ in range(cfg["num_res_blocks"]):
            o = nf * cfg["ch_mult"][i]
            res(in_ch, o); in_ch = o
            if image >> i == cfg["attn_resolution"]:
                attn(in_ch)
            hs_c.append(in_ch)
        if i != L - 1:
            res(in_ch, in_ch, True)
            p = pfx(); out.extend([(p + "Conv_0.weight", (in_ch, ch, 1, 1)), (p + "Conv_0.bias", (in_ch,))]); nxt()
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    res(in_ch, in_ch); attn(in_ch); res(in_ch, in_ch)
    for i in reversed(range(L)):
        for _ in range(cfg["num_res_blocks"] + 1):
            o = nf * cfg["ch_mult"][i]
            res(in_ch + hs_c.pop(), o); in_ch = o
        if image >> i == cfg["attn_resolution"]:
            attn(in_ch)
        p = pfx(); out.extend([(p + "weight", (in_ch,)), (p + "bias", (in_ch,))]); nxt()
        p = pfx(); out.extend([(p + "weight", (ch, in_ch, 3, 3)), (p + "bias", (ch,))]); nxt()
        if i != 0:
            res(in_ch, in_ch, True)
    assert not hs_c
    return out


# ----------------------------------------------------------------------------- FIR resampling
def fir_down2(x):
    """downsample_2d(x, [1,3,3,1], 2)  up_or_down_sampling.py:242-273 -> upfirdn2d(pad=(1,1), down=2)
    op/upfirdn2d.py:159-200.  Closed form per axis: y[m] = (x[2m-1] + 3x[2m] + 3x[2m+1] + x[2m+2])/8."""
    k = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=x.dtype) / 8.0
    xp = F.pad(x, (1, 1, 1, 1))
    H2, W2 = x.shape[-2] // 2, x.shape[-1] // 2
    y = 0
    for i in range(4):
        for j in range(4):
            y = y + (k[i] * k[j]) * xp[..., i:i + 2 * H2:2, j:j + 2 * W2:2]
    return y


def fir_up2(x):
    """upsample_2d(x, [1,3,3,1], 2)  up_or_down_sampling.py:206-239 -> upfirdn2d(up=2, pad=(2,1)).
    Closed form per axis: y[2m] = x[m-1]/4 + 3x[m]/4, y[2m+1] = 3x[m]/4 + x[m+1]/4 (zeros outside)."""
    def up_axis(v, dim):
        n = v.shape[dim]
        pad = [0, 0] * v.dim()
        pad[2 * (v.dim() - 1 - dim)] = 1
        pad[2 * (v.dim() - 1 - dim) + 1] = 1
        vp = F.pad(v, pad)
        prev, cur, nex = vp.narrow(dim, 0, n), vp.narrow(dim, 1, n), vp.narrow(dim, 2, n)
        even = 0.25 * prev + 0.75 * cur
        odd = 0.75 * cur + 0.25 * nex
        st = torch.stack((even, odd), dim=dim + 1)
        shp = list(v.shape)
        shp[dim] = 2 * n
        return st.reshape(shp)
    return up_axis(up_axis(x, x.dim() - 2), x.dim() - 1)


# ----------------------------------------------------------------------------- NCSN++ pieces
def _gn(x, w, b):
    """nn.GroupNorm(min(C//4, 32), C, eps=1e-6)  layerspp.py:264-266."""
    C = x.shape[1]
    return F.group_norm(x, min(C // 4, 32), w, b, eps=1e-6)


def _res_block(p, pre, x, temb, up=False, down=False):
    """ResnetBlockBigGANpp.forward  layerspp.py:291-323."""
    h = F.silu(_gn(x, p[pre + "GroupNorm_0.weight"], p[pre + "GroupNorm_0.bias"]))
    if up:
        h, x = fir_up2(h), fir_up2(x)
    elif down:
        h, x = fir_down2(h), fir_down2(x)
    h = F.conv2d(h, p[pre + "Conv_0.weight"], p[pre + "Conv_0.bias"], padding=1)
    h = h + F.linear(F.silu(temb), p[pre + "Dense_0.weight"], p[pre + "Dense_0.bias"])[:, :, None, None]
    h = F.silu(_gn(h, p[pre + "GroupNorm_1.weight"], p[pre + "GroupNorm_1.bias"]))
    h = F.conv2d(h, p[pre + "Conv_1.weight"], p[pre + "Conv_1.bias"], padding=1)
    if (pre + "Conv_2.weight") in p:
        x = F.conv2d(x, p[pre + "Conv_2.weight"], p[pre + "Conv_2.bias"])
    return (x + h) / math.sqrt(2.0)


def _nin(x, W, b):
    """NIN.forward  layers.py:678-689: per-pixel x @ W + b with W [in, out]."""
    return torch.einsum("bchw,cd->bdhw", x, W) + b[None, :, None, None]


def _attn_block(p, pre, x):
    """AttnBlockpp.forward  layerspp.py:76-92 (single head over all H*W positions)."""
    B, C, H, W = x.shape
    h = _gn(x, p[pre + "GroupNorm_0.weight"], p[pre + "GroupNorm_0.bias"])
    q = _nin(h, p[pre + "NIN_0.W"], p[pre + "NIN_0.b"]).reshape(B, C, H * W)
    k = _nin(h, p[pre + "NIN_1.W"], p[pre + "NIN_1.b"]).reshape(B, C, H * W)
    v = _nin(h, p[pre + "NIN_2.W"], p[pre + "NIN_2.b"]).reshape(B, C, H * W)
    w = torch.softmax(torch.einsum("bci,bcj->bij", q, k) * (int(C) ** (-0.5)), dim=-1)
    o = torch.einsum("bij,bcj->bci", w, v).reshape(B, C, H, W)
    o = _nin(o, p[pre + "NIN_3.W"], p[pre + "NIN_3.b"])
    return (x + o) / math.sqrt(2.0)


def time_embedding(p, t):
    """ncsnpp.py:324-343 + layerspp.py:37-47: GaussianFourierProjection(log t) -> Linear -> SiLU -> Linear; [B] -> [B, 4 nf]."""
    xp = torch.log(t)[:, None] * p["all_modules.0.W"][None, :] * 2 * np.pi
    temb = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
    temb = F.linear(temb, p["all_modules.1.weight"], p["all_modules.1.bias"])
    return F.linear(F.silu(temb), p["all_modules.2.weight"], p["all_modules.2.bias"])


def ncsnpp_forward(p, cfg, x, t):
    """NCSNpp.forward  ncsnpp.py:319-478 for the default (biggan / fir / output_skip / input_skip /
    sum / fourier / scale_by_sigma / not centered) configuration.  x [B,2S+2,H,W], t [B]."""
    nf, L, nrb = cfg["nf"], len(cfg["ch_mult"]), cfg["num_res_blocks"]
    mi = [0]

    def pre():
        s = f"all_modules.{mi[0]}."
        mi[0] += 1
        return s

    # ncsnpp.py:324-343
    xp = torch.log(t)[:, None] * p[pre() + "W"][None, :] * 2 * np.pi
    temb = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
    s = pre(); temb = F.linear(temb, p[s + "weight"], p[s + "bias"])
    s = pre(); temb = F.linear(F.silu(temb), p[s + "weight"], p[s + "bias"])
    x = 2 * x - 1.0  # ncsnpp.py:347-349 (centered=False)
    pyr_in = x
    s = pre(); hs = [F.conv2d(x, p[s + "weight"], p[s + "bias"], padding=1)]
    for i in range(L):
        for _ in range(nrb):
            h = _res_block(p, pre(), hs[-1], temb)
            if h.shape[-2] == cfg["attn_resolution"]:
                h = _attn_block(p, pre(), h)
            hs.append(h)
        if i != L - 1:
            h = _res_block(p, pre(), hs[-1], temb, down=True)
            pyr_in = fir_down2(pyr_in)
            s = pre(); h = F.conv2d(pyr_in, p[s + "Conv_0.weight"], p[s + "Conv_0.bias"]) + h  # Combine 'sum'
            hs.append(h)
    h = hs[-1]
    h = _res_block(p, pre(), h, temb)
    h = _attn_block(p, pre(), h)
    h = _res_block(p, pre(), h, temb)
    pyramid = None
    for i in reversed(range(L)):
        for _ in range(nrb + 1):
            h = _res_block(p, pre(), torch.cat([h, hs.pop()], dim=1), temb)
        if h.shape[-2] == cfg["attn_resolution"]:
            h = _attn_block(p, pre(), h)
        s = pre(); ph = F.silu(_gn(h, p[s + "weight"], p[s + "bias"]))
        s = pre(); ph = F.conv2d(ph, p[s + "weight"], p[s + "bias"], padding=1)
        pyramid = ph if pyramid is None else fir_up2(pyramid) + ph
        if i != 0:
            h = _res_block(p, pre(), h, temb, up=True)
    assert not hs
    h = pyramid / t[:, None, None, None]  # scale_by_sigma, ncsnpp.py:472-474
    return F.conv2d(h, p["output_layer.weight"], p["output_layer.bias"])


# ----------------------------------------------------------------------------- STFT front end
def num_frames(cfg, T):
    return 1 + (T + cfg["n_fft"] - cfg["hop"]) // cfg["hop"]


def pre_process(cfg, x):
    """ScoreModelNCSNpp.pre_process  score_models.py:107-116 (+ :41-48, :72-76, :83-91).
    torchaudio.transforms.Spectrogram(power=None, n_fft, hop_length, center=True, pad_mode='constant')
    is torch.stft(..., window=hann_window(n_fft), win_length=n_fft, normalized=False, onesided=True)."""
    n_fft, hop = cfg["n_fft"], cfg["hop"]
    T = x.shape[-1]
    xp = F.pad(x, (0, n_fft - hop))
    B, C, Tp = xp.shape
    win = torch.hann_window(n_fft, dtype=x.dtype)
    X = torch.stft(xp.reshape(B * C, Tp), n_fft, hop_length=hop, win_length=n_fft, window=win, center=True,
                   pad_mode="constant", normalized=False, onesided=True, return_complex=True)
    X = X.reshape(B, C, X.shape[-2], X.shape[-1])
    e = abs(cfg["spec_abs_exponent"])
    if e != 1:
        X = X.abs() ** e * torch.exp(1j * X.angle())
    X = X * cfg["spec_factor"]
    Y = torch.cat([X.real, X.imag], dim=1)  # [re c0..cC-1, im c0..cC-1]
    rem = Y.shape[-1] % 64
    n_pad = 0 if rem == 0 else 64 - rem
    if n_pad:
        Y = F.pad(Y, (0, n_pad))
    return Y, T, n_pad


def post_process(cfg, y, T, n_pad):
    """ScoreModelNCSNpp.post_process  score_models.py:118-124 (+ :59-64, :78-81, :99-105)."""
    n_fft, hop = cfg["n_fft"], cfg["hop"]
    if n_pad:
        y = y[..., :-n_pad]
    S = y.shape[1] // 2
    Z = torch.complex(y[:, :S].contiguous(), y[:, S:].contiguous())
    Z = Z / abs(cfg["spec_factor"])
    e = abs(cfg["spec_abs_exponent"])
    if e != 1:
        Z = Z.abs() ** (1 / e) * torch.exp(1j * Z.angle())
    B = Z.shape[0]
    win = torch.hann_window(n_fft, dtype=y.dtype)
    out = torch.istft(Z.reshape(B * S, Z.shape[-2], Z.shape[-1]), n_fft, hop_length=hop, win_length=n_fft, window=win,
                      center=True, normalized=False, onesided=True, length=None)
    out = out.reshape(B, S, -1)
    if out.shape[-1] < T:
        out = F.pad(out, (0, T - out.shape[-1]))
    return out[..., :T]


def score_forward(p, cfg, xt, t, mix):
    """ScoreModelNCSNpp.forward  score_models.py:126-138 (== DiffSepModel.forward pl_model.py:407-409)."""
    x, T, n_pad = pre_process(cfg, torch.cat((xt, mix), dim=1))
    y = ncsnpp_forward(p, cfg, x, t)
    return post_process(cfg, y, T, n_pad)


# ----------------------------------------------------------------------------- SDE / sampler
def mix_mats(S, dtype=torch.float32):
    """MixSDE.get_mix_mat  sdes/sdes.py:242-248."""
    A = torch.ones((S, S), dtype=dtype) / S
    return A[None], (torch.eye(S, dtype=dtype) - A)[None]


def cov_eigval(cfg, t):
    """MixSDE._cov_eigval  sdes/sdes.py:296-309."""
    r = cfg["sigma_max"] / cfg["sigma_min"]
    mult = cfg["sigma_min"] ** 2
    srp = r ** (2 * t)
    ev1 = mult * (srp - 1)
    ev2 = mult * (srp - torch.exp(-2.0 * cfg["d_lambda"] * t)) / (1.0 + cfg["d_lambda"] / math.log(r))
    return ev1, ev2


def mix_std(cfg, t, S):
    """MixSDE._std  sdes/sdes.py:315-320: L = sqrt(ev1) A + sqrt(ev2) P, [B,S,S]."""
    A, P = mix_mats(S, t.dtype)
    ev1, ev2 = cov_eigval(cfg, t)
    return ev1[:, None, None].sqrt() * A + ev2[:, None, None].sqrt() * P


def sigma_mix(mix, avg_len=510):
    """PriorMixSDE._std_sigma_mix  sdes/sdes.py:477-489: [B,1,T] -> [B,1,T]."""
    sm = F.avg_pool1d(mix ** 2, kernel_size=avg_len, stride=1, padding=avg_len // 2)
    sm = sm.clamp(min=1e-4).sqrt()
    if avg_len % 2 == 0:
        sm = sm[..., :-1]
    return 0.5 * sm


def _apply_std(L, v, smix):
    """MixSDE.mult_std (L @ v)  sdes.py:326-328, or PriorMixSDE.mult_std with L * sigma_mix per sample
    (einsum 'bcdt,bdt->bct')  sdes.py:515-537."""
    out = L @ v
    return out if smix is None else out * smix


def prior_sampling(cfg, y, z, smix=None):
    """MixSDE.prior_sampling  sdes/sdes.py:334-346 (mean 0.5*y broadcast to 2 sources: quirk Q2);
    PriorMixSDE.prior_sampling :564-587 when smix (= sigma_mix(y)) is given: mean 0.5*mix for any S."""
    S = z.shape[1]
    t = torch.ones((y.shape[0],), dtype=y.dtype)
    c = 0.5 if (S == 2 or smix is not None) else 1.0 / S
    return torch.broadcast_to(c * y, z.shape) + _apply_std(mix_std(cfg, t, S), z, smix)


def corrector_ald2(cfg, x, t, score, z, snr, smix=None):
    """AnnealedLangevinDynamics2.update_fn body  sdes/correctors.py:115-126."""
    L = mix_std(cfg, t, x.shape[1])
    g = _apply_std(L, _apply_std(L, score, smix), smix)
    x_mean = x + 2 * snr ** 2 * g
    return x_mean + _apply_std(2 * snr * L, z, smix), x_mean


def corrector_ald(cfg, x, t, score, z, snr):
    """AnnealedLangevinDynamics.update_fn body  sdes/correctors.py:73-89: scalar std from the first row of L L."""
    L = mix_std(cfg, t, x.shape[1])
    std = (L @ L)[:, 0, :].sum(dim=-1, keepdim=True).sqrt()[..., None]
    step = (snr * std) ** 2 * 2
    x_mean = x + step * score
    return x_mean + z * torch.sqrt(step * 2), x_mean


def corrector_langevin(x, score, z, snr):
    """LangevinCorrector.update_fn body  sdes/correctors.py:43-53: ONE step size from batch-mean norms."""
    gn = torch.norm(score.reshape(score.shape[0], -1), dim=-1).mean()
    zn = torch.norm(z.reshape(z.shape[0], -1), dim=-1).mean()
    step = (snr * zn / gn) ** 2 * 2
    x_mean = x + step * score
    return x_mean + z * torch.sqrt(step * 2), x_mean


def scheduled_timesteps(N, eps, schedule, T=1.0):
    """get_pc_scheduled_sampler's N+1 time points  sdes/__init__.py:91-111 (the step stays 1/N: quirk Q1)."""
    if schedule == "linear":
        return torch.linspace(T, eps, N + 1)
    if schedule == "log":
        return torch.logspace(math.log(T) / math.log(10), math.log(eps) / math.log(10), N + 1, base=10)
    if schedule == "revlog":
        return torch.logspace(math.log(eps) / math.log(10), math.log(T) / math.log(10), N + 1, base=10).flip(dims=(0,))
    raise NotImplementedError(schedule)


def predictor_reverse_diffusion(cfg, x, t, score, z, N, smix=None):
    """ReverseDiffusionPredictor.update_fn  sdes/predictors.py:60-66 with RSDE.discretize
    sdes/sdes.py:163-171, SDE.discretize :93-107 (dt = 1/N always: quirk Q1), MixSDE.sde :275-284
    (PriorMixSDE.sde :451-470: the diffusion is additionally scaled per sample by sigma_mix)."""
    S = x.shape[1]
    _, P = mix_mats(S, x.dtype)
    r = cfg["sigma_max"] / cfg["sigma_min"]
    drift = -cfg["d_lambda"] * P @ x
    diffusion = (cfg["sigma_min"] * r ** t * np.sqrt(2 * math.log(r)))[:, None, None]
    if smix is not None:
        diffusion = diffusion * smix
    dt = 1 / N
    f = drift * dt
    G = diffusion * torch.sqrt(torch.tensor(dt, dtype=x.dtype))
    rev_f = f - G ** 2 * score
    x_mean = x - rev_f
    return x_mean + G * z, x_mean


def pc_sampler(p, cfg, y, noise, N=None, corrector_steps=None, snr=None, eps=None, denoise=True, score_fn=None,
               timesteps=None, priormix_avg_len=None, corrector="ald2"):
    """sdes.get_pc_sampler(...)()  sdes/__init__.py:166-188 with predictor 'reverse_diffusion' and
    corrector 'ald2'.  `noise` is the list of N(0,1) draws in the reference's RNG order (Q7):
    prior, then per step: corrector draw(s), predictor draw."""
    N = cfg["N"] if N is None else N
    cs = cfg["corrector_steps"] if corrector_steps is None else corrector_steps
    snr = cfg["snr"] if snr is None else snr
    eps = cfg["t_eps"] if eps is None else eps
    if score_fn is None:
        def score_fn(x, t, m):
            return score_forward(p, cfg, x, t, m)
    it = iter(noise)
    smix = None if priormix_avg_len is None else sigma_mix(y, priormix_avg_len)  # PriorMixSDE (enhancement)
    with torch.no_grad():
        xt = prior_sampling(cfg, y, next(it), smix)
        xm = xt
        ts = torch.linspace(1.0, eps, N, dtype=torch.float32) if timesteps is None else timesteps
        for i in range(N):
            vec_t = torch.ones(y.shape[0], dtype=y.dtype) * ts[i].to(y.dtype)
            for _ in range(cs):
                sc = score_fn(xt, vec_t, y)
                if corrector == "ald2":
                    xt, xm = corrector_ald2(cfg, xt, vec_t, sc, next(it), snr, smix)
                elif corrector == "ald":
                    xt, xm = corrector_ald(cfg, xt, vec_t, sc, next(it), snr)
                else:
                    xt, xm = corrector_langevin(xt, sc, next(it), snr)
            xt, xm = predictor_reverse_diffusion(cfg, xt, vec_t, score_fn(xt, vec_t, y), next(it), N, smix)
    return (xm if denoise else xt), N * (cs + 1)


# ----------------------------------------------------------------------------- SDE object surface
def sde_coefficients(cfg, x, t, smix=None):
    """MixSDE.sde  sdes/sdes.py:275-284 / PriorMixSDE.sde :451-470: drift = -lambda P x, diffusion = sigma_min r^t
    sqrt(2 ln r) [B] (x sigma_mix broadcast to [B,S,T] for PriorMixSDE)."""
    S = x.shape[1]
    _, P = mix_mats(S, x.dtype)
    drift = -cfg["d_lambda"] * (P @ x)
    r = cfg["sigma_max"] / cfg["sigma_min"]
    diffusion = cfg["sigma_min"] * r ** t * math.sqrt(2.0 * math.log(r))
    if smix is not None:
        diffusion = diffusion[:, None, None] * smix[:, None, :].expand(-1, S, -1)
    return drift, diffusion


def sde_discretize(cfg, x, t, N, smix=None):
    """SDE.discretize  sdes/sdes.py:93-107 with dt = 1/N always (quirk Q1)."""
    dt = 1.0 / N
    drift, diffusion = sde_coefficients(cfg, x, t, smix)
    return drift * dt, diffusion * torch.sqrt(torch.tensor(dt))


def sde_mean(cfg, x0, t):
    """MixSDE._mean  sdes/sdes.py:286-294: (A + exp(-lambda t) P) x0."""
    A, P = mix_mats(x0.shape[1], x0.dtype)
    return (A + torch.exp(-t[:, None, None] * cfg["d_lambda"]) * P) @ x0


def sde_std(cfg, t, S, smix=None):
    """MixSDE._std  sdes/sdes.py:315-320 -> [B,S,S]; PriorMixSDE._std :515-532 -> [B,S,S,T]."""
    L = mix_std(cfg, t, S)
    return L if smix is None else L[..., None] * smix[:, None, None, :]


def sde_mult_std(std, x):
    """mult_std  sdes/sdes.py:326-328 (std @ x) / :534-537 (einsum bcdt,bdt->bct)."""
    return std @ x if std.dim() == 3 else torch.einsum("bcdt,bdt->bct", std, x)


def rsde_discretize(cfg, x, t, score, N, smix=None, probability_flow=False):
    """RSDE.discretize  sdes/sdes.py:163-171."""
    f, G = sde_discretize(cfg, x, t, N, smix)
    Gp = G if G.dim() == 3 else G[:, None, None]
    rev_f = f - Gp ** 2 * score * (0.5 if probability_flow else 1.0)
    return rev_f, (torch.zeros_like(G) if probability_flow else G)


# ----------------------------------------------------------------------------- separation metrics
def si_bss_eval_sources(ref, est, clamp_db=100.0):
    """SI-SDR / SI-SIR / SI-SAR with the best permutation as the reference's compute_metrics obtains them
    (evaluate.py:103-111: fast_bss_eval.si_bss_eval_sources(ref, est, zero_mean=False, compute_permutation=True,
    clamp_db=100)).  fast_bss_eval is a third-party package (environment.yaml:32, unpinned) that is NOT under
    /root/reference and not installed here: PARITY UNPINNED for this function.  It restates the package's published
    algorithm (Scheibler, "SDR - medium rare with fast computations", ICASSP 2022, scale-invariant case): with
    unit-norm references r_i and estimates e_j (time domain, float64),
        c_ij  = <r_i, e_j>^2                                  coherence of estimate j with reference i
        p_j   = |projection of e_j on span{r_1..r_S}|^2        (normal equations  R a = <r, e_j>,  R = r r^T)
        SI-SDR_ij = c_ij / (1 - c_ij),  SI-SIR_ij = c_ij / (p_j - c_ij),  SI-SAR_j = p_j / (1 - p_j)
    each mapped to dB through a coherence clamped to [eps', 1 - eps'], eps' = e / (1 + e), e = 10^(-clamp_db / 10)
    (so every figure lies in [-clamp_db, clamp_db]); the permutation maximises the mean SI-SDR.
    ref, est: [B,S,T] arrays.  Returns (sdr, sir, sar [B,S] at the best permutation, perm [B,S]: est[:, perm] ~ ref)."""
    import itertools
    r = np.asarray(ref, dtype=np.float64)
    e = np.asarray(est, dtype=np.float64)
    B, S, _ = r.shape
    tiny = np.finfo(np.float64).tiny
    rn = r / np.maximum(np.linalg.norm(r, axis=-1, keepdims=True), tiny)
    en = e / np.maximum(np.linalg.norm(e, axis=-1, keepdims=True), tiny)
    eps = 10.0 ** (-clamp_db / 10.0)
    eps = eps / (1.0 + eps)

    def to_db(coh):  # coherence -> ratio in dB (10 log10(coh / (1 - coh))), clamped
        coh = np.clip(coh, eps, 1.0 - eps)
        return 10.0 * np.log10(coh / (1.0 - coh))

    sdr = np.zeros((B, S, S)); sir = np.zeros((B, S, S)); sar = np.zeros((B, S, S))
    for b in range(B):
        xc = rn[b] @ en[b].T                      # [ref i, est j]
        R = rn[b] @ rn[b].T
        a = np.linalg.lstsq(R, xc, rcond=None)[0]  # projection coefficients (pseudo-inverse: silent / equal refs)
        p = np.einsum("ij,ij->j", xc, a)          # |proj e_j|^2
        c = xc ** 2
        sdr[b] = to_db(c)
        sir[b] = to_db(c / np.maximum(p[None, :], tiny))
        sar[b] = to_db(p)[None, :].repeat(S, axis=0)
    perms = list(itertools.permutations(range(S)))
    out = [np.zeros((B, S)) for _ in range(3)]
    best = np.zeros((B, S), dtype=np.int64)
    for b in range(B):
        k = int(np.argmax([np.mean([sdr[b, i, q[i]] for i in range(S)]) for q in perms]))
        best[b] = perms[k]
        for i in range(S):
            j = perms[k][i]
            out[0][b, i], out[1][b, i], out[2][b, i] = sdr[b, i, j], sir[b, i, j], sar[b, i, j]
    return out[0], out[1], out[2], best


def normalize_batch(mix):
    """normalize_batch  pl_model.py:81-88."""
    mean = mix.mean(dim=(1, 2), keepdim=True)
    std = mix.std(dim=(1, 2), keepdim=True).clamp(min=1e-5)
    return (mix - mean) / std, mean, std


def scale_output(mix, sep):
    """scale_output  separate.py:73-78."""
    num = (mix * sep).sum(dim=-1, keepdim=True)
    den = (sep * sep + 1e-10).sum(dim=-1, keepdim=True)
    return (num / den) * sep


def separate(p, cfg, mix, noise, **kw):
    """separate()  separate.py:81-99 on a batch mix [B,1,T] (the reference adds the batch dim itself)."""
    mix_norm, _, _ = normalize_batch(mix)
    sep, nfe = pc_sampler(p, cfg, mix_norm, noise, **kw)
    return scale_output(mix, sep), nfe


def to_torch(state, dtype=torch.float32):
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in state.items()}
