"""Separation metrics from Gram matrices (the waveform reductions run in the HIP `gram` kernel; the S x S algebra
below is host-side bookkeeping on a handful of float64 numbers per utterance).

Definitions follow fast_bss_eval.si_bss_eval_sources(ref, est, zero_mean=False, compute_permutation=True) as called by
the reference (evaluate.py:105-111): for estimate j scored against reference i
    s_target = <est_j, ref_i>/<ref_i, ref_i> ref_i
    e_proj   = orthogonal projection of est_j on span{ref_1..ref_S};  e_interf = e_proj - s_target;  e_artif = est_j - e_proj
    SI-SDR = |s_target|^2 / |est_j - s_target|^2,  SI-SIR = |s_target|^2 / |e_interf|^2,  SI-SAR = |e_proj|^2 / |e_artif|^2
and the permutation maximising the mean SI-SDR is returned.  fast_bss_eval itself is not installed here, so SIR/SAR are
pinned by these definitions (tests compare against a float64 time-domain restatement), SI-SDR also by closed-form cases.
"""
import itertools

import numpy as np
import torch

from . import ops


def _db(num, den):
    return 10.0 * np.log10(np.maximum(num, 1e-300) / np.maximum(den, 1e-300))


def si_bss_eval_sources(ref, est, clamp_db=100.0):
    """ref, est [B,S,T] device tensors -> (si_sdr, si_sir, si_sar [B,S] numpy, perm [B,S] int) with est[:, perm] aligned
    to ref."""
    Gd = ops.gram(ref.float(), est.float())
    # device -> pinned host on the CURRENT stream only (a pageable copy would stall every other stream's work)
    Gh = torch.empty(Gd.shape, dtype=Gd.dtype, pin_memory=True)
    Gh.copy_(Gd, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    G = Gh.numpy()
    B, _, S, _ = G.shape
    sdr = np.zeros((B, S, S)); sir = np.zeros((B, S, S)); sar = np.zeros((B, S, S))
    for b in range(B):
        Grr, Gre, Gee = G[b, 0], G[b, 1], G[b, 2]
        for j in range(S):  # estimate j
            c = np.linalg.lstsq(Grr, Gre[:, j], rcond=None)[0]      # projection coefficients on all references
            proj2 = float(c @ Grr @ c)
            for i in range(S):  # reference i
                a = Gre[i, j] / max(Grr[i, i], 1e-300)
                tgt2 = a * a * Grr[i, i]
                sdr[b, i, j] = _db(tgt2, Gee[j, j] - 2 * a * Gre[i, j] + tgt2)
                interf2 = proj2 - 2 * a * float(c @ Grr[:, i]) + tgt2
                sir[b, i, j] = _db(tgt2, interf2)
                sar[b, i, j] = _db(proj2, Gee[j, j] - proj2)
    for m in (sdr, sir, sar):
        np.clip(m, -clamp_db, clamp_db, out=m)
    perms = list(itertools.permutations(range(S)))
    out = [np.zeros((B, S)) for _ in range(3)]
    best = np.zeros((B, S), dtype=np.int64)
    for b in range(B):
        scores = [np.mean([sdr[b, i, p[i]] for i in range(S)]) for p in perms]
        p = perms[int(np.argmax(scores))]
        best[b] = p
        for i in range(S):
            out[0][b, i], out[1][b, i], out[2][b, i] = sdr[b, i, p[i]], sir[b, i, p[i]], sar[b, i, p[i]]
    return out[0], out[1], out[2], best


# ---------------------------------------------------------------------------------------------------------------- STOI / ESTOI
# The reference scores every source with pystoi's stoi(ref, est, fs, extended=True) (evaluate.py:113-130).  pystoi is a
# third-party package that is not installed here and is not under /root/reference: what follows restates the PUBLISHED
# algorithm — Taal, Hendriks, Heusdens, Jensen, "An Algorithm for Intelligibility Prediction of Time-Frequency Weighted Noisy
# Speech" (IEEE TASLP 2011) and Jensen, Taal, "An Algorithm for Predicting the Intelligibility of Speech Masked by Modulated
# Noise Maskers" (IEEE/ACM TASLP 2016, the extended measure) — with the constants of the authors' Matlab code that pystoi
# follows: 10 kHz, 256-sample Hann frames at 50 % overlap zero-padded to 512 bins, 15 one-third octave bands from 150 Hz,
# 30-frame (384 ms) segments, -15 dB clipping, 40 dB dynamic range for the silent-frame removal.  PARITY UNPINNED against the
# package (no vectors can be fetched); pinned by an independent loop-form restatement (oracle/stoi_oracle.py) and by the
# properties the papers state (tests/test_metrics_cpu.py).  Host-side numpy on a few seconds of audio per source.
_STOI_FS, _STOI_FRAME, _STOI_NFFT, _STOI_BANDS, _STOI_MINFREQ, _STOI_SEG, _STOI_BETA, _STOI_DYN = 10000, 256, 512, 15, 150.0, 30, -15.0, 40.0
_EPS = float(np.finfo(np.float64).eps)


def _hann_matlab(n):
    """Matlab's hanning(n): the n interior points of a symmetric (n + 2)-point Hann window (no zeros at the ends)"""
    return np.hanning(n + 2)[1:-1]


def _resample_filter(p, q):
    """Anti-aliasing FIR of Octave's resample(x, p, q) (Kaiser-windowed sinc, 60 dB rejection), unit DC gain"""
    g = int(np.gcd(p, q))
    p, q = p // g, q // g
    cutoff = 1.0 / (2.0 * max(p, q))
    rolloff = cutoff / 10.0
    rej_db = 60.0
    L = int(np.ceil((rej_db - 8.0) / (28.714 * rolloff)))
    t = np.arange(-L, L + 1)
    ideal = 2.0 * p * cutoff * np.sinc(2.0 * cutoff * t)
    beta = 0.1102 * (rej_db - 8.7)
    h = np.kaiser(2 * L + 1, beta) * ideal
    return h / np.sum(h), p, q


def _resample(x, fs_to, fs_from):
    from scipy.signal import resample_poly
    h, p, q = _resample_filter(int(fs_to), int(fs_from))
    return resample_poly(x, p, q, window=h)


def _thirdoct(fs, nfft, num_bands, min_freq):
    """[num_bands, nfft/2+1] 0/1 matrix: DFT bins of each one-third octave band (band edges snapped to the nearest bin)"""
    f = np.linspace(0, fs, nfft + 1)[: nfft // 2 + 1]
    k = np.arange(num_bands, dtype=np.float64)
    lo = min_freq * 2.0 ** ((2 * k - 1) / 6.0)
    hi = min_freq * 2.0 ** ((2 * k + 1) / 6.0)
    obm = np.zeros((num_bands, f.size))
    for i in range(num_bands):
        a = int(np.argmin((f - lo[i]) ** 2))
        b = int(np.argmin((f - hi[i]) ** 2))
        obm[i, a:b] = 1.0
    return obm


def _frames(x, n, hop):
    """[number of frames, n] view-free framing; the last start is < len(x) - n (the authors' loop bound)"""
    starts = np.arange(0, len(x) - n, hop)
    if starts.size == 0:
        return np.zeros((0, n))
    return x[starts[:, None] + np.arange(n)[None, :]]


def _remove_silent_frames(x, y, dyn_range, n, hop):
    w = _hann_matlab(n)
    xf, yf = _frames(x, n, hop) * w, _frames(y, n, hop) * w
    if xf.shape[0] == 0:
        return np.zeros(0), np.zeros(0)
    energy = 20.0 * np.log10(np.linalg.norm(xf, axis=1) + _EPS)
    keep = (np.max(energy) - dyn_range - energy) < 0
    xf, yf = xf[keep], yf[keep]
    out = []
    for fr in (xf, yf):  # overlap-add of the kept frames
        sig = np.zeros((fr.shape[0] - 1) * hop + n if fr.shape[0] else 0)
        for i in range(fr.shape[0]):
            sig[i * hop:i * hop + n] += fr[i]
        out.append(sig)
    return out[0], out[1]


_OBM = None


def stoi(ref, est, fs, extended=True):
    """Short-time objective intelligibility of `est` with respect to the clean `ref` (1-D arrays, same length), sample rate
    fs.  extended=True: ESTOI (the reference's default, evaluate.py:215-217 --stoi-no-extended switches it off).  Signals with
    fewer than 30 frames after the removal of silent frames return 1e-5, like the package."""
    global _OBM
    x = np.asarray(ref, dtype=np.float64).reshape(-1)
    y = np.asarray(est, dtype=np.float64).reshape(-1)
    if x.shape != y.shape:
        raise ValueError("stoi: ref and est must have the same length")
    if int(fs) != _STOI_FS:
        x, y = _resample(x, _STOI_FS, fs), _resample(y, _STOI_FS, fs)
    x, y = _remove_silent_frames(x, y, _STOI_DYN, _STOI_FRAME, _STOI_FRAME // 2)
    w = _hann_matlab(_STOI_FRAME)
    X = np.fft.rfft(_frames(x, _STOI_FRAME, _STOI_FRAME // 2) * w, n=_STOI_NFFT).T  # [bins, frames]
    Y = np.fft.rfft(_frames(y, _STOI_FRAME, _STOI_FRAME // 2) * w, n=_STOI_NFFT).T
    if X.shape[-1] < _STOI_SEG:
        return 1e-5
    if _OBM is None:
        _OBM = _thirdoct(_STOI_FS, _STOI_NFFT, _STOI_BANDS, _STOI_MINFREQ)
    xt = np.sqrt(_OBM @ (np.abs(X) ** 2))  # [bands, frames]
    yt = np.sqrt(_OBM @ (np.abs(Y) ** 2))
    n = xt.shape[1] - _STOI_SEG + 1
    idx = np.arange(n)[:, None] + np.arange(_STOI_SEG)[None, :]
    xs, ys = xt[:, idx].transpose(1, 0, 2), yt[:, idx].transpose(1, 0, 2)  # [segments, bands, 30]
    if extended:
        def rowcol(a):  # rows (bands) then columns (frames) to zero mean / unit norm
            a = a - a.mean(axis=2, keepdims=True)
            a = a / (np.sqrt((a * a).sum(axis=2, keepdims=True)) + _EPS)
            a = a - a.mean(axis=1, keepdims=True)
            return a / (np.sqrt((a * a).sum(axis=1, keepdims=True)) + _EPS)
        return float(np.sum(rowcol(xs) * rowcol(ys)) / _STOI_SEG / n)
    norm = np.linalg.norm(xs, axis=2, keepdims=True) / (np.linalg.norm(ys, axis=2, keepdims=True) + _EPS)
    yn = ys * norm
    yp = np.minimum(yn, xs * (1.0 + 10.0 ** (-_STOI_BETA / 20.0)))
    yp = yp - yp.mean(axis=2, keepdims=True)
    xz = xs - xs.mean(axis=2, keepdims=True)
    yp = yp / (np.linalg.norm(yp, axis=2, keepdims=True) + _EPS)
    xz = xz / (np.linalg.norm(xz, axis=2, keepdims=True) + _EPS)
    return float(np.sum(yp * xz) / (n * _STOI_BANDS))
