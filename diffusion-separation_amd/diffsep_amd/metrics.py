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
