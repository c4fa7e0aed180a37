"""Host-side metrics on the CPU: STOI / ESTOI (diffsep_amd.metrics.stoi, what evaluate.py:113-130
gets from pystoi) against an independent loop-form restatement of the papers (oracle/stoi_oracle.py) and against the
properties the papers state; SI-SDR / SIR / SAR: the oracle's coherence form against the explicit time-domain decomposition
(target / interference / artefact components built sample by sample).  Third-party packages (pystoi, fast_bss_eval) are
absent: no vectors of theirs can be generated here — these tests are what pins the restatements."""
import numpy as np
import pytest

import diffsep_oracle as O
import stoi_oracle as SO
from diffsep_amd import synth
from diffsep_amd.metrics import _resample, _thirdoct, stoi


def _speechlike(i, T, fs=8000):
    return synth.synth_mixture(i, T=T, fs=fs)[1][0].astype(np.float64)


@pytest.mark.parametrize("fs", [8000, 16000, 10000])
@pytest.mark.parametrize("extended", [True, False])
def test_stoi_matches_the_loop_form_restatement(fs, extended):
    T = int(1.6 * fs)
    x = _speechlike(1, T, fs)
    rng = np.random.default_rng(fs)
    for snr in (15.0, 0.0):
        y = x + rng.standard_normal(T) * np.std(x) * 10 ** (-snr / 20)
        a, b = stoi(x, y, fs, extended), SO.stoi(x, y, fs, extended)
        assert abs(a - b) < 1e-9, (a, b)


def test_stoi_properties():
    fs, T = 8000, 32000
    x = _speechlike(0, T)
    rng = np.random.default_rng(0)
    n = rng.standard_normal(T) * np.std(x)
    for ext in (True, False):
        assert abs(stoi(x, x, fs, ext) - 1.0) < 1e-9                      # identical signals: 1
        vals = [stoi(x, x + n * 10 ** (-snr / 20), fs, ext) for snr in (30, 20, 10, 0, -10)]
        assert all(a > b for a, b in zip(vals, vals[1:])) and vals[0] > 0.99 and vals[-1] < 0.45, vals   # monotone in the SNR
        y = x + n * 0.3
        assert abs(stoi(x, 7.5 * y, fs, ext) - stoi(x, y, fs, ext)) < 1e-9  # the gain of the processed signal does not matter
        assert abs(stoi(x, rng.standard_normal(T), fs, ext)) < 0.2        # unrelated noise: about 0
    # fewer than 30 frames (384 ms) after the removal of silent frames: the package's sentinel
    assert stoi(x[:2000], x[:2000], fs) == 1e-5
    # silence (40 dB below the loudest frame) is removed before anything is compared
    pad = np.zeros(8000)
    y = x + n * 0.3
    a = stoi(x, y, fs)
    b = stoi(np.concatenate([pad, x, pad]), np.concatenate([pad + 1e-3 * rng.standard_normal(8000), y, pad]), fs)
    assert abs(a - b) < 0.01
    with pytest.raises(ValueError):
        stoi(x, x[:-1], fs)


def test_third_octave_bands_and_resampler():
    obm = _thirdoct(10000, 512, 15, 150.0)
    assert obm.shape == (15, 257) and set(np.unique(obm)) == {0.0, 1.0}
    f = np.linspace(0, 10000, 513)[:257]
    cf = 150.0 * 2.0 ** (np.arange(15) / 3.0)
    for k in range(15):
        bins = np.nonzero(obm[k])[0]
        assert bins.size > 0 and np.all(np.diff(bins) == 1)               # one contiguous run of bins per band
        assert f[bins[0]] <= cf[k] <= f[bins[-1]] + 10000 / 512            # ... around its centre frequency
    assert (obm.sum(0) <= 1).all()                                         # bands do not overlap
    assert abs(f[np.nonzero(obm[-1])[0][-1]] - 150.0 * 2 ** (14 / 3) * 2 ** (1 / 6)) < 40  # top edge ~ 4.3 kHz
    assert SO.band_edges() == [(int(np.nonzero(r)[0][0]), int(np.nonzero(r)[0][-1]) + 1) for r in obm]
    for fs in (8000, 16000):                                               # a 1 kHz tone keeps frequency and amplitude
        t = np.arange(fs) / fs
        y = _resample(np.sin(2 * np.pi * 1000 * t), 10000, fs)
        assert len(y) == 10000
        want = np.sin(2 * np.pi * 1000 * np.arange(10000) / 10000)
        assert np.max(np.abs(y[500:-500] - want[500:-500])) < 2e-3
        assert np.max(np.abs(SO.resample_10k(np.sin(2 * np.pi * 1000 * t), fs) - y)) < 1e-9
    # a tone above the new Nyquist frequency is rejected (16 kHz -> 10 kHz: 6.5 kHz must not alias to 3.5 kHz)
    t = np.arange(16000) / 16000
    assert np.max(np.abs(_resample(np.sin(2 * np.pi * 6500 * t), 10000, 16000)[500:-500])) < 2e-3


@pytest.mark.parametrize("S", [2, 3])
def test_si_bss_eval_against_the_time_domain_decomposition(S):
    # est_j = s_target + e_interf + e_artif built explicitly (projection on ONE reference, on ALL references, the rest):
    # the three ratios of those sample vectors must equal the oracle's coherence form and the Gram form of the product
    rng = np.random.default_rng(S)
    T = 6000
    ref = rng.standard_normal((2, S, T))
    mixm = np.eye(S) + 0.25 * rng.standard_normal((S, S))
    est = np.einsum("ij,bjt->bit", mixm, ref) + 0.1 * rng.standard_normal((2, S, T))
    order = list(range(S))[::-1]
    est = est[:, order]
    sdr, sir, sar, perm = O.si_bss_eval_sources(ref, est)
    for b in range(2):
        assert [order[p] for p in perm[b]] == list(range(S))             # est[:, perm] is aligned with ref
        for i in range(S):
            e = est[b, perm[b][i]]
            s_t = (e @ ref[b, i]) / (ref[b, i] @ ref[b, i]) * ref[b, i]
            coef, *_ = np.linalg.lstsq(ref[b].T, e, rcond=None)
            p_all = coef @ ref[b]
            e_i, e_a = p_all - s_t, e - p_all
            assert abs(sdr[b, i] - 10 * np.log10((s_t @ s_t) / ((e - s_t) @ (e - s_t)))) < 1e-8
            assert abs(sir[b, i] - 10 * np.log10((s_t @ s_t) / (e_i @ e_i))) < 1e-8
            assert abs(sar[b, i] - 10 * np.log10((p_all @ p_all) / (e_a @ e_a))) < 1e-8
