"""CPU ORACLE — test infrastructure only (tests/ may import it; the product never does).

STOI / ESTOI in loop form: one frame, one band, one segment at a time, written from the papers' equations —
  [1] C. H. Taal, R. C. Hendriks, R. Heusdens, J. Jensen, "An Algorithm for Intelligibility Prediction of Time-Frequency
      Weighted Noisy Speech", IEEE Trans. Audio, Speech, Lang. Process. 19(7), 2011: eqs. (1)-(6);
  [2] J. Jensen, C. H. Taal, "An Algorithm for Predicting the Intelligibility of Speech Masked by Modulated Noise Maskers",
      IEEE/ACM Trans. Audio, Speech, Lang. Process. 24(11), 2016: Section III (row / column normalised segments)
— as the check of the vectorised diffsep_amd.metrics.stoi, which is what the reference's evaluate.py:113-130 obtains from the
third-party package pystoi (not installed here, not under /root/reference: PARITY UNPINNED against the package; the two
restatements pin each other and the papers' properties are asserted in tests/test_metrics_cpu.py).
Constants of the authors' implementation: fs 10 kHz, 256-sample frames (Hann, 50 % overlap) zero-padded to 512, J = 15
one-third octave bands from 150 Hz, N = 30 frames per segment, beta = -15 dB, 40 dB dynamic range for silent frames.
"""
import math

import numpy as np

FS, NW, NFFT, J, FMIN, N, BETA, DYN = 10000, 256, 512, 15, 150.0, 30, -15.0, 40.0
EPS = float(np.finfo(np.float64).eps)


def hann(n):
    # Matlab hanning(n): w[k] = 0.5 (1 - cos(2 pi (k + 1) / (n + 1))), k = 0 .. n-1
    return np.array([0.5 * (1.0 - math.cos(2.0 * math.pi * (k + 1) / (n + 1))) for k in range(n)])


def resample_10k(x, fs):
    """polyphase resampling by p / q = 10000 / fs with a Kaiser-windowed sinc (Octave's resample design, 60 dB), direct form:
    y[m] = sum_k h[k] u[m q - k + delay] with u = x zero-stuffed by p"""
    g = math.gcd(FS, int(fs))
    p, q = FS // g, int(fs) // g
    fc = 1.0 / (2.0 * max(p, q))
    L = int(math.ceil((60.0 - 8.0) / (28.714 * fc / 10.0)))
    t = np.arange(-L, L + 1)
    h = np.kaiser(2 * L + 1, 0.1102 * (60.0 - 8.7)) * (2.0 * p * fc * np.sinc(2.0 * fc * t))
    h = h / h.sum() * p  # unit DC gain after the zero stuffing
    u = np.zeros(len(x) * p)
    u[::p] = x
    full = np.convolve(u, h)[L:L + len(u)]  # centred (zero-phase) filter
    n_out = -(-len(x) * p // q)
    return full[::q][:n_out]


def band_edges():
    f = [FS * i / NFFT for i in range(NFFT // 2 + 1)]
    edges = []
    for k in range(J):
        lo = FMIN * 2.0 ** ((2 * k - 1) / 6.0)
        hi = FMIN * 2.0 ** ((2 * k + 1) / 6.0)
        a = min(range(len(f)), key=lambda i: (f[i] - lo) ** 2)
        b = min(range(len(f)), key=lambda i: (f[i] - hi) ** 2)
        edges.append((a, b))
    return edges


def stoi(x, y, fs, extended=True):
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    if int(fs) != FS:
        x, y = resample_10k(x, fs), resample_10k(y, fs)
    w = hann(NW)
    hop = NW // 2
    # ---- silent frames: energy of the windowed CLEAN frame more than 40 dB below the loudest one
    starts = list(range(0, len(x) - NW, hop))
    xf = [w * x[s:s + NW] for s in starts]
    yf = [w * y[s:s + NW] for s in starts]
    if not xf:
        return 1e-5
    e = [20.0 * math.log10(math.sqrt(float(np.dot(f, f))) + EPS) for f in xf]
    keep = [i for i in range(len(xf)) if max(e) - DYN - e[i] < 0]
    xs = np.zeros((len(keep) - 1) * hop + NW if keep else 0)
    ys = np.zeros_like(xs)
    for n_, i in enumerate(keep):
        xs[n_ * hop:n_ * hop + NW] += xf[i]
        ys[n_ * hop:n_ * hop + NW] += yf[i]
    # ---- one-third octave band magnitudes, eq. (1) of [1]
    edges = band_edges()
    X, Y = [], []
    for s in range(0, len(xs) - NW, hop):
        fx = np.abs(np.fft.rfft(w * xs[s:s + NW], NFFT)) ** 2
        fy = np.abs(np.fft.rfft(w * ys[s:s + NW], NFFT)) ** 2
        X.append([math.sqrt(float(fx[a:b].sum())) for a, b in edges])
        Y.append([math.sqrt(float(fy[a:b].sum())) for a, b in edges])
    X, Y = np.array(X).T if X else np.zeros((J, 0)), np.array(Y).T if Y else np.zeros((J, 0))  # [band, frame]
    M = X.shape[1]
    if M < N:
        return 1e-5
    total, count = 0.0, 0
    c = 10.0 ** (-BETA / 20.0)
    for m in range(N, M + 1):  # segment of frames m-N .. m-1
        xseg, yseg = X[:, m - N:m].copy(), Y[:, m - N:m].copy()
        if extended:  # [2]: rows then columns to zero mean and unit norm; d = mean over segments of (1/N) sum of products
            for seg in (xseg, yseg):
                for j in range(J):
                    seg[j] -= seg[j].mean()
                    seg[j] /= math.sqrt(float(np.dot(seg[j], seg[j]))) + EPS
                for n_ in range(N):
                    seg[:, n_] -= seg[:, n_].mean()
                    seg[:, n_] /= math.sqrt(float(np.dot(seg[:, n_], seg[:, n_]))) + EPS
            total += float((xseg * yseg).sum()) / N
            count += 1
        else:  # [1] eqs. (2)-(5): normalise, clip, correlate per band
            for j in range(J):
                alpha = math.sqrt(float(np.dot(xseg[j], xseg[j]))) / (math.sqrt(float(np.dot(yseg[j], yseg[j]))) + EPS)
                yp = np.minimum(alpha * yseg[j], (1.0 + c) * xseg[j])
                xc, yc = xseg[j] - xseg[j].mean(), yp - yp.mean()
                total += float(np.dot(xc / (math.sqrt(float(np.dot(xc, xc))) + EPS), yc / (math.sqrt(float(np.dot(yc, yc))) + EPS)))
                count += 1
    return total / count
