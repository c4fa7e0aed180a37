"""Portable, counter-based synthetic data: weights, noise and 2-speaker-like mixtures.

Everything here is a pure function of (name/seed, element index) computed with numpy uint64
arithmetic (splitmix64), so the golden-vector generator that runs beside the reference, the
CPU oracle and the GPU tests all regenerate bit-identical tensors without storing them.
There is no network in the build environment: no checkpoint or dataset can be fetched, so
weights follow the reference's *initialiser family* (variance-scaling fan_avg uniform,
models/ncsnpp_utils/layers.py:61-102) but with scale 1.0 on EVERY layer — the reference's
init_scale=0 layers would make the network output ~1e-10 and hide every kernel from the test.
"""
import zlib

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _key(name, seed):
    crc = np.uint64(zlib.crc32(name.encode("utf-8")))
    with np.errstate(over="ignore"):
        return (crc << np.uint64(32)) ^ (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15))


def uniform01(name, n, seed=0, offset=0):
    """n float64 draws in (0, 1), stream keyed by (name, seed)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64(_key(name, seed) + idx)
    return ((z >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(name, n, seed=0):
    """n float64 N(0,1) draws (Box–Muller on two decorrelated uniform streams)."""
    u1 = uniform01(name + "#u1", n, seed)
    u2 = uniform01(name + "#u2", n, seed)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def synth_param(name, shape, seed=0):
    """One parameter tensor (float32 numpy) for the reference key `name` with `shape`."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape))
    leaf = name.split(".")[-1]
    parent = name.split(".")[-2] if "." in name else ""
    if leaf == "W" and len(shape) == 1:  # GaussianFourierProjection.W ~ N(0, 16^2)  (layerspp.py:35-37)
        v = normal(name, n, seed) * 16.0
    elif leaf in ("weight", "W") and len(shape) >= 2:
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        a = np.sqrt(3.0 * 1.0 / ((fan_in + fan_out) / 2.0))
        v = (uniform01(name, n, seed) * 2.0 - 1.0) * a
    elif leaf == "weight" and parent.startswith("GroupNorm") or (leaf == "weight" and len(shape) == 1):
        v = 1.0 + (uniform01(name, n, seed) * 2.0 - 1.0) * 0.2  # GroupNorm gamma
    else:  # biases (conv / dense / NIN.b / GroupNorm beta)
        v = (uniform01(name, n, seed) * 2.0 - 1.0) * 0.1
    return v.astype(np.float32).reshape(shape)


def synth_state_dict(param_table, seed=0):
    """param_table: iterable of (name, shape). Returns {name: float32 ndarray}."""
    return {name: synth_param(name, shape, seed) for name, shape in param_table}


def synth_noise(tag, shape, seed=0):
    """Standard-normal float32 tensor keyed by `tag` (used as injected sampler noise)."""
    n = int(np.prod(shape))
    return normal("noise:" + tag, n, seed).astype(np.float32).reshape(shape)


def synth_mixture(index, T=32000, fs=8000, n_src=2, seed=0):
    """Speech-like 2-speaker mixture: each source = unit Gaussian noise through a
    random 1-pole low-pass with a ~4 Hz amplitude envelope; mix = sum, peak-normalised to 0.9.
    Returns (mix[1,T], sources[n_src,T]) float32."""
    srcs = []
    t = np.arange(T, dtype=np.float64) / fs
    for s in range(n_src):
        tag = f"mix{index}:src{s}"
        g = normal(tag, T, seed)
        u = uniform01(tag + ":par", 3, seed)
        pole = 0.6 + 0.35 * u[0]
        # 1-pole IIR y[n] = (1-p) g[n] + p y[n-1] evaluated by recursive doubling (pure numpy)
        y = g * (1.0 - pole)
        shift, p = 1, pole
        while shift < T:
            y[shift:] = y[shift:] + p * y[:-shift]
            p, shift = p * p, shift * 2
        y = y / (np.std(y) + 1e-12)
        f_am = 3.0 + 2.0 * u[1]
        env = 0.55 + 0.45 * np.sin(2.0 * np.pi * f_am * t + 2.0 * np.pi * u[2])
        srcs.append(y * env)
    srcs = np.stack(srcs, 0)
    mix = srcs.sum(0, keepdims=True)
    g = 0.9 / (np.abs(mix).max() + 1e-12)
    return (mix * g).astype(np.float32), (srcs * g).astype(np.float32)


def synth_batch(B, T=32000, fs=8000, n_src=2, seed=0, start=0):
    mixes, tgts = zip(*[synth_mixture(start + i, T, fs, n_src, seed) for i in range(B)])
    return np.stack(mixes, 0), np.stack(tgts, 0)
