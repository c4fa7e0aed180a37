"""Test-side launcher: one rank of `python -m diffsep_amd.evaluate` on a machine WITHOUT a GPU.  Everything that touches
the device is replaced here, in the test process only, by CPU stand-ins (the product has no CPU path and none is added):
the model's sampler returns a deterministic function of (normalised mixture, per-utterance seed), the metric reductions
run in torch on the CPU, "nccl" becomes "gloo".  What runs unmodified is the host logic under test: argument handling,
the rank's share of the utterances (contiguous / --balance), width-bucketed batch planning, per-utterance seeds, the
K-stream launch / finish loop, the gather to rank 0 and the two JSON files.
usage: evaluate_gloo_rank.py <evaluate argv...>   (RANK / WORLD_SIZE / MASTER_* from the environment)"""
import contextlib
import itertools
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import _lib, evaluate  # noqa: E402
from diffsep_amd.pl_model import default_config  # noqa: E402


class _Stream:
    def synchronize(self):
        pass


class _Engine:
    def __init__(self):
        self.cfg = _lib.model_config(nf=16, num_sources=2)
        self.reserved = []

    def padded_frames(self, T):  # the library's own host-side frame arithmetic (no GPU involved)
        import ctypes as C
        return int(_lib.lib().diffsep_padded_frames(C.byref(self.cfg), int(T)))

    def bucket_length(self, W):
        return (W - 1) * 128

    def reserve(self, B, T):
        self.reserved.append((B, T))


class _ScoreModel:
    def __init__(self):
        self._e = _Engine()

    def engine(self):
        return self._e


class _Model:
    calls = []  # (batch size, padded length) of every sampler call of this rank

    def __init__(self, *a, dtype="auto", **k):
        self.config = default_config(nf=16)
        self.dtype = "f16" if dtype == "auto" else dtype
        self.score_model = _ScoreModel()

    def tail_engine(self):
        return None

    def eval(self):
        return self

    def replica(self):
        return _Model(dtype=self.dtype)

    def set_throughput_mode(self, on=True):
        return self

    def fallback_model(self):
        return None if self.dtype == "split" else _Model(dtype="split")

    from diffsep_amd.pl_model import DiffSepModel as _Real
    rerun_if_nonfinite = _Real.rerun_if_nonfinite  # the real overflow net on the stand-in model

    def normalize_batch(self, batch):
        mix, tgt = batch
        mean, std = mix.mean(dim=(1, 2), keepdim=True), mix.std(dim=(1, 2), keepdim=True).clamp(min=1e-5)
        return ((mix - mean) / std, (tgt - mean) / std), mean, std

    def get_pc_sampler(self, pred, corr, y, N=30, corrector_steps=1, lengths=None, seeds=None, check_finite=True, **kw):
        assert pred == "reverse_diffusion" and corr == "ald2" and len(lengths) == len(seeds) == y.shape[0]

        def fn():
            _Model.calls.append((y.shape[0], y.shape[-1]))
            est = torch.zeros(y.shape[0], 2, y.shape[-1])
            for b, (L, s) in enumerate(zip(lengths, seeds)):  # a function of the utterance and ITS seed only
                z = torch.randn(2, L, generator=torch.Generator().manual_seed(int(s) % (2 ** 31)))
                est[b, :, :L] = torch.stack([0.7 * y[b, 0, :L], 0.3 * y[b, 0, :L].flip(-1)]) + 0.05 * z
            if os.environ.get("EVAL_TEST_OVERFLOW") and self.dtype != "split" and y.shape[0] == 3:
                est[0, 0, 5] = float("inf")  # (an f16 overflow in the full batches: evaluate must repeat them on the split twin)
            return est, N * (1 + corrector_steps)
        return fn


def _cpu_metrics(est, ref, n_src=None):
    out = []
    for b in range(est.shape[0]):
        best = None
        for perm in itertools.permutations(range(ref.shape[1])):
            v = []
            for k, p in enumerate(perm):
                r, e = ref[b, k].double(), est[b, p].double()
                a = (e * r).sum() / (r * r).sum().clamp(min=1e-20)
                v.append(float(10 * torch.log10((a * r).pow(2).sum() / (e - a * r).pow(2).sum().clamp(min=1e-20))))
            if best is None or sum(v) > sum(best[0]):
                best = (v, perm)
        k = len(best[0]) if n_src is None else n_src
        out.append({"si_sdr": [best[0][:k]], "si_sir": [best[0][:k]], "si_sar": [best[0][:k]], "perm": list(best[1])})
    return out


def main():
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    torch.cuda.Stream = _Stream
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.Tensor.pin_memory = lambda self: self
    to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: to(self, *["cpu" if x == "cuda" else x for x in a], **k)
    init = dist.init_process_group
    dist.init_process_group = lambda backend, **k: init("gloo")
    evaluate.DiffSepModel = _Model
    evaluate.compute_metrics = _cpu_metrics
    evaluate.main(sys.argv[1:])
    print("CALLS %s %r" % (os.environ.get("RANK", "0"), _Model.calls))


if __name__ == "__main__":
    main()
