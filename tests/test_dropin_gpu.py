"""The drop-in of INTEGRATION.md section 1 on the real engine: a parent nn.Module in the shape of the reference's LightningModule
(pl_model.py:95-146, 642-670: score_model as a child module, an EMA over self.parameters(), the swap in train() / eval())
holds diffsep_amd's ScoreModelNCSNpp, is moved with .to(device) and switched with .eval() / .train() — and the engine must
run on the weights the parameters hold at that moment.  The reference itself cannot travel to the GPU box; the same cycle
against the reference's own class runs in the build container (tests/test_reference_dropin_cpu.py)."""
import numpy as np
import pytest
import torch
from torch import nn

import diffsep_oracle as O
from diffsep_amd import _lib, synth
from diffsep_amd.engine import Engine, pack_state_dict, param_table
from diffsep_amd.score_models import ScoreModelNCSNpp

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


class Ema:
    """torch_ema.ExponentialMovingAverage's store / copy_to / restore (through param.data.copy_, like the package)"""

    def __init__(self, parameters):
        self.shadow_params = [p.clone().detach() for p in parameters if p.requires_grad]
        self.collected_params = None

    def _get(self, parameters):
        return [p for p in parameters if p.requires_grad]

    def store(self, parameters):
        self.collected_params = [p.clone() for p in self._get(parameters)]

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, self._get(parameters)):
            p.data.copy_(s.data)

    def restore(self, parameters):
        for c, p in zip(self.collected_params, self._get(parameters)):
            p.data.copy_(c.data)
        self.collected_params = None

    def to(self, device):
        self.shadow_params = [p.to(device) for p in self.shadow_params]


class LightningLikeParent(nn.Module):
    def __init__(self, nf=16, dtype="f32"):
        super().__init__()
        self.score_model = ScoreModelNCSNpp(num_sources=2, stft_args=dict(n_fft=510, hop_length=128, center=True,
                                                                           pad_mode="constant"),
                                            backbone_args=dict(_target_="models.ncsnpp.NCSNpp", nf=nf), transform="exponent",
                                            spec_abs_exponent=0.5, spec_factor=0.33, spec_trans_learnable=False, dtype=dtype)
        self.ema = Ema(self.parameters())

    def train(self, mode=True, no_ema=False):  # pl_model.py:650-667
        res = super().train(mode)
        if mode is False and not no_ema:
            self.ema.store(self.parameters())
            self.ema.copy_to(self.parameters())
        elif self.ema.collected_params is not None:
            self.ema.restore(self.parameters())
        return res

    def eval(self, no_ema=False):
        return self.train(False, no_ema=no_ema)

    def to(self, *args, **kwargs):  # pl_model.py:675-678
        self.ema.to(*args, **kwargs)
        return super().to(*args, **kwargs)

    def forward(self, xt, time, mix):  # pl_model.py:407-409
        return self.score_model(xt, time, mix)


def test_parent_module_ema_swap_reaches_the_engine(golden):
    g, meta = golden
    parent = LightningLikeParent()
    sm = parent.score_model
    table = [(n, s) for n, s, _ in param_table(sm.cfg)]
    raw, ema = synth.synth_state_dict(table, 1), synth.synth_state_dict(table, 7)
    res = parent.load_state_dict({"score_model.backbone." + k: torch.from_numpy(v) for k, v in raw.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    names = [n for n, p in sm.named_parameters() if p.requires_grad]
    parent.ema.shadow_params = [torch.from_numpy(ema[n[len("backbone."):]]) for n in names]
    ema_eff = dict(ema)
    ema_eff["all_modules.0.W"] = raw["all_modules.0.W"]  # (the frozen Fourier projection is not shadowed)

    parent.to("cuda:0")
    assert all(p.is_cuda for p in parent.parameters())
    parent.eval()
    assert sm.engine().device == torch.device("cuda", 0)

    B, S, T, N = 2, 2, 4000, 3
    xt = torch.from_numpy(synth.synth_noise("g7.xt", (B, S, T))).cuda() * 0.5
    mix = torch.from_numpy(synth.synth_noise("g7.mix", (B, 1, T))).cuda() * 0.5
    tt = torch.tensor([0.7, 0.05], device="cuda")

    def direct(weights):
        e = Engine(sm.cfg, pack_state_dict(sm.cfg, weights), device="cuda:0")
        out = e.score(xt, tt, mix)
        torch.cuda.synchronize()
        e.close()
        return out

    s_eval = parent(xt, tt, mix)
    assert torch.equal(s_eval, direct(ema_eff))
    builds = sm._slot.builds
    assert torch.equal(parent(xt, tt, mix), s_eval) and sm._slot.builds == builds  # (no re-pack per call)

    # the reference's sampler loop (sdes/__init__.py:166-188: torch arithmetic for the SDE, the model as score function) on the
    # EMA weights.  The golden sampler result g9_sep was produced by the reference with the seed-7 weights, Fourier projection
    # included — shadow everything for this part, as torch_ema 0.3 does
    with torch.no_grad():
        getattr(sm.backbone.all_modules, "0").W.data.copy_(torch.from_numpy(ema["all_modules.0.W"]))
    sm.weights_changed()  # (a write through .data by hand: said so, as the docstring asks)
    cfg = O.default_config(16, 2)
    mixb = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    mix_norm = O.normalize_batch(mixb)[0]
    draws = [torch.from_numpy(synth.synth_noise(f"g9.z{i}", (B, S, T))) for i in range(1 + 2 * N)]
    x, nfe = O.pc_sampler(None, cfg, mix_norm, draws, N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True,
                          score_fn=lambda a, t, m: parent(a.cuda(), t.cuda(), m.cuda()).cpu())
    assert nfe == meta["g9_nfe"]
    err = float(np.sqrt(np.mean((x.numpy().astype(np.float64) - g["g9_sep"]) ** 2)) / np.sqrt(np.mean(g["g9_sep"] ** 2.0)))
    assert err < 1e-4, err

    # train(): the raw weights come back (pl_model.py:662-666) and the engine follows
    parent.train()
    raw_now = dict(raw)
    raw_now["all_modules.0.W"] = ema["all_modules.0.W"]  # (written by hand above; frozen, so not part of the swap)
    assert torch.equal(parent(xt, tt, mix), direct(raw_now))
    assert not torch.equal(parent(xt, tt, mix), s_eval)
    parent.eval(no_ema=True)
    assert torch.equal(parent(xt, tt, mix), direct(raw_now))
