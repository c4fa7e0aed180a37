#!/usr/bin/env python3
"""Round-2 golden vectors (tests/golden/golden_ref2.npz), again produced by RUNNING THE REFERENCE ITSELF on CPU.

    python tests/golden/gen_golden_r2.py          (build container only: needs /root/reference)

Same method and stand-ins as gen_golden.py (imported from there); kept in a second file so that golden_ref.npz stays
byte-identical.  Contents:
  g13_*  the SDE object surface a user-written predictor / corrector reaches: MixSDE / PriorMixSDE .sde(),
         .marginal_prob(), .discretize(), .reverse(score).discretize() (sdes/sdes.py:93-173,275-328,451-537)
  g14_*  the published model width nf = 128 (config/experiment/icassp-separation.yaml:14-18, config/model/nr.yaml):
         one score evaluation (spec_factor 0.15) and a PriorMixSDE N=2 sampler at 16 kHz settings
  g15_*  three sources: ald2 corrector and reverse-diffusion predictor updates with S = 3 (the reference's
         MixSDE.prior_sampling is undefined for S = 3 — quirk Q2 — so there is no full-sampler vector)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402
from gen_golden import AD, InjectedNoise, load_synth_weights, model_config, synth  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    G._install_stubs()
    sys.path.insert(0, G.REF)
    import sdes as ref_sdes  # reference
    import pl_model as ref_pl  # reference

    out = {}
    B, S, T, N = 2, 2, 4000, 3
    cfg = model_config(16, S)
    model = ref_pl.DiffSepModel(cfg)
    load_synth_weights(model.score_model.backbone, 7)
    model.eval()
    mixb = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    (mix_norm, _), _, _ = model.normalize_batch((mixb, None))
    x0 = torch.from_numpy(synth.synth_noise("g9.x0", (B, S, T))) * 0.5
    tv = torch.tensor([0.8, 0.2])

    # ---- G13 SDE surface
    sde = ref_sdes.sdes.MixSDE(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, N=N)
    psde = ref_sdes.sdes.PriorMixSDE(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, N=N)
    for tag, s in (("mix", sde), ("pmix", psde)):
        drift, diff = s.sde(x0, tv, mix_norm)
        out[f"g13_{tag}_drift"], out[f"g13_{tag}_diffusion"] = drift.numpy(), diff.numpy()
        mean, std = s.marginal_prob(x0, tv, mix_norm)
        out[f"g13_{tag}_mean"], out[f"g13_{tag}_std"] = mean.numpy(), std.numpy()
        out[f"g13_{tag}_mult_std"] = s.mult_std(std, x0).numpy()
        f, Gd = s.discretize(x0, tv, mix_norm)
        out[f"g13_{tag}_f"], out[f"g13_{tag}_G"] = f.numpy(), Gd.numpy()
        rs = s.reverse(model)
        rf, rG = rs.discretize(x0, tv, mix_norm)
        out[f"g13_{tag}_rev_f"], out[f"g13_{tag}_rev_G"] = rf.numpy(), rG.numpy()
        td, dd = rs.sde(x0, tv, mix_norm)
        out[f"g13_{tag}_rsde_drift"] = td.numpy()

    # ---- G14 nf = 128, spec_factor 0.15
    cfg128 = model_config(128, S, spec_factor=0.15)
    cfg128["model"]["sde"] = AD(_target_="sdes.sdes.PriorMixSDE", ndim=S, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, N=30)
    m128 = ref_pl.DiffSepModel(cfg128)
    load_synth_weights(m128.score_model.backbone, 7)
    m128.eval()
    xt = torch.from_numpy(synth.synth_noise("g7.xt", (1, S, T))) * 0.5
    mx = torch.from_numpy(synth.synth_noise("g7.mix", (1, 1, T))) * 0.5
    out["g14_score_nf128"] = m128(xt, torch.tensor([0.6]), mx).numpy()
    draws = [torch.from_numpy(synth.synth_noise(f"g14.z{i}", (1, S, T))) for i in range(1 + 2 * 2)]
    with InjectedNoise(draws) as inj:
        sampler = m128.get_pc_sampler("reverse_diffusion", "ald2", mix_norm[:1], N=2, denoise=True, intermediate=False,
                                      corrector_steps=1, snr=0.5, schedule=None)
        out["g14_priormix_sep_nf128"] = sampler()[0].numpy()
        assert inj.i == len(draws)

    # ---- G15 three sources: isolated updates
    cfg3 = model_config(16, 3)
    m3 = ref_pl.DiffSepModel(cfg3)
    load_synth_weights(m3.score_model.backbone, 7)
    m3.eval()
    sde3 = m3.sde.copy()
    sde3.N = N
    x03 = torch.from_numpy(synth.synth_noise("g15.x0", (B, 3, T))) * 0.5
    z3 = [torch.from_numpy(synth.synth_noise(f"g15.z{i}", (B, 3, T))) for i in range(2)]
    corr = ref_sdes.CorrectorRegistry.get_by_name("ald2")(sde3, m3, snr=0.5, n_steps=1)
    pred = ref_sdes.PredictorRegistry.get_by_name("reverse_diffusion")(sde3, m3)
    with InjectedNoise([z3[0]]):
        xc, xcm = corr.update_fn(x03, tv, mix_norm)
    with InjectedNoise([z3[1]]):
        xp, xpm = pred.update_fn(x03, tv, mix_norm)
    out["g15_corr_x"], out["g15_corr_mean"] = xc.numpy(), xcm.numpy()
    out["g15_pred_x"], out["g15_pred_mean"] = xp.numpy(), xpm.numpy()
    out["g15_std"] = sde3._std(tv).numpy()

    np.savez_compressed(os.path.join(HERE, "golden_ref2.npz"), **{k: np.asarray(v) for k, v in out.items()})
    tot = sum(np.asarray(v).nbytes for v in out.values())
    print(f"wrote {len(out)} arrays, {tot/1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
