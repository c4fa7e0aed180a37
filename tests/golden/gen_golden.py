#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (CPU, float32).

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/gen_golden.py

The reference is imported unmodified from /root/reference.  Its third-party packages that are not
installed here are replaced by minimal stand-ins that carry no arithmetic of the hot path:
  pytorch_lightning (LightningModule = nn.Module), hydra.utils.instantiate (imports `_target_` and
  calls it), omegaconf.open_dict, torch_ema (unused at inference), fast_bss_eval / pesq / matplotlib /
  huggingface_hub (unused here), torch.utils.cpp_extension.load (the CUDA ops are never called on
  CPU tensors: op/upfirdn2d.py:146-149 takes upfirdn2d_native), and
  torchaudio.transforms.{Spectrogram,InverseSpectrogram} = thin wrappers over torch.stft / torch.istft
  with window=hann_window(n_fft), win_length=n_fft, normalized=False, onesided=True — which is all the
  real classes do for power=None (torchaudio is not installed and cannot be fetched).
Inputs, weights and injected noise come from diffsep_amd.synth (a counter-based PRNG), so only the
reference OUTPUTS are stored; the tests regenerate the inputs.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "diffusion-separation_amd"))
from diffsep_amd import synth  # noqa: E402


# --------------------------------------------------------------------------- stand-ins
def _install_stubs():
    import torch.utils.cpp_extension as cpp_ext
    cpp_ext.load = lambda *a, **k: types.SimpleNamespace()

    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.ModelCheckpoint = type("ModelCheckpoint", (), {})
    pl.callbacks = cb
    sys.modules["pytorch_lightning"] = pl
    sys.modules["pytorch_lightning.callbacks"] = cb

    hydra = types.ModuleType("hydra")
    hutils = types.ModuleType("hydra.utils")

    def instantiate(cfg, *args, _recursive_=True, **kwargs):
        cfg = dict(cfg)
        target = cfg.pop("_target_")
        mod, name = target.rsplit(".", 1)
        cls = getattr(importlib.import_module(mod), name)
        cfg.update(kwargs)
        if _recursive_:
            cfg = {k: (instantiate(v) if isinstance(v, dict) and "_target_" in v else v) for k, v in cfg.items()}
        return cls(*args, **cfg)

    hutils.instantiate = instantiate
    hutils.to_absolute_path = lambda p: p
    hydra.utils = hutils
    sys.modules["hydra"] = hydra
    sys.modules["hydra.utils"] = hutils

    oc = types.ModuleType("omegaconf")
    oco = types.ModuleType("omegaconf.omegaconf")
    import contextlib
    oco.open_dict = lambda c: contextlib.nullcontext()
    oc.omegaconf = oco
    oc.OmegaConf = type("OmegaConf", (), {})
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.omegaconf"] = oco

    te = types.ModuleType("torch_ema")

    class ExponentialMovingAverage:
        def __init__(self, params, decay=0.0):
            self.collected_params = None

        def to(self, *a, **k):
            return self

        def store(self, *a):
            pass

        def copy_to(self, *a):
            pass

        def restore(self, *a):
            pass

    te.ExponentialMovingAverage = ExponentialMovingAverage
    sys.modules["torch_ema"] = te
    for name in ("fast_bss_eval", "pesq", "matplotlib", "matplotlib.pyplot"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["pesq"].pesq = lambda *a, **k: 0.0
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]

    ta = types.ModuleType("torchaudio")
    tt = types.ModuleType("torchaudio.transforms")

    class Spectrogram(torch.nn.Module):
        def __init__(self, n_fft=400, hop_length=None, power=2.0, center=True, pad_mode="reflect", **kw):
            super().__init__()
            assert power is None
            self.n_fft, self.hop, self.center, self.pad_mode = n_fft, hop_length, center, pad_mode
            self.register_buffer("window", torch.hann_window(n_fft), persistent=False)

        def forward(self, x):
            shp = x.shape
            X = torch.stft(x.reshape(-1, shp[-1]), self.n_fft, hop_length=self.hop, win_length=self.n_fft,
                           window=self.window, center=self.center, pad_mode=self.pad_mode, normalized=False,
                           onesided=True, return_complex=True)
            return X.reshape(shp[:-1] + X.shape[-2:])

    class InverseSpectrogram(torch.nn.Module):
        def __init__(self, n_fft=400, hop_length=None, center=True, pad_mode="reflect", **kw):
            super().__init__()
            self.n_fft, self.hop, self.center = n_fft, hop_length, center
            self.register_buffer("window", torch.hann_window(n_fft), persistent=False)

        def forward(self, X, length=None):
            shp = X.shape
            x = torch.istft(X.reshape(-1, shp[-2], shp[-1]), self.n_fft, hop_length=self.hop, win_length=self.n_fft,
                            window=self.window, center=self.center, normalized=False, onesided=True, length=length)
            return x.reshape(shp[:-2] + x.shape[-1:])

    tt.Spectrogram, tt.InverseSpectrogram = Spectrogram, InverseSpectrogram
    ta.transforms = tt
    ta.load = ta.save = None
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.transforms"] = tt


class AD(dict):
    """attribute dict standing in for an OmegaConf node."""
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def copy(self):
        return AD(self)


def to_ad(d):
    return AD({k: to_ad(v) if isinstance(v, dict) else v for k, v in d.items()})


def model_config(nf, S, spec_factor=0.33):
    """config/model/default.yaml restated as data (values only)."""
    return to_ad(dict(model=dict(
        n_speakers=S, fs=8000, t_eps=0.03, t_rev_init=0.03, ema_decay=0.999, valid_max_sep_batches=1,
        time_sampling_strategy="uniform", train_source_order="power", init_hack=False,
        score_model=dict(_target_="models.score_models.ScoreModelNCSNpp", num_sources=S,
                         stft_args=dict(n_fft=510, hop_length=128, center=True, pad_mode="constant"),
                         backbone_args=dict(_target_="models.ncsnpp.NCSNpp", nf=nf),
                         transform="exponent", spec_abs_exponent=0.5, spec_factor=spec_factor,
                         spec_trans_learnable=False),
        sde=dict(_target_="sdes.sdes.MixSDE", ndim=S, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, N=30),
        sampler=dict(N=30, snr=0.5, corrector_steps=1),
        loss=dict(_target_="torch.nn.MSELoss"),
        val_losses={},
        optimizer=dict(_target_="torch.optim.Adam", lr=1e-4, weight_decay=0.0),
    )))


def load_synth_weights(module, seed, prefix=""):
    sd = module.state_dict()
    table = []
    for k, v in sd.items():
        if k.endswith("window"):
            continue
        name = k[len(prefix):] if prefix and k.startswith(prefix) else k
        table.append((name, tuple(v.shape)))
        sd[k] = torch.from_numpy(synth.synth_param(name, v.shape, seed))
    module.load_state_dict(sd)
    return table


class InjectedNoise:
    """Replace torch.randn_like by a queue of synthetic draws while the reference sampler runs."""
    def __init__(self, draws):
        self.draws = list(draws)
        self.i = 0

    def __enter__(self):
        self.orig = torch.randn_like

        def fake(x, **kw):
            z = self.draws[self.i]
            self.i += 1
            assert z.shape == x.shape
            return z.to(x.dtype)
        torch.randn_like = fake
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def main():
    torch.set_grad_enabled(False)
    torch.manual_seed(0)
    _install_stubs()
    sys.path.insert(0, REF)
    from models.ncsnpp_utils import layerspp, up_or_down_sampling  # reference
    from models.score_models import ScoreModelNCSNpp  # reference
    import sdes as ref_sdes  # reference
    import pl_model as ref_pl  # reference
    import separate as ref_separate  # reference

    out = {}
    meta = {}

    # ---- G1 FIR resampling (the only reference-owned native op; CPU path = upfirdn2d_native)
    x = torch.from_numpy(synth.synth_noise("g1.x", (2, 8, 6, 10)))
    out["g1_up"] = up_or_down_sampling.upsample_2d(x, [1, 3, 3, 1], factor=2).numpy()
    out["g1_down"] = up_or_down_sampling.downsample_2d(x, [1, 3, 3, 1], factor=2).numpy()

    # ---- G4 ResnetBlockBigGANpp {plain, Cin != Cout, up, down}, G5 AttnBlockpp
    act = torch.nn.SiLU()
    temb = torch.from_numpy(synth.synth_noise("g4.temb", (2, 32)))
    for tag, kw, cin in (("plain", dict(), 16), ("widen", dict(out_ch=24), 16), ("up", dict(up=True), 16),
                         ("down", dict(down=True), 16)):
        blk = layerspp.ResnetBlockBigGANpp(act, cin, temb_dim=32, dropout=0.0, fir=True, fir_kernel=[1, 3, 3, 1],
                                           skip_rescale=True, init_scale=0.0, **kw)
        load_synth_weights(blk, 4, prefix="")
        xin = torch.from_numpy(synth.synth_noise("g4.x." + tag, (2, cin, 8, 12)))
        out["g4_" + tag] = blk(xin, temb).numpy()
    for tag, hw in (("16x16", (16, 16)), ("4x4", (4, 4))):
        ab = layerspp.AttnBlockpp(channels=16, skip_rescale=True, init_scale=0.0)
        load_synth_weights(ab, 5)
        xin = torch.from_numpy(synth.synth_noise("g5.x." + tag, (2, 16) + hw))
        out["g5_" + tag] = ab(xin).numpy()

    # ---- G6/G7 ScoreModelNCSNpp: pre/post process, frame counts, full forward (tiny backbone nf=16)
    S, nf = 2, 16
    cfg = model_config(nf, S)
    sm = ScoreModelNCSNpp(**{k: v for k, v in cfg.model.score_model.items() if k != "_target_"})
    table = load_synth_weights(sm.backbone, 7)
    meta["param_table_nf16_S2"] = [[n, list(s)] for n, s in table]
    sm.eval()
    frames = {}
    for T in (4000, 31999, 32000, 32001):
        xx = torch.from_numpy(synth.synth_noise(f"g6.x.{T}", (1, 3, T))) * 0.3
        spec, n_samples, n_pad = sm.pre_process(xx)
        frames[str(T)] = [int(spec.shape[-1] - n_pad), int(spec.shape[-1])]
        if T == 4000:
            out["g6_pre_4000"] = spec.numpy()
            yy = torch.from_numpy(synth.synth_noise("g6.y.4000", (1, 4, 256, 64))) * 0.2
            out["g6_post_4000"] = sm.post_process(yy, n_samples, n_pad).numpy()
        else:  # first / last valid frame + one interior frame: pins the frame indexing at the boundaries
            F = spec.shape[-1] - n_pad
            out[f"g6_pre_{T}_frames"] = spec[..., [0, 1, F // 2, F - 2, F - 1]].numpy()
    meta["frames"] = frames
    T = 4000
    xt = torch.from_numpy(synth.synth_noise("g7.xt", (2, S, T))) * 0.5
    mix = torch.from_numpy(synth.synth_noise("g7.mix", (2, 1, T))) * 0.5
    tt = torch.tensor([0.7, 0.05], dtype=torch.float32)
    out["g7_score"] = sm(xt, tt, mix).numpy()
    xb = torch.from_numpy(synth.synth_noise("g7.xb", (1, 6, 256, 64))) * 0.3
    out["g7_backbone"] = sm.backbone(xb, torch.tensor([0.4])).numpy()

    # ---- 3-source variant (Cin 8 / Cout 6) builds and runs
    cfg3 = model_config(16, 3)
    sm3 = ScoreModelNCSNpp(**{k: v for k, v in cfg3.model.score_model.items() if k != "_target_"})
    table3 = load_synth_weights(sm3.backbone, 7)
    meta["param_table_nf16_S3"] = [[n, list(s)] for n, s in table3]
    xt3 = torch.from_numpy(synth.synth_noise("g7.xt3", (1, 3, T))) * 0.5
    mix3 = torch.from_numpy(synth.synth_noise("g7.mix3", (1, 1, T))) * 0.5
    out["g7_score_S3"] = sm3(xt3, torch.tensor([0.3]), mix3).numpy()

    # ---- parameter tables of the real sizes (names + shapes only)
    for nfx in (64, 128):
        from models.ncsnpp import NCSNpp
        net = NCSNpp(nf=nfx, num_channels_in=6, num_channels_out=4)
        meta[f"param_table_nf{nfx}_S2"] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
        meta[f"param_count_nf{nfx}_S2"] = int(sum(v.numel() for v in net.state_dict().values()))

    # ---- G8 SDE scalar tables (float32 tensors exactly as the sampler computes them)
    sde = ref_sdes.sdes.MixSDE(ndim=2, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, N=30)
    ts = torch.linspace(sde.T, 0.03, 30)
    out["g8_timesteps"] = ts.numpy()
    out["g8_std"] = sde._std(ts).numpy()
    ev1, ev2 = sde._cov_eigval(ts)
    out["g8_ev1"], out["g8_ev2"] = ev1.numpy(), ev2.numpy()
    _, diff = sde.sde(torch.zeros(30, 2, 4), ts, None)
    out["g8_diffusion"] = diff.numpy()
    out["g8_timesteps_N7"] = torch.linspace(1.0, 0.03, 7).numpy()
    out["g8_timesteps_N200"] = torch.linspace(1.0, 0.03, 200).numpy()

    # ---- G9 the full PC sampler through DiffSepModel.get_pc_sampler (N=3, 1 corrector step)
    model = ref_pl.DiffSepModel(cfg)
    load_synth_weights(model.score_model.backbone, 7)
    model.eval()
    B = 2
    mixb = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    (mix_norm, _), mean, std = model.normalize_batch((mixb, None))
    out["g10_mix_norm"] = mix_norm.numpy()
    N, cs = 3, 1
    draws = [torch.from_numpy(synth.synth_noise(f"g9.z{i}", (B, S, T))) for i in range(1 + N * (cs + 1))]
    with InjectedNoise(draws) as inj:
        sampler = model.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=N, denoise=True, intermediate=False,
                                       corrector_steps=cs, snr=0.5, schedule=None)
        sep, nfe = sampler()
        assert inj.i == len(draws)
    out["g9_sep"] = sep.numpy()
    meta["g9_nfe"] = int(nfe)
    with InjectedNoise(draws) as inj:
        sampler = model.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=N, denoise=False, intermediate=False,
                                       corrector_steps=cs, snr=0.5, schedule=None)
        out["g9_sep_nodenoise"] = sampler()[0].numpy()
    # one predictor and one corrector update in isolation
    pred = ref_sdes.PredictorRegistry.get_by_name("reverse_diffusion")(sde.copy(), model)
    sdec = sde.copy(); sdec.N = N
    pred = ref_sdes.PredictorRegistry.get_by_name("reverse_diffusion")(sdec, model)
    corr = ref_sdes.CorrectorRegistry.get_by_name("ald2")(sdec, model, snr=0.5, n_steps=1)
    x0 = torch.from_numpy(synth.synth_noise("g9.x0", (B, S, T))) * 0.5
    tv = torch.tensor([0.8, 0.2])
    with InjectedNoise([draws[1]]):
        xc, xcm = corr.update_fn(x0, tv, mix_norm)
    with InjectedNoise([draws[2]]):
        xp, xpm = pred.update_fn(x0, tv, mix_norm)
    out["g9_corr_x"], out["g9_corr_mean"] = xc.numpy(), xcm.numpy()
    out["g9_pred_x"], out["g9_pred_mean"] = xp.numpy(), xpm.numpy()
    with InjectedNoise([draws[0]]):
        out["g9_prior"] = sdec.prior_sampling(mix_norm.shape, mix_norm).numpy()

    # ---- G10 separate.separate(): normalize -> sampler -> scale_output, per utterance (batch dim added inside)
    kw = dict(N=2, denoise=True, intermediate=False, corrector_steps=1, snr=0.5, schedule=None)
    d1 = [torch.from_numpy(synth.synth_noise(f"g10.z{i}", (1, S, T))) for i in range(1 + 2 * 2)]
    with InjectedNoise(d1):
        out["g10_separate"] = ref_separate.separate(mixb[0], model, kw, "cpu").numpy()
    out["g10_scale_output"] = ref_separate.scale_output(mixb, sep).numpy()

    # ---- G11 PriorMixSDE (speech enhancement, config/model/nr.yaml): sigma_mix, prior, isolated updates, full sampler
    cfgp = model_config(nf, S)
    cfgp["model"]["sde"] = AD(_target_="sdes.sdes.PriorMixSDE", ndim=S, d_lambda=2.0, sigma_min=0.05, sigma_max=0.5, N=30)
    modelp = ref_pl.DiffSepModel(cfgp)
    load_synth_weights(modelp.score_model.backbone, 7)
    modelp.eval()
    psde = modelp.sde.copy(); psde.N = N
    out["g11_sigma_mix"] = psde._std_sigma_mix(mix_norm).numpy()
    with InjectedNoise([draws[0]]):
        out["g11_prior"] = psde.prior_sampling(mix_norm.shape, mix_norm).numpy()
    predp = ref_sdes.PredictorRegistry.get_by_name("reverse_diffusion")(psde, modelp)
    corrp = ref_sdes.CorrectorRegistry.get_by_name("ald2")(psde, modelp, snr=0.5, n_steps=1)
    with InjectedNoise([draws[1]]):
        xc, xcm = corrp.update_fn(x0, tv, mix_norm)
    with InjectedNoise([draws[2]]):
        xp, xpm = predp.update_fn(x0, tv, mix_norm)
    out["g11_corr_x"], out["g11_corr_mean"] = xc.numpy(), xcm.numpy()
    out["g11_pred_x"], out["g11_pred_mean"] = xp.numpy(), xpm.numpy()
    with InjectedNoise(draws) as inj:
        sampler = modelp.get_pc_sampler("reverse_diffusion", "ald2", mix_norm, N=N, denoise=True, intermediate=False,
                                        corrector_steps=cs, snr=0.5, schedule=None)
        out["g11_sep"] = sampler()[0].numpy()
        assert inj.i == len(draws)

    # ---- G12 remaining sampler surface: euler_maruyama predictor, ald / langevin correctors,
    # scheduled sampler, probability-flow discretisation of the reverse SDE
    em = ref_sdes.PredictorRegistry.get_by_name("euler_maruyama")(sdec, model)
    with InjectedNoise([draws[2]]):
        xe, xem = em.update_fn(x0, tv, mix_norm)
    out["g12_em_x"], out["g12_em_mean"] = xe.numpy(), xem.numpy()
    ald = ref_sdes.CorrectorRegistry.get_by_name("ald")(sdec, model, snr=0.5, n_steps=1)
    with InjectedNoise([draws[1]]):
        xa, xam = ald.update_fn(x0, tv, mix_norm)
    out["g12_ald_x"], out["g12_ald_mean"] = xa.numpy(), xam.numpy()
    lan = ref_sdes.CorrectorRegistry.get_by_name("langevin")(sdec, model, snr=0.5, n_steps=1)
    with InjectedNoise([draws[1]]):
        xl, xlm = lan.update_fn(x0, tv, mix_norm)
    out["g12_langevin_x"], out["g12_langevin_mean"] = xl.numpy(), xlm.numpy()
    rs_pf = sdec.reverse(model, probability_flow=True)
    fpf, gpf = rs_pf.discretize(x0, tv, mix_norm)
    out["g12_pflow_mean"] = (x0 - fpf).numpy()
    meta["g12_pflow_G_is_zero"] = bool((gpf == 0).all())
    with InjectedNoise(draws) as inj:
        sampler = model.get_pc_sampler("euler_maruyama", "ald", mix_norm, N=N, denoise=True, intermediate=False,
                                       corrector_steps=cs, snr=0.5, schedule="log")
        out["g12_sep_em_ald_log"] = sampler()[0].numpy()
    with InjectedNoise(draws) as inj:
        sampler = model.get_pc_sampler("reverse_diffusion", "langevin", mix_norm, N=N, denoise=True, intermediate=False,
                                       corrector_steps=cs, snr=0.5, schedule=None)
        out["g12_sep_rd_langevin"] = sampler()[0].numpy()

    np.savez_compressed(os.path.join(HERE, "golden_ref.npz"), **{k: np.asarray(v) for k, v in out.items()})
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f)
    tot = sum(np.asarray(v).nbytes for v in out.values())
    print(f"wrote {len(out)} arrays, {tot/1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
