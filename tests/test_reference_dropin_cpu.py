"""INTEGRATION.md §1, tested where it can be tested: in the build container, against the REFERENCE's own LightningModule.

The reference's `pl_model.DiffSepModel` (imported unmodified from /root/reference, third-party packages replaced by the
stand-ins of tests/golden/gen_golden.py) is built with the `_target_` string a maintainer would put into
config/model/default.yaml:15 — `diffsep_amd.score_models.ScoreModelNCSNpp`; the SDE stays `sdes.sdes.MixSDE`, because the
reference's own correctors accept only their own SDE classes (sdes/correctors.py:100, an isinstance check) — and then
driven the way separate.py:36-48 drives it: load_from_checkpoint (strict) of a checkpoint WRITTEN BY THE REFERENCE model,
on_load_checkpoint (EMA), .eval() (EMA shadow -> parameters, pl_model.py:655-660), .train() (restore, :662-666).  What an
engine would be created from (`packed_blob`, and the blob a recording stand-in engine receives — engines need a GPU) must
be the EMA weights after .eval() and the raw weights after .train().

Skipped where /root/reference does not exist (the GPU box); tests/test_dropin_gpu.py runs the same cycle with a stand-in
parent module on the real engine.

torch_ema is not installed (no network): ExponentialMovingAverage below restates its published behaviour — shadow_params =
detached clones, store / copy_to / restore through `param.data.copy_`, state_dict / load_state_dict with the length check
— in the two variants that exist in the wild: 0.2 shadows only parameters with requires_grad (a checkpoint holds one tensor
fewer than there are parameters: the frozen Fourier projection), 0.3 shadows every parameter passed.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")


def _make_torch_ema(filter_requires_grad):
    te = types.ModuleType("torch_ema")

    class ExponentialMovingAverage:
        def __init__(self, parameters, decay=0.0, use_num_updates=True):
            if decay < 0.0 or decay > 1.0:
                raise ValueError("Decay must be between 0 and 1")
            self.decay, self.num_updates = decay, (0 if use_num_updates else None)
            self.shadow_params = [p.clone().detach() for p in self._pick(parameters)]
            self.collected_params = None

        @staticmethod
        def _pick(parameters):
            parameters = list(parameters)
            return [p for p in parameters if p.requires_grad] if filter_requires_grad else parameters

        def _get(self, parameters):
            parameters = self._pick(parameters)
            if len(parameters) != len(self.shadow_params):
                raise ValueError("Number of parameters passed as argument is different from number of shadow parameters "
                                 "maintained by this ExponentialMovingAverage")
            return parameters

        def store(self, parameters):
            self.collected_params = [p.clone() for p in self._get(parameters)]

        def copy_to(self, parameters):
            for s, p in zip(self.shadow_params, self._get(parameters)):
                p.data.copy_(s.data)

        def restore(self, parameters):
            if self.collected_params is None:
                raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
            for c, p in zip(self.collected_params, self._get(parameters)):
                p.data.copy_(c.data)
            self.collected_params = None  # (pl_model.py:663 tests this attribute)

        def to(self, device=None, dtype=None):
            self.shadow_params = [p.to(device=device, dtype=dtype) if p.is_floating_point() else p.to(device=device)
                                  for p in self.shadow_params]
            return self

        def state_dict(self):
            return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow_params,
                    "collected_params": self.collected_params}

        def load_state_dict(self, state):
            self.decay, self.num_updates = state["decay"], state["num_updates"]
            shadow = [p.clone() for p in state["shadow_params"]]
            if len(shadow) != len(self.shadow_params) or any(a.shape != b.shape for a, b in zip(shadow, self.shadow_params)):
                raise ValueError("Tried to `load_state_dict()` with the wrong number of parameters in the saved state.")
            self.shadow_params = shadow
            self.collected_params = state["collected_params"]

    te.ExponentialMovingAverage = ExponentialMovingAverage
    return te


@pytest.fixture(scope="module")
def harness():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "golden", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    sys.modules["gen_golden"] = gg  # (its attribute-dict config class is pickled into the checkpoint)
    spec.loader.exec_module(gg)
    saved = dict(sys.modules)
    saved_path = list(sys.path)
    import torch.utils.cpp_extension as cpp_ext
    saved_load = cpp_ext.load
    gg._install_stubs()
    # real torchaudio keeps the window as a PERSISTENT buffer: the reference's state dict carries
    # score_model.stft.window / score_model.stft_inv.window — make the stand-ins do the same
    tt = sys.modules["torchaudio.transforms"]
    for cls in (tt.Spectrogram, tt.InverseSpectrogram):
        orig = cls.__init__

        def init(self, *a, _orig=orig, **k):
            _orig(self, *a, **k)
            w = self.window
            del self._buffers["window"]
            self._non_persistent_buffers_set.discard("window")
            self.register_buffer("window", w, persistent=True)
        cls.__init__ = init
    sys.path.insert(0, REF)
    yield gg
    cpp_ext.load = saved_load
    sys.path[:] = saved_path
    for k in list(sys.modules):
        if k not in saved:
            del sys.modules[k]
    sys.modules.update(saved)


def _fresh_ref_pl(filter_requires_grad):
    """the reference's pl_model imported against one torch_ema variant"""
    sys.modules["torch_ema"] = _make_torch_ema(filter_requires_grad)
    for k in ("pl_model",):
        sys.modules.pop(k, None)
    import pl_model as ref_pl
    return ref_pl


@pytest.mark.parametrize("filter_requires_grad", [True, False], ids=["torch_ema_0.2", "torch_ema_0.3"])
@pytest.mark.parametrize("with_window_keys", [True, False])
def test_reference_lightning_module_holds_loads_and_ema_swaps_the_hip_score_model(harness, tmp_path, filter_requires_grad,
                                                                                  with_window_keys):
    gg = harness
    from diffsep_amd import synth
    from diffsep_amd.engine import pack_state_dict, param_table
    from diffsep_amd.score_models import ScoreModelNCSNpp as HipScoreModel
    import diffsep_amd.sdes as hip_sdes
    ref_pl = _fresh_ref_pl(filter_requires_grad)
    nf, S = 16, 2

    # ---- a checkpoint as the REFERENCE writes it: raw weights in state_dict, other weights in the EMA shadow
    cfg_ref = gg.model_config(nf, S)
    ref = ref_pl.DiffSepModel(cfg_ref)
    gg.load_synth_weights(ref.score_model.backbone, 1)
    names = [n for n, p in ref.score_model.named_parameters()]
    shadow_names = [n for n, p in ref.score_model.named_parameters() if p.requires_grad or not filter_requires_grad]
    assert len(ref.ema.shadow_params) == len(shadow_names) == (646 if filter_requires_grad else 647)
    ema_w = {n: torch.from_numpy(synth.synth_param(n[len("backbone."):], p.shape, 2))
             for n, p in ref.score_model.named_parameters()}
    ref.ema.shadow_params = [ema_w[n].clone() for n in shadow_names]
    ckpt = {"state_dict": {k: v.clone() for k, v in ref.state_dict().items()}, "hyper_parameters": {"config": None}}
    ref.on_save_checkpoint(ckpt)  # (pl_model.py:672-673: checkpoint["ema"] = self.ema.state_dict())
    assert ("score_model.stft.window" in ckpt["state_dict"]) and ("score_model.stft_inv.window" in ckpt["state_dict"])
    if not with_window_keys:
        ckpt["state_dict"] = {k: v for k, v in ckpt["state_dict"].items() if not k.endswith(".window")}
    raw_w = {k[len("score_model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("score_model.backbone.")}

    # ---- the drop-in: the reference's DiffSepModel built from a config that names the HIP classes
    cfg = gg.model_config(nf, S)
    cfg.model.score_model["_target_"] = "diffsep_amd.score_models.ScoreModelNCSNpp"
    cfg.model.score_model.backbone_args["_target_"] = "models.ncsnpp.NCSNpp"  # (left as it is in the yaml: ignored)
    ckpt["hyper_parameters"] = {"config": cfg}
    torch.save(ckpt, tmp_path / "ref.ckpt")

    # Lightning's load_from_checkpoint, restated: cls(**hyper_parameters); on_load_checkpoint(ckpt); load_state_dict(strict)
    ck = torch.load(tmp_path / "ref.ckpt", map_location="cpu", weights_only=False)
    model = ref_pl.DiffSepModel(**ck["hyper_parameters"])
    assert isinstance(model.score_model, HipScoreModel) and isinstance(model.score_model, torch.nn.Module)
    assert model.sde.N == 30 and model.t_max == 1.0
    # the parameter tree is the reference's: names, order, shapes, which ones require a gradient
    assert [n for n, _ in model.score_model.named_parameters()] == names
    assert [tuple(p.shape) for p in model.score_model.parameters()] == [tuple(p.shape) for p in ref.score_model.parameters()]
    assert [p.requires_grad for p in model.score_model.parameters()] == [p.requires_grad for p in ref.score_model.parameters()]
    assert len(list(model.parameters())) == 647 and len(model.ema.shadow_params) == len(shadow_names)
    assert set(model.state_dict().keys()) == set(ref.state_dict().keys())

    built = []

    class RecordingEngine:  # (an Engine needs a GPU; this shows what it would be created from)
        def __init__(self, cfg_c, blob, device=None, lib_kind=None):
            self.blob = blob.copy()
            built.append(self)

        def close(self):
            pass

    model.score_model._engine_factory = RecordingEngine
    model.on_load_checkpoint(ck)
    res = model.load_state_dict(ck["state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert not model._error_loading_ema

    def blob_of(weights):
        return pack_state_dict(model.score_model.cfg, {k[len("backbone."):]: v for k, v in weights.items()})

    sm = model.score_model
    np.testing.assert_array_equal(sm.packed_blob(), blob_of(raw_w))
    np.testing.assert_array_equal(sm.engine().blob, blob_of(raw_w))

    # separate.py:47-48: model.to(device); model.eval()  -> inference runs on the EMA weights (quirk Q4)
    model.to("cpu")
    model.eval()
    want = {k: (ema_w[k] if k in shadow_names else raw_w[k]) for k in raw_w}
    np.testing.assert_array_equal(sm.packed_blob(), blob_of(want))
    e_eval = sm.engine()
    np.testing.assert_array_equal(e_eval.blob, blob_of(want))
    assert sm.engine() is e_eval  # (nothing changed: no re-pack per call)
    assert not np.array_equal(blob_of(want), blob_of(raw_w))
    # a twin (the split-precision fallback / hybrid head of diffsep_amd.pl_model) sees the same swap
    np.testing.assert_array_equal(sm.twin("split").engine().blob, blob_of(want))

    # eval(no_ema=True) after train(): raw weights again (pl_model.py:662-666 restores)
    model.train()
    np.testing.assert_array_equal(sm.engine().blob, blob_of(raw_w))
    model.eval(no_ema=True)
    np.testing.assert_array_equal(sm.engine().blob, blob_of(raw_w))
    model.eval()
    np.testing.assert_array_equal(sm.engine().blob, blob_of(want))
    n_built = len(built)
    model.eval()  # (a second eval() stores the EMA weights and copies them onto themselves: same content, no rebuild)
    sm.engine()
    assert len(built) == n_built

    # (a) the reference's sampler factory with the reference's SDE and the model as score function: the Python loop of
    # sdes/__init__.py:166-188 with the HIP engine behind every score evaluation (construction only here: a score needs the GPU)
    y = torch.zeros(1, 1, 4000)
    sampler = model.get_pc_sampler("reverse_diffusion", "ald2", y, N=3, corrector_steps=1, snr=0.5)
    assert callable(sampler)

    # (b) the one-line switch `import sdes` -> `from diffsep_amd import sdes` in pl_model.py: the
    # reference's DiffSepModel.get_pc_sampler (pl_model.py:687-721) then builds the fused sampler — the whole loop as one engine
    # call — because the engine is found behind the LightningModule's score_model
    cfg_b = gg.model_config(nf, S)
    cfg_b.model.score_model["_target_"] = "diffsep_amd.score_models.ScoreModelNCSNpp"
    cfg_b.model.sde["_target_"] = "diffsep_amd.sdes.sdes.MixSDE"
    saved_sdes = ref_pl.sdes
    ref_pl.sdes = hip_sdes
    try:
        model_b = ref_pl.DiffSepModel(cfg_b)
        assert isinstance(model_b.sde, hip_sdes.MixSDE)
        calls = []

        class SamplingEngine(RecordingEngine):
            def pc_sample(self, y_, sde_cfg, **kw):
                calls.append((tuple(y_.shape), sde_cfg, kw))
                return torch.zeros(y_.shape[0], 2, y_.shape[2]), 0

        model_b.score_model._engine_factory = SamplingEngine
        model_b.on_load_checkpoint(ck)
        model_b.load_state_dict(ck["state_dict"], strict=True)
        model_b.eval()
        x, nfe = model_b.get_pc_sampler("reverse_diffusion", "ald2", y, N=3, corrector_steps=1, snr=0.5)()
        assert nfe == 6 and tuple(x.shape) == (1, 2, 4000) and len(calls) == 1
        assert calls[0][2]["N"] == 3 and calls[0][2]["eps"] == 0.03 and calls[0][2]["snr"] == 0.5
        np.testing.assert_array_equal(built[-1].blob, blob_of(want))  # (that engine was created from the EMA weights)
        # the HIP SDE in the reference's OWN sampler is refused by the reference's corrector (isinstance on its own classes)
        with pytest.raises(NotImplementedError):
            saved_sdes.get_pc_sampler("reverse_diffusion", "ald2", sde=model_b.sde, score_fn=model_b, y=y)
    finally:
        ref_pl.sdes = saved_sdes


def test_reference_config_with_a_trained_variant_the_engine_does_not_implement_is_refused(harness):
    gg = harness
    ref_pl = _fresh_ref_pl(True)
    cfg = gg.model_config(16, 2)
    cfg.model.score_model["_target_"] = "diffsep_amd.score_models.ScoreModelNCSNpp"
    cfg.model.score_model.backbone_args["progressive"] = "none"
    with pytest.raises(NotImplementedError):
        ref_pl.DiffSepModel(cfg)
