"""Readers of the two evaluation corpora, inference only (host side; no torchaudio / lightning).

The contract is a directory layout -> `(mixture [1,T], targets [S,T])` float32 items:

* WSJ0-mix         `<root>/{2,3}speakers/wav{8,16}k/{min,max}/{tr,cv,tt}/{mix,s1..sS}/<name>.wav`; item = (mix, [s1..sS]);
                   names sorted (reference datasets/wsj0_mix.py:64-92)
* VoiceBank-DEMAND `<root>/{train,test}/{noisy,clean}/<name>.wav`; item = (noisy, [clean, noisy - clean]); directory
                   order (reference datasets/vctk_demand.py:33-61)

Both are one index class, `WavPairs` — a list of utterance names plus a mixture folder and the folders / rule the
targets come from — built by a layout resolver (`wsj0_mix`, `voicebank_demand`).  `WSJ0_mix` / `NoisyDataset` keep the
reference's constructor keywords so that its hydra `datamodule` entries resolve, but only the evaluation behaviour
exists: whole utterances, file order, `max_n_samples`.  The training-time behaviour of the reference classes (random
crops to `max_len_s` / `audio_len`, tiling of short files, noise shuffling) is out of scope
and asking for it raises.
"""
import os
from pathlib import Path

import torch

from . import wavio

_WSJ0_SPLITS = {"train": "tr", "val": "cv", "test": "tt", "libri2mix_test": "test"}


class WavPairs(torch.utils.data.Dataset):
    """names[i] -> (mixture [1,T], targets [S,T]).  `target_dirs`: one folder per target channel; `residual=True`
    appends mixture - sum(targets) as a last channel (the noise of an enhancement pair)."""

    def __init__(self, mix_dir, target_dirs, names, fs, residual=False):
        self.mix_dir, self.target_dirs = Path(mix_dir), [Path(d) for d in target_dirs]
        self.file_list, self.fs, self.residual = list(names), int(fs), bool(residual)

    def __len__(self):
        return len(self.file_list)

    def num_samples(self, idx):
        """length of item idx from the wav header alone (evaluate sorts / buckets by it without decoding audio)"""
        return wavio.info(self.mix_dir / self.file_list[idx])[1]

    def __getitem__(self, idx):
        name = self.file_list[idx]
        mix, sr = wavio.load(self.mix_dir / name)
        if sr != self.fs:
            raise ValueError(f"{self.mix_dir / name}: sample rate {sr}, the dataset was opened for {self.fs}")
        rows = [wavio.load(d / name)[0] for d in self.target_dirs]
        if self.residual:
            rows.append(mix - torch.stack(rows).sum(0))
        return mix, torch.cat(rows, dim=0)


def wsj0_mix(root, n_spkr=2, fs=16000, cut="max", split="test", max_n_samples=None, mix_dir="mix"):
    """Resolve the WSJ0-2mix / 3mix layout (pywsj0-mix output) to a WavPairs index."""
    if fs not in (8000, 16000):
        raise ValueError(f"The sampling frequency fs can be only 8000 or 16000 (passed {fs})")
    if n_spkr not in (2, 3):
        raise ValueError(f"The number of speakers can only be 2 or 3 (passed {n_spkr})")
    if cut not in ("min", "max"):
        raise ValueError(f"The cut parameter has to be 'min' or 'max' (passed {cut})")
    if split not in _WSJ0_SPLITS:
        raise ValueError(f"The split parameter must be 'train', 'val', or 'test' (passed {split})")
    base = Path(root).absolute() / f"{n_spkr}speakers" / f"wav{int(fs) // 1000}k" / cut / _WSJ0_SPLITS[split]
    names = sorted(os.listdir(base / mix_dir))
    if max_n_samples is not None:
        names = names[:max_n_samples]
    return WavPairs(base / mix_dir, [base / f"s{k}" for k in range(1, n_spkr + 1)], names, fs)


def voicebank_demand(root, fs=16000, split="test"):
    """Resolve the VoiceBank-DEMAND layout to a WavPairs index: targets = [clean, noisy - clean]."""
    if split not in ("test", "train"):
        raise ValueError(f"The split parameter must be 'train' or 'test' (passed {split})")
    base = Path(root).absolute() / split
    return WavPairs(base / "noisy", [base / "clean"], os.listdir(base / "noisy"), fs, residual=True)


def WSJ0_mix(path, n_spkr=2, fs=16000, cut="max", split="train", max_len_s=None, max_n_samples=None, mix_dir="mix"):
    """The reference's constructor keywords (datasets/wsj0_mix.py:24-62) over `wsj0_mix`."""
    if max_len_s is not None:
        raise NotImplementedError("max_len_s (random training crops) is not part of the inference path")
    return wsj0_mix(path, n_spkr=n_spkr, fs=fs, cut=cut, split=split, max_n_samples=max_n_samples, mix_dir=mix_dir)


def NoisyDataset(audio_path, audio_len=4, fs=16000, augmentation=False, split="train"):
    """The reference's constructor keywords and defaults (datasets/vctk_demand.py:21-28) over `voicebank_demand`.  Only
    whole utterances are served — what the reference does for split == "test" (`audio_len` is ignored there too,
    vctk_demand.py:59-61); its split == "train" items are random crops / tilings of audio_len seconds with shuffled
    noise, a training-time behaviour that is not part of the inference path: asking for it raises."""
    if augmentation:
        raise NotImplementedError("augmentation is a training-time feature; not part of the inference path")
    if split == "train":
        raise NotImplementedError("NoisyDataset(split='train') serves random training crops in the reference "
                                  "(vctk_demand.py:63-99): not part of the inference path; use split='test' or "
                                  "voicebank_demand(root, split='train') for whole utterances")
    return voicebank_demand(audio_path, fs=fs, split=split)


def pad_batch(items, side="center", to=None):
    """[(mix [1,T_i], tgt [S,T_i])] -> (mix [B,1,Tmax], tgt [B,S,Tmax], lengths [B]).  side="center" spreads the padding
    on both ends (left half rounded down) like the reference's max_collator (datasets/wsj0_mix.py:95-111); side="right"
    appends it (what the engine's mixed-length batches take: Engine.pc_sample(lengths=...)).  to: pad to this length
    instead of the longest item (Engine.bucket_length: one workspace plan per padded width)."""
    lengths = [int(m.shape[-1]) for m, _ in items]
    Tmax = max(lengths) if to is None else int(to)
    assert Tmax >= max(lengths)
    out_m, out_t = [], []
    for (m, t), n in zip(items, lengths):
        left = (Tmax - n) // 2 if side == "center" else 0
        out_m.append(torch.nn.functional.pad(m, (left, Tmax - n - left)))
        out_t.append(torch.nn.functional.pad(t, (left, Tmax - n - left)))
    return torch.stack(out_m), torch.stack(out_t), lengths


def max_collator(batch):
    m, t, _ = pad_batch(batch, side="center")
    return m, t


def summarize(results):
    """Mean of every record field over the utterances + their "number" (evaluate.py:148-161); None fields (metrics
    this build does not compute: pesq, stoi) are skipped."""
    keys, acc = [], {}
    for res in results:
        for k, v in res.items():
            if v is None:
                continue
            if k not in acc:
                keys.append(k)
                acc[k] = 0.0
            acc[k] += float(torch.as_tensor(v, dtype=torch.float64).mean())
    n = len(results)
    out = {k: acc[k] / max(n, 1) for k in keys}
    out["number"] = n
    return out
