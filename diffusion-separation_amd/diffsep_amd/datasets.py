"""On-disk data contracts of the evaluation path (host side, no torchaudio/lightning dependency).

* WSJ0_mix        <root>/{2,3}speakers/wav{8,16}k/{min,max}/{tr,cv,tt}/{mix,s1,s2[,s3]}/*.wav -> (mix [1,T], tgt [S,T])
                  (reference datasets/wsj0_mix.py:24-92; same ctor arguments, same errors, same sorted file order)
* NoisyDataset    <root>/{train,test}/{noisy,clean}/*.wav -> (noisy [1,T], [clean, noisy - clean] [2,T])
                  (reference datasets/vctk_demand.py:22-79)
* max_collator    centre-pads every tensor of a batch to the longest item (datasets/wsj0_mix.py:95-111)
"""
import os
import random
from pathlib import Path

import torch

from . import wavio

split_map = {"test": "tt", "val": "cv", "train": "tr", "libri2mix_test": "test"}


class WSJ0_mix(torch.utils.data.Dataset):
    def __init__(self, path, n_spkr=2, fs=16000, cut="max", split="train", max_len_s=None, max_n_samples=None,
                 mix_dir="mix"):
        super().__init__()
        if fs not in [8000, 16000]:
            raise ValueError(f"The sampling frequency fs can be only 8000 or 16000 (passed {fs})")
        if n_spkr not in [2, 3]:
            raise ValueError(f"The number of speakers can only be 2 or 3 (passed {n_spkr})")
        if cut not in ["min", "max"]:
            raise ValueError(f"The cut parameter has to be 'min' or 'max' (passed {cut})")
        if split not in split_map:
            raise ValueError(f"The split parameter must be 'train', 'val', or 'test' (passed {split})")
        self.base_folder = Path(path).absolute()
        self.n_spkr, self.fs, self.cut = n_spkr, int(fs), cut
        self.max_len = int(self.fs * max_len_s) if max_len_s is not None else None
        self.path = self.base_folder / f"{n_spkr}speakers/wav{self.fs // 1000}k/{cut}/{split_map[split]}"
        self.path_mix = self.path / mix_dir
        self.path_src = [self.path / f"s{i + 1}" for i in range(n_spkr)]
        self.file_list = sorted(os.listdir(self.path_mix))
        if max_n_samples is not None:
            self.file_list = self.file_list[:max_n_samples]

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        name = self.file_list[idx]
        mix, _ = wavio.load(self.path_mix / name)
        tgt = torch.cat([wavio.load(p / name)[0] for p in self.path_src], dim=0)
        if self.max_len is not None and tgt.shape[-1] > self.max_len:
            p = int(torch.randint(0, tgt.shape[-1] - self.max_len, size=(1,)))  # random cut of the right size
            tgt, mix = tgt[..., p:p + self.max_len], mix[..., p:p + self.max_len]
        return mix, tgt


class NoisyDataset(torch.utils.data.Dataset):
    def __init__(self, audio_path, audio_len=4, fs=16000, augmentation=False, split="train"):
        if split not in ("test", "train"):
            raise ValueError(f"The split parameter must be 'train' or 'test' (passed {split})")
        root = Path(audio_path).absolute() / split
        self.noisy_path, self.clean_path = root / "noisy", root / "clean"
        self.file_list = os.listdir(self.noisy_path)  # directory order, as the reference (not sorted)
        self.audio_len, self.fs, self.aug, self.split = int(audio_len * fs), fs, augmentation, split

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        noisy, _ = wavio.load(self.noisy_path / self.file_list[idx])
        clean, _ = wavio.load(self.clean_path / self.file_list[idx])
        if self.split == "test":
            return noisy, torch.cat([clean, noisy - clean], dim=0)
        n = noisy.shape[-1]
        if n < self.audio_len:
            noisy = torch.tile(noisy, dims=(2,))[..., :self.audio_len]
            clean = torch.tile(clean, dims=(2,))[..., :self.audio_len]
        else:
            st = random.randint(0, n - self.audio_len)
            noisy, clean = noisy[..., st:st + self.audio_len], clean[..., st:st + self.audio_len]
        if self.aug:
            noise = noisy - clean
            noisy = noise[torch.randperm(clean.size(0))] + clean
        return noisy, torch.cat([clean, noisy - clean], dim=0)


def max_collator(batch):
    max_len = max(s[0].shape[-1] for s in batch)
    rows = []
    for row in batch:
        new = []
        for el in row:
            if isinstance(el, torch.Tensor):
                off = max_len - el.shape[-1]
                new.append(torch.nn.functional.pad(el, (off // 2, off - off // 2)))
        rows.append(tuple(new))
    return torch.utils.data.default_collate(rows)


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
