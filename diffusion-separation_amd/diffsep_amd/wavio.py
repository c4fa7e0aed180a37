"""Minimal RIFF/WAVE reader / writer (PCM16, PCM32, float32) — torchaudio is not part of the image.
load(path) -> (float32 tensor [channels, T] in [-1, 1], sample_rate), save(path, [C,T] tensor, sr)."""
import struct

import numpy as np
import torch


def load(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 1 and bits == 16:
        x = np.frombuffer(payload, "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(payload, "<i4").astype(np.float32) / 2147483648.0
    elif tag == 3 and bits == 32:
        x = np.frombuffer(payload, "<f4").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (tag {tag}, {bits} bits)")
    x = x[: (x.size // ch) * ch].reshape(-1, ch).T
    return torch.from_numpy(np.ascontiguousarray(x)), sr


def info(path):
    """(sample_rate, frames per channel) from the header only."""
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt = None
        while True:
            ch = f.read(8)
            if len(ch) < 8:
                raise ValueError(f"{path}: missing fmt/data chunk")
            cid, size = ch[:4], struct.unpack("<I", ch[4:])[0]
            if cid == b"fmt ":
                fmt = struct.unpack("<HHIIHH", f.read(size)[:16])
                if size & 1:
                    f.read(1)
            elif cid == b"data":
                if fmt is None:
                    raise ValueError(f"{path}: data chunk before fmt")
                return fmt[2], size // (fmt[1] * (fmt[5] // 8))
            else:
                f.seek(size + (size & 1), 1)


def save(path, wav, sr, bits=16):
    x = wav.detach().cpu().float().numpy() if isinstance(wav, torch.Tensor) else np.asarray(wav, np.float32)
    if x.ndim == 1:
        x = x[None]
    ch, n = x.shape
    inter = np.ascontiguousarray(x.T)
    if bits == 16:
        body = np.clip(np.round(inter * 32767.0), -32768, 32767).astype("<i2").tobytes()
        tag = 1
    else:
        body, tag, bits = inter.astype("<f4").tobytes(), 3, 32
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, tag, ch, sr,
                      sr * ch * bits // 8, ch * bits // 8, bits, b"data", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)
