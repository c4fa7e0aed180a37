"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every
symbol include/diffsep_hip.h declares, and its parameter table / frame arithmetic (pure host code)
agree with the oracle and therefore with the reference.  No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import diffsep_oracle as O
from diffsep_amd import _lib
from diffsep_amd.engine import pack_state_dict, param_table
from diffsep_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_library_loads_and_exports_header_symbols(kind):
    # both builds of the library (16-bit tensors as bfloat16 / as IEEE half precision) export the same C-ABI
    l = _lib.lib(kind)
    hdr = open(os.path.join(ROOT, "include", "diffsep_hip.h")).read()
    declared = set(re.findall(r"\b(diffsep_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert b"gfx950" in l.diffsep_version()
    assert (b"storage f16" in l.diffsep_version()) == (kind == "f16")


@pytest.mark.parametrize("kind", ["bf16", "f16"])
def test_no_packed_fp32_valu_in_any_kernel(kind, tmp_path):
    """The wrong GroupNorm sums under co-resident kernels (profiles/experiments/README.md, "Streams") came from a site
    where a scalar VALU result fed a packed-FP32 instruction.  The build bans the instruction class (Makefile:
    -fno-slp-vectorize, -fno-vectorize for sde.hip); this disassembles every gfx950 code object of the library and
    checks that none is left."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain is not installed")
    so = shutil.copy(_lib.LIB_PATHS[kind], tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(p for p in os.listdir(tmp_path) if p.endswith("gfx950"))
    assert len(objs) >= 8  # one code object per .hip source
    kernels = 0
    for o in objs:
        asm = subprocess.run([objdump, "-d", str(tmp_path / o)], check=True, capture_output=True, text=True).stdout
        kernels += len(re.findall(r"^[0-9a-f]+ <_Z\w+>:", asm, re.M))
        hits = re.findall(r"v_pk_(?:fma|mul|add)_f32", asm)
        assert not hits, f"{len(hits)} packed-FP32 instructions in {o}"
    assert kernels > 100


@pytest.mark.parametrize("nf,S", [(16, 2), (16, 3), (64, 2), (128, 2)])
def test_param_table_is_reference_state_dict_order(golden, nf, S):
    _, meta = golden
    cfg = _lib.model_config(nf=nf, num_sources=S)
    mine = [[n, list(s)] for n, s, _ in param_table(cfg)]
    assert mine == meta[f"param_table_nf{nf}_S{S}"]
    assert mine == [[n, list(s)] for n, s in O.param_table(O.default_config(nf, S))]
    offs = [o for _, _, o in param_table(cfg)]
    sizes = [int(np.prod(s)) for _, s, _ in param_table(cfg)]
    assert offs == list(np.cumsum([0] + sizes[:-1]))
    assert _lib.lib().diffsep_param_total(C.byref(cfg)) == sum(sizes)


def test_frame_arithmetic_is_bit_exact(golden):
    _, meta = golden
    cfg = _lib.model_config()
    l = _lib.lib()
    for T, (F, W) in meta["frames"].items():
        assert l.diffsep_num_frames(C.byref(cfg), int(T)) == F
        assert l.diffsep_padded_frames(C.byref(cfg), int(T)) == W
    for T in (1, 127, 128, 129, 382, 383, 48000, 64000, 64001):
        F = 1 + (T + 382) // 128
        assert l.diffsep_num_frames(C.byref(cfg), T) == F
        assert l.diffsep_padded_frames(C.byref(cfg), T) == 64 * ((F + 63) // 64)


def test_pack_state_dict_roundtrip_and_errors():
    cfg = _lib.model_config(nf=16)
    table = [(n, s) for n, s, _ in param_table(cfg)]
    sd = synth.synth_state_dict(table, 3)
    blob = pack_state_dict(cfg, sd)
    for n, s, off in param_table(cfg):
        assert np.array_equal(blob[off:off + sd[n].size], sd[n].reshape(-1))
    bad = dict(sd)
    bad.pop("all_modules.3.weight")
    with pytest.raises(KeyError):
        pack_state_dict(cfg, bad)
    bad = dict(sd)
    bad["all_modules.3.weight"] = bad["all_modules.3.weight"][:, :, :1]
    with pytest.raises(ValueError):
        pack_state_dict(cfg, bad)


def test_bad_config_reports_error_without_gpu():
    cfg = _lib.model_config(nf=12)  # not a multiple of 8
    assert _lib.lib().diffsep_param_count(C.byref(cfg)) < 0
    assert b"nf" in _lib.lib().diffsep_last_error()


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffsep_amd.engine import Engine
    cfg = _lib.model_config(nf=16)
    with pytest.raises(_lib.DiffsepError):
        Engine(cfg, np.zeros(10, np.float32))


def test_ctypes_structs_mirror_the_header():
    # field order and C types of every struct of the boundary: a silent mismatch would shift every argument after it
    hdr = open(os.path.join(ROOT, "include", "diffsep_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    ctype = {"int32_t": C.c_int32, "float": C.c_float, "const int64_t*": C.POINTER(C.c_int64),
             "const uint64_t*": C.POINTER(C.c_uint64), "diffsep_engine*": C.c_void_p}
    want = {"diffsep_model_config": _lib.ModelConfig, "diffsep_sde_config": _lib.SdeConfig,
            "diffsep_sampler_config": _lib.SamplerConfig, "diffsep_sampler_ext": _lib.SamplerExt}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(\w+);", hdr, flags=re.S):
        if name not in want:
            continue
        fields = []
        for decl in [d.strip() for d in body.split(";") if d.strip()]:
            m = re.match(r"(.+?[\s*])([\w\[\], ]+)$", decl)
            typ, names = m.group(1).strip().replace(" *", "*"), m.group(2)
            for nm in [n.strip() for n in names.split(",")]:
                arr = re.match(r"(\w+)\[(\d+)\]", nm)
                fields.append((arr.group(1), ctype[typ] * int(arr.group(2))) if arr else (nm, ctype[typ]))
        mine = [(n, t) for n, t in want.pop(name)._fields_]
        assert [n for n, _ in mine] == [n for n, _ in fields], name
        assert all(C.sizeof(a) == C.sizeof(b) for (_, a), (_, b) in zip(mine, fields)), name
    assert not want, f"structs not found in the header: {list(want)}"


def test_dtype_codes_match_the_header():
    hdr = open(os.path.join(ROOT, "include", "diffsep_hip.h")).read()
    codes = {n: int(v) for n, v in re.findall(r"#define (DIFFSEP_(?:F32|BF16|F32_SPLIT))\s+(\d+)", hdr)}
    assert codes == {"DIFFSEP_F32": _lib.F32, "DIFFSEP_BF16": _lib.BF16, "DIFFSEP_F32_SPLIT": _lib.F32_SPLIT}
    cfg = _lib.model_config(nf=16, dtype=_lib.F32_SPLIT)  # the parameter table does not depend on the precision mode
    assert [n for n, _, _ in param_table(cfg)] == [n for n, _, _ in param_table(_lib.model_config(nf=16))]
