"""ctypes binding of libdiffsep_hip.so (include/diffsep_hip.h).  No fallback: if the HIP library
is missing or a call fails, this raises — the product path never routes around the GPU code."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same sources: the 16-bit storage format of dtype code 1 is bfloat16 in libdiffsep_hip.so and IEEE half
# precision in libdiffsep_hip_f16.so (csrc/Makefile, -DDS_HALF_F16).  DIFFSEP_LIB overrides the first one (profiling builds).
LIB_PATH = os.environ.get("DIFFSEP_LIB", os.path.join(os.path.dirname(_HERE), "libdiffsep_hip.so"))
LIB_PATHS = {"bf16": LIB_PATH, "f16": os.environ.get("DIFFSEP_LIB_F16", os.path.join(os.path.dirname(_HERE), "libdiffsep_hip_f16.so"))}

F32, BF16 = 0, 1
F32_SPLIT = 2  # fp32 tensors; matrix products as 3 bf16 MFMAs on hi / lo halves (include/diffsep_hip.h)
F16 = 3        # Python-side only: dtype code 1 (16-bit storage) of the half-precision build
SDE_MIX, SDE_PRIORMIX = 0, 1
PRED_REVERSE_DIFFUSION, PRED_EULER_MARUYAMA, PRED_NONE = 0, 1, 2
CORR_ALD2, CORR_NONE, CORR_ALD, CORR_LANGEVIN = 0, 1, 2, 3


class ModelConfig(C.Structure):
    _fields_ = [("nf", C.c_int32), ("num_sources", C.c_int32), ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * 8),
                ("num_res_blocks", C.c_int32), ("attn_resolution", C.c_int32), ("n_fft", C.c_int32),
                ("hop", C.c_int32), ("spec_abs_exponent", C.c_float), ("spec_factor", C.c_float),
                ("dtype", C.c_int32)]


class SdeConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ndim", C.c_int32), ("d_lambda", C.c_float), ("sigma_min", C.c_float),
                ("sigma_max", C.c_float), ("avg_len", C.c_int32)]


class SamplerConfig(C.Structure):
    _fields_ = [("N", C.c_int32), ("corrector_steps", C.c_int32), ("snr", C.c_float), ("eps", C.c_float),
                ("denoise", C.c_int32), ("predictor", C.c_int32), ("corrector", C.c_int32)]


class SamplerExt(C.Structure):
    _fields_ = [("lengths_host", C.POINTER(C.c_int64)), ("seeds_host", C.POINTER(C.c_uint64)),
                ("tail_engine", C.c_void_p), ("tail_steps", C.c_int32), ("head_steps", C.c_int32)]


class ProfRecord(C.Structure):  # diffsep_prof_record
    _fields_ = [("kernel", C.c_char * 128), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("Cout", C.c_int32), ("taps", C.c_int32), ("skip_cin", C.c_int32), ("has_res", C.c_int32),
                ("cls", C.c_int32), ("_pad", C.c_int32), ("flops", C.c_double), ("bytes", C.c_double), ("ms", C.c_double)]


class DiffsepError(RuntimeError):
    pass


_lib = None

_P, _I, _L, _F, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64
_SIGS = {
    "diffsep_last_error": (C.c_char_p, []),
    "diffsep_version": (C.c_char_p, []),
    "diffsep_param_count": (_I, [C.POINTER(ModelConfig)]),
    "diffsep_param_info": (_I, [C.POINTER(ModelConfig), _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_I),
                                C.POINTER(_L)]),
    "diffsep_param_total": (_L, [C.POINTER(ModelConfig)]),
    "diffsep_engine_create": (_I, [C.POINTER(ModelConfig), _P, _L, C.POINTER(_P)]),
    "diffsep_engine_destroy": (None, [_P]),
    "diffsep_engine_device_bytes": (_L, [_P]),
    "diffsep_engine_reserve": (_I, [_P, _I, _L, _P]),
    "diffsep_engine_debug_absmax": (_I, [_P, C.POINTER(C.c_double), _I, C.POINTER(_I)]),
    "diffsep_engine_debug_arena": (_I, [_P, C.POINTER(_P), C.POINTER(_L), C.POINTER(_L)]),
    "diffsep_num_frames": (_I, [C.POINTER(ModelConfig), _L]),
    "diffsep_padded_frames": (_I, [C.POINTER(ModelConfig), _L]),
    "diffsep_score_forward": (_I, [_P, _P, _P, _P, _P, _I, _L, _P]),
    "diffsep_backbone_forward": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "diffsep_pc_sample": (_I, [_P, C.POINTER(SdeConfig), C.POINTER(SamplerConfig), _P, _P, _I, _L, _P, _U64, _P,
                               C.POINTER(_I), _P]),
    "diffsep_pc_sample_ex": (_I, [_P, C.POINTER(SdeConfig), C.POINTER(SamplerConfig), C.POINTER(SamplerExt), _P, _P, _I,
                                  _L, _P, _U64, _P, C.POINTER(_I), _P]),
    "diffsep_engine_set_graph": (_I, [_P, _I]),
    "diffsep_set_option": (_I, [C.c_char_p, _L]),
    "diffsep_engine_set_option": (_I, [_P, C.c_char_p, _L]),
    "diffsep_engine_get_option": (_L, [_P, C.c_char_p]),
    "diffsep_engine_profile_begin": (_I, [_P]),
    "diffsep_engine_profile_end": (_I, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_L),
                                        C.POINTER(C.c_double)]),
    "diffsep_engine_profile_end_n": (_I, [_P, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_L),
                                          C.POINTER(C.c_double), C.POINTER(_I)]),
    "diffsep_num_kernel_classes": (_I, []),
    "diffsep_engine_profile_records": (_I, [_P, C.POINTER(ProfRecord), _I, C.POINTER(_I)]),
    "diffsep_time_embedding": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _L, _P]),
    "diffsep_upfirdn2d": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "diffsep_groupnorm_act": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _P, _L, _P]),
    "diffsep_conv2d": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "diffsep_groupnorm_stats": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P, _L, _P]),
    "diffsep_conv2d_fused": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F,
                                  _I, _P, _I, _P, _P, _P, _P, _I, _P]),
    "diffsep_conv2d_chunk": (_I, [_I, _I]),
    "diffsep_conv3x3_streamed": (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P,
                                      _P, _P]),
    "diffsep_frag_index": (_L, [_I, _I, _I, _I, _I]),
    "diffsep_frag_index_split": (_L, [_I, _I, _I, _I, _I, _I]),
    "diffsep_resblock_forward": (_I, [_I, _I, _I, _I, _I, _I, _P, _L, _P, _P, _P, _I, _I, _I, _P]),
    "diffsep_attnblock_forward": (_I, [_I, _I, _P, _L, _P, _P, _I, _I, _I, _P]),
    "diffsep_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "diffsep_stft_pack": (_I, [_P, _P, _P, _I, _I, _L, _I, _I, _F, _F, _I, _I, _I, _I, _P, _L, _P]),
    "diffsep_istft_unpack": (_I, [_P, _P, _I, _I, _L, _I, _I, _F, _F, _I, _I, _I, _P, _L, _P]),
    "diffsep_sde_sigma_mix": (_I, [_P, _P, _I, _L, _I, _P]),
    "diffsep_sde_prior": (_I, [C.POINTER(SdeConfig), _P, _P, _P, _I, _I, _L, _P, _P]),
    "diffsep_sde_corrector_update": (_I, [C.POINTER(SdeConfig), _F, _P, _P, _P, _P, _P, _P, _I, _I, _L, _P, _I, _P]),
    "diffsep_sde_predictor_update": (_I, [C.POINTER(SdeConfig), _I, _P, _P, _P, _P, _P, _P, _I, _I, _L, _P, _I, _P]),
    "diffsep_sde_coefficients": (_I, [C.POINTER(SdeConfig), _P, _P, _P, _P, _P, _I, _I, _L, _F, _F, _P]),
    "diffsep_sde_mean": (_I, [C.POINTER(SdeConfig), _P, _P, _P, _I, _I, _L, _P]),
    "diffsep_sde_std": (_I, [C.POINTER(SdeConfig), _P, _P, _P, _I, _I, _L, _P]),
    "diffsep_sde_mult_std": (_I, [_P, _P, _P, _I, _I, _L, _I, _P]),
    "diffsep_sde_reverse_drift": (_I, [_P, _P, _P, _P, _I, _L, _I, _I, _P]),
    "diffsep_sde_langevin_update": (_I, [_F, _P, _P, _P, _P, _P, _I, _L, _P, _L, _P]),
    "diffsep_normalize_batch": (_I, [_P, _P, _P, _P, _I, _L, _P]),
    "diffsep_scale_output": (_I, [_P, _P, _I, _I, _L, _P]),
    "diffsep_gram": (_I, [_P, _P, _P, _I, _I, _L, _P]),
    "diffsep_randn": (_I, [_P, _L, _U64, _U64, _P]),
    "diffsep_randn_batch": (_I, [_P, _I, _I, _L, _P, _P, _U64, _P]),
    "diffsep_convert": (_I, [_P, _P, _L, _I, _I, _P]),
}
EXPORTS = tuple(_SIGS.keys())


def lib(kind="bf16"):
    """Load (once) and return the HIP library of a storage variant ("bf16": libdiffsep_hip.so, "f16":
    libdiffsep_hip_f16.so); raises if it has not been built."""
    global _lib
    if _lib is None:
        _lib = {}
    if kind not in _lib:
        path = LIB_PATHS[kind]
        if not os.path.exists(path):
            raise DiffsepError(f"{path} not found: build it with __graft_entry__.build() or "
                               "`make -C diffusion-separation_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
        # On a GPU box torch must have brought the HIP runtime up BEFORE this library is loaded: loaded first (e.g.
        # __graft_entry__.build() followed by smoke() in one process), its device calls then failed with "no
        # ROCm-capable device is detected" (measured; the library and torch share one libamdhip64).  No-op without a GPU.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
        l = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib[kind] = l
    return _lib[kind]


def check(rc, l=None):
    if rc != 0:
        raise DiffsepError((l or lib()).diffsep_last_error().decode("utf-8", "replace"))


def half_kind(dtype_code):
    """storage variant of a dtype code: F16 lives in the half-precision build, everything else in the default one"""
    return "f16" if dtype_code == F16 else "bf16"


def model_config(nf=64, num_sources=2, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolution=16,
                 n_fft=510, hop=128, spec_abs_exponent=0.5, spec_factor=0.33, dtype=F32):
    c = ModelConfig()
    c.nf, c.num_sources, c.n_levels = nf, num_sources, len(ch_mult)
    for i, m in enumerate(ch_mult):
        c.ch_mult[i] = m
    c.num_res_blocks, c.attn_resolution, c.n_fft, c.hop = num_res_blocks, attn_resolution, n_fft, hop
    c.spec_abs_exponent, c.spec_factor, c.dtype = spec_abs_exponent, spec_factor, dtype
    return c
