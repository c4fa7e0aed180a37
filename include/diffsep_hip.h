/*
 * diffsep_hip.h — C-ABI of the MI355X-native reverse-diffusion separation engine.
 *
 * This is the drop-in boundary for the ONE hot path of fakufaku/diffusion-separation:
 *   separate.py / evaluate.py -> DiffSepModel.get_pc_sampler -> sdes.get_pc_sampler
 *   -> {ald2 corrector, reverse_diffusion predictor} -> ScoreModelNCSNpp (STFT -> NCSN++ -> iSTFT).
 * The reference has no C-ABI/FFI of its own on this path: its plug points are
 * Python call signatures.  Each entry point below cites the reference interface it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C types, raw DEVICE pointers (e.g. torch.Tensor.data_ptr()), sizes as int / int64_t;
 *   - every function returns 0 on success, non-zero on failure; diffsep_last_error() returns a
 *     thread-local message; no C++ exception crosses this boundary;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     asynchronous on it, the caller synchronises;
 *   - waveforms are float32 [B, S, T] / [B, 1, T] contiguous, exactly as the reference passes them;
 *   - image-like activations at the unit-op level are NHWC ("pixel-major") with an explicit pixel
 *     stride `ld` (elements): element (b,h,w,c) lives at ((b*H + h)*W + w)*ld + c.  dtype is
 *     DIFFSEP_F32 or DIFFSEP_BF16 (16-bit storage), C and ld multiples of 8;
 *   - the library is built twice from the same sources and exports this same interface both times:
 *     libdiffsep_hip.so stores the 16-bit tensors (code DIFFSEP_BF16) as raw bfloat16 bits,
 *     libdiffsep_hip_f16.so (-DDS_HALF_F16) as IEEE half precision: the same kernels and speed with 11
 *     instead of 8 significand bits (one score evaluation 2.4e-3 instead of 2.5e-2 from fp32).
 *     diffsep_version() names the build.  Handles and 16-bit buffers of one build mean nothing to the other;
 *   - an engine handle is bound to the device that was current at creation, owns the repacked
 *     weights + workspace, and is not thread-safe (the reference is one Python thread per process
 *     per GPU: evaluate_mp.py:339,510-513).  The parameter-table queries (diffsep_param_count / _info /
 *     _total) share ONE process-wide cache of the last architecture asked about: call them from one
 *     thread at a time.
 */
#ifndef DIFFSEP_HIP_H
#define DIFFSEP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIFFSEP_F32 0
#define DIFFSEP_BF16 1
/* fp32 tensors and fp32 everywhere except the matrix products: every MFMA k-block runs as three bf16 MFMAs on the
 * hi / lo bf16 halves of both operands (hi*hi + hi*lo + lo*hi, fp32 accumulation: products good to ~2^-17 instead of
 * 2^-24), ~2x the speed of DIFFSEP_F32.  Accepted by diffsep_engine_create and by the unit convolution / attention
 * entry points; tensors are fp32 exactly as for DIFFSEP_F32. */
#define DIFFSEP_F32_SPLIT 2

#define DIFFSEP_SDE_MIX 0      /* sdes/sdes.py:180-349  MixSDE      */
#define DIFFSEP_SDE_PRIORMIX 1 /* sdes/sdes.py:352-590  PriorMixSDE */

#define DIFFSEP_PRED_REVERSE_DIFFUSION 0 /* sdes/predictors.py:55-66 */
#define DIFFSEP_PRED_EULER_MARUYAMA 1    /* sdes/predictors.py:39-52 */
#define DIFFSEP_PRED_NONE 2              /* sdes/predictors.py:69-77 */

#define DIFFSEP_CORR_ALD2 0     /* sdes/correctors.py:94-128  */
#define DIFFSEP_CORR_NONE 1     /* sdes/correctors.py:131-141 */
#define DIFFSEP_CORR_ALD 2      /* sdes/correctors.py:58-91 (MixSDE only) */
#define DIFFSEP_CORR_LANGEVIN 3 /* sdes/correctors.py:35-55 (one step size for the whole batch) */

typedef struct diffsep_engine diffsep_engine; /* opaque */

/* Hyper-parameters of ScoreModelNCSNpp + its NCSNpp backbone.
 * Replaces: models/score_models.py:11-39 (ctor kwargs), models/ncsnpp.py:45-70 (ctor defaults),
 * config/model/default.yaml:14-31. */
typedef struct {
  int32_t nf;               /* backbone_args.nf (64 default.yaml:25; 128 icassp-separation.yaml:16) */
  int32_t num_sources;      /* S; backbone in = 2S+2, out = 2S (score_models.py:24-26) */
  int32_t n_levels;         /* len(ch_mult) = 7 */
  int32_t ch_mult[8];       /* (1,1,2,2,2,2,2) */
  int32_t num_res_blocks;   /* 2 */
  int32_t attn_resolution;  /* 16 (tested against the frequency axis, ncsnpp.py:367,415) */
  int32_t n_fft;            /* 510 -> image height n_fft/2+1 = 256 */
  int32_t hop;              /* 128 */
  float spec_abs_exponent;  /* 0.5 */
  float spec_factor;        /* 0.33 (0.15 published) */
  int32_t dtype;            /* DIFFSEP_F32 (exact), DIFFSEP_F32_SPLIT (parity-grade, 2x faster) or DIFFSEP_BF16 (16-bit storage: throughput) */
} diffsep_model_config;

/* sdes/sdes.py:217-240 (MixSDE ctor), :397-450 (PriorMixSDE ctor). */
typedef struct {
  int32_t kind; /* DIFFSEP_SDE_* */
  int32_t ndim; /* number of sources */
  float d_lambda, sigma_min, sigma_max;
  int32_t avg_len; /* PriorMixSDE only: length of the mix^2 moving average (510, sdes.py:383) */
} diffsep_sde_config;

/* sdes/__init__.py:132-145 (get_pc_sampler kwargs) + pl_model.py:687-701. */
typedef struct {
  int32_t N;               /* reverse steps */
  int32_t corrector_steps; /* n_steps of the corrector */
  float snr;               /* corrector step size */
  float eps;               /* t_eps: last time step */
  int32_t denoise;         /* return x_mean of the last predictor step */
  int32_t predictor;       /* DIFFSEP_PRED_* */
  int32_t corrector;       /* DIFFSEP_CORR_* */
} diffsep_sampler_config;

const char* diffsep_last_error(void);
const char* diffsep_version(void);

/* ---- parameter table: the canonical weight order is the reference state_dict order of
 * ScoreModelNCSNpp.backbone (models/ncsnpp.py:106-308: all_modules.{i}.*, then output_layer.*,
 * see SURVEY.md Appendix A).  The host assembles one flat float32 blob in this order. ---- */
int32_t diffsep_param_count(const diffsep_model_config* cfg);
/* name: e.g. "all_modules.4.Conv_0.weight"; shape: up to 4 dims (reference layout, e.g. OIHW);
 * offset: position (in floats) inside the flat blob. */
int32_t diffsep_param_info(const diffsep_model_config* cfg, int32_t idx, char* name, int32_t name_cap,
                           int64_t shape[4], int32_t* ndim, int64_t* offset);
int64_t diffsep_param_total(const diffsep_model_config* cfg);

/* Create from HOST weights (EMA already applied by the caller: pl_model.py:655-660).
 * Replaces DiffSepModel.__init__/load_from_checkpoint + .to(device) (separate.py:44-48). */
int32_t diffsep_engine_create(const diffsep_model_config* cfg, const float* weights_host, int64_t n_floats,
                              diffsep_engine** out);
void diffsep_engine_destroy(diffsep_engine* e);
/* bytes of device memory currently held (weights + workspace). */
int64_t diffsep_engine_device_bytes(const diffsep_engine* e);
/* Size the workspace for batches of up to B utterances of T samples now (the engine otherwise grows it on demand, and
 * hipFree / hipMalloc synchronise the whole device — i.e. every other stream).  Host-side counterpart in the
 * reference: none (torch's caching allocator). */
int32_t diffsep_engine_reserve(diffsep_engine* e, int32_t B, int64_t T, void* stream);
/* Debug aid: the engine's workspace arena (sampler state, then one forward's activations in launch order from
 * fwd_base).  Lets a test snapshot every intermediate tensor of a forward; no reference counterpart. */
int32_t diffsep_engine_debug_arena(const diffsep_engine* e, void** base, int64_t* bytes, int64_t* fwd_base);
/* Debug / test aid (engine option "track_tensors" = 1, eager forwards: diffsep_score_forward): for every activation tensor of the
 * last forward, in allocation order, out[4 i ..] = {largest finite |value|, number of non-finite values, rows H, channels C};
 * *n = number of tensors (out may be NULL to query).  How far the half-precision engines are from 65504. */
int32_t diffsep_engine_debug_absmax(diffsep_engine* e, double* out, int32_t cap, int32_t* n);
/* frames F = 1 + (T + n_fft - hop)/hop and padded width W = 64*ceil(F/64) for T samples
 * (score_models.py:83-91,107-112). Pure host arithmetic. */
int32_t diffsep_num_frames(const diffsep_model_config* cfg, int64_t T);
int32_t diffsep_padded_frames(const diffsep_model_config* cfg, int64_t T);

/* score_fn(x, t, mix): DiffSepModel.forward (pl_model.py:407-409) -> ScoreModelNCSNpp.forward
 * (models/score_models.py:126-138).  xt [B,S,T], t [B], mix [B,1,T], out [B,S,T]; all device f32. */
int32_t diffsep_score_forward(diffsep_engine* e, const float* xt, const float* t, const float* mix, float* out,
                              int32_t B, int64_t T, void* stream);

/* backbone only: NCSNpp.forward (models/ncsnpp.py:319-478) on an already-packed NHWC input
 * x [B,256,W,Cpad] (dtype of the engine, BEFORE the 2x-1 of ncsnpp.py:347-349), t [B] f32;
 * y [B,256,W,Cpad_out].  Used by parity tests to isolate the network from the STFT front end. */
int32_t diffsep_backbone_forward(diffsep_engine* e, const void* x, const float* t, void* y, int32_t B, int32_t W,
                                 void* stream);

/* The whole sampler: sdes.get_pc_sampler(...)() (sdes/__init__.py:132-190) as called from
 * DiffSepModel.get_pc_sampler (pl_model.py:687-711) with score_fn = the engine.
 *   mix_norm [B,1,T] : output of normalize_batch (pl_model.py:81-88);
 *   out      [B,S,T] : x_mean of the last predictor step if denoise else x (sdes/__init__.py:183);
 *   noise            : NULL -> on-device Philox N(0,1) draws keyed by `seed`; otherwise
 *                      [1 + N*(corrector_steps+1)][B][S][T] float32 draws consumed in the
 *                      reference's RNG order: prior, then per step corrector
 *                      draw(s), predictor draw;
 *   timesteps        : NULL -> torch.linspace(T=1, eps, N) semantics (sdes/__init__.py:175);
 *                      otherwise N+1 HOST floats for the scheduled sampler
 *                      (sdes/__init__.py:91-111; dt is still 1/N — reference quirk Q1);
 *   nfe_out          : N*(corrector_steps+1) (sdes/__init__.py:184), may be NULL. */
int32_t diffsep_pc_sample(diffsep_engine* e, const diffsep_sde_config* sde, const diffsep_sampler_config* smp,
                          const float* mix_norm, float* out, int32_t B, int64_t T, const float* noise,
                          uint64_t seed, const float* timesteps_host, int32_t* nfe_out, void* stream);

/* Extensions of the sampler call that have no counterpart in the reference (which runs one utterance at a time in
 * one precision: evaluate.py:348-376).  All fields optional (zero / NULL = diffsep_pc_sample behaviour).
 *   lengths_host [B]  : utterance b has lengths_host[b] <= T samples; the batch is zero-padded to T and every
 *                       utterance must have the padded frame count of T (diffsep_padded_frames).  Samples beyond an
 *                       utterance's length stay exactly zero through the whole sampler, so each utterance comes out
 *                       bit-for-bit as a B = 1 call on it alone with the same seed (fp32 engine; within the
 *                       GroupNorm-sum rounding of the weight-stationary bf16 kernel otherwise).  The 'langevin'
 *                       corrector couples the batch entries and is refused with lengths.
 *   seeds_host [B]    : per-utterance Philox seeds (with noise == NULL): utterance b draws exactly what a B = 1 call
 *                       with seed = seeds_host[b] draws, whatever batch it rides in.  Without it `seed` keys the
 *                       whole batch as in diffsep_pc_sample (with lengths: utterance b uses seed + b * 0x9E3779B97F4A7C15).
 *   tail_engine       : a second engine of the same architecture and weights (an fp32 or split-fp32 engine behind a bf16 one) that
 *                       evaluates the score of the FIRST head_steps and / or the LAST tail_steps reverse steps (their
 *                       corrector and predictor evaluations).  A score error enters the state scaled by the step size
 *                       G(t)^2, ~100x larger at t = 1 than at t = eps: it is the early steps whose precision decides
 *                       how closely the bf16 sampler follows the fp32 one (tools/hybrid_probe.py). */
typedef struct {
  const int64_t* lengths_host;
  const uint64_t* seeds_host;
  diffsep_engine* tail_engine;
  int32_t tail_steps;
  int32_t head_steps;
} diffsep_sampler_ext;
int32_t diffsep_pc_sample_ex(diffsep_engine* e, const diffsep_sde_config* sde, const diffsep_sampler_config* smp,
                             const diffsep_sampler_ext* ext, const float* mix_norm, float* out, int32_t B, int64_t T,
                             const float* noise, uint64_t seed, const float* timesteps_host, int32_t* nfe_out,
                             void* stream);

/* Use hipGraph replay of the per-NFE launch sequence inside diffsep_pc_sample (default 1). */
int32_t diffsep_engine_set_graph(diffsep_engine* e, int32_t enable);

/* Options (no counterpart in the reference: A/B switches, test aids and the bound of the graph cache).
 * Process-wide defaults of the kernel-dispatch switches "no_rw", "no_rw128", "rw_small", "no_rw_res" (0 / 1): read from the
 * environment variables DIFFSEP_NO_RW, DIFFSEP_NO_RW128, DIFFSEP_RW_SMALL, DIFFSEP_NO_RW_RES ONCE, changed here; they apply to
 * the unit entry points (diffsep_conv2d, ...) and are copied by every engine created afterwards. */
int32_t diffsep_set_option(const char* name, int64_t value);
/* Per engine: the same four switches, plus "graph_cache" (captured graphs kept, least recently used evicted; default 12),
 * "dbg_alloc" (log workspace allocations) and "ablate" (measurement aid: bit mask of launch classes that are skipped — the
 * results are garbage).  Synchronises the device and drops the engine's captured graphs. */
int32_t diffsep_engine_set_option(diffsep_engine* e, const char* name, int64_t value);
/* Current value of an engine option, "graphs_cached" = number of captured graphs held; -1 for an unknown name. */
int64_t diffsep_engine_get_option(const diffsep_engine* e, const char* name);

/* Measurement hook (bench.py): between begin and end every MFMA conv/GEMM launch of the engine is
 * bracketed by HIP events on its launch stream (graph replay bypassed).  Outputs are 12-entry arrays
 * indexed by kernel class: 3x3 {8x32 tile x 64 cout, 8x32 x 32, 8x8 x 64}, then the same
 * tiles for 1x1 / GEMM, the weight-stationary 64 -> 64 3x3 kernel, the small-image 3x3 kernel, the register-weight
 * 3x3 kernel (conv3x3_rw.hip), the fused attention block (attn_fused.hip), the streamed-weight 3x3 kernel (conv3x3_sw.hip) and
 * its split-precision sibling (conv3x3_sws.hip) — TWELVE entries: summed algorithmic flops, summed milliseconds, launch counts and (bytes, nullable)
 * summed algorithmic HBM bytes = every operand read once + the output written once. */
#define DIFFSEP_NUM_KERNEL_CLASSES 12
int32_t diffsep_engine_profile_begin(diffsep_engine* e);
/* Writes DIFFSEP_NUM_KERNEL_CLASSES entries per array — the caller's buffers must hold that many (ABI note: the count was 9
 * until round 3 and 10 until round 5; a caller compiled against an older header must use the _n form below or be rebuilt). */
int32_t diffsep_engine_profile_end(diffsep_engine* e, double* flops, double* ms, int64_t* launches, double* bytes);
/* The same with the capacity of the caller's arrays stated: writes min(n_classes, DIFFSEP_NUM_KERNEL_CLASSES) entries per
 * array and never more; *n_written (nullable) receives that number. */
int32_t diffsep_engine_profile_end_n(diffsep_engine* e, int32_t n_classes, double* flops, double* ms, int64_t* launches,
                                     double* bytes, int32_t* n_written);
int32_t diffsep_num_kernel_classes(void);
/* The launches of that span one by one (call after profile_end; out may be NULL to query *n): the kernel instantiation
 * with its template arguments, the problem shape, algorithmic flops / bytes and the measured duration.  Round 5: the
 * HBM-bound launches of the path (GroupNorm apply / FIR x2 resampling, STFT, iSTFT, SDE updates, RNG) are bracketed too and
 * appear here with cls = -1, flops = 0, Cin = channels, bytes = algorithmic HBM bytes (inputs read once, outputs written once);
 * they are not part of the per-class arrays of profile_end. */
typedef struct diffsep_prof_record {
  char kernel[128];
  int32_t B, H, W, Cin, Cout, taps, skip_cin, has_res, cls, _pad;
  double flops, bytes;
  double ms;
} diffsep_prof_record;
int32_t diffsep_engine_profile_records(diffsep_engine* e, diffsep_prof_record* out, int32_t cap, int32_t* n);

/* ------------------------------------------------------------------ unit entry points
 * (used by the parity tests; the engine calls the same launchers internally). */

/* Time embedding of the backbone: temb[B][4 nf] = Linear_2(SiLU(Linear_1(GaussianFourierProjection(log t)))).
 * Replaces: models/ncsnpp.py:324-343 (all_modules[0..2]), models/ncsnpp_utils/layerspp.py:37-47 (the projection).
 * fourier_w [nf] (the frozen W, scale included), w1 [4 nf][2 nf], b1 [4 nf], w2 [4 nf][4 nf], b2 [4 nf] (nn.Linear layout),
 * all float32 device pointers; workspace >= B * 6 nf floats. */
int32_t diffsep_time_embedding(const float* t, const float* fourier_w, const float* w1, const float* b1, const float* w2,
                               const float* b2, float* temb, int32_t B, int32_t nf, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* upfirdn2d with the [1,3,3,1] FIR, factor 2: upsample_2d / downsample_2d
 * (models/ncsnpp_utils/up_or_down_sampling.py:206-273; native op op/upfirdn2d.cpp:12-23,
 * op/upfirdn2d_kernel.cu:107-207).  up!=0: [B,H,W,C] -> [B,2H,2W,C]; else -> [B,H/2,W/2,C]. */
int32_t diffsep_upfirdn2d(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx,
                          int32_t ldy, int32_t up, int32_t dtype, void* stream);

/* y = act(GroupNorm(x)) with G = min(C/4,32), eps, affine (layerspp.py:264-266,292,313);
 * act: 0 none (attention GN, layerspp.py:78) / 1 SiLU.  resample: 0 none / 1 FIR up / 2 FIR down
 * applied to act(GN(x)) (layerspp.py:294-299); xr (nullable) receives the same resampling of raw x. */
int32_t diffsep_groupnorm_act(const void* x, const float* gamma, const float* beta, void* y, void* xr, int32_t B,
                              int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t ldy, int32_t ldxr,
                              int32_t groups, float eps, int32_t act, int32_t resample, int32_t dtype,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* 3x3 (pad 1) or 1x1 convolution, NHWC, implicit GEMM on MFMA:
 * y = (conv(x, w) + bias[co] + bias_b[b,co] + res) * out_scale     (layers.py:112-119,141-156;
 * the fused terms are layerspp.py:306-323).  w: [Cout][taps][Cin] in `dtype` (taps = ky*3+kx). */
int32_t diffsep_conv2d(const void* x, const void* w, const float* bias, const float* bias_b, const void* res,
                       void* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                       int32_t ldx, int32_t ldr, int32_t ldy, float out_scale, int32_t dtype, void* stream);

/* GroupNorm statistics only: per-(b,c) scale = rstd*gamma, shift = beta - mean*scale ([B][C] fp32), optionally
 * over the in-place channel concat cat([x, x2]) (x holds C1 channels).  The convolutions consume them. */
int32_t diffsep_groupnorm_stats(const void* x, const void* x2, int32_t C1, const float* gamma, const float* beta,
                                float* scale, float* shift, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx,
                                int32_t ldx2, int32_t groups, float eps, int32_t dtype, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* diffsep_conv2d with the producer-side fusions the engine uses: the input may be cat([x, x2], C) read in
 * place (ncsnpp.py:411) and act(GroupNorm(.)) (layerspp.py:292,313) is applied to it on the fly from
 * per-(b,c) scale/shift (gn_act: 0 none, 1 SiLU); zero padding is applied AFTER the activation. */
int32_t diffsep_conv2d_fused(const void* x, const void* x2, int32_t C1, const float* gn_scale, const float* gn_shift,
                             int32_t gn_act, const void* w, const float* bias, const float* bias_b, const void* res,
                             void* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                             int32_t ldx, int32_t ldx2, int32_t ldr, int32_t ldy, float out_scale, int32_t dtype,
                             int64_t* stats, int32_t w_chunk, const int64_t* gn_acc1, const int64_t* gn_acc2,
                             const float* gn_gamma, const float* gn_beta, int32_t gn_groups, void* stream);
/* `w_chunk` = 0: w is [Cout][k*k][Cin_pad]; = kc = diffsep_conv2d_chunk(ksize, dtype): w is chunk-major
 * [Cin_pad / kc][k*k][Cout][kc] (the layout the engine keeps its weights in: one K stage of the kernel is then
 * contiguous in memory and is fetched in full 128-byte lines). */
int32_t diffsep_conv2d_chunk(int32_t ksize, int32_t dtype);
/* The streamed-weight 3x3 kernel (csrc/conv3x3_sw.hip; the 128-cout layers with 192 / 256 input channels or a folded 1x1 skip
 * on up to 256 raw channels: ncsnpp.py:409-417, layerspp.py:291-323) as a unit, whatever the dispatch would pick.  Dense NHWC
 * 16-bit tensors: x [B][H][W][C1 or Cin], x2 (nullable) the other Cin - C1 channels; act(GroupNorm(.)) from per-(b, c)
 * scale / shift when gn_scale != NULL (SiLU), raw input otherwise; optional folded skip y += sw * cat([sx, sx2]) on RAW channels;
 * y [B][H][W][Cout] = (conv + bias + bias_b[b]) * out_scale; `stats` as in diffsep_conv2d_fused.  w_frag / sw_frag: the
 * [Cout][3 x 3][Cin] / [Cout][sCin] weights in FRAGMENT-major order — element (cout, tap, cin) at diffsep_frag_index(cout, tap,
 * cin, taps, Cout): one k-step (64-channel chunk, tap, 16-channel block) of all couts is contiguous, cout group by cout group,
 * so that a wave's 1 KB load instruction is one MFMA B-operand.
 * dtype = DIFFSEP_F32_SPLIT: the split mode's sibling (csrc/conv3x3_sws.hip; sdes/__init__.py:166-188 runs on it in the head of a
 * hybrid run): fp32 tensors, Cout = 64 / 128 / 256, every product as three bfloat16 MFMAs on hi / lo planes; w_frag / sw_frag
 * are then PAIRS of bfloat16 planes (hi = bf16(w), lo = bf16(w - hi)) at diffsep_frag_index_split(cout, tap, cin, taps, Cout,
 * plane).  `res` (nullable, no skip beside it; 16-bit: Cout = 128): residual [B][H][W][Cout] added before out_scale — it rides
 * through the matrix cores against `ident_frag`, the fragment copy (of that mode) of the Cout x Cout identity (taps = 1).
 * 16-bit tile shapes: 8 x 32 pixels x 128 couts, 4 x 32 x 128 where H % 8 != 0 (and, in the engine, on levels with fewer 8-row
 * tiles than compute units), 8 x 32 x 64 for the one 64-cout layer the register-weight kernel does not hold (Cin = 192). */
int32_t diffsep_conv3x3_streamed(const void* x, const void* x2, int32_t C1, const float* gn_scale, const float* gn_shift,
                                 const void* w_frag, const float* bias, const float* bias_b, const void* sx, const void* sx2,
                                 int32_t sC1, int32_t sCin, const void* sw_frag, void* y, int32_t B, int32_t H, int32_t W,
                                 int32_t Cin, int32_t Cout, float out_scale, int32_t dtype, int64_t* stats, const void* res,
                                 const void* ident_frag, void* stream);
int64_t diffsep_frag_index(int32_t cout, int32_t tap, int32_t cin, int32_t taps, int32_t Cout);
int64_t diffsep_frag_index_split(int32_t cout, int32_t tap, int32_t cin, int32_t taps, int32_t Cout, int32_t plane);
/* `stats` (nullable): channel-sum accumulators of the OUTPUT for the next GroupNorm, [B][Cout][2] int64 fixed point
 * (sum * 2^24, sum of squares * 2^16; exact to 6e-8 / 1.5e-5 per tile, no overflow while the per-image sum of
 * squares of a channel stays below 1.4e14).  Every block ADDS its tile's totals with integer atomics (associative:
 * bit-reproducible), the caller zeroes the buffer first — so no separate pass over the tensor is needed
 * (layerspp.py:313: GroupNorm_1 follows Conv_0).
 * `gn_acc1` (nullable, instead of gn_scale / gn_shift): such accumulators of x (and gn_acc2 of x2) plus the
 * GroupNorm affine parameters gamma / beta [Cin] and the group count; the kernel derives scale / shift itself
 * (eps = 1e-6, statistics over H * W * Cin / groups elements per group). */

/* ResnetBlockBigGANpp.forward (layerspp.py:291-323) through the engine's own block code (GroupNorm + SiLU fused into
 * the convolutions' input staging, FIR up / down of both branches in one pass, Conv_2 folded into Conv_1, temb
 * projection Dense_0(act(temb))): x [B,H,W,in_ch] NHWC (dtype), temb [B,temb_dim] f32 -> y [B,H',W',out_ch].
 * params_host: the block's parameters as one float32 blob in reference state_dict order (GroupNorm_0.{weight,bias},
 * Conv_0.{weight,bias}, Dense_0.{weight,bias}, GroupNorm_1.{weight,bias}, Conv_1.{weight,bias}[, Conv_2.{weight,
 * bias} when in_ch != out_ch or up or down]).  Test aid: builds a one-module engine per call and synchronises. */
int32_t diffsep_resblock_forward(int32_t in_ch, int32_t out_ch, int32_t up, int32_t down, int32_t temb_dim,
                                 int32_t dtype, const float* params_host, int64_t n_floats, const void* x,
                                 const float* temb, void* y, int32_t B, int32_t H, int32_t W, void* stream);
/* AttnBlockpp.forward (layerspp.py:76-92) through the engine's block code: x, y [B,H,W,C] NHWC (dtype);
 * params_host = GroupNorm_0.{weight,bias}, NIN_0.{W,b} .. NIN_3.{W,b} as one float32 blob.  Test aid. */
int32_t diffsep_attnblock_forward(int32_t channels, int32_t dtype, const float* params_host, int64_t n_floats,
                                  const void* x, void* y, int32_t B, int32_t H, int32_t W, void* stream);

/* AttnBlockpp core (layerspp.py:83-87): o = softmax(q k^T * C^-0.5) v over L = H*W tokens.
 * q,k [B,L,C] (ld), vt [B,C,Lp] (V transposed, Lp = L rounded up to 8), o [B,L,C];
 * ws >= B*L*Lp*(2*elt) bytes.  QK^T and PV run on MFMA. */
int32_t diffsep_attention(const void* q, const void* k, const void* vt, void* o, int32_t B, int32_t L, int32_t C,
                          int32_t ld, int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream);

/* pre_process (score_models.py:107-116) + the 2x-1 of ncsnpp.py:347-349: cat(xt, mix) -> right pad
 * n_fft-hop -> STFT(n_fft, hop, periodic Hann, center, zero pad) -> |z|^e e^{j angle} * factor ->
 * [re x S+1 | im x S+1] channels -> zero-pad frames to W (then 2x-1).
 * xt [B,S,T], mix [B,1,T] f32 -> y [B, n_fft/2+1, W, Cpad] (dtype), Cpad = roundup(2S+2, 8).
 * centered_shift != 0 applies 2x-1 (what the engine does); 0 leaves the packed spectrogram.
 * ws >= 2*((B*(S+1)*F + 8)*512*4 + 256) bytes (windowed frames + transposed spectrum of the DFT GEMM). */
int32_t diffsep_stft_pack(const float* xt, const float* mix, void* y, int32_t B, int32_t S, int64_t T,
                          int32_t n_fft, int32_t hop, float exponent, float factor, int32_t W, int32_t Cpad,
                          int32_t centered_shift, int32_t dtype, void* workspace, int64_t workspace_bytes,
                          void* stream);

/* post_process (score_models.py:118-124): unpad frames -> channels to complex -> z/|factor| ->
 * |z|^(1/e) e^{j angle} -> iSTFT -> crop to T.  x [B,256,W,Cpad] (first 2S channels used) -> out [B,S,T];
 * ws >= 2*(B*S*F*512*4 + 256) bytes (decompressed bins + inverse-DFT frames). */
int32_t diffsep_istft_unpack(const void* x, float* out, int32_t B, int32_t S, int64_t T, int32_t n_fft,
                             int32_t hop, float exponent, float factor, int32_t W, int32_t Cpad, int32_t dtype,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* PriorMixSDE._std_sigma_mix (sdes/sdes.py:477-489): sigma_mix[b,t] = 0.5*sqrt(clamp(avg_pool1d(mix^2, avg_len,
 * stride 1, pad avg_len/2), 1e-4)); mix [B,1,T] -> sigma_mix [B,T].  The three updates below take it as their
 * `sigma_mix` argument for kind = DIFFSEP_SDE_PRIORMIX (time-varying std, sdes.py:451-470,515-532) and NULL
 * for MixSDE. */
int32_t diffsep_sde_sigma_mix(const float* mix, float* sigma_mix, int32_t B, int64_t T, int32_t avg_len, void* stream);
/* MixSDE.prior_sampling (sdes/sdes.py:334-346) / PriorMixSDE.prior_sampling (:564-587):
 * x_T = 0.5*y (bcast) + L(T) @ z. */
int32_t diffsep_sde_prior(const diffsep_sde_config* sde, const float* y, const float* z, float* x, int32_t B,
                          int32_t S, int64_t T, const float* sigma_mix, void* stream);
/* AnnealedLangevinDynamics2.update_fn body for one step given the score
 * (sdes/correctors.py:115-126). t [B] device. */
int32_t diffsep_sde_corrector_update(const diffsep_sde_config* sde, float snr, const float* x, const float* t,
                                     const float* score, const float* z, float* x_out, float* x_mean_out,
                                     int32_t B, int32_t S, int64_t T, const float* sigma_mix, int32_t variant,
                                     void* stream); /* variant: 0 = ald2, 1 = ald (sdes/correctors.py:58-91) */
/* ReverseDiffusionPredictor.update_fn given the score (sdes/predictors.py:60-66 ->
 * sdes/sdes.py:163-171,93-107,275-284); dt = 1/N.  EulerMaruyamaPredictor (predictors.py:39-52) is the same
 * update.  probability_flow != 0: RSDE.discretize with half the score term and no noise (sdes.py:165-171). */
int32_t diffsep_sde_predictor_update(const diffsep_sde_config* sde, int32_t N, const float* x, const float* t,
                                     const float* score, const float* z, float* x_out, float* x_mean_out,
                                     int32_t B, int32_t S, int64_t T, const float* sigma_mix,
                                     int32_t probability_flow, void* stream);
/* ---- the SDE object surface a user-written Predictor / Corrector reaches through the reference API.  The fused
 * sampler never calls these; diffsep_amd/sdes/sdes.py builds sde() / marginal_prob() / mult_std() / discretize() /
 * reverse() on them. ----
 * SDE.sde (MixSDE sdes/sdes.py:275-284, PriorMixSDE :451-470) scaled for SDE.discretize (:93-107):
 *   drift_out [B,S,T] = f_scale * (-lambda P x);  diffusion_out = g_scale * g(t) [B]  (MixSDE)
 *                                                              or g_scale * g(t) sigma_mix [B,S,T] (PriorMixSDE).
 *   sde(): f_scale = g_scale = 1; discretize(): f_scale = dt, g_scale = sqrt(dt), dt = 1/N (quirk Q1). */
int32_t diffsep_sde_coefficients(const diffsep_sde_config* sde, const float* x, const float* t, const float* sigma_mix,
                                 float* drift_out, float* diffusion_out, int32_t B, int32_t S, int64_t T, float f_scale,
                                 float g_scale, void* stream);
/* MixSDE._mean (sdes/sdes.py:286-294): (A + exp(-lambda t) P) x0, the mean of marginal_prob (:322-324). */
int32_t diffsep_sde_mean(const diffsep_sde_config* sde, const float* x0, const float* t, float* mean_out, int32_t B,
                         int32_t S, int64_t T, void* stream);
/* MixSDE._std (sdes/sdes.py:315-320) -> std_out [B,S,S]; PriorMixSDE._std (:515-532, sigma_mix != NULL) ->
 * std_out [B,S,S,T]: the second member of marginal_prob. */
int32_t diffsep_sde_std(const diffsep_sde_config* sde, const float* t, const float* sigma_mix, float* std_out, int32_t B,
                        int32_t S, int64_t T, void* stream);
/* MixSDE.mult_std (std @ x, sdes/sdes.py:326-328; per_sample = 0, std [B,S,S]) / PriorMixSDE.mult_std
 * (einsum "bcdt,bdt->bct", :534-537; per_sample = 1, std [B,S,S,T]) for any dense std. */
int32_t diffsep_sde_mult_std(const float* std, const float* x, float* out, int32_t B, int32_t S, int64_t T,
                             int32_t per_sample, void* stream);
/* RSDE.discretize (sdes/sdes.py:163-171) given the forward discretisation: rev_f = f - G^2 score (x 0.5 when
 * probability_flow); G is [B] (g_full = 0) or [B, n_per_batch] (g_full = 1). */
int32_t diffsep_sde_reverse_drift(const float* f, const float* G, const float* score, float* rev_f_out, int32_t B,
                                  int64_t n_per_batch, int32_t g_full, int32_t probability_flow, void* stream);

/* LangevinCorrector.update_fn body for one step (sdes/correctors.py:43-53): step = 2 (snr <||z_b||> / <||g_b||>)^2
 * from batch-mean norms, x_mean = x + step g, x = x_mean + sqrt(2 step) z.  x etc. are [B, n_per_batch];
 * ws >= 16*B + 16 bytes. */
int32_t diffsep_sde_langevin_update(float snr, const float* x, const float* score, const float* z, float* x_out,
                                    float* x_mean_out, int32_t B, int64_t n_per_batch, void* workspace,
                                    int64_t workspace_bytes, void* stream);

/* normalize_batch (pl_model.py:81-88): per-utterance mean / unbiased std (clamped 1e-5) over (1,T).
 * mix [B,1,T] -> mix_norm; mean,std [B] (nullable). */
int32_t diffsep_normalize_batch(const float* mix, float* mix_norm, float* mean, float* std, int32_t B, int64_t T,
                                void* stream);
/* scale_output (separate.py:73-78): alpha = <mix,sep>/sum(sep^2+1e-10); sep*alpha, in place. */
int32_t diffsep_scale_output(const float* mix, float* sep, int32_t B, int32_t S, int64_t T, void* stream);

/* Gram matrices of references and estimates for the separation metrics (evaluate.py:103-132:
 * fast_bss_eval.si_bss_eval_sources(ref, est, zero_mean=False, compute_permutation=True)): out [B][3][S][S] float64 =
 * { ref ref^T, ref est^T, est est^T }.  SI-SDR / SI-SIR / SI-SAR and the best permutation follow from these alone. */
int32_t diffsep_gram(const float* ref, const float* est, double* out, int32_t B, int32_t S, int64_t T, void* stream);

/* on-device standard normal draws (Philox4x32-10 + Box-Muller); used when noise == NULL. */
int32_t diffsep_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream);
/* the same for a zero-padded batch of utterances with their own seeds and lengths (device arrays [B]): row (b, s) of
 * out [B,S,T] holds values s*len_b .. (s+1)*len_b - 1 of the stream diffsep_randn(seed_b, stream_id) draws, then zeros. */
int32_t diffsep_randn_batch(float* out, int32_t B, int32_t S, int64_t T, const uint64_t* seeds, const int32_t* lengths,
                            uint64_t stream_id, void* stream);

/* float32 <-> engine dtype conversion helper for the tests (n elements). */
int32_t diffsep_convert(const void* src, void* dst, int64_t n, int32_t src_dtype, int32_t dst_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSEP_HIP_H */
