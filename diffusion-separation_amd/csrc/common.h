// common.h — shared device helpers for the gfx950 kernels (wave64, NHWC activations).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define DS_F32 0
#define DS_BF16 1
#define DS_F32_SPLIT 2  // boundary value only: fp32 tensors, MFMA products as 3 bf16 MFMAs (ConvArgs.split); kernels see DS_F32
#define DS_MAX_SRC 4
// fixed-point scales of the GroupNorm channel-sum accumulators (int64): sums 2^-24, sums of squares 2^-16
#define DS_STAT_SUM_SCALE 16777216.0
#define DS_STAT_SQ_SCALE 65536.0
// Add into an accumulator (device-scope integer atomic: associative, so totals are bit-reproducible).
__device__ inline void ds_stat_add(long long* acc, long long v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)v);
}

// ---------------------------------------------------------------- error plumbing (host)
void ds_set_error(const std::string& s);
#define DS_CHECK(cond, msg)                                                            \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      ds_set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + (msg)); \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)
#define DS_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ds_set_error(std::string(__FILE__) + ":" + std::to_string(__LINE__) + ": " + #expr + " -> " + \
                   hipGetErrorString(_e));                                             \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)
#define DS_LAUNCH_CHECK() DS_HIP(hipGetLastError())
// hipFuncAttributeMaxDynamicSharedMemorySize lives per DEVICE: set it once per device ordinal (an engine may sit on any GPU
// of the process; legal during stream capture)
#define DS_FUNC_LDS_ONCE(kern, bytes)                                                                              \
  do {                                                                                                             \
    static bool done_[32] = {};                                                                                    \
    int dev_ = 0;                                                                                                  \
    (void)hipGetDevice(&dev_);                                                                                     \
    if (dev_ < 0 || dev_ >= 32 || !done_[dev_]) {                                                                  \
      DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
      if (dev_ >= 0 && dev_ < 32) done_[dev_] = true;                                                              \
    }                                                                                                              \
  } while (0)

// ---------------------------------------------------------------- 16-bit storage format
// The 16-bit tensors of the engine ("bf16_t" = raw 16 bits) are bfloat16 by default and IEEE half precision when the
// library is built with -DDS_HALF_F16 (libdiffsep_hip_f16.so): the same kernels at the same MFMA rate with 11 instead
// of 8 significand bits — one score evaluation is 2.3e-3 instead of 1.9e-2 from fp32 (tools/probes/storage_dtype_probe.py).
// Everything that touches stored 16-bit values goes through h2f / f2h / pack_h2 / h_lo / h_hi / mfma_h*; the split mode's
// hi / lo planes (fp32 tensors as two bfloat16 halves) are bfloat16 in both builds (pack_bf16x2, bf_lo / bf_hi, mfma_bf*).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4_acc;
__host__ __device__ inline float bf2f(bf16_t v) {
  union { uint32_t u; float f; } c; c.u = ((uint32_t)v) << 16; return c.f;
}
// two floats -> packed bf16x2 (RNE): one v_cvt_pk_bf16_f32 on gfx950
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ inline float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ inline float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__host__ __device__ inline bf16_t f2bf(float f) {  // round-to-nearest-even
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(unsigned short, (__bf16)f);
#endif
  union { uint32_t u; float f; } c; c.f = f;
  uint32_t u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ inline f32x16 mfma_bf32(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ inline f32x4_acc mfma_bf16(const uint4& a, const uint4& b, const f32x4_acc& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
#ifdef DS_HALF_F16
#define DS_HALF_NAME "f16"
#define DS_MFMA_H32_ASM "v_mfma_f32_32x32x16_f16"
#define DS_CVT_PK_H_ASM "v_cvt_pk_f16_f32"
#define DS_H_ONE 0x3c00u   // 1.0
__host__ __device__ inline float h2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__host__ __device__ inline bf16_t f2h(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ inline uint32_t pack_h2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
__device__ inline float h_lo(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[0]; }
__device__ inline float h_hi(uint32_t w) { return (float)__builtin_bit_cast(f16x2_t, w)[1]; }
__device__ inline f32x16 mfma_h32(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ inline f32x4_acc mfma_h16(const uint4& a, const uint4& b, const f32x4_acc& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
#else
#define DS_HALF_NAME "bf16"
#define DS_MFMA_H32_ASM "v_mfma_f32_32x32x16_bf16"
#define DS_CVT_PK_H_ASM "v_cvt_pk_bf16_f32"
#define DS_H_ONE 0x3f80u   // 1.0
__host__ __device__ inline float h2f(bf16_t v) { return bf2f(v); }
__host__ __device__ inline bf16_t f2h(float f) { return f2bf(f); }
__device__ inline uint32_t pack_h2(float lo, float hi) { return pack_bf16x2(lo, hi); }
__device__ inline float h_lo(uint32_t w) { return bf_lo(w); }
__device__ inline float h_hi(uint32_t w) { return bf_hi(w); }
__device__ inline f32x16 mfma_h32(const uint4& a, const uint4& b, const f32x16& c) { return mfma_bf32(a, b, c); }
__device__ inline f32x4_acc mfma_h16(const uint4& a, const uint4& b, const f32x4_acc& c) { return mfma_bf16(a, b, c); }
#endif

template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int KV = 4;  // elements per 16 bytes
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct Elt<bf16_t> {
  static constexpr int KV = 8;
  __device__ static inline float ld(const bf16_t* p) { return h2f(*p); }
  __device__ static inline void st(bf16_t* p, float v) { *p = f2h(v); }
};

// 8 consecutive channels <-> 8 floats (the elementwise kernels' unit of work: C % 8 == 0 always).
template <typename T> __device__ inline void load8(const T* p, float* f);
template <> __device__ inline void load8<float>(const float* p, float* f) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <> __device__ inline void load8<bf16_t>(const bf16_t* p, float* f) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  f[0] = h_lo(u.x); f[1] = h_hi(u.x);
  f[2] = h_lo(u.y); f[3] = h_hi(u.y);
  f[4] = h_lo(u.z); f[5] = h_hi(u.z);
  f[6] = h_lo(u.w); f[7] = h_hi(u.w);
}
template <typename T> __device__ inline void store8(T* p, const float* f);
template <> __device__ inline void store8<float>(float* p, const float* f) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}
template <> __device__ inline void store8<bf16_t>(bf16_t* p, const float* f) {
  uint4 u;
  u.x = pack_h2(f[0], f[1]);
  u.y = pack_h2(f[2], f[3]);
  u.z = pack_h2(f[4], f[5]);
  u.w = pack_h2(f[6], f[7]);
  *reinterpret_cast<uint4*>(p) = u;
}

// SiLU / exp: the fp32 (parity) instantiations use the accurate ocml expf, bf16 the hardware v_exp_f32.
template <typename T> __device__ inline float exp_t(float v);
template <> __device__ inline float exp_t<float>(float v) { return expf(v); }
template <> __device__ inline float exp_t<bf16_t>(float v) { return __expf(v); }
template <typename T> __device__ inline float silu_t(float v);
template <> __device__ inline float silu_t<float>(float v) { return v / (1.0f + expf(-v)); }
// bf16 mode: v_exp_f32 + v_rcp_f32 (1 ulp) instead of the 11-instruction IEEE division
template <> __device__ inline float silu_t<bf16_t>(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// wave64 butterfly reductions (no LDS)
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- launcher prototypes (host)
struct ConvArgs {
  const void* x; long x_bs; int ldx;     // A operand: [B][M][ldx]; x_bs = batch stride (elements)
  const void* x2; long x2_bs; int ldx2;  // optional second A source: channels [C1, Cin) come from x2 (concat in place)
  int C1;                                // channels taken from x when x2 != null
  const void* w; long w_bs;              // Bt operand: [Cout][taps][Cin]; w_bs batch stride (0 = shared)
  int w_chunked;                         // != 0: weights are chunk-major [Cin/kc][taps][Cout][kc] with kc = this value
  // optional second copies of w / sw in the register-weight kernel's FRAGMENT-major order (conv3x3_rw.hip, ds_rw_frag_index):
  // one load instruction of a wave = 1 KB of contiguous memory.  Null: that kernel gathers its fragments from w / sw.
  const void* w_frag; const void* sw_frag;
  // split mode (conv3x3_sws.hip): w_frag / sw_frag are hi / lo fragment copies (ds_sws_frag_index); ident_frag = that copy of the
  // Cout x Cout identity matrix (taps = 1), which a residual meets as a folded skip
  const void* ident_frag;
  // optional fused 1x1 skip convolution on the raw block input (3x3 launches only): y += sw * cat([sx, sx2])
  const void* sx; long sx_bs; int ldsx;  //   [B][M][ldsx], sCin channels (sC1 from sx when sx2 != null)
  const void* sx2; long sx2_bs; int ldsx2; int sC1; int sCin;
  const void* sw; int sw_chunked;        //   [Cout][sCin] or chunk-major [sCin/kc][Cout][kc] (kc = the 3x3 kernel's)
  const float* gn_scale;                 // optional fused GroupNorm-apply on A: f(x) = act(x*scale[b,c] + shift[b,c])
  const float* gn_shift; int gn_act;     //   scale/shift are [B][Cin] fp32; gn_act: 0 none, 1 SiLU
  const float* bias;                     // [Cout] (mode 0) or [M] (mode 1) or null
  const float* bias_b; int bias_b_ld;    // per-batch column bias [B][bias_b_ld] or null
  int bias_mode;                         // 0 = along Cout, 1 = along rows (M)
  const float* div_b;                    // per-batch divisor applied to the accumulator BEFORE bias (h / t), or null
  const void* res; long res_bs; int ldr; // residual added before out_scale, or null
  float out_scale;
  void* y; long y_bs; int ldy;
  // optional [B][Cout][2] accumulators (sum, sum of squares of the output, fixed point DS_STAT_*_SCALE, int64):
  // every block ADDS its tile's totals with integer atomics (bit-reproducible); the caller zeroes them first
  long long* stats_acc;
  // GroupNorm of the INPUT straight from such accumulators (of x and x2) instead of gn_scale / gn_shift:
  const long long* gn_acc1; const long long* gn_acc2; const float* gn_gamma; const float* gn_beta;
  int gn_groups; float gn_inv_count; float gn_eps;  // inv_count = 1 / (H * W * Cin / groups)
  int B, H, W;                           // taps==9: image H x W; taps==1: M = H*W rows
  int Cin, Cout, taps;
  int dtype;
  int split;                             // fp32 only: 3 bf16 MFMAs per k-block on hi / lo halves (2^-17 products) instead of fp32 MFMAs
  unsigned opts;                         // DS_OPT_* dispatch switches (an engine's copy, or ds_default_opts() at the unit entry points)
};
// Dispatch switches of the convolution launchers (A/B and test aids).  The process defaults are read from
// the environment ONCE (DIFFSEP_NO_RW, DIFFSEP_NO_RW128, DIFFSEP_RW_SMALL, DIFFSEP_NO_RW_RES) and changed with
// diffsep_set_option; an engine copies them at creation (diffsep_engine_set_option changes its copy and drops its captured
// graphs), so a launch decision never reads the environment.
#define DS_OPT_NO_RW 1u       // no register-weight 3x3 kernel (conv3x3_rw.hip): generic / weight-stationary tiles instead
#define DS_OPT_NO_RW128 2u    // ... for its 128-cout variants only
#define DS_OPT_RW_SMALL 4u    // register-weight kernel also for launches with fewer tiles than CUs (unit tests of small shapes)
#define DS_OPT_NO_RW_RES 8u   // residual launches stay on the weight-stationary kernel
#define DS_OPT_NO_ATTN_FUSED 64u  // attention blocks as the 11 separate launches of rounds 1 - 3 (A/B)
#define DS_OPT_NO_SPLIT256 128u   // cat(128, 128) -> 128 blocks as ONE launch per convolution on the generic tile (A/B)
#define DS_OPT_NO_STFT_FUSED 256u // (process default only) STFT / iSTFT as the launch sequences of rounds 1 - 4 instead of the fused kernels (A/B)
#define DS_OPT_RW_QUARTER 1024u   // ... on a quarter of the CUs (A/B)
#define DS_OPT_RW_BIG_HALF 2048u  // register-weight launches with > 4 tiles per block (the 256-row level) on half the CUs (A/B)
#define DS_OPT_RW_HALF 512u       // register-weight launches with <= 4 tiles per block on half the CUs (A/B)
#define DS_OPT_NO_SW 4096u       // no streamed-weight 3x3 kernel (conv3x3_sw.hip): generic tile / two-launch cat route instead (A/B)
#define DS_OPT_NO_SW_RW 8192u    // ... only for the launches the register-weight kernel does not take (A/B: by default it also takes the
                                 // 128-cout launches with < 2 tiles of 8 x 32 per CU, where that kernel pays its weight prologue per tile)
#define DS_OPT_NO_SW_ROWS4 32768u // ... not on levels with fewer 8 x 32 tiles than compute units (its 4 x 32 tile variant; A/B)
#define DS_OPT_SW_ROWS4 65536u    // (tests, with rw_small: the 4 x 32 tile variant on every shape)
#define DS_OPT_NO_SWS 16384u     // no split-mode streamed-weight kernel (conv3x3_sws.hip): the generic tile in split mode instead (A/B)
#define DS_OPT_NO_WFRAG 32u   // the engine does not hand the fragment-major weight copies to the register-weight kernel (A/B)
unsigned ds_default_opts();
int ds_num_cus();  // compute units of the current device (cached per device ordinal)
int ds_launch_conv(const ConvArgs& a, hipStream_t st);
// name (with its template arguments) of the kernel instantiation the calling thread's last ds_launch_conv ran
const char* ds_last_conv_kernel();
void ds_set_last_conv_kernel(const char* name);
int ds_conv_config_id(const ConvArgs& a);
int ds_conv_chunk(int taps, int dtype);
bool ds_conv_skip_supported(int H, int W, int Cout, int dtype);
// layout of the fragment-major copies: element (cout co, tap, input channel ch) of a [Cout][taps][Cin] weight tensor lives at
//   ((((ch / 64 * taps + tap) * 4 + ch % 64 / 16) * (Cout / 32) + co / 32) * 64 + (ch % 16 / 8) * 32 + co % 32) * 8 + ch % 8
// (k-step = (64-channel chunk, tap, 16-channel block); then cout group, lane = (k-half, cout), 8 channels)
__host__ __device__ inline long ds_rw_frag_index(int co, int tap, int ch, int taps, int Cout) {
  return (((((long)(ch / 64) * taps + tap) * 4 + (ch % 64) / 16) * (Cout / 32) + co / 32) * 64 + ((ch % 16) / 8) * 32 + co % 32) * 8 + ch % 8;
}
inline bool ds_rw_frag_shape(int taps, int Cin, int Cout) {  // the weight shapes conv3x3_rw.hip can take (3x3 and folded 1x1 skip)
  return (Cout == 64 || Cout == 128) && (Cin == 64 || Cin == 128) && (taps == 1 || !(Cout == 128 && Cin == 64));
}
bool ds_conv_rw_eligible(const ConvArgs& a);   // conv3x3_rw.hip: register-resident weights, 64 / 128 -> 64 bf16, >= 32-row images
int ds_launch_conv_rw(const ConvArgs& a, hipStream_t st);
// the weight shapes conv3x3_sw.hip streams (fragment-major copies of 3x3 weights and folded 1x1 skips)
// (Cout 64 too: ds_conv_sw_supported takes e.g. the 192 -> 64 layer of the 128-row level, which is no register-weight shape; until
// round 6 those copies existed only because the split kernel's shape test below happened to cover them)
inline bool ds_sw_frag_shape(int taps, int Cin, int Cout) { return (Cout == 64 || Cout == 128 || Cout == 256) && Cin % 64 == 0 && Cin >= 64 && Cin <= 512; }
bool ds_conv_sw_supported(const ConvArgs& a);  // conv3x3_sw.hip: streamed weights, 64 .. 256 -> 128 n couts, 16-bit
bool ds_conv_sw_eligible(const ConvArgs& a);   // ... and dispatched there
int ds_launch_conv_sw(const ConvArgs& a, hipStream_t st);
// conv3x3_sws.hip: the split mode's streamed-weight kernel (fp32 tensors, hi / lo bfloat16 planes).  Element (cout co, tap, input
// channel ch) of a [Cout][taps][Cin] weight tensor, plane 0 = bf16(w), plane 1 = bf16(w - plane 0), lives at
//   ((((ch / 32 * taps + tap) * 2 + ch % 32 / 16) * 2 + plane) * (Cout / 32) + co / 32) * 64 + (ch % 16 / 8) * 32 + co % 32) * 8 + ch % 8
__host__ __device__ inline long ds_sws_frag_index(int co, int tap, int ch, int taps, int Cout, int plane) {
  return ((((((long)(ch / 32) * taps + tap) * 2 + (ch % 32) / 16) * 2 + plane) * (Cout / 32) + co / 32) * 64 + ((ch % 16) / 8) * 32 + co % 32) * 8 + ch % 8;
}
inline bool ds_sws_frag_shape(int taps, int Cin, int Cout) { return (Cout == 64 || Cout == 128 || Cout == 256) && Cin % 64 == 0 && Cin >= 64 && Cin <= 256; }
bool ds_conv_sws_supported(const ConvArgs& a);
bool ds_conv_sws_eligible(const ConvArgs& a);
int ds_launch_conv_sws(const ConvArgs& a, hipStream_t st);
bool ds_conv_ws_eligible(const ConvArgs& a);   // conv3x3_ws.hip: weight-stationary 64 -> 64 bf16 kernel
int ds_launch_conv_ws(const ConvArgs& a, hipStream_t st);
bool ds_conv_thin_eligible(const ConvArgs& a);   // conv3x3_ws.hip: the 8 -> 64 first layer
int ds_launch_conv_thin(const ConvArgs& a, hipStream_t st);
bool ds_conv_thin_out_eligible(const ConvArgs& a);   // conv3x3_ws.hip: the <= 8-cout pyramid heads
int ds_launch_conv_thin_out(const ConvArgs& a, hipStream_t st);
bool ds_conv_small_eligible(const ConvArgs& a);  // conv3x3_small.hip: <= 16-row images, 16-cout slabs, bf16
int ds_launch_conv_small(const ConvArgs& a, hipStream_t st);

// attn_fused.hip: AttnBlockpp as one kernel (16-bit storage, 128 channels, <= 256 pixels per sample).  Weights in the
// fragment-major order of ds_rw_frag_index(row, 0, column, 1, 128): wv / wo = NIN_2 / NIN_3 as [out][in], wqk = Wk^T Wq (the
// query and key projections folded: rows = key-side input channel, columns = query-side input channel), bqk = Wk^T b_q;
// x: [B][L][ldx]; GroupNorm of x from its producer's accumulators (gn_acc) or from scale / shift arrays [B][C].
struct AttnFusedArgs {
  const void* x; long x_bs; int ldx;
  const long long* gn_acc; const float* gn_gamma; const float* gn_beta; int gn_groups; float gn_inv_count; float gn_eps;
  const float* gn_scale; const float* gn_shift;
  const void* wqk; const void* wv; const void* wo;
  const float* bqk; const float* bv; const float* bo;
  void* y; long y_bs; int ldy;
  long long* stats;  // [B][C][2] accumulators of the output (nullable)
  int B, L, C;
};
bool ds_attn_fused_eligible(int dtype, int channels, int L);
int ds_launch_attn_fused(const AttnFusedArgs& a, hipStream_t st);

// GroupNorm: stats -> per-(b,c) scale/shift -> apply(+SiLU)(+FIR resample)
// ws layout: doubles [B][nblk][C][2] then floats scale[B][C], shift[B][C]
long ds_gn_workspace_bytes(int B, int H, int W, int C);
// x2 != null: channels [C1, C) are read from x2 (pixel stride ldx2) — statistics of cat([x, x2]) in place.
int ds_launch_gn_stats(const void* x, int ldx, const void* x2, int ldx2, int C1, int B, int H, int W, int C, int groups,
                       float eps, const float* gamma, const float* beta, void* ws, float* scale, float* shift,
                       int dtype, hipStream_t st);
// scale/shift from the channel-sum accumulators filled by the conv epilogues (two sources = concat view)
int ds_launch_gn_finalize_acc(const long long* a1, int C1, const long long* a2, int C2, int B, long npix, int groups,
                              float eps, const float* gamma, const float* beta, float* scale, float* shift,
                              hipStream_t st);
// mode: 0 none, 1 up, 2 down. scale/shift null => identity & no activation (pure FIR on x -> xr only).
int ds_launch_gn_apply(const void* x, int ldx, const float* scale, const float* shift, int C, void* y, int ldy,
                       void* xr, int ldxr, int B, int H, int W, int act, int mode, int dtype, hipStream_t st);
int ds_launch_concat(const void* a, int lda, int Ca, const void* b, int ldb, int Cb, void* y, int ldy, long npix,
                     int dtype, hipStream_t st);
int ds_launch_softmax(const void* x, void* y, long rows, int L, int ld, int dtype, hipStream_t st);

long ds_stft_workspace_bytes(int B, int S, long T, int n_fft, int hop);
long ds_istft_workspace_bytes(int B, int S, long T, int n_fft, int hop);
int ds_launch_stft_pack(const float* xt, const float* mix, void* y, int B, int S, long T, int n_fft, int hop,
                        float exponent, float factor, int W, int Cpad, int shift, int dtype, const float* tab,
                        float* ws, hipStream_t st, int split = 0);  // split: the DFT GEMM in split mode (ConvArgs.split)
int ds_launch_istft(const void* x, float* out, int B, int S, long T, int n_fft, int hop, float exponent, float factor,
                    int W, int Cpad, int dtype, const float* tab, float* ws, hipStream_t st, int split = 0,
                    const float* ow = nullptr, const float* ob = nullptr, const float* tdiv = nullptr, int ow_cin = 0);
// (ow [2S][ow_cin], ob [2S], tdiv [B]: the network's output layer fused into the unpack pass; x is then the last pyramid)
// tab: device floats: cos | sin | hann (n_fft each), then the [512][512] forward and inverse(+window) DFT matrices
int ds_build_stft_table(int n_fft, float** dev_tab);

struct SdeP { int kind; int ndim; float d_lambda, sigma_min, sigma_max; };
// smix: per-sample sigma_mix [B][T] of PriorMixSDE (kind 1), null for MixSDE (kind 0)
int ds_launch_sigma_mix(const float* mix, float* out, int B, long T, int avg_len, hipStream_t st);
// lens (nullable): per-utterance lengths [B] (device); the state beyond lens[b] is kept at zero
int ds_launch_sde_prior(const SdeP& s, const float* y, const float* z, float* x, int B, int S, long T,
                        const float* smix, hipStream_t st, const int* lens = nullptr);
int ds_launch_sde_corrector(const SdeP& s, float snr, const float* x, const float* t, const float* score,
                            const float* z, float* xo, float* xm, int B, int S, long T, const float* smix,
                            int variant, hipStream_t st, const int* lens = nullptr);  // variant 0 = ald2, 1 = ald
int ds_launch_sde_predictor(const SdeP& s, int N, const float* x, const float* t, const float* score, const float* z,
                            float* xo, float* xm, int B, int S, long T, const float* smix, int pflow, hipStream_t st,
                            const int* lens = nullptr);
// the SDE object surface (sde / marginal_prob / mult_std / discretize / reverse) as unit kernels
int ds_launch_sde_coeff(const SdeP& s, const float* x, const float* t, const float* smix, float* drift,
                        float* diffusion, int B, int S, long T, float fs, float gs, hipStream_t st);
int ds_launch_sde_mean(const SdeP& s, const float* x0, const float* t, float* out, int B, int S, long T, hipStream_t st);
int ds_launch_sde_std(const SdeP& s, const float* t, const float* smix, float* L, int B, int S, long T, hipStream_t st);
int ds_launch_sde_mult_std(const float* L, const float* x, float* out, int B, int S, long T, int per_t, hipStream_t st);
int ds_launch_sde_reverse(const float* f, const float* G, const float* score, float* out, int B, long n_per_batch,
                          int g_full, int pflow, hipStream_t st);
int ds_launch_randn_batch(float* out, int B, int S, long T, const uint64_t* seeds, const int* lens, uint64_t stream_id,
                          hipStream_t st);
// Langevin corrector step; ws >= 16*B + 16 bytes
int ds_launch_langevin(float snr, const float* x, const float* score, const float* z, float* xo, float* xm, int B,
                       long n_per_batch, void* ws, hipStream_t st);
int ds_launch_normalize(const float* mix, float* out, float* mean, float* std, int B, long T, hipStream_t st);
int ds_launch_scale_output(const float* mix, float* sep, int B, int S, long T, hipStream_t st);
int ds_launch_randn(float* out, long n, uint64_t seed, uint64_t stream_id, hipStream_t st);
int ds_launch_gram(const float* ref, const float* est, double* out, int B, int S, long T, hipStream_t st);
int ds_launch_convert(const void* src, void* dst, long n, int sd, int dd, hipStream_t st);
int ds_launch_fill(float* p, float v, long n, hipStream_t st);

// time embedding: y[b][o] = sum_k act_in(x[b][k]) * W[o][k] + bias[o]   (fp32)
int ds_launch_linear(const float* x, const float* W, const float* bias, float* y, int B, int K, int O, int silu_in,
                     hipStream_t st);
// wide outputs, weights transposed [K][O] (ds_launch_dense_transpose builds that layout block by block)
int ds_launch_linear_t(const float* x, const float* Wt, const float* bias, float* y, int B, int K, int O, int silu_in,
                       hipStream_t st);
int ds_launch_dense_transpose(const float* src, float* dst, int O, int K, int ldo, int off, hipStream_t st);
int ds_launch_fourier(const float* t, const float* Wf, float* emb, int B, int nf, hipStream_t st);
