// conv_mfma.hip — implicit-GEMM convolution / batched NT-GEMM on the gfx950 matrix cores.
//
// One kernel template covers every dense contraction on the hot path:
//   * conv3x3 (pad 1)      layers.py:141-156  (ddpm_conv3x3)             TAPS = 9, spatial halo tile
//   * conv1x1 / NIN        layers.py:112-119, 678-689                    TAPS = 1, rows = pixels
//   * attention Q K^T, P V layerspp.py:83-87 (einsum)                    TAPS = 1, batched "weights"
//
//   Y[b, m, n] = ( sum_{tap,k} f(X[b, m (+) tap, k]) * Wt[b?, n, tap, k] / div_b[b] + bias + res ) * out_scale
//
// f is the identity or the fused GroupNorm-apply + SiLU of the producer-side normalisation
// (f(x) = silu(x * scale[b,k] + shift[b,k]), layerspp.py:292,313) evaluated while the tile is written
// to LDS, so act(GN(x)) never exists in HBM.  X may be the channel concatenation of two tensors
// (torch.cat([h, skip], 1), ncsnpp.py:411) read in place.
//
// Layout: X is NHWC with pixel stride ld (k contiguous), Wt is [n][tap][k] (k contiguous) — both
// operands are "K-major", so every MFMA fragment is one 16-byte LDS read:
//   lane l -> row/col (l & 31), k-half h = l >> 5 reads bytes [(kb*2 + h)*16, +16) of its LDS row.
//   bf16: those 8 values are exactly the A/B fragment of v_mfma_f32_32x32x16_bf16 (k = h*8 + j).
//   f32 : the 4 values feed 4 x v_mfma_f32_32x32x2_f32 (k-slot h of MFMA s is channel kb*8 + h*4 + s);
//         f32 MFMA is bit-for-bit an fmaf chain, so the parity path keeps exact fp32 products.
// LDS rows are padded by 16 B (row stride 80 or 144 B): any 16 consecutive rows hit 16 distinct
// 16-byte bank slots, so ds_read_b128 is conflict-free for the 32 consecutive pixels of a fragment.
// Block = 256 threads = 4 waves (one per SIMD); wave tile = (32*WM) x (32*WN), fp32 accumulators.
// Pipeline: the next K-chunk's global loads are issued into registers before the MFMA loop of the
// current chunk and written to LDS after it (issue-early / write-late), so HBM/L2 latency hides
// under the matrix work.  Epilogue: accumulators go through LDS (fp32) and leave as 16-byte,
// channel-contiguous stores with coalesced residual reads.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

#ifdef CONV_TIMING  // profiling build only (tools/conv_timing.sh): per-phase cycle totals of wave 0 of every block
__device__ unsigned long long g_conv_dbg[16];
#define CT_DECL unsigned long long ct_prev = __builtin_readcyclecounter(), ct_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CT_MARK(i) { unsigned long long ct_now = __builtin_readcyclecounter(); ct_acc[i] += ct_now - ct_prev; ct_prev = ct_now; }
#define CT_WAIT __builtin_amdgcn_s_waitcnt(0);
#define CT_FLUSH if (threadIdx.x == 0) { for (int q = 0; q < 12; ++q) atomicAdd(&g_conv_dbg[q], ct_acc[q]); atomicAdd(&g_conv_dbg[15], 1ull); }
extern "C" int diffsep_debug_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_dbg), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_dbg), z, sizeof(z)); }
  return 0;
}
#else
#define CT_DECL
#define CT_MARK(i)
#define CT_WAIT
#define CT_FLUSH
#endif

template <typename T> struct Mma;
template <> struct Mma<float> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {  // the engine's 16-bit storage format (bfloat16, or half precision with -DDS_HALF_F16)
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
#ifdef ABL_NOMFMA
    c[0] += __uint_as_float(a.x ^ b.x);
    return;
#endif
    c = mfma_h32(a, b, c);
  }
};
struct MmaBf {  // true bfloat16 operands: the hi / lo planes of the split mode
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) { c = mfma_bf32(a, b, c); }
};

// fp32 value = hi + lo with hi, lo bf16 (|error| <= 2^-17 |v|): the operands of the "split" mode, in which an fp32 conv
// runs as three bf16 MFMAs per k-block (hi*hi + hi*lo + lo*hi, fp32 accumulation) instead of eight fp32 MFMAs
__device__ inline void split4(const uint4& v, uint2& hi, uint2& lo) {
  const float f0 = __uint_as_float(v.x), f1 = __uint_as_float(v.y), f2 = __uint_as_float(v.z), f3 = __uint_as_float(v.w);
  hi.x = pack_bf16x2(f0, f1);
  hi.y = pack_bf16x2(f2, f3);
  lo.x = pack_bf16x2(f0 - bf_lo(hi.x), f1 - bf_hi(hi.x));
  lo.y = pack_bf16x2(f2 - bf_lo(hi.y), f3 - bf_hi(hi.y));
}

// GN-apply (+SiLU) on one 16-byte vector of KV channels
template <typename T> struct GnVec;
template <> struct GnVec<float> {
  template <bool ACT>
  __device__ static inline uint4 run(const uint4& u, const float* sc, const float* sh) {
    float f[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = f[j] * sc[j] + sh[j];
      f[j] = ACT ? silu_t<float>(v) : v;
    }
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct GnVec<bf16_t> {
  template <bool ACT>
  __device__ static inline uint4 run(const uint4& u, const float* sc, const float* sh) {
    float f[8];
    f[0] = h_lo(u.x); f[1] = h_hi(u.x);
    f[2] = h_lo(u.y); f[3] = h_hi(u.y);
    f[4] = h_lo(u.z); f[5] = h_hi(u.z);
    f[6] = h_lo(u.w); f[7] = h_hi(u.w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = f[j] * sc[j] + sh[j];
      f[j] = ACT ? silu_t<bf16_t>(v) : v;
    }
    uint4 o;
    o.x = pack_h2(f[0], f[1]);
    o.y = pack_h2(f[2], f[3]);
    o.z = pack_h2(f[4], f[5]);
    o.w = pack_h2(f[6], f[7]);
    return o;
  }
};

// 8 channels held as raw 16-byte vectors -> floats
template <typename T> __device__ inline void unpack8(const uint4* u, float* f);
template <> __device__ inline void unpack8<float>(const uint4* u, float* f) {
  f[0] = __uint_as_float(u[0].x); f[1] = __uint_as_float(u[0].y); f[2] = __uint_as_float(u[0].z);
  f[3] = __uint_as_float(u[0].w); f[4] = __uint_as_float(u[1].x); f[5] = __uint_as_float(u[1].y);
  f[6] = __uint_as_float(u[1].z); f[7] = __uint_as_float(u[1].w);
}
template <> __device__ inline void unpack8<bf16_t>(const uint4* u, float* f) {
  f[0] = h_lo(u[0].x); f[1] = h_hi(u[0].x);
  f[2] = h_lo(u[0].y); f[3] = h_hi(u[0].y);
  f[4] = h_lo(u[0].z); f[5] = h_hi(u[0].z);
  f[6] = h_lo(u[0].w); f[7] = h_hi(u[0].w);
}

template <typename T> __device__ inline void pack8(const float* f, uint4* u);
template <> __device__ inline void pack8<float>(const float* f, uint4* u) {
  u[0] = make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  u[1] = make_uint4(__float_as_uint(f[4]), __float_as_uint(f[5]), __float_as_uint(f[6]), __float_as_uint(f[7]));
}
template <> __device__ inline void pack8<bf16_t>(const float* f, uint4* u) {
  u[0] = make_uint4(pack_h2(f[0], f[1]), pack_h2(f[2], f[3]), pack_h2(f[4], f[5]), pack_h2(f[6], f[7]));
}

struct ConvK {  // kernel-side copy of ConvArgs (typed by the template)
  const void* x; long x_bs; int ldx; int C1;
  const void* x2; long x2_bs; int ldx2;
  const void* w; long w_bs; int w_chunked; int w_shift;   // w_shift = log2(w_chunked): no integer division in the kernel
  // fused 1x1 skip convolution (ResnetBlockBigGANpp Conv_2): extra K chunks with one tap, raw input
  const void* sx; long sx_bs; int ldsx; const void* sx2; long sx2_bs; int ldsx2; int sC1; int sCin;
  const void* sw; int sw_chunked; int sw_shift;
  const float* gn_scale; const float* gn_shift; int gn_act;
  // GroupNorm from channel-sum accumulators of the producer(s) (fixed point, common.h): the block turns them
  // into its per-channel scale / shift table while its first loads are in flight
  const long long* gn_acc1; const long long* gn_acc2; const float* gn_gamma; const float* gn_beta;
  int gn_groups; float gn_inv_count; float gn_eps;
  const float* bias; const float* bias_b; int bias_b_ld; int bias_mode;
  const float* div_b;
  const void* res; long res_bs; int ldr;
  float out_scale;
  void* y; long y_bs; int ldy;
  long long* stats;  // optional [B][Cout][2] fixed-point accumulators: sum / sum of squares of the OUTPUT channels
  int H, W, Cin, Cout;
  int tiles_x;
};

template <typename T, int TAPS, int TH, int TW, int BN, int KC, int EP = 1, int NT = 256>
struct ConvGeom {
  static constexpr int KV = 16 / (int)sizeof(T);
  static constexpr int R = (TAPS == 9) ? 1 : 0;
  static constexpr int HW_ = TW + 2 * R, HH_ = TH + 2 * R, HP = HW_ * HH_;
  // LDS row length of the halo tile (pixels).  A wave's 32 fragment pixels are one tile row when TW = 32, two when TW = 16,
  // four when TW = 8; ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... in one pass each, and with the
  // 80 / 144-byte pitch those 16 rows fall into 16 distinct 16-byte bank slots only if the tile rows start 32 (TW = 16) / 24
  // (TW = 8) LDS rows apart (exhaustive search; unpadded, the half-width tile of the 32^2 level had 2 of 16 lanes conflicting
  // in every pass: 2.8 conflict cycles per LDS instruction in round 3's counters)
  static constexpr int HWP = (TAPS == 9 && TW == 16) ? 32 : ((TAPS == 9 && TW == 8) ? 24 : HW_);
  static constexpr int BM = TH * TW;
  static constexpr int ROWB = KC * (int)sizeof(T) + 16;
  static constexpr int NVEC = KC / KV;
  static constexpr int NKB = KC / (2 * KV);
  static constexpr int NA = (HP * NVEC + NT - 1) / NT;
  static constexpr int NB = (TAPS * BN * NVEC + NT - 1) / NT;
  static constexpr int OROW = BN * 4 + 16;  // fp32 output staging row pitch
  static constexpr int LDS_A = HH_ * HWP * ROWB;   // the halo tile
  static constexpr int LDS_STAGE = LDS_A + TAPS * BN * ROWB;
  static constexpr int LDS_OUT = (BM / EP) * OROW;  // the epilogue streams the tile out in EP passes
  static constexpr int LDS = LDS_STAGE > LDS_OUT ? LDS_STAGE : LDS_OUT;
  // fused 1x1 skip convolution: needs one weight staging pass per tap (256 / NVEC rows per pass == BN)
  static constexpr bool SKIP_OK = TAPS == 9 && (NT / NVEC) <= BN && BN % (NT / NVEC) == 0;
};

// ---- buffer addressing (SRSRC): a wave-uniform descriptor + a 32-bit per-lane byte offset + a uniform
// scalar offset.  Lanes that must not touch memory get an offset >= num_records: the hardware returns 0 for
// such loads and drops such stores, so the halo zero padding, ragged tiles and channel tails cost no
// branches, no exec masking and no zero-initialisation.  The per-lane offsets are loop invariant: inside the
// K loop a load is ONE instruction (the chunk's channel offset travels in the scalar offset).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define DS_OOB 0x80000000u
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
#ifndef DS_A_AUX
#define DS_A_AUX 0  // cache-policy bits of the activation tile loads (experiment: 2 = nt)
#endif
__device__ inline uint4 buf_load16a(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, DS_A_AUX);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline void buf_store16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const uint4& d) {
  u32x4_t v = {d.x, d.y, d.z, d.w};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

// SP (fp32 only): split mode.  The staged chunk is written to LDS as two bf16 planes per row ([KC hi][KC lo], the same
// KC * 4 bytes as fp32) and every k-block of 16 channels is three bf16 MFMAs.
template <typename T, int TAPS, int TH, int TW, int BN, int WM, int WN, int KC, int EP = 1, int OCC = 2, int SP = 0, int NT = 256>
__global__ __launch_bounds__(NT, OCC) void conv_mfma_kernel(ConvK p) {
  using G = ConvGeom<T, TAPS, TH, TW, BN, KC, EP, NT>;
  static_assert(EP == 1 || (WM % EP == 0), "epilogue passes split the wave's M blocks");
  static_assert(SP == 0 || (sizeof(T) == 4 && KC % 16 == 0), "split mode: fp32 storage, whole 16-channel k-blocks");
  constexpr int KV = G::KV, R = G::R, HW_ = G::HW_, HWP = G::HWP, HP = G::HP, BM = G::BM, ROWB = G::ROWB, NVEC = G::NVEC,
                NKB = SP ? KC / 16 : G::NKB, NA = G::NA, NB = G::NB, OROW = G::OROW;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int WAVES_N = BN / (32 * WN);
  constexpr int WAVES_M = BM / (32 * WM);
  static_assert(WAVES_M * WAVES_N == NT / 64, "the wave tiles cover the block tile");
  static_assert(KC % (2 * KV) == 0, "KC must hold whole k-blocks");
  static_assert(NT % NVEC == 0, "a thread keeps one channel offset across its vectors");

  CT_DECL
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + G::LDS_A;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int b = blockIdx.z;
  const int n0 = blockIdx.y * BN;
  int y0 = 0, x0 = 0;
  int m0 = 0;
  const int M = p.H * p.W;
  if (TAPS == 9) {
    y0 = (blockIdx.x / p.tiles_x) * TH;
    x0 = (blockIdx.x % p.tiles_x) * TW;
  } else {
    m0 = blockIdx.x * BM;
  }

  const bool has_gn = p.gn_scale != nullptr || p.gn_acc1 != nullptr;
  float* sGN = reinterpret_cast<float*>(smem + G::LDS);  // [Cin] scale, [Cin] shift (accumulator mode)
  const int C1 = p.C1, C2 = p.Cin - p.C1;
  const __amdgpu_buffer_rsrc_t rx1 =
      make_rsrc(reinterpret_cast<const T*>(p.x) + (long)b * p.x_bs, (unsigned)M * p.ldx * ESZ);
  const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(
      p.x2 ? reinterpret_cast<const T*>(p.x2) + (long)b * p.x2_bs : reinterpret_cast<const T*>(p.x), (unsigned)M * (p.x2 ? p.ldx2 : p.ldx) * ESZ);
  const __amdgpu_buffer_rsrc_t rw =
      make_rsrc(reinterpret_cast<const T*>(p.w) + (long)b * p.w_bs, (unsigned)p.Cout * TAPS * p.Cin * ESZ);

  // ---- per-thread staging descriptors.  Vector i = tid + 256 k of a stage: row (pixel / weight row)
  // row0 + k*RPS, 16-byte slot tid % NVEC: LDS offsets are linear in k (immediates), global byte offsets
  // are computed once (pixels: one per source because the pixel strides differ).
  constexpr int RPS = NT / NVEC;  // rows covered by one pass of the block
  constexpr bool B_TAPSTEP = (RPS % BN == 0);
  static_assert(B_TAPSTEP || BN % RPS == 0, "a pass of the block covers whole taps or a whole fraction of one");
  const int vch = (tid % NVEC) * KV;  // channel offset of this thread's vectors inside a chunk
  const int row0 = tid / NVEC;
  const int lds0 = row0 * ROWB + (tid % NVEC) * 16;  // + k * RPS * ROWB
  auto a_in = [&](int k) __attribute__((always_inline)) { return row0 + k * RPS < HP; };
  auto b_in = [&](int k) __attribute__((always_inline)) { return row0 + k * RPS < TAPS * BN; };
  // pixel index of the thread's halo vectors (-1: outside the image).  The byte offset in a source is
  // (pixi * ld + vch) * ESZ; for pixi = -1 that is negative = far beyond num_records as unsigned: reads zero.
  int pixi_[NA];
  // LDS byte offset of the thread's halo vectors (tile rows are HWP pixels apart there); linear in k — a compile-time
  // immediate, no registers — when the LDS rows carry no padding (TW = 32, 1x1)
  constexpr bool LDSA_LIN = TAPS != 9 || HWP == HW_;
  int ldsa_[LDSA_LIN ? 1 : NA];
  bool aval[NA];
  bool ain[NA];  // the vector belongs to the tile proper (not its halo): all the fused 1x1 skip conv needs
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    int pixi = -1;
    if (a_in(k)) {
      const int pix = row0 + k * RPS;
      if (TAPS == 9) {
        const int hy = pix / HW_, hx = pix - hy * HW_;
        const int gy = y0 + hy - R, gx = x0 + hx - R;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) pixi = gy * p.W + gx;
      } else {
        if (m0 + pix < M) pixi = m0 + pix;
      }
    }
    aval[k] = pixi >= 0;
    pixi_[k] = pixi;
    if constexpr (!LDSA_LIN) {
      const int pix = row0 + k * RPS;
      const int hy = TAPS == 9 ? pix / HW_ : 0, hx = TAPS == 9 ? pix - hy * HW_ : pix;
      ldsa_[k] = (hy * HWP + hx) * ROWB;
    }
    {
      const int pix = row0 + k * RPS;
      const int hy = pix / HW_, hx = pix - hy * HW_;
      ain[k] = TAPS != 9 || (hy >= R && hy < R + TH && hx >= R && hx < R + TW);
    }
  }
  auto voa = [&](int k, int ld) __attribute__((always_inline)) { return (unsigned)((pixi_[k] * ld + vch) * ESZ); };
  // byte offsets of the thread's weight vectors (row = cout, tap): DS_OOB past Cout.  When a pass of the block covers whole
  // taps the offsets are base + k * step (one register + a scalar instead of NB registers: the 8 x 32 x 64 tile sat at 256
  // VGPRs with 5 of them in scratch memory, and a scratch reload next to in-flight loads is a full vmcnt(0))
  constexpr bool VOB_LIN = B_TAPSTEP;
  unsigned vob[VOB_LIN ? 1 : NB];
  unsigned vob_step = 0;
  int vtap0 = 0;
  bool vcol_ok = false;
  if constexpr (VOB_LIN) {
    const int col = row0 % BN;
    vtap0 = row0 / BN;
    vcol_ok = n0 + col < p.Cout;
    vob[0] = p.w_chunked ? (unsigned)((((vch >> p.w_shift) * TAPS + vtap0) * p.Cout + n0 + col) * p.w_chunked +
                                      (vch & (p.w_chunked - 1))) * ESZ
                         : (unsigned)(((n0 + col) * TAPS + vtap0) * p.Cin + vch) * ESZ;
    vob_step = (unsigned)(RPS / BN) * (p.w_chunked ? (unsigned)(p.Cout * p.w_chunked) : (unsigned)p.Cin) * ESZ;
  }
#pragma unroll
  for (int k = 0; k < (VOB_LIN ? 0 : NB); ++k) {
    int col, tap;
    if (B_TAPSTEP) {  // a pass covers whole taps
      col = row0 % BN;
      tap = row0 / BN + k * (RPS / BN);
    } else {          // BN / RPS passes per tap
      constexpr int Q = BN / RPS;
      col = row0 + (k % Q) * RPS;
      tap = k / Q;
    }
    const bool ok = b_in(k) && n0 + col < p.Cout && tap < TAPS;
    // weights: [Cout][tap][Cin] or, chunk-major, [Cin / KC][tap][Cout][KC] (a stage's rows are then contiguous:
    // full 128-byte lines per request instead of 64-byte pieces)
    vob[k] = !ok ? DS_OOB
                 : p.w_chunked ? (unsigned)((((vch >> p.w_shift) * TAPS + tap) * p.Cout + n0 + col) * p.w_chunked +
                                            (vch & (p.w_chunked - 1))) * ESZ
                               : (unsigned)(((n0 + col) * TAPS + tap) * p.Cin + vch) * ESZ;
  }
  auto vobk = [&](int k) __attribute__((always_inline)) {
    if constexpr (VOB_LIN) return (vcol_ok && vtap0 + k * (RPS / BN) < TAPS) ? vob[0] + (unsigned)k * vob_step : DS_OOB;
    else return vob[k];
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane LDS byte offsets of the fragment rows
  int aoff[WM], boff[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int pp = (wm * WM + i) * 32 + l32;
    const int row = (TAPS == 9) ? ((pp / TW) * HWP + (pp % TW)) : pp;
    aoff[i] = row * ROWB + h * 16;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) boff[j] = ((wn * WN + j) * 32 + l32) * ROWB + h * 16;

  uint4 pa[NA], pb[NB];
  float gsc[KV], gsh[KV];
  bool ch_ok = true;

  // K is walked source by source so that a chunk never straddles the concat seam: chunks [0, nch1) read
  // channels of x, chunks [nch1, nch) channels of x2 (weights follow the concatenated channel index).
  const int nch1 = (C1 + KC - 1) / KC;
  const int nch = nch1 + (C2 + KC - 1) / KC;
  // fused skip convolution: chunks [nch, ncht) carry the raw block input through ONE tap (the centre)
  const int sC1 = p.sC1, sC2 = p.sCin - p.sC1;
  const int nchs1 = p.sx ? (sC1 + KC - 1) / KC : 0;
  const int ncht = nch + nchs1 + (p.sx ? (sC2 + KC - 1) / KC : 0);
  // (supported by the instantiations whose weight staging makes one pass per tap; the launcher checks)
  constexpr bool SKIP_OK = G::SKIP_OK;
  constexpr int QS = SKIP_OK ? BN / RPS : 1;   // staging passes per tap
  constexpr int KSKIP = SKIP_OK ? 4 * QS : 0;  // first staging pass of the centre tap
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    if (SKIP_OK && c >= nch) {  // skip chunk: raw input, centre-tap weights only
      const int cs = c - nch;
      const bool second = cs >= nchs1;
      const int cb = (second ? cs - nchs1 : cs) * KC;
      const int width = (second ? sC2 : sC1) - cb;
      const int wb = second ? sC1 + cb : cb;
      ch_ok = vch < width;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(
          second ? reinterpret_cast<const T*>(p.sx2) + (long)b * p.sx2_bs : reinterpret_cast<const T*>(p.sx) + (long)b * p.sx_bs,
          (unsigned)M * (second ? p.ldsx2 : p.ldsx) * ESZ);
      const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.sw, (unsigned)p.Cout * p.sCin * ESZ);
      const int lds_ = second ? p.ldsx2 : p.ldsx;
      const unsigned so = (unsigned)cb * ESZ;
#pragma unroll
      for (int k = 0; k < NA; ++k) pa[k] = buf_load16(rs, (ch_ok && ain[k]) ? voa(k, lds_) : DS_OOB, so);
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        const int col = row0 + q * RPS;  // (RPS == BN: row0 < BN)
        const bool okw = ch_ok && col < BN && n0 + col < p.Cout;
        const unsigned vo = p.sw_chunked ? (unsigned)((((wb + vch) >> p.sw_shift) * p.Cout + n0 + col) * p.sw_chunked +
                                                      ((wb + vch) & (p.sw_chunked - 1))) * ESZ
                                         : (unsigned)((n0 + col) * p.sCin + wb + vch) * ESZ;
        pb[KSKIP + q] = buf_load16(rsw, okw ? vo : DS_OOB, 0);
      }
      return;
    }
    const bool second = c >= nch1;
    const int cb = (second ? c - nch1 : c) * KC;        // channel offset inside the source
    const int width = (second ? C2 : C1) - cb;           // channels left in the source (>= 1)
    const int wb = second ? C1 + cb : cb;                // channel offset inside the weights / GN tables
    ch_ok = vch < width;
    const unsigned so = (unsigned)cb * ESZ;
    const unsigned sw = p.w_chunked ? (unsigned)(wb >> p.w_shift) * (unsigned)(TAPS * p.Cout * p.w_chunked * ESZ)
                                    : (unsigned)wb * ESZ;
#ifdef ABL_NOLOAD
    return;
#endif
    const int ld_ = second ? p.ldx2 : p.ldx;
    if (width >= KC) {  // uniform fast path
      if (!second) {
#pragma unroll
        for (int k = 0; k < NA; ++k) pa[k] = buf_load16a(rx1, voa(k, ld_), so);
      } else {
#pragma unroll
        for (int k = 0; k < NA; ++k) pa[k] = buf_load16a(rx2, voa(k, ld_), so);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) pb[k] = buf_load16(rw, vobk(k), sw);
    } else {  // channel tail of a source: lanes past the end read zeros
      if (!second) {
#pragma unroll
        for (int k = 0; k < NA; ++k) pa[k] = buf_load16(rx1, ch_ok ? voa(k, ld_) : DS_OOB, so);
      } else {
#pragma unroll
        for (int k = 0; k < NA; ++k) pa[k] = buf_load16(rx2, ch_ok ? voa(k, ld_) : DS_OOB, so);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) pb[k] = buf_load16(rw, ch_ok ? vobk(k) : DS_OOB, sw);
    }
  };
  auto load_gn = [&](int c) __attribute__((always_inline)) {  // scale / shift of this thread's KV channels of chunk c
    const bool second = c >= nch1;
    const int cb = (second ? c - nch1 : c) * KC;
    const int wb = second ? C1 + cb : cb;
    if (has_gn && vch < (second ? C2 : C1) - cb) {
      const float4* ps = p.gn_acc1 ? reinterpret_cast<const float4*>(sGN + wb + vch)
                                   : reinterpret_cast<const float4*>(p.gn_scale + (long)b * p.Cin + wb + vch);
      const float4* ph = p.gn_acc1 ? reinterpret_cast<const float4*>(sGN + p.Cin + wb + vch)
                                   : reinterpret_cast<const float4*>(p.gn_shift + (long)b * p.Cin + wb + vch);
#pragma unroll
      for (int j = 0; j < KV / 4; ++j) {
        const float4 a = ps[j], cc = ph[j];
        gsc[4 * j] = a.x; gsc[4 * j + 1] = a.y; gsc[4 * j + 2] = a.z; gsc[4 * j + 3] = a.w;
        gsh[4 * j] = cc.x; gsh[4 * j + 1] = cc.y; gsh[4 * j + 2] = cc.z; gsh[4 * j + 3] = cc.w;
      }
    }
  };
  // accumulator mode: per-channel scale = rstd * gamma, shift = beta - mean * scale of image b, into LDS
  auto build_gn_table = [&]() __attribute__((always_inline)) {
    // one global round trip: every thread fetches the sums of ITS channel into LDS (the staging area is still
    // unused), then the group totals come from LDS
    long long* tmp = reinterpret_cast<long long*>(smem);  // [Cin][2]
    for (int c = tid; c < p.Cin; c += NT) {
      const long long* src = c < C1 ? p.gn_acc1 + ((long)b * C1 + c) * 2 : p.gn_acc2 + ((long)b * C2 + (c - C1)) * 2;
      const longlong2 v = *reinterpret_cast<const longlong2*>(src);
      tmp[2 * c] = v.x;
      tmp[2 * c + 1] = v.y;
    }
    __syncthreads();
    const int cpg = p.Cin / p.gn_groups;
    for (int c = tid; c < p.Cin; c += NT) {
      const int g0 = (c / cpg) * cpg;
      long long ssum = 0, ssq = 0;
      for (int j = 0; j < cpg; ++j) {
        ssum += tmp[2 * (g0 + j)];
        ssq += tmp[2 * (g0 + j) + 1];
      }
      const double mean = (double)ssum * (1.0 / DS_STAT_SUM_SCALE) * (double)p.gn_inv_count;
      double var = (double)ssq * (1.0 / DS_STAT_SQ_SCALE) * (double)p.gn_inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
      const float sc = rstd * (p.gn_gamma ? p.gn_gamma[c] : 1.f);
      sGN[c] = sc;
      sGN[p.Cin + c] = (p.gn_beta ? p.gn_beta[c] : 0.f) - (float)mean * sc;
    }
    __syncthreads();
  };
  // The chunk in flight is activated IN REGISTERS (GN affine + SiLU) while the matrix pipe works on the chunk
  // that is resident in LDS: the activation's VALU is spread over the MFMA loop of the same wave, so between
  // the two barriers of a chunk only the LDS writes remain.
  auto put = [&](char* dst, const uint4& v) __attribute__((always_inline)) {
    if constexpr (SP) {  // row = [KC bf16 hi][KC bf16 lo]; this thread's 4 channels: 8 bytes in each plane
      uint2 hi, lo;
      split4(v, hi, lo);
      *reinterpret_cast<uint2*>(dst) = hi;
      *reinterpret_cast<uint2*>(dst + KC * 2) = lo;
    } else {
      *reinterpret_cast<uint4*>(dst) = v;
    }
  };
  const int ldsw0 = SP ? row0 * ROWB + (tid % NVEC) * 8 : lds0;
  auto write_chunk = [&](bool skip) __attribute__((always_inline)) {
#ifdef ABL_NOLDSW
    return;
#endif
#pragma unroll
    for (int k = 0; k < NA; ++k)
      if (a_in(k)) put(LDSA_LIN ? sA + ldsw0 + k * RPS * ROWB : sA + ldsa_[LDSA_LIN ? 0 : k] + ldsw0 - row0 * ROWB, pa[k]);
    if (skip) {  // only the centre tap's weight rows exist (and only they are read)
#pragma unroll
      for (int q = 0; q < QS; ++q) put(sB + ldsw0 + (KSKIP + q) * RPS * ROWB, pb[KSKIP + q]);
      return;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (b_in(k)) put(sB + ldsw0 + k * RPS * ROWB, pb[k]);
  };
  constexpr int SLOTS = TAPS * NKB;                 // k-blocks of one chunk
  constexpr int S0 = SLOTS / 3;                     // the loads of the next chunk get this long to land
  auto run_chunks = [&](auto MODE_) __attribute__((always_inline)) {               // 0: raw input, 1: GN affine, 2: GN affine + SiLU
    constexpr int MODE = decltype(MODE_)::value;
    auto act = [&](int k) __attribute__((always_inline)) {  // branch free: zero padding (outside pixels, channel tails) keeps its loaded zeros
#ifdef ABL_NOACT
      return;
#endif
      if constexpr (MODE != 0) {
        uint4 r = GnVec<T>::template run<MODE == 2>(pa[k], gsc, gsh);
        // keep the arithmetic unconditional (a branch here would cut the MFMA loop into pieces)
        asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
        const bool ok = aval[k] && ch_ok;
        pa[k].x = ok ? r.x : pa[k].x;
        pa[k].y = ok ? r.y : pa[k].y;
        pa[k].z = ok ? r.z : pa[k].z;
        pa[k].w = ok ? r.w : pa[k].w;
      }
    };
    auto mma_chunk = [&](auto NEXT_, auto SKIP_) __attribute__((always_inline)) {
      constexpr bool NEXT = decltype(NEXT_)::value && MODE != 0;
      constexpr bool SKIP = decltype(SKIP_)::value;  // fused skip convolution: the centre tap only
#pragma unroll
      for (int tap = (SKIP ? 4 : 0); tap < (SKIP ? 5 : TAPS); ++tap) {
        const int toff = (TAPS == 9) ? ((tap / 3) * HWP + (tap % 3)) * ROWB : 0;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          uint4 af[WM], bfr[WN];
#pragma unroll
          for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const uint4*>(sA + aoff[i] + toff + kb * 32);
#pragma unroll
          for (int j = 0; j < WN; ++j)
            bfr[j] = *reinterpret_cast<const uint4*>(sB + boff[j] + tap * BN * ROWB + kb * 32);
          if constexpr (SP) {
            uint4 afl[WM], bfl[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) afl[i] = *reinterpret_cast<const uint4*>(sA + aoff[i] + toff + kb * 32 + KC * 2);
#pragma unroll
            for (int j = 0; j < WN; ++j)
              bfl[j] = *reinterpret_cast<const uint4*>(sB + boff[j] + tap * BN * ROWB + kb * 32 + KC * 2);
            // term-major: consecutive MFMAs go to different accumulators (small terms first)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
              for (int j = 0; j < WN; ++j) MmaBf::run(bfl[j], af[i], acc[i][j]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
              for (int j = 0; j < WN; ++j) MmaBf::run(bfr[j], afl[i], acc[i][j]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
              for (int j = 0; j < WN; ++j) MmaBf::run(bfr[j], af[i], acc[i][j]);
          } else {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
              for (int j = 0; j < WN; ++j) Mma<T>::run(bfr[j], af[i], acc[i][j]);  // D[cout][pixel]
          }
          if constexpr (NEXT) {
            const int s = tap * NKB + kb;
            if (s >= S0) {
              const int k0 = (s - S0) * NA / (SLOTS - S0), k1 = (s + 1 - S0) * NA / (SLOTS - S0);
#pragma unroll
              for (int k = 0; k < NA; ++k)
                if (k >= k0 && k < k1) act(k);
            }
          }
        }
      }
    };
    CT_MARK(0)
    load_chunk(0);
    if (p.gn_acc1) build_gn_table();  // (behind the first chunk's loads: they are needed next anyway)
    load_gn(0);
    CT_MARK(1)
    CT_WAIT
    CT_MARK(2)
#pragma unroll
    for (int k = 0; k < NA; ++k) act(k);
    CT_MARK(3)
    for (int c = 0; c < nch; ++c) {
      __syncthreads();  // previous chunk's fragment reads are done
      write_chunk(false);
      __syncthreads();
      CT_MARK(4)
      if (c + 1 < nch) {
        load_chunk(c + 1);  // in flight during the first third of the MFMA loop below
        load_gn(c + 1);
        CT_MARK(5)
        mma_chunk(std::true_type{}, std::false_type{});
      } else {
        if (c + 1 < ncht) load_chunk(c + 1);  // first skip chunk (raw: nothing to activate)
        CT_MARK(5)
        mma_chunk(std::false_type{}, std::false_type{});
      }
      CT_MARK(6)
    }
    if constexpr (SKIP_OK) {
      for (int c = nch; c < ncht; ++c) {  // fused 1x1 skip convolution: extra K through the centre tap
        __syncthreads();
        write_chunk(true);
        __syncthreads();
        if (c + 1 < ncht) load_chunk(c + 1);
        mma_chunk(std::false_type{}, std::true_type{});
      }
    }
  };
  if (!has_gn) run_chunks(std::integral_constant<int, 0>{});
  else if (p.gn_act) run_chunks(std::integral_constant<int, 2>{});
  else run_chunks(std::integral_constant<int, 1>{});

  // ---- epilogue: accumulators -> LDS (fp32, [pixel][cout]) -> bias / temb / residual / scale -> 16-byte
  // stores, in EP passes over the wave's M blocks (a smaller staging buffer lets more blocks share a CU).
  const __amdgpu_buffer_rsrc_t ry =
      make_rsrc(reinterpret_cast<T*>(p.y) + (long)b * p.y_bs, (unsigned)M * p.ldy * ESZ);
  const __amdgpu_buffer_rsrc_t rr = make_rsrc(
      p.res ? reinterpret_cast<const T*>(p.res) + (long)b * p.res_bs : reinterpret_cast<const T*>(p.y),
      p.res ? (unsigned)M * p.ldr * ESZ : 0u);  // no residual: zero records -> every load returns 0
#ifdef ABL_NOEPI
  {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 12345.678f) reinterpret_cast<float*>(p.y)[tid] = t;
    return;
  }
#endif
  const int cout8 = (p.Cout + 7) & ~7;
  const float dvs = p.div_b ? p.div_b[b] : 1.0f;
  constexpr int NCG = BN / 8;
  static_assert(NT % NCG == 0, "a thread keeps one cout group across the epilogue loop");
  constexpr int WME = WM / EP;  // M blocks of a wave per pass
  constexpr int RV = ESZ * 8 / 16;  // 16-byte vectors per 8 channels
  const int cg = tid % NCG;
  const int co = n0 + cg * 8;
  // per-thread column bias (conv bias + per-batch temb bias): 16 independent loads issued back to back
  float bv[8];
  {
    float b1[8], b2[8];
    const float* pb1 = (p.bias_mode == 0 && p.bias) ? p.bias : nullptr;
    const float* pb2 = (p.bias_mode == 0 && p.bias_b) ? p.bias_b + (long)b * p.bias_b_ld : nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cc = (co + j < p.Cout) ? co + j : p.Cout - 1;
      b1[j] = pb1 ? pb1[cc] : 0.f;
      b2[j] = pb2 ? pb2[cc] : 0.f;
    }
    // channels past Cout (padding up to a multiple of 8) have zero weights and a zero residual: a zero bias
    // makes them come out as exact zeros without any per-element select
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = (co + j < p.Cout) ? b1[j] + b2[j] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(bv[j]));  // computed once, not rematerialised per row
  }
  // tiles that overhang the image must keep their outside rows out of the statistics
  const bool overhang = (TAPS == 9) ? (y0 + TH > p.H || x0 + TW > p.W) : (m0 + BM > M);
  // the common case (column bias, no division, whole tile inside the image, whole cout groups) runs without any
  // per-element selects; everything else takes the general path
  const bool lean = p.bias_mode == 0 && !p.div_b && !overhang && n0 + BN <= cout8;
  float ssum[8], ssq[8];  // GroupNorm statistics of what this thread writes (consumed by the next GN)
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    __syncthreads();  // fragment reads (e = 0) / the previous pass's reads are done
    // C/D layout of the 32x32 MFMA with the weights as the A operand: lane = pixel l32 of the M block, register
    // quad g = couts 8 g + 4 h .. + 3 of the N block: one 16-byte LDS write per quad
#pragma unroll
    for (int i = 0; i < WME; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int lp = (wm * WME + i) * 32 + l32;
          const int cc = (wn * WN + j) * 32 + 8 * g + 4 * h;
          *reinterpret_cast<float4*>(smem + lp * OROW + cc * 4) =
              make_float4(acc[e * WME + i][j][4 * g], acc[e * WME + i][j][4 * g + 1], acc[e * WME + i][j][4 * g + 2],
                          acc[e * WME + i][j][4 * g + 3]);
        }
    // rows of this thread: pixel index (or -1) -> byte offsets of its 8-channel vector in y / res
    constexpr int NROW = (BM / EP) / (NT / NCG);
    static_assert((BM / EP) % (NT / NCG) == 0, "whole rows per thread");
    int mrow[NROW];
    unsigned voy[NROW], vor[NROW];
    uint4 rraw[NROW][RV];
    // EP == 1: local row lp IS the tile pixel, so successive rows of a thread advance linearly
    // ((256/NCG)/TW image rows, or 256/NCG flat pixels): one add per row instead of div/mod/mul
    constexpr int RSTEP = NT / NCG;
    const int pp0 = tid / NCG;
    const int gy0 = y0 + pp0 / TW, gx0 = x0 + pp0 % TW;
    const int mlin0 = (TAPS == 9) ? gy0 * p.W + gx0 : m0 + pp0;
    const int mstep = (TAPS == 9) ? (RSTEP / TW) * p.W : RSTEP;
    const bool col_ok = co < cout8 && (TAPS != 9 || gx0 < p.W);
#pragma unroll
    for (int it = 0; it < NROW; ++it) {
      int m = -1;
      if (EP == 1 && (TAPS != 9 || RSTEP % TW == 0)) {
        const bool ok = col_ok && ((TAPS == 9) ? (gy0 + it * (RSTEP / TW) < p.H) : (mlin0 + it * mstep < M));
        m = ok ? mlin0 + it * mstep : -1;
      } else {
        const int lp = tid / NCG + it * (NT / NCG);
        const int pp = ((lp / (32 * WME)) * WM + e * WME + (lp / 32) % WME) * 32 + (lp & 31);
        if (TAPS == 9) {
          const int gy = y0 + pp / TW, gx = x0 + pp % TW;
          if (gy < p.H && gx < p.W) m = gy * p.W + gx;
        } else {
          if (m0 + pp < M) m = m0 + pp;
        }
        if (co >= cout8) m = -1;
      }
      mrow[it] = m;
      voy[it] = m >= 0 ? (unsigned)(m * p.ldy + co) * ESZ : DS_OOB;
      vor[it] = m >= 0 ? (unsigned)(m * p.ldr + co) * ESZ : DS_OOB;
#pragma unroll
      for (int q = 0; q < RV; ++q) rraw[it][q] = buf_load16(rr, vor[it], 16u * q);
    }
    __syncthreads();
    CT_MARK(7)
    uint4 oraw[NROW][RV];
    if (lean) {
      const float osc = p.out_scale;
      float bs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bs[j] = bv[j] * osc;
      const bool has_res = p.res != nullptr, has_stats = p.stats != nullptr;
#pragma unroll
      for (int it = 0; it < NROW; ++it) {
        const int lp = tid / NCG + it * (NT / NCG);
        const float4 a0 = *reinterpret_cast<const float4*>(smem + lp * OROW + cg * 32);
        const float4 a1 = *reinterpret_cast<const float4*>(smem + lp * OROW + cg * 32 + 16);
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], osc, bs[j]);
        if (has_res) {
          float rv[8];
          unpack8<T>(rraw[it], rv);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaf(rv[j], osc, v[j]);
        }
        if (has_stats) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            ssum[j] += v[j];
            ssq[j] = fmaf(v[j], v[j], ssq[j]);
          }
        }
        pack8<T>(v, oraw[it]);
      }
    } else {
#pragma unroll
      for (int it = 0; it < NROW; ++it) {
        const int m = mrow[it];
        const int lp = tid / NCG + it * (NT / NCG);
        float v[8];
        const float4 a0 = *reinterpret_cast<const float4*>(smem + lp * OROW + cg * 32);
        const float4 a1 = *reinterpret_cast<const float4*>(smem + lp * OROW + cg * 32 + 16);
        v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
        float rv[8];
        unpack8<T>(rraw[it], rv);
        if (p.div_b) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] / dvs;
        }
        if (p.bias_mode == 1 && p.bias) {  // row bias (the V^T GEMM of the attention block)
          const float rowb = p.bias[m < 0 ? 0 : m];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (co + j < p.Cout) ? v[j] + rowb : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] + bv[j] + rv[j]) * p.out_scale;
        const float keep = m < 0 ? 0.f : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // the barriers keep these as scalar FMAs.  Without them the compiler pairs (sum, sum of squares) into
          // v_mul_f32 + v_pk_fma_f32 op_sel sequences, and that code was measured returning a wrong SUM lane (the
          // square lane and the stored output stayed right) in ~1 % of forwards once kernels of other hardware
          // queues shared the CUs — never with a single stream (profiles/experiments/README.md, "streams")
          ssum[j] = fmaf(keep, v[j], ssum[j]);
          asm volatile("" : "+v"(ssum[j]));
          ssq[j] = fmaf(keep * v[j], v[j], ssq[j]);
          asm volatile("" : "+v"(ssq[j]));
        }
        pack8<T>(v, oraw[it]);
      }
    }
    CT_MARK(8)
#pragma unroll
    for (int it = 0; it < NROW; ++it)
#pragma unroll
      for (int q = 0; q < RV; ++q) buf_store16(ry, voy[it], 16u * q, oraw[it][q]);
    CT_MARK(9)
  }
  if (p.stats) {  // block-reduce the per-thread partials: 256/NCG threads share a cout group
    __syncthreads();
    float* sr = reinterpret_cast<float*>(smem);  // [256/NCG][BN][2]
    const int r = tid / NCG;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sr[(r * BN + cg * 8 + j) * 2 + 0] = ssum[j];
      sr[(r * BN + cg * 8 + j) * 2 + 1] = ssq[j];
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      double a = 0.0, q = 0.0;
      for (int rr2 = 0; rr2 < NT / NCG; ++rr2) {
        a += (double)sr[(rr2 * BN + tid) * 2 + 0];
        q += (double)sr[(rr2 * BN + tid) * 2 + 1];
      }
      // integer (fixed-point) atomics: associative, so the image totals are bit-reproducible whatever the order
      long long* o = p.stats + ((long)b * p.Cout + n0 + tid) * 2;
      ds_stat_add(o, (long long)llrint(a * DS_STAT_SUM_SCALE));
      ds_stat_add(o + 1, (long long)llrint(q * DS_STAT_SQ_SCALE));
    }
  }
  CT_MARK(10)
  CT_FLUSH
}

template <typename T, int TAPS, int TH, int TW, int BN, int WM, int WN, int KC, int EP = 1, int OCC = 2, int SP = 0, int NT = 256>
static int launch_cfg(const ConvArgs& a, hipStream_t st) {
  using G = ConvGeom<T, TAPS, TH, TW, BN, KC, EP, NT>;
  constexpr int LDS = G::LDS + 4096;  // + the [2][Cin <= 512] GroupNorm table of the accumulator mode
  auto kern = conv_mfma_kernel<T, TAPS, TH, TW, BN, WM, WN, KC, EP, OCC, SP, NT>;
  DS_FUNC_LDS_ONCE(kern, LDS);
  ConvK k;
  k.x = a.x; k.x_bs = a.x_bs; k.ldx = a.ldx; k.C1 = a.x2 ? a.C1 : a.Cin;
  k.x2 = a.x2; k.x2_bs = a.x2_bs; k.ldx2 = a.ldx2;
  k.w = a.w; k.w_bs = a.w_bs;
  // chunk-major weights [Cin/kc][taps][Cout][kc]: a K stage of KC channels is KC / kc consecutive layout chunks
  DS_CHECK(a.w_chunked == 0 || (a.w_chunked >= 8 && KC % a.w_chunked == 0 && a.Cin % KC == 0 && (!a.x2 || a.C1 % KC == 0)),
           "conv: chunk-major weights need kc | the kernel's chunk width and whole chunks per source");
  k.w_chunked = a.w_chunked; k.w_shift = a.w_chunked ? __builtin_ctz(a.w_chunked) : 0;
  DS_CHECK((a.w_chunked & (a.w_chunked - 1)) == 0 && (a.sw_chunked & (a.sw_chunked - 1)) == 0, "conv: weight chunks are powers of two");
  DS_CHECK(!a.sx || (G::SKIP_OK && a.sw && a.sCin % 8 == 0 && a.ldsx % 8 == 0 &&
                     (a.sw_chunked == 0 || (a.sw_chunked >= 8 && KC % a.sw_chunked == 0 && a.sCin % KC == 0 &&
                                            (!a.sx2 || a.sC1 % KC == 0)))),
           "conv: bad fused skip convolution arguments");
  k.sx = a.sx; k.sx_bs = a.sx_bs; k.ldsx = a.ldsx; k.sx2 = a.sx2; k.sx2_bs = a.sx2_bs; k.ldsx2 = a.ldsx2;
  k.sC1 = a.sx2 ? a.sC1 : a.sCin; k.sCin = a.sCin; k.sw = a.sw; k.sw_chunked = a.sw_chunked; k.sw_shift = a.sw_chunked ? __builtin_ctz(a.sw_chunked) : 0;
  k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift; k.gn_act = a.gn_act;
  k.bias = a.bias; k.bias_b = a.bias_b; k.bias_b_ld = a.bias_b_ld; k.bias_mode = a.bias_mode; k.div_b = a.div_b;
  k.res = a.res; k.res_bs = a.res_bs; k.ldr = a.ldr; k.out_scale = a.out_scale;
  k.y = a.y; k.y_bs = a.y_bs; k.ldy = a.ldy; k.stats = a.stats_acc;
  k.gn_acc1 = a.gn_acc1; k.gn_acc2 = a.gn_acc2; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta;
  k.gn_groups = a.gn_groups; k.gn_inv_count = a.gn_inv_count; k.gn_eps = a.gn_eps;
  DS_CHECK(!a.gn_acc1 || (a.Cin <= 512 && a.gn_groups > 0 && a.Cin % a.gn_groups == 0 && (!a.x2 || a.gn_acc2)),
           "conv: bad GroupNorm accumulator arguments");
  k.H = a.H; k.W = a.W; k.Cin = a.Cin; k.Cout = a.Cout;
  dim3 grid;
  if (TAPS == 9) {
    k.tiles_x = cdiv(a.W, TW);
    grid.x = k.tiles_x * cdiv(a.H, TH);
  } else {
    k.tiles_x = 1;
    grid.x = cdiv((long)a.H * a.W, TH * TW);
  }
  grid.y = cdiv(a.Cout, BN);
  grid.z = a.B;
  hipLaunchKernelGGL(kern, grid, dim3(NT), LDS, st, k);
  DS_LAUNCH_CHECK();
  {
    static char name[128] = {0};
    if (!name[0])
      snprintf(name, sizeof(name), "conv_mfma_kernel<%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d>", sizeof(T) == 2 ? DS_HALF_NAME : "float", TAPS,
               TH, TW, BN, WM, WN, KC, EP, OCC, SP, NT);
    ds_set_last_conv_kernel(name);
  }
  return 0;
}

static thread_local const char* g_last_conv_kernel = "";
const char* ds_last_conv_kernel() { return g_last_conv_kernel; }
void ds_set_last_conv_kernel(const char* name) { g_last_conv_kernel = name; }

template <typename T, int SP = 0>
static int launch_typed(const ConvArgs& a, hipStream_t st) {
  constexpr int KC9 = (sizeof(T) == 4) ? 16 : 32;
  constexpr int KC1 = (sizeof(T) == 4) ? 32 : 64;
  switch (ds_conv_config_id(a)) {
    case 0: {
      // few tiles (the 32^2 level: 128 blocks of 8 x 32 pixels on 256 CUs): half-width tiles fill the chip
      // (one batch alone 325.6 -> 320.6 ms, four in flight unchanged)
      const long blocks = (long)cdiv(a.W, 32) * cdiv(a.H, 8) * cdiv(a.Cout, 64) * a.B;
      if constexpr (sizeof(T) == 2) {
      // (64-channel K stages at one block per CU for these launches — option kc64 of round 4 — shortened a block's life by
      // 13 % and moved neither the one-batch latency nor the in-flight throughput: profiles/experiments/README.md)
      if (blocks <= 128) return launch_cfg<T, 9, 8, 16, 64, 1, 2, KC9, 1, 2, SP>(a, st);
      // many tiles and >= 128 couts: one 8-wave block computes 128 couts of a tile, so the halo tile is loaded and
      // activated once per 128 couts instead of once per 64 (nf = 128: 20.3 -> 20.9 utt/s; nf = 64: unchanged)
      // weight-heavy 64-cout launches without a folded skip (Conv_0 of the 256^2 / 128^2 up path): 16 x 32 pixels on 8 waves,
      // one weight stage per 512 pixels (+1 % end to end)
      if (!a.sx && a.Cin >= 128 && a.Cout == 64 && a.H % 16 == 0 && blocks >= 1024)
        return launch_cfg<T, 9, 16, 32, 64, 2, 2, KC9, 2, 2, SP, 512>(a, st);
      if (a.Cout % 128 == 0 && blocks >= 1024)
        return launch_cfg<T, 9, 8, 32, 128, 2, 2, KC9, 2, 2, SP, 512>(a, st);
      }
      return launch_cfg<T, 9, 8, 32, 64, 2, 2, KC9, 1, 2, SP>(a, st);
    }
    case 1: return launch_cfg<T, 9, 8, 32, 32, 2, 1, KC9, 1, 2, SP>(a, st);
    case 2: {  // small images: a chain of dependent chunk round trips (1.7 us each) -> chunks twice as deep (+1 %)
      const bool deep = a.Cin % (2 * KC9) == 0 && (!a.x2 || a.C1 % (2 * KC9) == 0) &&
                        (!a.sx || (a.sCin % (2 * KC9) == 0 && (!a.sx2 || a.sC1 % (2 * KC9) == 0)));
      if (deep) return launch_cfg<T, 9, 8, 8, 64, 1, 1, KC9 * 2, 1, 2, SP>(a, st);
      return launch_cfg<T, 9, 8, 8, 64, 1, 1, KC9, 1, 2, SP>(a, st);
    }
    case 3: return launch_cfg<T, 1, 8, 32, 64, 2, 2, KC1, 1, 2, SP>(a, st);
    case 4: return launch_cfg<T, 1, 8, 32, 32, 2, 1, KC1, 1, 2, SP>(a, st);
    default: return launch_cfg<T, 1, 8, 8, 64, 1, 1, KC1, 1, 2, SP>(a, st);
  }
}

// whether the instantiation that would run a 3x3 launch of this shape can take the fused 1x1 skip convolution
bool ds_conv_skip_supported(int H, int W, int Cout, int dtype) {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.H = H; a.W = W; a.Cout = Cout; a.taps = 9; a.dtype = dtype; a.sx = &a;
  const int id = ds_conv_config_id(a);
  return id == 0 || id == 2;  // (bf16 small images go to conv3x3_small.hip, which takes the skip too: checked there)
}

// chunk width (channels per K stage) of the kernel that would run this problem: the kc of chunk-major weights
int ds_conv_chunk(int taps, int dtype) {
  if (dtype == DS_F32) return taps == 9 ? 16 : 32;
  return taps == 9 ? 32 : 64;
}

// Which instantiation ds_launch_conv picks (profiling label): 0/1/2 = 3x3 {8x32xBN64, 8x32xBN32, 8x8xBN64},
// 3/4/5 = the same tiles for 1x1 / GEMM, 6 = the weight-stationary 64 -> 64 kernel (conv3x3_ws.hip), 7 = the
// small-image kernel (conv3x3_small.hip), 8 = the register-weight kernel (conv3x3_rw.hip), 10 = the streamed-weight kernel
// (conv3x3_sw.hip), 11 = its split-mode sibling (conv3x3_sws.hip).  (9 = the fused attention block, launched by engine.hip.)
int ds_conv_config_id(const ConvArgs& a) {
  if (ds_conv_sws_eligible(a)) return 11;
  if (ds_conv_sw_eligible(a)) return 10;
  if (ds_conv_rw_eligible(a)) return 8;
  if (ds_conv_ws_eligible(a) || ds_conv_thin_eligible(a) || ds_conv_thin_out_eligible(a)) return 6;
  if (ds_conv_small_eligible(a)) return 7;
  if (a.taps == 9) {
    if (a.W >= 32 && a.H >= 8) return a.Cout <= 32 ? 1 : 0;
    return 2;
  }
  // small-M problems with a wide N (the STFT GEMM: M = 512 bins, N = frames) still want the 256-row tile
  if ((long)a.H * a.W >= 1024 || ((long)a.H * a.W >= 256 && a.Cout >= 1024)) return a.Cout <= 32 ? 4 : 3;
  return 5;
}

int ds_launch_conv(const ConvArgs& a, hipStream_t st) {
  DS_CHECK(a.taps == 1 || a.taps == 9, "conv: taps must be 1 or 9");
  DS_CHECK(a.Cin % 8 == 0 && a.ldx % 8 == 0, "conv: Cin and ldx must be multiples of 8");
  DS_CHECK(a.B > 0 && a.H > 0 && a.W > 0 && a.Cout > 0, "conv: empty problem");
  DS_CHECK(a.x && a.w && a.y, "conv: null pointer");
  DS_CHECK(a.ldy >= ((a.Cout + 7) & ~7), "conv: output pixel stride must cover Cout rounded up to 8");
  DS_CHECK(!a.x2 || (a.C1 % 8 == 0 && a.C1 > 0 && a.C1 < a.Cin && a.ldx2 % 8 == 0), "conv: bad concat split");
  {  // 32-bit buffer offsets with bit 31 reserved as the out-of-range marker: < 2 GiB per batch entry
    const long esz = a.dtype == DS_F32 ? 4 : 2, M = (long)a.H * a.W;
    long mld = a.ldx > a.ldy ? a.ldx : a.ldy;
    if (a.x2 && a.ldx2 > mld) mld = a.ldx2;
    if (a.res && a.ldr > mld) mld = a.ldr;
    DS_CHECK(M * mld * esz < 2147483647L, "conv: image too large for 32-bit buffer offsets");
    DS_CHECK((long)a.Cout * a.taps * a.Cin * esz < 2147483647L, "conv: weight tensor too large");
  }
  if (ds_conv_sws_eligible(a)) return ds_launch_conv_sws(a, st);
  if (ds_conv_sw_eligible(a)) return ds_launch_conv_sw(a, st);
  if (ds_conv_rw_eligible(a)) return ds_launch_conv_rw(a, st);
  if (ds_conv_ws_eligible(a)) return ds_launch_conv_ws(a, st);
  if (ds_conv_thin_eligible(a)) return ds_launch_conv_thin(a, st);
  if (ds_conv_thin_out_eligible(a)) return ds_launch_conv_thin_out(a, st);
  if (ds_conv_small_eligible(a)) return ds_launch_conv_small(a, st);
  if (a.dtype == DS_F32) return a.split ? launch_typed<float, 1>(a, st) : launch_typed<float, 0>(a, st);
  if (a.dtype == DS_BF16) return launch_typed<bf16_t>(a, st);
  DS_CHECK(false, "conv: unknown dtype");
}
