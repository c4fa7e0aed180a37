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
#include "common.h"

template <typename T> struct Mma;
template <> struct Mma<float> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};
template <> struct Mma<bf16_t> {
  __device__ static inline void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0,
                                                0);
  }
};

// GN-apply (+SiLU) on one 16-byte vector of KV channels
template <typename T> struct GnVec;
template <> struct GnVec<float> {
  __device__ static inline uint4 run(const uint4& u, const float* sc, const float* sh, int act) {
    float f[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = f[j] * sc[j] + sh[j];
      f[j] = act ? silu_t<float>(v) : v;
    }
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct GnVec<bf16_t> {
  __device__ static inline uint4 run(const uint4& u, const float* sc, const float* sh, int act) {
    float f[8];
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = f[j] * sc[j] + sh[j];
      f[j] = act ? silu_t<bf16_t>(v) : v;
    }
    uint4 o;
    o.x = pack_bf16x2(f[0], f[1]);
    o.y = pack_bf16x2(f[2], f[3]);
    o.z = pack_bf16x2(f[4], f[5]);
    o.w = pack_bf16x2(f[6], f[7]);
    return o;
  }
};

struct ConvK {  // kernel-side copy of ConvArgs (typed by the template)
  const void* x; long x_bs; int ldx; int C1;
  const void* x2; long x2_bs; int ldx2;
  const void* w; long w_bs;
  const float* gn_scale; const float* gn_shift; int gn_act;
  const float* bias; const float* bias_b; int bias_b_ld; int bias_mode;
  const float* div_b;
  const void* res; long res_bs; int ldr;
  float out_scale;
  void* y; long y_bs; int ldy;
  double* stats;  // optional [B][gridDim.x][Cout][2]: per-tile sum / sum of squares of the OUTPUT channels
  int H, W, Cin, Cout;
  int tiles_x;
};

template <typename T, int TAPS, int TH, int TW, int BN, int KC>
struct ConvGeom {
  static constexpr int KV = 16 / (int)sizeof(T);
  static constexpr int R = (TAPS == 9) ? 1 : 0;
  static constexpr int HW_ = TW + 2 * R, HH_ = TH + 2 * R, HP = HW_ * HH_;
  static constexpr int BM = TH * TW;
  static constexpr int ROWB = KC * (int)sizeof(T) + 16;
  static constexpr int NVEC = KC / KV;
  static constexpr int NKB = KC / (2 * KV);
  static constexpr int NA = (HP * NVEC + 255) / 256;
  static constexpr int NB = (TAPS * BN * NVEC + 255) / 256;
  static constexpr int OROW = BN * 4 + 16;  // fp32 output staging row pitch
  static constexpr int LDS_STAGE = HP * ROWB + TAPS * BN * ROWB;
  static constexpr int LDS_OUT = BM * OROW;
  static constexpr int LDS = LDS_STAGE > LDS_OUT ? LDS_STAGE : LDS_OUT;
};

template <typename T, int TAPS, int TH, int TW, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvK p) {
  using G = ConvGeom<T, TAPS, TH, TW, BN, KC>;
  constexpr int KV = G::KV, R = G::R, HW_ = G::HW_, HP = G::HP, BM = G::BM, ROWB = G::ROWB, NVEC = G::NVEC,
                NKB = G::NKB, NA = G::NA, NB = G::NB, OROW = G::OROW;
  constexpr int WAVES_N = BN / (32 * WN);
  constexpr int WAVES_M = BM / (32 * WM);
  static_assert(WAVES_M * WAVES_N == 4, "block is 4 waves");
  static_assert(KC % (2 * KV) == 0, "KC must hold whole k-blocks");
  static_assert(256 % NVEC == 0, "a thread keeps one channel offset across its vectors");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  char* sB = smem + HP * ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  const int b = blockIdx.z;
  const int n0 = blockIdx.y * BN;
  int y0 = 0, x0 = 0;
  long m0 = 0;
  const long M = (long)p.H * p.W;
  if (TAPS == 9) {
    y0 = (blockIdx.x / p.tiles_x) * TH;
    x0 = (blockIdx.x % p.tiles_x) * TW;
  } else {
    m0 = (long)blockIdx.x * BM;
  }

  const T* xb1 = reinterpret_cast<const T*>(p.x) + (long)b * p.x_bs;
  const T* xb2 = p.x2 ? reinterpret_cast<const T*>(p.x2) + (long)b * p.x2_bs : nullptr;
  const T* wb = reinterpret_cast<const T*>(p.w) + (long)b * p.w_bs;
  const bool has_gn = p.gn_scale != nullptr;

  // ---- per-thread staging descriptors (chunk independent)
  const int vch = (tid % NVEC) * KV;  // channel offset of this thread's vectors inside a chunk
  int apix[NA];                       // pixel index inside the image (or -1: outside / unused)
  int alds[NA];                       // LDS byte offset of the vector
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const int i = tid + k * 256;
    const int pix = i / NVEC;
    apix[k] = -1;
    alds[k] = (i < HP * NVEC) ? pix * ROWB + (i - pix * NVEC) * 16 : -1;
    if (i < HP * NVEC) {
      if (TAPS == 9) {
        const int hy = pix / HW_, hx = pix - hy * HW_;
        const int gy = y0 + hy - R, gx = x0 + hx - R;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) apix[k] = gy * p.W + gx;
      } else {
        const long m = m0 + pix;
        if (m < M) apix[k] = (int)m;
      }
    }
  }
  int bsrc[NB];  // element offset of (co, tap, 0) in the weight tensor (or -1)
  int blds[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int i = tid + k * 256;
    const int row = i / NVEC;
    const int tap = row / BN, col = row - tap * BN;
    const int co = n0 + col;
    const bool in = i < TAPS * BN * NVEC;
    blds[k] = in ? row * ROWB + (i - row * NVEC) * 16 : -1;
    bsrc[k] = (in && co < p.Cout) ? (co * TAPS + tap) * p.Cin : -1;
  }

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
    const int row = (TAPS == 9) ? ((pp / TW) * HW_ + (pp % TW)) : pp;
    aoff[i] = row * ROWB + h * 16;
  }
#pragma unroll
  for (int j = 0; j < WN; ++j) boff[j] = ((wn * WN + j) * 32 + l32) * ROWB + h * 16;

  uint4 pa[NA], pb[NB];
  float gsc[KV], gsh[KV];
  bool ch_ok = false;

  auto load_chunk = [&](int ci0) {
    const int ci = ci0 + vch;
    ch_ok = ci < p.Cin;
    const bool second = xb2 != nullptr && ci >= p.C1;
    const T* src = second ? xb2 + (ci - p.C1) : xb1 + ci;
    const int ld = second ? p.ldx2 : p.ldx;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      pa[k] = make_uint4(0u, 0u, 0u, 0u);
      if (apix[k] >= 0 && ch_ok) pa[k] = *reinterpret_cast<const uint4*>(src + (long)apix[k] * ld);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      pb[k] = make_uint4(0u, 0u, 0u, 0u);
      if (bsrc[k] >= 0 && ch_ok) pb[k] = *reinterpret_cast<const uint4*>(wb + bsrc[k] + ci);
    }
    if (has_gn && ch_ok) {
#pragma unroll
      for (int j = 0; j < KV; ++j) {
        gsc[j] = p.gn_scale[(long)b * p.Cin + ci + j];
        gsh[j] = p.gn_shift[(long)b * p.Cin + ci + j];
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      if (alds[k] >= 0) {
        uint4 v = pa[k];
        if (has_gn && apix[k] >= 0 && ch_ok) v = GnVec<T>::run(v, gsc, gsh, p.gn_act);
        *reinterpret_cast<uint4*>(sA + alds[k]) = v;
      }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (blds[k] >= 0) *reinterpret_cast<uint4*>(sB + blds[k]) = pb[k];
  };

  load_chunk(0);
  for (int ci0 = 0; ci0 < p.Cin; ci0 += KC) {
    __syncthreads();  // previous chunk's fragment reads are done
    store_chunk();
    __syncthreads();
    if (ci0 + KC < p.Cin) load_chunk(ci0 + KC);  // in flight during the MFMA loop below
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int toff = (TAPS == 9) ? ((tap / 3) * HW_ + (tap % 3)) * ROWB : 0;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        uint4 af[WM], bfr[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) af[i] = *reinterpret_cast<const uint4*>(sA + aoff[i] + toff + kb * 32);
#pragma unroll
        for (int j = 0; j < WN; ++j)
          bfr[j] = *reinterpret_cast<const uint4*>(sB + boff[j] + tap * BN * ROWB + kb * 32);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
      }
    }
  }

  // ---- epilogue, part 1: accumulators -> LDS (fp32, [pixel][cout]).
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h.
  __syncthreads();
  float* so = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pp = (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int cc = (wn * WN + j) * 32 + l32;
        *reinterpret_cast<float*>(smem + pp * OROW + cc * 4) = acc[i][j][r];
      }
  __syncthreads();
  // ---- part 2: one thread = one pixel x 8 couts: bias / temb / residual / scale, 16-byte stores
  T* yb = reinterpret_cast<T*>(p.y) + (long)b * p.y_bs;
  const T* rb = p.res ? reinterpret_cast<const T*>(p.res) + (long)b * p.res_bs : nullptr;
  const int cout8 = (p.Cout + 7) & ~7;
  const float dvs = p.div_b ? p.div_b[b] : 1.0f;
  constexpr int NCG = BN / 8;
  static_assert(256 % NCG == 0, "a thread keeps one cout group across the epilogue loop");
  const int cg = tid % NCG;
  const int co = n0 + cg * 8;
  float bv[8];  // per-thread column bias (conv bias + per-batch temb bias), loaded once
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float t = 0.f;
    const int c = co + j;
    if (p.bias_mode == 0 && c < p.Cout) {
      if (p.bias) t += p.bias[c];
      if (p.bias_b) t += p.bias_b[(long)b * p.bias_b_ld + c];
    }
    bv[j] = t;
  }
  float ssum[8], ssq[8];  // GroupNorm statistics of what this thread writes (consumed by the next GN)
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  if (co < cout8) {
    for (int pp = tid / NCG; pp < BM; pp += 256 / NCG) {
      long m;
      if (TAPS == 9) {
        const int gy = y0 + pp / TW, gx = x0 + pp % TW;
        if (gy >= p.H || gx >= p.W) continue;
        m = (long)gy * p.W + gx;
      } else {
        m = m0 + pp;
        if (m >= M) continue;
      }
      float v[8];
      const float4 a0 = *reinterpret_cast<const float4*>(smem + pp * OROW + cg * 32);
      const float4 a1 = *reinterpret_cast<const float4*>(smem + pp * OROW + cg * 32 + 16);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      float rv[8];
      if (rb) load8<T>(rb + m * p.ldr + co, rv);
      const float rowb = (p.bias_mode == 1 && p.bias) ? p.bias[m] : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = v[j];
        if (p.div_b) t = t / dvs;
        t = (co + j < p.Cout) ? t + bv[j] + rowb : 0.f;
        if (rb) t += rv[j];
        v[j] = t * p.out_scale;
        ssum[j] += v[j];
        ssq[j] = fmaf(v[j], v[j], ssq[j]);
      }
      store8<T>(yb + m * p.ldy + co, v);
    }
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
      for (int rr = 0; rr < 256 / NCG; ++rr) {
        a += (double)sr[(rr * BN + tid) * 2 + 0];
        q += (double)sr[(rr * BN + tid) * 2 + 1];
      }
      double* o = p.stats + (((long)b * gridDim.x + blockIdx.x) * p.Cout + n0 + tid) * 2;
      o[0] = a;
      o[1] = q;
    }
  }
  (void)so;
}

template <typename T, int TAPS, int TH, int TW, int BN, int WM, int WN, int KC>
static int launch_cfg(const ConvArgs& a, hipStream_t st) {
  using G = ConvGeom<T, TAPS, TH, TW, BN, KC>;
  constexpr int LDS = G::LDS;
  auto kern = conv_mfma_kernel<T, TAPS, TH, TW, BN, WM, WN, KC>;
  static bool attr_done = false;
  if (!attr_done) {
    DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done = true;
  }
  ConvK k;
  k.x = a.x; k.x_bs = a.x_bs; k.ldx = a.ldx; k.C1 = a.x2 ? a.C1 : a.Cin;
  k.x2 = a.x2; k.x2_bs = a.x2_bs; k.ldx2 = a.ldx2;
  k.w = a.w; k.w_bs = a.w_bs;
  k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift; k.gn_act = a.gn_act;
  k.bias = a.bias; k.bias_b = a.bias_b; k.bias_b_ld = a.bias_b_ld; k.bias_mode = a.bias_mode; k.div_b = a.div_b;
  k.res = a.res; k.res_bs = a.res_bs; k.ldr = a.ldr; k.out_scale = a.out_scale;
  k.y = a.y; k.y_bs = a.y_bs; k.ldy = a.ldy; k.stats = a.stats_out;
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
  hipLaunchKernelGGL(kern, grid, dim3(256), LDS, st, k);
  DS_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int launch_typed(const ConvArgs& a, hipStream_t st) {
  constexpr int KC9 = (sizeof(T) == 4) ? 16 : 32;
  constexpr int KC1 = (sizeof(T) == 4) ? 32 : 64;
  switch (ds_conv_config_id(a)) {
    case 0: return launch_cfg<T, 9, 8, 32, 64, 2, 2, KC9>(a, st);
    case 1: return launch_cfg<T, 9, 8, 32, 32, 2, 1, KC9>(a, st);
    case 2: return launch_cfg<T, 9, 8, 8, 64, 1, 1, KC9>(a, st);
    case 3: return launch_cfg<T, 1, 8, 32, 64, 2, 2, KC1>(a, st);
    case 4: return launch_cfg<T, 1, 8, 32, 32, 2, 1, KC1>(a, st);
    default: return launch_cfg<T, 1, 8, 8, 64, 1, 1, KC1>(a, st);
  }
}

// grid.x of the launch = number of output tiles per image (the stride of the statistics partials)
int ds_conv_tiles(const ConvArgs& a) {
  const int id = ds_conv_config_id(a);
  if (id <= 1) return cdiv(a.W, 32) * cdiv(a.H, 8);
  if (id == 2) return cdiv(a.W, 8) * cdiv(a.H, 8);
  const long M = (long)a.H * a.W;
  return id == 5 ? cdiv(M, 64) : cdiv(M, 256);
}

// Which instantiation ds_launch_conv picks (profiling label): 0/1/2 = 3x3 {8x32xBN64, 8x32xBN32, 8x8xBN64},
// 3/4/5 = the same tiles for 1x1 / GEMM.
int ds_conv_config_id(const ConvArgs& a) {
  if (a.taps == 9) {
    if (a.W >= 32 && a.H >= 8) return a.Cout <= 32 ? 1 : 0;
    return 2;
  }
  if ((long)a.H * a.W >= 1024) return a.Cout <= 32 ? 4 : 3;
  return 5;
}

int ds_launch_conv(const ConvArgs& a, hipStream_t st) {
  DS_CHECK(a.taps == 1 || a.taps == 9, "conv: taps must be 1 or 9");
  DS_CHECK(a.Cin % 8 == 0 && a.ldx % 8 == 0, "conv: Cin and ldx must be multiples of 8");
  DS_CHECK(a.B > 0 && a.H > 0 && a.W > 0 && a.Cout > 0, "conv: empty problem");
  DS_CHECK(a.x && a.w && a.y, "conv: null pointer");
  DS_CHECK(a.ldy >= ((a.Cout + 7) & ~7), "conv: output pixel stride must cover Cout rounded up to 8");
  DS_CHECK(!a.x2 || (a.C1 % 8 == 0 && a.C1 > 0 && a.C1 < a.Cin && a.ldx2 % 8 == 0), "conv: bad concat split");
  DS_CHECK((long)a.H * a.W * (a.ldx > a.ldy ? a.ldx : a.ldy) < 2147483647L, "conv: image too large for 32-bit offsets");
  DS_CHECK((long)a.Cout * a.taps * a.Cin < 2147483647L, "conv: weight tensor too large");
  if (a.dtype == DS_F32) return launch_typed<float>(a, st);
  if (a.dtype == DS_BF16) return launch_typed<bf16_t>(a, st);
  DS_CHECK(false, "conv: unknown dtype");
}
