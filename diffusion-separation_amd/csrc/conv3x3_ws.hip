// conv3x3_ws.hip — weight-stationary 3x3 convolution for the 64 -> 64 channel layers of NCSN++ (bf16, gfx950).
//
// These layers (ddpm_conv3x3 in ResnetBlockBigGANpp at the 256^2 and 128^2 levels, reference layers.py:141-156,
// layerspp.py:291-323) are HBM-bound: 77 GFLOP against 400 MB of activations per launch at B = 16.  The generic
// implicit-GEMM kernel (conv_mfma.hip) re-stages the whole 73 KB weight tensor for every 256-pixel tile, which is
// 63 % of its L2 -> CU traffic and of its LDS writes.  Here a block is persistent:
//
//   * one block of 8 waves per CU; the [9][64][64] weights are copied into LDS ONCE per block (83 KB, rows padded
//     to 144 B) and stay there for all of the block's tiles (a contiguous raster range of one image);
//   * all 8 waves run the same phase: a tile is two chunks of 32 input channels that go through a 2-slot LDS ring with
//     ONE LDS-only barrier per chunk; two register sets keep every global load in flight for a whole chunk phase (the
//     loads are issued one per k-step inside the MFMA loop), GroupNorm affine + SiLU is applied in registers between
//     the MFMAs of the same wave, and the next tile's first chunk is loading while this tile's epilogue runs;
//   * a wave owns 1 pixel row (32 pixels) x 64 couts of the 8 x 32 tile (2 MFMA tiles: 36 MFMAs per chunk).  MFMA
//     operands are swapped (A = weights, B = pixels), so a lane ends up holding 4 consecutive output channels of ONE
//     pixel per accumulator quad = one 16-byte LDS write.  The epilogue turns the wave's result into pixel rows
//     through a private 2.3 KB LDS scratch, 8 pixels at a time (no block barrier: LDS operations of one wave execute
//     in order); a lane then owns 8 consecutive couts of a pixel: bias / residual / scale / statistics / bf16 packing
//     and full 128-byte-line stores;
//   * GroupNorm statistics of the output are accumulated in 16 registers per lane over all tiles of the block
//     and added once at the end to the [B][64][2] fixed-point accumulators of the tensor.
//
// (Measured alternatives — ping-pong wave groups, 16 x 32 tiles, role-specialised waves, register-resident weights with
// LDS-DMA — are kept as text under profiles/experiments/ with their numbers.)
//
// K order (chunk, tap, 16-channel block) is the same as in conv_mfma.hip.
//
// The file also holds the two other persistent kernels on the same 8 x 32 tiles: conv3x3_thin_in_kernel (the network's
// first layer, 8 -> 64) and conv3x3_thin_out_kernel (the output-pyramid heads, 64 k -> <= 8 couts); each is described at
// its definition.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef WS_TIMING  // profiling build only: per-phase cycle totals of wave 0 (group 0) and wave 4 (group 1)
__device__ unsigned long long g_ws_dbg[32];
#define WT_DECL unsigned wt_prev = (unsigned)__builtin_readcyclecounter(), wt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define WT_MARK(i) { unsigned wt_now = (unsigned)__builtin_readcyclecounter(); wt_acc[i] += wt_now - wt_prev; wt_prev = wt_now; }
#define WT_FLUSH if ((threadIdx.x & 255) == 0) { for (int q = 0; q < 12; ++q) atomicAdd(&g_ws_dbg[(threadIdx.x >> 8) * 16 + q], (unsigned long long)wt_acc[q]); atomicAdd(&g_ws_dbg[(threadIdx.x >> 8) * 16 + 15], 1ull); }
extern "C" int diffsep_ws_debug_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_dbg), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ws_dbg), z, sizeof(z)); }
  return 0;
}
#else
#define WT_DECL
#define WT_MARK(i)
#define WT_FLUSH
#endif

#ifdef ABL_NOSTORE  // profiling: keep the epilogue arithmetic, drop (almost) all output traffic
#define WS_STORE(v, r, off, so, aux) if ((v).x == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, so, aux)
#else
#define WS_STORE(v, r, off, so, aux) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, so, aux)
#endif

namespace {

// (fragments are declared as bf16x8 = 8 raw 16-bit values; the MFMA is the storage format's)
__device__ inline f32x16 mfma_h32x(bf16x8 a, bf16x8 b, f32x16 c, int, int, int) {
  return mfma_h32(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, b), c);
}
__device__ inline f32x4_acc mfma_h16x(bf16x8 a, bf16x8 b, f32x4_acc c, int, int, int) {
  return mfma_h16(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, b), c);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr unsigned OOB = 0x80000000u;

__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ inline uint4 ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ inline uint2 ld8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
  return make_uint2(v.x, v.y);
}
__device__ inline void st8(__amdgpu_buffer_rsrc_t r, unsigned voff, uint2 d) {
  const u32x2_t v = {d.x, d.y};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 0);
}

// Block barrier that only orders LDS traffic.  __syncthreads() is a workgroup-scope fence: it drains vmcnt, i.e. it
// would wait for the global prefetch loads issued just before it and serialize them with the barrier.
__device__ inline void sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int C = 64;                 // input and output channels
constexpr int TH = 8, TW = 32;        // output tile
constexpr int HW_ = TW + 2, HH_ = TH + 2, HP = HW_ * HH_;  // halo tile: 340 pixels
constexpr int KC = 32;                // channels per chunk
constexpr int AROW = KC * 2 + 16;     // 80 B: halo pixel row in LDS (bank-conflict padding)
constexpr int WROW = C * 2 + 16;      // 144 B: weight row (tap, cout) in LDS
constexpr int LDS_W = 9 * C * WROW;   // 82,944
constexpr int LDS_A = HP * AROW;      // 27,200 per ring slot
constexpr int LDS_TAB = 3 * C * 4;    // GN scale, GN shift, (bias + temb bias) * out_scale
constexpr int NA = (HP * (KC / 8) + 255) / 256;  // 16-byte vectors per thread (of a 4-wave group) per chunk: 6
constexpr int EROW = 288;             // epilogue scratch: 8 pixel rows of 64 fp32 (+32 B: conflict-free quad writes)
constexpr int LDS_E = 8 * EROW;       // per wave
constexpr int RED_ROW = 20;           // final statistics reduce: 16 floats per thread (+4 pad)
constexpr int LDS_DESC = NA * 256 * 4;  // (variant B: NA1 * 512 * 4, the same) // staging descriptors (global byte offset | border flags) per group thread
constexpr int LDS_MAIN = LDS_W + 2 * LDS_A + LDS_TAB;
constexpr int LDS_TOTAL = LDS_MAIN + 8 * LDS_E + LDS_DESC;
static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget of one CU");
static_assert(512 * RED_ROW * 4 <= LDS_TOTAL, "the statistics reduce reuses the block's LDS");

struct WsK {
  const bf16_t* x; long x_bs; int ldx;
  const bf16_t* w; int w_chunked;
  const float* gn_scale; const float* gn_shift;
  const long long* gn_acc; const float* gn_gamma; const float* gn_beta; int gn_groups; float gn_inv_count; float gn_eps;
  const float* bias; const float* bias_b; int bias_b_ld;
  const bf16_t* res; long res_bs; int ldr;
  float out_scale;
  bf16_t* y; long y_bs; int ldy;
  long long* stats;
  // folded 1x1 skip convolution on the raw block input (SKB > 0): y += sw * cat([sx, sx2])
  const bf16_t* sx; long sx_bs; int ldsx;
  const bf16_t* sx2; long sx2_bs; int ldsx2;
  int sC1, sCin;
  const bf16_t* sw; int sw_chunked, sw_shift;
  int H, W, G, tiles_x, tiles_per_img;
};

// GN affine (+ SiLU) on 8 bf16 channels
template <bool ACT>
__device__ inline uint4 gn8(const uint4& u, const float* sc, const float* sh) {
#if defined(DS_HALF_F16) && !defined(DS_GN8_F32)
  // half-precision build, round 5: affine + SiLU in packed half precision, 8 instructions per dword (as the register-weight
  // convolution: DESIGN.md section 2 for what it costs in agreement — in front of a convolution, nothing measurable)
  if constexpr (ACT) {
    auto one = [&](unsigned w, int d) __attribute__((always_inline)) {
      const unsigned ps = pack_h2(sc[2 * d], sc[2 * d + 1]), pb = pack_h2(sh[2 * d], sh[2 * d + 1]);
      unsigned z, xx, e, dd, r, o;
      asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(z) : "v"(w), "v"(ps), "v"(pb));
      asm("v_pk_mul_f16 %0, %1, %2" : "=v"(xx) : "v"(z), "s"(0xbdc5bdc5u));  // x -log2(e)
      asm("v_exp_f16 %0, %1" : "=v"(e) : "v"(xx));
      asm("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(e) : "v"(xx));
      asm("v_pk_add_f16 %0, %1, %2" : "=v"(dd) : "v"(e), "s"(0x3c003c00u));
      asm("v_rcp_f16 %0, %1" : "=v"(r) : "v"(dd));
      asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r) : "v"(dd));
      asm("v_pk_mul_f16 %0, %1, %2" : "=v"(o) : "v"(z), "v"(r));
      return o;
    };
    uint4 o;
    o.x = one(u.x, 0); o.y = one(u.y, 1); o.z = one(u.z, 2); o.w = one(u.w, 3);
    return o;
  }
#endif
  float f[8];
  f[0] = h_lo(u.x); f[1] = h_hi(u.x);
  f[2] = h_lo(u.y); f[3] = h_hi(u.y);
  f[4] = h_lo(u.z); f[5] = h_hi(u.z);
  f[6] = h_lo(u.w); f[7] = h_hi(u.w);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = f[j] * sc[j] + sh[j];
    f[j] = ACT ? silu_t<bf16_t>(v) : v;
  }
  uint4 o;
  o.x = pack_h2(f[0], f[1]);
  o.y = pack_h2(f[2], f[3]);
  o.z = pack_h2(f[4], f[5]);
  o.w = pack_h2(f[6], f[7]);
  return o;
}

// ---- the kernel: all 8 waves in the same phase.  A wave owns ONE pixel row x 64 couts
// (36 MFMAs per chunk); the halo chunks go through a 2-slot ring with ONE barrier per chunk; the chunk in flight
// is activated in registers between the MFMAs of the same wave; the next tile's first chunk is loading while
// this tile's epilogue runs.
constexpr int NA1 = (HP * (KC / 8) + 511) / 512;  // 16-byte vectors per thread per chunk: 3
// SKB: 16-channel k-blocks of the folded 1x1 skip convolution (0 = none, 4 = 64 raw channels, 8 = 128 = two sources of
// 64).  Conv_2 of a residual block is linear in the RAW block input, so its product can be added to the conv
// accumulators at any time.  A 1x1 convolution has no halo and no reuse between waves, so it stays private to the wave:
// the wave keeps its 64 couts x 16 SKB channels of Conv_2.weight in registers for the whole launch (8 SKB VGPRs), reads
// the raw input of ITS 32 pixels with full-line loads (8 pixels x 128 bytes per instruction) behind chunk c's MFMA loop
// (source c), turns them into MFMA fragments through its private epilogue scratch (16 pixels at a time, wave-local LDS
// order, no barrier) and issues the 8 extra MFMAs at the end of the chunk.  These instantiations take no residual.
template <int MODE, int SKB = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_ws1_kernel(WsK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;
  char* sA = smem + LDS_W;
  float* sTab = reinterpret_cast<float*>(smem + LDS_W + 2 * LDS_A);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / p.G, part = blockIdx.x % p.G;
  const int t0 = (int)((long)part * p.tiles_per_img / p.G);
  const int nt = (int)((long)(part + 1) * p.tiles_per_img / p.G) - t0;
  const int Q = 2 * nt;
  {
    const __amdgpu_buffer_rsrc_t rw = rsrc(p.w, 9u * C * C * 2u);
    uint4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld16(rw, (unsigned)(tid + 512 * k) * 16u, 0);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int v = tid + 512 * k;
      int dst;
      if (p.w_chunked) {  // [chunk][tap][cout][32 ch]
        const int row = v >> 2, piece = v & 3;
        const int co = row & (C - 1), tap = (row >> 6) % 9, chunk = row / (9 * C);
        dst = (tap * C + co) * WROW + chunk * (KC * 2) + piece * 16;
      } else {            // [cout][tap][64 ch]
        const int row = v >> 3, piece = v & 7;
        const int co = row / 9, tap = row - co * 9;
        dst = (tap * C + co) * WROW + piece * 16;
      }
      *reinterpret_cast<uint4*>(sW + dst) = wv[k];
    }
    if (tid < C) {
      float sc = 1.f, sh = 0.f;
      if (p.gn_acc) {  // GroupNorm statistics straight from the producer's channel-sum accumulators
        const int cpg = C / p.gn_groups, g0 = (tid / cpg) * cpg;
        long long ssum = 0, ssq = 0;
        for (int j = 0; j < cpg; ++j) {
          ssum += p.gn_acc[((long)b * C + g0 + j) * 2];
          ssq += p.gn_acc[((long)b * C + g0 + j) * 2 + 1];
        }
        const double mean = (double)ssum * (1.0 / DS_STAT_SUM_SCALE) * (double)p.gn_inv_count;
        double var = (double)ssq * (1.0 / DS_STAT_SQ_SCALE) * (double)p.gn_inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        sc = (float)(1.0 / sqrt(var + (double)p.gn_eps)) * (p.gn_gamma ? p.gn_gamma[tid] : 1.f);
        sh = (p.gn_beta ? p.gn_beta[tid] : 0.f) - (float)mean * sc;
      } else if (p.gn_scale) {
        sc = p.gn_scale[(long)b * C + tid];
        sh = p.gn_shift[(long)b * C + tid];
      }
      sTab[tid] = sc;
      sTab[C + tid] = sh;
      sTab[2 * C + tid] =
          ((p.bias ? p.bias[tid] : 0.f) + (p.bias_b ? p.bias_b[(long)b * p.bias_b_ld + tid] : 0.f)) * p.out_scale;
    }
  }
  const int slot = tid & 3;
  int* sDesc = reinterpret_cast<int*>(smem + LDS_MAIN + 8 * LDS_E);
#pragma unroll
  for (int k = 0; k < NA1; ++k) {
    const int v = tid + 512 * k;
    const int pix = v >> 2, hy = pix / HW_, hx = pix - hy * HW_;
    const int rel = (((hy - 1) * p.W + (hx - 1)) * p.ldx + slot * 8) * 2;
    const int flg = (hy == 0 ? 1 : 0) | (hy == HH_ - 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == HW_ - 1 ? 8 : 0);
    sDesc[k * 512 + tid] = rel | flg;
  }
  const int ldo0 = (tid >> 2) * AROW + slot * 16;  // vector k is 128 pixels further
  const bool in_last = tid + 512 * (NA1 - 1) < HP * 4;
  const __amdgpu_buffer_rsrc_t rx = rsrc(p.x + (long)b * p.x_bs, (unsigned)(p.H * p.W) * p.ldx * 2u);
  const __amdgpu_buffer_rsrc_t ry = rsrc(p.y + (long)b * p.y_bs, (unsigned)(p.H * p.W) * p.ldy * 2u);
  const __amdgpu_buffer_rsrc_t rr =
      rsrc(p.res ? p.res + (long)b * p.res_bs : p.y, p.res ? (unsigned)(p.H * p.W) * p.ldr * 2u : 0u);

  // two register sets: while set (q+1)&1 (the chunk after the one in LDS) is activated and written, set q&1
  // already receives chunk q+2: every global load has a full chunk phase to land before its first use
  uint4 pa[2][NA1];
  bool pval[2][NA1];
  // global loads are issued ONE per k-step inside the MFMA loop: the texture addresser takes ~25 cycles per
  // 16-byte wave load, and a burst of them right after the barrier would hold back every wave's first MFMA
  unsigned ld_edge = 0, ld_soff = 0;
  int ld_tbase = 0;
  auto prep = [&](int q) {  // geometry of chunk q (tile q / 2, channels 32 (q & 1) ...)
    const int t = t0 + (q >> 1);
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;
    ld_edge = (y0 == 0 ? 1u : 0u) | (y0 + TH == p.H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == p.W ? 8u : 0u);
    ld_tbase = (y0 * p.W + x0) * p.ldx * 2;
    ld_soff = (unsigned)(q & 1) * (KC * 2);
  };
  auto issue_one = [&](int set, int k) {
    const int d = sDesc[k * 512 + tid];
    pval[set][k] = (k < NA1 - 1 || in_last) && !((unsigned)d & ld_edge);
#ifdef ABL2_NOLOAD
    pa[set][k] = make_uint4(d, d + 1, d + 2, ld_tbase);
#else
    pa[set][k] = ld16(rx, pval[set][k] ? (unsigned)((d & ~15) + ld_tbase) : OOB, ld_soff);
#endif
  };
  auto issue = [&](int q, int set) {
    prep(q);
#pragma unroll
    for (int k = 0; k < NA1; ++k) issue_one(set, k);
  };
  float gsc[8], gsh[8];
  auto act_tab = [&](int c) {
    const float4* ts = reinterpret_cast<const float4*>(sTab + c * KC + slot * 8);
    const float4* th = reinterpret_cast<const float4*>(sTab + C + c * KC + slot * 8);
    const float4 s0 = ts[0], s1 = ts[1], h0 = th[0], h1 = th[1];
    gsc[0] = s0.x; gsc[1] = s0.y; gsc[2] = s0.z; gsc[3] = s0.w; gsc[4] = s1.x; gsc[5] = s1.y; gsc[6] = s1.z; gsc[7] = s1.w;
    gsh[0] = h0.x; gsh[1] = h0.y; gsh[2] = h0.z; gsh[3] = h0.w; gsh[4] = h1.x; gsh[5] = h1.y; gsh[6] = h1.z; gsh[7] = h1.w;
  };
  auto act_one = [&](int set, int k) {
#ifdef ABL2_NOACT
    return;
#endif
    uint4 r = gn8<MODE == 2>(pa[set][k], gsc, gsh);
    asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w));
    pa[set][k].x = pval[set][k] ? r.x : pa[set][k].x;
    pa[set][k].y = pval[set][k] ? r.y : pa[set][k].y;
    pa[set][k].z = pval[set][k] ? r.z : pa[set][k].z;
    pa[set][k].w = pval[set][k] ? r.w : pa[set][k].w;
  };
  auto write = [&](int set) {  // chunk parity = register set = ring slot
    char* dst = sA + set * LDS_A;
#pragma unroll
    for (int k = 0; k < NA1; ++k)
#ifdef ABL2_NOLDSW
      if (pa[set][k].x == 0x12345678u)
#else
      if (k < NA1 - 1 || in_last)
#endif
        *reinterpret_cast<uint4*>(dst + ldo0 + k * 128 * AROW) = pa[set][k];
  };
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  const int aoff = (wave * HW_ + l32) * AROW + h * 16;
  const int woff = l32 * WROW + h * 16;
  char* sE = smem + LDS_MAIN + wave * LDS_E;
  const int epx = lane >> 3, ecg = lane & 7;
  const bool has_res = SKB == 0 && p.res != nullptr, has_stats = p.stats != nullptr;
  const float osc = p.out_scale;
  unsigned res_o = 0;
  uint4 rres[4];
  // folded skip: this lane's weight fragments (cout 32 j + l32, channels 16 kb + 8 h ..) and pixel fragments
  constexpr int NSRC = SKB / 4;  // 64-channel sources of the skip input
  uint4 wsk[2][SKB > 0 ? SKB : 1], xsk[4], xg[2];
  __amdgpu_buffer_rsrc_t rs1 = rx, rs2 = rx;
  unsigned skip_o1 = 0, skip_o2 = 0;
  if constexpr (SKB > 0) {
    const __amdgpu_buffer_rsrc_t rsw = rsrc(p.sw, (unsigned)C * p.sCin * 2u);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kb = 0; kb < SKB; ++kb) {
        const int co = 32 * j + l32, c0 = 16 * kb + 8 * h;
        const unsigned vo = p.sw_chunked ? (unsigned)((((c0 >> p.sw_shift) * C + co) * p.sw_chunked + (c0 & (p.sw_chunked - 1))) * 2)
                                         : (unsigned)((co * p.sCin + c0) * 2);
        wsk[j][kb] = ld16(rsw, vo, 0);
      }
    rs1 = rsrc(p.sx + (long)b * p.sx_bs, (unsigned)(p.H * p.W) * p.ldsx * 2u);
    rs2 = p.sx2 ? rsrc(p.sx2 + (long)b * p.sx2_bs, (unsigned)(p.H * p.W) * p.ldsx2 * 2u) : rs1;
  }
  // lane i of a staging load: pixel 8 m + (i >> 3) of the wave's row, 16-byte piece i & 7 of its 64 channels
  auto skip_load = [&](int c, int half) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const unsigned pixo = (unsigned)(16 * half + 8 * m + (lane >> 3));
      xg[m] = c == 0 ? ld16(rs1, skip_o1 + pixo * (unsigned)(p.ldsx * 2) + (unsigned)((lane & 7) * 16), 0)
                     : ld16(rs2, skip_o2 + pixo * (unsigned)(p.ldsx2 * 2) + (unsigned)((lane & 7) * 16), 0);
    }
  };
  auto skip_stage = [&](int half) {  // 16 pixels through the wave's private scratch: rows of 128 + 16 bytes
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < 2; ++m)
      *reinterpret_cast<uint4*>(sE + (8 * m + (lane >> 3)) * 144 + (lane & 7) * 16) = xg[m];
    __builtin_amdgcn_wave_barrier();
    if ((l32 >> 4) == half) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) xsk[kb] = *reinterpret_cast<const uint4*>(sE + (l32 & 15) * 144 + (2 * kb + h) * 16);
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto mma = [&](int c, int sl, auto NEXT_, bool loads, bool resl) {
    constexpr bool NEXT = decltype(NEXT_)::value;
    if constexpr (NEXT) act_tab(c ^ 1);
    const char* a = sA + sl * LDS_A + aoff;
    const char* w = sW + woff + c * (KC * 2);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 pf = __builtin_bit_cast(
            bf16x8, *reinterpret_cast<const uint4*>(a + ((tap / 3) * HW_ + (tap % 3)) * AROW + kb * 32));
        const bf16x8 w0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(w + tap * C * WROW + kb * 32));
        const bf16x8 w1 =
            __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(w + (tap * C + 32) * WROW + kb * 32));
#ifdef ABL2_NOMFMA
        acc[0][0] += (float)(__builtin_bit_cast(uint4, pf).x ^ __builtin_bit_cast(uint4, w0).x);
        acc[1][0] += (float)(__builtin_bit_cast(uint4, pf).y ^ __builtin_bit_cast(uint4, w1).y);
#else
        acc[0] = mfma_h32x(w0, pf, acc[0], 0, 0, 0);
        acc[1] = mfma_h32x(w1, pf, acc[1], 0, 0, 0);
#endif
        const int s = tap * 2 + kb;
        if (s < NA1 && loads) issue_one(c, s);                    // chunk q + 2 -> the set chunk q came from
        if (s >= NA1 && s < NA1 + 4 && resl)                      // the tile's residual rows
          rres[s - NA1] = ld16(rr, res_o + (unsigned)((s - NA1) * 8 * p.ldr * 2), 0);
        if constexpr (SKB > 0) {  // raw block input of source c for the folded 1x1 convolution
          if (c < NSRC) {
            if (s == NA1) skip_load(c, 0);
            if (s == 9) { skip_stage(0); skip_load(c, 1); }
            if (s == 15) skip_stage(1);
          }
        }
        if constexpr (NEXT) {
          if (s == 8) act_one(c ^ 1, 0);
          if (s == 11) act_one(c ^ 1, 1);
          if (s == 14) act_one(c ^ 1, 2);
        }
      }
    }
    if constexpr (SKB > 0) {
      if (c < NSRC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bf16x8 xf = __builtin_bit_cast(bf16x8, xsk[i]);
          acc[0] = mfma_h32x(__builtin_bit_cast(bf16x8, wsk[0][(SKB > 4 ? 4 : 0) * c + i]), xf, acc[0], 0, 0, 0);
          acc[1] = mfma_h32x(__builtin_bit_cast(bf16x8, wsk[1][(SKB > 4 ? 4 : 0) * c + i]), xf, acc[1], 0, 0, 0);
        }
      }
    }
  };
  auto prep_res = [&](int t) {
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    res_o = (unsigned)((((ty * TH + wave) * p.W + tx * TW + epx) * p.ldr + ecg * 8) * 2);
  };
  auto prep_skip = [&](int t) {
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const int pix = (ty * TH + wave) * p.W + tx * TW;  // first pixel of the wave's row
    skip_o1 = (unsigned)(pix * p.ldsx * 2);
    skip_o2 = (unsigned)(pix * p.ldsx2 * 2) + (p.sx2 ? 0u : 128u);  // one 128-channel source: its second half
  };
  auto epilogue = [&](int t) {
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const unsigned o = (unsigned)((((ty * TH + wave) * p.W + tx * TW + epx) * p.ldy + ecg * 8) * 2);
#ifdef ABL2_NOEPI
    if (acc[0][0] + acc[1][5] == 12345.678f) reinterpret_cast<float*>(p.y)[tid] = acc[0][1];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    return;
#endif
    float bz[8];
    {
      const float4 b0 = *reinterpret_cast<const float4*>(sTab + 2 * C + ecg * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(sTab + 2 * C + ecg * 8 + 4);
      bz[0] = b0.x; bz[1] = b0.y; bz[2] = b0.z; bz[3] = b0.w; bz[4] = b1.x; bz[5] = b1.y; bz[6] = b1.z; bz[7] = b1.w;
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if ((l32 >> 3) == s4) {
        char* dst = sE + (l32 & 7) * EROW + h * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + (j * 32 + g * 8) * 4) =
                make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
      }
      __builtin_amdgcn_wave_barrier();
      const float4 a0 = *reinterpret_cast<const float4*>(sE + epx * EROW + ecg * 32);
      const float4 a1 = *reinterpret_cast<const float4*>(sE + epx * EROW + ecg * 32 + 16);
      __builtin_amdgcn_wave_barrier();
      float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], osc, bz[j]);
      if (has_res) {
        const uint4 u = rres[s4];
        v[0] = fmaf(h_lo(u.x), osc, v[0]); v[1] = fmaf(h_hi(u.x), osc, v[1]);
        v[2] = fmaf(h_lo(u.y), osc, v[2]); v[3] = fmaf(h_hi(u.y), osc, v[3]);
        v[4] = fmaf(h_lo(u.z), osc, v[4]); v[5] = fmaf(h_hi(u.z), osc, v[5]);
        v[6] = fmaf(h_lo(u.w), osc, v[6]); v[7] = fmaf(h_hi(u.w), osc, v[7]);
      }
      if (has_stats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ssum[j] += v[j];
          ssq[j] = fmaf(v[j], v[j], ssq[j]);
        }
      }
      u32x4_t ov = {pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
      WS_STORE(ov, ry, o + (unsigned)(s4 * 8 * p.ldy * 2), 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  };

  WT_DECL
  sync_lds();  // weights, tables, descriptors visible
  issue(0, 0);
  if (Q > 1) issue(1, 1);
  act_tab(0);
#pragma unroll
  for (int k = 0; k < NA1; ++k) act_one(0, k);
  write(0);
  WT_MARK(4)
  for (int q = 0; q < Q; q += 2) {
    const int t = t0 + (q >> 1);
    const bool more = q + 2 < Q;
    // ---- chunk 0 (slot 0); set 1 = the tile's second chunk (issued a phase ago), set 0 <- next tile's first
    sync_lds();
    WT_MARK(0)
    if (more) prep(q + 2);
    if (has_res) prep_res(t);        // (the residual is consumed by the epilogue at the end of the next phase)
    if constexpr (SKB > 0) prep_skip(t);
    WT_MARK(1)
    mma(0, 0, std::true_type{}, more, has_res);   // activates set 1, fetches set 0 + the residual meanwhile
    WT_MARK(2)
    write(1);
    WT_MARK(3)
    // ---- chunk 1 (slot 1); set 0 = next tile's first chunk, set 1 <- next tile's second
    sync_lds();
    WT_MARK(0)
    if (more) {
      prep(q + 3);
      WT_MARK(1)
      mma(1, 1, std::true_type{}, true, false);   // activates set 0, fetches set 1 meanwhile
      WT_MARK(2)
      write(0);
      WT_MARK(3)
    } else {
      mma(1, 1, std::false_type{}, false, false);
      WT_MARK(2)
    }
    epilogue(t);
    WT_MARK(6)
  }
  if (has_stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    *reinterpret_cast<float4*>(red + tid * RED_ROW) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
    *reinterpret_cast<float4*>(red + tid * RED_ROW + 4) = make_float4(ssum[4], ssum[5], ssum[6], ssum[7]);
    *reinterpret_cast<float4*>(red + tid * RED_ROW + 8) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
    *reinterpret_cast<float4*>(red + tid * RED_ROW + 12) = make_float4(ssq[4], ssq[5], ssq[6], ssq[7]);
    __syncthreads();
    if (tid < 128) {
      const int co = tid >> 1, st = tid & 1;
      const float* src = red + (co >> 3) * RED_ROW + st * 8 + (co & 7);
      double a = 0.0;
#pragma unroll 8
      for (int i = 0; i < 64; ++i) a += (double)src[i * 8 * RED_ROW];
      ds_stat_add(p.stats + ((long)b * C + co) * 2 + st, (long long)llrint(a * (st ? DS_STAT_SQ_SCALE : DS_STAT_SUM_SCALE)));
    }
  }
  WT_MARK(5)
  WT_FLUSH
}

// ---- the network's first layer: conv3x3 8 -> 64 (ncsnpp.py:352-357; 6 real input channels padded to 8) on the same
// 8 x 32 tiles and persistent blocks.  K = 9 taps x 8 channels = 72: a 16-wide k-block is a PAIR of taps, so the B
// fragment of lane (pixel l32, half h) for k-block kb is the 16-byte pixel vector of tap 2 kb + h — read straight from a
// [340 halo pixels][16 B] LDS tile (5.4 KB, double-buffered: one barrier per tile) — and the 64 x 72 weights live in 40
// registers per lane.  10 MFMAs per wave and tile; the launch is bound by writing its 64-channel output (and its
// statistics), which the generic tile did at 1.5 TB/s (102 us at 256^2, B = 16).
struct ThinK {
  const bf16_t* x; long x_bs; int ldx;
  const bf16_t* w;              // [64][9][8]
  const float* bias;
  bf16_t* y; long y_bs; int ldy;
  long long* stats;
  int H, W, G, tiles_x, tiles_per_img;
};
constexpr int THIN_LDS = 512 * RED_ROW * 4;  // input tiles 2 x 5.4 KB + 8 epilogue scratches 18 KB; the final statistics reduce needs 40 KB
static_assert(2 * HP * 16 + 8 * LDS_E <= THIN_LDS, "LDS of the thin-input kernel");
__global__ __launch_bounds__(512, 2) void conv3x3_thin_in_kernel(ThinK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;                       // [2][HP][16 B]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l32 = lane & 31, h = lane >> 5;
  char* sE = smem + 2 * HP * 16 + wave * LDS_E;
  const int b = blockIdx.x / p.G, part = blockIdx.x % p.G;
  const int t0 = (int)((long)part * p.tiles_per_img / p.G);
  const int nt = (int)((long)(part + 1) * p.tiles_per_img / p.G) - t0;
  const __amdgpu_buffer_rsrc_t rx = rsrc(p.x + (long)b * p.x_bs, (unsigned)(p.H * p.W) * p.ldx * 2u);
  const __amdgpu_buffer_rsrc_t ry = rsrc(p.y + (long)b * p.y_bs, (unsigned)(p.H * p.W) * p.ldy * 2u);
  // weight fragments: cout 32 j + l32, tap 2 kb + h (tap 9 does not exist: zero)
  uint4 wf[2][5];
  {
    const __amdgpu_buffer_rsrc_t rw = rsrc(p.w, 64u * 9u * 8u * 2u);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kb = 0; kb < 5; ++kb) {
        const int tap = 2 * kb + h;
        wf[j][kb] = ld16(rw, tap < 9 ? (unsigned)(((32 * j + l32) * 9 + tap) * 16) : OOB, 0);
      }
  }
  float bz[8];
  const int epx = lane >> 3, ecg = lane & 7;
#pragma unroll
  for (int j = 0; j < 8; ++j) bz[j] = p.bias ? p.bias[ecg * 8 + j] : 0.f;
  // staging: thread i < HP owns halo pixel i of the tile
  const int hy = tid / HW_, hx = tid - hy * HW_;
  uint4 px = make_uint4(0, 0, 0, 0);
  auto issue = [&](int t) {
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1;
    const bool ok = tid < HP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    px = ld16(rx, ok ? (unsigned)((gy * p.W + gx) * p.ldx * 2) : OOB, 0);
  };
  float ssum[8], ssq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  const bool has_stats = p.stats != nullptr;
  issue(t0);
  for (int i = 0; i < nt; ++i) {
    const int t = t0 + i;
    char* sx = sX + (i & 1) * HP * 16;
    if (tid < HP) *reinterpret_cast<uint4*>(sx + tid * 16) = px;
    if (i + 1 < nt) issue(t + 1);
    sync_lds();  // tile i visible; the slot written next iteration was last read two barriers ago
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
      const int tap = 2 * kb + h < 9 ? 2 * kb + h : 8;  // (the missing tenth tap multiplies zero weights)
      const bf16x8 pf = __builtin_bit_cast(
          bf16x8, *reinterpret_cast<const uint4*>(sx + ((wave + tap / 3) * HW_ + l32 + tap % 3) * 16));
      acc[0] = mfma_h32x(__builtin_bit_cast(bf16x8, wf[0][kb]), pf, acc[0], 0, 0, 0);
      acc[1] = mfma_h32x(__builtin_bit_cast(bf16x8, wf[1][kb]), pf, acc[1], 0, 0, 0);
    }
    // epilogue: the wave's 32 pixels x 64 couts through its private scratch, 8 pixels at a time, as full 128-byte lines
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const unsigned o = (unsigned)((((ty * TH + wave) * p.W + tx * TW + epx) * p.ldy + ecg * 8) * 2);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if ((l32 >> 3) == s4) {
        char* dst = sE + (l32 & 7) * EROW + h * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(dst + (j * 32 + g * 8) * 4) =
                make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
      }
      __builtin_amdgcn_wave_barrier();
      const float4 a0 = *reinterpret_cast<const float4*>(sE + epx * EROW + ecg * 32);
      const float4 a1 = *reinterpret_cast<const float4*>(sE + epx * EROW + ecg * 32 + 16);
      __builtin_amdgcn_wave_barrier();
      float v[8] = {a0.x + bz[0], a0.y + bz[1], a0.z + bz[2], a0.w + bz[3], a1.x + bz[4], a1.y + bz[5], a1.z + bz[6], a1.w + bz[7]};
      if (has_stats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          ssum[j] += v[j];
          ssq[j] = fmaf(v[j], v[j], ssq[j]);
        }
      }
      u32x4_t ov = {pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
      WS_STORE(ov, ry, o + (unsigned)(s4 * 8 * p.ldy * 2), 0, 0);
    }
  }
  if (has_stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    *reinterpret_cast<float4*>(red + tid * RED_ROW) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
    *reinterpret_cast<float4*>(red + tid * RED_ROW + 4) = make_float4(ssum[4], ssum[5], ssum[6], ssum[7]);
    *reinterpret_cast<float4*>(red + tid * RED_ROW + 8) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
    *reinterpret_cast<float4*>(red + tid * RED_ROW + 12) = make_float4(ssq[4], ssq[5], ssq[6], ssq[7]);
    __syncthreads();
    if (tid < 128) {
      const int co = tid >> 1, st = tid & 1;
      const float* src = red + (co >> 3) * RED_ROW + st * 8 + (co & 7);
      double a = 0.0;
#pragma unroll 8
      for (int i = 0; i < 64; ++i) a += (double)src[i * 8 * RED_ROW];
      ds_stat_add(p.stats + ((long)b * C + co) * 2 + st, (long long)llrint(a * (st ? DS_STAT_SQ_SCALE : DS_STAT_SUM_SCALE)));
    }
  }
}

// ---- the output-pyramid heads: conv3x3 (64 | 128 | 256) -> <= 8 channels on act(GN(h)) (+ the FIR-upsampled previous
// pyramid as residual), ncsnpp.py:419-440.  With 8 couts the layer is all input handling: a block stages the activated
// 8 x 32 (+halo) tile of 64 input channels in LDS, a wave multiplies its pixel row as two 16-pixel groups with
// v_mfma_f32_16x16x32_bf16 (A = the 8 real couts padded to 16, kept in LDS: 18 KB per 64 channels), and lanes 0 - 31 of
// each group write 8 bytes of output.  Wider inputs run as consecutive 64-channel passes into the same accumulators.
// The generic tile (32 couts of MFMA work for 8, 256 VGPRs, 2 waves per SIMD) took 109 us at 256^2, B = 16.
struct ThinOutK {
  const bf16_t* x; long x_bs; int ldx; int Cin;
  const bf16_t* w; int w_chunked, w_shift;  // [Cout][9][Cin] or chunk-major [Cin/kc][9][Cout][kc]
  const float* gn_scale; const float* gn_shift;
  const long long* gn_acc; const float* gn_gamma; const float* gn_beta; int gn_cpg; float gn_inv_count; float gn_eps;
  int act;
  const float* bias;
  const bf16_t* res; long res_bs; int ldr;
  bf16_t* y; long y_bs; int ldy;
  int Cout;
  int H, W, G, tiles_x, tiles_per_img;
};
// LDS pitch of a halo pixel (64 channels) / of a (tap, cout) weight row: 160 B.  A 16 x 16 x 32 fragment read is lane = (row l16,
// k-quarter q) -> row * pitch + 16 q; ds_read_b128 serves lanes {0-3, 12-15, 20-27} (rows 0-3, 12-15 at q, rows 4-11 at q + 1)
// etc. in one pass, and those 16 addresses fall into 16 distinct 16-byte bank slots iff pitch / 16 = 2 (mod 4): with the 144 B
// of rounds 2 - 3 (right for the 32 x 32 x 16 layout of the other kernels) the counters showed 3.4 conflict cycles per read.
constexpr int TO_PROW = 64 * 2 + 32;
constexpr int TO_LDS_X = HP * TO_PROW;            // 48,960
constexpr int TO_LDS_W = 9 * 16 * TO_PROW;        // 20,736
constexpr int TO_LDS = TO_LDS_X + TO_LDS_W + 2 * 64 * 4;
// WPE = waves per SIMD the register allocation is held to: 4 = 128 VGPRs, TWO blocks per CU, so that one block's loads / activation
// / LDS writes overlap the other's MFMA phase (64 -> 6 at 256^2: 73.7 -> 58.8 us; at 134 VGPRs only one block fitted); 2 for
// launches with a single tile per block, where the few spilled registers of the tight allocation cost more than they buy
// (128 -> 6 at 64^2: 11.9 vs 13.1 us)
template <int WPE>
__global__ __launch_bounds__(512, WPE) void conv3x3_thin_out_kernel(ThinOutK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sWt = smem + TO_LDS_X;
  float* sTab = reinterpret_cast<float*>(smem + TO_LDS_X + TO_LDS_W);  // scale[64], shift[64] of the current pass
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.x / p.G, part = blockIdx.x % p.G;
  const int t0 = (int)((long)part * p.tiles_per_img / p.G);
  const int nt = (int)((long)(part + 1) * p.tiles_per_img / p.G) - t0;
  const int M = p.H * p.W;
  const __amdgpu_buffer_rsrc_t rx = rsrc(p.x + (long)b * p.x_bs, (unsigned)M * p.ldx * 2u);
  const __amdgpu_buffer_rsrc_t ry = rsrc(p.y + (long)b * p.y_bs, (unsigned)M * p.ldy * 2u);
  const __amdgpu_buffer_rsrc_t rr = rsrc(p.res ? p.res + (long)b * p.res_bs : p.y, p.res ? (unsigned)M * p.ldr * 2u : 0u);
  const __amdgpu_buffer_rsrc_t rw = rsrc(p.w, (unsigned)p.Cout * 9u * p.Cin * 2u);
  const int npass = p.Cin / 64;
  const int slot = tid & 7;  // this thread's 8 channels of a pixel in the staging pass
  constexpr int NV = (HP * 8 + 511) / 512;  // 6 vectors per thread and tile
  float bz[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bz[i] = (p.bias && 4 * q + i < p.Cout) ? p.bias[4 * q + i] : 0.f;
  const int foff = l16 * TO_PROW + q * 16;
  // the (tile, 64-channel pass) steps of the block as one sequence: the loads of step s + 1 are issued right after step
  // s's tile is in LDS, so they are in flight during its MFMAs and epilogue
  uint4 pv[NV];
  bool ok[NV];
  auto issue = [&](int s_) {
    const int t = t0 + s_ / npass, cb = (s_ % npass) * 64;
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = tid + 512 * k, pix = v >> 3;
      const int hy = pix / HW_, hx = pix - hy * HW_;
      const int gy = ty * TH + hy - 1, gx = tx * TW + hx - 1;
      ok[k] = pix < HP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      pv[k] = ld16(rx, ok[k] ? (unsigned)(((gy * p.W + gx) * p.ldx + cb + slot * 8) * 2) : OOB, 0);
    }
  };
  const int nsteps = nt * npass;
  if (nsteps > 0) issue(0);
  f32x4 acc[2];
  uint2 rvq[2] = {make_uint2(0u, 0u), make_uint2(0u, 0u)};
  for (int s_ = 0; s_ < nsteps; ++s_) {
    const int i = s_ / npass, ps = s_ - i * npass;
    const int t = t0 + i;
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    if (ps == 0) {
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    {
      const int cb = ps * 64;
      const bool new_tab = s_ == 0 || npass > 1;  // weights / GroupNorm table of this pass (resident when Cin == 64)
      uint4 wv[3];
      if (new_tab) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {  // 9 x 16 rows x 8 vectors = 1152 vectors
          const int v = tid + 512 * k, row = v >> 3, tap = row >> 4, co = row & 15;
          const int ch = cb + slot * 8;
          const unsigned vo = p.w_chunked ? (unsigned)(((((ch >> p.w_shift) * 9 + tap) * p.Cout + co) * p.w_chunked + (ch & (p.w_chunked - 1))) * 2)
                                          : (unsigned)(((co * 9 + tap) * p.Cin + ch) * 2);
          wv[k] = ld16(rw, (v < 9 * 16 * 8 && co < p.Cout) ? vo : OOB, 0);
        }
      }
      __syncthreads();  // the previous step's fragment reads are done
      if (new_tab) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int v = tid + 512 * k;
          if (v < 9 * 16 * 8) *reinterpret_cast<uint4*>(sWt + (v >> 3) * TO_PROW + slot * 16) = wv[k];
        }
        if (tid < 64) {
          const int c = cb + tid;
          float sc = 1.f, sh = 0.f;
          if (p.gn_acc) {
            const int g0 = (int)(((float)c + 0.5f) / (float)p.gn_cpg) * p.gn_cpg;
            long long ssum = 0, ssq = 0;
            for (int j = 0; j < p.gn_cpg; ++j) {
              ssum += p.gn_acc[((long)b * p.Cin + g0 + j) * 2];
              ssq += p.gn_acc[((long)b * p.Cin + g0 + j) * 2 + 1];
            }
            const double mean = (double)ssum * (1.0 / DS_STAT_SUM_SCALE) * (double)p.gn_inv_count;
            double var = (double)ssq * (1.0 / DS_STAT_SQ_SCALE) * (double)p.gn_inv_count - mean * mean;
            if (var < 0.0) var = 0.0;
            sc = (float)(1.0 / sqrt(var + (double)p.gn_eps)) * (p.gn_gamma ? p.gn_gamma[c] : 1.f);
            sh = (p.gn_beta ? p.gn_beta[c] : 0.f) - (float)mean * sc;
          } else if (p.gn_scale) {
            sc = p.gn_scale[(long)b * p.Cin + c];
            sh = p.gn_shift[(long)b * p.Cin + c];
          }
          sTab[tid] = sc;
          sTab[64 + tid] = sh;
        }
        __syncthreads();
      }
      float gsc[8], gsh[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { gsc[j] = sTab[slot * 8 + j]; gsh[j] = sTab[64 + slot * 8 + j]; }
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = tid + 512 * k;
        if (v < HP * 8) {
          uint4 r = pv[k];
          if (ok[k]) r = p.act ? gn8<true>(pv[k], gsc, gsh) : gn8<false>(pv[k], gsc, gsh);
          *reinterpret_cast<uint4*>(sX + (v >> 3) * TO_PROW + slot * 16) = r;
        }
      }
      __syncthreads();
      // the residual of this tile's outputs goes out BEFORE the next step's loads (loads return in order: the epilogue
      // then waits for these two only) and lands during the MFMAs instead of in front of the stores
      if (ps == npass - 1 && q < 2) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int m = (ty * TH + wave) * p.W + tx * TW + 16 * g + l16;
          rvq[g] = ld8(rr, (unsigned)((m * p.ldr + 4 * q) * 2));
        }
      }
      if (s_ + 1 < nsteps) issue(s_ + 1);
      // ---- wave = pixel row `wave` of the tile: two 16-pixel groups x 16 (8 real) couts
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const bf16x8 wf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sWt + tap * 16 * TO_PROW + foff + kb * 64));
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            const bf16x8 xf = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const uint4*>(sX + ((wave + tap / 3) * HW_ + 16 * g + tap % 3) * TO_PROW + foff + kb * 64));
            acc[g] = mfma_h16x(wf, xf, acc[g], 0, 0, 0);
          }
        }
    }
    if (ps != npass - 1) continue;
    // ---- epilogue: lane (pixel 16 g + l16, couts 4 q .. 4 q + 3); couts >= 8 do not exist
    if (q < 2) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int m = (ty * TH + wave) * p.W + tx * TW + 16 * g + l16;
        const uint2 rv = rvq[g];
        float v[4];
        v[0] = acc[g][0] + bz[0] + h_lo(rv.x);
        v[1] = acc[g][1] + bz[1] + h_hi(rv.x);
        v[2] = acc[g][2] + bz[2] + h_lo(rv.y);
        v[3] = acc[g][3] + bz[3] + h_hi(rv.y);
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2)
          if (4 * q + i2 >= p.Cout) v[i2] = 0.f;
        st8(ry, (unsigned)((m * p.ldy + 4 * q) * 2), make_uint2(pack_h2(v[0], v[1]), pack_h2(v[2], v[3])));
      }
    }
  }
}

int ws_blocks_per_image(const ConvArgs& a) {
  const int cus = ds_num_cus();
  const int tiles = (a.H / TH) * (a.W / TW);
  int g = cus / a.B;
  if (g < 1) g = 1;
  if (g > tiles) g = tiles;
  return g;
}

}  // namespace

// The layers this kernel takes over from conv_mfma.hip.
bool ds_conv_ws_eligible(const ConvArgs& a) {
  if (!(a.dtype == DS_BF16 && a.taps == 9 && a.Cin == C && a.Cout == C && !a.x2 && a.w_bs == 0 &&
        (a.w_chunked == 0 || a.w_chunked == KC) && a.bias_mode == 0 && !a.div_b && a.H % TH == 0 && a.W % TW == 0 &&
        a.ldx >= C && a.ldy >= C && (!a.res || a.ldr >= C)))
    return false;
  if (a.sx) {  // folded 1x1 skip convolution: 64 or 128 raw channels, whole 16-channel k-blocks per source, no residual
    const int sC1 = a.sx2 ? a.sC1 : a.sCin;
    if (!(a.sw && !a.res && (a.sCin == 64 || a.sCin == 128) && (!a.sx2 || sC1 == 64) && a.ldsx % 8 == 0 && (!a.sx2 || a.ldsx2 % 8 == 0) &&
          (a.sw_chunked == 0 || ((a.sw_chunked & (a.sw_chunked - 1)) == 0 && a.sw_chunked >= 8))))
      return false;
  }
  return true;
}

int ds_launch_conv_ws(const ConvArgs& a, hipStream_t st) {
  WsK k;
  k.x = reinterpret_cast<const bf16_t*>(a.x); k.x_bs = a.x_bs; k.ldx = a.ldx;
  k.w = reinterpret_cast<const bf16_t*>(a.w); k.w_chunked = a.w_chunked;
  k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift;
  k.bias = a.bias; k.bias_b = a.bias_b; k.bias_b_ld = a.bias_b_ld;
  k.res = reinterpret_cast<const bf16_t*>(a.res); k.res_bs = a.res_bs; k.ldr = a.ldr;
  k.out_scale = a.out_scale;
  k.y = reinterpret_cast<bf16_t*>(a.y); k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.stats = a.stats_acc;
  k.gn_acc = a.gn_acc1; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta; k.gn_groups = a.gn_groups;
  k.gn_inv_count = a.gn_inv_count; k.gn_eps = a.gn_eps;
  k.H = a.H; k.W = a.W; k.G = ws_blocks_per_image(a);
  k.tiles_x = a.W / TW; k.tiles_per_img = (a.H / TH) * (a.W / TW);
  const int mode = ((a.gn_scale || a.gn_acc1) && a.gn_act) ? 2 : 1;  // raw input = affine with scale 1, shift 0 (exact in bf16)
  k.sx = reinterpret_cast<const bf16_t*>(a.sx); k.sx_bs = a.sx_bs; k.ldsx = a.ldsx;
  k.sx2 = reinterpret_cast<const bf16_t*>(a.sx2); k.sx2_bs = a.sx2_bs; k.ldsx2 = a.sx2 ? a.ldsx2 : a.ldsx;
  k.sC1 = a.sx2 ? a.sC1 : a.sCin; k.sCin = a.sCin;
  k.sw = reinterpret_cast<const bf16_t*>(a.sw); k.sw_chunked = a.sw_chunked; k.sw_shift = a.sw_chunked ? __builtin_ctz(a.sw_chunked) : 0;
  const int skb = a.sx ? a.sCin / 16 : 0;
  const dim3 grid(a.B * k.G), block(512);
#define DS_WS_LAUNCH(M_, S_)                                                                                       \
  do {                                                                                                             \
    DS_FUNC_LDS_ONCE((conv3x3_ws1_kernel<M_, S_>), LDS_TOTAL);                                                     \
    hipLaunchKernelGGL((conv3x3_ws1_kernel<M_, S_>), grid, block, LDS_TOTAL, st, k);                               \
    ds_set_last_conv_kernel("conv3x3_ws1_kernel<" #M_ "," #S_ ">");                                                \
  } while (0)
  if (skb == 8) { if (mode == 1) DS_WS_LAUNCH(1, 8); else DS_WS_LAUNCH(2, 8); }
  else if (skb == 4) { if (mode == 1) DS_WS_LAUNCH(1, 4); else DS_WS_LAUNCH(2, 4); }
  else { if (mode == 1) DS_WS_LAUNCH(1, 0); else DS_WS_LAUNCH(2, 0); }
#undef DS_WS_LAUNCH
  DS_LAUNCH_CHECK();
  return 0;
}

// The first layer of the network: 8 (padded) input channels, no GroupNorm, no residual.
bool ds_conv_thin_eligible(const ConvArgs& a) {
  return a.dtype == DS_BF16 && a.taps == 9 && a.Cin == 8 && a.Cout == C && !a.x2 && !a.sx && a.w_bs == 0 &&
         a.w_chunked == 0 && !a.gn_scale && !a.gn_acc1 && !a.res && !a.bias_b && a.bias_mode == 0 && !a.div_b &&
         a.out_scale == 1.f && a.H % TH == 0 && a.W % TW == 0 && a.ldx % 8 == 0 && a.ldy >= C;
}

int ds_launch_conv_thin(const ConvArgs& a, hipStream_t st) {
  ThinK k;
  k.x = reinterpret_cast<const bf16_t*>(a.x); k.x_bs = a.x_bs; k.ldx = a.ldx;
  k.w = reinterpret_cast<const bf16_t*>(a.w);
  k.bias = a.bias;
  k.y = reinterpret_cast<bf16_t*>(a.y); k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.stats = a.stats_acc;
  k.H = a.H; k.W = a.W; k.G = ws_blocks_per_image(a) * 2;  // two blocks per CU
  const int tiles = (a.H / TH) * (a.W / TW);
  if (k.G > tiles) k.G = tiles;
  k.tiles_x = a.W / TW; k.tiles_per_img = tiles;
  hipLaunchKernelGGL(conv3x3_thin_in_kernel, dim3(a.B * k.G), dim3(512), THIN_LDS, st, k);
  ds_set_last_conv_kernel("conv3x3_thin_in_kernel");
  DS_LAUNCH_CHECK();
  return 0;
}

// The output-pyramid heads: <= 8 couts, 64-channel multiples in, GroupNorm (+ SiLU) on the input, optional residual.
bool ds_conv_thin_out_eligible(const ConvArgs& a) {
  return a.dtype == DS_BF16 && a.taps == 9 && a.Cout <= 8 && a.Cin % 64 == 0 && a.Cin <= 512 && !a.x2 && !a.sx && a.w_bs == 0 &&
         ((a.w_chunked & (a.w_chunked - 1)) == 0) && (a.w_chunked == 0 || a.w_chunked >= 8) && !a.bias_b && a.bias_mode == 0 &&
         !a.div_b && a.out_scale == 1.f && !a.stats_acc && a.H % TH == 0 && a.W % TW == 0 && a.ldx % 8 == 0 && a.ldy >= 8 &&
         a.ldy % 4 == 0 && (!a.res || (a.ldr >= 8 && a.ldr % 4 == 0)) &&
         (!a.gn_acc1 || (a.gn_groups > 0 && a.Cin % a.gn_groups == 0));
}

int ds_launch_conv_thin_out(const ConvArgs& a, hipStream_t st) {
  ThinOutK k;
  k.x = reinterpret_cast<const bf16_t*>(a.x); k.x_bs = a.x_bs; k.ldx = a.ldx; k.Cin = a.Cin;
  k.w = reinterpret_cast<const bf16_t*>(a.w); k.w_chunked = a.w_chunked; k.w_shift = a.w_chunked ? __builtin_ctz(a.w_chunked) : 0;
  k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift;
  k.gn_acc = a.gn_acc1; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta;
  k.gn_cpg = a.gn_acc1 ? a.Cin / a.gn_groups : 1; k.gn_inv_count = a.gn_inv_count; k.gn_eps = a.gn_eps;
  k.act = a.gn_act;
  k.bias = a.bias;
  k.res = reinterpret_cast<const bf16_t*>(a.res); k.res_bs = a.res_bs; k.ldr = a.ldr;
  k.y = reinterpret_cast<bf16_t*>(a.y); k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.Cout = a.Cout;
  const int tiles = (a.H / TH) * (a.W / TW);
  k.H = a.H; k.W = a.W; k.G = ws_blocks_per_image(a) * 2;  // two blocks per CU
  if (k.G > tiles) k.G = tiles;
  k.tiles_x = a.W / TW; k.tiles_per_img = tiles;
  if (tiles >= 2 * k.G) {
    DS_FUNC_LDS_ONCE(conv3x3_thin_out_kernel<4>, TO_LDS);
    hipLaunchKernelGGL(conv3x3_thin_out_kernel<4>, dim3(a.B * k.G), dim3(512), TO_LDS, st, k);
  } else {
    DS_FUNC_LDS_ONCE(conv3x3_thin_out_kernel<2>, TO_LDS);
    hipLaunchKernelGGL(conv3x3_thin_out_kernel<2>, dim3(a.B * k.G), dim3(512), TO_LDS, st, k);
  }
  ds_set_last_conv_kernel("conv3x3_thin_out_kernel");
  DS_LAUNCH_CHECK();
  return 0;
}
