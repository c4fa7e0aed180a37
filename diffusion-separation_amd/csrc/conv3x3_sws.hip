// conv3x3_sws.hip — 3x3 convolution of the SPLIT mode (fp32 tensors, every product as three bfloat16 MFMAs on hi / lo halves:
// hi x hi + hi x lo + lo x hi, fp32 accumulation, 2^-17 relative product error) with STREAMED weights, for the 64- / 128- / 256-cout
// layers (<= 256 input channels) of the >= 32-row levels; gfx950.  The split engine is the head of every hybrid run (pl_model dtype "hybrid": what "auto"
// ships above nf = 64) and the overflow net of the half-precision engine; until round 5 all of its >= 32-row layers ran on the
// generic tile (conv_mfma.hip, SP = 1) at ~0.27 of what the three-MFMA product allows.
// Reference: layers.py:141-156 (ddpm_conv3x3), layerspp.py:291-323 (ResnetBlockBigGANpp), ncsnpp.py:409-417 (the concat).
//
// Structure = conv3x3_sw.hip (read that header first: persistent blocks of 4 waves, one wave per SIMD, a 2-slot halo ring fed chunk
// by chunk through registers, half-phases whose epilogue runs under the other half's MFMAs, a weight ring in the accumulator file
// primed at the top of every tile), with what fp32 storage changes:
//   * a chunk is 32 channels: a pixel's chunk is still ONE full 128-byte line, a thread's staging piece 4 fp32 values.  The piece is
//     activated in fp32 (GroupNorm affine + SiLU), split into bfloat16 hi and lo = bf16(v - hi) and written as two 8-byte pieces
//     into the pixel's LDS row [32 hi | 32 lo] (the same 144-byte pitch as the 16-bit kernels);
//   * a k-step (tap, 16-channel block) is 3 MFMAs per pixel row; its weights are TWO fragments (hi, lo: 2 KB per wave) of a
//     fragment-major copy the engine prepares (ds_sws_frag_index), streamed SWS_D k-steps ahead into the accumulator file;
//   * 4 pixel rows per wave (8 accumulators would leave no room for the doubled pixel fragments): NCG = 4 cout groups on one
//     4 x 32 tile (128 couts) or NCG = 2 cout groups x 2 pixel groups on an 8 x 32 tile (64 couts);
//   * MFMA order inside a k-step: product-major, row-minor — consecutive MFMAs never share an accumulator;
//   * the epilogue stores fp32 quads straight from the accumulator layout (lane (pixel, half h): couts 8 q + 4 h .. + 3 = 16 bytes);
//   * a residual rides as skip chunks against an identity matrix the engine provides (hi = 1, lo = 0: the residual enters the fp32
//     accumulators as hi + lo, 2^-17 relative — the product error of every other term of the sum).
// The staging and epilogue arithmetic is plain C++ here (compiler-placed between the inline-asm MFMAs, one scheduling barrier per
// k-step): with 6 MFMAs per k-step and ~2 other instructions per MFMA the single wave's issue slots are not the bound they are in
// the 16-bit kernels.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef SWS_TIMING  // profiling build only: per-phase cycle totals of wave 0
__device__ unsigned long long g_sws_dbg[16];
#define RT_DECL unsigned rt_prev = (unsigned)__builtin_readcyclecounter(), rt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_MARK(i) { unsigned rt_now = (unsigned)__builtin_readcyclecounter(); rt_acc[i] += rt_now - rt_prev; rt_prev = rt_now; }
#define RT_FLUSH if (threadIdx.x == 0) { for (int q = 0; q < 8; ++q) atomicAdd(&g_sws_dbg[q], (unsigned long long)rt_acc[q]); atomicAdd(&g_sws_dbg[15], 1ull); }
extern "C" int diffsep_sws_debug_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sws_dbg), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sws_dbg), z, sizeof(z)); }
  return 0;
}
#else
#define RT_DECL
#define RT_MARK(i)
#define RT_FLUSH
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
constexpr unsigned OOB = 0x80000000u;

__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ inline u32x4_t ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
// block barrier that orders LDS traffic only (a __syncthreads() would also drain the global prefetch)
__device__ inline void sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int TW = 32, HW_ = TW + 2;  // tile width, halo row
constexpr int KC = 32;                // channels per chunk: a pixel's chunk is ONE full 128-byte line of fp32 values
constexpr int NKB = KC / 16;          // 16-channel k-blocks per tap
constexpr int KSC = 9 * NKB;          // k-steps of a 3x3 chunk
constexpr int AROW = KC * 4 + 16;     // 144 B: LDS pitch of a halo pixel: [32 bf16 hi][32 bf16 lo] + pad
constexpr int LO_OFF = KC * 2;        // byte offset of the lo plane in a pixel's row
constexpr int PPL = KC / 4;           // 16-byte pieces (4 fp32 values) per pixel
constexpr int NT = 256;
constexpr int RPW = 4, RH = RPW / 2;  // pixel rows per wave, rows per half-phase
#ifndef SWS_D
#define SWS_D 10  // ring depth in k-steps (each: a hi and a lo fragment = 8 registers)
#endif
constexpr int RING = SWS_D;

struct SwsK {
  const float* x; long x_bs; int ldx; int C1;      // channels [0, C1) from x, [C1, Cin) from x2
  const float* x2; long x2_bs; int ldx2;
  const bf16_t* wfrag; const bf16_t* swfrag;       // ds_sws_frag_index order: [k-step][hi | lo][Cout / 32][lane][8]
  unsigned frag_step;                              // bytes of one k-step: 2 planes x Cout / 32 KB
  const float* gn_scale; const float* gn_shift;    // [B][Cin] or null
  const long long* gn_acc1; const long long* gn_acc2; const float* gn_gamma; const float* gn_beta;
  int gn_groups; float gn_inv_count; float gn_eps;
  const float* bias; const float* bias_b; int bias_b_ld;
  float out_scale;
  float* y; long y_bs; int ldy;
  long long* stats;
  const float* sx; long sx_bs; int ldsx; int sC1;  // folded skip / residual: raw channels [0, sC1) from sx, the rest from sx2
  const float* sx2; long sx2_bs; int ldsx2;
  int H, W, G, ncb, cout, tiles_x, tiles_per_img;  // G blocks per image and cout block; ncb cout blocks; cout = the layer's
  int dbg;
};

// NCH / NSK: 32-channel chunks of the 3x3 input / of the folded 1x1 skip; NCG: cout groups of 32 per block (4: one pixel group,
// tile 4 x 32; 2: two pixel groups, tile 8 x 32)
template <int NCH, int NSK, int NCG>
struct SwsGeom {
  static constexpr int CO = 32 * NCG, PGN = 4 / NCG;
  static constexpr int TH = PGN * RPW, HH_ = TH + 2, HP = HH_ * HW_;
  static constexpr int NI = TH * TW * PPL / NT, NBP = HP - TH * TW, NB = (NBP * PPL + NT - 1) / NT;
  static constexpr int NL = NI + NB;
  static_assert(NT / PPL == TW && TH * TW * PPL % NT == 0, "one staging pass = one tile row");
  static constexpr int LDS_A = NL * (NT / PPL) * AROW;    // one ring slot (whole passes of the block: no predicated writes)
  static constexpr int CIN = NCH * KC, SCIN = NSK * KC;
  static constexpr int LDS_TAB = (2 * CIN + CO) * 4;      // GN scale, GN shift, (bias + temb bias) * out_scale
  static constexpr int LDS_DESC = NB * NT * 4;            // relative pixel index of the border pieces
  static_assert((NI + NB) * (NT / PPL) >= HP + NB * (NT / PPL) - NBP, "dummy pixels of the last border pass fit the slot");
  static constexpr int NPH = NCH + NSK;
  static constexpr int NKS = NCH * KSC + NSK * NKB;
  static constexpr int OFF_TAB = 2 * LDS_A, OFF_DESC = OFF_TAB + ((LDS_TAB + 15) & ~15);
  static constexpr int LDS_TOTAL = OFF_DESC + LDS_DESC;
  // chunk of phase P: [0, NCH) = 3x3 chunk, NCH + s = skip chunk s.  Order: 3x3 chunk 0, the skip chunks, the other 3x3 chunks
  // (the first and the last phase are long ones: each carries the epilogue of half a tile)
  static constexpr int chunk_of(int P) { return P == 0 ? 0 : (P <= NSK ? NCH + P - 1 : P - NSK); }
  static_assert(NCH >= 2, "a 3x3 chunk at either end of the tile");
  static constexpr int nk_of(int P) { return chunk_of(P) < NCH ? KSC : NKB; }
  static constexpr int S = 2 * NKS;  // the weight stream of one tile: position = (phase, half, k-step of the half)
  static constexpr int pos0(int P) { int s = 0; for (int q = 0; q < P; ++q) s += 2 * nk_of(q); return s; }
  static constexpr int widx(bool conv, int ks) {  // loop order ((kx, block) groups outside, ky inside) -> (tap, block)
    if (!conv) return ks;
    const int g = ks / 3, dy = ks % 3, dx = g / NKB, kb = g % NKB;
    return (dy * 3 + dx) * NKB + kb;
  }
  static constexpr int frag_of(int s) {  // k-step index into wfrag (>= 0) or -1 - index into swfrag
    int P = 0;
    while (s >= 2 * nk_of(P)) { s -= 2 * nk_of(P); ++P; }
    const int nk = nk_of(P), ks = s % nk, c = chunk_of(P);
    return c < NCH ? c * KSC + widx(true, ks) : -1 - ((c - NCH) * NKB + ks);
  }
  struct FragTab { int v[S]; };
  static constexpr FragTab frag_tab() { FragTab t{}; for (int s = 0; s < S; ++s) t.v[s] = frag_of(s); return t; }
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget of one CU");
  static_assert(NT * 36 * 4 <= LDS_TOTAL, "the statistics reduce reuses the block's LDS");
};

// fp32 value = hi + lo with hi, lo bfloat16 (round to nearest even both): as split4 of conv_mfma.hip
__device__ inline void split4(const float (&f)[4], u32x2_t& hi, u32x2_t& lo) {
  hi.x = pack_bf16x2(f[0], f[1]);
  hi.y = pack_bf16x2(f[2], f[3]);
  lo.x = pack_bf16x2(f[0] - bf_lo(hi.x), f[1] - bf_hi(hi.x));
  lo.y = pack_bf16x2(f[2] - bf_lo(hi.y), f[3] - bf_hi(hi.y));
}

// MODE: 0 raw input, 2 GroupNorm + SiLU
template <int NCH, int NSK, int MODE, int NCG>
__global__ __launch_bounds__(NT, 1) void conv3x3_sws_kernel(SwsK p) {
  using G = SwsGeom<NCH, NSK, NCG>;
  constexpr int CO = G::CO, PGN = G::PGN;
  constexpr int TH = G::TH, HP = G::HP, LDS_A = G::LDS_A, NL = G::NL, NPH = G::NPH, CIN = G::CIN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  float* sTab = reinterpret_cast<float*>(smem + G::OFF_TAB);
  int* sDesc = reinterpret_cast<int*>(smem + G::OFF_DESC);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int cg = wave % NCG, pg = wave / NCG;
  const int b = blockIdx.x / (p.G * p.ncb), cb = (blockIdx.x / p.G) % p.ncb, part = blockIdx.x % p.G;
  const int t0 = (int)((long)part * p.tiles_per_img / p.G);
  const int nt = (int)((long)(part + 1) * p.tiles_per_img / p.G) - t0;
  RT_DECL

  // ---- tables: GroupNorm scale / shift of image b, bias (as conv3x3_rw.hip: operands loaded first, tables built behind the
  // first loads)
  static_assert(CIN <= NT && CO <= NT, "one table entry per thread");
  constexpr int CPG_MAX = 8;
  long long t_s[CPG_MAX], t_q[CPG_MAX];
  float t_gam = 1.f, t_bet = 0.f, t_sc = 1.f, t_sh = 0.f, t_bias = 0.f;
#pragma unroll
  for (int j = 0; j < CPG_MAX; ++j) { t_s[j] = 0; t_q[j] = 0; }
  if (tid < CIN) {
    const int c = tid;
    if (p.gn_acc1) {
      const int C1 = p.C1, C2 = CIN - C1;
      const int cpg = CIN / p.gn_groups, g0 = (c / cpg) * cpg;
#pragma unroll
      for (int j = 0; j < CPG_MAX; ++j) {
        if (j < cpg) {
          const int cc = g0 + j;
          const long long* src = cc < C1 ? p.gn_acc1 + ((long)b * C1 + cc) * 2 : p.gn_acc2 + ((long)b * C2 + (cc - C1)) * 2;
          t_s[j] = src[0];
          t_q[j] = src[1];
        }
      }
      t_gam = p.gn_gamma ? p.gn_gamma[c] : 1.f;
      t_bet = p.gn_beta ? p.gn_beta[c] : 0.f;
    } else if (p.gn_scale) {
      t_sc = p.gn_scale[(long)b * CIN + c];
      t_sh = p.gn_shift[(long)b * CIN + c];
    }
  }
  if (tid < CO) t_bias = (p.bias ? p.bias[cb * CO + tid] : 0.f) + (p.bias_b ? p.bias_b[(long)b * p.bias_b_ld + cb * CO + tid] : 0.f);
  auto build_tables = [&]() __attribute__((always_inline)) {
    if (tid < CIN) {
      float sc = t_sc, sh = t_sh;
      if (p.gn_acc1) {
        long long t_ssum = 0, t_ssq = 0;
#pragma unroll
        for (int j = 0; j < CPG_MAX; ++j) { t_ssum += t_s[j]; t_ssq += t_q[j]; }
        const double mean = (double)t_ssum * (1.0 / DS_STAT_SUM_SCALE) * (double)p.gn_inv_count;
        double var = (double)t_ssq * (1.0 / DS_STAT_SQ_SCALE) * (double)p.gn_inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        sc = (float)(1.0 / sqrt(var + (double)p.gn_eps)) * t_gam;
        sh = t_bet - (float)mean * sc;
      }
      sTab[tid] = sc;
      sTab[CIN + tid] = sh;
    }
    if (tid < CO) sTab[2 * CIN + tid] = t_bias * p.out_scale;
  };
  const int slot = tid & (PPL - 1);  // this thread's 4 channels of a chunk: 4 slot ..
  // staging pieces of this thread (16 bytes = 4 fp32 channels of one pixel): piece k < NI = pixel (row k, column tid / PPL) of the
  // tile itself; piece NI + kb = border pixel tid / PPL + 32 kb of the halo line (top row, bottom row, left column, right column);
  // border pieces carry 5 flag bits (top / bottom / left / right halo line, past the last border pixel)
  constexpr int NI = G::NI, NB = G::NB, NBP = G::NBP;
  const int ixp = tid / PPL;
  unsigned fl = 0;
  int dstb[NB];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    const int bi = ixp + (NT / PPL) * kb;
    int hy, hx;
    if (bi < HW_) { hy = 0; hx = bi; }
    else if (bi < 2 * HW_) { hy = G::HH_ - 1; hx = bi - HW_; }
    else if (bi < 2 * HW_ + TH) { hy = 1 + bi - 2 * HW_; hx = 0; }
    else { hy = 1 + bi - 2 * HW_ - TH; hx = HW_ - 1; }
    const bool in = bi < NBP;
    const unsigned flg = in ? (hy == 0 ? 1u : 0u) | (hy == G::HH_ - 1 ? 2u : 0u) | (hx == 0 ? 4u : 0u) | (hx == HW_ - 1 ? 8u : 0u) : 16u;
    sDesc[kb * NT + tid] = in ? (hy - 1) * p.W + (hx - 1) : 0;
    dstb[kb] = (in ? hy * HW_ + hx : HP + (bi - NBP)) * AROW + slot * 8;  // (hi piece; the lo piece LO_OFF further)
    fl |= flg << (5 * kb);
  }
  const int ldi0 = (HW_ + 1 + ixp) * AROW + slot * 8;  // tile pixel (0, ixp): piece k < NI is one halo row further

  RT_MARK(6)
  const int M = p.H * p.W;
  const __amdgpu_buffer_rsrc_t rx1 = rsrc(p.x + (long)b * p.x_bs, (unsigned)M * p.ldx * 4u);
  const __amdgpu_buffer_rsrc_t rx2 = p.x2 ? rsrc(p.x2 + (long)b * p.x2_bs, (unsigned)M * p.ldx2 * 4u) : rx1;
  const __amdgpu_buffer_rsrc_t rs1 = NSK ? rsrc(p.sx + (long)b * p.sx_bs, (unsigned)M * p.ldsx * 4u) : rx1;
  const __amdgpu_buffer_rsrc_t rs2 = (NSK && p.sx2) ? rsrc(p.sx2 + (long)b * p.sx2_bs, (unsigned)M * p.ldsx2 * 4u) : rs1;
  const __amdgpu_buffer_rsrc_t ry = rsrc(p.y + (long)b * p.y_bs, (unsigned)M * p.ldy * 4u);

  // ---- the weight stream: the (hi, lo) fragments of stream position s -> ring slot s % RING, loaded RING positions ahead
  u32x4_t rhi[RING], rlo[RING];
  const __amdgpu_buffer_rsrc_t rw = rsrc(p.wfrag, (unsigned)(NCH * KSC) * p.frag_step);
  const __amdgpu_buffer_rsrc_t rsw = NSK ? rsrc(p.swfrag, (unsigned)(NSK * NKB) * p.frag_step) : rw;
  const unsigned vfrag = (unsigned)(((cb * NCG + cg) * 64 + lane) * 16);
  const unsigned lo_step = p.frag_step >> 1;
  auto load_frag = [&](int s) __attribute__((always_inline)) {  // (s: compile time)
    constexpr typename G::FragTab FT = G::frag_tab();
    const int f = FT.v[s];
    if (f >= 0) {
      rhi[s % RING] = ld16(rw, vfrag, (unsigned)f * p.frag_step);
      rlo[s % RING] = ld16(rw, vfrag, (unsigned)f * p.frag_step + lo_step);
    } else {
      rhi[s % RING] = ld16(rsw, vfrag, (unsigned)(-1 - f) * p.frag_step);
      rlo[s % RING] = ld16(rsw, vfrag, (unsigned)(-1 - f) * p.frag_step + lo_step);
    }
  };

  // ---- staging state: pa[] holds the chunk AFTER the one in LDS (in flight or landed)
  u32x4_t pa[NL];
  struct TileG { int pix0; unsigned edge; };
  auto geom_at = [&](int ty, int tx, bool valid) {
    TileG g;
    const int y0 = ty * TH, x0 = tx * TW;
    g.edge = (y0 == 0 ? 1u : 0u) | (y0 + TH == p.H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == p.W ? 8u : 0u);
    // (no tile: the first pixel PAST the image — every offset pixel x pitch is then >= the tensor's size, and stays below 2^32 for
    // every pitch; a constant like 0x3fffff times a 1 KB pitch plus a tile offset WRAPS into the tensor)
    g.pix0 = valid ? y0 * p.W + x0 : M;
    return g;
  };
  auto tile_geom = [&](int i) {
    const int t = t0 + i;
    const int ty = t / p.tiles_x;
    return geom_at(ty, t - ty * p.tiles_x, i < nt);
  };
  auto piece_ok = [&](auto P_, const TileG& g, int k) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
    const unsigned em = (P < NCH ? (g.edge | 16u) : 31u) << (5 * (k - NI));  // (scalar; border pieces only)
    return (fl & em) == 0u;
  };
  auto issue_one = [&](auto P_, const TileG& g, int k, int rel) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
    constexpr bool CONV = P < NCH;
    constexpr int CB = (CONV ? P : P - NCH) * KC;
    const int c1 = CONV ? p.C1 : p.sC1;
    const bool second = CB >= c1;  // (wave-uniform)
    const unsigned ld4 = (unsigned)(CONV ? (second ? p.ldx2 : p.ldx) : (second ? p.ldsx2 : p.ldsx)) * 4u;
    const unsigned co4 = (unsigned)((second ? CB - c1 : CB) * 4) + (unsigned)slot * 16u;
    const __amdgpu_buffer_rsrc_t r = CONV ? (second ? rx2 : rx1) : (second ? rs2 : rs1);
    if constexpr (!CONV) {
      if (k >= NI) return;  // (a skip chunk meets the centre tap only: its border pieces are never read)
    }
    if (k < NI) {
      const unsigned off = __umul24((unsigned)(g.pix0 + k * p.W + ixp), ld4) + co4;
      pa[k] = ld16(r, off, 0);
      return;
    }
    const unsigned off = __umul24((unsigned)(rel + g.pix0), ld4) + co4;
    pa[k] = ld16(r, piece_ok(P_, g, k) ? off : OOB, 0);
  };
  // every input of the launch is activated: the activation may leave a constant factor (-1 / ln 2) to the epilogue
  constexpr bool FOLD = MODE == 2 && NSK == 0;
  float gsc[4], gsh[4];
  auto act_tab = [&](int c) __attribute__((always_inline)) {  // scale / shift of this thread's 4 channels of chunk c
    if constexpr (MODE != 0) {
      const float4 s0 = *reinterpret_cast<const float4*>(sTab + c * KC + slot * 4);
      const float4 h0 = *reinterpret_cast<const float4*>(sTab + CIN + c * KC + slot * 4);
      const float k = FOLD ? -1.4426950408889634f : 1.f;
      gsc[0] = s0.x * k; gsc[1] = s0.y * k; gsc[2] = s0.z * k; gsc[3] = s0.w * k;
      gsh[0] = h0.x * k; gsh[1] = h0.y * k; gsh[2] = h0.z * k; gsh[3] = h0.w * k;
    }
  };
  // Staging runs in UNITS of one piece: activate its 4 values, split them, write the two planes, re-issue the registers as the
  // load of the chunk after next
  auto unit = [&](auto P1_, auto P2_, const TileG& g1, const TileG& g2, int sl, int k, int rel) __attribute__((always_inline)) {
    constexpr int P1 = decltype(P1_)::value;
    float v[4] = {__uint_as_float(pa[k].x), __uint_as_float(pa[k].y), __uint_as_float(pa[k].z), __uint_as_float(pa[k].w)};
    if constexpr (P1 < NCH && MODE != 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = fmaf(v[j], gsc[j], gsh[j]);
        // FOLD: the affine carries the factor -log2(e): z IS the exponent of the sigmoid's exp2 and the staged value is
        // silu(GN(x)) / -ln 2; the epilogue multiplies the accumulators back
        const float e = __builtin_amdgcn_exp2f(FOLD ? z : z * -1.4426950408889634f);
        v[j] = z * __builtin_amdgcn_rcpf(1.0f + e);
      }
    }
    u32x2_t hi, lo;
    split4(v, hi, lo);
    if (P1 < NCH && MODE != 0 && k >= NI) {  // zero padding stays zero (silu(GN(0)) != 0): border pieces only
      const bool ok = piece_ok(P1_, g1, k);
      hi.x = ok ? hi.x : 0u; hi.y = ok ? hi.y : 0u; lo.x = ok ? lo.x : 0u; lo.y = ok ? lo.y : 0u;
    }
    if (P1 < NCH || k < NI) {  // (border pieces of a skip chunk: nothing was loaded, nothing is read)
      char* d = sA + sl * LDS_A + (k < NI ? ldi0 + k * HW_ * AROW : dstb[k < NI ? 0 : k - NI]);
      *reinterpret_cast<u32x2_t*>(d) = hi;
      *reinterpret_cast<u32x2_t*>(d + LO_OFF) = lo;
    }
    issue_one(P2_, g2, k, rel);
  };

  // the wave's accumulators live in the accumulator half of the register file; they start UNDEFINED (see conv3x3_sw.hip)
  f32x16 acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) asm volatile("" : "=a"(acc[r]));
  float ssum[16], ssq[16];  // per lane: its 16 couts (8 q + 4 h + i), summed over its pixels
#pragma unroll
  for (int j = 0; j < 16; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  const bool has_stats = p.stats != nullptr;
  const float osc = FOLD ? p.out_scale * -0.6931471805599453f : p.out_scale;
  // fragment base of this lane: pixel (row pg * RPW, column l32) of the halo tile, k-half h (8 channels = 16 bytes of a plane)
  const int fbase = (pg * RPW * HW_ + l32) * AROW + h * 16;
  int relreg[NB];
  float4 breg[4];  // (bias + temb bias) * out_scale of the lane's quads q = 0 .. 3: couts 8 q + 4 h ..

  // ---- epilogue of row r (of tile g) in the accumulator layout: lane (pixel l32, half h) holds the cout quads 8 q + 4 h .. + 3 of
  // the wave's 32 couts = four 16-byte stores
  auto epi_quad = [&](const TileG& g, int r, int q) __attribute__((always_inline)) {
    const float4 t = breg[q];
    float4 v;
    v.x = fmaf(acc[r][4 * q + 0], osc, t.x);
    v.y = fmaf(acc[r][4 * q + 1], osc, t.y);
    v.z = fmaf(acc[r][4 * q + 2], osc, t.z);
    v.w = fmaf(acc[r][4 * q + 3], osc, t.w);
    ssum[4 * q + 0] += v.x; ssq[4 * q + 0] = fmaf(v.x, v.x, ssq[4 * q + 0]);
    ssum[4 * q + 1] += v.y; ssq[4 * q + 1] = fmaf(v.y, v.y, ssq[4 * q + 1]);
    ssum[4 * q + 2] += v.z; ssq[4 * q + 2] = fmaf(v.z, v.z, ssq[4 * q + 2]);
    ssum[4 * q + 3] += v.w; ssq[4 * q + 3] = fmaf(v.w, v.w, ssq[4 * q + 3]);
    const int pix = g.pix0 + (pg * RPW + r) * p.W + l32;
    const unsigned o = __umul24((unsigned)pix, (unsigned)p.ldy * 4u) + (unsigned)(((cb * NCG + cg) * 32 + 8 * q + 4 * h) * 4);
    u32x4_t ov = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(ov, ry, o, 0, 0);
  };

  // ---- one HALF of a phase: the MFMAs of chunk (phase P, ring slot P & 1) for the wave's rows [HF * RH, HF * RH + RH), with its
  // share of the staging of the next phase's chunk and (EPI) the epilogue of the OTHER half's rows
  auto half = [&](auto P_, auto HF_, auto EPI_, int slot_r, const TileG& ge, const TileG& g1, const TileG& g2) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value, HF = decltype(HF_)::value;
    constexpr bool EPI = decltype(EPI_)::value;
    constexpr int C = G::chunk_of(P);
    constexpr bool CONV = C < NCH;
    constexpr int NK = CONV ? KSC : NKB;
    constexpr int C1 = G::chunk_of((P + 1) % NPH), C2 = G::chunk_of((P + 2) % NPH);
    constexpr int R0 = HF * RH, ER0 = HF ? 0 : RH;
    const char* fb = sA + slot_r * LDS_A + fbase + R0 * HW_ * AROW;
    if constexpr (EPI) {  // the rows this half finishes were last written by asm MFMAs: 12 wait states before they are read
      static_assert(RH == 2, "one guard for the half's two accumulators");
      asm volatile("s_nop 11" : "+a"(acc[ER0]), "+a"(acc[ER0 + 1]));
    }
    if constexpr (HF == 0 && C1 < NCH) act_tab(C1);
    // K order inside a 3x3 chunk: (kx, 16-channel block) groups outside, ky inside: the RH + 2 input-row fragment PAIRS (hi, lo) of a
    // group serve three k-steps.  (A skip chunk has one k-step per group: the centre tap.)
    constexpr int SUB = CONV ? 3 : 1, RFN = CONV ? RH + 2 : RH, NG = NK / SUB;
    auto ldg = [&](int g, int j, int plane) __attribute__((always_inline)) {
      const int dx = CONV ? g / NKB : 1, kb = g % NKB, row = CONV ? j : j + 1;
      return *reinterpret_cast<const u32x4_t*>(fb + (row * HW_ + dx) * AROW + plane * LO_OFF + kb * 32);
    };
    u32x4_t rfh[2][RFN], rfl[2][RFN];
#pragma unroll
    for (int j = 0; j < RFN; ++j) { rfh[0][j] = ldg(0, j, 0); rfl[0][j] = ldg(0, j, 1); }
    constexpr int SP0 = G::pos0(P) + HF * NK;  // stream position of this half's first k-step
    // staging pieces / epilogue quads of this half, spread over its k-steps: half-phase HF carries pieces [HF NL / 2 ..)
    constexpr int K0 = HF ? NL / 2 : 0, K1 = HF ? NL : NL / 2, NPC = K1 - K0;
    constexpr int NEQ = RH * 4;  // epilogue quads of an EPI half
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const u32x4_t whi = rhi[(SP0 + ks) % RING], wlo = rlo[(SP0 + ks) % RING];
      const int gq = (ks / SUB) & 1, o = CONV ? ks % SUB : 0;
      // product-major, row-minor: consecutive MFMAs never share an accumulator
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
        for (int r = 0; r < RH; ++r) {
          const u32x4_t& wv = pr == 2 ? wlo : whi;
          const u32x4_t& av = pr == 1 ? rfl[gq][r + o] : rfh[gq][r + o];
          if (P == 0 && ks == 0 && pr == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[R0 + r]) : "a"(wv), "v"(av));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[R0 + r]) : "a"(wv), "v"(av));
          // ---- the gap behind this MFMA: the next group's fragment pairs, one plane per MFMA slot
          const int g = ks / SUB, slotq = ((ks % SUB) * 3 + pr) * RH + r;
          if (g + 1 < NG && slotq < 2 * RFN) {
            if (slotq & 1) rfl[(g + 1) & 1][slotq >> 1] = ldg(g + 1, slotq >> 1, 1);
            else rfh[(g + 1) & 1][slotq >> 1] = ldg(g + 1, slotq >> 1, 0);
          }
        }
      }
      // this k-step's share of the staging (pieces) ...
#pragma unroll
      for (int k = K0 + ks * NPC / NK; k < K0 + (ks + 1) * NPC / NK; ++k) {
        const int rel = k >= NI ? relreg[k >= NI ? k - NI : 0] : 0;
        unit(std::integral_constant<int, C1>{}, std::integral_constant<int, C2>{}, g1, g2, slot_r ^ 1, k, rel);
      }
      // ... and of the other half's epilogue
      if constexpr (EPI) {
#pragma unroll
        for (int e = ks * NEQ / NK; e < (ks + 1) * NEQ / NK; ++e) epi_quad(ge, ER0 + e / 4, e % 4);
      }
      // this k-step's ring slot is free: the fragment pair RING positions ahead
      if (SP0 + ks + RING < G::S) load_frag(SP0 + ks + RING);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: the first phase's chunk into slot 0, the second phase's chunk in flight
  TileG gc = tile_geom(0);
  constexpr int CH0 = G::chunk_of(0), CH1 = G::chunk_of(1 % NPH);
  const TileG g1st = gc;  // (NPH >= 2: the second phase belongs to the same tile)
  {
#pragma unroll
    for (int k = 0; k < NL; ++k) issue_one(std::integral_constant<int, CH0>{}, gc, k, k < NI ? 0 : sDesc[(k < NI ? 0 : k - NI) * NT + tid]);
    build_tables();
    sync_lds();  // tables visible
    RT_MARK(7)
#pragma unroll
    for (int k = 0; k < NB; ++k) relreg[k] = sDesc[k * NT + tid];
#pragma unroll
    for (int q = 0; q < 4; ++q) breg[q] = *reinterpret_cast<const float4*>(sTab + 2 * CIN + cg * 32 + 8 * q + 4 * h);
  }
  TileG gp = tile_geom(nt);  // "previous tile" of the first one: no tile (its stores fall outside every tensor)
  int ph = 0;                 // phases done: the chunk of phase ph sits in ring slot ph & 1
  TileG gnc = tile_geom(1);
  int ty2 = (t0 + 2) / p.tiles_x, tx2 = (t0 + 2) - ty2 * p.tiles_x;
  for (int i = 0; i < nt; ++i) {
    const TileG gn = gnc, gnn = geom_at(ty2, tx2, i + 2 < nt);
    // the tile's first RING weight fragment pairs (the ring is not carried across the loop's back edge)
#pragma unroll
    for (int s = 0; s < RING && s < G::S; ++s) load_frag(s);
    if (i == 0) {  // the block's first chunk: activated and written in one go, under the first fragments' flight
      act_tab(CH0);
#pragma unroll
      for (int k = 0; k < NL; ++k)
        unit(std::integral_constant<int, CH0>{}, std::integral_constant<int, CH1>{}, gc, g1st, 0, k, k < NI ? 0 : relreg[k < NI ? 0 : k - NI]);
      RT_MARK(0)
    }
    auto run = [&](auto self, auto P_) __attribute__((always_inline)) {
      constexpr int P = decltype(P_)::value;
      sync_lds();
      RT_MARK(1)
      const int slot_r = ph & 1;
      half(P_, std::integral_constant<int, 0>{}, std::integral_constant<bool, P == 0>{}, slot_r, gp, ((P + 1) / NPH == 0 ? gc : gn), ((P + 2) / NPH == 0 ? gc : ((P + 2) / NPH == 1 ? gn : gnn)));
      if constexpr (P == 0) {
        if (i == 0) {  // (the first tile has no predecessor: what that epilogue summed up was not an output)
#pragma unroll
          for (int j = 0; j < 16; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
        }
      }
      half(P_, std::integral_constant<int, 1>{}, std::integral_constant<bool, P == NPH - 1>{}, slot_r, gc, ((P + 1) / NPH == 0 ? gc : gn), ((P + 2) / NPH == 0 ? gc : ((P + 2) / NPH == 1 ? gn : gnn)));
      ++ph;
      RT_MARK(G::chunk_of(P) < NCH ? 2 : 3)
      if constexpr (P + 1 < NPH) self(self, std::integral_constant<int, P + 1>{});
    };
    run(run, std::integral_constant<int, 0>{});
    gp = gc;
    gc = gn;
    gnc = gnn;
    if (++tx2 == p.tiles_x) { tx2 = 0; ++ty2; }
  }
  // the second half of the last tile's rows
  asm volatile("s_nop 11" : "+a"(acc[RH]), "+a"(acc[RH + 1]));
#pragma unroll
  for (int e = 0; e < RH * 4; ++e) epi_quad(gp, RH + e / 4, e % 4);
  RT_MARK(4)
  if (has_stats) {
    __syncthreads();
    // per lane 16 couts (cg * 32 + 8 q + 4 h + i) of pixel column l32 of its rows: sum over the 32 columns and the pixel groups
    float* red = reinterpret_cast<float*>(smem);  // [256 threads][32 (+4 pad)]
    constexpr int RED_ROW = 36;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(red + tid * RED_ROW + 4 * q) = make_float4(ssum[4 * q], ssum[4 * q + 1], ssum[4 * q + 2], ssum[4 * q + 3]);
      *reinterpret_cast<float4*>(red + tid * RED_ROW + 16 + 4 * q) = make_float4(ssq[4 * q], ssq[4 * q + 1], ssq[4 * q + 2], ssq[4 * q + 3]);
    }
    __syncthreads();
    if (tid < 2 * CO) {
      const int co = tid >> 1, st = tid & 1;
      const int wcg = co >> 5, c32 = co & 31, q = c32 >> 3, hh = (c32 >> 2) & 1, i = c32 & 3;
      double a = 0.0;
      for (int wpg = 0; wpg < PGN; ++wpg)
        for (int l = 0; l < 32; ++l) {
          const int t = (wpg * NCG + wcg) * 64 + hh * 32 + l;
          a += (double)red[t * RED_ROW + st * 16 + 4 * q + i];
        }
      ds_stat_add(p.stats + ((long)b * p.cout + cb * CO + co) * 2 + st, (long long)llrint(a * (st ? DS_STAT_SQ_SCALE : DS_STAT_SUM_SCALE)));
    }
  }
  RT_MARK(5)
  RT_FLUSH
}

template <int NCH, int NSK, int MODE, int NCG>
int sws_launch(const SwsK& k0, const ConvArgs& a, hipStream_t st) {
  using G = SwsGeom<NCH, NSK, NCG>;
  SwsK k = k0;
  const int tiles = (a.H / G::TH) * (a.W / TW);
  int g = ds_num_cus() / (a.B * k.ncb);
  if (g < 1) g = 1;
  if (g > tiles) g = tiles;
  k.G = g;
  k.tiles_x = a.W / TW;
  k.tiles_per_img = tiles;
  auto kern = conv3x3_sws_kernel<NCH, NSK, MODE, NCG>;
  DS_FUNC_LDS_ONCE(kern, G::LDS_TOTAL);
  hipLaunchKernelGGL(kern, dim3(a.B * k.ncb * k.G), dim3(NT), G::LDS_TOTAL, st, k);
  DS_LAUNCH_CHECK();
  {
    static char name[64] = {0};
    if (!name[0]) snprintf(name, sizeof(name), "conv3x3_sws_kernel<%d,%d,%d,%d>", NCH, NSK, MODE, NCG);
    ds_set_last_conv_kernel(name);
  }
  return 0;
}

// (Cin / 32, skip or residual channels / 32) pairs instantiated per cout width: the layers of the >= 32-row levels of nf = 64 / 128
template <int NCG>
int sws_dispatch(const SwsK& k, const ConvArgs& a, int nch, int nsk, int mode, hipStream_t st) {
#define SWS_CASE(NCH_, NSK_, MODE_) if (nch == NCH_ && nsk == NSK_ && mode == MODE_) return sws_launch<NCH_, NSK_, MODE_, NCG>(k, a, st)
  SWS_CASE(2, 0, 0); SWS_CASE(4, 0, 0); SWS_CASE(8, 0, 0);                     // raw input (behind a resampling)
  SWS_CASE(2, 0, 2); SWS_CASE(4, 0, 2); SWS_CASE(6, 0, 2); SWS_CASE(8, 0, 2);  // Conv_0 of plain blocks (one tensor or a concat)
  SWS_CASE(2, 2, 2); SWS_CASE(2, 4, 2); SWS_CASE(2, 6, 2);                      // 64 -> 64 + residual / skip on 64, 128, 192 raw channels
  SWS_CASE(4, 2, 2); SWS_CASE(4, 4, 2); SWS_CASE(4, 6, 2); SWS_CASE(4, 8, 2);  // 128 -> 128 + residual / skip on 64 .. 256 raw channels
#undef SWS_CASE
  return -1;
}

}  // namespace

// The launches this kernel takes: fp32 tensors in split mode, 3x3, 64 / 128 / 256 couts, input channels and skip channels in the
// instantiated set (sws_dispatch), whole tiles (W % 32 == 0, H % 8 == 0), fragment-major hi / lo weight copies at hand; a residual
// needs the identity copy (ConvArgs.ident_frag).
static bool sws_shape(int Cout, int nch, int nsk, int mode) {
  if (!(Cout == 64 || Cout == 128 || Cout == 256)) return false;  // (256: two cout blocks of 128)
  if (mode == 0) return nsk == 0 && (nch == 2 || nch == 4 || nch == 8);
  if (nsk == 0) return nch == 2 || nch == 4 || nch == 6 || nch == 8;
  if (nch == 2) return nsk == 2 || nsk == 4 || nsk == 6;
  if (nch == 4) return nsk == 2 || nsk == 4 || nsk == 6 || nsk == 8;
  return false;
}
bool ds_conv_sws_supported(const ConvArgs& a) {
  if (!(a.dtype == DS_F32 && a.split && a.taps == 9 && a.Cin % KC == 0 && a.w_frag && a.w_bs == 0 && a.bias_mode == 0 && !a.div_b &&
        a.W % TW == 0 && a.H % 8 == 0 && a.H >= 8 && a.ldy >= a.Cout && a.ldy % 4 == 0 && a.Cout <= 256))
    return false;
  if (a.x2 ? !(a.C1 % KC == 0 && a.C1 > 0 && a.C1 < a.Cin && a.ldx % 4 == 0 && a.ldx2 % 4 == 0) : a.ldx % 4 != 0) return false;
  const bool gn = a.gn_scale || a.gn_acc1;
  if (gn && !a.gn_act) return false;
  if (a.gn_acc1 && !(a.gn_groups > 0 && a.Cin % a.gn_groups == 0 && a.Cin / a.gn_groups <= 8 && (!a.x2 || a.gn_acc2))) return false;
  int nsk = 0;
  if (a.sx) {
    if (!(a.sw && a.sw_frag && !a.res && a.sCin % KC == 0 && a.ldsx % 4 == 0 &&
          (!a.sx2 || (a.sC1 % KC == 0 && a.sC1 > 0 && a.sC1 < a.sCin && a.ldsx2 % 4 == 0))))
      return false;
    nsk = a.sCin / KC;
  } else if (a.res) {
    if (!(a.ident_frag && a.ldr >= a.Cout && a.ldr % 4 == 0)) return false;
    nsk = a.Cout / KC;
  }
  return sws_shape(a.Cout, a.Cin / KC, nsk, gn ? 2 : 0);
}
bool ds_conv_sws_eligible(const ConvArgs& a) {
  if ((a.opts & DS_OPT_NO_SWS) || !ds_conv_sws_supported(a)) return false;
  // at least one tile per two compute units (nf = 64 at 32^2, B = 16: 128 blocks of 1728 MFMAs per wave, ~40 us, against 93 us on
  // the generic tile, which splits the same work over 256 blocks but re-streams and re-splits the weights through LDS)
  const long tiles = (long)a.B * (a.H / (a.Cout == 64 ? 8 : 4)) * (a.W / TW) * (a.Cout == 64 ? 1 : a.Cout / 128);
  return 2 * tiles >= ds_num_cus() || (a.opts & DS_OPT_RW_SMALL);
}

int ds_launch_conv_sws(const ConvArgs& a, hipStream_t st) {
  SwsK k;
  k.x = reinterpret_cast<const float*>(a.x); k.x_bs = a.x_bs; k.ldx = a.ldx; k.C1 = a.x2 ? a.C1 : a.Cin;
  k.x2 = reinterpret_cast<const float*>(a.x2); k.x2_bs = a.x2_bs; k.ldx2 = a.x2 ? a.ldx2 : a.ldx;
  k.wfrag = reinterpret_cast<const bf16_t*>(a.w_frag);
  k.swfrag = nullptr;
  k.frag_step = (unsigned)(a.Cout / 32) * 2048u;
  k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift;
  k.gn_acc1 = a.gn_acc1; k.gn_acc2 = a.gn_acc2; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta;
  k.gn_groups = a.gn_groups; k.gn_inv_count = a.gn_inv_count; k.gn_eps = a.gn_eps;
  k.bias = a.bias; k.bias_b = a.bias_b; k.bias_b_ld = a.bias_b_ld;
  k.out_scale = a.out_scale;
  k.y = reinterpret_cast<float*>(a.y); k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.stats = a.stats_acc;
  k.sx = nullptr; k.sx_bs = 0; k.ldsx = 0; k.sC1 = 0; k.sx2 = nullptr; k.sx2_bs = 0; k.ldsx2 = 0;
  int nsk = 0;
  if (a.sx) {
    k.sx = reinterpret_cast<const float*>(a.sx); k.sx_bs = a.sx_bs; k.ldsx = a.ldsx; k.sC1 = a.sx2 ? a.sC1 : a.sCin;
    k.sx2 = reinterpret_cast<const float*>(a.sx2); k.sx2_bs = a.sx2_bs; k.ldsx2 = a.sx2 ? a.ldsx2 : a.ldsx;
    k.swfrag = reinterpret_cast<const bf16_t*>(a.sw_frag);
    nsk = a.sCin / KC;
  } else if (a.res) {  // the residual [B][H][W][Cout] as a folded skip against the identity matrix
    k.sx = reinterpret_cast<const float*>(a.res); k.sx_bs = a.res_bs; k.ldsx = a.ldr; k.sC1 = a.Cout; k.ldsx2 = a.ldr;
    k.swfrag = reinterpret_cast<const bf16_t*>(a.ident_frag);
    nsk = a.Cout / KC;
  }
  k.H = a.H; k.W = a.W; k.G = 0; k.cout = a.Cout; k.tiles_x = 0; k.tiles_per_img = 0;
  k.dbg = 0;
  const int mode = ((a.gn_scale || a.gn_acc1) && a.gn_act) ? 2 : 0;
  int rc;
  if (a.Cout == 64) {
    k.ncb = 1;
    rc = sws_dispatch<2>(k, a, a.Cin / KC, nsk, mode, st);
  } else {
    k.ncb = a.Cout / 128;
    rc = sws_dispatch<4>(k, a, a.Cin / KC, nsk, mode, st);
  }
  DS_CHECK(rc >= 0, "conv3x3_sws: shape outside the instantiated set");
  return rc;
}
