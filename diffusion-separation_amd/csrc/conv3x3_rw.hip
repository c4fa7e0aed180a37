// conv3x3_rw.hip — 3x3 convolution with REGISTER-resident weights for the 16-bit layers of the large levels: 64 couts
// (64 -> 64, cat(64, 64) -> 64, optionally with the folded 1x1 skip of ResnetBlockBigGANpp on 64 / 128 raw channels) and
// 128 -> 128 (optional folded 64 / 128-channel skip or residual); gfx950, bfloat16 or (-DDS_HALF_F16) fp16 storage.
// Reference: layers.py:141-156 (ddpm_conv3x3), layerspp.py:291-323 (ResnetBlockBigGANpp), ncsnpp.py:411 (the concat).
//
// Why another kernel: the generic tile (conv_mfma.hip) re-streams the weights for every 256-pixel tile and runs
// 2 waves per SIMD in lock step (load -> activate -> LDS write -> barrier -> MFMA -> epilogue add up: 0.21-0.24 of the
// MFMA peak); the weight-stationary kernel (conv3x3_ws.hip) keeps the weights in LDS and spends its MFMA phase on LDS
// fragment traffic (1.5 reads per MFMA).  Here:
//
//   * one block of 4 waves per CU, ONE wave per SIMD with the whole 512-entry register file.  A wave owns 32 couts
//     (cg) and keeps ALL of their weight fragments in registers for the whole launch (144 registers per 64 input
//     channels, pinned to the accumulator half of the file by inline-asm MFMAs with "a" operands) — no weight traffic
//     at all after the prologue, through HBM, L2, L1 or LDS.  What the file does not hold beside the accumulators (the
//     last k-steps of the 128-channel layers, the skip fragments of the 128-cout variant) is read from LDS one k-step
//     ahead (RwGeom::NWL);
//   * NCG = 2 (64 couts): the block's waves are 2 cout groups x 2 pixel groups of RPW rows x 32 pixels, tile (2 RPW) x 32;
//     NCG = 4 (128 couts): 4 cout groups on ONE pixel group, tile RPW x 32 — the same staging and epilogue work per wave
//     for twice the MFMAs.  Every weight fragment is used by RPW / 2 consecutive MFMAs on independent accumulators; the
//     only LDS traffic of the MFMA stream is the pixel fragments: K runs (kx, 16-channel block) outside and ky inside, so
//     the RPW / 2 + 2 input-row fragments of a group serve three k-steps (2/3 of a ds_read_b128 per MFMA);
//   * persistent blocks walk a contiguous raster range of tiles of one image; the input goes chunk by chunk (KC = 64
//     channels: a pixel's chunk is one full 128-byte line) through a 2-slot halo ring: while the MFMAs of chunk q run,
//     the same wave activates chunk q + 1 in registers (GroupNorm affine + SiLU, one dword = "unit" at a time), writes
//     it to the other slot and re-issues the freed registers as the loads of chunk q + 2 — a global load has a whole
//     chunk phase to land, and one LDS-only barrier per chunk is the only synchronisation;
//     a thread's staging pieces are first the tile's own pixels (one tile row per pass: always inside the image — no
//     flags, no zero-padding selects) and then the pixels of the halo border, the only ones that carry flags;
//   * the K loop is fully unrolled (every MFMA names its own weight registers; needs -mllvm
//     -pragma-unroll-threshold=1000000, see the Makefile); a scheduling barrier per k-step keeps the compiler's
//     interleave local (MFMAs + fragment reads + the k-step's share of the staging and epilogue units);
//   * the folded 1x1 skip (Conv_2 on the raw block input) is NSK extra chunks that use only the centre tap; a residual
//     rides through the same path as a skip with identity weights (a residual LOAD between the epilogue's stores would
//     wait for every earlier store: vmcnt retires in order);
//   * each phase runs in two halves of the wave's rows: the epilogue of one half (bias + temb bias, 1/sqrt(2),
//     statistics for the next GroupNorm, packing, v_permlane32_swap / v_permlane16_swap into 64-byte row pieces,
//     buffer_store_b128 — all in the accumulator layout, no LDS) runs under the MFMAs of the other half.
//
// What bounds it: the issue slots of the single in-order wave (<= 5 instructions fit under a
// 32-cycle MFMA, the GroupNorm variant has ~9).  K order (chunk, tap, 16-channel block) as in conv_mfma.hip / conv3x3_ws.hip.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef RW_TIMING  // profiling build only: per-phase cycle totals of wave 0
__device__ unsigned long long g_rw_dbg[16];
#define RT_DECL unsigned rt_prev = (unsigned)__builtin_readcyclecounter(), rt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_MARK(i) { unsigned rt_now = (unsigned)__builtin_readcyclecounter(); rt_acc[i] += rt_now - rt_prev; rt_prev = rt_now; }
#define RT_FLUSH if (threadIdx.x == 0) { for (int q = 0; q < 8; ++q) atomicAdd(&g_rw_dbg[q], (unsigned long long)rt_acc[q]); atomicAdd(&g_rw_dbg[15], 1ull); }
extern "C" int diffsep_rw_debug_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rw_dbg), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rw_dbg), z, sizeof(z)); }
  return 0;
}
#else
#define RT_DECL
#define RT_MARK(i)
#define RT_FLUSH
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
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
constexpr int KC = 64;                // channels per chunk: a pixel's chunk is ONE full 128-byte line of a 64-channel tensor
constexpr int NKB = KC / 16;          // 16-channel k-blocks per tap
constexpr int KSC = 9 * NKB;          // k-steps of a 3x3 chunk
constexpr int AROW = KC * 2 + 16;     // 144 B: LDS pitch of a halo pixel (16 consecutive rows = 16 distinct bank slots)
constexpr int PPL = KC / 8;           // 16-byte pieces (lanes) per pixel
constexpr int NT = 256;

struct RwK {
  const bf16_t* x; long x_bs; int ldx; int C1;     // channels [0, C1) from x, [C1, Cin) from x2
  const bf16_t* x2; long x2_bs; int ldx2;
  const bf16_t* w; int w_chunked;                  // [64][9][Cin] or chunk-major [Cin/32][9][64][32]
  const bf16_t* wfrag; const bf16_t* swfrag;       // optional fragment-major copies of w / sw (ds_rw_frag_index): [k-step][cout group][lane][8]
  const float* gn_scale; const float* gn_shift;    // [B][Cin] or null
  const long long* gn_acc1; const long long* gn_acc2; const float* gn_gamma; const float* gn_beta;
  int gn_groups; float gn_inv_count; float gn_eps;
  const float* bias; const float* bias_b; int bias_b_ld;
  const bf16_t* res; long res_bs; int ldr;
  float out_scale;
  bf16_t* y; long y_bs; int ldy;
  long long* stats;
  const bf16_t* sx; long sx_bs; int ldsx; int sC1;  // folded skip: raw channels [0, sC1) from sx, the rest from sx2
  const bf16_t* sx2; long sx2_bs; int ldsx2;
  const bf16_t* sw; int sw_chunked; int sw_shift; int sCin;
  int H, W, G, tiles_x, tiles_per_img;
  int dbg;  // profiling builds: bit 0 = stores fall outside the tensor, bit 1 = loads do
};

// k-steps whose weight fragments live in LDS instead of registers (the LAST ones in K order): the fourth chunk of the
// 128-channel layers and the 128-channel skip — what the 512-entry register file does not hold beside the accumulators
// With 4 cout groups (128 couts, 128 input channels) a wave holds 72 fragments of the 3x3 weights: 64 fill the 256
// accumulator registers, the last 8 and every skip fragment live in LDS
#ifndef RW_NWL2
#define RW_NWL2 8  // (64 fragments fill the 256 accumulator registers; tools/rw_ab.sh "-DRW_NWL2=18" "" for the A/B)
#endif
constexpr int rw_lds_ksteps(int nch, int rpw, int nsk, int ncg) { return ncg == 4 ? nch * KSC - 64 + nsk * NKB : (nch == 2 ? RW_NWL2 : 0); }

// NCG: cout groups of 32 (2: 64 couts, the block's 4 waves = 2 cout groups x 2 pixel groups; 4: 128 couts, 4 cout
// groups on ONE pixel group — the same staging and epilogue work per wave for twice the MFMAs)
template <int NCH, int RPW, int NSK, int NCG>
struct RwGeom {
  static constexpr int CO = 32 * NCG, PGN = 4 / NCG;
  static constexpr int TH = PGN * RPW, HH_ = TH + 2, HP = HH_ * HW_;
  // 16-byte staging pieces per thread and chunk: NI passes over the tile's own pixels (a pass = NT / PPL = 32 pixels = one
  // tile row: always inside the image, no flags, no selects), then NB passes over the NBP pixels of the halo border
  static constexpr int NI = TH * TW * PPL / NT, NBP = HP - TH * TW, NB = (NBP * PPL + NT - 1) / NT;
  static constexpr int NL = NI + NB;
  static_assert(NT / PPL == TW && TH * TW * PPL % NT == 0, "one staging pass = one tile row");
  static constexpr int LDS_A = NL * (NT / PPL) * AROW;    // one ring slot (whole passes of the block: no predicated writes)
  static constexpr int CIN = NCH * KC, SCIN = NSK * KC;
  static constexpr int LDS_TAB = (2 * CIN + CO) * 4;      // GN scale, GN shift, (bias + temb bias) * out_scale
  static constexpr int LDS_DESC = NB * NT * 4;            // relative pixel index of the border pieces
  static_assert((NI + NB) * (NT / PPL) >= HP + NB * (NT / PPL) - NBP, "dummy pixels of the last border pass fit the slot");
  static constexpr int NPH = NCH + NSK;                   // phases (chunks) per tile
  static constexpr int NKS = NCH * KSC + NSK * NKB;       // k-steps = weight fragments per wave
  static constexpr int NWL = rw_lds_ksteps(NCH, RPW, NSK, NCG), NWR = NKS - NWL;  // fragments in LDS / in registers
  static constexpr int LDS_WL = NWL * NCG * 64 * 16;      // [k-step][cout group][lane] x 16 B
  static constexpr int WL_STEP = NCG * 64 * 16;
  static constexpr int OFF_TAB = 2 * LDS_A, OFF_WL = OFF_TAB + ((LDS_TAB + 15) & ~15), OFF_DESC = OFF_WL + LDS_WL;
  static constexpr int LDS_TOTAL = OFF_DESC + LDS_DESC;
  // chunk of phase P: [0, NCH) = 3x3 chunk, NCH + s = skip chunk s.  The skip chunks come first: the tile's LAST phase is
  // a long one (it carries the epilogue of the first half-tile)
  static constexpr int chunk_of(int P) { return P < NSK ? NCH + P : P - NSK; }
  static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget of one CU");
  static_assert(NT * 36 * 4 <= LDS_TOTAL, "the statistics reduce reuses the block's LDS");
};

// NCH: 64-channel chunks of the 3x3 input (1: 64 channels, 2: 128 = one or two sources); RPW: pixel rows per wave
// (tile = 2 RPW x 32); NSK: 64-channel chunks of the folded 1x1 skip (0, 1, 2); MODE: 0 raw input, 2 GroupNorm + SiLU
template <int NCH, int RPW, int NSK, int MODE, int NCG>
__global__ __launch_bounds__(NT, 1) void conv3x3_rw_kernel(RwK p) {
  using G = RwGeom<NCH, RPW, NSK, NCG>;
  constexpr int CO = G::CO, PGN = G::PGN;
  constexpr int TH = G::TH, HP = G::HP, LDS_A = G::LDS_A, NL = G::NL, NPH = G::NPH, NKS = G::NKS, CIN = G::CIN, NWR = G::NWR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;
  float* sTab = reinterpret_cast<float*>(smem + G::OFF_TAB);
  int* sDesc = reinterpret_cast<int*>(smem + G::OFF_DESC);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int cg = wave % NCG, pg = wave / NCG;
  const int b = blockIdx.x / p.G, part = blockIdx.x % p.G;
  const int t0 = (int)((long)part * p.tiles_per_img / p.G);
  const int nt = (int)((long)(part + 1) * p.tiles_per_img / p.G) - t0;
  RT_DECL

  // ---- tables: GroupNorm scale / shift of image b, bias.  The operands are LOADED here, ahead of the first chunk's loads and
  // the weight fragments (loads return in order), and turned into the LDS tables after those have been issued: the fp64
  // statistics run while 70 - 300 KB of weights are in flight
  static_assert(CIN <= NT && CO <= NT, "one table entry per thread");
  constexpr int CPG_MAX = 8;  // channels per GroupNorm group: min(C / 4, 32) groups -> 4 (C <= 128) or 8 (C = 256)
  long long t_s[CPG_MAX], t_q[CPG_MAX];
  float t_gam = 1.f, t_bet = 0.f, t_sc = 1.f, t_sh = 0.f, t_bias = 0.f;
#pragma unroll
  for (int j = 0; j < CPG_MAX; ++j) { t_s[j] = 0; t_q[j] = 0; }
  if (tid < CIN) {
    const int c = tid;
    if (p.gn_acc1) {  // statistics straight from the producers' channel-sum accumulators
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
  if (tid < CO) t_bias = (p.bias ? p.bias[tid] : 0.f) + (p.bias_b ? p.bias_b[(long)b * p.bias_b_ld + tid] : 0.f);
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
  const int slot = tid & (PPL - 1);
  // staging pieces of this thread, 16-byte slot tid % PPL of a pixel: piece k < NI = pixel (row k, column tid / PPL) of the tile
  // itself; piece NI + kb = border pixel tid / PPL + 32 kb of the halo line (top row, bottom row, left column, right column).
  // Border pieces carry 5 flag bits (bits 0..3 = top / bottom / left / right halo line, bit 4 = past the last border
  // pixel), their relative pixel index (LDS table or registers) and their LDS offset (registers).
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
    dstb[kb] = (in ? hy * HW_ + hx : HP + (bi - NBP)) * AROW + slot * 16;  // (the pieces past the border: unused pixels of the slot)
    fl |= flg << (5 * kb);
  }
  const int ldi0 = (HW_ + 1 + ixp) * AROW + slot * 16;  // tile pixel (0, ixp): piece k < NI is HW_ pixels (one halo row) further

  RT_MARK(6)
  const int M = p.H * p.W;
  const __amdgpu_buffer_rsrc_t rx1 = rsrc(p.x + (long)b * p.x_bs, (unsigned)M * p.ldx * 2u);
  const __amdgpu_buffer_rsrc_t rx2 = p.x2 ? rsrc(p.x2 + (long)b * p.x2_bs, (unsigned)M * p.ldx2 * 2u) : rx1;
  const __amdgpu_buffer_rsrc_t rs1 = NSK ? rsrc(p.sx + (long)b * p.sx_bs, (unsigned)M * p.ldsx * 2u) : rx1;
  const __amdgpu_buffer_rsrc_t rs2 = (NSK && p.sx2) ? rsrc(p.sx2 + (long)b * p.sx2_bs, (unsigned)M * p.ldsx2 * 2u) : rs1;
  const __amdgpu_buffer_rsrc_t ry = rsrc(p.y + (long)b * p.y_bs, (unsigned)M * p.ldy * 2u);

  // ---- this wave's weight fragments: cout cg * 32 + l32, k-step (chunk, tap, kb): channels 32 chunk + 16 kb + 8 h ..
  u32x4_t wf[NWR > 0 ? NWR : 1];
  char* sWl = smem + G::OFF_WL + (cg * 64 + lane) * 16;  // + (ks - NWR) * WL_STEP
  auto put_w = [&](int ks, const u32x4_t& v) __attribute__((always_inline)) {
    if (ks < NWR) wf[ks] = v;
    else if (pg == 0) *reinterpret_cast<u32x4_t*>(sWl + (ks - NWR) * G::WL_STEP) = v;
  };
  auto load_weights = [&]() __attribute__((always_inline)) {
    // fragment-major copies (the engine's layers): a wave's load instruction = 1 KB of contiguous memory, 8 full 128-byte lines,
    // instead of 32-byte pieces of 32 rows (16 half-used lines that the k-block's other half touches again later)
    const bool wfm = p.wfrag != nullptr, sfm = NSK && p.swfrag != nullptr;  // (uniform)
    const __amdgpu_buffer_rsrc_t rw = rsrc(wfm ? p.wfrag : p.w, 9u * CO * CIN * 2u);
    const __amdgpu_buffer_rsrc_t rsw = (NSK && p.sw) ? rsrc(sfm ? p.swfrag : p.sw, (unsigned)CO * G::SCIN * 2u) : rsrc(p.w, 0u);
    const unsigned vfrag = (unsigned)((cg * 64 + lane) * 16);
    const int co = cg * 32 + l32;
    // 3x3: [64][9][Cin]: (co * 9 + tap) * Cin + ch; chunk-major [Cin/32][9][64][32]: ((ch / 32 * 9 + tap) * 64 + co) * 32 + ch % 32
    const bool wc = p.w_chunked != 0;  // (uniform: the per-fragment part of the offset is a scalar select)
    const unsigned vlane = wfm ? vfrag : (wc ? (unsigned)((co * 32 + h * 8) * 2) : (unsigned)((co * 9 * CIN + h * 8) * 2));
    // skip: [64][sCin]: co * sCin + ch; chunk-major [sCin/kc][64][kc] (kc >= 16: a 16-channel k-block never straddles a
    // layout chunk): ((ch / kc) * 64 + co) * kc + ch % kc
    const bool sc_ = NSK && p.sw_chunked != 0;
    const int kcs = sc_ ? p.sw_chunked : 16;
    const unsigned vl2 = sfm ? vfrag : (sc_ ? (unsigned)((co * kcs + h * 8) * 2) : (unsigned)((co * G::SCIN + h * 8) * 2));
    auto load_one = [&](int ks) __attribute__((always_inline)) {
      if (ks < NCH * KSC) {
        const int c = ks / KSC, tap = (ks % KSC) / NKB, kb = ks % NKB;
        const int chb = c * KC + kb * 16;  // first channel of the k-block (this lane: + 8 h)
        const unsigned so = wfm ? (unsigned)(ks * NCG * 1024)
                                : (wc ? (unsigned)((((chb >> 5) * 9 + tap) * CO * 32 + (chb & 31)) * 2) : (unsigned)((tap * CIN + chb) * 2));
        put_w(ks, ld16(rw, vlane, so));
      } else {
        const int chb = (ks - NCH * KSC) * 16;
        const unsigned so = sfm ? (unsigned)((ks - NCH * KSC) * NCG * 1024)
                                : (sc_ ? (unsigned)((((chb >> p.sw_shift) * CO) * kcs + (chb & (kcs - 1))) * 2) : (unsigned)(chb * 2));
        u32x4_t f = ld16(rsw, vl2, so);
        if (!p.sw) {  // residual as a skip with identity weights: channel chb + 8 h + j feeds cout co with weight 1
          const int d = co - (chb + 8 * h);  // the lane's 8 channels hold the 1 at position d (if 0 <= d < 8)
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = d == 2 * e ? DS_H_ONE : (d == 2 * e + 1 ? DS_H_ONE << 16 : 0u);
        }
        put_w(ks, f);
      }
    };
    // the LDS-resident fragments first: they have to be written before the block's first barrier
#pragma unroll
    for (int ks = NWR; ks < NKS; ++ks) load_one(ks);
#pragma unroll
    for (int ks = 0; ks < NWR; ++ks) load_one(ks);
  };

  // ---- staging state: pa[] holds the chunk AFTER the one in LDS (in flight or landed)
  u32x4_t pa[NL];
  // geometry of the tile a chunk belongs to (wave-uniform): first pixel, border mask (which halo lines lie outside
  // the image); tiles past the block's range get a pixel index beyond every tensor (the hardware returns zeros)
  struct TileG { int pix0; unsigned edge; };
  auto geom_at = [&](int ty, int tx, bool valid) {
    TileG g;
    const int y0 = ty * TH, x0 = tx * TW;
    g.edge = (y0 == 0 ? 1u : 0u) | (y0 + TH == p.H ? 2u : 0u) | (x0 == 0 ? 4u : 0u) | (x0 + TW == p.W ? 8u : 0u);
    g.pix0 = valid ? y0 * p.W + x0 : 0x3fffff;
    return g;
  };
  auto tile_geom = [&](int i) {
    const int t = t0 + i;
    const int ty = t / p.tiles_x;
    return geom_at(ty, t - ty * p.tiles_x, i < nt);
  };
  // piece k of a chunk of type P is valid (inside the image / an interior pixel for the skip chunks)
  auto piece_ok = [&](auto P_, const TileG& g, int k) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
    const unsigned em = (P < NCH ? (g.edge | 16u) : 31u) << (5 * (k - NI));  // (scalar; border pieces only)
    return (fl & em) == 0u;
  };
  // source of chunk P (compile time): descriptor, pixel pitch in bytes, byte offset of this thread's 8 channels
  auto issue_one = [&](auto P_, const TileG& g, int k, int rel) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value;
    constexpr bool CONV = P < NCH;
    constexpr int CB = (CONV ? P : P - NCH) * KC;
    const int c1 = CONV ? p.C1 : p.sC1;
    const bool second = CB >= c1;  // (wave-uniform)
    const unsigned ld2 = (unsigned)(CONV ? (second ? p.ldx2 : p.ldx) : (second ? p.ldsx2 : p.ldsx)) * 2u;
    const unsigned co2 = (unsigned)((second ? CB - c1 : CB) * 2) + (unsigned)slot * 16u;
    const __amdgpu_buffer_rsrc_t r = CONV ? (second ? rx2 : rx1) : (second ? rs2 : rs1);
    if constexpr (!CONV) {
      if (k >= NI) return;  // (a skip chunk meets the centre tap only: its border pieces are never read)
    }
    if (k < NI) {  // a pixel of the tile itself: inside the image whenever the tile is (a tile past the block's range has a
                   // pixel index beyond every tensor: the hardware returns zeros)
      const unsigned off = __umul24((unsigned)(g.pix0 + k * p.W + ixp), ld2) + co2;
      pa[k] = ld16(r, off, 0);
      return;
    }
    const unsigned off = __umul24((unsigned)(rel + g.pix0), ld2) + co2;
#ifdef RW_TIMING
    pa[k] = ld16(r, (piece_ok(P_, g, k) && !(p.dbg & 2)) ? off : OOB, 0);
#else
    pa[k] = ld16(r, piece_ok(P_, g, k) ? off : OOB, 0);
#endif
  };
  // every input of the launch is activated (no raw skip / residual chunk shares the accumulators): the activation may
  // leave a constant factor to the epilogue
  constexpr bool FOLD = MODE == 2 && NSK == 0;
  // Round 5.  (i) RW_ACT_PK (half-precision build): GroupNorm affine + SiLU in PACKED half precision — per dword (two channels)
  // v_pk_fma_f16, 2 x v_exp_f16, v_pk_add_f16, 2 x v_rcp_f16, v_pk_mul_f16 = 7 instructions (8 without FOLD) instead of 11
  // (2 fma_mix, 2 exp, 2 add, 2 rcp, 2 mul, cvt_pk); the upper halves go through SDWA forms of the transcendentals, no
  // unpack / pack.  What it costs in rounding is gated by tests/test_engine_gpu.py against the CPU oracle (-DRW_ACT_F32
  // restores the fp32 arithmetic for the A/B).  (ii) RW_PIPE: a unit runs in THREE stages, one unit apart (affine + exp |
  // 1 + e, rcp | multiply, write, re-issue): the single in-order wave no longer issues a transcendental's consumer straight
  // behind it (a k-step carries 0.6 units: inside one unit every instruction depends on the previous one).
#if defined(DS_HALF_F16) && !defined(RW_ACT_F32)
  constexpr bool ACT_PK = true;
#else
  constexpr bool ACT_PK = false;
#endif
#ifdef RW_NO_PIPE
  constexpr int PIPE_LAG = 0;
#else
  constexpr int PIPE_LAG = 2;
#endif
  float gsc[ACT_PK ? 1 : 8], gsh[ACT_PK ? 1 : 8];
  unsigned psc[ACT_PK ? 4 : 1], psh[ACT_PK ? 4 : 1];
  auto act_tab = [&](int c) __attribute__((always_inline)) {  // scale / shift of this thread's 8 channels of chunk c
    if constexpr (MODE != 0) {
      const float4* ts = reinterpret_cast<const float4*>(sTab + c * KC + slot * 8);
      const float4* th = reinterpret_cast<const float4*>(sTab + CIN + c * KC + slot * 8);
      const float4 s0 = ts[0], s1 = ts[1], h0 = th[0], h1 = th[1];
      float a[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, b_[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
      if constexpr (FOLD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] *= -1.4426950408889634f; b_[j] *= -1.4426950408889634f; }
      }
      if constexpr (ACT_PK) {
#pragma unroll
        for (int d = 0; d < 4; ++d) { psc[d] = pack_h2(a[2 * d], a[2 * d + 1]); psh[d] = pack_h2(b_[2 * d], b_[2 * d + 1]); }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { gsc[j] = a[j]; gsh[j] = b_[j]; }
      }
    }
  };
  // Staging runs in UNITS of one dword (two channels) so that its VALU work spreads evenly over the k-steps of a phase:
  // unit u = dword u & 3 of piece u >> 2.  The piece's last unit writes it to the ring and re-issues its registers as
  // the load of the chunk after next.  Stage registers of the units in flight: ring of 3 by unit index (compile time).
  u32x4_t so;  // the piece being assembled
  float uz[3][2], ut[3][2];
  unsigned uzp[3], utp[3], uxp[3], udp[3], urp[3];
  // stages 0 (unit u0: affine, exp2 of both channels) and 1 (unit u1: 1 + e, reciprocal) of two DIFFERENT units, emitted
  // interleaved: an SDWA transcendental that completes a register (upper half, the lower one preserved) straight behind the
  // instruction that wrote the lower half costs a wait state (s_nop) — the other unit's instruction sits between them.
  // u0 / u1 < 0: that stage has nothing to do in this step.
  auto unit_s01 = [&](auto P1_, int u0, int u1) __attribute__((always_inline)) {
    constexpr int P1 = decltype(P1_)::value;
    if constexpr (P1 < NCH && MODE != 0) {
      const int q0 = u0 >= 0 ? u0 % 3 : 0, q1 = u1 >= 0 ? u1 % 3 : 0;
      if constexpr (ACT_PK) {
        unsigned z = 0, x = 0, e = 0, dd = 0, r = 0;
        if (u0 >= 0) {
          asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(z) : "v"(pa[u0 >> 2][u0 & 3]), "v"(psc[u0 & 3]), "v"(psh[u0 & 3]));
          x = z;
          if constexpr (MODE == 2 && !FOLD) asm("v_pk_mul_f16 %0, %1, %2" : "=v"(x) : "v"(z), "s"(0xbdc5bdc5u));  // -log2(e) in both halves
        }
        if constexpr (MODE == 2) {
          if (u1 >= 0) asm("v_pk_add_f16 %0, %1, %2" : "=v"(dd) : "v"(utp[q1]), "s"(0x3c003c00u));
          if (u0 >= 0) asm("v_exp_f16 %0, %1" : "=v"(e) : "v"(x));
          if (u1 >= 0) asm("v_rcp_f16 %0, %1" : "=v"(r) : "v"(dd));
          if (u0 >= 0) asm("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(e) : "v"(x));
          if (u1 >= 0) asm("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r) : "v"(dd));
          if (u1 >= 0) urp[q1] = r;
          if (u0 >= 0) utp[q0] = e;
        }
        if (u0 >= 0) uzp[q0] = z;
      } else {
        if (u1 >= 0 && MODE == 2) {
          ut[q1][0] = __builtin_amdgcn_rcpf(1.0f + ut[q1][0]);
          ut[q1][1] = __builtin_amdgcn_rcpf(1.0f + ut[q1][1]);
        }
        if (u0 >= 0) {
          const unsigned w = pa[u0 >> 2][u0 & 3];
          const int d = u0 & 3;
          const float lo = h_lo(w), hi = h_hi(w);
          const float z0 = fmaf(lo, gsc[2 * d], gsh[2 * d]), z1 = fmaf(hi, gsc[2 * d + 1], gsh[2 * d + 1]);
          uz[q0][0] = z0; uz[q0][1] = z1;
          if (MODE == 2) {
            // FOLD: the affine carries the factor -log2(e), z IS the exponent of the sigmoid's exp2 and the staged value is
            // silu(GN(x)) / -ln 2; the epilogue multiplies the accumulators back (exact in fp32, one multiply per element less)
            ut[q0][0] = __builtin_amdgcn_exp2f(FOLD ? z0 : z0 * -1.4426950408889634f);
            ut[q0][1] = __builtin_amdgcn_exp2f(FOLD ? z1 : z1 * -1.4426950408889634f);
          }
        }
      }
    }
  };
  // stage 2: z * sigmoid, the dword into the piece (unit_s2); the piece's last unit writes it and re-issues its registers
  // (unit_fin: that part alone, for the hand-placed stream whose multiply sits in an earlier lump)
  // (RAW: the chunk's registers already hold the ACTIVATED values — the pre-activation of round 5, see half())
  auto unit_s2x = [&](auto P1_, auto P2_, auto FIN_, auto RAW_, const TileG& g1, const TileG& g2, int sl, int u, int rel) __attribute__((always_inline)) {
    constexpr int P1 = decltype(P1_)::value;
    constexpr bool FIN_ONLY = decltype(FIN_)::value, RAW = decltype(RAW_)::value;
    const int k = u >> 2, d = u & 3, q = u % 3;
    if constexpr (!FIN_ONLY) {
    if constexpr (P1 < NCH && MODE != 0 && !RAW) {
      if constexpr (ACT_PK) {
        unsigned v = uzp[q];
        if constexpr (MODE == 2) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(v) : "v"(uzp[q]), "v"(urp[q]));
        so[d] = v;
      } else {
        float z0 = uz[q][0], z1 = uz[q][1];
        if (MODE == 2) { z0 *= ut[q][0]; z1 *= ut[q][1]; }
        so[d] = pack_h2(z0, z1);
      }
    } else {
      so[d] = pa[k][d];
    }
    }
    if (d == 3) {
      if (P1 < NCH && MODE != 0 && k >= NI) {  // zero padding stays zero (silu(GN(0)) != 0): border pieces only
        const bool ok = piece_ok(P1_, g1, k);
        so.x = ok ? so.x : 0u;
        so.y = ok ? so.y : 0u;
        so.z = ok ? so.z : 0u;
        so.w = ok ? so.w : 0u;
      }
#ifdef RW_ABL_NOLDSW
      asm volatile("" :: "v"(so));
#else
      if (P1 < NCH || k < NI)  // (border pieces of a skip chunk: nothing was loaded, nothing is read)
        *reinterpret_cast<u32x4_t*>(sA + sl * LDS_A + (k < NI ? ldi0 + k * HW_ * AROW : dstb[k < NI ? 0 : k - NI])) = so;
#endif
#ifdef RW_ABL_NOLOAD
      pa[k][0] += rel;
#else
      issue_one(P2_, g2, k, rel);
#endif
    }
  };
  auto unit_s2 = [&](auto P1_, auto P2_, const TileG& g1, const TileG& g2, int sl, int u, int rel) __attribute__((always_inline)) {
    unit_s2x(P1_, P2_, std::false_type{}, std::false_type{}, g1, g2, sl, u, rel);
  };
  auto unit_fin = [&](auto P1_, auto P2_, const TileG& g1, const TileG& g2, int sl, int u, int rel) __attribute__((always_inline)) {
    unit_s2x(P1_, P2_, std::true_type{}, std::false_type{}, g1, g2, sl, u, rel);
  };
  auto unit_raw = [&](auto P1_, auto P2_, const TileG& g1, const TileG& g2, int sl, int u, int rel) __attribute__((always_inline)) {
    unit_s2x(P1_, P2_, std::false_type{}, std::true_type{}, g1, g2, sl, u, rel);
  };
  // a whole unit at once (the prologue)
  auto unit = [&](auto P1_, auto P2_, const TileG& g1, const TileG& g2, int sl, int u, int rel) __attribute__((always_inline)) {
    unit_s01(P1_, u, -1);
    unit_s01(P1_, -1, u);
    unit_s2(P1_, P2_, g1, g2, sl, u, rel);
  };

  f32x16 acc[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
  float ssum[16], ssq[16];  // per lane: its 16 couts (8 q + 4 h + i), summed over its pixels
#pragma unroll
  for (int j = 0; j < 16; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  const bool has_stats = p.stats != nullptr;
  const float osc = FOLD ? p.out_scale * -0.6931471805599453f : p.out_scale;
  // fragment base of this lane: pixel (row pg * RPW, column l32) of the halo tile, k-half h
  const int fbase = (pg * RPW * HW_ + l32) * AROW + h * 16;

  // ---- epilogue of one tile, in the accumulator layout (no LDS): lane (pixel l32, half h) holds, per row, the cout
  // quads 8 q + 4 h .. + 3 of the wave's 32 couts.  A 16-byte load / store of a lane covers couts 16 j + 8 h .. + 7 of its
  // pixel; two v_permlane32_swap per 16 bytes turn that into the two quads (q = 2 j, 2 j + 1) of the lane and back.
  auto swap_halves = [&](u32x4_t& v) __attribute__((always_inline)) {
    auto r0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
    v.x = r0[0]; v.z = r0[1]; v.y = r1[0]; v.w = r1[1];
  };
  // ---- epilogue work in UNITS of half a row (8 couts of the lane's pixel): bias, statistics, packing, one 16-byte
  // store.  (There are no loads in the epilogue: a residual rides through the ring as two raw chunks that meet identity
  // fragments — see the launcher.  vmcnt retires in order and counts stores: residual rows loaded between the stores
  // were measured waiting for the acknowledgement of every earlier store, 1350 cycles per row on an idle chip.)
  // The two units of a row leave together: v_permlane16_swap regroups their 16-byte pieces so that one store covers
  // 16 pixels x 64 contiguous bytes (the wave's 32 couts) instead of 32 pixels x 32 bytes — half the write requests.
  u32x4_t ov0;  // the row's first half, waiting for the second
  const int srow = lane >> 4;  // after the regrouping lane L holds, of pixel L & 15 (+ 16 in the second store), the piece:
  const unsigned spiece = (unsigned)(cg * 64 + (srow & 1) * 32 + (srow >> 1) * 16);  // byte offset in the pixel's 128 B
  auto epi_unit = [&](const TileG& g, int r, int j, const float4& t0, const float4& t1) __attribute__((always_inline)) {
    float v[8];
    v[0] = fmaf(acc[r][8 * j + 0], osc, t0.x); v[1] = fmaf(acc[r][8 * j + 1], osc, t0.y);
    v[2] = fmaf(acc[r][8 * j + 2], osc, t0.z); v[3] = fmaf(acc[r][8 * j + 3], osc, t0.w);
    v[4] = fmaf(acc[r][8 * j + 4], osc, t1.x); v[5] = fmaf(acc[r][8 * j + 5], osc, t1.y);
    v[6] = fmaf(acc[r][8 * j + 6], osc, t1.z); v[7] = fmaf(acc[r][8 * j + 7], osc, t1.w);
    // (always taken: a branch here would cut the half-phase's instruction stream into separately scheduled pieces)
#ifndef RW_ABL_NOSTATS
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ssum[8 * j + e] += v[e];
      ssq[8 * j + e] = fmaf(v[e], v[e], ssq[8 * j + e]);
    }
#endif
    u32x4_t ov = {pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
    swap_halves(ov);  // lane (pixel l32, half h): couts 16 j + 8 h .. + 7
    if (j == 0) {
      ov0 = ov;
    } else {
      u32x4_t a = ov0, b2 = ov;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        auto q = __builtin_amdgcn_permlane16_swap(a[d], b2[d], false, false);
        a[d] = q[0]; b2[d] = q[1];
      }
      const int pix = g.pix0 + (pg * RPW + r) * p.W + (lane & 15);
      const unsigned o = __umul24((unsigned)pix, (unsigned)p.ldy * 2u) + spiece;
#ifdef RW_TIMING
      const unsigned o1 = (p.dbg & 1) ? OOB : o, o2 = (p.dbg & 1) ? OOB : o + 16u * (unsigned)p.ldy * 2u;
#else
      const unsigned o1 = o, o2 = o + 16u * (unsigned)p.ldy * 2u;
#endif
#ifdef RW_ABL_NOSTORE
      asm volatile("" :: "v"(a), "v"(b2), "v"(o1), "v"(o2));
#else
      __builtin_amdgcn_raw_buffer_store_b128(a, ry, o1, 0, 0);    // pixels 0 .. 15 of the row
      __builtin_amdgcn_raw_buffer_store_b128(b2, ry, o2, 0, 0);   // pixels 16 .. 31
#endif
    }
  };
  // (bias + temb bias) * out_scale of the unit's two cout quads: LDS broadcast reads, issued ahead of the k-step's MFMAs
  // Where the register file has room (64-channel layers: 144 - 176 of the 256 accumulator registers hold weights) the
  // lane's 16 values stay in registers for the whole launch: 8 LDS reads per half-phase less
  // (64-channel layers and the 4-row tiles of the 128-cout variants have the VGPRs for both tables; the 128 -> 64 variant
  // for the bias only)
  constexpr bool REL_REGS = NWR * 4 <= 176 || NCG == 4;
  constexpr bool BIAS_REGS = true;
  float4 breg[2][2];
  int relreg[G::NB];  // ... and so do the relative pixel indices of the thread's border pieces
  auto epi_bias = [&](int j, float4& t0, float4& t1) __attribute__((always_inline)) {
    if constexpr (BIAS_REGS) {
      t0 = breg[j][0];
      t1 = breg[j][1];
    } else {
      t0 = *reinterpret_cast<const float4*>(sTab + 2 * CIN + cg * 32 + 8 * (2 * j) + 4 * h);
      t1 = *reinterpret_cast<const float4*>(sTab + 2 * CIN + cg * 32 + 8 * (2 * j + 1) + 4 * h);
    }
  };

  // ---- one HALF of a phase: the MFMAs of chunk (phase P, ring slot P & 1) for the wave's rows [HF * RH, HF * RH + RH),
  // interleaved with (i) its share of the staging of the next phase's chunk and of the loads of the chunk after that,
  // (ii) EPI: the epilogue of the OTHER half's rows — rows [RH, RPW) of the previous tile during the first half of phase
  // 0, rows [0, RH) of this tile during the second half of the last phase.  The epilogue of one half of the accumulators
  // thus always runs under the MFMAs of the other half: no second accumulator set, no phase in which all waves of the
  // chip store at once.
  //
  // Round 5: the stream is HAND-PLACED.  A single in-order wave hides at most ~5 other instructions under one 32-cycle
  // MFMA, and only if they sit in the gap behind it; the compiler's placement (sched_group_barrier sees neither the
  // inline-asm MFMAs nor inline-asm VALU) put two MFMAs back to back and the k-step's VALU behind them.  Now every
  // MFMA is followed by its GAP: the fragment read scheduled there, then the LUMPS (<= 4 - 5 instructions each, volatile
  // inline asm in the order written) that a compile-time schedule assigns to it:
  //   * unit lumps: virtual staging step v = {stage 0 of unit v, stage 1 of unit v - 1 | stage 2 of unit v - 2} in two
  //     lumps; spread evenly over the phase's gaps, a gap of an epilogue half counting W_E / W_N of another one;
  //   * epilogue lumps (EPI halves): per row 20 lumps — per half row (j) 4 x {2 fma + pack | the pair's statistics}, the
  //     permlane32 swap; then the permlane16 regrouping and the two stores — spread evenly over the half's gaps.
  constexpr int RH = RPW / 2;
#ifndef RW_W_E
#define RW_W_E 2
#define RW_W_N 5
#endif
  constexpr int W_E = RW_W_E, W_N = RW_W_N;  // capacity of a gap for unit lumps: epilogue half / other half (A/B: -DRW_W_E=.. -DRW_W_N=..)
  float et[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // the two cout pairs in flight through the epilogue lumps
  u32x4_t osa = {0, 0, 0, 0}, osb = {0, 0, 0, 0};  // the row's two store pieces after the regrouping
  // PRE-ACTIVATION (round 5, the 64 -> 64 layers with ONE folded skip / residual chunk, fp16 build).  A tile of those layers is a
  // SHORT phase (the skip chunk: 16 MFMAs per wave) and a long one (the 3x3 chunk: 144); the short phase staged the next 3x3 chunk,
  // i.e. carried its whole activation under 16 MFMAs, VALU-bound.  The registers of that chunk are re-issued as loads DURING the
  // long phase (as the raw skip pieces leave them): now the raw units are placed in the first third of the long phase and, from there
  // on, the landed 3x3 pieces are activated IN PLACE (pa[k][d] <- silu(GN(pa[k][d])), same lumps, the multiply's destination is the
  // register itself): the short phase only copies, zeroes the padding and writes.  The first tile of a block has no long phase in
  // front of it and activates as before (PD_ = false).
  // MEASURED, NOT SHIPPED (-DRW_PREACT builds it): same box, 64 -> 64 + residual @256^2 136.9 / 139.8 us with, 136.5 / 133.6 without —
  // the short phase was not waiting for its activation VALU (its first half also carries the previous tile's 44 epilogue lumps).
  constexpr bool PRE_CFG = NCH == 1 && NSK == 1 && MODE == 2 && NCG == 2 && ACT_PK && PIPE_LAG == 2
#ifndef RW_PREACT
                           && false
#endif
      ;
  auto half = [&](auto P_, auto HF_, auto EPI_, auto PD_, int slot_r, const TileG& ge, const TileG& g1, const TileG& g2) __attribute__((always_inline)) {
    constexpr int P = decltype(P_)::value, HF = decltype(HF_)::value;
    constexpr bool EPI = decltype(EPI_)::value;
    constexpr bool PREDONE = decltype(PD_)::value;      // the chunk this phase stages was activated in registers by the phase before
    constexpr bool PRE = PRE_CFG && P == NPH - 1;       // this phase activates, in place, the chunk whose loads it issues
    constexpr int C = G::chunk_of(P);
    constexpr bool CONV = C < NCH;
    constexpr int NK = CONV ? KSC : NKB;
    constexpr int W0 = CONV ? C * KSC : NCH * KSC + (C - NCH) * NKB;
    constexpr int C1 = G::chunk_of((P + 1) % NPH), C2 = G::chunk_of((P + 2) % NPH);
    constexpr int NU = NL * 4;
    constexpr bool ACT1 = C1 < NCH && MODE != 0 && !PREDONE;
    // virtual unit index v: stage 0 of unit v, stage 1 of unit v - 1, stage 2 of unit v - LAG (LAG = 0: the whole unit at v)
    constexpr int LAG = ACT1 ? PIPE_LAG : 0, NUV = NU + LAG, NLU = 2 * NUV;
    constexpr int NGH = NK * RH;  // gaps (MFMAs) of this half
    constexpr int w0 = (P == 0) ? W_E : W_N, w1 = (P == NPH - 1) ? W_E : W_N, CAP = NGH * (w0 + w1);
    // first unit lump of phase gap GP in [0, 2 NGH]
    // (PRE: the raw units of this phase in its first third, the in-place activation lumps in the rest)
    constexpr int GPRE = 2 * NGH / 3, NLP = PRE ? 2 * (NU + 2) : 0;
    auto lub = [](int GP) constexpr {
      return PRE ? NLU * (GP < GPRE ? GP : GPRE) / GPRE : NLU * (GP <= NGH ? GP * w0 : NGH * w0 + (GP - NGH) * w1) / CAP;
    };
    auto lpb = [](int GP) constexpr { return GP <= GPRE ? 0 : NLP * (GP - GPRE) / (2 * NGH - GPRE); };
    constexpr int NLE = RH * 22;  // epilogue lumps of an EPI half
    constexpr int R0 = HF * RH, ER0 = HF ? 0 : RH;
    const char* fb = sA + slot_r * LDS_A + fbase + R0 * HW_ * AROW;
#ifndef RW_BUILTIN_MFMA
    if constexpr (EPI) {  // the rows this half finishes were last written by asm MFMAs: 12 wait states before a VALU read
#pragma unroll
      for (int r = 0; r < RH; ++r) asm volatile("s_nop 11" : "+v"(acc[ER0 + r]));
    }
#endif
    if constexpr (HF == 0 && C1 < NCH) act_tab(C1);
    // K order inside a 3x3 chunk: (kx, 16-channel block) groups outside, ky inside.  The RH rows of this half use the
    // pixel fragments of input rows R0 .. R0 + RH + 1 at the group's (kx, block): each is read ONCE per group and serves up
    // to three k-steps (ky) — RH + 2 fragment reads per 3 RH MFMAs instead of 3 RH, and one wait per group.  (A skip
    // chunk has one k-step per group: the centre tap.)
    constexpr int SUB = CONV ? 3 : 1, RFN = CONV ? RH + 2 : RH, NG = NK / SUB;
    static_assert(SUB * RH >= RFN, "the next group's fragments are read in the MFMA slots of the current one");
    auto ldg = [&](int g, int j) __attribute__((always_inline)) {  // fragment j of group g = kx * NKB + block
      const int dx = CONV ? g / NKB : 1, kb = g % NKB, row = CONV ? j : j + 1;
      return *reinterpret_cast<const u32x4_t*>(fb + (row * HW_ + dx) * AROW + kb * 32);
    };
    u32x4_t rf[2][RFN];
#pragma unroll
    for (int j = 0; j < RFN; ++j) rf[0][j] = ldg(0, j);
    // k-step ks of this order -> its weight fragment in the (tap, block) order of load_weights
    auto widx = [](int ks) {
      if (!CONV) return ks;
      const int g = ks / 3, dy = ks % 3, dx = g / NKB, kb = g % NKB;
      return (dy * 3 + dx) * NKB + kb;
    };
#define RW_FRAG(ks, r) rf[((ks) / SUB) & 1][(r) + (CONV ? (ks) % SUB : 0)]
#define RW_FRAG_LAST(ks) rf[((ks) / SUB) & 1][RFN - 1]
    // weight fragment of k-step ks: a register, or (the last NWL fragments) an LDS read issued one k-step ahead
    u32x4_t wl = {0, 0, 0, 0}, wln = {0, 0, 0, 0};
    if constexpr (W0 + widx(0) >= NWR) wl = *reinterpret_cast<const u32x4_t*>(sWl + (W0 + widx(0) - NWR) * G::WL_STEP);
    // relative pixel index of the border pieces that complete in a k-step, read (registers or LDS table) one k-step ahead
    int rels[2][NL];
    auto fetch_rels = [&](int ks, int par) __attribute__((always_inline)) {  // for the unit lumps of k-step ks of this half
      const int gp0 = HF * NGH + ks * RH;
#pragma unroll
      for (int L = lub(gp0); L < lub(gp0 + RH); ++L) {
        const int u = (L >> 1) - LAG;
        if ((L & 1) && u >= 0 && u < NU && (u & 3) == 3 && (u >> 2) >= NI)
          rels[par][u >> 2] = REL_REGS ? relreg[(u >> 2) - NI] : sDesc[((u >> 2) - NI) * NT + tid];
      }
    };
    fetch_rels(0, 0);  // (the first k-step's operands: the one exposed round trip of the half)

    // ---- lumps.  (An inline-asm instruction that reads a register written by one of the two instructions in front of it
    // gets a wait state from the compiler, which cannot see whether the producer was a transcendental: the orders below
    // keep every consumer three instructions behind its producer.)
    auto unit_lump = [&](int L, int par) __attribute__((always_inline)) {
      const int v = L >> 1;
      const int u0 = v < NU ? v : -1, u1 = LAG ? ((v >= 1 && v - 1 < NU) ? v - 1 : -1) : v, u2 = v - LAG;
      const bool s2 = u2 >= 0 && u2 < NU;
      if ((L & 1) == 0) {
        if constexpr (ACT1 && ACT_PK) {
          // affine of unit u0 | 1 + e of unit u1 | z * sigmoid of unit u2 | exp2 (lower half) of u0 | reciprocal (lower half) of u1
          if (u0 >= 0) {
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(uzp[u0 % 3]) : "v"(pa[u0 >> 2][u0 & 3]), "v"(psc[u0 & 3]), "v"(psh[u0 & 3]));
            if constexpr (MODE == 2 && !FOLD) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(uxp[u0 % 3]) : "v"(uzp[u0 % 3]), "s"(0xbdc5bdc5u));  // x -log2(e)
          }
          if constexpr (MODE == 2) {
            if (u1 >= 0) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(udp[u1 % 3]) : "v"(utp[u1 % 3]), "s"(0x3c003c00u));
            if (s2) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(so[u2 & 3]) : "v"(uzp[u2 % 3]), "v"(urp[u2 % 3]));
            if (u0 >= 0) asm volatile("v_exp_f16 %0, %1" : "=v"(utp[u0 % 3]) : "v"(FOLD ? uzp[u0 % 3] : uxp[u0 % 3]));
            if (u1 >= 0) asm volatile("v_rcp_f16 %0, %1" : "=v"(urp[u1 % 3]) : "v"(udp[u1 % 3]));
          } else {
            if (s2) so[u2 & 3] = uzp[u2 % 3];
          }
        } else if constexpr (ACT1) {
          if (u0 >= 0) unit_s01(std::integral_constant<int, C1>{}, u0, -1);
        }
      } else {
        if constexpr (ACT1 && ACT_PK) {
          if constexpr (MODE == 2) {
            if (u0 >= 0) asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(utp[u0 % 3]) : "v"(FOLD ? uzp[u0 % 3] : uxp[u0 % 3]));
            if (u1 >= 0) asm volatile("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(urp[u1 % 3]) : "v"(udp[u1 % 3]));
          }
        } else if constexpr (ACT1) {
          if (u1 >= 0) unit_s01(std::integral_constant<int, C1>{}, -1, u1);
        }
#ifndef RW_ABL_NOSTAGE
        if (s2) {
          const int rel = ((u2 & 3) == 3 && (u2 >> 2) >= NI) ? rels[par][u2 >> 2] : 0;
          if constexpr (ACT1 && ACT_PK) unit_fin(std::integral_constant<int, C1>{}, std::integral_constant<int, C2>{}, g1, g2, slot_r ^ 1, u2, rel);
          else if constexpr (PREDONE) unit_raw(std::integral_constant<int, C1>{}, std::integral_constant<int, C2>{}, g1, g2, slot_r ^ 1, u2, rel);
          else unit_s2(std::integral_constant<int, C1>{}, std::integral_constant<int, C2>{}, g1, g2, slot_r ^ 1, u2, rel);
        }
#endif
      }
    };
    // in-place activation lumps (PRE): virtual step v = {affine + exp of unit v, 1 + e and reciprocal of unit v - 1, multiply of unit
    // v - 2 INTO ITS OWN REGISTER}; the chunk is C2 — its scale / shift table is the one in psc / psh (NCH = 1: one table)
    auto pre_lump = [&](int L) __attribute__((always_inline)) {
      if constexpr (PRE) {
        const int v = L >> 1;
        const int u0 = v < NU ? v : -1, u1 = (v >= 1 && v - 1 < NU) ? v - 1 : -1, u2 = v - 2;
        const bool s2 = u2 >= 0 && u2 < NU;
        if ((L & 1) == 0) {
          if (u0 >= 0) {
            asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(uzp[u0 % 3]) : "v"(pa[u0 >> 2][u0 & 3]), "v"(psc[u0 & 3]), "v"(psh[u0 & 3]));
            if constexpr (!FOLD) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(uxp[u0 % 3]) : "v"(uzp[u0 % 3]), "s"(0xbdc5bdc5u));
          }
          if (u1 >= 0) asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(udp[u1 % 3]) : "v"(utp[u1 % 3]), "s"(0x3c003c00u));
          if (s2) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(pa[u2 >> 2][u2 & 3]) : "v"(uzp[u2 % 3]), "v"(urp[u2 % 3]));
          if (u0 >= 0) asm volatile("v_exp_f16 %0, %1" : "=v"(utp[u0 % 3]) : "v"(FOLD ? uzp[u0 % 3] : uxp[u0 % 3]));
          if (u1 >= 0) asm volatile("v_rcp_f16 %0, %1" : "=v"(urp[u1 % 3]) : "v"(udp[u1 % 3]));
        } else {
          if (u0 >= 0) asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(utp[u0 % 3]) : "v"(FOLD ? uzp[u0 % 3] : uxp[u0 % 3]));
          if (u1 >= 0) asm volatile("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(urp[u1 % 3]) : "v"(udp[u1 % 3]));
        }
      }
    };
    // epilogue lumps of row rr, 22 per row: per half row j the pairs p = 0 .. 3 of the lane's 8 couts as a two-deep
    // pipeline — A(p) = {v = acc * scale + bias of pair p; pack pair p - 1}, B(p) = {statistics of pair p} in the order
    // A0 A1 B0 A2 B1 A3 B2 A4 B3 — then the permlane32 swap; x = 20: permlane16 regrouping; x = 21: the stores
    auto epi_lump = [&](int LE) __attribute__((always_inline)) {
      const int rr = ER0 + LE / 22, x = LE % 22;
      if (x < 20) {
        const int j = x / 10, y = x % 10;
        // y: 0 A0, 1 A1, 2 B0, 3 A2, 4 B1, 5 A3, 6 B2, 7 A4, 8 B3, 9 swap
        const bool isA = y == 0 || y == 1 || y == 3 || y == 5 || y == 7;
        const int pr = y == 0 ? 0 : (y == 1 ? 1 : (y == 2 ? 0 : (y == 3 ? 2 : (y == 4 ? 1 : (y == 5 ? 3 : (y == 6 ? 2 : (y == 7 ? 4 : 3)))))));
        if (y == 9) {
          // lane (pixel l32, half h): couts 16 j + 8 h .. + 7
          if (j == 0) {
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(osa[0]), "+v"(osa[2]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(osa[1]), "+v"(osa[3]));
          } else {
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(osb[0]), "+v"(osb[2]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(osb[1]), "+v"(osb[3]));
          }
        } else if (isA) {
          if (pr < 4) {
            const int e0 = 8 * j + 2 * pr;
            const float b0 = pr == 0 ? breg[j][0].x : (pr == 1 ? breg[j][0].z : (pr == 2 ? breg[j][1].x : breg[j][1].z));
            const float b1 = pr == 0 ? breg[j][0].y : (pr == 1 ? breg[j][0].w : (pr == 2 ? breg[j][1].y : breg[j][1].w));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(et[pr & 1][0]) : "v"(acc[rr][e0]), "s"(osc), "v"(b0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(et[pr & 1][1]) : "v"(acc[rr][e0 + 1]), "s"(osc), "v"(b1));
          }
          if (pr >= 1) {
            const int q = pr - 1;
            if (j == 0) asm volatile(DS_CVT_PK_H_ASM " %0, %1, %2" : "=v"(osa[q]) : "v"(et[q & 1][0]), "v"(et[q & 1][1]));
            else asm volatile(DS_CVT_PK_H_ASM " %0, %1, %2" : "=v"(osb[q]) : "v"(et[q & 1][0]), "v"(et[q & 1][1]));
          }
        } else {
#ifndef RW_ABL_NOSTATS
          const int e0 = 8 * j + 2 * pr;
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(ssum[e0]) : "v"(et[pr & 1][0]));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(ssum[e0 + 1]) : "v"(et[pr & 1][1]));
          asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(ssq[e0]) : "v"(et[pr & 1][0]));
          asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(ssq[e0 + 1]) : "v"(et[pr & 1][1]));
#endif
        }
      } else if (x == 20) {
#pragma unroll
        for (int d = 0; d < 4; ++d) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(osa[d]), "+v"(osb[d]));
      } else {
        const int pix = ge.pix0 + (pg * RPW + rr) * p.W + (lane & 15);
        const unsigned o = __umul24((unsigned)pix, (unsigned)p.ldy * 2u) + spiece;
#ifdef RW_TIMING
        const unsigned o1 = (p.dbg & 1) ? OOB : o, o2 = (p.dbg & 1) ? OOB : o + 16u * (unsigned)p.ldy * 2u;
#else
        const unsigned o1 = o, o2 = o + 16u * (unsigned)p.ldy * 2u;
#endif
#ifdef RW_ABL_NOSTORE
        asm volatile("" :: "v"(osa), "v"(osb), "v"(o1), "v"(o2));
#else
        __builtin_amdgcn_raw_buffer_store_b128(osa, ry, o1, 0, 0);   // pixels 0 .. 15 of the row
        __builtin_amdgcn_raw_buffer_store_b128(osb, ry, o2, 0, 0);   // pixels 16 .. 31
#endif
      }
    };

#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      if (ks + 1 < NK) fetch_rels(ks + 1, (ks + 1) & 1);
      const int wi = W0 + widx(ks), win = W0 + widx(ks + 1 < NK ? ks + 1 : ks);
      if (win >= NWR && ks + 1 < NK) wln = *reinterpret_cast<const u32x4_t*>(sWl + (win - NWR) * G::WL_STEP);
      const u32x4_t wk = wi < NWR ? wf[wi < NWR ? wi : 0] : wl;
      // RW_DEP: the k-step's LAST pixel fragment rides along as an unused operand of every MFMA of the k-step: the
      // compiler then waits ONCE per k-step (for the newest fragment) instead of once per MFMA
#ifdef RW_NO_DEP
#define RW_DEP
#else
#define RW_DEP , "v"(RW_FRAG_LAST(ks))
#endif
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        // Inline asm: the register-resident weight fragment is pinned to the accumulator half of the register file ("a")
        // and feeds the MFMA from there; accumulators, pixel fragments and everything the VALU touches live in the
        // architectural half.  What the compiler does not know about an asm MFMA: the 12 wait states between its result and
        // a VALU read — the guard at the start of every half.
        if (wi < NWR) {
          if (P == 0 && ks == 0) asm volatile(DS_MFMA_H32_ASM " %0, %1, %2, 0" : "=v"(acc[R0 + r]) : "a"(wk), "v"(RW_FRAG(ks, r)) RW_DEP);
          else asm volatile(DS_MFMA_H32_ASM " %0, %1, %2, %0" : "+v"(acc[R0 + r]) : "a"(wk), "v"(RW_FRAG(ks, r)) RW_DEP);
        } else {  // LDS-resident fragment: straight from the ds_read's VGPRs (no VALU copy in front of the MFMA)
          if (P == 0 && ks == 0) asm volatile(DS_MFMA_H32_ASM " %0, %1, %2, 0" : "=v"(acc[R0 + r]) : "v"(wk), "v"(RW_FRAG(ks, r)) RW_DEP);
          else asm volatile(DS_MFMA_H32_ASM " %0, %1, %2, %0" : "+v"(acc[R0 + r]) : "v"(wk), "v"(RW_FRAG(ks, r)) RW_DEP);
        }
        // ---- the gap behind this MFMA
        {  // the next group's fragments, one per MFMA slot of this group
          const int g = ks / SUB, q = (ks % SUB) * RH + r;
          if (q < RFN && g + 1 < NG) rf[(g + 1) & 1][q] = ldg(g + 1, q);
        }
        const int gh = ks * RH + r, gp = HF * NGH + gh;
#pragma unroll
        for (int L = lub(gp); L < lub(gp + 1); ++L) unit_lump(L, ks & 1);
        if constexpr (PRE) {
#pragma unroll
          for (int L = lpb(gp); L < lpb(gp + 1); ++L) pre_lump(L);
        }
#ifndef RW_ABL_NOEPI
        if constexpr (EPI) {
#pragma unroll
          for (int LE = gh * NLE / NGH; LE < (gh + 1) * NLE / NGH; ++LE) epi_lump(LE);
        }
#endif
      }
      wl = wln;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: the first phase's chunk into slot 0, the second phase's chunk in flight
  TileG gc = tile_geom(0);
  constexpr int CH0 = G::chunk_of(0), CH1 = G::chunk_of(1 % NPH);
  const TileG g1st = NPH > 1 ? gc : tile_geom(1);  // the tile of the second phase
  {
    // the first chunk's loads go out ahead of the weights (loads return in order: the tile can be staged while the
    // weight fragments are still streaming in; the MFMAs then wait for them fragment by fragment)
#pragma unroll
    for (int k = 0; k < NL; ++k) issue_one(std::integral_constant<int, CH0>{}, gc, k, k < NI ? 0 : sDesc[(k < NI ? 0 : k - NI) * NT + tid]);
    load_weights();
    build_tables();
    sync_lds();  // tables (and the LDS-resident weight fragments) visible
    RT_MARK(7)
    if constexpr (REL_REGS) {
#pragma unroll
      for (int k = 0; k < NB; ++k) relreg[k] = sDesc[k * NT + tid];
    }
    if constexpr (BIAS_REGS) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        breg[j][0] = *reinterpret_cast<const float4*>(sTab + 2 * CIN + cg * 32 + 8 * (2 * j) + 4 * h);
        breg[j][1] = *reinterpret_cast<const float4*>(sTab + 2 * CIN + cg * 32 + 8 * (2 * j + 1) + 4 * h);
      }
    }
    act_tab(CH0);
    // (staging the first chunk re-issues every piece as the second one)
#pragma unroll
    for (int u = 0; u < NL * 4; ++u)
      unit(std::integral_constant<int, CH0>{}, std::integral_constant<int, CH1>{}, gc, g1st, 0, u,
           (u >> 2) < NI ? 0 : sDesc[((u >> 2) < NI ? 0 : (u >> 2) - NI) * NT + tid]);
  }
  RT_MARK(0)
  TileG gp = tile_geom(nt);  // "previous tile" of the first one: no tile (its stores fall outside every tensor)
  int ph = 0;                 // phases done: the chunk of phase ph sits in ring slot ph & 1
  // tile i + 2 by increments (a division per tile and geometry is ~45 scalar instructions in a stream whose issue slots
  // are the bottleneck)
  TileG gnc = tile_geom(1);
  int ty2 = (t0 + 2) / p.tiles_x, tx2 = (t0 + 2) - ty2 * p.tiles_x;
  for (int i = 0; i < nt; ++i) {
    const TileG gn = gnc, gnn = geom_at(ty2, tx2, i + 2 < nt);
    // phase P stages the chunk of phase P + 1 and issues that of phase P + 2: both belong to the next tile once they wrap
    auto run = [&](auto self, auto P_) __attribute__((always_inline)) {
      constexpr int P = decltype(P_)::value;
      sync_lds();
#ifdef RW_SKEW
      if (wave == 1) __builtin_amdgcn_s_sleep(RW_SKEW);
      if (wave == 2) __builtin_amdgcn_s_sleep(2 * RW_SKEW);
      if (wave == 3) __builtin_amdgcn_s_sleep(3 * RW_SKEW);
#endif
      RT_MARK(1)
      const int slot_r = ph & 1;
      if constexpr (P == 0 && PRE_CFG) {  // (the first tile's 3x3 chunk was not pre-activated)
        if (i == 0) half(P_, std::integral_constant<int, 0>{}, std::true_type{}, std::false_type{}, slot_r, gp, gc, gn);
        else half(P_, std::integral_constant<int, 0>{}, std::true_type{}, std::true_type{}, slot_r, gp, gc, gn);
      } else {
        half(P_, std::integral_constant<int, 0>{}, std::integral_constant<bool, P == 0>{}, std::false_type{}, slot_r, gp, ((P + 1) / NPH == 0 ? gc : gn), ((P + 2) / NPH == 0 ? gc : ((P + 2) / NPH == 1 ? gn : gnn)));
      }
      if constexpr (P == 0) {
        if (i == 0) {  // (the first tile has no predecessor: what that epilogue summed up was not an output)
#pragma unroll
          for (int j = 0; j < 16; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
        }
      }
      if constexpr (P == 0 && PRE_CFG) {
        if (i == 0) half(P_, std::integral_constant<int, 1>{}, std::false_type{}, std::false_type{}, slot_r, gc, gc, gn);
        else half(P_, std::integral_constant<int, 1>{}, std::false_type{}, std::true_type{}, slot_r, gc, gc, gn);
      } else {
        half(P_, std::integral_constant<int, 1>{}, std::integral_constant<bool, P == NPH - 1>{}, std::false_type{}, slot_r, gc, ((P + 1) / NPH == 0 ? gc : gn), ((P + 2) / NPH == 0 ? gc : ((P + 2) / NPH == 1 ? gn : gnn)));
      }
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
#ifndef RW_BUILTIN_MFMA
#pragma unroll
  for (int r = 0; r < RH; ++r) asm volatile("s_nop 11" : "+v"(acc[RH + r]));
#endif
#pragma unroll
  for (int e = 0; e < RH * 2; ++e) {
    float4 t0, t1;
    epi_bias(e & 1, t0, t1);
    epi_unit(gp, RH + (e >> 1), e & 1, t0, t1);
  }
  RT_MARK(4)
  if (has_stats) {
    __syncthreads();
    // per lane 16 couts (cg * 32 + 8 q + 4 h + i) of pixel column l32 of its rows: sum over the 32 columns and the two
    // pixel groups
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
      ds_stat_add(p.stats + ((long)b * CO + co) * 2 + st, (long long)llrint(a * (st ? DS_STAT_SQ_SCALE : DS_STAT_SUM_SCALE)));
    }
  }
  RT_MARK(5)
  RT_FLUSH
}

int rw_num_cus() { return ds_num_cus(); }

int rw_blocks_per_image(const ConvArgs& a, int tiles) {
  const int cus = rw_num_cus();
  int g = cus / a.B;
#ifdef RW_TIMING  // (profiling builds only: fewer, fatter blocks)
  if (getenv("DIFFSEP_RW_G")) g = atoi(getenv("DIFFSEP_RW_G"));
#endif
  // option rw_half (A/B, round 5): launches whose blocks would get <= 4 tiles (the 128-row level at B = 16: a 295 KB weight
  // prologue per 4 tiles) run on HALF the CUs with twice the tiles per block — pays only if another stream's kernel takes the rest
  if ((a.opts & DS_OPT_RW_HALF) && g >= 2 && tiles / g <= 4) g /= 2;
  if ((a.opts & DS_OPT_RW_QUARTER) && g >= 4 && tiles / g <= 4) g /= 4;
  if ((a.opts & DS_OPT_RW_BIG_HALF) && g >= 2 && tiles / g > 4) g /= 2;
  if (g < 1) g = 1;
  if (g > tiles) g = tiles;
  return g;
}

template <int NCH, int RPW, int NSK, int MODE, int NCG>
int rw_launch(const RwK& k0, const ConvArgs& a, hipStream_t st) {
  using G = RwGeom<NCH, RPW, NSK, NCG>;
  RwK k = k0;
  const int tiles = (a.H / G::TH) * (a.W / TW);
  k.G = rw_blocks_per_image(a, tiles);
  k.tiles_x = a.W / TW;
  k.tiles_per_img = tiles;
  auto kern = conv3x3_rw_kernel<NCH, RPW, NSK, MODE, NCG>;
  DS_FUNC_LDS_ONCE(kern, G::LDS_TOTAL);
  hipLaunchKernelGGL(kern, dim3(a.B * k.G), dim3(NT), G::LDS_TOTAL, st, k);
  DS_LAUNCH_CHECK();
  {
    static char name[64] = {0};
    if (!name[0]) snprintf(name, sizeof(name), "conv3x3_rw_kernel<%d,%d,%d,%d,%d>", NCH, RPW, NSK, MODE, NCG);
    ds_set_last_conv_kernel(name);
  }
  return 0;
}

}  // namespace

// The layers this kernel takes over: bf16 3x3, 64 couts, 64 or 128 input channels (one tensor or the in-place concat of
// two), input raw or GroupNorm + SiLU, optional folded 1x1 skip on 64 / 128 raw channels, whole tiles.
bool ds_conv_rw_eligible(const ConvArgs& a) {
  if (a.opts & DS_OPT_NO_RW) return false;
  const int CO = a.Cout;
  if (!(a.dtype == DS_BF16 && a.taps == 9 && ((CO == 64 && (a.Cin == 64 || a.Cin == 128)) || (CO == 128 && a.Cin == 128)) &&
        a.w_bs == 0 && (a.w_chunked == 0 || a.w_chunked == 32) && a.bias_mode == 0 && !a.div_b && a.W % TW == 0 && a.H % 8 == 0 &&
        a.H >= 32 && a.W >= 32 && a.ldy >= CO && a.ldy % 8 == 0 && (!a.res || (a.ldr >= CO && a.ldr % 8 == 0))))
    return false;
  if (CO == 128 && (a.opts & DS_OPT_NO_RW128)) return false;
  if (a.x2 ? !(a.C1 % KC == 0 && a.C1 > 0 && a.C1 < a.Cin && a.ldx % 8 == 0 && a.ldx2 % 8 == 0) : a.ldx % 8 != 0) return false;
  const bool gn = a.gn_scale || a.gn_acc1;
  if (gn && !a.gn_act) return false;  // (affine without SiLU does not occur in front of a 3x3 convolution)
  if (a.gn_acc1 && !(a.gn_groups > 0 && a.Cin % a.gn_groups == 0 && a.Cin / a.gn_groups <= 8 && (!a.x2 || a.gn_acc2))) return false;
  if (a.sx) {
    // (192 skip channels = the cat(64, 128) block of the 128^2 up path: 36 + 12 fragments fill 192 of the 256 weight registers)
    if (!(a.sw && !a.res && (a.sCin == 64 || a.sCin == 128 || (a.sCin == 192 && CO == 64 && a.Cin == 64)) && a.ldsx % 8 == 0 &&
          (!a.sx2 || (a.sC1 % KC == 0 && a.sC1 > 0 && a.sC1 < a.sCin && a.ldsx2 % 8 == 0)) &&
          (a.sw_chunked == 0 || ((a.sw_chunked & (a.sw_chunked - 1)) == 0 && a.sw_chunked >= 16))))
      return false;
    if (a.Cin == 128 && CO == 64) return false;  // (no layer of the network has both; the register file would not hold it either)
  }
  if (a.res && a.Cin == 128 && CO == 64) return false;  // (the residual rides as a skip: same limit)
  if (CO == 128) {  // 4 cout groups: the skip / residual fragments live in LDS; every such layer normalises its input
    if ((a.sx || a.res) && !gn) return false;
    // fewer 4 x 32 tiles than CUs (the 32^2 level at B = 16): a block's 295 KB weight prologue serves one tile and half
    // the chip idles — the generic tile is as fast there (22.6 vs 23.6 us) and leaves room for the other streams' blocks
    if ((long)a.B * (a.H / 4) * (a.W / TW) < rw_num_cus() && !(a.opts & DS_OPT_RW_SMALL)) return false;
    return true;
  }
  // a residual rides as an identity-weight skip chunk: 138 us at 256^2 against 159 us on the weight-stationary kernel
  // (+1 % end to end; option no_rw_res for the A/B)
  if (a.res && (a.opts & DS_OPT_NO_RW_RES)) return false;
  // fewer 8 x 32 tiles than CUs: a block's weight prologue would serve a single tile and part of the chip idles — those
  // launches stay on the weight-stationary / generic kernels.  (Round 3 first kept every 64-channel launch at <= 128^2
  // there: 41.3 vs 37 us; with the tile geometry by increments the register-weight kernel is the faster one at 128^2 too —
  // Conv_0 41.8 vs 46.5 us, + residual 45.0 vs 48.2, + skip64 46.1 vs 52.0, raw 35.1 vs 42.3; +1 % end to end.)
  if ((long)a.B * (a.H / 8) * (a.W / TW) < rw_num_cus() && !(a.opts & DS_OPT_RW_SMALL)) return false;
  return true;
}

int ds_launch_conv_rw(const ConvArgs& a, hipStream_t st) {
  RwK k;
  k.x = reinterpret_cast<const bf16_t*>(a.x); k.x_bs = a.x_bs; k.ldx = a.ldx; k.C1 = a.x2 ? a.C1 : a.Cin;
  k.x2 = reinterpret_cast<const bf16_t*>(a.x2); k.x2_bs = a.x2_bs; k.ldx2 = a.x2 ? a.ldx2 : a.ldx;
  k.w = reinterpret_cast<const bf16_t*>(a.w); k.w_chunked = a.w_chunked;
  k.wfrag = reinterpret_cast<const bf16_t*>(a.w_frag);
  k.swfrag = (a.sx && a.sw) ? reinterpret_cast<const bf16_t*>(a.sw_frag) : nullptr;
  k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift;
  k.gn_acc1 = a.gn_acc1; k.gn_acc2 = a.gn_acc2; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta;
  k.gn_groups = a.gn_groups; k.gn_inv_count = a.gn_inv_count; k.gn_eps = a.gn_eps;
  k.bias = a.bias; k.bias_b = a.bias_b; k.bias_b_ld = a.bias_b_ld;
  k.res = reinterpret_cast<const bf16_t*>(a.res); k.res_bs = a.res_bs; k.ldr = a.ldr;
  k.out_scale = a.out_scale;
  k.y = reinterpret_cast<bf16_t*>(a.y); k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.stats = a.stats_acc;
  k.sx = reinterpret_cast<const bf16_t*>(a.sx); k.sx_bs = a.sx_bs; k.ldsx = a.ldsx; k.sC1 = a.sx2 ? a.sC1 : a.sCin;
  k.sx2 = reinterpret_cast<const bf16_t*>(a.sx2); k.sx2_bs = a.sx2_bs; k.ldsx2 = a.sx2 ? a.ldsx2 : a.ldsx;
  k.sw = reinterpret_cast<const bf16_t*>(a.sw); k.sw_chunked = a.sw_chunked; k.sw_shift = a.sw_chunked ? __builtin_ctz(a.sw_chunked) : 0;
  k.sCin = a.sCin;
  k.H = a.H; k.W = a.W; k.G = 0; k.tiles_x = 0; k.tiles_per_img = 0;
#ifdef RW_TIMING  // (profiling builds only: stores / loads outside the tensors)
  k.dbg = getenv("DIFFSEP_RW_DBG") ? atoi(getenv("DIFFSEP_RW_DBG")) : 0;
#else
  k.dbg = 0;
#endif
  if (a.res) {  // the residual [B][H][W][64] as a folded skip with identity weights (sw = null): exact in the fp32 accumulators
    k.sx = reinterpret_cast<const bf16_t*>(a.res); k.sx_bs = a.res_bs; k.ldsx = a.ldr; k.sC1 = a.Cout;
    k.sx2 = nullptr; k.sx2_bs = 0; k.ldsx2 = a.ldr; k.sw = nullptr; k.sw_chunked = 0; k.sw_shift = 0; k.sCin = a.Cout;
  }
  const int mode = ((a.gn_scale || a.gn_acc1) && a.gn_act) ? 2 : 0;
  const int nsk = a.sx ? a.sCin / KC : (a.res ? a.Cout / KC : 0);
  if (a.Cout == 128) {  // 128 -> 128: 4 cout groups
    if (mode == 0) return rw_launch<2, 4, 0, 0, 4>(k, a, st);
    if (nsk == 0) return rw_launch<2, 4, 0, 2, 4>(k, a, st);
    if (nsk == 1) return rw_launch<2, 4, 1, 2, 4>(k, a, st);
    return rw_launch<2, 4, 2, 2, 4>(k, a, st);
  }
#define RW_GO(NCH_, NSK_) return mode == 2 ? rw_launch<NCH_, 4, NSK_, 2, 2>(k, a, st) : rw_launch<NCH_, 4, NSK_, 0, 2>(k, a, st)
  if (a.Cin == 128) RW_GO(2, 0);
  if (nsk == 0) RW_GO(1, 0);
  if (nsk == 1) RW_GO(1, 1);
  if (nsk == 2) RW_GO(1, 2);
  RW_GO(1, 3);
#undef RW_GO
}
