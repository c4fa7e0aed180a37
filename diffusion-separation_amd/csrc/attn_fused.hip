// attn_fused.hip — AttnBlockpp (layerspp.py:76-92) as ONE kernel for the 16-bit engines, C = 128 channels, L = H * W <= 256
// pixels (the 16 x 16 attention level and the 4 x 4 bottleneck of nf = 64 at 4 s): gfx950.
//
//   h = GroupNorm(x); q, k, v = NIN_0..2(h); w = softmax(q k^T / sqrt(C)); out = (x + NIN_3(w v)) / sqrt(2)
//
// Unfused, the block is 11 graph nodes (GroupNorm finalize + apply, three projections, V^T, scores, softmax, P V, output
// projection) of 5 - 9 us each on tensors of a few hundred KB: a 45 - 65 us latency chain per block, four blocks per network
// evaluation.  Here one workgroup of 8 waves owns one sample and nothing leaves the CU between the input and the output:
//   * h (GroupNorm affine applied, 16-bit) and V^T (16-bit) live in LDS (69.6 + 67.6 KB); every other intermediate lives in
//     registers: the 32 x 32 MFMA leaves a lane with register quads of 4 consecutive "weight-side" indices for its pixel row,
//     two v_permlane32_swap per 16 bytes turn that into the B fragment of the next product (conv3x3_rw.hip's epilogue trick),
//     so Q', P and O go from accumulator to operand without touching LDS;
//   * Q and K are never formed: S[i][j] = q_i . k_j = h_j . (Wk^T Wq h_i + Wk^T b_q) + (a term that does not depend on j and
//     that the softmax removes), so the engine folds M = Wk^T Wq and b' = Wk^T b_q once at creation (fp32, then the storage
//     type) and the kernel computes Q' = M h + b' with one product on the wave's own rows, then S = Q' h^T;
//   * V^T = Wv h^T + b_v is computed once per sample by all 8 waves (its rows are the A operand of P V: K-major in j);
//   * wave w owns query rows [32 w, 32 w + 32): Q' (32 MFMAs), S (L / 4 = 64), softmax in registers (row statistics: one
//     cross-half exchange), O = P V (64), out = O Wo^T (32), then bias, residual, 1 / sqrt(2), 8-byte stores and the output's
//     GroupNorm statistics (the consumer's normalisation reads accumulators, conv_mfma.hip);
//   * weights are read as MFMA fragments straight from global memory in the fragment-major order of ds_rw_frag_index (1 KB
//     contiguous per wave instruction; the 8 waves read the same fragments: L1 hits) a whole matrix ahead of their use: Wv and
//     M at kernel start, Wo while P V frees the score registers; LDS fragments are read one k-block ahead of their MFMAs and
//     consecutive MFMAs go to different accumulators (the first version, with every operand fetched where it is used, took
//     40 us per block: tools/attn_timing.sh).
// Rounding points: h, V, P and the output in the storage type like the unfused path; M in the storage type instead of Q and K.
// Both paths are compared with the CPU oracle at the 16-bit tolerance (tests/test_round4_gpu.py: the fused one is the closer).
#include <type_traits>

#include "common.h"

#ifdef ATTN_TIMING  // profiling build only (tools/attn_timing.sh): per-phase cycle totals of wave 0 of every block
__device__ unsigned long long g_attn_dbg[16];
#define AT_DECL unsigned long long at_prev = __builtin_readcyclecounter(), at_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define AT_MARK(i) { unsigned long long at_now = __builtin_readcyclecounter(); at_acc[i] += at_now - at_prev; at_prev = at_now; }
#define AT_FLUSH if (threadIdx.x == 0) { for (int q = 0; q < 12; ++q) atomicAdd(&g_attn_dbg[q], at_acc[q]); atomicAdd(&g_attn_dbg[15], 1ull); }
extern "C" int diffsep_attn_debug_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_dbg), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), z, sizeof(z)); }
  return 0;
}
#else
#define AT_DECL
#define AT_MARK(i)
#define AT_FLUSH
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int C = 128, NKB = C / 16, NT_ = 512, LMAX = 256;
constexpr int PH = C * 2 + 16;         // sH pitch: pixel row of 128 channels (+16 B: 16 consecutive rows = 16 distinct bank slots)
constexpr int PV = LMAX * 2 + 16;      // sVt pitch: channel row of 256 pixels
constexpr int OFF_VT = LMAX * PH, OFF_TAB = OFF_VT + C * PV, LDS_BYTES = OFF_TAB + 5 * C * 4;  // table: scale, shift, b', b_v, b_o
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget of one CU");
static_assert(LMAX * (C + 4) * 4 <= OFF_TAB, "the statistics pass reuses the h / V^T area for the fp32 output tile");

struct AttnK {
  const bf16_t* x; long x_bs; int ldx;          // [B][L][ldx]
  const long long* gn_acc; const float* gn_gamma; const float* gn_beta; int gn_groups; float gn_inv_count; float gn_eps;
  const float* gn_scale; const float* gn_shift;  // [B][C] (used when gn_acc is null)
  const bf16_t* wqk; const bf16_t* wv; const bf16_t* wo;  // fragment-major [kb][n tile][lane][8]; wqk = Wk^T Wq
  const float* bqk; const float* bv; const float* bo;       // bqk = Wk^T b_q
  bf16_t* y; long y_bs; int ldy;
  long long* stats;                              // [B][C][2] or null
  int L;                                         // pixels (multiple of 16, <= 256)
  float qk_scale;                                // C^-0.5 * log2(e)
};

__device__ inline void swap_halves(u32x4_t& v) {
  auto r0 = __builtin_amdgcn_permlane32_swap(v.x, v.z, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(v.y, v.w, false, false);
  v.x = r0[0]; v.z = r0[1]; v.y = r1[0]; v.w = r1[1];
}
// accumulator tile (lane = pixel row, register 4 g + e = weight-side index 8 g + 4 h + e) -> the two B fragments (k-blocks
// 2 t and 2 t + 1 of the next product: lane = the same pixel row, 8 consecutive indices 16 s + 8 h ..) in the storage type
__device__ inline void acc_to_frags(const f32x16& a, u32x4_t& f0, u32x4_t& f1) {
  f0 = u32x4_t{pack_h2(a[0], a[1]), pack_h2(a[2], a[3]), pack_h2(a[4], a[5]), pack_h2(a[6], a[7])};
  f1 = u32x4_t{pack_h2(a[8], a[9]), pack_h2(a[10], a[11]), pack_h2(a[12], a[13]), pack_h2(a[14], a[15])};
  swap_halves(f0);
  swap_halves(f1);
}
__device__ inline f32x16 mma(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
  return mfma_h32(__builtin_bit_cast(uint4, a), __builtin_bit_cast(uint4, b), c);
}
__device__ inline f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}

// LT: 32-pixel tiles of the sample (compile time: every product loop is straight-line code; L <= 32 LT, rows past L masked)
template <int LT>
__global__ __launch_bounds__(NT_) void attn_fused_kernel(AttnK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;
  char* sVt = smem + OFF_VT;
  float* sTab = reinterpret_cast<float*>(smem + OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int b = blockIdx.x, L = p.L;
  const bf16_t* xb = p.x + (long)b * p.x_bs;
  AT_DECL

  // Weight fragments: global loads issued a whole matrix ahead of their use (file header)
  const unsigned wlane = (unsigned)lane * 16u;  // fragment (kb, n tile) at ((kb * 4 + nt) * 64 + lane) * 16 B
  auto wfrag = [&](const bf16_t* w, int kb, int nt) {
    return *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(w) + (kb * 4 + nt) * 1024 + wlane);
  };
  // the raw input: 16 pieces of 16 bytes per pixel, issued first (the GroupNorm table is built while they are in flight)
  constexpr int PPP = C / 8;                         // pieces per pixel
  constexpr int NPC = LMAX * PPP / NT_;              // pieces per thread (8)
  u32x4_t raw[NPC];
#pragma unroll
  for (int k = 0; k < NPC; ++k) {
    const int pc = tid + k * NT_, px = pc / PPP, sl = pc % PPP;
    raw[k] = px < L ? *reinterpret_cast<const u32x4_t*>(xb + (long)px * p.ldx + sl * 8) : u32x4_t{0, 0, 0, 0};
  }
  // (issued behind the input tile: loads return in order)
  const int ct = wave & 3;                       // V^T: 32-channel tile of this wave
  u32x4_t wv[NKB], wa[4][NKB];                   // wa: the weight matrix of the wave's next 128 x 128 product
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) wv[kb] = wfrag(p.wv, kb, ct);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) wa[nt][kb] = wfrag(p.wqk, kb, nt);
  // ---- GroupNorm scale / shift of sample b (layerspp.py:78: no activation) and the bias vectors -> LDS table
  if (tid < C) {
    float sc, sh;
    if (p.gn_acc) {
      const int cpg = C / p.gn_groups, g0 = (tid / cpg) * cpg;
      long long vs[8], vq[8];  // (C / min(C / 4, 32) = 4 channels per group; all loads in flight at once)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long long* src = p.gn_acc + ((long)b * C + g0 + (j < cpg ? j : 0)) * 2;
        vs[j] = j < cpg ? src[0] : 0;
        vq[j] = j < cpg ? src[1] : 0;
      }
      long long ssum = 0, ssq = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ssum += vs[j]; ssq += vq[j]; }
      const double mean = (double)ssum * (1.0 / DS_STAT_SUM_SCALE) * (double)p.gn_inv_count;
      double var = (double)ssq * (1.0 / DS_STAT_SQ_SCALE) * (double)p.gn_inv_count - mean * mean;
      if (var < 0.0) var = 0.0;
      sc = (float)(1.0 / sqrt(var + (double)p.gn_eps)) * (p.gn_gamma ? p.gn_gamma[tid] : 1.f);
      sh = (p.gn_beta ? p.gn_beta[tid] : 0.f) - (float)mean * sc;
    } else {
      sc = p.gn_scale[(long)b * C + tid];
      sh = p.gn_shift[(long)b * C + tid];
    }
    sTab[tid] = sc;
    sTab[C + tid] = sh;
  } else if (tid < 4 * C) {
    const int w = tid / C - 1, c = tid % C;  // waves 2 - 7: the three bias vectors
    sTab[(2 + w) * C + c] = (w == 0 ? p.bqk : (w == 1 ? p.bv : p.bo))[c];
  }
  AT_MARK(0)
  __syncthreads();
  AT_MARK(1)
  {
    const int sl = tid % PPP;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = sTab[sl * 8 + j]; sh[j] = sTab[C + sl * 8 + j]; }
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int px = (tid + k * NT_) / PPP;
      u32x4_t o;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        o[d] = pack_h2(fmaf(h_lo(raw[k][d]), sc[2 * d], sh[2 * d]), fmaf(h_hi(raw[k][d]), sc[2 * d + 1], sh[2 * d + 1]));
      if (px >= L) o = u32x4_t{0, 0, 0, 0};          // rows past the image read as zeros (their scores are masked below)
      if (px < LT * 32) *reinterpret_cast<u32x4_t*>(sH + px * PH + sl * 16) = o;
    }
  }
  AT_MARK(2)
  __syncthreads();
  AT_MARK(3)

  auto hfrag = [&](int tile, int kb) {  // h rows 32 tile + l32, channels 16 kb + 8 h ..
    return *reinterpret_cast<const u32x4_t*>(sH + (tile * 32 + l32) * PH + kb * 32 + h * 16);
  };
  const float* sBqk = sTab + 2 * C;
  const float* sBv = sTab + 3 * C;
  const float* sBo = sTab + 4 * C;

  // ---- V^T[c][j] = sum_c' Wv[c][c'] h[j][c'] + b_v[c]: lane = channel c (B operand = Wv rows), registers = pixels j (A = h rows)
  {  // (the two waves of a channel tile split the pixel tiles; h fragments through a ring of 8, read 6 MFMAs ahead)
    const float bias = sBv[ct * 32 + l32];
    const int j0 = wave >> 2;
    constexpr int NIT = (LT + 1) / 2, NTV = NIT * NKB, LA = 6;   // (tile, k-block) steps of a wave, lookahead
    if (LT % 2 == 0 || j0 == 0) {
      u32x4_t ring[8];
#pragma unroll
      for (int t = 0; t < LA && t < NTV; ++t) ring[t % 8] = hfrag(j0 + 2 * (t / NKB), t % NKB);
      f32x16 acc = zero16();
#pragma unroll
      for (int t = 0; t < NTV; ++t) {
        const int jt = j0 + 2 * (t / NKB), kb = t % NKB;
        if (t + LA < NTV) ring[(t + LA) % 8] = hfrag(j0 + 2 * ((t + LA) / NKB), (t + LA) % NKB);
        acc = mma(ring[t % 8], wv[kb], acc);
        if (kb == NKB - 1) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint2 v = make_uint2(pack_h2(acc[4 * g] + bias, acc[4 * g + 1] + bias), pack_h2(acc[4 * g + 2] + bias, acc[4 * g + 3] + bias));
            *reinterpret_cast<uint2*>(sVt + (ct * 32 + l32) * PV + (jt * 32 + 8 * g + 4 * h) * 2) = v;
          }
          acc = zero16();
        }
      }
    }
  }
  AT_MARK(4)
  __syncthreads();
  AT_MARK(5)

  // ---- this wave's query rows
  const int i0 = wave * 32;
  f32x16 oacc[4];
  if (i0 < LT * 32) {
    u32x4_t q2[NKB];
    {  // Q' = h M^T + b' (M = Wk^T Wq)
      u32x4_t hf[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) hf[kb] = hfrag(wave, kb);
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {  // two output tiles at a time (two independent accumulator chains; four would not fit)
        f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[q] = mma(wa[2 * hp + q][kb], hf[kb], acc[q]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int nt = 2 * hp + q;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 bq = *reinterpret_cast<const float4*>(sBqk + nt * 32 + 8 * g + 4 * h);
            acc[q][4 * g] += bq.x; acc[q][4 * g + 1] += bq.y; acc[q][4 * g + 2] += bq.z; acc[q][4 * g + 3] += bq.w;
          }
          acc_to_frags(acc[q], q2[2 * nt], q2[2 * nt + 1]);
        }
      }
    }
    AT_MARK(6)
    // S[i][j] = Q'[i] . h[j]: lane = row i, registers = pixels j (LT tiles of 32); k-blocks outside (consecutive MFMAs on
    // different accumulators), h fragments through a ring of 8 read 6 MFMAs ahead
    f32x16 s[LT];
    {
      constexpr int NTS = NKB * LT, LA = 6;
      u32x4_t ring[8];
#pragma unroll
      for (int jt = 0; jt < LT; ++jt) s[jt] = zero16();
#pragma unroll
      for (int t = 0; t < LA && t < NTS; ++t) ring[t % 8] = hfrag(t % LT, t / LT);
#pragma unroll
      for (int t = 0; t < NTS; ++t) {
        if (t + LA < NTS) ring[(t + LA) % 8] = hfrag((t + LA) % LT, (t + LA) / LT);
        s[t % LT] = mma(ring[t % 8], q2[t / LT], s[t % LT]);
      }
    }
    AT_MARK(7)
    // softmax over j: the lane holds half of its row (the other k-half lane the rest); pixels past L are masked
    float mx = -3.0e38f, sum = 0.f;
    auto soft = [&](auto FULL_) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(FULL_)::value;  // every pixel of the LT tiles is inside the sample: no masks
#pragma unroll
      for (int jt = 0; jt < LT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = jt * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
          mx = (FULL || j < L) ? fmaxf(mx, s[jt][e]) : mx;
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float nm = -mx * p.qk_scale;
#pragma unroll
      for (int jt = 0; jt < LT; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int j = jt * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
          const float pe = (FULL || j < L) ? __builtin_amdgcn_exp2f(fmaf(s[jt][e], p.qk_scale, nm)) : 0.f;
          s[jt][e] = pe;
          sum += pe;
        }
    };
    if (L == LT * 32) soft(std::true_type{}); else soft(std::false_type{});
    sum += __shfl_xor(sum, 32, 64);
    const float inv_sum = __builtin_amdgcn_rcpf(sum);
    AT_MARK(8)
    // O = P V (P normalised: the storage-type rounding of the probabilities matches the unfused path's softmax output); the
    // output projection's fragments are issued as the score tiles die
    const char* vr = sVt + l32 * PV + h * 16;
    auto vfrag = [&](int nt, int kb) { return *reinterpret_cast<const u32x4_t*>(vr + nt * 32 * PV + kb * 32); };
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = zero16();
#pragma unroll
    for (int jt = 0; jt < LT; ++jt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[jt][e] *= inv_sum;
      u32x4_t pf0, pf1;
      acc_to_frags(s[jt], pf0, pf1);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) oacc[nt] = mma(vfrag(nt, 2 * jt), pf0, oacc[nt]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) oacc[nt] = mma(vfrag(nt, 2 * jt + 1), pf1, oacc[nt]);
      constexpr int WPT = 32 / LT;  // fragments of Wo per score tile (k-block major, the order of their use)
#pragma unroll
      for (int q = 0; q < WPT; ++q) wa[(jt * WPT + q) % 4][(jt * WPT + q) / 4] = wfrag(p.wo, (jt * WPT + q) / 4, (jt * WPT + q) % 4);
    }
    AT_MARK(9)
    u32x4_t of[NKB];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc_to_frags(oacc[nt], of[2 * nt], of[2 * nt + 1]);
    // O Wo^T (bias, residual and scaling follow in the store pass)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = zero16();
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) oacc[nt] = mma(wa[nt][kb], of[kb], oacc[nt]);
  }
  AT_MARK(10)
  // ---- store pass.  The projected tile goes through LDS (fp32, the h / V^T area: free once every wave is past its P V) so
  // that the residual is read and the output written in full 128-byte lines — in the accumulator layout a wave instruction
  // touches 32 rows, 16 bytes each (measured: the slowest phase of the first version) — and the statistics of the output for
  // the consumer's GroupNorm (channel sums of the fp32 values before the storage rounding, as the convolution epilogues
  // produce them) come out of the same pass.
  constexpr int RPT = LT * 32 * PPP / NT_;  // rows per thread in the store pass (its 8-channel slot is fixed)
  u32x4_t xres[RPT];                        // the residual rows: issued here, they land under the barrier and the LDS transposition
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const int r = tid / PPP + k * (NT_ / PPP);
    xres[k] = r < L ? *reinterpret_cast<const u32x4_t*>(xb + (long)r * p.ldx + (tid % PPP) * 8) : u32x4_t{0, 0, 0, 0};
  }
  __syncthreads();
  float* sO = reinterpret_cast<float*>(smem);  // [rows][C + 4]
  constexpr int OP = C + 4;
  if (i0 < LT * 32) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(sO + (i0 + l32) * OP + nt * 32 + 8 * g + 4 * h) =
            make_float4(oacc[nt][4 * g], oacc[nt][4 * g + 1], oacc[nt][4 * g + 2], oacc[nt][4 * g + 3]);
  }
  __syncthreads();
  {
    const int sl = tid % PPP, r0 = tid / PPP;
    float bo[8], ssum[8], ssq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { bo[j] = sBo[sl * 8 + j]; ssum[j] = 0.f; ssq[j] = 0.f; }
    bf16_t* yb = p.y + (long)b * p.y_bs;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int r = r0 + k * (NT_ / PPP);
      const float4 a0 = *reinterpret_cast<const float4*>(sO + r * OP + sl * 8), a1 = *reinterpret_cast<const float4*>(sO + r * OP + sl * 8 + 4);
      const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      float v[8];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        v[2 * d] = (a[2 * d] + bo[2 * d] + h_lo(xres[k][d])) * 0.70710678118654752440f;
        v[2 * d + 1] = (a[2 * d + 1] + bo[2 * d + 1] + h_hi(xres[k][d])) * 0.70710678118654752440f;
      }
      if (r < L) {
        *reinterpret_cast<u32x4_t*>(yb + (long)r * p.ldy + sl * 8) = u32x4_t{pack_h2(v[0], v[1]), pack_h2(v[2], v[3]), pack_h2(v[4], v[5]), pack_h2(v[6], v[7])};
#pragma unroll
        for (int j = 0; j < 8; ++j) { ssum[j] += v[j]; ssq[j] = fmaf(v[j], v[j], ssq[j]); }
      }
    }
    if (!p.stats) { AT_FLUSH return; }
    __syncthreads();  // (sO is read: reuse it for the partial sums [32 row groups][C][2])
    float* sP = sO;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sP[(r0 * C + sl * 8 + j) * 2] = ssum[j];
      sP[(r0 * C + sl * 8 + j) * 2 + 1] = ssq[j];
    }
    __syncthreads();
    if (tid < 2 * C) {
      const int c = tid >> 1, st = tid & 1;
      double a = 0.0;
#pragma unroll 8
      for (int g = 0; g < NT_ / PPP; ++g) a += (double)sP[(g * C + c) * 2 + st];
      ds_stat_add(p.stats + ((long)b * C + c) * 2 + st, (long long)llrint(a * (st ? DS_STAT_SQ_SCALE : DS_STAT_SUM_SCALE)));
    }
  }
  AT_MARK(11)
  AT_FLUSH
}

}  // namespace

bool ds_attn_fused_eligible(int dtype, int channels, int L) { return dtype == DS_BF16 && channels == C && L >= 16 && L <= LMAX && L % 16 == 0; }

int ds_launch_attn_fused(const AttnFusedArgs& a, hipStream_t st) {
  DS_CHECK(ds_attn_fused_eligible(DS_BF16, a.C, a.L), "attn_fused: unsupported shape");
  DS_CHECK(a.x && a.y && a.wqk && a.wv && a.wo && a.bqk && a.bv && a.bo, "attn_fused: null pointer");
  DS_CHECK(a.gn_acc || (a.gn_scale && a.gn_shift), "attn_fused: no GroupNorm statistics");
  // (x is read and y written with 16-byte vector accesses at pixel * ld + 8 k elements: strides in multiples of 8 elements,
  // 16-byte aligned bases)
  DS_CHECK(a.ldx % 8 == 0 && a.ldy % 8 == 0 && a.ldx >= C && a.ldy >= C, "attn_fused: bad pixel stride");
  DS_CHECK(((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.y & 15) == 0 && (a.x_bs % 8) == 0 && (a.y_bs % 8) == 0, "attn_fused: x / y must be 16-byte aligned");
  AttnK k;
  k.x = reinterpret_cast<const bf16_t*>(a.x); k.x_bs = a.x_bs; k.ldx = a.ldx;
  k.gn_acc = a.gn_acc; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta; k.gn_groups = a.gn_groups; k.gn_inv_count = a.gn_inv_count;
  k.gn_eps = a.gn_eps; k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift;
  k.wqk = reinterpret_cast<const bf16_t*>(a.wqk);
  k.wv = reinterpret_cast<const bf16_t*>(a.wv); k.wo = reinterpret_cast<const bf16_t*>(a.wo);
  k.bqk = a.bqk; k.bv = a.bv; k.bo = a.bo;
  k.y = reinterpret_cast<bf16_t*>(a.y); k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.stats = a.stats;
  k.L = a.L;
  k.qk_scale = 1.4426950408889634f / sqrtf((float)C);
#define ATTN_GO(LT_)                                                                     \
  do {                                                                                   \
    DS_FUNC_LDS_ONCE(attn_fused_kernel<LT_>, LDS_BYTES);                                 \
    hipLaunchKernelGGL(attn_fused_kernel<LT_>, dim3(a.B), dim3(NT_), LDS_BYTES, st, k);  \
    DS_LAUNCH_CHECK();                                                                   \
    return 0;                                                                            \
  } while (0)
  if (a.L <= 32) ATTN_GO(1);
  if (a.L <= 64) ATTN_GO(2);
  if (a.L <= 128) ATTN_GO(4);
  ATTN_GO(8);
#undef ATTN_GO
}
