// attn_fused.hip — AttnBlockpp (layerspp.py:76-92) as ONE kernel for the 16-bit engines, C = 128 channels, L = H * W <= 256
// pixels (the 16 x 16 attention level and the 4 x 4 bottleneck of nf = 64 at 4 s): gfx950.
//
//   h = GroupNorm(x); q, k, v = NIN_0..2(h); w = softmax(q k^T / sqrt(C)); out = (x + NIN_3(w v)) / sqrt(2)
//
// Unfused, the block is 11 graph nodes (GroupNorm finalize + apply, three projections, V^T, scores, softmax, P V, output
// projection) of 5 - 9 us each on tensors of a few hundred KB: 60 - 65 us of latency chain per block, four blocks per network
// evaluation.  Here one workgroup of 8 waves owns one sample and nothing leaves the CU between the input and the output:
//   * h (GroupNorm affine applied, 16-bit) and V^T (16-bit) live in LDS (69.6 + 67.6 KB); every other intermediate lives in
//     registers: the 32 x 32 MFMA leaves a lane with register quads of 4 consecutive "weight-side" indices for its pixel row,
//     two v_permlane32_swap per 16 bytes turn that into the B fragment of the next product (conv3x3_rw.hip's epilogue trick),
//     so Q, Q', P and O go from accumulator to operand without touching LDS;
//   * K is never formed: S = Q K^T = (Q Wk) h^T, and the key bias adds a per-row constant that the softmax removes
//     (S[i][j] += q_i . b_k for every j) — one 32-MFMA product on the wave's own rows instead of a 256 x 128 tensor;
//   * V^T = Wv h^T + b_v is computed once per sample by all 8 waves (its rows are the B operand of P V: K-major in j);
//   * wave w owns query rows [32 w, 32 w + 32): Q (32 MFMAs), Q' = Q Wk (32), S = Q' h^T (L / 4 = 64), softmax in registers
//     (row statistics: one cross-half exchange), O = P V (64), out = O Wo^T (32), then bias, residual, 1 / sqrt(2), 8-byte
//     stores and the output's GroupNorm statistics (the consumer's normalisation reads accumulators, conv_mfma.hip);
//   * the four weight matrices are read as MFMA fragments straight from global memory in the fragment-major order of
//     ds_rw_frag_index (1 KB contiguous per wave instruction; the 8 waves read the same fragments: L1 hits).
// Rounding points = the unfused path's (h, V, P and the output in the storage type, everything else fp32) except that the
// key projection is folded into the query side; both are compared with the CPU oracle at the 16-bit tolerance.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int C = 128, NKB = C / 16, NT_ = 512, LMAX = 256;
constexpr int PH = C * 2 + 16;         // sH pitch: pixel row of 128 channels (+16 B: 16 consecutive rows = 16 distinct bank slots)
constexpr int PV = LMAX * 2 + 16;      // sVt pitch: channel row of 256 pixels
constexpr int OFF_VT = LMAX * PH, OFF_TAB = OFF_VT + C * PV, LDS_BYTES = OFF_TAB + 2 * C * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget of one CU");
static_assert(LMAX * (C + 4) * 4 <= OFF_TAB, "the statistics pass reuses the h / V^T area for the fp32 output tile");

struct AttnK {
  const bf16_t* x; long x_bs; int ldx;          // [B][L][ldx]
  const long long* gn_acc; const float* gn_gamma; const float* gn_beta; int gn_groups; float gn_inv_count; float gn_eps;
  const float* gn_scale; const float* gn_shift;  // [B][C] (used when gn_acc is null)
  const bf16_t* wq; const bf16_t* wkt; const bf16_t* wv; const bf16_t* wo;  // fragment-major [kb][n tile][lane][8]
  const float* bq; const float* bv; const float* bo;
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

__global__ __launch_bounds__(NT_) void attn_fused_kernel(AttnK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sH = smem;
  char* sVt = smem + OFF_VT;
  float* sTab = reinterpret_cast<float*>(smem + OFF_TAB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, h = lane >> 5;
  const int b = blockIdx.x, L = p.L;
  const int LT = (L + 31) >> 5;  // 32-pixel tiles
  const bf16_t* xb = p.x + (long)b * p.x_bs;

  // ---- GroupNorm scale / shift of sample b (layerspp.py:78: no activation)
  if (tid < C) {
    float sc, sh;
    if (p.gn_acc) {
      const int cpg = C / p.gn_groups, g0 = (tid / cpg) * cpg;
      long long ssum = 0, ssq = 0;
      for (int j = 0; j < cpg; ++j) {
        const long long* src = p.gn_acc + ((long)b * C + g0 + j) * 2;
        ssum += src[0];
        ssq += src[1];
      }
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
  }
  // the raw input: 16 pieces of 16 bytes per pixel, issued before the table is visible
  constexpr int PPP = C / 8;                         // pieces per pixel
  constexpr int NPC = LMAX * PPP / NT_;              // pieces per thread (8)
  u32x4_t raw[NPC];
#pragma unroll
  for (int k = 0; k < NPC; ++k) {
    const int pc = tid + k * NT_, px = pc / PPP, sl = pc % PPP;
    raw[k] = px < L ? *reinterpret_cast<const u32x4_t*>(xb + (long)px * p.ldx + sl * 8) : u32x4_t{0, 0, 0, 0};
  }
  __syncthreads();
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
  __syncthreads();

  const unsigned wlane = (unsigned)lane * 16u;  // fragment-major weights: fragment (kb, n tile) at ((kb * 4 + nt) * 64 + lane) * 16 B
  auto wfrag = [&](const bf16_t* w, int kb, int nt) {
    return *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(w) + (kb * 4 + nt) * 1024 + wlane);
  };
  auto hfrag = [&](int tile, int kb) {  // h rows 32 tile + l32, channels 16 kb + 8 h ..
    return *reinterpret_cast<const u32x4_t*>(sH + (tile * 32 + l32) * PH + kb * 32 + h * 16);
  };

  // ---- V^T[c][j] = sum_c' Wv[c][c'] h[j][c'] + b_v[c]: lane = channel c (B operand = Wv rows), registers = pixels j (A = h rows)
  {
    const int ct = wave & 3;                       // 32-channel tile of this wave; the two waves of a tile split the pixel tiles
    u32x4_t wv[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) wv[kb] = wfrag(p.wv, kb, ct);
    const float bias = p.bv[ct * 32 + l32];
    for (int jt = wave >> 2; jt < LT; jt += 2) {
      f32x16 acc = zero16();
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) acc = mma(hfrag(jt, kb), wv[kb], acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint2 v = make_uint2(pack_h2(acc[4 * g] + bias, acc[4 * g + 1] + bias), pack_h2(acc[4 * g + 2] + bias, acc[4 * g + 3] + bias));
        *reinterpret_cast<uint2*>(sVt + (ct * 32 + l32) * PV + (jt * 32 + 8 * g + 4 * h) * 2) = v;
      }
    }
  }
  __syncthreads();

  // ---- this wave's query rows
  const int i0 = wave * 32;
  f32x16 oacc[4];
  float inv_sum = 0.f;
  if (i0 < L) {
    u32x4_t qf[NKB];
    {  // Q = h Wq^T + b_q -> fragments
      u32x4_t hf[NKB];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) hf[kb] = hfrag(wave, kb);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        f32x16 acc = zero16();
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) acc = mma(wfrag(p.wq, kb, nt), hf[kb], acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bq = *reinterpret_cast<const float4*>(p.bq + nt * 32 + 8 * g + 4 * h);
          acc[4 * g] += bq.x; acc[4 * g + 1] += bq.y; acc[4 * g + 2] += bq.z; acc[4 * g + 3] += bq.w;
        }
        acc_to_frags(acc, qf[2 * nt], qf[2 * nt + 1]);
      }
    }
    u32x4_t q2[NKB];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {  // Q' = Q Wk (the key projection moved to the query side)
      f32x16 acc = zero16();
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) acc = mma(wfrag(p.wkt, kb, nt), qf[kb], acc);
      acc_to_frags(acc, q2[2 * nt], q2[2 * nt + 1]);
    }
    // S[i][j] = Q'[i] . h[j]: lane = row i, registers = pixels j (8 tiles of 32)
    f32x16 s[LMAX / 32];
#pragma unroll
    for (int jt = 0; jt < LMAX / 32; ++jt) {
      s[jt] = zero16();
      if (jt < LT) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) s[jt] = mma(hfrag(jt, kb), q2[kb], s[jt]);
      }
    }
    // softmax over j: the lane holds half of its row (the other k-half lane the rest)
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < LMAX / 32; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = jt * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
        if (jt < LT && j < L) mx = fmaxf(mx, s[jt][e]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    const float nm = -mx * p.qk_scale;
#pragma unroll
    for (int jt = 0; jt < LMAX / 32; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int j = jt * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
        const float pe = (jt < LT && j < L) ? __builtin_amdgcn_exp2f(fmaf(s[jt][e], p.qk_scale, nm)) : 0.f;
        s[jt][e] = pe;
        sum += pe;
      }
    sum += __shfl_xor(sum, 32, 64);
    inv_sum = __builtin_amdgcn_rcpf(sum);
    // O = P V (P normalised: the storage-type rounding of the probabilities matches the unfused path's softmax output)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) oacc[nt] = zero16();
#pragma unroll
    for (int jt = 0; jt < LMAX / 32; ++jt) {
      if (jt < LT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[jt][e] *= inv_sum;
        u32x4_t pf0, pf1;
        acc_to_frags(s[jt], pf0, pf1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const char* vr = sVt + (nt * 32 + l32) * PV + h * 16;
          oacc[nt] = mma(*reinterpret_cast<const u32x4_t*>(vr + (2 * jt) * 32), pf0, oacc[nt]);
          oacc[nt] = mma(*reinterpret_cast<const u32x4_t*>(vr + (2 * jt + 1) * 32), pf1, oacc[nt]);
        }
      }
    }
    u32x4_t of[NKB];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc_to_frags(oacc[nt], of[2 * nt], of[2 * nt + 1]);
    // out = (x + O Wo^T + b_o) / sqrt(2)
    const int i = i0 + l32;
    const bool rok = i < L;
    const bf16_t* xr = xb + (long)(rok ? i : 0) * p.ldx;
    bf16_t* yr = p.y + (long)b * p.y_bs + (long)(rok ? i : 0) * p.ldy;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x16 acc = zero16();
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) acc = mma(wfrag(p.wo, kb, nt), of[kb], acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nt * 32 + 8 * g + 4 * h;
        const float4 bo = *reinterpret_cast<const float4*>(p.bo + n);
        const uint2 xv = *reinterpret_cast<const uint2*>(xr + n);
        float v0 = (acc[4 * g] + bo.x + h_lo(xv.x)) * 0.70710678118654752440f;
        float v1 = (acc[4 * g + 1] + bo.y + h_hi(xv.x)) * 0.70710678118654752440f;
        float v2 = (acc[4 * g + 2] + bo.z + h_lo(xv.y)) * 0.70710678118654752440f;
        float v3 = (acc[4 * g + 3] + bo.w + h_hi(xv.y)) * 0.70710678118654752440f;
        if (rok) *reinterpret_cast<uint2*>(yr + n) = make_uint2(pack_h2(v0, v1), pack_h2(v2, v3));
        oacc[nt][4 * g] = rok ? v0 : 0.f; oacc[nt][4 * g + 1] = rok ? v1 : 0.f;
        oacc[nt][4 * g + 2] = rok ? v2 : 0.f; oacc[nt][4 * g + 3] = rok ? v3 : 0.f;
      }
    }
  }
  if (!p.stats) return;
  // ---- statistics of the output for the consumer's GroupNorm: channel sums over the sample's pixels (fp32 values before the
  // storage rounding, as the convolution epilogues do).  The h / V^T area is free once every wave is past its P V.
  __syncthreads();
  float* sO = reinterpret_cast<float*>(smem);  // [L rows][C + 4]
  constexpr int OP = C + 4;
  if (i0 < L) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(sO + (i0 + l32) * OP + nt * 32 + 8 * g + 4 * h) =
            make_float4(oacc[nt][4 * g], oacc[nt][4 * g + 1], oacc[nt][4 * g + 2], oacc[nt][4 * g + 3]);
  }
  __syncthreads();
  {
    const int c = tid & (C - 1), part = tid >> 7;  // 4 row quarters
    const int rows = LT * 32, r0 = part * (rows / 4), r1 = r0 + rows / 4;
    double a = 0.0, q = 0.0;
    for (int r = r0; r < r1; ++r) {
      const float v = sO[r * OP + c];
      a += (double)v;
      q += (double)v * (double)v;
    }
    ds_stat_add(p.stats + ((long)b * C + c) * 2, (long long)llrint(a * DS_STAT_SUM_SCALE));
    ds_stat_add(p.stats + ((long)b * C + c) * 2 + 1, (long long)llrint(q * DS_STAT_SQ_SCALE));
  }
}

}  // namespace

bool ds_attn_fused_eligible(int dtype, int channels, int L) { return dtype == DS_BF16 && channels == C && L >= 16 && L <= LMAX && L % 16 == 0; }

int ds_launch_attn_fused(const AttnFusedArgs& a, hipStream_t st) {
  DS_CHECK(ds_attn_fused_eligible(DS_BF16, a.C, a.L), "attn_fused: unsupported shape");
  DS_CHECK(a.x && a.y && a.wq && a.wkt && a.wv && a.wo && a.bq && a.bv && a.bo, "attn_fused: null pointer");
  DS_CHECK(a.gn_acc || (a.gn_scale && a.gn_shift), "attn_fused: no GroupNorm statistics");
  DS_CHECK(a.ldx % 8 == 0 && a.ldy % 4 == 0 && a.ldx >= C && a.ldy >= C, "attn_fused: bad pixel stride");
  AttnK k;
  k.x = reinterpret_cast<const bf16_t*>(a.x); k.x_bs = a.x_bs; k.ldx = a.ldx;
  k.gn_acc = a.gn_acc; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta; k.gn_groups = a.gn_groups; k.gn_inv_count = a.gn_inv_count;
  k.gn_eps = a.gn_eps; k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift;
  k.wq = reinterpret_cast<const bf16_t*>(a.wq); k.wkt = reinterpret_cast<const bf16_t*>(a.wkt);
  k.wv = reinterpret_cast<const bf16_t*>(a.wv); k.wo = reinterpret_cast<const bf16_t*>(a.wo);
  k.bq = a.bq; k.bv = a.bv; k.bo = a.bo;
  k.y = reinterpret_cast<bf16_t*>(a.y); k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.stats = a.stats;
  k.L = a.L;
  k.qk_scale = 1.4426950408889634f / sqrtf((float)C);
  DS_FUNC_LDS_ONCE(attn_fused_kernel, LDS_BYTES);
  hipLaunchKernelGGL(attn_fused_kernel, dim3(a.B), dim3(NT_), LDS_BYTES, st, k);
  DS_LAUNCH_CHECK();
  return 0;
}
