// stft.hip — the time <-> compressed-spectrogram front end of ScoreModelNCSNpp, fused:
//   stft_pack   : cat(xt, mix) -> right pad (n_fft - hop) -> STFT (periodic Hann, center, zero pad,
//                 one-sided) -> |z|^e e^{j angle} * factor -> [re.. | im..] channels (NHWC) -> frame
//                 padding to W -> (2x - 1)                       score_models.py:107-116, 41-48, 72-91;
//                                                                ncsnpp.py:347-349
//   istft_frames: channels -> complex -> z/|factor| -> |z|^(1/e) e^{j angle} -> irfft * window
//   istft_ola   : overlap-add / window envelope, trim n_fft/2, crop to T
//                                                                score_models.py:118-124, 59-64, 78-81, 99-105
// n_fft = 510 = 2*3*5*17 is not a power of two, so both transforms are evaluated as dense fp32 matrix
// products on the matrix cores (the f32 MFMA is an exact fmaf chain): frames [rows x 512] against a
// precomputed [512 x 512] real DFT / inverse-DFT(+window) matrix, through the same NT-GEMM kernel as the
// 1x1 convolutions.  Thin kernels do framing+window, compress+pack, unpack+decompress and overlap-add.
// Frame indexing is integer arithmetic identical to torch.stft(center=True):
//   frame f, tap n reads sample hop*f - n_fft/2 + n of the ORIGINAL signal (zero outside [0, T)).
#include <math.h>

#include <string.h>

#include <type_traits>
#include <vector>

#include "common.h"

#define DS_MAXC 4  // num_sources + 1 <= 4

// table layout (floats): [cos n_fft | sin n_fft | hann n_fft] then, 64-float aligned,
//   dft_fwd [512][512]: row k < 256: cos(2 pi k n / n_fft), row 256 + k: -sin(.), columns n >= n_fft zero
//   dft_inv [512][512]: row n < n_fft: w[n]/n_fft * { c_j cos(2 pi j n/n_fft) | -c_j sin(.) } for column j | 256 + j,
//                       c_0 = c_Nyq = 1 (their imaginary parts are ignored like c2r does), c_j = 2 otherwise.
long ds_stft_fwd_offset(int n_fft) { return ((long)3 * n_fft + 63) & ~63L; }
long ds_stft_inv_offset(int n_fft) { return ds_stft_fwd_offset(n_fft) + 512L * 512L; }
// round 5: MFMA-fragment-major copy of the forward DFT with the Hann window folded in, as hi / lo BFLOAT16 halves in both builds
// (like the split mode's planes: the decompressed spectrum |z|^2 of the inverse transform reaches 1e6 at t = 0.03 — past the range
// of IEEE half precision, whose hi half became inf and lo half NaN in a first version) (the fused STFT kernel below): [part hi | lo][k-step 32][row tile 16][lane 64][8] — 2 x 512 KB, in float units
long ds_stft_ffrag_offset(int n_fft) { return ds_stft_inv_offset(n_fft) + 512L * 512L; }
constexpr long DS_STFT_FFRAG_FLOATS = 2L * 16 * 32 * 64 * 8 / 2;
// ... and of the inverse DFT (window, 1 / n_fft and the one-sided weights folded in): [part][k-step 32][column tile 16 (taps n)][lane][8]
long ds_stft_ifrag_offset(int n_fft) { return ds_stft_ffrag_offset(n_fft) + DS_STFT_FFRAG_FLOATS; }
int ds_build_stft_table(int n_fft, float** dev_tab) {
  if (n_fft > 510 || n_fft % 2) { ds_set_error("stft: n_fft must be even and <= 510"); return 1; }
  const int bins = n_fft / 2 + 1;
  std::vector<float> t((size_t)ds_stft_ifrag_offset(n_fft) + DS_STFT_FFRAG_FLOATS, 0.f);
  for (int n = 0; n < n_fft; ++n) {
    const double a = 2.0 * M_PI * (double)n / (double)n_fft;
    t[n] = (float)cos(a);
    t[n_fft + n] = (float)sin(a);
    t[2 * n_fft + n] = (float)(0.5 * (1.0 - cos(a)));  // torch.hann_window(n_fft) (periodic)
  }
  float* fw = t.data() + ds_stft_fwd_offset(n_fft);
  float* iv = t.data() + ds_stft_inv_offset(n_fft);
  for (int k = 0; k < bins; ++k)
    for (int n = 0; n < n_fft; ++n) {
      const double a = 2.0 * M_PI * (double)(((long)k * n) % n_fft) / (double)n_fft;
      fw[(size_t)k * 512 + n] = (float)cos(a);
      fw[(size_t)(256 + k) * 512 + n] = (float)(-sin(a));
      const double cj = (k == 0 || k == bins - 1) ? 1.0 : 2.0;
      const double wn = 0.5 * (1.0 - cos(2.0 * M_PI * (double)n / (double)n_fft)) / (double)n_fft;
      iv[(size_t)n * 512 + k] = (float)(wn * cj * cos(a));
      iv[(size_t)n * 512 + 256 + k] = (k == 0 || k == bins - 1) ? 0.f : (float)(-wn * cj * sin(a));
    }
  {  // DFT rows interleaved (row 2 k = Re bin k, 2 k + 1 = Im bin k): a lane's accumulator quad holds two whole complex bins
    bf16_t* fr = reinterpret_cast<bf16_t*>(t.data() + ds_stft_ffrag_offset(n_fft));
    const size_t part = (size_t)16 * 32 * 64 * 8;
    for (int rt = 0; rt < 16; ++rt)
      for (int ks = 0; ks < 32; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int r = rt * 32 + (lane & 31), kb = r >> 1, im = r & 1, n = 16 * ks + 8 * (lane >> 5) + j;
            float v = 0.f;
            if (n < n_fft && kb < bins) {
              const double a = 2.0 * M_PI * (double)(((long)kb * n) % n_fft) / (double)n_fft;
              const double wn = 0.5 * (1.0 - cos(2.0 * M_PI * (double)n / (double)n_fft));
              v = (float)(wn * (im ? -sin(a) : cos(a)));
            }
            const bf16_t hi = f2bf(v);
            const size_t o = (((size_t)ks * 16 + rt) * 64 + lane) * 8 + j;
            fr[o] = hi;
            fr[part + o] = f2bf(v - bf2f(hi));
          }
  }
  {  // inverse: B operand of frames[r][n] = sum_K U[r][K] dft_inv[n][K]: lane = tap n of the column tile, 8 consecutive K
    bf16_t* fr = reinterpret_cast<bf16_t*>(t.data() + ds_stft_ifrag_offset(n_fft));
    const size_t part = (size_t)16 * 32 * 64 * 8;
    for (int ct = 0; ct < 16; ++ct)
      for (int ks = 0; ks < 32; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int n = ct * 32 + (lane & 31), K = 16 * ks + 8 * (lane >> 5) + j;
            const float v = iv[(size_t)n * 512 + K];
            const bf16_t hi = f2bf(v);
            const size_t o = (((size_t)ks * 16 + ct) * 64 + lane) * 8 + j;
            fr[o] = hi;
            fr[part + o] = f2bf(v - bf2f(hi));
          }
  }
  float* d = nullptr;
  DS_HIP(hipMalloc(&d, t.size() * sizeof(float)));
  DS_HIP(hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  *dev_tab = d;
  return 0;
}

// rows = (b, ch, f): frames[row][n] = x_ch[b, hop f - n_fft/2 + n] * hann[n]  (n < n_fft), 0 for the 2 pad taps
__global__ __launch_bounds__(256) void stft_frame_kernel(const float* __restrict__ xt, const float* __restrict__ mix,
                                                         float* __restrict__ frames, int S, long Tlen, int n_fft,
                                                         int hop, int F, long rows, const float* __restrict__ tab) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one thread = 4 taps
  if (i >= rows * 128) return;
  const long row = i >> 7;
  const int n0 = (int)(i & 127) * 4;
  const int f = (int)(row % F);
  const long bc = row / F;
  const int NC = S + 1;
  const int ch = (int)(bc % NC);
  const long b = bc / NC;
  const float* src = (ch < S) ? xt + (b * S + ch) * Tlen : mix + b * Tlen;
  const long base = (long)f * hop - n_fft / 2;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + j;
    const long idx = base + n;
    v[j] = (n < n_fft && idx >= 0 && idx < Tlen) ? src[idx] * tab[2 * n_fft + n] : 0.f;
  }
  *reinterpret_cast<float4*>(frames + row * 512 + n0) = make_float4(v[0], v[1], v[2], v[3]);
}

// specT [512][rows] (row k: Re bin k, row 256+k: Im bin k) -> compress -> channel pack -> (2x-1) -> NHWC
template <typename T>
__global__ __launch_bounds__(256) void stft_pack_kernel(const float* __restrict__ specT, T* __restrict__ y, int S,
                                                        int bins, int F, int W, int Cpad, long rows, float expo,
                                                        float factor, int shift, int B) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * bins * W;
  if (i >= total) return;
  const int f = (int)(i % W);
  const int k = (int)((i / W) % bins);
  const long b = i / ((long)W * bins);
  const int NC = S + 1;
  float o[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) o[j] = 0.f;
  if (f < F) {
#pragma unroll
    for (int ch = 0; ch < DS_MAXC; ++ch) {
      if (ch < NC) {
        const long row = (b * NC + ch) * F + f;
        const float re = specT[(long)k * rows + row], im = specT[(long)(256 + k) * rows + row];
        // |z|^e e^{j angle(z)} * factor == z * |z|^(e-1) * factor (0 at z = 0); then 2x - 1
        const float mag = sqrtf(re * re + im * im);
        float sc = 0.f;
        if (mag > 0.f) sc = (expo == 0.5f) ? (1.0f / sqrtf(mag)) : ((expo == 1.0f) ? 1.0f : powf(mag, expo - 1.0f));
        sc *= factor;
        float vr = re * sc, vi = im * sc;
        if (shift) { vr = 2.f * vr - 1.f; vi = 2.f * vi - 1.f; }
        if (ch == 0) { o[0] = vr; }
        if (ch == 1) { o[1] = vr; }
        if (ch == 2) { o[2] = vr; }
        if (ch == 3) { o[3] = vr; }
        // imaginary parts follow the NC real parts
        const int q = NC + ch;
        if (q == 1) o[1] = vi; else if (q == 2) o[2] = vi; else if (q == 3) o[3] = vi; else if (q == 4) o[4] = vi;
        else if (q == 5) o[5] = vi; else if (q == 6) o[6] = vi; else if (q == 7) o[7] = vi;
      }
    }
  } else {  // zero-padded frame (score_models.py:83-91), then 2x-1
    const float pv = shift ? -1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = (j < 2 * NC) ? pv : 0.f;
  }
  T* dst = y + ((b * bins + k) * W + f) * Cpad;
  for (int c0 = 0; c0 < Cpad; c0 += 8) store8<T>(dst + c0, o + c0);
}

// ---- round 5: the forward transform as ONE kernel for the 16-bit engines (n_fft = 510, hop = 128).
// The three launches above move 25 MB of fp32 frames and 25 MB of fp32 spectra through HBM and run the DFT on the generic
// GEMM tile (18 + 42 + 14 us per evaluation at B = 16, T = 32000).  Here a block owns 32 consecutive frames of one utterance
// and one half of the DFT rows:
//   * the 32 * 128 + 382 samples its frames cover go to LDS ONCE per channel, as hi / lo bfloat16 halves (16 significand bits, the
//     range of fp32), 128 samples per 272-byte row: frame r, tap n is sample 128 r + n, so the frame
//     matrix is never formed — a B fragment (32 frames x 16 taps) is one conflict-free ds_read_b128 per lane;
//   * the A operand is the DFT matrix with the window folded in, fragment-major (ds_build_stft_table), hi / lo as well; three
//     MFMAs per product (hi hi + lo hi + hi lo); rows interleaved Re / Im: a lane's accumulator quad = two whole complex bins of
//     ONE frame for every channel, i.e. exactly the 16 bytes of two output pixels' channel vectors;
//   * epilogue in the accumulator layout: |z|^e e^{j angle} * factor, (2x - 1), channel pack, frame padding — 16-byte stores
//     that run 512 bytes contiguous along the frame axis.
#ifdef ST_TIMING  // profiling build only: phase ticks of wave 0 of every block (tools/stft_timing.sh)
__device__ unsigned long long g_st_dbg[16];
#define ST_DECL unsigned long long st_prev = __builtin_readcyclecounter();
#define ST_MARK(i) { const unsigned long long st_now = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_st_dbg[i], st_now - st_prev); st_prev = st_now; }
extern "C" int diffsep_st_debug_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_st_dbg), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_dbg), z, sizeof(z)); }
  return 0;
}
#else
#define ST_DECL
#define ST_MARK(i)
#endif
constexpr int SF_PITCH = 272, SF_ROWS = 35, SF_CH = SF_ROWS * SF_PITCH, SF_NS = 32 * 128 + 382;
// (waves per block: 8 — one 32-row DFT tile per wave — measured the same as 4, tools/istft_ab.sh: 32.7 vs 33.2 us: unlike the inverse
// kernel's, this product phase is within 1.4x of its MFMA time and the staging / epilogue phases are short)
#ifndef SF_NW
#define SF_NW 4
#endif
constexpr int SF_NT = 64 * SF_NW, SF_RT = 8 / SF_NW;  // threads per block; 32-row tiles of the block's half of the DFT per wave
template <int NC>
__global__ __launch_bounds__(SF_NT) void stft_fused_kernel(const float* __restrict__ xt, const float* __restrict__ mix,
                                                         bf16_t* __restrict__ y, long Tlen, int F, int W, float expo,
                                                         float factor, int shift, const uint4* __restrict__ dfrag) {
  extern __shared__ __attribute__((aligned(16))) char sm[];  // [part][channel][SF_CH]
  constexpr int S = NC - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
  const int ntile = W >> 5;
  const int rh = blockIdx.x & 1, ft = (blockIdx.x >> 1) % ntile, b = (blockIdx.x >> 1) / ntile;
  const int f0 = ft * 32;
  const float pv = shift ? -1.f : 0.f;
  ST_DECL
  if (f0 >= F) {  // a tile of padding frames only (score_models.py:83-91): the pad value, no transform
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = j < 2 * NC ? pv : 0.f;
    for (int i = tid; i < 128 * 32; i += SF_NT) {
      const int kb = rh * 128 + (i >> 5), f = f0 + (i & 31);
      store8<bf16_t>(y + (((long)b * 256 + kb) * W + f) * 8, o);
    }
    return;
  }
  // ---- the samples of the tile's frames: sample j of the tile = original sample 128 f0 - 255 + j (zeros outside [0, T))
  const long g0 = 128L * f0 - 255;
  // (SF_NSP = SF_NS rounded up to whole fragments: taps 510 and 511 of the last frame meet zero DFT columns, but they must be
  // FINITE — uninitialised LDS there turned into NaN x 0 for one utterance of a batch)
  constexpr int SF_NSP = (SF_NS + 7) & ~7;
  // (loads first, nine iterations at a time: one load -> convert -> LDS store per iteration is a chain of 27 memory latencies,
  // 60 of the first version's 72 us)
  constexpr int NPAIR = NC * (SF_NSP / 2), NIT = (NPAIR + SF_NT - 1) / SF_NT, GRP = 9;
#pragma unroll
  for (int base = 0; base < NIT; base += GRP) {
    float v0[GRP], v1[GRP];
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int idx = tid + (base + u) * SF_NT;
      v0[u] = 0.f; v1[u] = 0.f;
      if (base + u < NIT && idx < NPAIR) {
        const int c = idx / (SF_NSP / 2), j = 2 * (idx - c * (SF_NSP / 2));
        const float* src = c < S ? xt + ((long)b * S + c) * Tlen : mix + (long)b * Tlen;
        const long t0 = g0 + j;
        if (t0 >= 0 && t0 < Tlen) v0[u] = src[t0];
        if (t0 + 1 >= 0 && t0 + 1 < Tlen) v1[u] = src[t0 + 1];
      }
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int idx = tid + (base + u) * SF_NT;
      if (base + u < NIT && idx < NPAIR) {
        const int c = idx / (SF_NSP / 2), j = 2 * (idx - c * (SF_NSP / 2));
        const bf16_t h0 = f2bf(v0[u]), h1 = f2bf(v1[u]);
        const bf16_t l0 = f2bf(v0[u] - bf2f(h0)), l1 = f2bf(v1[u] - bf2f(h1));
        const int addr = 2 * j + 16 * (j >> 7);
        *reinterpret_cast<unsigned*>(sm + c * SF_CH + addr) = (unsigned)h0 | ((unsigned)h1 << 16);
        *reinterpret_cast<unsigned*>(sm + (NC + c) * SF_CH + addr) = (unsigned)l0 | ((unsigned)l1 << 16);
      }
    }
  }
  __syncthreads();
  ST_MARK(0)
  // ---- 32 frames x 32 RT DFT rows per wave (row tiles rt0 .. rt0 + RT - 1), NC channels
  constexpr int RT = SF_RT;
  f32x16 acc[RT][NC];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][c][e] = 0.f;
  const int rt0 = rh * 8 + RT * wave;
  // (k-step major: the 32 KB all waves of all blocks read for one k-step are CONTIGUOUS.  Tile-major, the same k-step of the 16
  // tiles sat 32 KB apart — on a quarter of the L2 channels: 2.5k cycles per k-step for 768 cycles of MFMAs)
  const uint4* ahi = dfrag + ((size_t)rt0 * 64 + lane);
  const uint4* alo = ahi + (size_t)16 * 32 * 64;
  constexpr int DEPTH = 3;  // A fragments in flight: DEPTH k-steps (L2 latency ~ 2 k-steps of 18 MFMAs)
  uint4 a[DEPTH][RT][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int t = 0; t < RT; ++t) { a[d][t][0] = ahi[((size_t)d * 16 + t) * 64]; a[d][t][1] = alo[((size_t)d * 16 + t) * 64]; }
  const char* bbase = sm + SF_PITCH * l32 + 16 * h;
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) {
    const int n = 16 * ks;  // (+ 8 h: the lane's half of the k-step, in bbase)
    const int boff = 2 * n + 16 * (n >> 7);  // (16 ks + 8 h stays inside one 128-sample row: the row's pad is the same for both halves)
    uint4 bh[NC], bl[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      bh[c] = *reinterpret_cast<const uint4*>(bbase + c * SF_CH + boff);
      bl[c] = *reinterpret_cast<const uint4*>(bbase + (NC + c) * SF_CH + boff);
    }
    uint4 ah[RT], al[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) { ah[t] = a[ks % DEPTH][t][0]; al[t] = a[ks % DEPTH][t][1]; }
    if (ks + DEPTH < 32) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        a[ks % DEPTH][t][0] = ahi[((size_t)(ks + DEPTH) * 16 + t) * 64];
        a[ks % DEPTH][t][1] = alo[((size_t)(ks + DEPTH) * 16 + t) * 64];
      }
    }
    // (product-major order — consecutive MFMAs on different accumulators — measured no faster here and slower in the inverse
    // kernel: the phase waits for the fragment stream, not for accumulator dependencies)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        acc[t][c] = mfma_bf32(ah[t], bh[c], acc[t][c]);
        acc[t][c] = mfma_bf32(al[t], bh[c], acc[t][c]);
        acc[t][c] = mfma_bf32(ah[t], bl[c], acc[t][c]);
      }
  }
  ST_MARK(1)
  // ---- epilogue: lane (frame l32, half h) holds rows 8 q + 4 h + i of each row tile = bins 16 rt + 4 q + 2 h + {0, 1}, Re / Im.
  // The exponent case is decided ONCE (uniform): with the three cases inside the loops every one of the 48 compressions carried
  // an inlined powf and IEEE sqrt / divide sequences — 11k VALU instructions per lane, 60 of the first version's 72 us.  The
  // 16-bit output (2^-9 / 2^-12 relative) does not see the 1-ulp hardware sqrt / rsq of the e = 0.5 path.
  const int f = f0 + l32;
  auto epilogue = [&](auto MODE_) __attribute__((always_inline)) {
    constexpr int MODE = decltype(MODE_)::value;  // 0: e = 0.5, 1: e = 1, 2: general
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int kb = 16 * (rt0 + t) + 4 * q + 2 * h + kk;
          float o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float re = acc[t][c][4 * q + 2 * kk], im = acc[t][c][4 * q + 2 * kk + 1];
            // |z|^e e^{j angle(z)} * factor == z * |z|^(e-1) * factor (0 at z = 0); then 2x - 1
            const float m2 = fmaf(re, re, im * im);
            float sc;
            if constexpr (MODE == 0) sc = m2 > 0.f ? __builtin_amdgcn_rsqf(__builtin_amdgcn_sqrtf(m2)) : 0.f;  // |z|^-1/2
            else if constexpr (MODE == 1) sc = 1.f;
            else sc = m2 > 0.f ? powf(sqrtf(m2), expo - 1.0f) : 0.f;
            sc *= factor;
            float vr = re * sc, vi = im * sc;
            if (shift) { vr = fmaf(2.f, vr, -1.f); vi = fmaf(2.f, vi, -1.f); }
            o[c] = vr;
            o[NC + c] = vi;
          }
          if (f >= F) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = j < 2 * NC ? pv : 0.f;
          }
          store8<bf16_t>(y + (((long)b * 256 + kb) * W + f) * 8, o);
        }
  };
  if (expo == 0.5f) epilogue(std::integral_constant<int, 0>{});
  else if (expo == 1.0f) epilogue(std::integral_constant<int, 1>{});
  else epilogue(std::integral_constant<int, 2>{});
  ST_MARK(2)
}

long ds_stft_workspace_bytes(int B, int S, long T, int n_fft, int hop) {
  const long F = 1 + (T + n_fft - hop) / hop;
  const long rows = (long)B * (S + 1) * F;
  return 2 * ((rows + 8) * 512 * 4 + 256);
}

int ds_launch_stft_pack(const float* xt, const float* mix, void* y, int B, int S, long T, int n_fft, int hop,
                        float exponent, float factor, int W, int Cpad, int shift, int dtype, const float* tab,
                        float* ws, hipStream_t st, int split) {
  DS_CHECK(S >= 1 && S + 1 <= DS_MAXC, "stft: num_sources must be in [1,3]");
  DS_CHECK(Cpad % 8 == 0 && Cpad >= 2 * (S + 1) && Cpad <= 16, "stft: bad channel padding");
  DS_CHECK(n_fft % 2 == 0 && n_fft >= 2 && n_fft <= 510 && hop >= 1, "stft: n_fft must be even and <= 510");
  const int F = 1 + (int)((T + n_fft - hop) / hop);
  DS_CHECK(W >= F, "stft: padded width smaller than the frame count");
  const int bins = n_fft / 2 + 1;
  if (dtype == DS_BF16 && n_fft == 510 && hop == 128 && Cpad == 8 && W % 32 == 0 && !(ds_default_opts() & DS_OPT_NO_STFT_FUSED)) {
    const uint4* dfrag = reinterpret_cast<const uint4*>(tab + ds_stft_ffrag_offset(n_fft));
    const unsigned nblk = (unsigned)(2 * (W / 32) * B);
#ifndef SF_LDS_MIN
#define SF_LDS_MIN 0  // (tools/istft_ab.sh: -DSF_LDS_MIN=84000 keeps two blocks from sharing a CU)
#endif
#define SFK(NC_)                                                                                                           \
  {                                                                                                                          \
    constexpr int LDS0_ = 2 * NC_ * SF_CH, LDS_ = LDS0_ > SF_LDS_MIN ? LDS0_ : SF_LDS_MIN;                                   \
    DS_FUNC_LDS_ONCE((stft_fused_kernel<NC_>), LDS_);                                                                        \
    hipLaunchKernelGGL((stft_fused_kernel<NC_>), dim3(nblk), dim3(SF_NT), LDS_, st, xt, mix, (bf16_t*)y, T, F, W, exponent, factor, shift, dfrag); \
  }
    if (S == 1) SFK(2) else if (S == 2) SFK(3) else SFK(4)
#undef SFK
    DS_LAUNCH_CHECK();
    return 0;
  }
  const long rows = (long)B * (S + 1) * F;
  const long rows_p = (rows + 7) & ~7L;  // pixel stride of specT (16-byte aligned vector stores)
  float* frames = ws;
  float* specT = ws + ((rows * 512 + 63) & ~63L);
  hipLaunchKernelGGL(stft_frame_kernel, dim3(cdiv(rows * 128, 256)), dim3(256), 0, st, xt, mix, frames, S, T, n_fft,
                     hop, F, rows, tab);
  DS_LAUNCH_CHECK();
  // specT[k][row] = sum_n dft_fwd[k][n] * frames[row][n]   (NT GEMM, fp32 MFMA)
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = tab + ds_stft_fwd_offset(n_fft); a.ldx = 512; a.x_bs = 0;
  a.w = frames; a.w_bs = 0;
  a.y = specT; a.ldy = (int)rows_p; a.y_bs = 0;
  a.B = 1; a.H = 1; a.W = 512; a.Cin = 512; a.Cout = (int)rows; a.taps = 1; a.dtype = DS_F32; a.split = split; a.out_scale = 1.f;
  if (ds_launch_conv(a, st)) return 1;
  const long total = (long)B * bins * W;
  if (dtype == DS_F32)
    hipLaunchKernelGGL(stft_pack_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, specT, (float*)y, S, bins, F,
                       W, Cpad, rows_p, exponent, factor, shift, B);
  else
    hipLaunchKernelGGL(stft_pack_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, specT, (bf16_t*)y, S, bins,
                       F, W, Cpad, rows_p, exponent, factor, shift, B);
  DS_LAUNCH_CHECK();
  return 0;
}

// rows = (b, s, f): U[row][j] = Re z_j, U[row][256 + j] = Im z_j with z = decompress(channels / |factor|):
// |z|^(1/e) e^{j angle} == z * |z|^(1/e - 1)                         score_models.py:59-64, 78-81
#define DS_FRAME_PITCH 512
// ow != null: x is the network's last pyramid tensor and the output layer — NCSNpp's `h = pyramid / t; output_layer(h)`,
// a 1x1 convolution on <= 8 channels (ncsnpp.py:472-477) — is applied on the fly: v[c] = (sum_k ow[c][k] x[k]) / t[b] + ob[c]
// (no separate launch, no packed output tensor in HBM)
template <typename T>
__global__ __launch_bounds__(256) void istft_unpack_kernel(const T* __restrict__ x, float* __restrict__ U, int S,
                                                           int bins, int F, int W, int Cpad, float expo, float factor,
                                                           long total, const float* __restrict__ ow,
                                                           const float* __restrict__ ob, const float* __restrict__ tdiv,
                                                           int ow_cin) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (b, f, j), j fastest: coalesced writes of U rows
  if (i >= total) return;
  const int j = (int)(i & 255);
  const int f = (int)((i >> 8) % F);
  const long b = (i >> 8) / F;
  const float inv_fac = 1.0f / fabsf(factor);
  float v[8];
  if (j < bins) load8<T>(x + ((b * bins + j) * W + f) * Cpad, v);
  if (ow && j < bins) {
    float h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = v[k];
    const float td = tdiv[b];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = 0.f;
      if (c < 2 * S) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < ow_cin) a = fmaf(ow[c * ow_cin + k], h[k], a);
        a = a / td + ob[c];
      }
      v[c] = a;
    }
  }
#pragma unroll
  for (int s = 0; s < DS_MAXC - 1; ++s) {
    if (s < S) {
      float re = 0.f, im = 0.f;
      if (j < bins) {
        const float vr = v[s] * inv_fac, vi = v[S + s] * inv_fac;
        const float mag = sqrtf(vr * vr + vi * vi);
        float sc = 0.f;
        if (mag > 0.f) sc = (expo == 0.5f) ? mag : ((expo == 1.0f) ? 1.0f : powf(mag, 1.0f / expo - 1.0f));
        re = vr * sc;
        im = vi * sc;
      }
      float* row = U + ((b * S + s) * F + f) * 512;
      row[j] = re;
      row[256 + j] = im;
    }
  }
}

// out[b,s,t] = sum_f frame_f[t + n_fft/2 - f hop] / sum_f w^2[...]   (torch.istft, center=True)
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, float* __restrict__ out,
                                                        long Tlen, int n_fft, int hop, int F,
                                                        const float* __restrict__ tab) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long bs = blockIdx.y;  // b * S + s
  if (t >= Tlen) return;
  float v = 0.f;
  if (t < (long)hop * (F - 1)) {  // beyond the iSTFT length adjust_length pads zeros (score_models.py:99-105)
    const long q = t + n_fft / 2;
    long f_hi = q / hop;
    if (f_hi > F - 1) f_hi = F - 1;
    long f_lo = (q - (n_fft - 1) + hop - 1) / hop;
    if (q - (n_fft - 1) <= 0) f_lo = 0;
    float num = 0.f, den = 0.f;
    for (long f = f_lo; f <= f_hi; ++f) {
      const int n = (int)(q - f * hop);
      const float w = tab[2 * n_fft + n];
      num += frames[(bs * F + f) * DS_FRAME_PITCH + n];
      den = fmaf(w, w, den);
    }
    v = num / den;
  }
  out[bs * Tlen + t] = v;
}

// ---- round 5: the inverse transform as ONE kernel for the 16-bit engines (n_fft = 510, hop = 128): output layer + decompress,
// inverse DFT, overlap-add and envelope division.  The three launches above write and re-read 2 x 12 MB of fp32 rows (26 + 36 +
// 8 us per evaluation).  A block owns one source of one utterance and 29 hops (3712 samples) of its output: the 32 frames
// g0 - 1 .. g0 + 30 are all that touch those samples, so the overlap-add is local to the block.
//   * prologue: the 32 x 256 pixels of the frames' columns -> `h = pyramid / t; output_layer(h)` for the source's two channels
//     (ncsnpp.py:472-477) -> z / |factor|, |z|^(1/e) e^{j angle} -> U[frame][Re 256 | Im 256] in LDS as hi / lo halves;
//   * frames = U dft_inv^T on the matrix cores, three MFMAs per product, U as the A operand (frames = rows), the fragment-major
//     inverse table as the B operand: a lane holds ONE tap n of 16 frames;
//   * overlap-add: frame r, tap n lands on sample 128 r + n - 383 of the block's segment: one contribution per wave, written to
//     per-wave copies of the segment and summed in a fixed order; then every sample is divided by its window envelope and written once.
constexpr int SI_PITCH = 1040, SI_U = 32 * SI_PITCH, SI_SEG = 29 * 128;
// NS sources per block (2 when S = 2: the fragment table — 1 MB per block out of L2, what the product phase waits for — then
// serves both; 1 otherwise)
// EM: 0 exponent 0.5, 1 exponent 1, 2 general — a TEMPLATE parameter: with a run-time case inside the pixel loop the compiler
// speculated the inlined powf of the general case for every pixel (200 instead of ~60 instructions)
// Waves per block.  The kernel is a chain of latency-bound phases (U prologue: global loads -> decompression -> LDS; products: the DFT
// fragment stream out of L2; overlap-add; envelope + stores) on ONE block per CU (133 KB of LDS), so what hides the latencies is more
// waves of the same block: 4 / 8 / 16 waves = 54.2 / 42.2 / 32.7 us per launch (B = 16, same bits; tools/istft_ab.sh, round 6).  16
// waves: one 32-tap column tile per wave, four waves share a copy of the overlap-add segment (their 128 taps = every residue once).
#ifndef SI_NW
#define SI_NW 16
#endif
constexpr int SI_NT = 64 * SI_NW, SI_CT = 16 / SI_NW;  // threads per block; 32-tap column tiles per wave
template <int NS, int EM>
__global__ __launch_bounds__(SI_NT, NS == 1 ? 2 : 1) void istft_fused_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int S,
                                                             long Tlen, int F, int W, int ld, float expo, float factor,
                                                             const float* __restrict__ ow, const float* __restrict__ ob,
                                                             const float* __restrict__ tdiv, int ow_cin,
                                                             const uint4* __restrict__ dfrag, const float* __restrict__ tab,
                                                             int nseg) {
  extern __shared__ __attribute__((aligned(16))) char sm[];  // per source: U hi [32][SI_PITCH] | U lo; after the products: 4 x ola [SI_SEG] fp32
  constexpr int SRC = 2 * SI_U;  // bytes per source
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
  const int nsg = S / NS;
  const int seg = blockIdx.x % nseg, s0 = ((blockIdx.x / nseg) % nsg) * NS, b = blockIdx.x / (nseg * nsg);
  const int g0 = 29 * seg, fA = g0 - 1;  // frame r of the block = frame fA + r
  ST_DECL
  // ---- U: thread -> (frame r fastest: 32 x 16 B contiguous per bin, then the bin)
  {
    const float inv_fac = 1.0f / fabsf(factor);
    float wre[NS][8], wim[NS][8], bre[NS], bim[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      const int s = s0 + ns;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        wre[ns][k] = (ow && k < ow_cin) ? ow[s * ow_cin + k] : (k == s ? 1.f : 0.f);
        wim[ns][k] = (ow && k < ow_cin) ? ow[(S + s) * ow_cin + k] : (k == S + s ? 1.f : 0.f);
      }
      bre[ns] = ow ? ob[s] : 0.f;
      bim[ns] = ow ? ob[S + s] : 0.f;
    }
    const float inv_td = ow ? 1.0f / tdiv[b] : 1.f;
    // thread -> frame r = tid & 31 (fixed), bins (tid >> 5) + BG it; eight loads in flight per pass (one load -> LDS store per
    // iteration was a chain of 32 memory latencies)
    constexpr int BG = SI_NT / 32;  // bin groups of the block
    const int r = tid & 31, f = fA + r;
    const bool fok = f >= 0 && f < F;
    char* u = sm + r * SI_PITCH;
#pragma unroll
    for (int base = 0; base < 256 / BG; base += 8) {
      uint4 raw[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int kb = (tid >> 5) + BG * (base + q);
        raw[q] = fok ? *reinterpret_cast<const uint4*>(x + (((long)b * 256 + kb) * W + f) * ld) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int kb = (tid >> 5) + BG * (base + q);
        float v[8];
        v[0] = h_lo(raw[q].x); v[1] = h_hi(raw[q].x); v[2] = h_lo(raw[q].y); v[3] = h_hi(raw[q].y);
        v[4] = h_lo(raw[q].z); v[5] = h_hi(raw[q].z); v[6] = h_lo(raw[q].w); v[7] = h_hi(raw[q].w);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          float re = 0.f, im = 0.f;
          if (fok) {
            float ar = 0.f, ai = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { ar = fmaf(wre[ns][k], v[k], ar); ai = fmaf(wim[ns][k], v[k], ai); }
            if (ow) { ar = fmaf(ar, inv_td, bre[ns]); ai = fmaf(ai, inv_td, bim[ns]); }
            const float vr = ar * inv_fac, vi = ai * inv_fac;
            const float m2 = fmaf(vr, vr, vi * vi);
            float sc;
            if constexpr (EM == 0) sc = __builtin_amdgcn_sqrtf(m2);   // |z|^(1/e - 1) = |z| at e = 0.5 (0 at z = 0)
            else if constexpr (EM == 1) sc = 1.f;
            else sc = m2 > 0.f ? powf(sqrtf(m2), 1.0f / expo - 1.0f) : 0.f;
            re = vr * sc;
            im = vi * sc;
          }
          const bf16_t rh_ = f2bf(re), ih_ = f2bf(im);
          char* un = u + ns * SRC;
          *reinterpret_cast<bf16_t*>(un + 2 * kb) = rh_;
          *reinterpret_cast<bf16_t*>(un + 2 * (256 + kb)) = ih_;
          *reinterpret_cast<bf16_t*>(un + SI_U + 2 * kb) = f2bf(re - bf2f(rh_));
          *reinterpret_cast<bf16_t*>(un + SI_U + 2 * (256 + kb)) = f2bf(im - bf2f(ih_));
        }
      }
    }
  }
  __syncthreads();
  ST_MARK(4)
  // ---- frames[r][n]: 32 frames x 32 SI_CT taps per wave (column tiles SI_CT wave .. + SI_CT - 1)
  constexpr int CT = SI_CT;
  f32x16 acc[NS][CT];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ns][c][e] = 0.f;
  const uint4* bhi = dfrag + ((size_t)(CT * wave) * 64 + lane);  // (k-step major, see stft_fused_kernel)
  const uint4* blo = bhi + (size_t)16 * 32 * 64;
#ifndef SI_DEPTH
#define SI_DEPTH 3  // k-steps of B fragments in flight per wave (tools/istft_ab.sh: -DSI_DEPTH=...)
#endif
  constexpr int DEPTH = SI_DEPTH;
  uint4 bf[DEPTH][CT][2];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int c = 0; c < CT; ++c) { bf[d][c][0] = bhi[((size_t)d * 16 + c) * 64]; bf[d][c][1] = blo[((size_t)d * 16 + c) * 64]; }
  const char* abase = sm + SI_PITCH * l32 + 16 * h;
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) {
    uint4 ah[NS], al[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      ah[ns] = *reinterpret_cast<const uint4*>(abase + ns * SRC + 32 * ks);
      al[ns] = *reinterpret_cast<const uint4*>(abase + ns * SRC + SI_U + 32 * ks);
    }
    uint4 bh_[CT], bl_[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) { bh_[c] = bf[ks % DEPTH][c][0]; bl_[c] = bf[ks % DEPTH][c][1]; }
    if (ks + DEPTH < 32) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        bf[ks % DEPTH][c][0] = bhi[((size_t)(ks + DEPTH) * 16 + c) * 64];
        bf[ks % DEPTH][c][1] = blo[((size_t)(ks + DEPTH) * 16 + c) * 64];
      }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        acc[ns][c] = mfma_bf32(ah[ns], bh_[c], acc[ns][c]);
        acc[ns][c] = mfma_bf32(al[ns], bh_[c], acc[ns][c]);
        acc[ns][c] = mfma_bf32(ah[ns], bl_[c], acc[ns][c]);
      }
  }
  ST_MARK(5)
  // ---- overlap-add: lane (tap n = 32 (4 wave + c) + l32, half h) holds frames r = 8 q + 4 h + i.  Sample p of the segment
  // receives frame r's tap n = p + 383 - 128 r: consecutive frames' taps are 128 apart, i.e. ONE contribution per wave — every wave
  // writes its own copy of the segment with plain stores (no atomics: float atomics would make the sum's order, and with it the
  // result's last bit, depend on the timing of the waves — tests/test_engine_gpu.py::test_concurrent_streams_are_bit_identical...),
  // over the U planes, which nobody reads any more
  __syncthreads();
  // (the squared window for the envelope, beside the four copies: the envelope loop below read it from global memory in a chain
  // of 15 x 4 dependent loads per thread — most of the first version's 100 us)
  float* w2 = reinterpret_cast<float*>(sm) + 4 * SI_SEG;
  for (int i = tid; i < 512; i += SI_NT) { const float w = i < 510 ? tab[2 * 510 + i] : 0.f; w2[i] = w * w; }
#pragma unroll
  for (int ns = 0; ns < NS; ++ns) {
    // copy j holds taps 128 j .. 128 j + 127 (every residue once: one contribution per sample) = the waves 4 j / CT ... of the block
    float* mine = reinterpret_cast<float*>(sm + ns * SRC) + (wave * CT / 4) * SI_SEG;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int n = 32 * (CT * wave + c) + l32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int r = 8 * (e >> 2) + 4 * h + (e & 3);
        const int p = 128 * r + n - 383;
        if (n < 510 && p >= 0 && p < SI_SEG) mine[p] = acc[ns][c][e];  // (frames outside [0, F) contributed zeros)
      }
    }
  }
  __syncthreads();
  ST_MARK(6)
  // ---- out[b, s, t] = ola / sum_f w^2[t + 255 - 128 f]   (torch.istft, center = True; zeros beyond 128 (F - 1): adjust_length)
  for (int i = tid; i < SI_SEG; i += SI_NT) {
    const long t = 128L * g0 + i;
    if (t >= Tlen) break;
    if (t < 128L * (F - 1)) {
      const long q = t + 255;
      long f_hi = q / 128;
      if (f_hi > F - 1) f_hi = F - 1;
      long f_lo = (q - 509 + 127) / 128;
      if (q - 509 <= 0) f_lo = 0;
      float den = 0.f;
      for (long f = f_lo; f <= f_hi; ++f) den += w2[(int)(q - f * 128)];
      const float inv_den = 1.0f / den;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        const float* ola = reinterpret_cast<const float*>(sm + ns * SRC);
        float num = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {  // wave w's contribution: tap n = 128 w + ((i + 383) & 127) of frame r = (i + 383 - n) / 128
          const int n = 128 * w + ((i + 383) & 127), r = (i + 383 - n) >> 7;
          if (n < 510 && r >= 0 && r < 32) num += ola[w * SI_SEG + i];
        }
        out[((long)b * S + s0 + ns) * Tlen + t] = num * inv_den;
      }
    } else {
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) out[((long)b * S + s0 + ns) * Tlen + t] = 0.f;
    }
  }
  ST_MARK(7)
}

long ds_istft_workspace_bytes(int B, int S, long T, int n_fft, int hop) {
  const long F = 1 + (T + n_fft - hop) / hop;
  return 2 * ((long)B * S * F * 512 * 4 + 256);
}

int ds_launch_istft(const void* x, float* out, int B, int S, long T, int n_fft, int hop, float exponent, float factor,
                    int W, int Cpad, int dtype, const float* tab, float* ws, hipStream_t st, int split, const float* ow,
                    const float* ob, const float* tdiv, int ow_cin) {
  DS_CHECK(n_fft % 2 == 0 && n_fft <= 510, "istft: n_fft must be even and <= 510");
  DS_CHECK(!ow || (ob && tdiv && ow_cin >= 1 && ow_cin <= 8), "istft: bad fused output layer");
  DS_CHECK(S >= 1 && S <= DS_MAXC - 1 && Cpad >= 2 * S && Cpad % 8 == 0 && 2 * S <= 8, "istft: bad source / channel count");
  const int F = 1 + (int)((T + n_fft - hop) / hop);
  DS_CHECK(W >= F, "istft: padded width smaller than the frame count");
  const int bins = n_fft / 2 + 1;
  if (dtype == DS_BF16 && n_fft == 510 && hop == 128 && Cpad % 8 == 0 && (!ow || ow_cin <= 8) && !(ds_default_opts() & DS_OPT_NO_STFT_FUSED)) {
    const uint4* dfrag = reinterpret_cast<const uint4*>(tab + ds_stft_ifrag_offset(n_fft));
    const int nseg = (int)cdiv(T, (long)SI_SEG);
    constexpr int LDS1 = 2 * SI_U;
    static_assert(4 * SI_SEG * 4 + 512 * 4 <= LDS1, "the four per-wave copies of the segment and the squared window fit the U planes");
#define ISK(NS_, EM_)                                                                                                                     \
  {                                                                                                                                          \
    DS_FUNC_LDS_ONCE((istft_fused_kernel<NS_, EM_>), NS_ * LDS1);                                                                            \
    hipLaunchKernelGGL((istft_fused_kernel<NS_, EM_>), dim3((unsigned)(B * (S / NS_) * nseg)), dim3(SI_NT), NS_ * LDS1, st, (const bf16_t*)x, out, \
                       S, T, F, W, Cpad, exponent, factor, ow, ob, tdiv, ow_cin, dfrag, tab, nseg);                                       \
  }
    const int em = exponent == 0.5f ? 0 : (exponent == 1.0f ? 1 : 2);
#ifdef SI_NS1  // (A/B: one source per block, two blocks per CU)
    if (false) {}
#else
    if (S == 2) { if (em == 0) ISK(2, 0) else if (em == 1) ISK(2, 1) else ISK(2, 2) }
#endif
    else { if (em == 0) ISK(1, 0) else if (em == 1) ISK(1, 1) else ISK(1, 2) }
#undef ISK
    DS_LAUNCH_CHECK();
    return 0;
  }
  const long rows = (long)B * S * F;
  float* U = ws;
  float* frames = ws + ((rows * 512 + 63) & ~63L);
  const long total = (long)B * F * 256;
  if (dtype == DS_F32)
    hipLaunchKernelGGL(istft_unpack_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const float*)x, U, S, bins,
                       F, W, Cpad, exponent, factor, total, ow, ob, tdiv, ow_cin);
  else
    hipLaunchKernelGGL(istft_unpack_kernel<bf16_t>, dim3(cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)x, U, S,
                       bins, F, W, Cpad, exponent, factor, total, ow, ob, tdiv, ow_cin);
  DS_LAUNCH_CHECK();
  // frames[row][n] = sum_K U[row][K] * dft_inv[n][K]   (NT GEMM, fp32 MFMA; window and 1/n_fft folded in)
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.x = U; a.ldx = 512; a.x_bs = 0;
  a.w = tab + ds_stft_inv_offset(n_fft); a.w_bs = 0;
  a.y = frames; a.ldy = DS_FRAME_PITCH; a.y_bs = 0;
  a.B = 1; a.H = 1; a.W = (int)rows; a.Cin = 512; a.Cout = 512; a.taps = 1; a.dtype = DS_F32; a.split = split; a.out_scale = 1.f;
  if (ds_launch_conv(a, st)) return 1;
  dim3 g2((unsigned)cdiv(T, 256), (unsigned)(B * S));
  hipLaunchKernelGGL(istft_ola_kernel, g2, dim3(256), 0, st, frames, out, T, n_fft, hop, F, tab);
  DS_LAUNCH_CHECK();
  return 0;
}
