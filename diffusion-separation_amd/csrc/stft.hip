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

#include <vector>

#include "common.h"

#define DS_MAXC 4  // num_sources + 1 <= 4

// table layout (floats): [cos n_fft | sin n_fft | hann n_fft] then, 64-float aligned,
//   dft_fwd [512][512]: row k < 256: cos(2 pi k n / n_fft), row 256 + k: -sin(.), columns n >= n_fft zero
//   dft_inv [512][512]: row n < n_fft: w[n]/n_fft * { c_j cos(2 pi j n/n_fft) | -c_j sin(.) } for column j | 256 + j,
//                       c_0 = c_Nyq = 1 (their imaginary parts are ignored like c2r does), c_j = 2 otherwise.
long ds_stft_fwd_offset(int n_fft) { return ((long)3 * n_fft + 63) & ~63L; }
long ds_stft_inv_offset(int n_fft) { return ds_stft_fwd_offset(n_fft) + 512L * 512L; }
int ds_build_stft_table(int n_fft, float** dev_tab) {
  if (n_fft > 510 || n_fft % 2) { ds_set_error("stft: n_fft must be even and <= 510"); return 1; }
  const int bins = n_fft / 2 + 1;
  std::vector<float> t((size_t)ds_stft_inv_offset(n_fft) + 512 * 512, 0.f);
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
