// stft.hip — the time <-> compressed-spectrogram front end of ScoreModelNCSNpp, fused:
//   stft_pack   : cat(xt, mix) -> right pad (n_fft - hop) -> STFT (periodic Hann, center, zero pad,
//                 one-sided) -> |z|^e e^{j angle} * factor -> [re.. | im..] channels (NHWC) -> frame
//                 padding to W -> (2x - 1)                       score_models.py:107-116, 41-48, 72-91;
//                                                                ncsnpp.py:347-349
//   istft_frames: channels -> complex -> z/|factor| -> |z|^(1/e) e^{j angle} -> irfft * window
//   istft_ola   : overlap-add / window envelope, trim n_fft/2, crop to T
//                                                                score_models.py:118-124, 59-64, 78-81, 99-105
// n_fft = 510 = 2*3*5*17 is not a power of two and a frame is tiny, so the DFT is evaluated directly
// from an LDS twiddle table (index k*n mod n_fft kept incrementally): 256 bins x 510 taps per frame,
// all fp32.  Frame indexing is integer arithmetic identical to torch.stft(center=True):
//   frame f, tap n reads sample 128 f - 255 + n of the ORIGINAL signal (zero outside [0, T)).
#include <math.h>

#include <vector>

#include "common.h"

#define DS_MAXC 4  // num_sources + 1 <= 4

int ds_build_stft_table(int n_fft, float** dev_tab) {
  std::vector<float> t(3 * (size_t)n_fft);
  for (int n = 0; n < n_fft; ++n) {
    const double a = 2.0 * M_PI * (double)n / (double)n_fft;
    t[n] = (float)cos(a);
    t[n_fft + n] = (float)sin(a);
    t[2 * n_fft + n] = (float)(0.5 * (1.0 - cos(a)));  // torch.hann_window(n_fft) (periodic)
  }
  float* d = nullptr;
  DS_HIP(hipMalloc(&d, t.size() * sizeof(float)));
  DS_HIP(hipMemcpy(d, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  *dev_tab = d;
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void stft_pack_kernel(const float* __restrict__ xt, const float* __restrict__ mix,
                                                        T* __restrict__ y, int S, long Tlen, int n_fft, int hop, int F,
                                                        int W, int Cpad, float expo, float factor, int shift,
                                                        const float* __restrict__ tab) {
  extern __shared__ float sm[];  // cos[n_fft] | sin[n_fft] | frames [NC][n_fft]
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int NC = S + 1;
  const int bins = n_fft / 2 + 1;
  if (f >= F) {  // zero-padded frame (score_models.py:83-91), then 2x-1
    const float pv = shift ? -1.f : 0.f;
    for (int k = tid; k < bins; k += 256) {
      T* dst = y + (((long)b * bins + k) * W + f) * Cpad;
      for (int c0 = 0; c0 < Cpad; c0 += 8) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (c0 + j < 2 * NC) ? pv : 0.f;
        store8<T>(dst + c0, o);
      }
    }
    return;
  }
  float* cs = sm;
  float* sn = sm + n_fft;
  float* fr = sm + 2 * n_fft;
  for (int n = tid; n < n_fft; n += 256) {
    cs[n] = tab[n];
    sn[n] = tab[n_fft + n];
  }
  const long base = (long)f * hop - n_fft / 2;
  for (int i = tid; i < NC * n_fft; i += 256) {
    const int ch = i / n_fft, n = i - ch * n_fft;
    const long idx = base + n;
    float v = 0.f;
    if (idx >= 0 && idx < Tlen) v = (ch < S) ? xt[((long)b * S + ch) * Tlen + idx] : mix[(long)b * Tlen + idx];
    fr[i] = v * tab[2 * n_fft + n];
  }
  __syncthreads();
  for (int k = tid; k < bins; k += 256) {
    float re[DS_MAXC], im[DS_MAXC];
#pragma unroll
    for (int c = 0; c < DS_MAXC; ++c) { re[c] = 0.f; im[c] = 0.f; }
    int idx = 0;
    for (int n = 0; n < n_fft; ++n) {
      const float c = cs[idx], s = sn[idx];
#pragma unroll
      for (int ch = 0; ch < DS_MAXC; ++ch) {
        if (ch < NC) {
          const float v = fr[ch * n_fft + n];
          re[ch] = fmaf(v, c, re[ch]);
          im[ch] = fmaf(-v, s, im[ch]);
        }
      }
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
    }
    // |z|^e e^{j angle(z)} * factor == z * |z|^(e-1) * factor (0 at z = 0); then 2x - 1
    float o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0.f;
#pragma unroll
    for (int ch = 0; ch < DS_MAXC; ++ch) {
      if (ch < NC) {
        const float mag = sqrtf(re[ch] * re[ch] + im[ch] * im[ch]);
        float sc = 0.f;
        if (mag > 0.f) sc = (expo == 0.5f) ? (1.0f / sqrtf(mag)) : ((expo == 1.0f) ? 1.0f : powf(mag, expo - 1.0f));
        sc *= factor;
        float vr = re[ch] * sc, vi = im[ch] * sc;
        if (shift) { vr = 2.f * vr - 1.f; vi = 2.f * vi - 1.f; }
        o[ch] = vr;
        o[NC + ch] = vi;
      }
    }
    T* dst = y + (((long)b * bins + k) * W + f) * Cpad;
    for (int c0 = 0; c0 < Cpad; c0 += 8) store8<T>(dst + c0, o + c0);
  }
}

int ds_launch_stft_pack(const float* xt, const float* mix, void* y, int B, int S, long T, int n_fft, int hop,
                        float exponent, float factor, int W, int Cpad, int shift, int dtype, const float* tab,
                        hipStream_t st) {
  DS_CHECK(S >= 1 && S + 1 <= DS_MAXC, "stft: num_sources must be in [1,3]");
  DS_CHECK(Cpad % 8 == 0 && Cpad >= 2 * (S + 1) && Cpad <= 16, "stft: bad channel padding");
  DS_CHECK(n_fft % 2 == 0 && n_fft >= 2 && hop >= 1, "stft: n_fft must be even");
  const int F = 1 + (int)((T + n_fft - hop) / hop);
  DS_CHECK(W >= F, "stft: padded width smaller than the frame count");
  const size_t lds = (size_t)(2 + S + 1) * n_fft * sizeof(float);
  dim3 grid((unsigned)W, (unsigned)B);
  if (dtype == DS_F32)
    hipLaunchKernelGGL(stft_pack_kernel<float>, grid, dim3(256), lds, st, xt, mix, (float*)y, S, T, n_fft, hop, F, W,
                       Cpad, exponent, factor, shift, tab);
  else
    hipLaunchKernelGGL(stft_pack_kernel<bf16_t>, grid, dim3(256), lds, st, xt, mix, (bf16_t*)y, S, T, n_fft, hop, F, W,
                       Cpad, exponent, factor, shift, tab);
  DS_LAUNCH_CHECK();
  return 0;
}

// one block per (frame, source, batch): decompress the 256 bins into LDS, then each thread
// evaluates output taps n = tid, tid + 256 of the length-n_fft inverse real DFT, times the window.
#define DS_FRAME_PITCH 512
template <typename T>
__global__ __launch_bounds__(256) void istft_frames_kernel(const T* __restrict__ x, float* __restrict__ frames, int S,
                                                           int n_fft, int F, int W, int Cpad, float expo, float factor,
                                                           const float* __restrict__ tab) {
  extern __shared__ float sm[];  // cos | sin | re[bins] | im[bins]
  const int f = blockIdx.x, s = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int bins = n_fft / 2 + 1;
  float* cs = sm;
  float* sn = sm + n_fft;
  float* re = sm + 2 * n_fft;
  float* im = re + bins;
  for (int n = tid; n < n_fft; n += 256) {
    cs[n] = tab[n];
    sn[n] = tab[n_fft + n];
  }
  const float inv_fac = 1.0f / fabsf(factor);
  for (int k = tid; k < bins; k += 256) {
    const T* src = x + (((long)b * bins + k) * W + f) * Cpad;
    float vr = Elt<T>::ld(src + s) * inv_fac, vi = Elt<T>::ld(src + S + s) * inv_fac;
    // |z|^(1/e) e^{j angle} == z * |z|^(1/e - 1)
    const float mag = sqrtf(vr * vr + vi * vi);
    float sc = 0.f;
    if (mag > 0.f) sc = (expo == 0.5f) ? mag : ((expo == 1.0f) ? 1.0f : powf(mag, 1.0f / expo - 1.0f));
    re[k] = vr * sc;
    im[k] = vi * sc;
  }
  __syncthreads();
  const float invn = 1.0f / (float)n_fft;
  for (int n = tid; n < n_fft; n += 256) {
    // c2r semantics: imaginary parts of the DC and Nyquist bins are ignored
    float acc = re[0] + ((n & 1) ? -re[bins - 1] : re[bins - 1]);
    float a2 = 0.f;
    int idx = n;  // k * n mod n_fft for k = 1
    for (int k = 1; k < bins - 1; ++k) {
      a2 = fmaf(re[k], cs[idx], a2);
      a2 = fmaf(-im[k], sn[idx], a2);
      idx += n;
      if (idx >= n_fft) idx -= n_fft;
    }
    acc = fmaf(2.f, a2, acc);
    frames[(((long)b * S + s) * F + f) * DS_FRAME_PITCH + n] = acc * invn * tab[2 * n_fft + n];
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

int ds_launch_istft(const void* x, float* out, int B, int S, long T, int n_fft, int hop, float exponent, float factor,
                    int W, int Cpad, int dtype, const float* tab, float* frames_ws, hipStream_t st) {
  DS_CHECK(n_fft % 2 == 0 && n_fft <= DS_FRAME_PITCH, "istft: n_fft must be even and <= 512");
  const int F = 1 + (int)((T + n_fft - hop) / hop);
  DS_CHECK(W >= F, "istft: padded width smaller than the frame count");
  const int bins = n_fft / 2 + 1;
  const size_t lds = (size_t)(2 * n_fft + 2 * bins) * sizeof(float);
  dim3 grid((unsigned)F, (unsigned)S, (unsigned)B);
  if (dtype == DS_F32)
    hipLaunchKernelGGL(istft_frames_kernel<float>, grid, dim3(256), lds, st, (const float*)x, frames_ws, S, n_fft, F, W,
                       Cpad, exponent, factor, tab);
  else
    hipLaunchKernelGGL(istft_frames_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)x, frames_ws, S, n_fft, F,
                       W, Cpad, exponent, factor, tab);
  DS_LAUNCH_CHECK();
  dim3 g2((unsigned)cdiv(T, 256), (unsigned)(B * S));
  hipLaunchKernelGGL(istft_ola_kernel, g2, dim3(256), 0, st, frames_ws, out, T, n_fft, hop, F, tab);
  DS_LAUNCH_CHECK();
  return 0;
}
