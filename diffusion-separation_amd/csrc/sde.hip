// sde.hip — the predictor–corrector state updates of the reverse-diffusion sampler, time-domain
// helpers and the time embedding (all fp32, HBM-bound, one thread = one time index x all sources).
//
// MixSDE matrices are all of the form  M = a*A + p*P  with A = 11^T/S, P = I - A  (sdes/sdes.py:242-248),
// so M @ v = a*mean_s(v) + p*(v - mean_s(v)) for any number of sources — no [B,S,S] matmul needed.
//   ev1 = smin^2 (r^{2t} - 1),  ev2 = smin^2 (r^{2t} - e^{-2 lambda t}) / (1 + lambda/ln r)   sdes.py:296-309
//   L   = sqrt(ev1) A + sqrt(ev2) P                                                      sdes.py:315-320
//   g(t)= smin r^t sqrt(2 ln r);  G = g sqrt(dt), dt = 1/N (quirk Q1)                     sdes.py:275-284, 93-107
#include "common.h"

struct MixCoef { float a, p; };  // sqrt(ev1), sqrt(ev2)

__device__ inline void mix_eig(const SdeP& s, float t, float& ev1, float& ev2) {
  const float r = s.sigma_max / s.sigma_min;
  const float logsig = logf(r);
  const float mult = s.sigma_min * s.sigma_min;
  const float srp = powf(r, 2.0f * t);
  ev1 = mult * (srp - 1.0f);
  const float ex = expf(-2.0f * s.d_lambda * t);
  const float denom = 1.0f + s.d_lambda / logsig;
  ev2 = mult * (srp - ex) / denom;
}

// PriorMixSDE._std_sigma_mix  sdes.py:477-489: 0.5 * sqrt(clamp(avg_pool1d(mix^2, k, stride 1, pad k/2), 1e-4)),
// zero padding counted in the average; for even k the extra trailing output is dropped.
__global__ __launch_bounds__(256) void sde_sigma_mix_kernel(const float* __restrict__ mix, float* __restrict__ out,
                                                            long T, int k) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const float* m = mix + (long)b * T;
  const long lo = t - k / 2;
  float acc = 0.f;
  for (int j = 0; j < k; ++j) {
    const long i = lo + j;
    if (i >= 0 && i < T) acc = fmaf(m[i], m[i], acc);
  }
  float v = acc / (float)k;
  if (v < 1e-4f) v = 1e-4f;
  out[(long)b * T + t] = 0.5f * sqrtf(v);
}
int ds_launch_sigma_mix(const float* mix, float* out, int B, long T, int avg_len, hipStream_t st) {
  DS_CHECK(avg_len >= 1, "sigma_mix: avg_len must be positive");
  hipLaunchKernelGGL(sde_sigma_mix_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, mix, out, T, avg_len);
  DS_LAUNCH_CHECK();
  return 0;
}

// x_T = c*y + L(T=1) @ z          MixSDE.prior_sampling  sdes.py:334-346 (c = 0.5 hard-coded for S = 2: quirk Q2)
// PriorMixSDE (smix != null): L is scaled per sample by sigma_mix, mean is 0.5*mix for any S (sdes.py:564-587)
// lens (nullable, here and in the two updates below): per-utterance lengths of a batch of utterances of different
// lengths that share one padded frame count; samples t >= lens[b] of the state are kept at exactly zero, which makes
// every utterance of the batch evolve bit-for-bit as it would alone (the STFT of the zero tail is the zero padding).
__global__ __launch_bounds__(256) void sde_prior_kernel(SdeP s, const float* __restrict__ y,
                                                        const float* __restrict__ z, float* __restrict__ x, int S,
                                                        long T, const float* __restrict__ smix,
                                                        const int* __restrict__ lens) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  if (lens && t >= lens[b]) {
    for (int i = 0; i < S; ++i) x[((long)b * S + i) * T + t] = 0.f;
    return;
  }
  float ev1, ev2;
  mix_eig(s, 1.0f, ev1, ev2);
  const float sm = smix ? smix[(long)b * T + t] : 1.0f;
  const float a = sqrtf(ev1) * sm, p = sqrtf(ev2) * sm;
  const float c = (S == 2 || smix) ? 0.5f : 1.0f / (float)S;
  const float m = c * y[(long)b * T + t];
  float zz[DS_MAX_SRC], mz = 0.f;
  for (int i = 0; i < S; ++i) { zz[i] = z[((long)b * S + i) * T + t]; mz += zz[i]; }
  mz /= (float)S;
  for (int i = 0; i < S; ++i) x[((long)b * S + i) * T + t] = m + (a * mz + p * (zz[i] - mz));
}

// ald2 corrector step given score g:  x_mean = x + 2 snr^2 L L g ;  x = x_mean + (2 snr L) z
// sdes/correctors.py:115-126
// (x may alias xo — the engine updates its state in place — so x / xo / xm carry no __restrict__)
__global__ __launch_bounds__(256) void sde_corrector_kernel(SdeP s, float snr, const float* x,
                                                            const float* __restrict__ tt,
                                                            const float* __restrict__ score,
                                                            const float* __restrict__ z, float* xo, float* xm, int S,
                                                            long T, const float* __restrict__ smix, int variant,
                                                            const int* __restrict__ lens) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  if (lens && t >= lens[b]) {
    for (int i = 0; i < S; ++i) {
      const long o = ((long)b * S + i) * T + t;
      if (xm) xm[o] = 0.f;
      xo[o] = 0.f;
    }
    return;
  }
  float ev1, ev2;
  mix_eig(s, tt[b], ev1, ev2);
  // variant 1 = 'ald' (sdes/correctors.py:58-91): a scalar std = sqrt(sum_j (L L)[0, j]) = sqrt(ev1) for every source,
  // i.e. the same update with L replaced by sqrt(ev1) I
  if (variant == 1) ev2 = ev1;
  const float sm = smix ? smix[(long)b * T + t] : 1.0f;  // PriorMixSDE._std: L * sigma_mix (sdes.py:515-532)
  const float a = sqrtf(ev1) * sm, p = sqrtf(ev2) * sm;
  const float step = 2.0f * snr * snr;
  float g[DS_MAX_SRC], n[DS_MAX_SRC], mg = 0.f, mn = 0.f;
  for (int i = 0; i < S; ++i) {
    g[i] = score[((long)b * S + i) * T + t];
    n[i] = z ? z[((long)b * S + i) * T + t] : 0.f;
    mg += g[i];
    mn += n[i];
  }
  mg /= (float)S;
  mn /= (float)S;
  for (int i = 0; i < S; ++i) {
    // L @ (L @ g), evaluated as two applications like the reference does
    const float u = a * mg + p * (g[i] - mg);        // (L g)_i ; mean_s(L g) = a * mg
    const float llg = a * (a * mg) + p * (u - a * mg);
    const long o = ((long)b * S + i) * T + t;
    const float mean = x[o] + step * llg;
    const float a2 = 2.0f * snr * a, p2 = 2.0f * snr * p;
    if (xm) xm[o] = mean;
    xo[o] = mean + (a2 * mn + p2 * (n[i] - mn));
  }
}

// reverse-diffusion predictor given score:  f = -lambda P x dt ; rev_f = f - G^2 score ;
// x_mean = x - rev_f ; x = x_mean + G z          sdes/predictors.py:60-66, sdes/sdes.py:163-171
__global__ __launch_bounds__(256) void sde_predictor_kernel(SdeP s, int N, const float* x,
                                                            const float* __restrict__ tt,
                                                            const float* __restrict__ score,
                                                            const float* __restrict__ z, float* xo, float* xm, int S,
                                                            long T, const float* __restrict__ smix, int pflow,
                                                            const int* __restrict__ lens) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  if (lens && t >= lens[b]) {
    for (int i = 0; i < S; ++i) {
      const long o = ((long)b * S + i) * T + t;
      if (xm) xm[o] = 0.f;
      xo[o] = 0.f;
    }
    return;
  }
  const float r = s.sigma_max / s.sigma_min;
  const float dt = 1.0f / (float)N;
  const float sigma = s.sigma_min * powf(r, tt[b]);
  // PriorMixSDE.sde: diffusion = sigma(t) sqrt(2 ln r) * sigma_mix per sample (sdes.py:451-470)
  const float diffusion = sigma * sqrtf(2.0f * logf(r)) * (smix ? smix[(long)b * T + t] : 1.0f);
  const float G = diffusion * sqrtf(dt);
  float xv[DS_MAX_SRC], mx = 0.f;
  for (int i = 0; i < S; ++i) { xv[i] = x[((long)b * S + i) * T + t]; mx += xv[i]; }
  mx /= (float)S;
  for (int i = 0; i < S; ++i) {
    const long o = ((long)b * S + i) * T + t;
    const float drift = -s.d_lambda * (xv[i] - mx);
    const float f = drift * dt;
    // probability-flow ODE (sdes.py:165-171): half the score term, no noise
    const float rev_f = f - G * G * score[o] * (pflow ? 0.5f : 1.0f);
    const float mean = xv[i] - rev_f;
    if (xm) xm[o] = mean;
    xo[o] = mean + ((z && !pflow) ? G * z[o] : 0.f);
  }
}

int ds_launch_sde_prior(const SdeP& s, const float* y, const float* z, float* x, int B, int S, long T,
                        const float* smix, hipStream_t st, const int* lens) {
  DS_CHECK((s.kind == 1) == (smix != nullptr), "sde: PriorMixSDE needs sigma_mix, MixSDE must not get one");
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC, "sde: too many sources");
  hipLaunchKernelGGL(sde_prior_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, s, y, z, x, S, T, smix, lens);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_sde_corrector(const SdeP& s, float snr, const float* x, const float* t, const float* score,
                            const float* z, float* xo, float* xm, int B, int S, long T, const float* smix,
                            int variant, hipStream_t st, const int* lens) {
  DS_CHECK((s.kind == 1) == (smix != nullptr), "sde: PriorMixSDE needs sigma_mix, MixSDE must not get one");
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC, "sde: too many sources");
  DS_CHECK(variant == 0 || (variant == 1 && s.kind == 0), "sde: corrector variant must be ald2 (0) or ald on MixSDE (1)");
  hipLaunchKernelGGL(sde_corrector_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, s, snr, x, t, score, z, xo, xm, S,
                     T, smix, variant, lens);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_sde_predictor(const SdeP& s, int N, const float* x, const float* t, const float* score, const float* z,
                            float* xo, float* xm, int B, int S, long T, const float* smix, int pflow,
                            hipStream_t st, const int* lens) {
  DS_CHECK((s.kind == 1) == (smix != nullptr), "sde: PriorMixSDE needs sigma_mix, MixSDE must not get one");
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC && N >= 1, "sde: bad arguments");
  hipLaunchKernelGGL(sde_predictor_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, s, N, x, t, score, z, xo, xm, S, T,
                     smix, pflow, lens);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ SDE object surface (the L3 plug point)
// The pieces a user-written Predictor / Corrector reaches through the reference's SDE API — sde(), marginal_prob()
// (= _mean, _std), mult_std(), discretize(), reverse().discretize() (sdes/sdes.py:61-66,93-173,275-328,451-470,
// 515-537) — as unit kernels.  The fused sampler above never calls them; they exist so that code written against
// the reference interface runs on the mirror (diffsep_amd/sdes/sdes.py).

// drift_out = fs * (-lambda P x);  diffusion = gs * g(t) [* sigma_mix]:  MixSDE.sde (sdes.py:275-284) with
// fs = gs = 1, SDE.discretize (sdes.py:93-107) with fs = dt, gs = sqrt(dt).  diffusion is [B] for MixSDE and
// [B,S,T] for PriorMixSDE (sdes.py:451-470).
__global__ __launch_bounds__(256) void sde_coeff_kernel(SdeP s, const float* x, const float* __restrict__ tt,
                                                        const float* __restrict__ smix, float* drift,
                                                        float* __restrict__ diffusion, int S, long T, float fs,
                                                        float gs) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const float r = s.sigma_max / s.sigma_min;
  const float g = s.sigma_min * powf(r, tt[b]) * sqrtf(2.0f * logf(r));
  float xv[DS_MAX_SRC], mx = 0.f;
  for (int i = 0; i < S; ++i) { xv[i] = x[((long)b * S + i) * T + t]; mx += xv[i]; }
  mx /= (float)S;
  for (int i = 0; i < S; ++i) {
    const long o = ((long)b * S + i) * T + t;
    drift[o] = (-s.d_lambda * (xv[i] - mx)) * fs;
    if (smix) diffusion[o] = (g * smix[(long)b * T + t]) * gs;
  }
  if (!smix && t == 0) diffusion[b] = g * gs;
}
// mean = (A + exp(-lambda t) P) x0          MixSDE._mean / _mean_mix_mat (sdes.py:286-294)
__global__ __launch_bounds__(256) void sde_mean_kernel(SdeP s, const float* __restrict__ x0,
                                                       const float* __restrict__ tt, float* __restrict__ out, int S,
                                                       long T) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const float decay = expf(-tt[b] * s.d_lambda);
  float xv[DS_MAX_SRC], mx = 0.f;
  for (int i = 0; i < S; ++i) { xv[i] = x0[((long)b * S + i) * T + t]; mx += xv[i]; }
  mx /= (float)S;
  for (int i = 0; i < S; ++i) out[((long)b * S + i) * T + t] = mx + decay * (xv[i] - mx);
}
// L = sqrt(ev1) A + sqrt(ev2) P as a dense tensor: [B,S,S] (MixSDE._std, sdes.py:315-320) or, scaled per sample by
// sigma_mix, [B,S,S,T] (PriorMixSDE._std, sdes.py:515-532)
__global__ __launch_bounds__(256) void sde_std_kernel(SdeP s, const float* __restrict__ tt,
                                                      const float* __restrict__ smix, float* __restrict__ L, int S,
                                                      long T) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const long Tn = smix ? T : 1;
  if (t >= Tn) return;
  float ev1, ev2;
  mix_eig(s, tt[b], ev1, ev2);
  const float a = sqrtf(ev1), p = sqrtf(ev2);
  const float sm = smix ? smix[(long)b * T + t] : 1.0f;
  const float inv = 1.0f / (float)S;
  for (int c = 0; c < S; ++c)
    for (int d = 0; d < S; ++d) {
      const float A = inv, Pn = (c == d ? 1.0f : 0.0f) - inv;
      L[(((long)b * S + c) * S + d) * Tn + t] = (a * A + p * Pn) * sm;
    }
}
// out[b,c,t] = sum_d std[b,c,d(,t)] x[b,d,t]      MixSDE.mult_std (std @ x, sdes.py:326-328) /
// PriorMixSDE.mult_std (einsum "bcdt,bdt->bct", sdes.py:534-537) for ANY dense std the caller built
__global__ __launch_bounds__(256) void sde_mult_std_kernel(const float* __restrict__ L, const float* __restrict__ x,
                                                           float* __restrict__ out, int S, long T, int per_t) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  float xv[DS_MAX_SRC];
  for (int d = 0; d < S; ++d) xv[d] = x[((long)b * S + d) * T + t];
  for (int c = 0; c < S; ++c) {
    float acc = 0.f;
    for (int d = 0; d < S; ++d) {
      const long li = ((long)b * S + c) * S + d;
      acc = fmaf(per_t ? L[li * T + t] : L[li], xv[d], acc);
    }
    out[((long)b * S + c) * T + t] = acc;
  }
}
// rev_f = f - G^2 score (x 0.5 for the probability-flow ODE)      RSDE.discretize (sdes.py:163-171); G is [B] or [B,S,T]
__global__ __launch_bounds__(256) void sde_reverse_kernel(const float* __restrict__ f, const float* __restrict__ G,
                                                          const float* __restrict__ score, float* __restrict__ out,
                                                          long n_per_batch, int g_full, float half) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n_per_batch) return;
  const long o = (long)b * n_per_batch + i;
  const float g = g_full ? G[o] : G[b];
  out[o] = f[o] - g * g * score[o] * half;
}
int ds_launch_sde_coeff(const SdeP& s, const float* x, const float* t, const float* smix, float* drift,
                        float* diffusion, int B, int S, long T, float fs, float gs, hipStream_t st) {
  DS_CHECK((s.kind == 1) == (smix != nullptr), "sde: PriorMixSDE needs sigma_mix, MixSDE must not get one");
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC, "sde: too many sources");
  hipLaunchKernelGGL(sde_coeff_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, s, x, t, smix, drift, diffusion, S, T,
                     fs, gs);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_sde_mean(const SdeP& s, const float* x0, const float* t, float* out, int B, int S, long T,
                       hipStream_t st) {
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC, "sde: too many sources");
  hipLaunchKernelGGL(sde_mean_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, s, x0, t, out, S, T);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_sde_std(const SdeP& s, const float* t, const float* smix, float* L, int B, int S, long T,
                      hipStream_t st) {
  DS_CHECK((s.kind == 1) == (smix != nullptr), "sde: PriorMixSDE needs sigma_mix, MixSDE must not get one");
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC, "sde: too many sources");
  hipLaunchKernelGGL(sde_std_kernel, dim3(cdiv(smix ? T : 1, 256), B), dim3(256), 0, st, s, t, smix, L, S, T);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_sde_mult_std(const float* L, const float* x, float* out, int B, int S, long T, int per_t,
                           hipStream_t st) {
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC, "sde: too many sources");
  hipLaunchKernelGGL(sde_mult_std_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, L, x, out, S, T, per_t);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_sde_reverse(const float* f, const float* G, const float* score, float* out, int B, long n_per_batch,
                          int g_full, int pflow, hipStream_t st) {
  hipLaunchKernelGGL(sde_reverse_kernel, dim3(cdiv(n_per_batch, 256), B), dim3(256), 0, st, f, G, score, out,
                     n_per_batch, g_full, pflow ? 0.5f : 1.0f);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ Langevin corrector (sdes/correctors.py:35-55)
// step = 2 (snr * mean_b ||z_b|| / mean_b ||g_b||)^2 is ONE scalar for the whole batch (the reference couples the
// batch entries here); x_mean = x + step g ; x = x_mean + sqrt(2 step) z.
__global__ __launch_bounds__(1024) void batch_norm2_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                           double* __restrict__ out, long n) {
  __shared__ double sh[16];
  const int b = blockIdx.x;
  double a = 0.0, c = 0.0;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const double u = (double)g[(long)b * n + i], v = (double)z[(long)b * n + i];
    a += u * u;
    c += v * v;
  }
  a = wave_sum_d(a);
  c = wave_sum_d(c);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = a;
  __syncthreads();
  double ta = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) ta += sh[i];
  __syncthreads();
  if (lane == 0) sh[w] = c;
  __syncthreads();
  double tc = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tc += sh[i];
  if (threadIdx.x == 0) { out[2 * b] = sqrt(ta); out[2 * b + 1] = sqrt(tc); }
}
__global__ void langevin_step_kernel(const double* __restrict__ norms, int B, float snr, float* __restrict__ step) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double gn = 0.0, zn = 0.0;
    for (int b = 0; b < B; ++b) { gn += norms[2 * b]; zn += norms[2 * b + 1]; }
    gn /= B; zn /= B;
    const float r = snr * (float)zn / (float)gn;
    step[0] = r * r * 2.0f;
  }
}
__global__ __launch_bounds__(256) void langevin_update_kernel(const float* x, const float* __restrict__ g,
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ step, float* xo, float* xm,
                                                              long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float st = step[0];
  const float mean = x[i] + st * g[i];
  if (xm) xm[i] = mean;
  xo[i] = mean + z[i] * sqrtf(st * 2.0f);
}
int ds_launch_langevin(float snr, const float* x, const float* score, const float* z, float* xo, float* xm, int B,
                       long n_per_batch, void* ws, hipStream_t st) {
  double* norms = reinterpret_cast<double*>(ws);
  float* step = reinterpret_cast<float*>(norms + 2 * (size_t)B);
  hipLaunchKernelGGL(batch_norm2_kernel, dim3(B), dim3(1024), 0, st, score, z, norms, n_per_batch);
  DS_LAUNCH_CHECK();
  hipLaunchKernelGGL(langevin_step_kernel, dim3(1), dim3(64), 0, st, norms, B, snr, step);
  DS_LAUNCH_CHECK();
  const long n = (long)B * n_per_batch;
  hipLaunchKernelGGL(langevin_update_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, x, score, z, step, xo, xm, n);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ block reductions (fp64)
__device__ inline double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}

// normalize_batch  pl_model.py:81-88: (mix - mean) / clamp(std_unbiased, 1e-5) over (chan=1, time)
__global__ __launch_bounds__(1024) void normalize_kernel(const float* __restrict__ mix, float* __restrict__ out,
                                                         float* __restrict__ mean_o, float* __restrict__ std_o,
                                                         long T) {
  __shared__ double sh[16];
  const int b = blockIdx.x;
  const float* m = mix + (long)b * T;
  double s = 0.0;
  for (long i = threadIdx.x; i < T; i += blockDim.x) s += (double)m[i];
  const double mean = block_sum_d(s, sh) / (double)T;
  double q = 0.0;
  for (long i = threadIdx.x; i < T; i += blockDim.x) {
    const double d = (double)m[i] - mean;
    q += d * d;
  }
  const double var = block_sum_d(q, sh) / (double)(T > 1 ? T - 1 : 1);
  float sd = (float)sqrt(var);
  if (sd < 1e-5f) sd = 1e-5f;
  const float mu = (float)mean;
  for (long i = threadIdx.x; i < T; i += blockDim.x) out[(long)b * T + i] = (m[i] - mu) / sd;
  if (threadIdx.x == 0) {
    if (mean_o) mean_o[b] = mu;
    if (std_o) std_o[b] = sd;
  }
}
int ds_launch_normalize(const float* mix, float* out, float* mean, float* std, int B, long T, hipStream_t st) {
  hipLaunchKernelGGL(normalize_kernel, dim3(B), dim3(1024), 0, st, mix, out, mean, std, T);
  DS_LAUNCH_CHECK();
  return 0;
}

// scale_output  separate.py:73-78: alpha = sum(mix*sep) / sum(sep^2 + 1e-10); sep *= alpha
__global__ __launch_bounds__(1024) void scale_output_kernel(const float* __restrict__ mix, float* __restrict__ sep,
                                                            int S, long T) {
  __shared__ double sh[16];
  const int s = blockIdx.x, b = blockIdx.y;
  const float* m = mix + (long)b * T;
  float* x = sep + ((long)b * S + s) * T;
  double num = 0.0, den = 0.0;
  for (long i = threadIdx.x; i < T; i += blockDim.x) {
    const double v = (double)x[i];
    num += (double)m[i] * v;
    den += v * v + 1e-10;
  }
  num = block_sum_d(num, sh);
  den = block_sum_d(den, sh);
  const float alpha = (float)(num / den);
  for (long i = threadIdx.x; i < T; i += blockDim.x) x[i] = alpha * x[i];
}
int ds_launch_scale_output(const float* mix, float* sep, int B, int S, long T, hipStream_t st) {
  hipLaunchKernelGGL(scale_output_kernel, dim3(S, B), dim3(1024), 0, st, mix, sep, S, T);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ Gram matrices for the BSS metrics
// out[b] = { ref ref^T, ref est^T, est est^T } (each [S][S], fp64): everything SI-SDR / SI-SIR / SI-SAR and the
// best permutation need (evaluate.py:103-132 -> fast_bss_eval.si_bss_eval_sources); one pass over the waveforms.
__global__ __launch_bounds__(1024) void gram_kernel(const float* __restrict__ ref, const float* __restrict__ est,
                                                    double* __restrict__ out, int S, long T) {
  __shared__ double sh[16];
  const int b = blockIdx.x;
  double acc[3 * DS_MAX_SRC * DS_MAX_SRC];
#pragma unroll
  for (int k = 0; k < 3 * DS_MAX_SRC * DS_MAX_SRC; ++k) acc[k] = 0.0;
  for (long i = threadIdx.x; i < T; i += blockDim.x) {
    double r[DS_MAX_SRC], e[DS_MAX_SRC];
#pragma unroll
    for (int a = 0; a < DS_MAX_SRC; ++a) {
      r[a] = a < S ? (double)ref[((long)b * S + a) * T + i] : 0.0;
      e[a] = a < S ? (double)est[((long)b * S + a) * T + i] : 0.0;
    }
#pragma unroll
    for (int a = 0; a < DS_MAX_SRC; ++a)
#pragma unroll
      for (int c = 0; c < DS_MAX_SRC; ++c) {
        acc[(0 * DS_MAX_SRC + a) * DS_MAX_SRC + c] += r[a] * r[c];
        acc[(1 * DS_MAX_SRC + a) * DS_MAX_SRC + c] += r[a] * e[c];
        acc[(2 * DS_MAX_SRC + a) * DS_MAX_SRC + c] += e[a] * e[c];
      }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int m = 0; m < 3; ++m)
    for (int a = 0; a < S; ++a)
      for (int c = 0; c < S; ++c) {
        double v = wave_sum_d(acc[(m * DS_MAX_SRC + a) * DS_MAX_SRC + c]);
        __syncthreads();
        if (lane == 0) sh[w] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
          double t = 0.0;
          for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
          out[(((long)b * 3 + m) * S + a) * S + c] = t;
        }
      }
}
int ds_launch_gram(const float* ref, const float* est, double* out, int B, int S, long T, hipStream_t st) {
  DS_CHECK(S >= 1 && S <= DS_MAX_SRC, "gram: too many sources");
  hipLaunchKernelGGL(gram_kernel, dim3(B), dim3(1024), 0, st, ref, est, out, S, T);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ Philox4x32-10 + Box–Muller
__device__ inline void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long n, uint64_t seed, uint64_t sid) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;  // one thread = 4 outputs
  if (q * 4 >= n) return;
  uint32_t c0 = (uint32_t)q, c1 = (uint32_t)((uint64_t)q >> 32), c2 = (uint32_t)sid, c3 = (uint32_t)(sid >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u3 = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
  float v[4];
  sincosf(6.28318530718f * u1, &v[1], &v[0]);
  sincosf(6.28318530718f * u3, &v[3], &v[2]);
  v[0] *= r0; v[1] *= r0; v[2] *= r1; v[3] *= r1;
  for (int j = 0; j < 4; ++j)
    if (q * 4 + j < n) out[q * 4 + j] = v[j];
}
// Batch of utterances with their own seeds and lengths: utterance b gets exactly the draws a single-utterance call
// with (seed[b], stream id, S * lens[b] values) produces — element (s, t) is value s * lens[b] + t of that stream — laid
// out in rows of T; the tail t >= lens[b] is zero.
__global__ __launch_bounds__(256) void randn_batch_kernel(float* __restrict__ out, int S, long T,
                                                          const uint64_t* __restrict__ seeds,
                                                          const int* __restrict__ lens, uint64_t sid) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;  // one thread = 4 values of utterance blockIdx.y's stream
  const int b = blockIdx.y;
  const long len = lens[b], n = (long)S * len;
  // (zero tail: element index space of the padded rows that no stream value maps to)
  if (q * 4 < (long)S * T) {
    for (int j = 0; j < 4; ++j) {
      const long e = q * 4 + j;
      if (e < (long)S * T && e % T >= len) out[(long)b * S * T + e] = 0.f;
    }
  }
  if (q * 4 >= n) return;
  const uint64_t seed = seeds[b];
  uint32_t c0 = (uint32_t)q, c1 = (uint32_t)((uint64_t)q >> 32), c2 = (uint32_t)sid, c3 = (uint32_t)(sid >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u3 = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
  float v[4];
  sincosf(6.28318530718f * u1, &v[1], &v[0]);
  sincosf(6.28318530718f * u3, &v[3], &v[2]);
  v[0] *= r0; v[1] *= r0; v[2] *= r1; v[3] *= r1;
  for (int j = 0; j < 4; ++j) {
    const long i = q * 4 + j;
    if (i < n) out[((long)b * S + i / len) * T + i % len] = v[j];
  }
}
int ds_launch_randn_batch(float* out, int B, int S, long T, const uint64_t* seeds, const int* lens, uint64_t stream_id,
                          hipStream_t st) {
  const long n = (long)S * T;
  hipLaunchKernelGGL(randn_batch_kernel, dim3(cdiv((n + 3) / 4, 256), B), dim3(256), 0, st, out, S, T, seeds, lens,
                     stream_id);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_randn(float* out, long n, uint64_t seed, uint64_t stream_id, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(randn_kernel, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, st, out, n, seed, stream_id);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ dtype conversion / fill
template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void convert_kernel(const SRC* __restrict__ s, DST* __restrict__ d, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    Elt<DST>::st(d + i, Elt<SRC>::ld(s + i));
}
int ds_launch_convert(const void* src, void* dst, long n, int sd, int dd, hipStream_t st) {
  if (n <= 0) return 0;
  long nb = (n + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (sd == DS_F32 && dd == DS_F32)
    hipLaunchKernelGGL((convert_kernel<float, float>), dim3(nb), dim3(256), 0, st, (const float*)src, (float*)dst, n);
  else if (sd == DS_F32 && dd == DS_BF16)
    hipLaunchKernelGGL((convert_kernel<float, bf16_t>), dim3(nb), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
  else if (sd == DS_BF16 && dd == DS_F32)
    hipLaunchKernelGGL((convert_kernel<bf16_t, float>), dim3(nb), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
  else
    hipLaunchKernelGGL((convert_kernel<bf16_t, bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst,
                       n);
  DS_LAUNCH_CHECK();
  return 0;
}
__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}
int ds_launch_fill(float* p, float v, long n, hipStream_t st) {
  if (n <= 0) return 0;
  long nb = (n + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(fill_kernel, dim3(nb), dim3(256), 0, st, p, v, n);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ time embedding
// GaussianFourierProjection(log t): cat(sin, cos)(((log t * W) * 2) * pi)      layerspp.py:32-41, ncsnpp.py:327
__global__ __launch_bounds__(256) void fourier_kernel(const float* __restrict__ t, const float* __restrict__ Wf,
                                                      float* __restrict__ emb, int nf) {
  const int b = blockIdx.x;
  const float lt = logf(t[b]);
  for (int j = threadIdx.x; j < nf; j += 256) {
    const float xp = ((lt * Wf[j]) * 2.0f) * 3.14159265358979323846f;
    emb[(long)b * 2 * nf + j] = sinf(xp);
    emb[(long)b * 2 * nf + nf + j] = cosf(xp);
  }
}
int ds_launch_fourier(const float* t, const float* Wf, float* emb, int B, int nf, hipStream_t st) {
  hipLaunchKernelGGL(fourier_kernel, dim3(B), dim3(256), 0, st, t, Wf, emb, nf);
  DS_LAUNCH_CHECK();
  return 0;
}
// y[b][o] = sum_k act(x[b][k]) W[o][k] + bias[o]; one wave per output (nn.Linear: ncsnpp.py:339-343, layerspp.py:311-312)
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ y, int K,
                                                     int O, int silu_in) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (o >= O) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) {
    float v = x[(long)b * K + k];
    if (silu_in) v = silu_t<float>(v);
    acc = fmaf(v, W[(long)o * K + k], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) y[(long)b * O + o] = acc + (bias ? bias[o] : 0.f);
}
// The same product for a WIDE output (every block's Dense_0 at once, O ~ 5k) with the weights stored transposed
// [K][O]: a thread owns one output and 4 batch entries, walks K with coalesced weight loads and LDS-broadcast
// activations (one wave per (output, batch entry) with lane-strided K took 29 us per forward: 22k blocks re-activating
// the embedding; this takes ~6).
__global__ __launch_bounds__(256) void linear_t_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                       const float* __restrict__ bias, float* __restrict__ y, int B,
                                                       int K, int O, int silu_in) {
  extern __shared__ float xs[];  // [16][K]: act(x) of this block's batch entries
  const int b00 = blockIdx.y * 16;
  for (int i = threadIdx.x; i < 16 * K; i += 256) {
    const int bb = b00 + i / K;
    float v = bb < B ? x[(long)bb * K + i % K] : 0.f;
    if (silu_in) v = silu_t<float>(v);
    xs[i] = v;
  }
  __syncthreads();
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  if (o >= O) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* xg = xs + 4 * g * K;
  for (int k = 0; k < K; k += 16) {  // 16 independent weight loads in flight (K % 16 != 0: the tail reads zeros)
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = (k + i < K) ? Wt[(long)(k + i) * O + o] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (k + 4 * q >= K) break;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(xg + j * K + k + 4 * q);
        acc[j] = fmaf(v.x, w[4 * q], acc[j]);
        acc[j] = fmaf(v.y, w[4 * q + 1], acc[j]);
        acc[j] = fmaf(v.z, w[4 * q + 2], acc[j]);
        acc[j] = fmaf(v.w, w[4 * q + 3], acc[j]);
      }
    }
  }
  const float bo = bias ? bias[o] : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int bb = b00 + 4 * g + j;
    if (bb < B) y[(long)bb * O + o] = acc[j] + bo;
  }
}
// K % 64 == 0 (the engine's case): 16 outputs x 16 K-slices per block — 340 blocks for O = 5440, every thread's K / 16
// weight loads independent and in flight together, the slices summed through LDS: one memory round trip (26 -> ~6 us).
template <int KL>  // K / 16
__global__ __launch_bounds__(256) void linear_t16_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B,
                                                         int O, int silu_in) {
  constexpr int K = 16 * KL;
  extern __shared__ float xs[];        // [16][K] act(x), then [16 slices][16 b][16 o] partial sums
  float* red = xs + 16 * K;
  const int b00 = blockIdx.y * 16;
  const int ol = threadIdx.x & 15, ks = threadIdx.x >> 4;
  const int o = blockIdx.x * 16 + ol;
  float w[KL];
#pragma unroll
  for (int i = 0; i < KL; ++i) w[i] = o < O ? Wt[(long)(ks * KL + i) * O + o] : 0.f;
  for (int i = threadIdx.x; i < 16 * K; i += 256) {
    const int bb = b00 + i / K;
    float v = bb < B ? x[(long)bb * K + i % K] : 0.f;
    if (silu_in) v = silu_t<float>(v);
    xs[i] = v;
  }
  __syncthreads();
#pragma unroll 4
  for (int bb = 0; bb < 16; ++bb) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < KL; i += 4) {
      const float4 v = *reinterpret_cast<const float4*>(xs + bb * K + ks * KL + i);
      a = fmaf(v.x, w[i], a);
      a = fmaf(v.y, w[i + 1], a);
      a = fmaf(v.z, w[i + 2], a);
      a = fmaf(v.w, w[i + 3], a);
    }
    red[(ks * 16 + bb) * 16 + ol] = a;
  }
  __syncthreads();
  const int bb = threadIdx.x >> 4;
  float sum = 0.f;
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) sum += red[(k2 * 16 + bb) * 16 + ol];
  if (o < O && b00 + bb < B) y[(long)(b00 + bb) * O + o] = sum + (bias ? bias[o] : 0.f);
}
int ds_launch_linear_t(const float* x, const float* Wt, const float* bias, float* y, int B, int K, int O, int silu_in,
                       hipStream_t st) {
  DS_CHECK(K % 4 == 0 && K <= 2048, "linear_t: K must be a multiple of 4 (at most 2048)");
  if (K == 256 || K == 512 || K == 64 || K == 128) {
    const dim3 grid(cdiv(O, 16), cdiv(B, 16));
    const size_t lds = (size_t)(16 * K + 16 * 16 * 16) * 4;
    if (K == 256) hipLaunchKernelGGL(linear_t16_kernel<16>, grid, dim3(256), lds, st, x, Wt, bias, y, B, O, silu_in);
    else if (K == 512) hipLaunchKernelGGL(linear_t16_kernel<32>, grid, dim3(256), lds, st, x, Wt, bias, y, B, O, silu_in);
    else if (K == 128) hipLaunchKernelGGL(linear_t16_kernel<8>, grid, dim3(256), lds, st, x, Wt, bias, y, B, O, silu_in);
    else hipLaunchKernelGGL(linear_t16_kernel<4>, grid, dim3(256), lds, st, x, Wt, bias, y, B, O, silu_in);
    DS_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(linear_t_kernel, dim3(cdiv(O, 64), cdiv(B, 16)), dim3(256), (size_t)16 * K * 4, st, x, Wt, bias, y, B, K,
                     O, silu_in);
  DS_LAUNCH_CHECK();
  return 0;
}
// dst[k][off + o] = src[o][k]: one block's Dense_0.weight into the transposed concatenation [K][ldo]
__global__ __launch_bounds__(256) void dense_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int O, int K, int ldo, int off) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < O * K) dst[(long)(i % K) * ldo + off + i / K] = src[i];
}
int ds_launch_dense_transpose(const float* src, float* dst, int O, int K, int ldo, int off, hipStream_t st) {
  hipLaunchKernelGGL(dense_transpose_kernel, dim3(cdiv((long)O * K, 256)), dim3(256), 0, st, src, dst, O, K, ldo, off);
  DS_LAUNCH_CHECK();
  return 0;
}
int ds_launch_linear(const float* x, const float* W, const float* bias, float* y, int B, int K, int O, int silu_in,
                     hipStream_t st) {
  hipLaunchKernelGGL(linear_kernel, dim3(cdiv(O, 4), B), dim3(256), 0, st, x, W, bias, y, K, O, silu_in);
  DS_LAUNCH_CHECK();
  return 0;
}
