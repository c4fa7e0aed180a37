// norm.hip — HBM-bound NHWC kernels around the convolutions:
//   GroupNorm statistics (fp64 accumulation, wave shuffles + LDS), per-(b,c) scale/shift,
//   apply + SiLU (+ the [1,3,3,1] FIR x2 up / down resampling of BOTH act(GN(x)) and raw x in one read),
//   channel concat, row softmax.
// Reference semantics: nn.GroupNorm(min(C/4,32), C, eps=1e-6) + SiLU  layerspp.py:264-266,292,313;
// upsample_2d / downsample_2d  up_or_down_sampling.py:206-273 (closed forms: SURVEY.md §8a-15).
#include <stdlib.h>

#include "common.h"

// ------------------------------------------------------------------ GroupNorm statistics
// grid (nblk, B); each block reduces `ppb` pixels of one batch entry for all C channels.
// Thread t owns channel group cg = t % (C/8) and pixel lane pl = t / (C/8): consecutive threads read
// consecutive 16/32-byte channel vectors of one pixel -> fully coalesced rows.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ x2,
                                                       int ldx2, int C1, long npix, int C, long ppb,
                                                       double* __restrict__ part) {
  __shared__ double sh[4096];  // [npl][C][2], npl*C <= 2048
  const int ncg = C >> 3;
  const int npl = 256 / ncg;
  const int tid = threadIdx.x;
  const int cg = tid % ncg, pl = tid / ncg;
  const int b = blockIdx.y;
  const long p0 = (long)blockIdx.x * ppb;
  long p1 = p0 + ppb;
  if (p1 > npix) p1 = npix;
  double s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.0; ss[j] = 0.0; }
  if (pl < npl) {
    const bool second = x2 != nullptr && cg * 8 >= C1;
    const int ld = second ? ldx2 : ldx;
    const T* xb = second ? x2 + (long)b * npix * ldx2 + (cg * 8 - C1) : x + (long)b * npix * ldx + cg * 8;
    for (long p = p0 + pl; p < p1; p += npl) {
      float f[8];
      load8<T>(xb + p * ld, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double d = (double)f[j];
        s[j] += d;
        ss[j] = fma(d, d, ss[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sh[((pl * C) + cg * 8 + j) * 2 + 0] = s[j];
      sh[((pl * C) + cg * 8 + j) * 2 + 1] = ss[j];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    double a = 0.0, q = 0.0;
    for (int l = 0; l < npl; ++l) {
      a += sh[(l * C + c) * 2 + 0];
      q += sh[(l * C + c) * 2 + 1];
    }
    double* o = part + (((long)b * gridDim.x + blockIdx.x) * C + c) * 2;
    o[0] = a;
    o[1] = q;
  }
}

// grid (B): fold block partials -> group mean / rstd -> per-channel scale & shift:
//   y = x * scale + shift,  scale = rstd * gamma, shift = beta - mean * scale.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, int nblk, int C, int groups,
                                                          double count, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
  __shared__ double cs[1024], cq[1024];
  __shared__ float gm[256], gr[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    double a = 0.0, q = 0.0;
    for (int k = 0; k < nblk; ++k) {
      const double* o = part + (((long)b * nblk + k) * C + c) * 2;
      a += o[0];
      q += o[1];
    }
    cs[c] = a;
    cq[c] = q;
  }
  __syncthreads();
  const int cpg = C / groups;
  for (int g = tid; g < groups; g += 256) {
    double a = 0.0, q = 0.0;
    for (int j = 0; j < cpg; ++j) {
      a += cs[g * cpg + j];
      q += cq[g * cpg + j];
    }
    const double mean = a / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    gm[g] = (float)mean;
    gr[g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    const float sc = gr[g] * (gamma ? gamma[c] : 1.f);
    scale[(long)b * C + c] = sc;
    shift[(long)b * C + c] = (beta ? beta[c] : 0.f) - gm[g] * sc;
  }
}

// GroupNorm scale/shift from the channel-sum accumulators the conv epilogues fill (fixed point, common.h) — only
// for consumers that are not convolutions (the FIR resampling / attention GroupNorm kernels); the convolutions
// build the same table in their own prologue.  grid (B) x 256 threads; two sources = in-place concat.
__global__ __launch_bounds__(256) void gn_finalize_acc_kernel(const long long* __restrict__ a1, int C1,
                                                              const long long* __restrict__ a2, int C2, int groups,
                                                              double inv_count, float eps,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ scale, float* __restrict__ shift) {
  const int b = blockIdx.x, C = C1 + C2, cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g0 = (c / cpg) * cpg;
    long long ssum = 0, ssq = 0;
    for (int j = 0; j < cpg; ++j) {
      const int cj = g0 + j;
      const long long* src = cj < C1 ? a1 + ((long)b * C1 + cj) * 2 : a2 + ((long)b * C2 + (cj - C1)) * 2;
      ssum += __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ssq += __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double mean = (double)ssum * (1.0 / DS_STAT_SUM_SCALE) * inv_count;
    double var = (double)ssq * (1.0 / DS_STAT_SQ_SCALE) * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float sc = (float)(1.0 / sqrt(var + (double)eps)) * (gamma ? gamma[c] : 1.f);
    scale[(long)b * C + c] = sc;
    shift[(long)b * C + c] = (beta ? beta[c] : 0.f) - (float)mean * sc;
  }
}

int ds_launch_gn_finalize_acc(const long long* a1, int C1, const long long* a2, int C2, int B, long npix, int groups,
                              float eps, const float* gamma, const float* beta, float* scale, float* shift,
                              hipStream_t st) {
  const int C = C1 + C2;
  DS_CHECK(C % groups == 0 && groups > 0, "groupnorm(acc): bad group count");
  const double inv_count = 1.0 / ((double)npix * (double)(C / groups));
  hipLaunchKernelGGL(gn_finalize_acc_kernel, dim3(B), dim3(256), 0, st, a1, C1, a2, C2, groups, inv_count, eps, gamma,
                     beta, scale, shift);
  DS_LAUNCH_CHECK();
  return 0;
}

long ds_gn_workspace_bytes(int B, int H, int W, int C) {
  const long npix = (long)H * W;
  long nblk = cdiv(npix, 64);
  if (nblk > 64) nblk = 64;
  return (long)B * nblk * C * 2 * 8 + 256;
}

int ds_launch_gn_stats(const void* x, int ldx, const void* x2, int ldx2, int C1, int B, int H, int W, int C, int groups,
                       float eps, const float* gamma, const float* beta, void* ws, float* scale, float* shift,
                       int dtype, hipStream_t st) {
  DS_CHECK(!x2 || (C1 % 8 == 0 && C1 > 0 && C1 < C), "groupnorm: bad concat split");
  DS_CHECK(C % 8 == 0 && C <= 1024 && C >= 8, "groupnorm: C must be a multiple of 8 in [8,1024]");
  DS_CHECK(groups > 0 && groups <= 256 && C % groups == 0, "groupnorm: bad group count");
  DS_CHECK((C >> 3) <= 256, "groupnorm: C too large");
  const long npix = (long)H * W;
  long nblk = cdiv(npix, 64);
  if (nblk > 64) nblk = 64;
  const long ppb = (npix + nblk - 1) / nblk;
  double* part = reinterpret_cast<double*>(ws);
  dim3 grid((unsigned)nblk, (unsigned)B);
  if (dtype == DS_F32)
    hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, (const float*)x2, ldx2,
                       C1, npix, C, ppb, part);
  else
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)x2, ldx2,
                       C1, npix, C, ppb, part);
  DS_LAUNCH_CHECK();
  const double count = (double)npix * (double)(C / groups);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, st, part, (int)nblk, C, groups, count, eps, gamma, beta,
                     scale, shift);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ apply (+SiLU) (+FIR resample)
// One thread = one OUTPUT pixel x 8 channels.  MODE 0: same size; 1: FIR x2 up; 2: FIR x2 down.
//   down: y[m]    = (x[2m-1] + 3x[2m] + 3x[2m+1] + x[2m+2]) / 8   per axis, zeros outside
//   up:   y[2m]   = x[m-1]/4 + 3x[m]/4 ;  y[2m+1] = 3x[m]/4 + x[m+1]/4
// The activation is applied BEFORE resampling (layerspp.py:292-299) and out-of-image taps are zero.
template <typename T, int MODE, bool AFFINE>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, int ldx,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int C, T* __restrict__ y,
                                                       int ldy, T* __restrict__ xr, int ldxr, int B, int H, int W,
                                                       int act) {
  const int Ho = MODE == 1 ? 2 * H : (MODE == 2 ? H / 2 : H);
  const int Wo = MODE == 1 ? 2 * W : (MODE == 2 ? W / 2 : W);
  const int ncg = C >> 3;
  const long total = (long)B * Ho * Wo * ncg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % ncg);
    long r = i / ncg;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float sc[8], sf[8];
    if (AFFINE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        sc[j] = scale[(long)b * C + cg * 8 + j];
        sf[j] = shift[(long)b * C + cg * 8 + j];
      }
    }
    const T* xb = x + (long)b * H * W * ldx + cg * 8;
    float ah[8], ax[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ah[j] = 0.f; ax[j] = 0.f; }
    if (MODE == 0) {
      float f[8];
      load8<T>(xb + ((long)oy * W + ox) * ldx, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = f[j] * sc[j] + sf[j];
        ah[j] = act ? silu_t<T>(v) : v;
      }
    } else {
      constexpr int NT = MODE == 2 ? 4 : 2;
      int iy0, ix0;
      float wy[4], wx[4];
      if (MODE == 2) {
        iy0 = 2 * oy - 1; ix0 = 2 * ox - 1;
        wy[0] = wx[0] = 0.125f; wy[1] = wx[1] = 0.375f; wy[2] = wx[2] = 0.375f; wy[3] = wx[3] = 0.125f;
      } else {
        const int ay = oy & 1, axp = ox & 1;
        iy0 = (oy >> 1) - 1 + ay; ix0 = (ox >> 1) - 1 + axp;
        wy[0] = ay ? 0.75f : 0.25f; wy[1] = ay ? 0.25f : 0.75f;
        wx[0] = axp ? 0.75f : 0.25f; wx[1] = axp ? 0.25f : 0.75f;
        wy[2] = wy[3] = wx[2] = wx[3] = 0.f;
      }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int iy = iy0 + a;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
          const int ix = ix0 + c;
          if (ix < 0 || ix >= W) continue;
          const float wgt = wy[a] * wx[c];
          float f[8];
          load8<T>(xb + ((long)iy * W + ix) * ldx, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            ax[j] += wgt * f[j];
            if (AFFINE) {
              float v = f[j] * sc[j] + sf[j];
              v = act ? silu_t<T>(v) : v;
              ah[j] += wgt * v;
            }
          }
        }
      }
    }
    const long opix = ((long)b * Ho + oy) * Wo + ox;
    if (AFFINE) store8<T>(y + opix * ldy + cg * 8, ah);
    if (MODE != 0 && xr) store8<T>(xr + opix * ldxr + cg * 8, ax);
  }
}

// Block variants of the resampling modes: one thread = a 2 x 2 OUTPUT block x 8 channels, built separably (rows of
// horizontally filtered values, then the vertical taps).  SiLU(GN(.)) is evaluated 9 (up) / 36 (down) times per 4
// outputs instead of 16 / 64, and the loads shrink likewise.  Same arithmetic as gn_apply_kernel up to fp32
// summation order.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void gn_resample2x2_kernel(const T* __restrict__ x, int ldx,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int C, T* __restrict__ y,
                                                             int ldy, T* __restrict__ xr, int ldxr, int B, int H, int W,
                                                             int act) {
  constexpr int NR = MODE == 1 ? 3 : 6;  // input rows / columns feeding one 2 x 2 output block
  const int Hb = MODE == 1 ? H : H / 4, Wb = MODE == 1 ? W : W / 4;  // grid of output blocks
  const int Ho = MODE == 1 ? 2 * H : H / 2, Wo = MODE == 1 ? 2 * W : W / 2;
  const int ncg = C >> 3;
  const long total = (long)B * Hb * Wb * ncg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % ncg);
    long r = i / ncg;
    const int bx = (int)(r % Wb);
    r /= Wb;
    const int by = (int)(r % Hb);
    const int b = (int)(r / Hb);
    float sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = scale[(long)b * C + cg * 8 + j];
      sf[j] = shift[(long)b * C + cg * 8 + j];
    }
    const T* xb = x + (long)b * H * W * ldx + cg * 8;
    const int iy0 = MODE == 1 ? by - 1 : 4 * by - 1, ix0 = MODE == 1 ? bx - 1 : 4 * bx - 1;
    float oh[2][2][8], ox[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { oh[a][c][j] = 0.f; ox[a][c][j] = 0.f; }
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
      const int iy = iy0 + ri;
      if (iy < 0 || iy >= H) continue;
      float hh[2][8], hx[2][8];  // horizontally filtered activated / raw values for the two output columns
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { hh[c][j] = 0.f; hx[c][j] = 0.f; }
#pragma unroll
      for (int ci = 0; ci < NR; ++ci) {
        const int ix = ix0 + ci;
        // horizontal tap weights of input column ci for output columns 0 / 1
        const float w0 = MODE == 1 ? (ci == 0 ? 0.25f : (ci == 1 ? 0.75f : 0.f))
                                   : (ci == 0 || ci == 3 ? 0.125f : (ci == 1 || ci == 2 ? 0.375f : 0.f));
        const float w1 = MODE == 1 ? (ci == 1 ? 0.75f : (ci == 2 ? 0.25f : 0.f))
                                   : (ci == 2 || ci == 5 ? 0.125f : (ci == 3 || ci == 4 ? 0.375f : 0.f));
        if (ix < 0 || ix >= W) continue;
        float f[8];
        load8<T>(xb + ((long)iy * W + ix) * ldx, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = f[j] * sc[j] + sf[j];
          v = act ? silu_t<T>(v) : v;
          if (w0 != 0.f) { hh[0][j] = fmaf(w0, v, hh[0][j]); hx[0][j] = fmaf(w0, f[j], hx[0][j]); }
          if (w1 != 0.f) { hh[1][j] = fmaf(w1, v, hh[1][j]); hx[1][j] = fmaf(w1, f[j], hx[1][j]); }
        }
      }
      const float v0 = MODE == 1 ? (ri == 0 ? 0.25f : (ri == 1 ? 0.75f : 0.f))
                                 : (ri == 0 || ri == 3 ? 0.125f : (ri == 1 || ri == 2 ? 0.375f : 0.f));
      const float v1 = MODE == 1 ? (ri == 1 ? 0.75f : (ri == 2 ? 0.25f : 0.f))
                                 : (ri == 2 || ri == 5 ? 0.125f : (ri == 3 || ri == 4 ? 0.375f : 0.f));
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (v0 != 0.f) { oh[0][c][j] = fmaf(v0, hh[c][j], oh[0][c][j]); ox[0][c][j] = fmaf(v0, hx[c][j], ox[0][c][j]); }
          if (v1 != 0.f) { oh[1][c][j] = fmaf(v1, hh[c][j], oh[1][c][j]); ox[1][c][j] = fmaf(v1, hx[c][j], ox[1][c][j]); }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const long opix = ((long)b * Ho + 2 * by + a) * Wo + 2 * bx + c;
        store8<T>(y + opix * ldy + cg * 8, oh[a][c]);
        if (xr) store8<T>(xr + opix * ldxr + cg * 8, ox[a][c]);
      }
  }
}


// FIR x2 DOWN for the large levels (16-bit tensors): the 2 x 2-block kernel above walks 36 input vectors per thread (each input
// element activated 2.25 times, rows re-read by the vertical neighbours) with little memory-level parallelism: 98 us at 256^2
// against ~40 us of HBM time.  Here a thread owns TWO adjacent output columns x 8 channels and walks DOWN a strip of RS output
// rows: an input row is 6 vectors (its 4-tap windows overlap by 2: 3 activations per output pixel and row instead of 4), loaded
// one row ahead, filtered horizontally into the two columns, and added into the two output rows it belongs to (taps 1/8, 3/8
// into the newer, 3/8, 1/8 into the older, which is then complete and stored) — every input row of the strip is read once.
// Same arithmetic as the kernels above up to fp32 summation order (horizontal taps first, then vertical; zeros outside).
template <int RS>
__global__ __launch_bounds__(256) void gn_fir_down_strip_kernel(const bf16_t* __restrict__ x, int ldx,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int C,
                                                                bf16_t* __restrict__ y, int ldy, bf16_t* __restrict__ xr,
                                                                int ldxr, int B, int H, int W, int act) {
  const int Ho = H / 2, Wo = W / 2, ncg = C >> 3, ntx = Wo / 2, nst = (Ho + RS - 1) / RS;
  const long total = (long)B * nst * ntx * ncg;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % ncg);
  long r = i / ncg;
  const int tx = (int)(r % ntx);
  r /= ntx;
  const int st = (int)(r % nst), b = (int)(r / nst);
  float sc[8], sf[8];
  {
    const float4 s0 = *reinterpret_cast<const float4*>(scale + (long)b * C + cg * 8), s1 = *reinterpret_cast<const float4*>(scale + (long)b * C + cg * 8 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + (long)b * C + cg * 8), h1 = *reinterpret_cast<const float4*>(shift + (long)b * C + cg * 8 + 4);
    sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    sf[0] = h0.x; sf[1] = h0.y; sf[2] = h0.z; sf[3] = h0.w; sf[4] = h1.x; sf[5] = h1.y; sf[6] = h1.z; sf[7] = h1.w;
  }
  const bf16_t* xb = x + (long)b * H * W * ldx + cg * 8;
  const int ix0 = 4 * tx - 1, oy0 = st * RS, iy0 = 2 * oy0 - 1;
  bool cok[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) cok[c] = ix0 + c >= 0 && ix0 + c < W;
  auto load_row = [&](int iy, uint4 (&raw)[6]) __attribute__((always_inline)) {
    const bool rok = iy >= 0 && iy < H;
#pragma unroll
    for (int c = 0; c < 6; ++c)
      raw[c] = (rok && cok[c]) ? *reinterpret_cast<const uint4*>(xb + ((long)iy * W + ix0 + c) * ldx) : make_uint4(0, 0, 0, 0);
  };
  // horizontally filtered row for the two output columns: ha (activated) / hx (raw)
  auto hrow = [&](int iy, const uint4 (&raw)[6], float (&ha)[2][8], float (&hx)[2][8]) __attribute__((always_inline)) {
    const bool rok = iy >= 0 && iy < H;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) { ha[q][j] = 0.f; hx[q][j] = 0.f; }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      float f[8];
      f[0] = h_lo(raw[c].x); f[1] = h_hi(raw[c].x); f[2] = h_lo(raw[c].y); f[3] = h_hi(raw[c].y);
      f[4] = h_lo(raw[c].z); f[5] = h_hi(raw[c].z); f[6] = h_lo(raw[c].w); f[7] = h_hi(raw[c].w);
      const bool ok = rok && cok[c];  // zero padding: the ACTIVATED value of a pixel outside the image is 0 too
      const float w0 = c == 0 || c == 3 ? 0.125f : (c == 1 || c == 2 ? 0.375f : 0.f);  // tap of column c for output column 0
      const float w1 = c == 2 || c == 5 ? 0.125f : (c == 3 || c == 4 ? 0.375f : 0.f);  // ... for output column 1
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = fmaf(f[j], sc[j], sf[j]);
        v = act ? silu_t<bf16_t>(v) : v;
        v = ok ? v : 0.f;
        if (w0 != 0.f) { ha[0][j] = fmaf(w0, v, ha[0][j]); hx[0][j] = fmaf(w0, f[j], hx[0][j]); }
        if (w1 != 0.f) { ha[1][j] = fmaf(w1, v, ha[1][j]); hx[1][j] = fmaf(w1, f[j], hx[1][j]); }
      }
    }
  };
  float pa[2][8], px[2][8], qa[2][8], qx[2][8];  // P: the older output row (gets taps 2, 3), Q: the newer (taps 0, 1)
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) { pa[q][j] = 0.f; px[q][j] = 0.f; qa[q][j] = 0.f; qx[q][j] = 0.f; }
  uint4 cur[6], nxt[6];
  load_row(iy0, cur);
#pragma unroll
  for (int k = 0; k < 2 * RS + 2; ++k) {
    if (k + 1 < 2 * RS + 2) load_row(iy0 + k + 1, nxt);
    float ha[2][8], hx[2][8];
    hrow(iy0 + k, cur, ha, hx);
    const float wq = (k & 1) ? 0.375f : 0.125f, wp = (k & 1) ? 0.125f : 0.375f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        qa[q][j] = (k & 1) ? fmaf(wq, ha[q][j], qa[q][j]) : wq * ha[q][j];
        qx[q][j] = (k & 1) ? fmaf(wq, hx[q][j], qx[q][j]) : wq * hx[q][j];
        pa[q][j] = fmaf(wp, ha[q][j], pa[q][j]);
        px[q][j] = fmaf(wp, hx[q][j], px[q][j]);
      }
    if (k & 1) {  // the older row is complete
      const int oy = oy0 + (k >> 1) - 1;
      if (k >= 3 && oy < Ho) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const long opix = ((long)b * Ho + oy) * Wo + 2 * tx + q;
          store8<bf16_t>(y + opix * ldy + cg * 8, pa[q]);
          if (xr) store8<bf16_t>(xr + opix * ldxr + cg * 8, px[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) { pa[q][j] = qa[q][j]; px[q][j] = qx[q][j]; }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) cur[c] = nxt[c];
  }
}

#ifdef DS_HALF_F16
// FIR x2 DOWN, round 5 (half-precision build): row tiles through LDS, every input element loaded and activated ONCE.
// The strip kernel above is VALU-bound: a thread re-activates the two columns it shares with each neighbour (48 SiLU per
// 32 new elements) in fp32 arithmetic and keeps 6 + 6 vectors in flight with most waves spilling — 80 us at 256^2 for
// 201 MB (0.31 of the HBM roof), 32 us at 128^2 (0.20).  Here a block of 256 threads walks DOWN a strip of RS output rows
// of a tile of 16 output columns x 64 channels:
//   phase 1 (per input row): thread (column c < 32, channel octet cg) loads its 16 bytes (one row ahead), applies
//     GroupNorm + SiLU (fp32, one rounding to the storage type) and
//     writes the activated and the raw vector to the row's LDS buffer (34 columns: threads 0 .. 15 also take the halo
//     columns -1 and 32); zeros outside the image for BOTH tensors;
//   phase 2: threads 0 .. 127 filter the activated tensor, 128 .. 255 the raw one: thread (output column m < 16, octet cg)
//     reads the 4 taps (ds_read_b128), filters horizontally in fp32 (v_fma_mix) and adds the row into its two running
//     output rows (taps 1/8, 3/8 into the newer, 3/8, 1/8 into the older, which is then complete: one 16-byte store).
// One barrier per input row (two row buffers).  Same arithmetic as the kernels above up to one storage rounding of the
// activated value in front of the filter and fp32 summation order.
#ifndef FD_MINW
#define FD_MINW 1  // (tools/fir_ab.sh: minimum waves per SIMD the register allocation must leave room for)
#endif
template <int RS>
__global__ __launch_bounds__(256, FD_MINW) void gn_fir_down_tiled_kernel(const bf16_t* __restrict__ x, int ldx,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int C,
                                                                bf16_t* __restrict__ y, int ldy, bf16_t* __restrict__ xr,
                                                                int ldxr, int B, int H, int W, int act, int tiles_x, int strips) {
  constexpr int CP = 144, NCOL = 34, TB = NCOL * CP;  // column pitch (bytes), staged columns, bytes per tensor and row
  __shared__ __attribute__((aligned(16))) char sm[2][2][TB];
  const int tid = threadIdx.x, cg = tid & 7, col = tid >> 3;
  const int ncgb = C >> 6;
  int bid = blockIdx.x;
  const int cgb = bid % ncgb; bid /= ncgb;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int st = bid % strips;
  const int b = bid / strips;
  const int ch0 = cgb * 64 + cg * 8;
  const int Ho = H >> 1, Wo = W >> 1;
  float sc[8], sf[8];
  {
    const float4 s0 = *reinterpret_cast<const float4*>(scale + (long)b * C + ch0), s1 = *reinterpret_cast<const float4*>(scale + (long)b * C + ch0 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + (long)b * C + ch0), h1 = *reinterpret_cast<const float4*>(shift + (long)b * C + ch0 + 4);
    sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
    sf[0] = h0.x; sf[1] = h0.y; sf[2] = h0.z; sf[3] = h0.w; sf[4] = h1.x; sf[5] = h1.y; sf[6] = h1.z; sf[7] = h1.w;
  }
  const bf16_t* xb = x + (long)b * H * W * ldx + ch0;
  const int ix = 32 * tx + col;                       // this thread's input column (always inside the image: W % 32 == 0)
  const bool halo = tid < 16;                         // threads 0 .. 7: column -1 of the tile, 8 .. 15: column 32
  const int hix = 32 * tx + (tid < 8 ? -1 : 32);
  const bool hok = halo && hix >= 0 && hix < W;
  const int iy0 = 2 * st * RS - 1;
  constexpr int NR = 2 * RS + 2;
  auto load_row = [&](int iy, uint4& v, uint4& hv) __attribute__((always_inline)) {
    const bool rok = iy >= 0 && iy < H;
    v = rok ? *reinterpret_cast<const uint4*>(xb + ((long)iy * W + ix) * ldx) : make_uint4(0, 0, 0, 0);
    hv = (rok && hok) ? *reinterpret_cast<const uint4*>(xb + ((long)iy * W + hix) * ldx) : make_uint4(0, 0, 0, 0);
  };
  // GroupNorm + SiLU in fp32, ONE rounding to the storage type on the way to LDS.  (A first version ran the activation in
  // packed half precision like the register-weight convolution does: 65 -> 63 us, but the sampler's agreement with the fp32
  // engine fell from 3.3e-3 to 4.3e-3 relative RMS — profiles/experiments/README.md round 5; behind a 16-tap average the
  // activation's rounding is not hidden by a following storage rounding of the same size, it adds to it.)
  auto silu2 = [&](unsigned w, int d) __attribute__((always_inline)) {
    float z0 = fmaf(h_lo(w), sc[2 * d], sf[2 * d]), z1 = fmaf(h_hi(w), sc[2 * d + 1], sf[2 * d + 1]);
    if (act) { z0 = silu_t<bf16_t>(z0); z1 = silu_t<bf16_t>(z1); }
    return pack_h2(z0, z1);
  };
  auto activate = [&](const uint4& v, bool ok) __attribute__((always_inline)) {  // zero padding: the activated value outside the image is 0 too
    uint4 a;
    a.x = silu2(v.x, 0); a.y = silu2(v.y, 1); a.z = silu2(v.z, 2); a.w = silu2(v.w, 3);
    if (!ok) a = make_uint4(0, 0, 0, 0);
    return a;
  };
  // phase-2 role
  const int tensor = tid >> 7, ocol = (tid >> 3) & 15;
  bf16_t* const out = tensor ? xr : y;
  const int ldo = tensor ? ldxr : ldy;
  float P[8], Q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { P[j] = 0.f; Q[j] = 0.f; }
  // input rows in flight: three rows ahead of the one being filtered (a row's phases take ~500 cycles, an HBM load under load
  // more: one row ahead left every row waiting for its data).  Four named register sets, four rows per trip of the loop.
  auto do_row = [&](int k, const uint4& cur, const uint4& hcur, bool odd) __attribute__((always_inline)) {
    const int iy = iy0 + k;
    const bool rok = iy >= 0 && iy < H;
    char* rowb = &sm[odd][0][0];
    {
      const uint4 a = activate(cur, rok);
      *reinterpret_cast<uint4*>(rowb + (col + 1) * CP + cg * 16) = a;
      *reinterpret_cast<uint4*>(rowb + TB + (col + 1) * CP + cg * 16) = cur;
      if (halo) {
        const uint4 ha = activate(hcur, rok && hok);
        const int hc = tid < 8 ? 0 : NCOL - 1;
        *reinterpret_cast<uint4*>(rowb + hc * CP + cg * 16) = ha;
        *reinterpret_cast<uint4*>(rowb + TB + hc * CP + cg * 16) = hcur;
      }
    }
    __syncthreads();
    {
      const char* src = rowb + tensor * TB + (2 * ocol) * CP + cg * 16;
      float hsum[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) hsum[j] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + t * CP);
        const float w = (t == 0 || t == 3) ? 0.125f : 0.375f;
        hsum[0] = fmaf(w, h_lo(v.x), hsum[0]); hsum[1] = fmaf(w, h_hi(v.x), hsum[1]);
        hsum[2] = fmaf(w, h_lo(v.y), hsum[2]); hsum[3] = fmaf(w, h_hi(v.y), hsum[3]);
        hsum[4] = fmaf(w, h_lo(v.z), hsum[4]); hsum[5] = fmaf(w, h_hi(v.z), hsum[5]);
        hsum[6] = fmaf(w, h_lo(v.w), hsum[6]); hsum[7] = fmaf(w, h_hi(v.w), hsum[7]);
      }
      const float wq = odd ? 0.375f : 0.125f, wp = odd ? 0.125f : 0.375f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        Q[j] = odd ? fmaf(wq, hsum[j], Q[j]) : wq * hsum[j];
        P[j] = fmaf(wp, hsum[j], P[j]);
      }
      if (odd) {  // the older row is complete
        const int oy = st * RS + (k >> 1) - 1;
        if (k >= 3 && oy < Ho && out) {
          const long opix = ((long)b * Ho + oy) * Wo + 16 * tx + ocol;
          store8<bf16_t>(out + opix * ldo + ch0, P);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) P[j] = Q[j];
      }
    }
  };
  uint4 r0, r1, r2, r3, h0, h1, h2, h3;
  load_row(iy0, r0, h0);
  load_row(iy0 + 1, r1, h1);
  load_row(iy0 + 2, r2, h2);
  for (int k = 0; k < NR; k += 4) {
    if (k + 3 < NR) load_row(iy0 + k + 3, r3, h3);
    do_row(k, r0, h0, false);
    if (k + 1 >= NR) break;
    if (k + 4 < NR) load_row(iy0 + k + 4, r0, h0);
    do_row(k + 1, r1, h1, true);
    if (k + 2 >= NR) break;
    if (k + 5 < NR) load_row(iy0 + k + 5, r1, h1);
    do_row(k + 2, r2, h2, false);
    if (k + 3 >= NR) break;
    if (k + 6 < NR) load_row(iy0 + k + 6, r2, h2);
    do_row(k + 3, r3, h3, true);
  }
}
#endif

// Tiled FIR x2 UP for the large levels: the 2 x 2-block kernel above evaluates SiLU(GN(.)) 9 times per input element in
// its up mode, which makes it VALU-bound (91 us at 256^2 output against 60 us of HBM time).  Here a block stages one input
// tile — 8 channel groups x (4 + 2) x (8 + 2) pixels — in LDS ONCE: the activated value as fp32 and the raw value in the
// storage type; then one thread per (input pixel, channel group) builds its 2 x 2 outputs from the 3 x 3 neighbourhood in
// LDS.  Same arithmetic as the kernels above (horizontal taps first, then vertical, fp32 FMAs; zeros outside the image).
// 91 -> 57 us at 256^2, 46 -> 27 us at 128^2.  (The FIR-down mode of this design measured slower than the 2 x 2-block
// kernel: profiles/experiments/gn_resample_tiled_down_r02.hip.txt.)
#ifndef UP_MINW
#define UP_MINW 1
#endif
template <typename T>
__global__ __launch_bounds__(256, UP_MINW) void gn_resample_up_tiled_kernel(const T* __restrict__ x, int ldx,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, int C,
                                                                   T* __restrict__ y, int ldy, T* __restrict__ xr,
                                                                   int ldxr, int B, int H, int W, int act, int tiles_x,
                                                                   int tiles_y) {
  constexpr int TH_ = 4, TW_ = 8, CG = 8;                         // tile of input pixels, channel groups per block
  constexpr int IH = TH_ + 2, IW = TW_ + 2;                       // staged input tile (1 pixel of halo)
  constexpr int RAWB = 8 * (int)sizeof(T);                        // raw vector bytes
  constexpr int PITCH = 32 + RAWB;                                // per (pixel, group): 8 fp32 activated + raw
  extern __shared__ __attribute__((aligned(16))) char sm[];       // IH * IW * CG * PITCH bytes
  const int tid = threadIdx.x;
  const int ncgb = (C >> 3) / CG;                                 // channel-group blocks (C % 64 == 0)
  int bid = blockIdx.x;
  const int cgb = bid % ncgb; bid /= ncgb;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const int cg = tid & 7;                                         // this thread's channel group in both phases
  const int ch0 = (cgb * CG + cg) * 8;
  float sc[8], sf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = scale[(long)b * C + ch0 + j];
    sf[j] = shift[(long)b * C + ch0 + j];
  }
  // ---- stage: input pixel (iy0 + r, ix0 + c), r < IH, c < IW
  const int iy0 = ty * TH_ - 1, ix0 = tx * TW_ - 1;
  const T* xb = x + (long)b * H * W * ldx + ch0;
  for (int i = tid; i < IH * IW * CG; i += 256) {
    const int pix = i >> 3;
    const int r = pix / IW, c = pix - r * IW;
    const int iy = iy0 + r, ix = ix0 + c;
    char* dst = sm + (pix * CG + cg) * PITCH;
    float v[8];
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      const T* src = xb + ((long)iy * W + ix) * ldx;
      float f[8];
      load8<T>(src, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = f[j] * sc[j] + sf[j];
        v[j] = act ? silu_t<T>(u) : u;
      }
      *reinterpret_cast<uint4*>(dst + 32) = reinterpret_cast<const uint4*>(src)[0];
      if (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst + 48) = reinterpret_cast<const uint4*>(src)[1];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      *reinterpret_cast<uint4*>(dst + 32) = make_uint4(0, 0, 0, 0);
      if (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst + 48) = make_uint4(0, 0, 0, 0);
    }
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(dst + 16) = make_float4(v[4], v[5], v[6], v[7]);
  }
  __syncthreads();
  // ---- FIR: thread = (input pixel p of the 4 x 8 tile, channel group cg) -> outputs (2 iy + a, 2 ix + c)
  const int p = tid >> 3, py = p / TW_, px = p - py * TW_;
  const int iy = ty * TH_ + py, ix = tx * TW_ + px;
  if (iy >= H || ix >= W) return;
  float oh[2][2][8], ox[2][2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { oh[a][c][j] = 0.f; ox[a][c][j] = 0.f; }
#pragma unroll
  for (int ri = 0; ri < 3; ++ri) {
    float hh[2][8], hx[2][8];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { hh[c][j] = 0.f; hx[c][j] = 0.f; }
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      const float w0 = ci == 0 ? 0.25f : (ci == 1 ? 0.75f : 0.f), w1 = ci == 1 ? 0.75f : (ci == 2 ? 0.25f : 0.f);
      const char* src = sm + (((py + ri) * IW + px + ci) * CG + cg) * PITCH;
      const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 16);
      const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      float f[8];
      load8<T>(reinterpret_cast<const T*>(src + 32), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (w0 != 0.f) { hh[0][j] = fmaf(w0, a[j], hh[0][j]); hx[0][j] = fmaf(w0, f[j], hx[0][j]); }
        if (w1 != 0.f) { hh[1][j] = fmaf(w1, a[j], hh[1][j]); hx[1][j] = fmaf(w1, f[j], hx[1][j]); }
      }
    }
    const float v0 = ri == 0 ? 0.25f : (ri == 1 ? 0.75f : 0.f), v1 = ri == 1 ? 0.75f : (ri == 2 ? 0.25f : 0.f);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (v0 != 0.f) { oh[0][c][j] = fmaf(v0, hh[c][j], oh[0][c][j]); ox[0][c][j] = fmaf(v0, hx[c][j], ox[0][c][j]); }
        if (v1 != 0.f) { oh[1][c][j] = fmaf(v1, hh[c][j], oh[1][c][j]); ox[1][c][j] = fmaf(v1, hx[c][j], ox[1][c][j]); }
      }
  }
  const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const long opix = ((long)b * Ho + 2 * iy + a) * Wo + 2 * ix + c;
      store8<T>(y + opix * ldy + ch0, oh[a][c]);
      if (xr) store8<T>(xr + opix * ldxr + ch0, ox[a][c]);
    }
}

template <typename T>
static constexpr int gn_resample_up_tiled_lds() { return 6 * 10 * 8 * (32 + 8 * (int)sizeof(T)); }

template <typename T>
static int gn_apply_typed(const void* x, int ldx, const float* scale, const float* shift, int C, void* y, int ldy,
                          void* xr, int ldxr, int B, int H, int W, int act, int mode, hipStream_t st) {
  const int Ho = mode == 1 ? 2 * H : (mode == 2 ? H / 2 : H);
  const int Wo = mode == 1 ? 2 * W : (mode == 2 ? W / 2 : W);
  const long total = (long)B * Ho * Wo * (C >> 3);
  long nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (nb < 1) nb = 1;
  const bool aff = scale != nullptr;
  const long tot2 = (mode == 1 ? (long)B * H * W : (long)B * (H / 4) * (W / 4)) * (C >> 3);
  // 2 x 2 output blocks per thread: fewer SiLU evaluations and loads per output, but a thread of the down mode walks
  // 36 input vectors — on the small levels (a few blocks' worth of threads) that is a 20 us latency chain, and the
  // one-output-per-thread kernel with 4x the threads takes 6 - 11 us (at 131072 block-threads: 26 vs 30 us, the blocks win)
  // large levels, FIR up: one activation per input element (LDS-staged tiles of 8 channel groups)
  if (aff && mode == 1 && C % 64 == 0 && tot2 >= 262144) {
    const int th = cdiv(H, 4), tw = cdiv(W, 8);
    const long nblk = (long)B * th * tw * (C / 64);
    DS_FUNC_LDS_ONCE((gn_resample_up_tiled_kernel<T>), gn_resample_up_tiled_lds<T>());
    hipLaunchKernelGGL((gn_resample_up_tiled_kernel<T>), dim3((unsigned)nblk), dim3(256), gn_resample_up_tiled_lds<T>(), st,
                       (const T*)x, ldx, scale, shift, C, (T*)y, ldy, (T*)xr, ldxr, B, H, W, act, tw, th);
    DS_LAUNCH_CHECK();
    return 0;
  }
  if constexpr (sizeof(T) == 2) {
    // large levels, FIR down, 16-bit tensors: column pairs walking down strips of 8 output rows (every input row read once)
#if defined(DS_HALF_F16) && !defined(DS_NO_FIR_TILED)
    // (half-precision build) row tiles through LDS: one activation per input element, 16 output columns x 64 channels per block
    if (aff && mode == 2 && H % 2 == 0 && W % 32 == 0 && C % 64 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (!xr || ldxr % 8 == 0) &&
        (long)B * H * W * C >= (1l << 21)) {
      const int Ho_ = H / 2, tiles_x = W / 32, ncgb = C / 64;
      // strips of RS output rows: as long as the strips still give every CU ~2 blocks
#ifndef FD_RS_MAX
#define FD_RS_MAX 8  // (16: 99.9 -> 97.0 us at 256^2 C = 64 and 68.7 -> 62.3 at 128^2 C = 128 incl. the statistics passes; tools/fir_ab.sh)
#endif
      int rs = FD_RS_MAX;
      while (rs > 4 && (long)B * tiles_x * cdiv(Ho_, rs) * ncgb < 2 * ds_num_cus()) rs >>= 1;
      const int strips = cdiv(Ho_, rs);
      const unsigned nblk = (unsigned)((long)B * tiles_x * strips * ncgb);
#define FD(RS_) hipLaunchKernelGGL((gn_fir_down_tiled_kernel<RS_>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)x, ldx, scale, shift, C, \
                                   (bf16_t*)y, ldy, (bf16_t*)xr, ldxr, B, H, W, act, tiles_x, strips)
      if (rs == 16) FD(16); else if (rs == 8) FD(8); else FD(4);
#undef FD
      DS_LAUNCH_CHECK();
      return 0;
    }
#endif
    if (aff && mode == 2 && H % 2 == 0 && W % 4 == 0 && tot2 >= 131072) {
      const long per_row = (long)B * (W / 4) * (C >> 3);
      // strips of 8 output rows where that still gives every CU two blocks (256^2: 131072 threads); strips of 4 below (nf = 64 at
      // 128^2: 32768 threads of 8-row strips were 128 blocks, 63 us with the statistics passes; 4-row strips 50, 2-row 53, the
      // 2 x 2-block kernel 55)
      if (per_row * cdiv(H / 2, 8) >= 131072) {
        constexpr int RS = 8;
        const long tot3 = per_row * cdiv(H / 2, RS);
        hipLaunchKernelGGL((gn_fir_down_strip_kernel<RS>), dim3((unsigned)cdiv(tot3, 256)), dim3(256), 0, st, (const bf16_t*)x, ldx, scale,
                           shift, C, (bf16_t*)y, ldy, (bf16_t*)xr, ldxr, B, H, W, act);
      } else {
        constexpr int RS = 4;
        const long tot3 = per_row * cdiv(H / 2, RS);
        hipLaunchKernelGGL((gn_fir_down_strip_kernel<RS>), dim3((unsigned)cdiv(tot3, 256)), dim3(256), 0, st, (const bf16_t*)x, ldx, scale,
                           shift, C, (bf16_t*)y, ldy, (bf16_t*)xr, ldxr, B, H, W, act);
      }
      DS_LAUNCH_CHECK();
      return 0;
    }
  }
  if (aff && (mode == 1 || (mode == 2 && H % 4 == 0 && W % 4 == 0 && tot2 >= 131072))) {
    long nb2 = (tot2 + 255) / 256;
    if (nb2 > 16384) nb2 = 16384;
    if (nb2 < 1) nb2 = 1;
    if (mode == 1)
      hipLaunchKernelGGL((gn_resample2x2_kernel<T, 1>), dim3((unsigned)nb2), dim3(256), 0, st, (const T*)x, ldx, scale,
                         shift, C, (T*)y, ldy, (T*)xr, ldxr, B, H, W, act);
    else
      hipLaunchKernelGGL((gn_resample2x2_kernel<T, 2>), dim3((unsigned)nb2), dim3(256), 0, st, (const T*)x, ldx, scale,
                         shift, C, (T*)y, ldy, (T*)xr, ldxr, B, H, W, act);
    DS_LAUNCH_CHECK();
    return 0;
  }
#define GA(M, A)                                                                                                    \
  hipLaunchKernelGGL((gn_apply_kernel<T, M, A>), dim3((unsigned)nb), dim3(256), 0, st, (const T*)x, ldx, scale, shift, \
                     C, (T*)y, ldy, (T*)xr, ldxr, B, H, W, act)
  if (mode == 0) { if (aff) GA(0, true); else GA(0, false); }
  else if (mode == 1) { if (aff) GA(1, true); else GA(1, false); }
  else { if (aff) GA(2, true); else GA(2, false); }
#undef GA
  DS_LAUNCH_CHECK();
  return 0;
}

int ds_launch_gn_apply(const void* x, int ldx, const float* scale, const float* shift, int C, void* y, int ldy,
                       void* xr, int ldxr, int B, int H, int W, int act, int mode, int dtype, hipStream_t st) {
  DS_CHECK(C % 8 == 0, "gn_apply: C must be a multiple of 8");
  DS_CHECK(mode >= 0 && mode <= 2, "gn_apply: bad resample mode");
  DS_CHECK(mode != 2 || (H % 2 == 0 && W % 2 == 0), "gn_apply: FIR down needs even H, W");
  DS_CHECK(scale != nullptr || (mode != 0 && xr != nullptr), "gn_apply: nothing to do");
  if (dtype == DS_F32) return gn_apply_typed<float>(x, ldx, scale, shift, C, y, ldy, xr, ldxr, B, H, W, act, mode, st);
  return gn_apply_typed<bf16_t>(x, ldx, scale, shift, C, y, ldy, xr, ldxr, B, H, W, act, mode, st);
}

// ------------------------------------------------------------------ channel concat  cat([a, b], dim=C)
template <typename T>
__global__ __launch_bounds__(256) void concat_kernel(const T* __restrict__ a, int lda, int Ca, const T* __restrict__ b,
                                                     int ldb, int Cb, T* __restrict__ y, int ldy, long npix) {
  const int ncg = (Ca + Cb) >> 3, nca = Ca >> 3;
  const long total = npix * ncg;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int cg = (int)(i % ncg);
    const long p = i / ncg;
    const T* src = cg < nca ? a + p * lda + cg * 8 : b + p * ldb + (cg - nca) * 8;
    T* dst = y + p * ldy + cg * 8;
    if (sizeof(T) == 4) {
      reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(src)[0];
      reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(src)[1];
    } else {
      reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(src)[0];
    }
  }
}

int ds_launch_concat(const void* a, int lda, int Ca, const void* b, int ldb, int Cb, void* y, int ldy, long npix,
                     int dtype, hipStream_t st) {
  DS_CHECK(Ca % 8 == 0 && Cb % 8 == 0, "concat: channels must be multiples of 8");
  const long total = npix * ((Ca + Cb) >> 3);
  long nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  if (nb < 1) nb = 1;
  if (dtype == DS_F32)
    hipLaunchKernelGGL(concat_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)a, lda, Ca,
                       (const float*)b, ldb, Cb, (float*)y, ldy, npix);
  else
    hipLaunchKernelGGL(concat_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)a, lda, Ca,
                       (const bf16_t*)b, ldb, Cb, (bf16_t*)y, ldy, npix);
  DS_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ row softmax (one wave per row)
// y[r, j] = exp(x[r,j] - max) / sum for j < L; columns [L, ld) are written as 0 (K padding of P V).
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int L,
                                                      int ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + row * ld;
  T* yr = y + row * ld;
  float mx = -INFINITY;
  for (int j = lane; j < L; j += 64) mx = fmaxf(mx, Elt<T>::ld(xr + j));
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < L; j += 64) sum += exp_t<T>(Elt<T>::ld(xr + j) - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < ld; j += 64) {
    float v = 0.f;
    if (j < L) v = exp_t<T>(Elt<T>::ld(xr + j) - mx) * inv;
    Elt<T>::st(yr + j, v);
  }
}

int ds_launch_softmax(const void* x, void* y, long rows, int L, int ld, int dtype, hipStream_t st) {
  DS_CHECK(L > 0 && ld >= L, "softmax: bad sizes");
  const unsigned nb = (unsigned)((rows + 3) / 4);
  if (dtype == DS_F32)
    hipLaunchKernelGGL(softmax_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)x, (float*)y, rows, L, ld);
  else
    hipLaunchKernelGGL(softmax_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, rows, L, ld);
  DS_LAUNCH_CHECK();
  return 0;
}
