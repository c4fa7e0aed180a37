// How many filler instructions does a 32-cycle MFMA hide, and does it matter WHICH wave of the SIMD issues them?
// (round 5: the question behind a wave-specialised register-weight convolution.)
//   T1  one wave per SIMD (the register-weight kernel's situation): stream of {MFMA, n fillers of kind X}: cycles per MFMA.
//   T2  two waves per SIMD, SPECIALISED: waves 0-3 issue MFMAs only, waves 4-7 issue the fillers only (the SiLU mix or
//       one kind); cycles per MFMA of the matrix waves, cycles per instruction of the vector waves; with / without s_setprio.
//   T3  two waves per SIMD, SYMMETRIC: both waves {MFMA, n fillers} (the 8-wave experiment of round 3).
// Every test runs on all CUs at once (grid = 256 blocks, random-ish non-zero data) so that the clock is the loaded one;
// cycles are s_memtime ticks of wave 0 / wave 4 of block 0, plus the kernel's wall time.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_mix.hip -o /tmp/mfma_valu_mix && /tmp/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define MFMA "v_mfma_f32_32x32x16_f16"
constexpr int ITERS = 512;  // x 8 MFMAs per iteration

// filler kinds
enum { F_MOV = 0, F_FMA32, F_EXP32, F_PKFMA16, F_EXP16, F_PKMUL16, F_CVTPK, F_DSREAD, F_SILU32, F_SILU16, F_DOT2, F_NKIND };
static const char* kind_name[F_NKIND] = {"v_mov_b32", "v_fma_f32", "v_exp_f32", "v_pk_fma_f16", "v_exp_f16", "v_pk_mul_f16",
                                         "v_cvt_pk_f16_f32", "ds_read_b128", "silu f32 mix (fma,exp,add,rcp,mul / cvt_pk)",
                                         "silu packed-f16 mix (pk_fma, 2 exp16, pk_add, 2 rcp16, pk_mul)", "v_dot2_f32_f16"};

struct Regs { float v[12]; };  // independent filler registers (no chain shorter than 6 instructions)

template <int KIND>
__device__ inline void filler(Regs& r, int i, const char* lds) {
  float& a = r.v[i % 12];
  float& b = r.v[(i + 5) % 12];
  float& c = r.v[(i + 7) % 12];
  if constexpr (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
  else if constexpr (KIND == F_FMA32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
  else if constexpr (KIND == F_EXP32) asm volatile("v_exp_f32 %0, %1" : "=v"(a) : "v"(b));
  else if constexpr (KIND == F_PKFMA16) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
  else if constexpr (KIND == F_EXP16) asm volatile("v_exp_f16 %0, %1" : "=v"(a) : "v"(b));
  else if constexpr (KIND == F_PKMUL16) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
  else if constexpr (KIND == F_CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
  else if constexpr (KIND == F_DOT2) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
  else if constexpr (KIND == F_DSREAD) {
    u32x4 t;
    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(threadIdx.x & 63) * 16u + (unsigned)(i & 7) * 1024u));
    asm volatile("" :: "v"(t));
  } else if constexpr (KIND == F_SILU32) {
    // the register-weight kernel's activation, 11 instructions per dword (2 elements), one of them per call:
    // fma, fma, exp, exp, add, add, rcp, rcp, mul, mul, cvt_pk
    switch (i % 11) {
      case 0: case 1: asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c)); break;
      case 2: case 3: asm volatile("v_exp_f32 %0, %1" : "=v"(a) : "v"(b)); break;
      case 4: case 5: asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(a) : "v"(b)); break;
      case 6: case 7: asm volatile("v_rcp_f32 %0, %1" : "=v"(a) : "v"(b)); break;
      case 8: case 9: asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c)); break;
      default: asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c)); break;
    }
  } else if constexpr (KIND == F_SILU16) {
    // packed form, 7 instructions per dword: pk_fma, exp16 lo, exp16 hi (sdwa), pk_add, rcp16 lo, rcp16 hi (sdwa), pk_mul
    switch (i % 7) {
      case 0: asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c)); break;
      case 1: asm volatile("v_exp_f16 %0, %1" : "=v"(a) : "v"(b)); break;
      case 2: asm volatile("v_exp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a) : "v"(b)); break;
      case 3: asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c)); break;
      case 4: asm volatile("v_rcp_f16 %0, %1" : "=v"(a) : "v"(b)); break;
      case 5: asm volatile("v_rcp_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(a) : "v"(b)); break;
      default: asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c)); break;
    }
  }
}

__device__ inline unsigned simd_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return (v >> 4) & 3u;
}

// out[block 0]: [0] cycles of wave 0, [1] cycles of wave 4 (if any), [2..9] simd ids of waves 0..7
// MODE 0: every wave {8 x (MFMA + NF fillers)} per iteration (T1 with 4 waves, T3 with 8)
// MODE 1: waves 0-3 MFMAs only, waves 4-7 NF x 8 fillers per iteration (T2); PRIO: 0 none, 1 matrix waves s_setprio 1,
//         2 vector waves s_setprio 1
template <int KIND, int NF, int MODE, int NW, int PRIO>
__global__ __launch_bounds__(64 * NW, NW / 4) void mix_kernel(unsigned long long* out, const float* seed) {
  __shared__ __attribute__((aligned(16))) char lds[8192 + 1024];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < (8192 + 1024) / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed[i & 255];
  __syncthreads();
  f32x16 acc[4];
  for (int r = 0; r < 4; ++r)
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
  u32x4 wa, wb;
  for (int e = 0; e < 4; ++e) {
    wa[e] = __builtin_bit_cast(unsigned, seed[(threadIdx.x + e) & 255]) & 0x3bff3bffu;  // finite halves < 1
    wb[e] = __builtin_bit_cast(unsigned, seed[(threadIdx.x + 7 * e + 3) & 255]) & 0x3bff3bffu;
  }
  Regs r;
  for (int i = 0; i < 12; ++i) r.v[i] = seed[(threadIdx.x + i) & 255] * 0.5f;
  const bool matrix = MODE == 0 || wave < 4;
  const bool vector_ = MODE == 0 || wave >= 4;
  if (PRIO == 1 && MODE == 1 && wave < 4) __builtin_amdgcn_s_setprio(1);
  if (PRIO == 2 && MODE == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE == 0) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        asm volatile(MFMA " %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(wa), "v"(wb));
#pragma unroll
        for (int f = 0; f < NF; ++f) filler<KIND>(r, m * NF + f, lds);
      }
    }
  } else if (wave < 4) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m) asm volatile(MFMA " %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(wa), "v"(wb));
    }
  } else {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int f = 0; f < 8 * NF; ++f) filler<KIND>(r, f, lds);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int q = 0; q < 4; ++q)
    for (int e = 0; e < 16; ++e) s += acc[q][e];
  for (int i = 0; i < 12; ++i) s += r.v[i];
  if (s == 12345.678f) out[15] = 1;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
    if (wave == 0) out[0] = t1 - t0;
    if (wave == 4) out[1] = t1 - t0;
    out[2 + wave] = simd_id();
  }
  (void)matrix; (void)vector_;
}

template <int KIND, int NF, int MODE, int NW, int PRIO>
void run(unsigned long long* d, const float* seed, const char* label) {
  unsigned long long h[16];
  hipMemset(d, 0, sizeof(h));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto k = mix_kernel<KIND, NF, MODE, NW, PRIO>;
  hipLaunchKernelGGL(k, dim3(256), dim3(64 * NW), 0, 0, d, seed);  // warm-up
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k, dim3(256), dim3(64 * NW), 0, 0, d, seed);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const double nm = ITERS * 8.0;
  if (MODE == 0)
    printf("%-34s NF=%d  %-18s %7.1f cyc/MFMA   (%.2f cyc per issued instruction)   wall %.1f us  ~%.2f GHz\n", label, NF,
           kind_name[KIND], h[0] / nm, h[0] / nm / (1 + NF), ms * 1e3, h[0] / (ms * 1e6));
  else
    printf("%-34s NF=%d  %-18s matrix wave %7.1f cyc/MFMA | vector wave %6.2f cyc/instr (%.1f cyc per 8 NF)  wall %.1f us  simd of waves 0..7: %llu%llu%llu%llu %llu%llu%llu%llu\n",
           label, NF, kind_name[KIND], h[0] / nm, NF ? h[1] / (nm * NF) : 0.0, h[1] / (double)ITERS, ms * 1e3, h[2], h[3], h[4], h[5], h[6],
           h[7], h[8], h[9]);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

#define T1(K)                                                              \
  run<K, 0, 0, 4, 0>(d, seed, "T1 one wave/SIMD");                          \
  run<K, 2, 0, 4, 0>(d, seed, "T1 one wave/SIMD");                          \
  run<K, 4, 0, 4, 0>(d, seed, "T1 one wave/SIMD");                          \
  run<K, 5, 0, 4, 0>(d, seed, "T1 one wave/SIMD");                          \
  run<K, 6, 0, 4, 0>(d, seed, "T1 one wave/SIMD");                          \
  run<K, 8, 0, 4, 0>(d, seed, "T1 one wave/SIMD");
#define T2(K)                                                              \
  run<K, 3, 1, 8, 0>(d, seed, "T2 specialised, no prio");                   \
  run<K, 5, 1, 8, 0>(d, seed, "T2 specialised, no prio");                   \
  run<K, 6, 1, 8, 0>(d, seed, "T2 specialised, no prio");                   \
  run<K, 8, 1, 8, 0>(d, seed, "T2 specialised, no prio");                   \
  run<K, 12, 1, 8, 0>(d, seed, "T2 specialised, no prio");                  \
  run<K, 6, 1, 8, 1>(d, seed, "T2 specialised, matrix prio 1");             \
  run<K, 8, 1, 8, 1>(d, seed, "T2 specialised, matrix prio 1");             \
  run<K, 6, 1, 8, 2>(d, seed, "T2 specialised, vector prio 1");
#define T3(K)                                                              \
  run<K, 2, 0, 8, 0>(d, seed, "T3 two waves/SIMD symmetric");               \
  run<K, 4, 0, 8, 0>(d, seed, "T3 two waves/SIMD symmetric");               \
  run<K, 6, 0, 8, 0>(d, seed, "T3 two waves/SIMD symmetric");

int main() {
  unsigned long long* d;
  float* seed;
  hipMalloc(&d, 16 * 8);
  hipMalloc(&seed, 256 * 4);
  float hs[256];
  for (int i = 0; i < 256; ++i) hs[i] = 0.25f + 0.5f * ((i * 2654435761u) % 1000) / 1000.f;
  hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
  T1(F_MOV) T1(F_FMA32) T1(F_EXP32) T1(F_PKFMA16) T1(F_EXP16) T1(F_CVTPK) T1(F_DSREAD) T1(F_SILU32) T1(F_SILU16) T1(F_DOT2)
  printf("\n");
  T2(F_MOV) T2(F_SILU32) T2(F_SILU16) T2(F_DSREAD)
  printf("\n");
  T3(F_MOV) T3(F_SILU32) T3(F_SILU16)
  return 0;
}
