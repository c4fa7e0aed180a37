// Issue cost of VALU instructions for ONE wave on a SIMD (the register-weight kernel's situation): cycles per
// instruction of a long stream of independent instructions, s_memtime around it.  hipcc --offload-arch=gfx950 -O2
// tools/probes/valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X X X X X X X X X X X X X X X X
#define BODY(NAME, ASM)                                                                                   \
  __global__ void NAME(unsigned long long* out, float seed) {                                            \
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
    unsigned long long t0 = __builtin_readcyclecounter();                                                \
    for (int i = 0; i < 256; ++i) {                                                                      \
      REP16(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)          \
    }                                                                                                    \
    unsigned long long t1 = __builtin_readcyclecounter();                                                \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                                              \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[1] = 1;                                    \
  }
#define EIGHT(OP) OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"
#define EIGHT3(OP) OP " %0, %0, %1, %2\n" OP " %1, %1, %2, %3\n" OP " %2, %2, %3, %4\n" OP " %3, %3, %4, %5\n" OP " %4, %4, %5, %6\n" OP " %5, %5, %6, %7\n" OP " %6, %6, %7, %0\n" OP " %7, %7, %0, %1\n"
BODY(k_exp32, EIGHT("v_exp_f32"))
BODY(k_rcp32, EIGHT("v_rcp_f32"))
BODY(k_exp16, EIGHT("v_exp_f16"))
BODY(k_rcp16, EIGHT("v_rcp_f16"))
BODY(k_fma32, EIGHT3("v_fma_f32"))
BODY(k_pkfma16, EIGHT3("v_pk_fma_f16"))
BODY(k_mov, EIGHT("v_mov_b32"))
BODY(k_cvtpk, "v_cvt_pk_f16_f32 %0, %0, %1\nv_cvt_pk_f16_f32 %1, %1, %2\nv_cvt_pk_f16_f32 %2, %2, %3\nv_cvt_pk_f16_f32 %3, %3, %4\nv_cvt_pk_f16_f32 %4, %4, %5\nv_cvt_pk_f16_f32 %5, %5, %6\nv_cvt_pk_f16_f32 %6, %6, %7\nv_cvt_pk_f16_f32 %7, %7, %0\n")

int main() {
  unsigned long long* d;
  hipMalloc(&d, 16);
  struct { const char* n; void (*k)(unsigned long long*, float); } ks[] = {
      {"v_exp_f32", k_exp32}, {"v_rcp_f32", k_rcp32}, {"v_exp_f16", k_exp16}, {"v_rcp_f16", k_rcp16},
      {"v_fma_f32", k_fma32}, {"v_pk_fma_f16", k_pkfma16}, {"v_mov_b32", k_mov}, {"v_cvt_pk_f16_f32", k_cvtpk}};
  for (auto& e : ks) {
    for (int waves = 1; waves <= 2; ++waves) {  // 1 wave alone on its SIMD, then 2 waves per SIMD (512 threads / 4 SIMDs)
      unsigned long long h[2] = {0, 0};
      hipMemset(d, 0, 16);
      hipLaunchKernelGGL(e.k, dim3(1), dim3(waves == 1 ? 64 : 512), 0, 0, d, 0.5f);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      printf("%-18s %s: %.2f cycles per wave-instruction\n", e.n, waves == 1 ? "1 wave on the SIMD " : "2 waves per SIMD   ", (double)h[0] / (256.0 * 16 * 8));
    }
  }
  return 0;
}
