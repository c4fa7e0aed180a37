// Probe: how fast does ONE wave per SIMD run the inner loop of conv3x3_rw.hip — v_mfma_f32_32x32x16_bf16 with the A
// operand (weights) in registers and the B operand (pixels) read from LDS by one ds_read_b128 per MFMA, RH MFMAs per
// weight fragment, reads issued DEPTH k-steps ahead?  No global traffic, no VALU: the ceiling of the loop structure.
// hipcc --offload-arch=gfx950 -O3 -mllvm -pragma-unroll-threshold=1000000 tools/probes/mfma_lds_loop.hip -o /tmp/mfma_lds_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int RH, int DEPTH, int PITCH, int NK>
__global__ __launch_bounds__(256, 1) void loop_kernel(const u32x4* __restrict__ w, float* __restrict__ y, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 40960 / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = w[i & 1023];
  u32x4 wf[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) wf[k] = w[(wave * NK + k) * 64 + lane];
  __syncthreads();
  const char* fb = smem + ((wave >> 1) * RH * 34 + (lane & 31)) * PITCH + (lane >> 5) * 16;
  f32x16 acc[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
  auto ldb = [&](int ks, int r) __attribute__((always_inline)) {
    const int tap = (ks / 2) % 9, kb = ks % 2;
    return *reinterpret_cast<const u32x4*>(fb + ((r + tap / 3) * 34 + tap % 3) * PITCH + kb * 32);
  };
  for (int it = 0; it < iters; ++it) {
    u32x4 bf[DEPTH][RH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int r = 0; r < RH; ++r) bf[d][r] = ldb(d, r);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, bf[ks % DEPTH][r]), acc[r], 0, 0, 0);
        if (ks + DEPTH < NK) bf[ks % DEPTH][r] = ldb(ks + DEPTH, r);
      }
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < RH; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) t += acc[r][e];
  y[blockIdx.x * 256 + tid] = t;
}

template <int RH, int DEPTH, int PITCH, int NK>
void run(const char* name, const u32x4* w, float* y) {
  auto k = loop_kernel<RH, DEPTH, PITCH, NK>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int iters = 64;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, w, y, iters);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 65536, 0, w, y, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double mf = 5.0 * iters * NK * RH;  // MFMAs per wave
  const double fl = mf * 32768.0 * 1024;    // x 1024 waves
  printf("%-40s %8.1f us/launch  %7.1f TF/s (%.3f of 2.5 PF)  %.1f ns per MFMA per wave\n", name, ms / 5 * 1e3, fl / (ms * 1e-3) / 1e12,
         fl / (ms * 1e-3) / 1e12 / 2500, ms * 1e6 / mf);
}

int main() {
  u32x4* w; float* y;
  hipMalloc(&w, 1 << 22); hipMalloc(&y, 1 << 20);
  hipMemset(w, 0x3c, 1 << 22);
  run<4, 2, 80, 36>("RH=4 depth 2 pitch 80", w, y);
  run<4, 3, 80, 36>("RH=4 depth 3 pitch 80", w, y);
  run<8, 1, 80, 36>("RH=8 depth 1 pitch 80", w, y);
  run<8, 2, 80, 36>("RH=8 depth 2 pitch 80", w, y);
  run<4, 2, 144, 36>("RH=4 depth 2 pitch 144", w, y);
  run<2, 4, 80, 36>("RH=2 depth 4 pitch 80", w, y);
  return 0;
}
