// How fast can every CU pull the SAME few hundred KB (a layer's weights) out of L2 into registers?  The register-weight
// kernel's prologue takes ~21k cycles for 295 KB per CU = 12.8 B/clk/CU; is that the rate of the L1 fill path or of this
// kernel's access pattern?  One block of 256 threads per CU; each wave issues NLD 16-byte-per-lane buffer loads (1 KB per
// instruction, contiguous) with all of them in flight, then consumes them.
//   mode 0: every block reads the same region (weights: L2 hits after the first touch per XCD)
//   mode 1: every block reads its own region (HBM streaming, for comparison)
//   mode 2: like 0 with the waves of a block reading the SAME 72 KB (L1 hits for three of four waves)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/l2_bcast.hip -o /tmp/l2_bcast && /tmp/l2_bcast
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr int NLD = 72;
__device__ unsigned long long g_cycles[4096];
template <int MODE, int AUX>
__global__ __launch_bounds__(256, 1) void probe(const char* base, unsigned bytes_per_block, unsigned* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* p = base + (MODE == 1 ? (size_t)blockIdx.x * bytes_per_block : 0);
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, bytes_per_block, 0x00020000);
  const unsigned voff = (MODE == 2 ? 0u : (unsigned)wave * NLD * 1024u) + lane * 16u;
  u32x4 v[NLD];
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < NLD; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, i * 1024, AUX);
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < NLD; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345678u) sink[tid] = acc;
  if (tid == 0) g_cycles[blockIdx.x] = t1 - t0;
}
template <int MODE, int AUX>
static void run(const char* name, char* buf, unsigned* sink, int blocks) {
  const unsigned bpb = 4 * NLD * 1024;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((probe<MODE, AUX>), dim3(blocks), dim3(256), 0, 0, buf, bpb, sink);
    hipDeviceSynchronize();
  }
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<MODE, AUX>), dim3(blocks), dim3(256), 0, 0, buf, bpb, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  static unsigned long long h[4096];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cycles), sizeof(unsigned long long) * blocks);
  double s = 0, mx = 0;
  for (int i = 0; i < blocks; ++i) { s += h[i]; if (h[i] > mx) mx = h[i]; }
  const double kb = (MODE == 2 ? NLD : 4 * NLD);
  printf("%-44s %d blocks: %7.0f cycles avg (max %7.0f) for %3.0f KB per CU = %5.1f B/clk/CU; launch %.1f us\n", name, blocks,
         s / blocks, mx, kb, kb * 1024 / (s / blocks), ms * 1e3);
}
int main() {
  char* buf; unsigned* sink;
  const size_t n = (size_t)256 * 4 * NLD * 1024;
  hipMalloc(&buf, n); hipMemset(buf, 1, n); hipMalloc(&sink, 4096);
  for (int blocks : {256, 64, 8}) {
    run<0, 0>("same region, default policy", buf, sink, blocks);
    run<0, 2>("same region, nt", buf, sink, blocks);
    run<0, 1>("same region, sc0", buf, sink, blocks);
    run<2, 0>("same region, 4 waves read the same 72 KB", buf, sink, blocks);
    run<1, 0>("own region (HBM)", buf, sink, blocks);
  }
  return 0;
}
