// Pure memory-pattern probe: the global traffic of the 64 -> 64 3x3 conv (halo tile reads in 64-byte pieces, residual
// rows, output rows) without LDS / MFMA / activation.  hipcc --offload-arch=gfx950 -O3 tile_stream.hip -o tile_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* b, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(b), 0, bytes, 0x00020000);
}
// mode bit0: halo (10x34) instead of 8x32 reads; bit1: residual; bit2: full 128-byte pixel per load instruction
__global__ __launch_bounds__(256, 2) void probe(const unsigned short* x, const unsigned short* res, unsigned short* y,
                                                int H, int W, int mode) {
  const int tid = threadIdx.x, b = blockIdx.z;
  const int tiles_x = W / 32;
  const int y0 = (blockIdx.x / tiles_x) * 8, x0 = (blockIdx.x % tiles_x) * 32;
  const long img = (long)H * W * 64;
  const __amdgpu_buffer_rsrc_t rx = rsrc(x + b * img, (unsigned)(img * 2));
  const __amdgpu_buffer_rsrc_t rr = rsrc(res + b * img, (unsigned)(img * 2));
  const __amdgpu_buffer_rsrc_t ry = rsrc(y + b * img, (unsigned)(img * 2));
  u32x4_t acc = {0, 0, 0, 0};
  const bool halo = mode & 1, full = mode & 4;
  const int HWt = halo ? 34 : 32, HHt = halo ? 10 : 8, HP = HWt * HHt;
  if (!full) {
    for (int c = 0; c < 2; ++c) {  // two 32-channel chunks: 4 lanes per pixel
      u32x4_t v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int vi = tid + 256 * k, pix = vi >> 2, slot = vi & 3;
        const int hy = pix / HWt, hx = pix - hy * HWt;
        const int gy = y0 + hy - (halo ? 1 : 0), gx = x0 + hx - (halo ? 1 : 0);
        const bool ok = pix < HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(((gy * W + gx) * 64 + c * 32 + slot * 8) * 2) : 0x80000000u, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) acc ^= v[k];
    }
  } else {
    u32x4_t v[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) {  // whole pixels: 8 lanes per pixel
      const int vi = tid + 256 * k, pix = vi >> 3, slot = vi & 7;
      const int hy = pix / HWt, hx = pix - hy * HWt;
      const int gy = y0 + hy - (halo ? 1 : 0), gx = x0 + hx - (halo ? 1 : 0);
      const bool ok = pix < HP && gy >= 0 && gy < H && gx >= 0 && gx < W;
      v[k] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (unsigned)(((gy * W + gx) * 64 + slot * 8) * 2) : 0x80000000u, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) acc ^= v[k];
  }
  const int row = tid >> 3, cg = tid & 7;  // 32 rows of threads x 8 cout groups; 8 passes
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pp = row + it * 32, gy = y0 + pp / 32, gx = x0 + pp % 32;
    const unsigned o = (unsigned)(((gy * W + gx) * 64 + cg * 8) * 2);
    u32x4_t r = acc;
    if (mode & 2) r ^= __builtin_amdgcn_raw_buffer_load_b128(rr, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(r, ry, o, 0, 0);
  }
}
int main(int argc, char** argv) {
  const int B = 16, H = 256, W = 256;
  const size_t n = (size_t)B * H * W * 64;
  unsigned short *x, *r, *y;
  hipMalloc(&x, n * 2); hipMalloc(&r, n * 2); hipMalloc(&y, n * 2);
  hipMemset(x, 1, n * 2); hipMemset(r, 2, n * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 8; ++mode) {
    dim3 grid((H / 8) * (W / 32), 1, B);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, grid, dim3(256), 0, 0, x, r, y, H, W, mode);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe, grid, dim3(256), 0, 0, x, r, y, H, W, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n * 2 * (1.0 + 1.0 + ((mode & 2) ? 1.0 : 0.0));
    printf("mode %d (%s%s%s): %7.1f us  %5.2f TB/s algorithmic\n", mode, (mode & 1) ? "halo " : "", (mode & 2) ? "residual " : "",
           (mode & 4) ? "128B-pixels" : "64B-pieces", ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
  }
  return 0;
}
