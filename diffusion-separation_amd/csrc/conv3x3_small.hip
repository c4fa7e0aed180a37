// conv3x3_small.hip — 3x3 convolution for the small-image levels of NCSN++ (16^2, 8^2, 4^2; bf16, gfx950).
//
// At these levels (ddpm_conv3x3 inside ResnetBlockBigGANpp, reference layers.py:141-156, layerspp.py:291-323) a launch
// is a few MFLOP per sample: with the 64-cout tile of conv_mfma.hip the grid is 32 - 128 blocks, each of which pulls
// 64 couts x 9 taps of weights through one CU in a chain of dependent stages (19 us per launch, 1 % MFMA use; 25k
// cycles per block of which 5k are address set-up).  The decomposition here is the opposite one — thin blocks, at
// most a few stages, everything in flight at once:
//
//   * a block owns 16 couts of one pixel tile of one sample — 8 rows x 16 columns (8 waves), 8 x 8 (8 waves, 4 of
//     them multiply) or 4 x 4 (4 waves, 1 multiplies) by image width: grid = tiles x Cout / 16 x B (128 - 256 blocks
//     for 128 couts at B = 16);
//   * K is walked in phases of PC = 128 (or 64) input channels of ONE source (x, then x2 of a concat view, then the raw
//     block input of the folded 1x1 skip convolution through the centre tap).  A phase stages the whole halo tile and
//     the block's [9][16][PC] weight slab through registers into LDS; TWO phases are in flight in two register sets,
//     so the first two phases (= the whole K of most launches) cost one memory round trip; GroupNorm affine + SiLU is
//     applied in registers on the way to LDS (scale / shift from materialised tables or, lazily, from the producers'
//     channel-sum accumulators: table built in LDS behind the first loads);
//   * v_mfma_f32_16x16x32_bf16 with A = weights (16 couts x 32 channels), B = 16 pixels of the tile: a lane ends up
//     with 4 consecutive couts of one pixel = one 8-byte store; wave w owns pixels 16 w .. 16 w + 15 of the tile;
//     fragment reads run one tap ahead of the MFMAs, two accumulator chains per output quad;
//   * epilogue in registers: bias (+ per-sample bias), residual, scale, bf16 rounding, GroupNorm statistics of the
//     output (DPP row sums, waves through LDS, integer atomics like every other conv epilogue).
//
// Measured (B = 16, 128 -> 128, inside the captured graph): 16^2 19.2 -> ~14 us, 8^2 18.8 -> ~11, 4^2 18.8 -> ~9.5 per
// launch, of which ~5 us is the floor of any kernel node; one batch alone 371 -> 350 ms, four in flight unchanged.
// tools/small_timing.sh (-DSM_TIMING) prints the per-phase cycles of a block.
//
// ESZ = 4 instantiates the same kernel for fp32 tensors in split mode (DIFFSEP_F32_SPLIT): 64-channel phases (the bytes of
// 128 bf16 channels), values split into hi / lo bf16 planes on the way to LDS, three MFMAs per k-block, fp32 epilogue
// (the split engine's <= 16-row levels: 31 -> ~14 us per launch, one batch alone 759 -> 702 ms).
//
// Same ConvArgs contract as ds_launch_conv (common.h); ds_conv_small_eligible says which launches come here.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#ifdef SM_TIMING  // profiling build only (tools/small_timing.sh): per-phase cycle totals of wave 0 of every block
__device__ unsigned long long g_sm_dbg[16];
#define ST_DECL unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime();
#define ST_MARK(i) { unsigned long long st_now = __builtin_readcyclecounter(); st_acc[i] += st_now - st_prev; st_prev = st_now; }
#define ST_WAIT asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define ST_FLUSH if (threadIdx.x == 0) { st_acc[11] = __builtin_amdgcn_s_memrealtime() - st_rt0; /* 100 MHz ticks */ for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_sm_dbg[i_], st_acc[i_]); atomicAdd(&g_sm_dbg[15], 1ull); }
extern "C" int diffsep_small_debug_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sm_dbg), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sm_dbg), z, sizeof(z)); }
  return 0;
}
#else
#define ST_DECL
#define ST_MARK(i)
#define ST_WAIT
#define ST_FLUSH
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr unsigned OOB = 0x80000000u;

__device__ inline __amdgpu_buffer_rsrc_t rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ inline uint4 ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ inline uint2 ld8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
  return make_uint2(v.x, v.y);
}
__device__ inline void st8(__amdgpu_buffer_rsrc_t r, unsigned voff, uint2 d) {
  const u32x2_t v = {d.x, d.y};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 0);
}

constexpr int GN_MAX = 512;         // channels of the lazy GroupNorm table

struct SmK {
  const void* x; long x_bs; int ldx;
  const void* x2; long x2_bs; int ldx2;
  int C1, Cin;
  const void* w; int w_chunked, w_shift;
  const void* sx; long sx_bs; int ldsx;
  const void* sx2; long sx2_bs; int ldsx2;
  int sC1, sCin;
  const void* sw; int sw_chunked, sw_shift;
  const float* gn_scale; const float* gn_shift;
  const long long* gn_acc1; const long long* gn_acc2; const float* gn_gamma; const float* gn_beta;
  int gn_cpg; float gn_inv_count; float gn_eps;
  const float* bias; const float* bias_b; int bias_b_ld;
  const void* res; long res_bs; int ldr;
  float out_scale;
  void* y; long y_bs; int ldy;
  long long* stats;
  int H, W, Cout, tiles_x;
};

// sum over the 16 lanes of a DPP row (every lane gets the total): rotations by 8 and 4, then the two quad permutations
template <int CTRL>
__device__ inline float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ inline float row16_sum(float v) {
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x4e>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0xb1>(v);   // quad_perm [1,0,3,2]
  return v;
}

// GN affine (+ SiLU) on 8 bf16 channels
template <bool ACT>
__device__ inline uint4 gn8(const uint4& u, const float* sc, const float* sh) {
  float f[8];
  f[0] = h_lo(u.x); f[1] = h_hi(u.x);
  f[2] = h_lo(u.y); f[3] = h_hi(u.y);
  f[4] = h_lo(u.z); f[5] = h_hi(u.z);
  f[6] = h_lo(u.w); f[7] = h_hi(u.w);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = f[j] * sc[j] + sh[j];
    f[j] = ACT ? silu_t<bf16_t>(v) : v;
  }
  uint4 o;
  o.x = pack_h2(f[0], f[1]);
  o.y = pack_h2(f[2], f[3]);
  o.z = pack_h2(f[4], f[5]);
  o.w = pack_h2(f[6], f[7]);
  return o;
}

// fp32 storage (DIFFSEP_F32_SPLIT): GN affine (+ SiLU) on 4 fp32 channels, and the hi / lo bf16 split of conv_mfma.hip
template <bool ACT>
__device__ inline uint4 gn4(const uint4& u, const float* sc, const float* sh) {
  float f[4] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = f[j] * sc[j] + sh[j];
    f[j] = ACT ? silu_t<float>(v) : v;
  }
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
__device__ inline void split4(const uint4& v, uint2& hi, uint2& lo) {
  const float f0 = __uint_as_float(v.x), f1 = __uint_as_float(v.y), f2 = __uint_as_float(v.z), f3 = __uint_as_float(v.w);
  hi.x = pack_bf16x2(f0, f1);
  hi.y = pack_bf16x2(f2, f3);
  lo.x = pack_bf16x2(f0 - bf_lo(hi.x), f1 - bf_hi(hi.x));
  lo.y = pack_bf16x2(f2 - bf_lo(hi.y), f3 - bf_hi(hi.y));
}

// tile shapes by image width: 16 columns x 8 rows on 8 waves, 8 x 8 on 8 waves (4 of them multiply: 512 threads keep the two
// staging register sets small), 4 x 4 (one MFMA group) on 4 waves
template <int GW> struct SmTile;
template <> struct SmTile<16> { static constexpr int THT = 8, NT = 512; };
template <> struct SmTile<8> { static constexpr int THT = 8, NT = 512; };
template <> struct SmTile<4> { static constexpr int THT = 4, NT = 256; };

// ESZ: storage bytes per element: 2 = bf16, 4 = fp32 with split (bf16x3) products — an LDS row is then two bf16 planes
// [PC hi][PC lo]
template <int GW, int PC, int NS, int ESZ>
struct SmGeom {
  static constexpr int THT = SmTile<GW>::THT, NT = SmTile<GW>::NT;
  static constexpr int HWS = GW + 2, HP = (THT + 2) * HWS;  // halo columns / pixels
  // LDS geometry of the halo tile: rows of HWSP pixels (HWS padded), PSTR bytes per pixel.  A fragment read of the
  // 16 x 16 x 32 MFMA is lane = (pixel l16 of the 16-pixel group, k-quarter q) -> (row * HWSP + column) * PSTR + 16 q, and
  // ds_read_b128 serves lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... in one pass each: 16 distinct 16-byte bank
  // slots per pass need PSTR / 16 = 2 (mod 4) and, where a group spans several tile rows (GW = 8: 2 rows, GW = 4: 4 rows), a
  // halo row of 16 / 12 pixels (exhaustive search over pitches and row lengths; rounds 2 - 3 had PSTR / 16 = 1 (mod 4) and
  // unpadded rows: 4.3 - 4.5 conflict cycles per LDS instruction in the counters).
  static constexpr int HWSP = GW == 16 ? HWS : (GW == 8 ? 16 : 12);
  static constexpr int KVE = 16 / ESZ;                       // elements per 16-byte vector
  static constexpr int NVEC = PC / KVE;                      // 16-byte vectors per pixel / weight row
  static constexpr int RPS = NT / NVEC;                      // rows staged by one pass of the block
  static constexpr int PSTR = PC * ESZ + 32;                 // LDS pitch of a halo pixel / of a (tap, cout) weight row
  static constexpr int NA = (HP + RPS - 1) / RPS;            // input vectors per thread and phase
  static constexpr int NWV = (9 * NS + RPS - 1) / RPS;       // weight vectors per thread and phase
  static constexpr int NGRP = THT * GW / 16;                 // 16-pixel groups of the tile (one per wave)
  static constexpr int NSK = (NS + RPS - 1) / RPS;           // staging passes of the folded skip convolution's weights
  static_assert(NGRP <= NT / 64 && NSK <= NWV && RPS % NS == 0, "one MFMA group per wave; a staging pass covers whole taps");
  static constexpr int LDS_IN = (THT + 2) * HWSP * PSTR, LDS_W = 9 * NS * PSTR;
  static constexpr int LDS = LDS_IN + LDS_W + 2 * GN_MAX * 4;
  static_assert(LDS_IN + LDS_W >= 2 * GN_MAX * 8, "the GroupNorm table is built in the staging area");
};

// GW: tile columns (16, 8, 4).  PC: channels per phase (64, 128).  NS: couts per block (16, 32).
// MODE: 0 raw input, 1 GroupNorm affine, 2 affine + SiLU.
template <int GW, int PC, int NS, int MODE, int ESZ = 2>
__global__ __launch_bounds__(SmTile<GW>::NT, 2) void conv3x3_small_kernel(SmK p) {
  using G = SmGeom<GW, PC, NS, ESZ>;
  constexpr int KVE = G::KVE;
  constexpr bool SPLIT = ESZ == 4;
  constexpr int NSK = G::NSK, NH = NS / 16;
  constexpr int THT = G::THT, NT = G::NT, NWAVES = NT / 64;
  constexpr int HWS = G::HWS, HWSP = G::HWSP, HP = G::HP, NVEC = G::NVEC, RPS = G::RPS, PSTR = G::PSTR, NA = G::NA, NWV = G::NWV,
                NGRP = G::NGRP;
  ST_DECL
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sIn = smem;
  char* sW = smem + G::LDS_IN;
  float* sGN = reinterpret_cast<float*>(smem + G::LDS_IN + G::LDS_W);  // [2][Cin] (lazy GroupNorm)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, q = lane >> 4;
  const int b = blockIdx.z;
  const int co0 = blockIdx.y * NS;
  const int y0 = (blockIdx.x / p.tiles_x) * THT, x0 = (blockIdx.x % p.tiles_x) * GW;
  const int M = p.H * p.W;

  // ---- staging geometry: vector i = tid + NT k is 16 bytes (8 channels, slot tid % NVEC) of halo pixel i / NVEC
  const int cv = tid % NVEC, row0 = tid / NVEC;
  int pixi[NA];    // linear pixel index in the image, or -1 (outside the image or past the halo)
  int ldsa[NA];    // LDS byte offset of the vector (halo rows are HWSP pixels long there)
  bool inner[NA];  // the pixel belongs to the tile proper (all the folded 1x1 convolution needs)
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const int pix = row0 + RPS * k;
    const int hy = pix / HWS, hx = pix - hy * HWS;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool ok = pix < HP && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    pixi[k] = ok ? gy * p.W + gx : -1;
    inner[k] = ok && hy >= 1 && hy <= THT && hx >= 1 && hx <= GW;
    ldsa[k] = (hy * HWSP + hx) * PSTR + cv * (SPLIT ? 8 : 16);
  }
  const int lds0 = row0 * PSTR + cv * (SPLIT ? 8 : 16);  // + RPS k PSTR (split: 8 bytes in each of the two planes)

  const int C1 = p.x2 ? p.C1 : p.Cin, C2 = p.Cin - C1;
  const int sC1 = p.sx2 ? p.sC1 : p.sCin, sC2 = p.sCin - sC1;
  const int nph1 = C1 / PC, nphc = nph1 + C2 / PC;                                    // conv phases
  const int nphs1 = p.sx ? sC1 / PC : 0, nph = nphc + nphs1 + (p.sx ? sC2 / PC : 0);  // + folded skip phases

  const __amdgpu_buffer_rsrc_t rw = rsrc(p.w, (unsigned)p.Cout * 9u * p.Cin * (unsigned)ESZ);
  const __amdgpu_buffer_rsrc_t rsw = rsrc(p.sw ? p.sw : p.w, p.sw ? (unsigned)p.Cout * p.sCin * (unsigned)ESZ : 0u);

  struct Stage {
    uint4 pa[NA], pw[NWV];
    float gsc[8], gsh[8];
    bool raw;  // a folded-skip phase: no activation, centre tap only
  };
  Stage s0, s1;

  auto issue = [&](int ph, Stage& S) __attribute__((always_inline)) {
    if (ph >= nphc) {  // folded 1x1 skip convolution: raw input, [Cout][sCin] weights into the centre tap's rows
      const int s = ph - nphc;
      const bool second = s >= nphs1;
      const int cb = (second ? s - nphs1 : s) * PC;
      const int wb = (second ? sC1 : 0) + cb + cv * KVE;  // channel in the skip weights
      const char* base = second ? (const char*)p.sx2 + (long)b * p.sx2_bs * ESZ : (const char*)p.sx + (long)b * p.sx_bs * ESZ;
      const int ld = second ? p.ldsx2 : p.ldsx;
      const __amdgpu_buffer_rsrc_t rs = rsrc(base, (unsigned)M * ld * (unsigned)ESZ);
#pragma unroll
      for (int k = 0; k < NA; ++k)
        S.pa[k] = ld16(rs, inner[k] ? (unsigned)((__umul24(pixi[k], ld) + cb + cv * KVE) * ESZ) : OOB, 0);
#pragma unroll
      for (int k = 0; k < NSK; ++k) {
        const int co = row0 + RPS * k;
        const unsigned vo = p.sw_chunked
                                ? (unsigned)((((wb >> p.sw_shift) * p.Cout + co0 + co) * p.sw_chunked + (wb & (p.sw_chunked - 1))) * ESZ)
                                : (unsigned)(((co0 + co) * p.sCin + wb) * ESZ);
        S.pw[k] = ld16(rsw, (co < NS && co0 + co < p.Cout) ? vo : OOB, 0);
      }
      S.raw = true;
      return;
    }
    const bool second = ph >= nph1;
    const int cb = (second ? ph - nph1 : ph) * PC;
    const int wb = (second ? C1 : 0) + cb + cv * KVE;  // channel in the weights / GroupNorm tables
    const char* base = second ? (const char*)p.x2 + (long)b * p.x2_bs * ESZ : (const char*)p.x + (long)b * p.x_bs * ESZ;
    const int ld = second ? p.ldx2 : p.ldx;
    const __amdgpu_buffer_rsrc_t rx = rsrc(base, (unsigned)M * ld * (unsigned)ESZ);
#pragma unroll
    for (int k = 0; k < NA; ++k)
      S.pa[k] = ld16(rx, pixi[k] >= 0 ? (unsigned)((__umul24(pixi[k], ld) + cb + cv * KVE) * ESZ) : OOB, 0);
    // weight row (tap, cout) = row0 + RPS k: RPS is a multiple of NS, so the cout stays and the tap advances by RPS / NS
    const int tap0 = row0 / NS, co = row0 % NS;
    const unsigned vo0 =
        p.w_chunked ? (unsigned)(((((wb >> p.w_shift) * 9 + tap0) * p.Cout + co0 + co) * p.w_chunked + (wb & (p.w_chunked - 1))) * ESZ)
                    : (unsigned)((((co0 + co) * 9 + tap0) * p.Cin + wb) * ESZ);
    const unsigned vstep = (unsigned)((RPS / NS) * (p.w_chunked ? p.Cout * p.w_chunked : p.Cin) * ESZ);
#pragma unroll
    for (int k = 0; k < NWV; ++k) S.pw[k] = ld16(rw, (row0 + RPS * k < 9 * NS && co0 + co < p.Cout) ? vo0 + k * vstep : OOB, 0);  // (couts past Cout — the <= 8-cout heads — multiply zeros)
    S.raw = false;
  };
  // GroupNorm scale / shift of the thread's 8 channels of conv phase ph (from the LDS table or the materialised arrays)
  auto fetch_gn = [&](int ph, Stage& S) __attribute__((always_inline)) {
    if (MODE == 0 || ph >= nphc) return;
    const bool second = ph >= nph1;
    const int wb = (second ? C1 + (ph - nph1) * PC : ph * PC) + cv * KVE;
    const float* ps = p.gn_acc1 ? sGN + wb : p.gn_scale + (long)b * p.Cin + wb;
    const float* ph_ = p.gn_acc1 ? sGN + p.Cin + wb : p.gn_shift + (long)b * p.Cin + wb;
#pragma unroll
    for (int j = 0; j < KVE; j += 4) {
      const float4 a0 = *reinterpret_cast<const float4*>(ps + j), h0 = *reinterpret_cast<const float4*>(ph_ + j);
      S.gsc[j] = a0.x; S.gsc[j + 1] = a0.y; S.gsc[j + 2] = a0.z; S.gsc[j + 3] = a0.w;
      S.gsh[j] = h0.x; S.gsh[j + 1] = h0.y; S.gsh[j + 2] = h0.z; S.gsh[j + 3] = h0.w;
    }
  };
  auto activate = [&](Stage& S) __attribute__((always_inline)) {  // zero padding keeps its loaded zeros
    if (MODE == 0 || S.raw) return;
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const uint4 r = SPLIT ? gn4<MODE == 2>(S.pa[k], S.gsc, S.gsh) : gn8<MODE == 2>(S.pa[k], S.gsc, S.gsh);
      const bool ok = pixi[k] >= 0;
      S.pa[k].x = ok ? r.x : S.pa[k].x;
      S.pa[k].y = ok ? r.y : S.pa[k].y;
      S.pa[k].z = ok ? r.z : S.pa[k].z;
      S.pa[k].w = ok ? r.w : S.pa[k].w;
    }
  };
  auto put = [&](char* dst, const uint4& v) __attribute__((always_inline)) {
    if constexpr (SPLIT) {  // row = [PC bf16 hi][PC bf16 lo]: this thread's 4 channels, 8 bytes in each plane
      uint2 hi, lo;
      split4(v, hi, lo);
      *reinterpret_cast<uint2*>(dst) = hi;
      *reinterpret_cast<uint2*>(dst + PC * 2) = lo;
    } else {
      *reinterpret_cast<uint4*>(dst) = v;
    }
  };
  auto write = [&](Stage& S) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NA; ++k)
      if (row0 + RPS * k < HP) put(sIn + ldsa[k], S.pa[k]);
    if (S.raw) {
#pragma unroll
      for (int k = 0; k < NSK; ++k)
        if (row0 + RPS * k < NS) put(sW + (4 * NS) * PSTR + lds0 + RPS * k * PSTR, S.pw[k]);
      return;
    }
#pragma unroll
    for (int k = 0; k < NWV; ++k)
      if (row0 + RPS * k < 9 * NS) put(sW + lds0 + RPS * k * PSTR, S.pw[k]);
  };

  // this lane's pixel of the tile: 16 w + l16 in raster order, couts co0 + 4 q .. + 3
  const int pl = wave * 16 + l16, ty = pl / GW, tx = pl % GW;
  const bool mma_wave = wave < NGRP;
  f32x4 acc[NH][2];  // two dependency chains per output quad (even / odd k-blocks)
#pragma unroll
  for (int h = 0; h < NH; ++h) { acc[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[h][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int woff = l16 * PSTR + q * 16;
  const int aoff = (ty * HWSP + tx) * PSTR + q * 16;
  // fragment reads run one tap ahead of the MFMAs (two register sets): the LDS latency is paid once per phase
  auto mma = [&](bool raw) __attribute__((always_inline)) {
    if (!mma_wave) return;
    constexpr int KB = PC / 32;
    constexpr int NP = SPLIT ? 2 : 1;  // bf16 planes of an LDS row (split: hi, lo)
    uint4 xf[2][NP][KB], wf[2][NP][NH][KB];
    auto frag = [&](int tap, int set) __attribute__((always_inline)) {
#pragma unroll
      for (int pl_ = 0; pl_ < NP; ++pl_)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          xf[set][pl_][kb] =
              *reinterpret_cast<const uint4*>(sIn + ((tap / 3) * HWSP + tap % 3) * PSTR + aoff + pl_ * PC * 2 + kb * 64);
#pragma unroll
          for (int h = 0; h < NH; ++h)
            wf[set][pl_][h][kb] = *reinterpret_cast<const uint4*>(sW + (tap * NS + 16 * h) * PSTR + woff + pl_ * PC * 2 + kb * 64);
        }
    };
    auto fma_tap = [&](int set) __attribute__((always_inline)) {
      if constexpr (SPLIT) {  // lo * hi, hi * lo, hi * hi (small terms first), term-major over the accumulators
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
          for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int h = 0; h < NH; ++h)
              acc[h][kb & 1] = mfma_bf16(wf[set][term == 0 ? 1 : 0][h][kb], xf[set][term == 1 ? 1 : 0][kb], acc[h][kb & 1]);
      } else {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int h = 0; h < NH; ++h)
            acc[h][kb & 1] = mfma_h16(wf[set][0][h][kb], xf[set][0][kb], acc[h][kb & 1]);
      }
    };
    if (raw) {  // folded 1x1 convolution: the centre tap only
      frag(4, 0);
      fma_tap(0);
      return;
    }
    frag(0, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) frag(tap + 1, (tap + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      fma_tap(tap & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: the first two phases in flight while the GroupNorm table is built
  ST_MARK(0)
  issue(0, s0);
  if (nph > 1) issue(1, s1);
  ST_MARK(1)
  if (MODE != 0 && p.gn_acc1) {
    long long* tmp = reinterpret_cast<long long*>(smem);  // [Cin][2]: the staging area is still unused
    for (int c = tid; c < p.Cin; c += NT) {
      const long long* src = c < C1 ? p.gn_acc1 + ((long)b * C1 + c) * 2 : p.gn_acc2 + ((long)b * C2 + (c - C1)) * 2;
      const longlong2 v = *reinterpret_cast<const longlong2*>(src);
      tmp[2 * c] = v.x;
      tmp[2 * c + 1] = v.y;
    }
    __syncthreads();
    const int cpg = p.gn_cpg;
    constexpr int NIT = GN_MAX / NT;
    float sc_[NIT], sh_[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + NT * it;
      sc_[it] = 0.f; sh_[it] = 0.f;
      if (c < p.Cin) {
        const int g0 = (int)(((float)c + 0.5f) / (float)cpg) * cpg;  // floor(c / cpg) * cpg, exact for c < 512
        long long ssum = 0, ssq = 0;
        for (int j = 0; j < cpg; ++j) {
          ssum += tmp[2 * (g0 + j)];
          ssq += tmp[2 * (g0 + j) + 1];
        }
        const double mean = (double)ssum * (1.0 / DS_STAT_SUM_SCALE) * (double)p.gn_inv_count;
        double var = (double)ssq * (1.0 / DS_STAT_SQ_SCALE) * (double)p.gn_inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
        sc_[it] = rstd * (p.gn_gamma ? p.gn_gamma[c] : 1.f);
        sh_[it] = (p.gn_beta ? p.gn_beta[c] : 0.f) - (float)mean * sc_[it];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + NT * it;
      if (c < p.Cin) { sGN[c] = sc_[it]; sGN[p.Cin + c] = sh_[it]; }
    }
    __syncthreads();
  }
  fetch_gn(0, s0);
  if (nph > 1) fetch_gn(1, s1);
  ST_MARK(2)

  // residual of this lane's output quad: in flight during the K loop
  const __amdgpu_buffer_rsrc_t rr = rsrc(p.res ? (const char*)p.res + (long)b * p.res_bs * ESZ : (const char*)p.y,
                                         p.res ? (unsigned)M * p.ldr * (unsigned)ESZ : 0u);
  const __amdgpu_buffer_rsrc_t ry = rsrc((char*)p.y + (long)b * p.y_bs * ESZ, (unsigned)M * p.ldy * (unsigned)ESZ);
  const int gy = y0 + ty, gx = x0 + tx;
  const int mo = (mma_wave && gy < p.H && gx < p.W) ? gy * p.W + gx : -1;
  const int cpad = (p.Cout + 7) & ~7;  // the output's (and residual's) channels incl. padding: quads past it are neither read nor written
  uint4 rres[NH];  // 4 couts: 8 bytes of bf16 or 16 bytes of fp32
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const unsigned ro = (mo >= 0 && co0 + 16 * h + 4 * q < cpad) ? (unsigned)((mo * p.ldr + co0 + 16 * h + 4 * q) * ESZ) : OOB;
    if constexpr (SPLIT) {
      rres[h] = ld16(rr, ro, 0);
    } else {
      const uint2 r2 = ld8(rr, ro);
      rres[h] = make_uint4(r2.x, r2.y, 0, 0);
    }
  }

  auto step = [&](int ph, Stage& S) __attribute__((always_inline)) {
    ST_WAIT
    ST_MARK(3)
    activate(S);
    ST_MARK(4)
    if (ph) __syncthreads();  // the previous phase's fragment reads are done
    write(S);
    const bool raw = S.raw;
    if (ph + 2 < nph) { issue(ph + 2, S); fetch_gn(ph + 2, S); }
    __syncthreads();
    ST_MARK(5)
    mma(raw);
    ST_MARK(6)
  };
  for (int ph = 0; ph < nph; ph += 2) {
    step(ph, s0);
    if (ph + 1 < nph) step(ph + 1, s1);
  }

  // ---- epilogue
  const float osc = p.out_scale;
  float v[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = co0 + 16 * h + 4 * q + i;
      const float bs = c < p.Cout ? ((p.bias ? p.bias[c] : 0.f) + (p.bias_b ? p.bias_b[(long)b * p.bias_b_ld + c] : 0.f)) * osc : 0.f;
      v[h][i] = fmaf(acc[h][0][i] + acc[h][1][i], osc, bs);
    }
    const unsigned yo = (mo >= 0 && co0 + 16 * h + 4 * q < cpad) ? (unsigned)((mo * p.ldy + co0 + 16 * h + 4 * q) * ESZ) : OOB;
    if constexpr (SPLIT) {
      v[h][0] = fmaf(__uint_as_float(rres[h].x), osc, v[h][0]);
      v[h][1] = fmaf(__uint_as_float(rres[h].y), osc, v[h][1]);
      v[h][2] = fmaf(__uint_as_float(rres[h].z), osc, v[h][2]);
      v[h][3] = fmaf(__uint_as_float(rres[h].w), osc, v[h][3]);
      const u32x4_t ov = {__float_as_uint(v[h][0]), __float_as_uint(v[h][1]), __float_as_uint(v[h][2]), __float_as_uint(v[h][3])};
      __builtin_amdgcn_raw_buffer_store_b128(ov, ry, yo, 0, 0);
    } else {
      v[h][0] = fmaf(h_lo(rres[h].x), osc, v[h][0]);
      v[h][1] = fmaf(h_hi(rres[h].x), osc, v[h][1]);
      v[h][2] = fmaf(h_lo(rres[h].y), osc, v[h][2]);
      v[h][3] = fmaf(h_hi(rres[h].y), osc, v[h][3]);
      st8(ry, yo, make_uint2(pack_h2(v[h][0], v[h][1]), pack_h2(v[h][2], v[h][3])));
    }
  }
  ST_MARK(7)
  if (p.stats) {
    const float keep = mo >= 0 ? 1.f : 0.f;
    float ssum[NH][4], ssq[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ssum[h][i] = row16_sum(keep * v[h][i]);
        ssq[h][i] = row16_sum(keep * v[h][i] * v[h][i]);
      }
    __syncthreads();  // the last phase's fragment reads are done: reuse the staging area
    float* sr = reinterpret_cast<float*>(smem);  // [waves][NS couts][2]
    if (l16 == 0) {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          sr[((wave * NS) + 16 * h + 4 * q + i) * 2] = ssum[h][i];
          sr[((wave * NS) + 16 * h + 4 * q + i) * 2 + 1] = ssq[h][i];
        }
    }
    __syncthreads();
    if (tid < NS) {
      double a = 0.0, s2 = 0.0;
      for (int w = 0; w < NWAVES; ++w) {
        a += (double)sr[(w * NS + tid) * 2];
        s2 += (double)sr[(w * NS + tid) * 2 + 1];
      }
      long long* o = p.stats + ((long)b * p.Cout + co0 + tid) * 2;
      ds_stat_add(o, (long long)llrint(a * DS_STAT_SUM_SCALE));
      ds_stat_add(o + 1, (long long)llrint(s2 * DS_STAT_SQ_SCALE));
    }
  }
  ST_MARK(8)
  ST_FLUSH
}

template <int GW, int PC, int NS, int MODE, int ESZ>
int launch_small(const SmK& k, const ConvArgs& a, hipStream_t st) {
  constexpr int LDS = SmGeom<GW, PC, NS, ESZ>::LDS;
  DS_FUNC_LDS_ONCE((conv3x3_small_kernel<GW, PC, NS, MODE, ESZ>), LDS);
  dim3 grid(k.tiles_x * cdiv(a.H, SmTile<GW>::THT), cdiv(a.Cout, NS), a.B);
  hipLaunchKernelGGL((conv3x3_small_kernel<GW, PC, NS, MODE, ESZ>), grid, dim3(SmTile<GW>::NT), LDS, st, k);
  {
    static char name[64] = {0};
    if (!name[0]) snprintf(name, sizeof(name), "conv3x3_small_kernel<%d,%d,%d,%d,%d>", GW, PC, NS, MODE, ESZ);
    ds_set_last_conv_kernel(name);
  }
  DS_LAUNCH_CHECK();
  return 0;
}
template <int GW, int PC, int NS, int ESZ>
int launch_small_mode(const SmK& k, const ConvArgs& a, int mode, hipStream_t st) {
  if (mode == 2) return launch_small<GW, PC, NS, 2, ESZ>(k, a, st);
  if (mode == 1) return launch_small<GW, PC, NS, 1, ESZ>(k, a, st);
  return launch_small<GW, PC, NS, 0, ESZ>(k, a, st);
}
template <int PC, int NS, int ESZ>
int launch_small_gw(const SmK& k, const ConvArgs& a, int gw, int mode, hipStream_t st) {
  if (gw == 4) return launch_small_mode<4, PC, NS, ESZ>(k, a, mode, st);
  if (gw == 8) return launch_small_mode<8, PC, NS, ESZ>(k, a, mode, st);
  return launch_small_mode<16, PC, NS, ESZ>(k, a, mode, st);
}

}  // namespace

// Small images (the launches ds_launch_conv would give to its 8 x 8 tile: W < 32 or H < 8; at most 16 rows), bf16,
// whole 64-channel phases per source, 16-cout slabs.
static bool small_sources_multiple_of(const ConvArgs& a, int pc) {
  const int C1 = a.x2 ? a.C1 : a.Cin;
  if (C1 % pc != 0 || (a.Cin - C1) % pc != 0) return false;
  if (a.sx) {
    const int sC1 = a.sx2 ? a.sC1 : a.sCin;
    if (sC1 % pc != 0 || (a.sCin - sC1) % pc != 0) return false;
  }
  return true;
}
bool ds_conv_small_eligible(const ConvArgs& a) {
  // bf16, or fp32 tensors in split mode (the exact fp32 engine keeps the generic kernel's fp32 MFMAs)
  if (!(a.dtype == DS_BF16 || (a.dtype == DS_F32 && a.split)) || a.taps != 9 || a.w_bs != 0 || a.bias_mode != 0 || a.div_b) return false;
  if (!(a.W < 32 || a.H < 8) || a.H > 16) return false;
  // whole 16-cout slabs, or the <= 8-cout heads of the output pyramid (one slab, couts past Cout masked: 14 - 17 us on the generic
  // 8 x 8 tile; no statistics, no folded skip there)
  const bool head = a.Cout <= 8 && !a.stats_acc && !a.sx && a.ldy >= 8 && (!a.res || a.ldr >= 8) && a.dtype == DS_BF16;
  if ((a.Cout % 16 != 0 && !head) || !small_sources_multiple_of(a, 64)) return false;
  if (a.ldx % 8 != 0 || (a.x2 && a.ldx2 % 8 != 0)) return false;
  if ((a.w_chunked & (a.w_chunked - 1)) || (a.w_chunked && a.w_chunked < 8)) return false;
  if (a.dtype == DS_F32 && ((a.w_chunked && a.w_chunked < 4) || (a.sx && a.sw_chunked && a.sw_chunked < 4))) return false;
  if (a.sx && (!a.sw || (a.sw_chunked & (a.sw_chunked - 1)) || (a.sw_chunked && a.sw_chunked < 8) || a.ldsx % 8 != 0 ||
               (a.sx2 && a.ldsx2 % 8 != 0)))
    return false;
  if (a.gn_acc1 && (a.Cin > GN_MAX || a.gn_groups <= 0 || a.Cin % a.gn_groups != 0 || (a.x2 && !a.gn_acc2))) return false;
  if (a.ldy % 4 != 0 || (a.res && a.ldr % 4 != 0)) return false;
  return true;
}

int ds_launch_conv_small(const ConvArgs& a, hipStream_t st) {
  SmK k;
  k.x = a.x; k.x_bs = a.x_bs; k.ldx = a.ldx;
  k.x2 = a.x2; k.x2_bs = a.x2_bs; k.ldx2 = a.ldx2;
  k.C1 = a.C1; k.Cin = a.Cin;
  k.w = a.w; k.w_chunked = a.w_chunked; k.w_shift = a.w_chunked ? __builtin_ctz(a.w_chunked) : 0;
  k.sx = a.sx; k.sx_bs = a.sx_bs; k.ldsx = a.ldsx;
  k.sx2 = a.sx2; k.sx2_bs = a.sx2_bs; k.ldsx2 = a.ldsx2;
  k.sC1 = a.sC1; k.sCin = a.sCin;
  k.sw = a.sw; k.sw_chunked = a.sw_chunked; k.sw_shift = a.sw_chunked ? __builtin_ctz(a.sw_chunked) : 0;
  k.gn_scale = a.gn_scale; k.gn_shift = a.gn_shift;
  k.gn_acc1 = a.gn_acc1; k.gn_acc2 = a.gn_acc2; k.gn_gamma = a.gn_gamma; k.gn_beta = a.gn_beta;
  k.gn_cpg = a.gn_acc1 ? a.Cin / a.gn_groups : 1; k.gn_inv_count = a.gn_inv_count; k.gn_eps = a.gn_eps;
  k.bias = a.bias; k.bias_b = a.bias_b; k.bias_b_ld = a.bias_b_ld;
  k.res = a.res; k.res_bs = a.res_bs; k.ldr = a.ldr;
  k.out_scale = a.out_scale;
  k.y = a.y; k.y_bs = a.y_bs; k.ldy = a.ldy;
  k.stats = a.stats_acc;
  const int gw = a.W <= 4 ? 4 : (a.W <= 8 ? 8 : 16);
  k.H = a.H; k.W = a.W; k.Cout = a.Cout; k.tiles_x = cdiv(a.W, gw);
  const int mode = (a.gn_scale || a.gn_acc1) ? (a.gn_act ? 2 : 1) : 0;
  // (32-cout slabs were measured slower: 406 vs 366 ms for one batch — twice the weight staging per block)
  if (a.dtype == DS_F32) return launch_small_gw<64, 16, 4>(k, a, gw, mode, st);  // 64 fp32 channels = the bytes of 128 bf16
  if (small_sources_multiple_of(a, 128)) return launch_small_gw<128, 16, 2>(k, a, gw, mode, st);
  return launch_small_gw<64, 16, 2>(k, a, gw, mode, st);
}
