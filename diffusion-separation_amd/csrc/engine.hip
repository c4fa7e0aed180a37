// engine.hip — host side of the engine: architecture walk of NCSN++ (models/ncsnpp.py:106-308 ctor,
// :319-478 forward), parameter table in reference state_dict order, weight repack, the workspace
// arena, the per-NFE launch sequence (eager or hipGraph replay) and the PC sampler driver
// (sdes/__init__.py:166-188).  Everything that touches data is a HIP kernel from the sibling files;
// this file only sequences launches.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/diffsep_hip.h"
#include "common.h"

// ------------------------------------------------------------------ error string
static thread_local std::string g_err;
void ds_set_error(const std::string& s) { g_err = s; }
extern "C" const char* diffsep_last_error(void) { return g_err.c_str(); }
extern "C" const char* diffsep_version(void) {
    // which build of the library this is: 16-bit tensors stored as bfloat16 or (-DDS_HALF_F16) as IEEE half precision
#ifdef DS_HALF_F16
    return "diffsep-hip 0.3 (gfx950, 16-bit storage f16)";
#else
    return "diffsep-hip 0.3 (gfx950, 16-bit storage bf16)";
#endif
}

// ------------------------------------------------------------------ architecture description
struct PRef { long off = -1; long numel = 0; };  // into the flat fp32 blob
struct ParamInfo { std::string name; int ndim; int64_t shape[4]; long off; };

enum ModKind { MK_FOURIER, MK_LINEAR, MK_CONV3, MK_RES, MK_ATTN, MK_COMBINE, MK_GN };

struct Module {
  ModKind kind;
  int in_ch = 0, out_ch = 0;
  int in_c1 = 0;  // residual blocks of the up path read an in-place concat: channels of its first source (0 = none)
  bool up = false, down = false, has_conv2 = false;
  int temb_off = 0;  // offset of this block's Dense_0 output inside the concatenated projection
  // fp32 parameter references
  PRef w0, b0;        // Fourier W / Linear / Conv (3x3 or 1x1) / GN gamma,beta
  PRef gn0_w, gn0_b, conv0_w, conv0_b, dense_w, dense_b, gn1_w, gn1_b, conv1_w, conv1_b, conv2_w, conv2_b;
  PRef nin_w[4], nin_b[4];
  // packed (engine dtype) weight offsets in elements
  long pk0 = -1, pk1 = -1, pk2 = -1, pk_nin[4] = {-1, -1, -1, -1};
  // second copies of Conv_0 / Conv_1 / Conv_2 in the register-weight kernel's fragment-major order (16-bit engines, the shapes
  // that kernel takes: ds_rw_frag_shape), -1 = none
  long pf0 = -1, pf1 = -1, pf2 = -1;
  // split engines: the copies are hi / lo plane pairs (ds_sws_frag_index); pf_id = that copy of the identity matrix, which the
  // residual of a block without Conv_2 meets as a folded skip (conv3x3_sws.hip)
  long pf_id = -1;
  // cat(128, 128) -> 128 blocks whose convolutions run as two 128-channel launches (res_block): fragment-major copies of the
  // two halves of Conv_0, of the first half of Conv_2, and the second half of Conv_2 packed for a stand-alone 1x1 launch
  long pf0a = -1, pf0b = -1, pf2a = -1, pk2b = -1;
  // attention block, fused kernel (attn_fused.hip; 128 channels only): fragment-major copies [0] = Wk^T Wq (query and key
  // projections folded at engine creation), [2] = Wv, [3] = Wo; ab_off = this block's Wk^T b_q in the engine's d_attn_b
  long pf_nin[4] = {-1, -1, -1, -1};
  long ab_off = -1;
};

struct Arch {
  std::vector<Module> mods;
  std::vector<ParamInfo> params;
  PRef out_w, out_b;
  long pk_out = -1;
  long total = 0;       // floats in the blob
  long pack_total = 0;  // elements in the packed weight buffer
  int dense_total = 0;  // sum of out_ch over residual blocks
  long attn_bias_total = 0;  // floats of folded attention biases (Module::ab_off)
  int chan_in = 0, chan_out = 0, cpad_in = 0, cpad_out = 0;
};

static int rup8(int c) { return (c + 7) & ~7; }

struct ArchBuilder {
  Arch& A;
  // which fragment-major weight copies get a slot in the pack buffer: bit 0 = the shapes of the 16-bit kernels (conv3x3_rw / _sw,
  // the fused attention block), bit 1 = the shapes of the split-precision kernel (conv3x3_sws).  An exact-fp32 engine reads none
  // of them (0); the unit entry points build their one-module engines with every copy (3).
  int frag;
  explicit ArchBuilder(Arch& a, int frag_mask = 3) : A(a), frag(frag_mask) {}
  bool frag_wanted(int taps, int cin, int cout) const {
    return ((frag & 1) && (ds_rw_frag_shape(taps, cin, cout) || ds_sw_frag_shape(taps, cin, cout))) ||
           ((frag & 2) && ds_sws_frag_shape(taps, cin, cout));
  }
  PRef add(const std::string& name, std::initializer_list<int64_t> shp) {
    ParamInfo p;
    p.name = name;
    p.ndim = (int)shp.size();
    long n = 1;
    int i = 0;
    for (auto s : shp) { p.shape[i++] = s; n *= s; }
    for (; i < 4; ++i) p.shape[i] = 1;
    p.off = A.total;
    A.params.push_back(p);
    PRef r;
    r.off = A.total;
    r.numel = n;
    A.total += n;
    return r;
  }
  long pack(long o, int taps, int cin) {
    long r = A.pack_total;
    A.pack_total += o * taps * (long)rup8(cin);
    A.pack_total = (A.pack_total + 63) & ~63L;
    return r;
  }
  std::string pfx() const { return "all_modules." + std::to_string(A.mods.size()) + "."; }
  void fourier(int nf) {
    Module m; m.kind = MK_FOURIER; m.out_ch = nf;
    m.w0 = add(pfx() + "W", {nf});
    A.mods.push_back(m);
  }
  void linear(int in, int out) {
    Module m; m.kind = MK_LINEAR; m.in_ch = in; m.out_ch = out;
    m.w0 = add(pfx() + "weight", {out, in});
    m.b0 = add(pfx() + "bias", {out});
    A.mods.push_back(m);
  }
  void conv3(int in, int out) {
    Module m; m.kind = MK_CONV3; m.in_ch = in; m.out_ch = out;
    m.w0 = add(pfx() + "weight", {out, in, 3, 3});
    m.b0 = add(pfx() + "bias", {out});
    m.pk0 = pack(out, 9, in);
    A.mods.push_back(m);
  }
  void gn(int c) {
    Module m; m.kind = MK_GN; m.in_ch = m.out_ch = c;
    m.w0 = add(pfx() + "weight", {c});
    m.b0 = add(pfx() + "bias", {c});
    A.mods.push_back(m);
  }
  void res(int in, int out, bool up, bool down, int temb_dim, int in_c1 = 0) {
    Module m; m.kind = MK_RES; m.in_ch = in; m.out_ch = out; m.up = up; m.down = down; m.in_c1 = in_c1;
    const std::string p = pfx();
    m.gn0_w = add(p + "GroupNorm_0.weight", {in});
    m.gn0_b = add(p + "GroupNorm_0.bias", {in});
    m.conv0_w = add(p + "Conv_0.weight", {out, in, 3, 3});
    m.conv0_b = add(p + "Conv_0.bias", {out});
    m.dense_w = add(p + "Dense_0.weight", {out, temb_dim});
    m.dense_b = add(p + "Dense_0.bias", {out});
    m.gn1_w = add(p + "GroupNorm_1.weight", {out});
    m.gn1_b = add(p + "GroupNorm_1.bias", {out});
    m.conv1_w = add(p + "Conv_1.weight", {out, out, 3, 3});
    m.conv1_b = add(p + "Conv_1.bias", {out});
    m.has_conv2 = (in != out) || up || down;
    if (m.has_conv2) {
      m.conv2_w = add(p + "Conv_2.weight", {out, in, 1, 1});
      m.conv2_b = add(p + "Conv_2.bias", {out});
      m.pk2 = pack(out, 1, in);
    }
    m.pk0 = pack(out, 9, in);
    m.pk1 = pack(out, 9, out);
    if (frag_wanted(9, in, out)) m.pf0 = pack(out, 9, in);
    if (frag_wanted(9, out, out)) m.pf1 = pack(out, 9, out);
    if (m.has_conv2 && frag_wanted(1, in, out)) m.pf2 = pack(out, 1, in);
    if (!m.has_conv2 && (((frag & 2) && ds_sws_frag_shape(1, out, out)) || ((frag & 1) && ds_sw_frag_shape(1, out, out))))
      m.pf_id = pack(out, 1, out);
    if ((frag & 1) && in == 256 && in_c1 == 128 && out == 128 && !up && !down) {
      m.pf0a = pack(out, 9, 128); m.pf0b = pack(out, 9, 128); m.pf2a = pack(out, 1, 128); m.pk2b = pack(out, 1, 128);
    }
    m.temb_off = A.dense_total;
    A.dense_total += out;
    A.mods.push_back(m);
  }
  void attn(int c) {
    Module m; m.kind = MK_ATTN; m.in_ch = m.out_ch = c;
    const std::string p = pfx();
    m.gn0_w = add(p + "GroupNorm_0.weight", {c});
    m.gn0_b = add(p + "GroupNorm_0.bias", {c});
    for (int i = 0; i < 4; ++i) {
      m.nin_w[i] = add(p + "NIN_" + std::to_string(i) + ".W", {c, c});
      m.nin_b[i] = add(p + "NIN_" + std::to_string(i) + ".b", {c});
      m.pk_nin[i] = pack(c, 1, c);
      if ((frag & 1) && c == 128 && i != 1) m.pf_nin[i] = pack(c, 1, c);
    }
    if (c == 128) { m.ab_off = A.attn_bias_total; A.attn_bias_total += c; }
    A.mods.push_back(m);
  }
  void combine(int d1, int d2) {
    Module m; m.kind = MK_COMBINE; m.in_ch = d1; m.out_ch = d2;
    const std::string p = pfx();
    m.w0 = add(p + "Conv_0.weight", {d2, d1, 1, 1});
    m.b0 = add(p + "Conv_0.bias", {d2});
    m.pk0 = pack(d2, 1, d1);
    A.mods.push_back(m);
  }
};

static int build_arch(const diffsep_model_config& c, Arch& A, int frag_mask = 3) {
  DS_CHECK(c.nf >= 8 && c.nf % 8 == 0, "config: nf must be a positive multiple of 8");
  DS_CHECK(c.num_sources >= 1 && c.num_sources <= 3, "config: num_sources must be 1..3");
  DS_CHECK(c.n_levels >= 1 && c.n_levels <= 8, "config: n_levels must be 1..8");
  DS_CHECK(c.num_res_blocks >= 1, "config: num_res_blocks");
  DS_CHECK(c.n_fft % 2 == 0 && c.n_fft <= 512 && c.hop > 0, "config: n_fft must be even and <= 512");
  A = Arch();
  const int nf = c.nf, channels = 2 * c.num_sources + 2;
  A.chan_in = channels;
  A.chan_out = 2 * c.num_sources;
  A.cpad_in = rup8(channels);
  A.cpad_out = rup8(A.chan_out);
  const int image_size = c.n_fft / 2 + 1;
  ArchBuilder b(A, frag_mask);
  // state_dict order: output_layer is registered before all_modules (ncsnpp.py:104-105 vs :308)
  A.out_w = b.add("output_layer.weight", {A.chan_out, channels, 1, 1});
  A.out_b = b.add("output_layer.bias", {A.chan_out});
  A.pk_out = b.pack(A.chan_out, 1, channels);
  b.fourier(nf);
  b.linear(2 * nf, 4 * nf);
  b.linear(4 * nf, 4 * nf);
  b.conv3(channels, nf);
  std::vector<int> hs_c{nf};
  int in_ch = nf;
  const int L = c.n_levels;
  for (int i = 0; i < L; ++i) {
    const int resl = image_size >> i;
    for (int k = 0; k < c.num_res_blocks; ++k) {
      const int out_ch = nf * c.ch_mult[i];
      b.res(in_ch, out_ch, false, false, 4 * nf);
      in_ch = out_ch;
      if (resl == c.attn_resolution) b.attn(in_ch);
      hs_c.push_back(in_ch);
    }
    if (i != L - 1) {
      b.res(in_ch, in_ch, false, true, 4 * nf);
      b.combine(channels, in_ch);
      hs_c.push_back(in_ch);
    }
  }
  in_ch = hs_c.back();
  b.res(in_ch, in_ch, false, false, 4 * nf);
  b.attn(in_ch);
  b.res(in_ch, in_ch, false, false, 4 * nf);
  for (int i = L - 1; i >= 0; --i) {
    const int resl = image_size >> i;
    for (int k = 0; k < c.num_res_blocks + 1; ++k) {
      const int out_ch = nf * c.ch_mult[i];
      const int skip = hs_c.back();
      hs_c.pop_back();
      b.res(in_ch + skip, out_ch, false, false, 4 * nf, in_ch);
      in_ch = out_ch;
    }
    if (resl == c.attn_resolution) b.attn(in_ch);
    b.gn(in_ch);
    b.conv3(in_ch, channels);
    if (i != 0) b.res(in_ch, in_ch, true, false, 4 * nf);
  }
  DS_CHECK(hs_c.empty(), "internal: skip stack not empty");
  return 0;
}

static diffsep_model_config g_tmp_cfg;
static Arch g_tmp_arch;
static bool g_tmp_valid = false;
static int cached_arch(const diffsep_model_config* cfg, Arch** out) {
  DS_CHECK(cfg != nullptr, "null config");
  if (!g_tmp_valid || memcmp(&g_tmp_cfg, cfg, sizeof(*cfg)) != 0) {
    g_tmp_valid = false;
    if (build_arch(*cfg, g_tmp_arch)) return 1;
    g_tmp_cfg = *cfg;
    g_tmp_valid = true;
  }
  *out = &g_tmp_arch;
  return 0;
}

extern "C" int32_t diffsep_param_count(const diffsep_model_config* cfg) {
  Arch* a;
  if (cached_arch(cfg, &a)) return -1;
  return (int32_t)a->params.size();
}
extern "C" int64_t diffsep_param_total(const diffsep_model_config* cfg) {
  Arch* a;
  if (cached_arch(cfg, &a)) return -1;
  return a->total;
}
extern "C" int32_t diffsep_param_info(const diffsep_model_config* cfg, int32_t idx, char* name, int32_t name_cap,
                                      int64_t shape[4], int32_t* ndim, int64_t* offset) {
  Arch* a;
  if (cached_arch(cfg, &a)) return 1;
  DS_CHECK(idx >= 0 && idx < (int)a->params.size(), "param index out of range");
  const ParamInfo& p = a->params[idx];
  if (name && name_cap > 0) {
    strncpy(name, p.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  if (ndim) *ndim = p.ndim;
  if (offset) *offset = p.off;
  return 0;
}
extern "C" int32_t diffsep_num_frames(const diffsep_model_config* cfg, int64_t T) {
  return 1 + (int32_t)((T + cfg->n_fft - cfg->hop) / cfg->hop);
}
extern "C" int32_t diffsep_padded_frames(const diffsep_model_config* cfg, int64_t T) {
  const int F = diffsep_num_frames(cfg, T);
  return 64 * ((F + 63) / 64);
}

// ------------------------------------------------------------------ weight repack kernel
// dst[o][tap][i] (i < Ipad, zero padded) = src[o*so + i*si + tap*st]; kc > 0: chunk-major dst[i / kc][tap][o][i % kc]
// (one K stage of the conv kernel contiguous in memory -> whole 128-byte lines per request)
template <typename T>
__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ src, T* __restrict__ dst, int O, int I,
                                                     int Ipad, int taps, long so, long si, long st, int kc) {
  const long total = (long)O * taps * Ipad;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx % Ipad);
    const long r = idx / Ipad;
    const int tap = (int)(r % taps);
    const int o = (int)(r / taps);
    const float v = (i < I) ? src[o * so + i * si + tap * st] : 0.f;
    const long d = kc ? ((((long)(i / kc) * taps + tap) * O + o) * kc + i % kc) : idx;
    Elt<T>::st(dst + d, v);
  }
}
// dst[ds_rw_frag_index(o, tap, i)] = src[o*so + i*si + tap*st]: the register-weight kernel's fragment-major copy (16-bit only)
__global__ __launch_bounds__(256) void repack_frag_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int O, int I,
                                                          int taps, long so, long si, long st) {
  const long total = (long)O * taps * I;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx % I);
    const long r = idx / I;
    const int tap = (int)(r % taps), o = (int)(r / taps);
    dst[ds_rw_frag_index(o, tap, i, taps, O)] = f2h(src ? src[o * so + i * si + tap * st] : (o == i ? 1.f : 0.f));  // (null: the identity)
  }
}
// ... and the split mode's: hi = bf16(w), lo = bf16(w - hi) at ds_sws_frag_index(o, tap, i, plane); src == null: the O x O identity
__global__ __launch_bounds__(256) void repack_frag_split_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int O, int I,
                                                                int taps, long so, long si, long st) {
  const long total = (long)O * taps * I;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx % I);
    const long r = idx / I;
    const int tap = (int)(r % taps), o = (int)(r / taps);
    const float w = src ? src[o * so + i * si + tap * st] : (o == i ? 1.f : 0.f);
    const uint32_t hi = pack_bf16x2(w, 0.f);
    const uint32_t lo = pack_bf16x2(w - bf_lo(hi), 0.f);
    dst[ds_sws_frag_index(o, tap, i, taps, O, 0)] = (bf16_t)(hi & 0xffffu);
    dst[ds_sws_frag_index(o, tap, i, taps, O, 1)] = (bf16_t)(lo & 0xffffu);
  }
}
// Fused attention block: M[c'][k] = sum_c Wk[c'][c] Wq[k][c] (NIN.W is [in][out]: Wq^T applied to h gives q) in fragment-major
// order, and b'[c'] = sum_c Wk[c'][c] b_q[c] — fp32 sums, one rounding to the storage type (attn_fused.hip)
__global__ __launch_bounds__(256) void attn_fold_qk_kernel(const float* __restrict__ wq, const float* __restrict__ wk,
                                                           const float* __restrict__ bq, bf16_t* __restrict__ m_frag,
                                                           float* __restrict__ b_fold, int Cc) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cc * Cc) return;
  const int cp = idx / Cc, k = idx % Cc;
  float a = 0.f;
  for (int c = 0; c < Cc; ++c) a = fmaf(wk[(long)cp * Cc + c], wq[(long)k * Cc + c], a);
  m_frag[ds_rw_frag_index(cp, 0, k, 1, Cc)] = f2h(a);
  if (k == 0) {
    float bb = 0.f;
    for (int c = 0; c < Cc; ++c) bb = fmaf(wk[(long)cp * Cc + c], bq[c], bb);
    b_fold[cp] = bb;
  }
}
// Which weights the engine keeps chunk-major: every conv whose input channels are a multiple of 64 and whose concat
// split (c1 channels from the first source, 0 = no concat) falls on a chunk boundary
static int weight_chunk(int taps, int cin, int c1, int dtype) {
  const int kc = ds_conv_chunk(taps, dtype);
  return (cin % 64 == 0 && c1 % kc == 0) ? kc : 0;
}
// Conv_2 of a block is folded into its second 3x3 convolution when that one runs on a 64-cout tile
static bool fuse_skip(const Module& m) { return m.has_conv2 && m.out_ch > 32; }

// ------------------------------------------------------------------ process-wide options, device properties
// Defaults of the dispatch switches: the environment is read ONCE, here (never inside a launch decision); diffsep_set_option
// changes them for engines created later and for the unit entry points.
static std::atomic<unsigned> g_opts{0};  // (read by every thread that creates an engine or calls a unit entry point)
static std::once_flag g_opts_once;
unsigned ds_default_opts() {
  std::call_once(g_opts_once, [] {
    auto on = [](const char* n) { const char* v = getenv(n); return v && *v && strcmp(v, "0") != 0; };
    if (on("DIFFSEP_NO_RW")) g_opts |= DS_OPT_NO_RW;
    if (on("DIFFSEP_NO_RW128")) g_opts |= DS_OPT_NO_RW128;
    if (on("DIFFSEP_RW_SMALL")) g_opts |= DS_OPT_RW_SMALL;
    if (on("DIFFSEP_NO_RW_RES")) g_opts |= DS_OPT_NO_RW_RES;
    if (on("DIFFSEP_RW_HALF")) g_opts |= DS_OPT_RW_HALF;
    if (on("DIFFSEP_RW_QUARTER")) g_opts |= DS_OPT_RW_QUARTER;
    if (on("DIFFSEP_RW_BIG_HALF")) g_opts |= DS_OPT_RW_BIG_HALF;
    if (on("DIFFSEP_NO_STFT_FUSED")) g_opts |= DS_OPT_NO_STFT_FUSED;
    if (on("DIFFSEP_NO_SW")) g_opts |= DS_OPT_NO_SW;
    if (on("DIFFSEP_NO_SWS")) g_opts |= DS_OPT_NO_SWS;
    if (on("DIFFSEP_NO_SW_ROWS4")) g_opts |= DS_OPT_NO_SW_ROWS4;
    if (on("DIFFSEP_NO_SW_RW")) g_opts |= DS_OPT_NO_SW_RW;
  });
  return g_opts;
}
int ds_num_cus() {
  static std::atomic<int> cus[32];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
  int c = cus[dev].load(std::memory_order_relaxed);
  if (c <= 0) {
    hipDeviceProp_t prop;
    c = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    if (c <= 0) c = 256;
    cus[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}
static int opt_bit(const char* name, unsigned* bit) {
  static const struct { const char* n; unsigned b; } tab[] = {
      {"no_rw", DS_OPT_NO_RW}, {"no_rw128", DS_OPT_NO_RW128}, {"rw_small", DS_OPT_RW_SMALL}, {"no_rw_res", DS_OPT_NO_RW_RES},
      {"no_wfrag", DS_OPT_NO_WFRAG},
      {"no_attn_fused", DS_OPT_NO_ATTN_FUSED},
      {"no_stft_fused", DS_OPT_NO_STFT_FUSED},
      {"rw_half", DS_OPT_RW_HALF},
      {"rw_quarter", DS_OPT_RW_QUARTER},
      {"rw_big_half", DS_OPT_RW_BIG_HALF},
      {"no_split256", DS_OPT_NO_SPLIT256},
      {"no_sw", DS_OPT_NO_SW},
      {"no_sws", DS_OPT_NO_SWS},
      {"no_sw_rows4", DS_OPT_NO_SW_ROWS4},
      {"sw_rows4", DS_OPT_SW_ROWS4},
      {"no_sw_rw", DS_OPT_NO_SW_RW}};
  for (const auto& t : tab)
    if (!strcmp(name, t.n)) { *bit = t.b; return 0; }
  return 1;
}
extern "C" int32_t diffsep_set_option(const char* name, int64_t value) {
  DS_CHECK(name, "set_option: null name");
  unsigned bit = 0;
  if (opt_bit(name, &bit)) { ds_set_error(std::string("set_option: unknown option '") + name + "'"); return 1; }
  ds_default_opts();
  if (value) g_opts.fetch_or(bit); else g_opts.fetch_and(~bit);
  return 0;
}

// ------------------------------------------------------------------ engine
struct Tn {  // NHWC view; optionally the in-place channel concat of two tensors (C1 channels from p, rest from p2)
  void* p = nullptr;
  int C = 0, ld = 0, H = 0, W = 0;
  void* p2 = nullptr;
  int C1 = 0, ld2 = 0;
  // channel-sum accumulators filled by the producing conv ([B][C][2] fixed-point int64, common.h), or null
  long long* sa = nullptr;
  long long* sa2 = nullptr;
};
static Tn cat_view(const Tn& a, const Tn& b) {
  Tn t = a;
  t.C = a.C + b.C; t.C1 = a.C; t.p2 = b.p; t.ld2 = b.ld;
  t.sa2 = b.sa;
  return t;
}

struct diffsep_engine {
  diffsep_model_config cfg;
  Arch arch;
  int esz = 4;
  float* d_blob = nullptr;
  char* d_pack = nullptr;
  float* d_dense_w = nullptr;
  float* d_dense_b = nullptr;
  float* d_attn_b = nullptr;  // folded query / key biases of the fused attention blocks
  float* d_tab = nullptr;
  // arena
  char* arena = nullptr;
  size_t cap = 0, top = 0, fwd_base = 0;
  size_t stats_need = 0, stats_used = 0;  // GroupNorm accumulator region of one forward (sized by the dry run)
  char* stats_ptr = nullptr;
  bool dry = false;
  int planB = -1;
  long planT = -1;
  // sampler state (inside the arena, below fwd_base)
  float *st_x = nullptr, *st_xm = nullptr, *st_score = nullptr, *st_t = nullptr, *st_noise = nullptr,
        *st_ts = nullptr, *st_mix = nullptr, *st_smix = nullptr, *st_lang = nullptr;
  int* st_lens = nullptr;             // per-utterance lengths of a mixed-length batch (diffsep_sampler_ext)
  unsigned long long* st_seeds = nullptr;
  char* ext_pin = nullptr;            // pinned staging of (lengths, seeds) + the event of its last upload
  size_t ext_pin_cap = 0;
  hipEvent_t ext_ev = nullptr;
  bool ext_ev_rec = false;
  // graph of one NFE: (st_x, st_t, st_mix) -> st_score
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  bool graph_ok = false;
  // the captured graphs of the plans seen so far, keyed by (B, T): every plan lays its tensors out in the ONE arena, so a
  // graph stays valid until the arena is reallocated.  graph / gexec / graph_ok above are the current plan's entry.
  // The cache is an LRU of graph_cap plans (option "graph_cache", default 12: a CLI meets a handful of (batch, width bucket)
  // pairs; the Python API passes raw signal lengths, and a loop over utterances of distinct lengths must not keep one graph of
  // several hundred nodes per length for ever).
  struct GraphRec { hipGraph_t g; hipGraphExec_t x; uint64_t used; };
  std::map<std::pair<int, long>, GraphRec> graphs;
  int graph_cap = 12;
  uint64_t graph_tick = 0;
  int use_graph = 1;
  unsigned opts = 0;      // DS_OPT_* dispatch switches of this engine's launches (copied from the process defaults at creation)
  unsigned ablate = 0;    // option "ablate" (measurement aid, tools/ablate_bench.py): launch classes that are SKIPPED
  bool warmed = false;
  int64_t weight_bytes = 0;
  // work never runs on the legacy null stream (it cannot be captured): a NULL `stream` argument is
  // mapped to this private stream, ordered against the null stream with events on both sides.
  hipStream_t own = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  float* ts_pin = nullptr;    // pinned staging buffer of the time-step upload (+ the event of its last use)
  size_t ts_pin_cap = 0;
  hipEvent_t ts_ev = nullptr;
  bool ts_ev_rec = false;
  std::vector<float> ts_dev;  // time steps currently in st_ts (for ts_B batch rows): re-uploaded only when they change
  int ts_B = 0;
  bool had_arena = false;
  bool dbg_alloc = false;  // DIFFSEP_DBG_ALLOC=1: log every arena allocation (offset, bytes) to stderr
  // option "track_tensors": every activation tensor of a forward is recorded so that diffsep_engine_debug_absmax can scan them
  // (the range margin of half-precision storage: tests/test_round5_gpu.py); off by default, eager forwards only
  bool track_tensors = false;
  struct Tracked { void* p; long n; int H, W, C; };
  std::vector<Tracked> tracked;
  // DIFFSEP_F32_SPLIT: fp32 tensors, every MFMA product as 3 bf16 MFMAs on hi / lo halves (cfg.dtype stays DS_F32)
  int split = 0;
  // optional per-launch timing of the MFMA kernels (HIP events on the launch stream)
  bool prof = false;
  struct ProfRec { hipEvent_t a, b; double flops, bytes; int cls; const char* kernel; int B, H, W, Cin, Cout, taps, sCin, res; float ms; };
  std::vector<ProfRec> prof_done;  // the records of the last profile_begin .. profile_end span, with their times
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool;
};
#define DS_NCLS 12
static hipEvent_t prof_event(diffsep_engine* e) {
  if (!e->ev_pool.empty()) { hipEvent_t v = e->ev_pool.back(); e->ev_pool.pop_back(); return v; }
  hipEvent_t v = nullptr;
  hipEventCreate(&v);
  return v;
}
static int conv_launch_prof(diffsep_engine* e, const ConvArgs& a, hipStream_t st) {
  if (!e->prof) return ds_launch_conv(a, st);
  diffsep_engine::ProfRec r;
  r.a = prof_event(e); r.b = prof_event(e);
  r.flops = 2.0 * ((double)a.taps * a.Cin + (a.sx ? a.sCin : 0)) * a.Cout * (double)a.H * a.W * a.B;
  {  // algorithmic HBM bytes: input (+ fused skip input) + output (+ residual) once each, weights once
    const double esz = a.dtype == DS_F32 ? 4.0 : 2.0;
    r.bytes = esz * ((double)a.B * a.H * a.W *
                         ((double)a.Cin + a.Cout + (a.res ? a.Cout : 0) + (a.sx ? a.sCin : 0)) +
                     ((double)a.taps * a.Cin + (a.sx ? a.sCin : 0)) * a.Cout);
  }
  r.cls = ds_conv_config_id(a);
  r.B = a.B; r.H = a.H; r.W = a.W; r.Cin = a.Cin; r.Cout = a.Cout; r.taps = a.taps; r.sCin = a.sx ? a.sCin : 0; r.res = a.res != nullptr;
  r.ms = 0.f;
  hipEventRecord(r.a, st);
  const int rc = ds_launch_conv(a, st);
  hipEventRecord(r.b, st);
  r.kernel = ds_last_conv_kernel();
  e->prof_recs.push_back(r);
  return rc;
}

// The HBM-bound launches of the path (GroupNorm apply / FIR resampling, STFT / iSTFT, SDE updates, RNG) in the same profile span:
// records with cls = -1 (not a member of the per-class arrays of profile_end), flops = 0, bytes = the ALGORITHMIC HBM bytes of the
// launch (every input read once, every output written once), kernel = a static name.  `body` issues the launch (or launch
// sequence: the STFT is frame + DFT + pack) on st.
template <typename F>
static int hbm_launch_prof(diffsep_engine* e, hipStream_t st, const char* name, double bytes, int B, int H, int W, int C, F&& body) {
  if (!e->prof) return body();
  diffsep_engine::ProfRec r;
  r.a = prof_event(e); r.b = prof_event(e);
  r.flops = 0.0; r.bytes = bytes; r.cls = -1; r.kernel = name;
  r.B = B; r.H = H; r.W = W; r.Cin = C; r.Cout = 0; r.taps = 0; r.sCin = 0; r.res = 0; r.ms = 0.f;
  hipEventRecord(r.a, st);
  const int rc = body();
  hipEventRecord(r.b, st);
  e->prof_recs.push_back(r);
  return rc;
}

struct StreamScope {
  diffsep_engine* e; hipStream_t user; hipStream_t st;
  StreamScope(diffsep_engine* e_, void* s) : e(e_), user((hipStream_t)s), st((hipStream_t)s) {
    if (!user) {
      // the private stream is only created for callers on the null stream: HIP maps streams onto its few hardware
      // queues in creation order, and streams nobody uses would alias the caller's streams onto one queue
      if (!e->own) hipStreamCreateWithFlags(&e->own, hipStreamNonBlocking);
      st = e->own;
      hipEventRecord(e->ev_in, nullptr);
      hipStreamWaitEvent(st, e->ev_in, 0);
    }
  }
  ~StreamScope() {
    if (!user) {
      hipEventRecord(e->ev_out, st);
      hipStreamWaitEvent(nullptr, e->ev_out, 0);
    }
  }
};

static void* e_alloc(diffsep_engine* e, size_t bytes) {
  const size_t a = (e->top + 255) & ~(size_t)255;
  e->top = a + bytes;
  if (e->dry) return (void*)(uintptr_t)(a + 256);  // fake non-null
  if (e->dbg_alloc) fprintf(stderr, "[diffsep alloc] %zu %zu\n", a, bytes);
  return e->arena + a;
}
// GroupNorm accumulators live in one region at the start of a forward's allocations: ONE fill launch per forward
// zeroes them all (they are filled by integer atomics)
static long long* e_alloc_stats(diffsep_engine* e, size_t bytes) {
  const size_t a = (e->stats_used + 255) & ~(size_t)255;
  e->stats_used = a + bytes;
  if (e->dry) { if (e->stats_used > e->stats_need) e->stats_need = e->stats_used; return (long long*)(uintptr_t)256; }
  if (e->stats_used > e->stats_need) { ds_set_error("internal: GroupNorm accumulator region overflow"); return nullptr; }
  return (long long*)(e->stats_ptr + a);
}
static int stats_begin(diffsep_engine* e, hipStream_t st) {  // call right after e->top = e->fwd_base
  e->stats_used = 0;
  if (e->dry) { e->stats_need = 0; return 0; }
  e->stats_ptr = (char*)e_alloc(e, e->stats_need);
  // zeroed by a kernel, not hipMemsetAsync: as a captured memset NODE it made every engine but the first return
  // different samples when the graph was replayed on another stream (B >= 4; profiles/experiments/README.md)
  if (e->stats_need && ds_launch_fill((float*)e->stats_ptr, 0.f, (long)(e->stats_need / 4), st)) return 1;
  return 0;
}
static Tn e_tensor(diffsep_engine* e, int B, int H, int W, int C) {
  Tn t;
  t.C = C; t.ld = C; t.H = H; t.W = W;
  t.p = e_alloc(e, (size_t)B * H * W * C * e->esz);
  if (e->track_tensors && !e->dry) e->tracked.push_back({t.p, (long)B * H * W * C, H, W, C});
  return t;
}
static float* e_f32(diffsep_engine* e, size_t n) { return (float*)e_alloc(e, n * 4); }
static const float* P(diffsep_engine* e, const PRef& r) { return e->d_blob + r.off; }
static const void* PK(diffsep_engine* e, long off) { return e->d_pack + off * e->esz; }
static const void* PKF(diffsep_engine* e, long off) { return off >= 0 ? e->d_pack + off * e->esz : nullptr; }

// option "ablate" (a measurement aid: what would the step cost if these launches were free?): bit 0 = every convolution /
// GEMM of the <= 16-row levels (attention included), 1 / 2 / 5 / 6 = those of the 32 / 64 / 128 / 256-row level, 3 = the
// gn_apply / FIR resampling kernels, 4 = gn_finalize, 7 = the STFT / iSTFT passes.  Results are garbage by construction.
static bool ablated(const diffsep_engine* e, int H) {
  if (!e->ablate) return false;
  const unsigned bit = H <= 16 ? 1u : H == 32 ? 2u : H == 64 ? 4u : H == 128 ? 32u : H == 256 ? 64u : 0u;
  return (e->ablate & bit) != 0;
}

// ---- launch helpers (skip when dry)
// GroupNorm of a conv input: either materialised per-(b,c) scale / shift arrays, or (lazy) the producers'
// accumulators + affine parameters, from which the consuming conv builds the table itself (no launch)
struct GnAff {
  float* scale = nullptr; float* shift = nullptr;
  const long long* acc1 = nullptr; const long long* acc2 = nullptr;
  const float* gamma = nullptr; const float* beta = nullptr; int groups = 0; float inv_count = 0.f;
};
struct SkipConv { const Tn* x; const void* w; int chunk; const void* w_frag; };  // fused 1x1 skip convolution on the raw block input
static int conv(diffsep_engine* e, const Tn& x, const void* w, const float* bias, const float* bias_b, int bias_b_ld,
                const Tn* res, float scale, Tn& y, int Cout, int taps, int B, const float* div_b,
                hipStream_t st, const GnAff* gn = nullptr, int gn_act = 0, bool want_stats = false,
                const SkipConv* skip = nullptr, const void* w_frag = nullptr, const void* ident_frag = nullptr) {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.H = x.H; a.W = x.W; a.Cin = x.C; a.Cout = Cout; a.taps = taps; a.dtype = e->cfg.dtype; a.split = e->split;
  a.opts = e->opts;
  a.x = x.p; a.x_bs = (long)x.H * x.W * x.ld; a.ldx = x.ld;
  a.x2 = x.p2; a.x2_bs = (long)x.H * x.W * x.ld2; a.ldx2 = x.ld2; a.C1 = x.C1;
  a.gn_scale = gn ? gn->scale : nullptr; a.gn_shift = gn ? gn->shift : nullptr; a.gn_act = gn_act;
  if (gn && gn->acc1) {
    a.gn_acc1 = gn->acc1; a.gn_acc2 = gn->acc2; a.gn_gamma = gn->gamma; a.gn_beta = gn->beta;
    a.gn_groups = gn->groups; a.gn_inv_count = gn->inv_count; a.gn_eps = 1e-6f;
  }
  a.w = w; a.w_bs = 0; a.w_chunked = weight_chunk(taps, x.C, x.p2 ? x.C1 : 0, e->cfg.dtype);
  const bool use_frag = (e->cfg.dtype == DS_BF16 || (e->cfg.dtype == DS_F32 && e->split)) && !(e->opts & DS_OPT_NO_WFRAG);
  a.w_frag = use_frag ? w_frag : nullptr;
  a.ident_frag = use_frag ? ident_frag : nullptr;
  a.bias = bias; a.bias_b = bias_b; a.bias_b_ld = bias_b_ld; a.bias_mode = 0; a.div_b = div_b;
  a.res = res ? res->p : nullptr; a.res_bs = res ? (long)res->H * res->W * res->ld : 0; a.ldr = res ? res->ld : 0;
  a.out_scale = scale;
  a.y = y.p; a.y_bs = (long)y.H * y.W * y.ld; a.ldy = y.ld;
  if (skip) {
    const Tn& sx = *skip->x;
    a.sx = sx.p; a.sx_bs = (long)sx.H * sx.W * sx.ld; a.ldsx = sx.ld;
    a.sx2 = sx.p2; a.sx2_bs = (long)sx.H * sx.W * sx.ld2; a.ldsx2 = sx.ld2; a.sC1 = sx.C1; a.sCin = sx.C;
    a.sw = skip->w; a.sw_chunked = skip->chunk;
    a.sw_frag = use_frag ? skip->w_frag : nullptr;
  }
  if (want_stats) {  // the consumer's GroupNorm reads these partials instead of re-reading the tensor
    y.sa = e_alloc_stats(e, (size_t)B * Cout * 2 * sizeof(long long));
    if (!y.sa) return 1;
    a.stats_acc = y.sa;
    if (e->dbg_alloc && !e->dry)
      fprintf(stderr, "[diffsep stats] %ld Cin %d Cout %d taps %d HxW %dx%d res %d skip %d bias_b %d cfg %d\n",
              (long)((char*)y.sa - e->stats_ptr), x.C, Cout, taps, x.H, x.W, res != nullptr, skip != nullptr,
              bias_b != nullptr, ds_conv_config_id(a));
  }
  if (e->dry || ablated(e, x.H)) return 0;
  return conv_launch_prof(e, a, st);
}

static int gn_stats(diffsep_engine* e, const Tn& x, const float* gamma, const float* beta, int B, GnAff& aff,
                    hipStream_t st, bool lazy = false) {
  const int groups = (x.C / 4 < 32) ? x.C / 4 : 32;
  aff = GnAff();
  const bool have_acc = x.sa && (!x.p2 || x.sa2);
  if (have_acc && lazy && x.C <= 512) {  // the consuming conv computes scale / shift in its prologue
    // (its LDS table holds 512 channels; wider inputs take the materialised arrays below)
    aff.acc1 = x.sa; aff.acc2 = x.sa2; aff.gamma = gamma; aff.beta = beta; aff.groups = groups;
    aff.inv_count = (float)(1.0 / ((double)x.H * x.W * (x.C / groups)));
    return 0;
  }
  aff.scale = e_f32(e, (size_t)B * x.C);
  aff.shift = e_f32(e, (size_t)B * x.C);
  if (have_acc) {
    if (e->dry || (e->ablate & 16u)) return 0;
    return ds_launch_gn_finalize_acc(x.sa, x.p2 ? x.C1 : x.C, x.sa2, x.p2 ? x.C - x.C1 : 0, B, (long)x.H * x.W, groups,
                                     1e-6f, gamma, beta, aff.scale, aff.shift, st);
  }
  void* ws = e_alloc(e, (size_t)ds_gn_workspace_bytes(B, x.H, x.W, x.C));
  if (e->dry) return 0;
  return ds_launch_gn_stats(x.p, x.ld, x.p2, x.ld2, x.C1, B, x.H, x.W, x.C, groups, 1e-6f, gamma, beta, ws, aff.scale,
                            aff.shift, e->cfg.dtype, st);
}
static int gn_apply(diffsep_engine* e, const Tn& x, const GnAff* aff, const Tn* y, const Tn* xr, int B, int act,
                    int mode, hipStream_t st) {
  if (e->dry || (e->ablate & 8u)) return 0;
  // algorithmic bytes: x read once; each output (act(GN(x)) and / or raw x, resampled: x 4 up, / 4 down) written once
  const double esz = e->cfg.dtype == DS_F32 ? 4.0 : 2.0, nin = (double)B * x.H * x.W * x.C;
  const double fo = mode == 1 ? 4.0 : (mode == 2 ? 0.25 : 1.0);
  const double bytes = esz * nin * (1.0 + fo * ((y ? 1 : 0) + (xr ? 1 : 0)));
  const char* name = mode == 1 ? (aff ? "gn_fir_up (GroupNorm + SiLU + FIR x2 up of act and raw)" : "fir_up (pyramid)")
                               : (mode == 2 ? (aff ? "gn_fir_down (GroupNorm + SiLU + FIR x2 down of act and raw)" : "fir_down (pyramid)")
                                            : "gn_apply (GroupNorm affine + SiLU)");
  return hbm_launch_prof(e, st, name, bytes, B, x.H, x.W, x.C, [&]() {
    return ds_launch_gn_apply(x.p, x.ld, aff ? aff->scale : nullptr, aff ? aff->shift : nullptr, x.C, y ? y->p : nullptr,
                              y ? y->ld : 0, xr ? xr->p : nullptr, xr ? xr->ld : 0, B, x.H, x.W, act, mode, e->cfg.dtype,
                              st);
  });
}

static const float kInvSqrt2 = 0.70710678118654752440f;

// ResnetBlockBigGANpp.forward  layerspp.py:291-323.  act(GN(.)) is never materialised for the plain blocks:
// both 3x3 convs apply it while staging their input tile; x may be an in-place concat view.
static int res_block(diffsep_engine* e, const Module& m, const Tn& x, const float* temb_proj, int B, Tn& out,
                     hipStream_t st) {
  DS_CHECK(x.C == m.in_ch, "internal: resblock channel mismatch");
  const int mode = m.up ? 1 : (m.down ? 2 : 0);
  const int Ho = m.up ? 2 * x.H : (m.down ? x.H / 2 : x.H);
  const int Wo = m.up ? 2 * x.W : (m.down ? x.W / 2 : x.W);
  GnAff a0, a1;
  // cat(128, 128) -> 128 on a level with at least one 4 x 32 tile per CU (nf = 128 at 256^2 / 128^2), 16-bit: no register-weight
  // kernel holds 256 input channels (section 8 of DESIGN.md), but conv(cat(a, b)) = conv_a(a) + conv_b(b) and no GroupNorm group
  // straddles the seam (256 / 32 = 8 channels per group): each convolution runs as TWO 128 -> 128 register-weight launches, the
  // second taking the first's result as its residual (one more storage rounding of a partial sum).  Conv_0: 850 -> 593 us at
  // 256^2; Conv_1 + the 256-channel 1x1 skip: the first half of the skip folded as before, the second as a 1x1 launch.
  // Only from 128 rows up: at 64^2 (where nf = 64 has these blocks) the register-weight launches carry their weight prologue for
  // two tiles per block and the pair is SLOWER than the generic tile (61.5 + 50.0 against 49.3 + 42.6 us per block in the graph).
  // Round 5, with the streamed-weight kernel (conv3x3_sw.hip; stand-alone, B = 16): Conv_0 as ONE launch 659 us at 256^2 against
  // 269 + 318 for the pair, 155 against 77 + 89 at 128^2 — the pair stays at 256 rows; Conv_1 with the WHOLE 256-channel skip
  // folded 438 us at 256^2 against 319 + 193 (3x3 with half of the skip + the 1x1 launch on the other half), 103 against 91 + 47 at
  // 128^2 — one launch at every size.  Option no_sw: the route of round 4.
  if (e->cfg.dtype == DS_BF16 && mode == 0 && x.p2 && m.pf0a >= 0 && x.C1 == 128 && x.sa && x.sa2 && x.W % 32 == 0 && x.H % 8 == 0 &&
      x.H >= ((e->opts & DS_OPT_NO_SW) ? 128 : 256) && (long)B * (x.H / 4) * (x.W / 32) >= ds_num_cus() &&
      !(e->opts & (DS_OPT_NO_RW | DS_OPT_NO_RW128 | DS_OPT_NO_SPLIT256 | DS_OPT_NO_WFRAG))) {
    Tn xa = x, xb = x;
    xa.C = 128; xa.p2 = nullptr; xa.C1 = 0; xa.ld2 = 0; xa.sa2 = nullptr;
    xb.p = x.p2; xb.ld = x.ld2; xb.C = 128; xb.p2 = nullptr; xb.C1 = 0; xb.ld2 = 0; xb.sa = x.sa2; xb.sa2 = nullptr;
    GnAff ga, gb;
    ga.acc1 = xa.sa; ga.gamma = P(e, m.gn0_w); ga.beta = P(e, m.gn0_b); ga.groups = 16;
    ga.inv_count = (float)(1.0 / ((double)x.H * x.W * 8));
    gb = ga; gb.acc1 = xb.sa; gb.gamma = ga.gamma + 128; gb.beta = ga.beta + 128;
    Tn h1p = e_tensor(e, B, Ho, Wo, m.out_ch), h1 = e_tensor(e, B, Ho, Wo, m.out_ch);
    const long half0 = 4L * 9 * m.out_ch * 32;  // chunk-major [Cin / 32][9][Cout][32]: the first source = the first 4 chunks
    if (conv(e, xa, PK(e, m.pk0), P(e, m.conv0_b), temb_proj + m.temb_off, e->arch.dense_total, nullptr, 1.f, h1p, m.out_ch, 9, B,
             nullptr, st, &ga, 1, false, nullptr, PKF(e, m.pf0a)))
      return 1;
    if (conv(e, xb, PK(e, m.pk0 + half0), nullptr, nullptr, 0, &h1p, 1.f, h1, m.out_ch, 9, B, nullptr, st, &gb, 1, true, nullptr,
             PKF(e, m.pf0b)))
      return 1;
    if (gn_stats(e, h1, P(e, m.gn1_w), P(e, m.gn1_b), B, a1, st, true)) return 1;
    out = e_tensor(e, B, Ho, Wo, m.out_ch);
    if (!(e->opts & DS_OPT_NO_SW) && m.pf2 >= 0) {
      const SkipConv skw = {&x, PK(e, m.pk2), weight_chunk(9, m.in_ch, m.in_c1, e->cfg.dtype), PKF(e, m.pf2)};
      return conv(e, h1, PK(e, m.pk1), P(e, m.conv1_b), P(e, m.conv2_b), 0, nullptr, kInvSqrt2, out, m.out_ch, 9, B, nullptr, st, &a1, 1,
                  true, &skw, PKF(e, m.pf1));
    }
    Tn outp = e_tensor(e, B, Ho, Wo, m.out_ch);
    const SkipConv sk = {&xa, PK(e, m.pk2), weight_chunk(9, 128, 0, e->cfg.dtype), PKF(e, m.pf2a)};
    if (conv(e, h1, PK(e, m.pk1), P(e, m.conv1_b), P(e, m.conv2_b), 0, nullptr, 1.f, outp, m.out_ch, 9, B, nullptr, st, &a1, 1,
             false, &sk, PKF(e, m.pf1)))
      return 1;
    return conv(e, xb, PK(e, m.pk2b), nullptr, nullptr, 0, &outp, kInvSqrt2, out, m.out_ch, 1, B, nullptr, st, nullptr, 0, true);
  }
  if (gn_stats(e, x, P(e, m.gn0_w), P(e, m.gn0_b), B, a0, st, mode == 0)) return 1;  // resampling needs the arrays
  Tn h1 = e_tensor(e, B, Ho, Wo, m.out_ch);
  Tn xr = x, h0m;
  if (mode) {
    DS_CHECK(x.p2 == nullptr, "internal: resampling block on a concat view");
    h0m = e_tensor(e, B, Ho, Wo, m.in_ch);
    xr = e_tensor(e, B, Ho, Wo, m.in_ch);
    if (gn_apply(e, x, &a0, &h0m, &xr, B, 1, mode, st)) return 1;
  }
  // Conv_2 (1x1 on the raw, possibly resampled block input; layerspp.py:317-318) is folded into the second 3x3
  // convolution as extra K through its centre tap: no separate launch, no skip tensor in HBM
  if (!m.has_conv2) DS_CHECK(xr.p2 == nullptr, "internal: identity skip on a concat view");
  if (mode) {
    if (conv(e, h0m, PK(e, m.pk0), P(e, m.conv0_b), temb_proj + m.temb_off, e->arch.dense_total, nullptr, 1.f, h1,
             m.out_ch, 9, B, nullptr, st, nullptr, 0, true, nullptr, PKF(e, m.pf0)))
      return 1;
  } else {
    if (conv(e, x, PK(e, m.pk0), P(e, m.conv0_b), temb_proj + m.temb_off, e->arch.dense_total, nullptr, 1.f, h1,
             m.out_ch, 9, B, nullptr, st, &a0, 1, true, nullptr, PKF(e, m.pf0)))
      return 1;
  }
  if (gn_stats(e, h1, P(e, m.gn1_w), P(e, m.gn1_b), B, a1, st, true)) return 1;
  out = e_tensor(e, B, Ho, Wo, m.out_ch);
  if (m.has_conv2 && fuse_skip(m)) {
    DS_CHECK(ds_conv_skip_supported(Ho, Wo, m.out_ch, e->cfg.dtype), "internal: fused skip conv on an unsupported tile");
    const SkipConv sk = {&xr, PK(e, m.pk2), weight_chunk(9, m.in_ch, m.in_c1, e->cfg.dtype), PKF(e, m.pf2)};
    // conv bias + Conv_2 bias: the second goes in as a "per-batch" bias with stride 0
    return conv(e, h1, PK(e, m.pk1), P(e, m.conv1_b), P(e, m.conv2_b), 0, nullptr, kInvSqrt2, out, m.out_ch, 9, B,
                nullptr, st, &a1, 1, true, &sk, PKF(e, m.pf1));
  }
  Tn skip = xr;
  if (m.has_conv2) {  // narrow blocks (<= 32 couts use the 32-cout tile, which has no skip path): separate 1x1 launch
    skip = e_tensor(e, B, Ho, Wo, m.out_ch);
    if (conv(e, xr, PK(e, m.pk2), P(e, m.conv2_b), nullptr, 0, nullptr, 1.f, skip, m.out_ch, 1, B, nullptr, st)) return 1;
  }
  return conv(e, h1, PK(e, m.pk1), P(e, m.conv1_b), nullptr, 0, &skip, kInvSqrt2, out, m.out_ch, 9, B, nullptr, st,
              &a1, 1, true, nullptr, PKF(e, m.pf1), m.has_conv2 ? nullptr : PKF(e, m.pf_id));
}

// attention core shared with the unit entry point: o = softmax(q k^T C^-1/2) v
static int attention_core(const void* q, const void* k, const void* vt, void* o, int B, int L, int C, int ldq, int ldo,
                          void* scores, void* probs, int dtype, hipStream_t st, int split = 0) {
  const int Lp = rup8(L);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.dtype = dtype; a.split = split; a.B = B; a.taps = 1; a.bias_mode = 0; a.out_scale = 1.f;
  // scores[b, i, j] = sum_c q[b,i,c] k[b,j,c] * C^-0.5
  a.x = q; a.x_bs = (long)L * ldq; a.ldx = ldq;
  a.w = k; a.w_bs = (long)L * C;
  a.y = scores; a.y_bs = (long)L * Lp; a.ldy = Lp;
  a.H = 1; a.W = L; a.Cin = C; a.Cout = L;
  a.out_scale = 1.0f / sqrtf((float)C);
  if (ds_launch_conv(a, st)) return 1;
  if (ds_launch_softmax(scores, probs, (long)B * L, L, Lp, dtype, st)) return 1;
  // o[b, i, c] = sum_j P[b,i,j] vt[b,c,j]
  a.x = probs; a.x_bs = (long)L * Lp; a.ldx = Lp;
  a.w = vt; a.w_bs = (long)C * Lp;
  a.y = o; a.y_bs = (long)L * ldo; a.ldy = ldo;
  a.H = 1; a.W = L; a.Cin = Lp; a.Cout = C;
  a.out_scale = 1.f;
  return ds_launch_conv(a, st);
}

// AttnBlockpp.forward  layerspp.py:76-92
static int attn_block(diffsep_engine* e, const Module& m, const Tn& x, int B, Tn& out, hipStream_t st) {
  const int C = m.in_ch, L = x.H * x.W, Lp = rup8(L);
  DS_CHECK(x.C == C, "internal: attention channel mismatch");
  if (ds_attn_fused_eligible(e->cfg.dtype, C, L) && !x.p2 && m.pf_nin[0] >= 0 && !(e->opts & DS_OPT_NO_ATTN_FUSED)) {
    // the whole block in one launch (attn_fused.hip): GroupNorm from the producer's accumulators (a tensor without them — the
    // unit entry point — gets the statistics kernels first), projections, softmax attention, output projection, residual,
    // statistics for the consumer
    GnAff ga;
    if (!x.sa && gn_stats(e, x, P(e, m.gn0_w), P(e, m.gn0_b), B, ga, st)) return 1;
    out = e_tensor(e, B, x.H, x.W, C);
    out.sa = e_alloc_stats(e, (size_t)B * C * 2 * sizeof(long long));
    if (!out.sa) return 1;
    if (e->dry || ablated(e, x.H)) return 0;
    AttnFusedArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x.p; a.x_bs = (long)L * x.ld; a.ldx = x.ld;
    a.gn_acc = x.sa; a.gn_scale = ga.scale; a.gn_shift = ga.shift; a.gn_gamma = P(e, m.gn0_w); a.gn_beta = P(e, m.gn0_b);
    a.gn_groups = (C / 4 < 32) ? C / 4 : 32; a.gn_inv_count = (float)(1.0 / ((double)L * (C / a.gn_groups))); a.gn_eps = 1e-6f;
    a.wqk = PKF(e, m.pf_nin[0]); a.wv = PKF(e, m.pf_nin[2]); a.wo = PKF(e, m.pf_nin[3]);
    a.bqk = e->d_attn_b + m.ab_off; a.bv = P(e, m.nin_b[2]); a.bo = P(e, m.nin_b[3]);
    a.y = out.p; a.y_bs = (long)L * out.ld; a.ldy = out.ld;
    a.stats = out.sa;
    a.B = B; a.L = L; a.C = C;
    if (!e->prof) return ds_launch_attn_fused(a, st);
    diffsep_engine::ProfRec r;  // (per-launch timing like the convolutions: class 9)
    r.a = prof_event(e); r.b = prof_event(e);
    // V^T, Q' and the output projection (2 L C^2 each), scores and P V (2 L^2 C each); input + output + three matrices once
    r.flops = (double)B * (6.0 * L * C * C + 4.0 * (double)L * L * C);
    r.bytes = (double)e->esz * (2.0 * B * L * C + 3.0 * C * C);
    r.cls = 9; r.B = B; r.H = x.H; r.W = x.W; r.Cin = C; r.Cout = C; r.taps = 1; r.sCin = 0; r.res = 1; r.ms = 0.f;
    r.kernel = "attn_fused_kernel";
    hipEventRecord(r.a, st);
    const int rc = ds_launch_attn_fused(a, st);
    hipEventRecord(r.b, st);
    e->prof_recs.push_back(r);
    return rc;
  }
  GnAff a0;
  if (gn_stats(e, x, P(e, m.gn0_w), P(e, m.gn0_b), B, a0, st)) return 1;
  Tn h = e_tensor(e, B, x.H, x.W, C);
  if (gn_apply(e, x, &a0, &h, nullptr, B, 0, 0, st)) return 1;
  Tn q = e_tensor(e, B, x.H, x.W, C), k = e_tensor(e, B, x.H, x.W, C);
  if (conv(e, h, PK(e, m.pk_nin[0]), P(e, m.nin_b[0]), nullptr, 0, nullptr, 1.f, q, C, 1, B, nullptr, st)) return 1;
  if (conv(e, h, PK(e, m.pk_nin[1]), P(e, m.nin_b[1]), nullptr, 0, nullptr, 1.f, k, C, 1, B, nullptr, st)) return 1;
  // V^T[b, c, l] = sum_c' Wv[c', c] h[b, l, c'] + b[c]: A = packed Wv^T ([C][C]), Bt = h, bias along rows
  void* vt = e_alloc(e, (size_t)B * C * Lp * e->esz);
  void* scores = e_alloc(e, (size_t)B * L * Lp * e->esz);
  void* probs = e_alloc(e, (size_t)B * L * Lp * e->esz);
  Tn o = e_tensor(e, B, x.H, x.W, C);
  out = e_tensor(e, B, x.H, x.W, C);
  if (!e->dry && !ablated(e, x.H)) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.dtype = e->cfg.dtype; a.split = e->split; a.B = B; a.taps = 1; a.out_scale = 1.f;
    a.x = PK(e, m.pk_nin[2]); a.x_bs = 0; a.ldx = C;
    a.w = h.p; a.w_bs = (long)L * C;
    a.bias = P(e, m.nin_b[2]); a.bias_mode = 1;
    a.y = vt; a.y_bs = (long)C * Lp; a.ldy = Lp;
    a.H = 1; a.W = C; a.Cin = C; a.Cout = L;
    if (ds_launch_conv(a, st)) return 1;
    if (attention_core(q.p, k.p, vt, o.p, B, L, C, C, C, scores, probs, e->cfg.dtype, st, e->split)) return 1;
  }
  // (the dry run must see this call too: it sizes the accumulator region)
  return conv(e, o, PK(e, m.pk_nin[3]), P(e, m.nin_b[3]), nullptr, 0, &x, kInvSqrt2, out, C, 1, B, nullptr, st, nullptr,
              0, true);
}

// NCSNpp.forward  ncsnpp.py:319-478.  x0: packed input AFTER 2x-1, [B,H,W,cpad_in]; y: [B,H,W,cpad_out]
// pyr_out != null: stop before the output layer and hand back the last pyramid tensor (the caller's iSTFT applies
// `output_layer(pyramid / t)` while unpacking)
static int net_forward(diffsep_engine* e, const Tn& x0, const float* t, const Tn& y, int B, hipStream_t st,
                       Tn* pyr_out = nullptr) {
  const Arch& A = e->arch;
  const diffsep_model_config& c = e->cfg;
  const int nf = c.nf;
  size_t mi = 0;
  // ---- time embedding (ncsnpp.py:324-343) and every block's Dense_0(act(temb)) (layerspp.py:311-312)
  float* emb = e_f32(e, (size_t)B * 2 * nf);
  float* t1 = e_f32(e, (size_t)B * 4 * nf);
  float* temb = e_f32(e, (size_t)B * 4 * nf);
  float* proj = e_f32(e, (size_t)B * A.dense_total);
  const Module& mf = A.mods[mi++];
  const Module& l1 = A.mods[mi++];
  const Module& l2 = A.mods[mi++];
  if (!e->dry) {
    if (ds_launch_fourier(t, P(e, mf.w0), emb, B, nf, st)) return 1;
    if (ds_launch_linear(emb, P(e, l1.w0), P(e, l1.b0), t1, B, 2 * nf, 4 * nf, 0, st)) return 1;
    if (ds_launch_linear(t1, P(e, l2.w0), P(e, l2.b0), temb, B, 4 * nf, 4 * nf, 1, st)) return 1;
    if (ds_launch_linear_t(temb, e->d_dense_w, e->d_dense_b, proj, B, 4 * nf, A.dense_total, 1, st)) return 1;
  }
  // ---- input conv
  const Module& cin = A.mods[mi++];
  std::vector<Tn> hs;
  {
    Tn h = e_tensor(e, B, x0.H, x0.W, nf);
    if (conv(e, x0, PK(e, cin.pk0), P(e, cin.b0), nullptr, 0, nullptr, 1.f, h, nf, 9, B, nullptr, st, nullptr, 0, true))
      return 1;
    hs.push_back(h);
  }
  Tn pyr_in = x0;
  const int L = c.n_levels;
  Tn h;
  for (int i = 0; i < L; ++i) {
    for (int k = 0; k < c.num_res_blocks; ++k) {
      if (res_block(e, A.mods[mi++], hs.back(), proj, B, h, st)) return 1;
      if (h.H == c.attn_resolution) {
        DS_CHECK(mi < A.mods.size() && A.mods[mi].kind == MK_ATTN, "attention placement mismatch (image height)");
        Tn ha;
        if (attn_block(e, A.mods[mi++], h, B, ha, st)) return 1;
        h = ha;
      }
      hs.push_back(h);
    }
    if (i != L - 1) {
      if (res_block(e, A.mods[mi++], hs.back(), proj, B, h, st)) return 1;
      // input pyramid: FIR down, Combine = conv1x1(pyr) + h   (ncsnpp.py:383-386, layerspp.py:52-57)
      Tn pd = e_tensor(e, B, pyr_in.H / 2, pyr_in.W / 2, pyr_in.C);
      if (gn_apply(e, pyr_in, nullptr, nullptr, &pd, B, 0, 2, st)) return 1;
      pyr_in = pd;
      const Module& cm = A.mods[mi++];
      DS_CHECK(cm.kind == MK_COMBINE, "internal: expected Combine");
      Tn hc = e_tensor(e, B, h.H, h.W, h.C);
      if (conv(e, pyr_in, PK(e, cm.pk0), P(e, cm.b0), nullptr, 0, &h, 1.f, hc, h.C, 1, B, nullptr, st, nullptr, 0, true))
        return 1;
      hs.push_back(hc);
    }
  }
  h = hs.back();
  Tn t2;
  if (res_block(e, A.mods[mi++], h, proj, B, t2, st)) return 1;
  if (attn_block(e, A.mods[mi++], t2, B, h, st)) return 1;
  if (res_block(e, A.mods[mi++], h, proj, B, t2, st)) return 1;
  h = t2;

  Tn pyramid;
  bool have_pyr = false;
  for (int i = L - 1; i >= 0; --i) {
    for (int k = 0; k < c.num_res_blocks + 1; ++k) {
      const Tn s = hs.back();
      hs.pop_back();
      const Tn cat = cat_view(h, s);  // torch.cat([h, hs.pop()], dim=1) read in place (ncsnpp.py:411)
      Tn hn2;
      if (res_block(e, A.mods[mi++], cat, proj, B, hn2, st)) return 1;
      h = hn2;
    }
    if (h.H == c.attn_resolution) {
      DS_CHECK(mi < A.mods.size() && A.mods[mi].kind == MK_ATTN, "attention placement mismatch (image height)");
      Tn ha;
      if (attn_block(e, A.mods[mi++], h, B, ha, st)) return 1;
      h = ha;
    }
    // output pyramid (ncsnpp.py:419-440): conv3x3(act(GN(h))) [+ FIR up of the previous pyramid]
    const Module& g = A.mods[mi++];
    const Module& cv = A.mods[mi++];
    DS_CHECK(g.kind == MK_GN && cv.kind == MK_CONV3, "internal: expected pyramid GN + conv");
    hipStream_t sp = st;  // (the pyramid chain on a side stream was measured 6 % slower: profiles/experiments)
    GnAff ga;
    if (gn_stats(e, h, P(e, g.w0), P(e, g.b0), B, ga, sp, true)) return 1;
    Tn pnew = e_tensor(e, B, h.H, h.W, A.cpad_in);
    if (have_pyr) {
      Tn pu = e_tensor(e, B, h.H, h.W, A.cpad_in);
      if (gn_apply(e, pyramid, nullptr, nullptr, &pu, B, 0, 1, sp)) return 1;
      if (conv(e, h, PK(e, cv.pk0), P(e, cv.b0), nullptr, 0, &pu, 1.f, pnew, A.chan_in, 9, B, nullptr, sp, &ga, 1))
        return 1;
    } else {
      if (conv(e, h, PK(e, cv.pk0), P(e, cv.b0), nullptr, 0, nullptr, 1.f, pnew, A.chan_in, 9, B, nullptr, sp, &ga, 1))
        return 1;
    }
    pyramid = pnew;
    have_pyr = true;
    if (i != 0) {
      Tn hu;
      if (res_block(e, A.mods[mi++], h, proj, B, hu, st)) return 1;
      h = hu;
    }
  }
  DS_CHECK(hs.empty() && mi == A.mods.size(), "internal: module walk did not consume all modules");
  // h = pyramid / t ; out = output_layer(h)   (ncsnpp.py:472-477)
  if (pyr_out) { *pyr_out = pyramid; return 0; }
  Tn yy = y;
  return conv(e, pyramid, PK(e, A.pk_out), P(e, A.out_b), nullptr, 0, nullptr, 1.f, yy, A.chan_out, 1, B, t, st);
}

// ScoreModelNCSNpp.forward  score_models.py:126-138
static int score_forward_impl(diffsep_engine* e, const float* xt, const float* t, const float* mix, float* out, int B,
                              long T, hipStream_t st) {
  const diffsep_model_config& c = e->cfg;
  const int W = diffsep_padded_frames(&c, T), H = c.n_fft / 2 + 1, S = c.num_sources;
  e->top = e->fwd_base;
  e->tracked.clear();
  if (stats_begin(e, st)) return 1;
  Tn x0 = e_tensor(e, B, H, W, e->arch.cpad_in);
  Tn y = e_tensor(e, B, H, W, e->arch.cpad_out);
  // the DFT GEMMs work on fp32 frames in every mode: exact fp32 MFMAs for the fp32 engine, bf16x3 products (4e-5, far
  // below the bf16 rounding of the packed spectrogram) for the split and the bf16 engine
  const int dft_split = e->split || c.dtype == DS_BF16;
  float* ws_f = (float*)e_alloc(e, (size_t)ds_stft_workspace_bytes(B, S, T, c.n_fft, c.hop));
  float* frames = (float*)e_alloc(e, (size_t)ds_istft_workspace_bytes(B, S, T, c.n_fft, c.hop));
  const double esz_t = c.dtype == DS_F32 ? 4.0 : 2.0;
  if (!e->dry && !(e->ablate & 128u))
    if (hbm_launch_prof(e, st, "stft (frame + real-DFT GEMM + compress / pack)",
                        4.0 * B * (S + 1) * (double)T + esz_t * B * H * (double)W * e->arch.cpad_in, B, H, W, e->arch.cpad_in, [&]() {
          return ds_launch_stft_pack(xt, mix, x0.p, B, S, T, c.n_fft, c.hop, c.spec_abs_exponent, c.spec_factor, W,
                                     e->arch.cpad_in, 1, c.dtype, e->d_tab, ws_f, st, dft_split);
        }))
      return 1;
  Tn pyr;
  if (net_forward(e, x0, t, y, B, st, &pyr)) return 1;
  if (!e->dry && !(e->ablate & 128u))
    if (hbm_launch_prof(e, st, "istft (unpack / decompress + inverse-DFT GEMM + overlap-add)",
                        esz_t * B * H * (double)W * pyr.ld + 4.0 * B * S * (double)T, B, H, W, pyr.ld, [&]() {
          return ds_launch_istft(pyr.p, out, B, S, T, c.n_fft, c.hop, c.spec_abs_exponent, c.spec_factor, W, pyr.ld, c.dtype,
                                 e->d_tab, frames, st, dft_split, P(e, e->arch.out_w), P(e, e->arch.out_b), t, e->arch.chan_in);
        }))
      return 1;
  return 0;
}

// Forget every captured graph (the arena moved, graphs were switched off, the engine goes away).  The caller has made sure
// that none of them is still executing.
static void drop_graph(diffsep_engine* e) {
  for (auto& kv : e->graphs) {
    if (kv.second.x) hipGraphExecDestroy(kv.second.x);
    if (kv.second.g) hipGraphDestroy(kv.second.g);
  }
  e->graphs.clear();
  e->gexec = nullptr;
  e->graph = nullptr;
  e->graph_ok = false;
}

// Size the arena for (B, T): sampler state + one forward's bump allocations.  kindW: if > 0 the
// plan is for backbone_forward with that width (no STFT) — handled by the caller via T = -W.
static int ensure_plan(diffsep_engine* e, int B, long T, hipStream_t st) {
  if (e->planB == B && e->planT == T && e->arena) return 0;
  DS_CHECK(B >= 1 && T >= 1, "empty batch or signal");
  // a plan seen before keeps its captured graph (one per (B, T): evaluate / separate alternate between a few widths and
  // the short last batch of each); only the layout and the zero padding of the arena are re-established below
  e->graph = nullptr; e->gexec = nullptr; e->graph_ok = false;
  const int S = e->cfg.num_sources;
  const size_t nst = (size_t)B * S * T;
  // state region
  e->top = 0;
  e->dry = true;
  e_alloc(e, nst * 4); e_alloc(e, nst * 4); e_alloc(e, nst * 4); e_alloc(e, nst * 4);
  e_alloc(e, (size_t)B * 4); e_alloc(e, (size_t)B * T * 4); e_alloc(e, (size_t)B * T * 4);
  e_alloc(e, 4096 * (size_t)B * 4); e_alloc(e, 16 * (size_t)B + 64);
  e_alloc(e, (size_t)B * 4); e_alloc(e, (size_t)B * 8);
  e->fwd_base = (e->top + 255) & ~(size_t)255;
  const int rc = score_forward_impl(e, nullptr, nullptr, nullptr, nullptr, B, T, st);
  e->dry = false;
  if (rc) return 1;
  const size_t need = e->top + e->stats_need + 8192;
  if (need > e->cap) {
    DS_HIP(hipStreamSynchronize(st));
    drop_graph(e);  // their addresses die with the old arena (nothing is in flight after the synchronisation)
    if (e->arena) DS_HIP(hipFree(e->arena));
    e->tracked.clear();  // (track_tensors: those pointers were into the old arena)
    e->arena = nullptr;
    e->cap = 0;
    // (hipFree / hipMalloc synchronise the whole device: grow with headroom so that a stream of utterances of
    // slowly increasing length does not reallocate — and stall every other stream — at each new maximum)
    const size_t grown = e->had_arena ? need + need / 4 : need;
    DS_HIP(hipMalloc((void**)&e->arena, grown));
    e->cap = grown;
    e->had_arena = true;
  }
  DS_HIP(hipMemsetAsync(e->arena, 0, e->cap, st));  // channel / K padding must read as zero
  e->top = 0;
  e->st_x = (float*)e_alloc(e, nst * 4);
  e->st_xm = (float*)e_alloc(e, nst * 4);
  e->st_score = (float*)e_alloc(e, nst * 4);
  e->st_noise = (float*)e_alloc(e, nst * 4);
  e->st_t = (float*)e_alloc(e, (size_t)B * 4);
  e->st_mix = (float*)e_alloc(e, (size_t)B * T * 4);
  e->st_smix = (float*)e_alloc(e, (size_t)B * T * 4);
  e->st_ts = (float*)e_alloc(e, 4096 * (size_t)B * 4);
  e->st_lang = (float*)e_alloc(e, 16 * (size_t)B + 64);
  e->st_lens = (int*)e_alloc(e, (size_t)B * 4);
  e->st_seeds = (unsigned long long*)e_alloc(e, (size_t)B * 8);
  e->planB = B;
  e->planT = T;
  {
    auto it = e->graphs.find(std::make_pair(B, T));
    if (it != e->graphs.end()) {
      e->graph = it->second.g; e->gexec = it->second.x; e->graph_ok = true;
      it->second.used = ++e->graph_tick;
    }
  }
  e->ts_dev.clear();
  // a new plan is captured at its first score evaluation (hipFuncSetAttribute inside the launchers is not a stream
  // operation and is legal during capture)
  e->warmed = true;
  return 0;
}

// ------------------------------------------------------------------ weight repack (fp32 blob -> engine dtype, kernel layout)
static int repack_weight(diffsep_engine* e, const PRef& src, long pk, int O, int I, int taps, long so, long si, long stp,
                         bool allow_chunk = true, int kc_taps = 0, int c1 = 0) {
  const int dtype = e->cfg.dtype;
  const int Ipad = rup8(I);
  // kc_taps: the kernel that will READ these weights (the fused skip conv is read by the 3x3 kernel)
  const int kc = allow_chunk ? weight_chunk(kc_taps ? kc_taps : taps, I, c1, dtype) : 0;
  const long total = (long)O * taps * Ipad;
  long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  if (dtype == DS_F32)
    hipLaunchKernelGGL(repack_kernel<float>, dim3(nb), dim3(256), 0, 0, e->d_blob + src.off, (float*)(e->d_pack) + pk, O,
                       I, Ipad, taps, so, si, stp, kc);
  else
    hipLaunchKernelGGL(repack_kernel<bf16_t>, dim3(nb), dim3(256), 0, 0, e->d_blob + src.off, (bf16_t*)(e->d_pack) + pk,
                       O, I, Ipad, taps, so, si, stp, kc);
  DS_LAUNCH_CHECK();
  return 0;
}
static int repack_frag(diffsep_engine* e, const PRef& src, long pf, int O, int I, int taps, long so, long si, long stp) {
  if (pf >= 0 && e->cfg.dtype == DS_F32 && e->split && ds_sws_frag_shape(taps, I, O)) {  // hi / lo planes: 2 x 2 bytes per weight = one slot
    hipLaunchKernelGGL(repack_frag_split_kernel, dim3(cdiv((long)O * taps * I, 256)), dim3(256), 0, 0, e->d_blob + src.off,
                       (bf16_t*)((float*)(e->d_pack) + pf), O, I, taps, so, si, stp);
    DS_LAUNCH_CHECK();
    return 0;
  }
  if (pf < 0 || e->cfg.dtype != DS_BF16) return 0;
  hipLaunchKernelGGL(repack_frag_kernel, dim3(cdiv((long)O * taps * I, 256)), dim3(256), 0, 0, e->d_blob + src.off,
                     (bf16_t*)(e->d_pack) + pf, O, I, taps, so, si, stp);
  DS_LAUNCH_CHECK();
  return 0;
}
static int repack_module(diffsep_engine* e, const Module& m) {
  int rc = 0;
  switch (m.kind) {
    case MK_CONV3: rc |= repack_weight(e, m.w0, m.pk0, m.out_ch, m.in_ch, 9, (long)m.in_ch * 9, 9, 1); break;
    case MK_COMBINE: rc |= repack_weight(e, m.w0, m.pk0, m.out_ch, m.in_ch, 1, m.in_ch, 1, 0); break;
    case MK_RES:
      rc |= repack_weight(e, m.conv0_w, m.pk0, m.out_ch, m.in_ch, 9, (long)m.in_ch * 9, 9, 1, true, 0, m.in_c1);
      rc |= repack_weight(e, m.conv1_w, m.pk1, m.out_ch, m.out_ch, 9, (long)m.out_ch * 9, 9, 1);
      if (m.has_conv2)
        rc |= repack_weight(e, m.conv2_w, m.pk2, m.out_ch, m.in_ch, 1, m.in_ch, 1, 0, true, fuse_skip(m) ? 9 : 0, m.in_c1);
      rc |= repack_frag(e, m.conv0_w, m.pf0, m.out_ch, m.in_ch, 9, (long)m.in_ch * 9, 9, 1);
      rc |= repack_frag(e, m.conv1_w, m.pf1, m.out_ch, m.out_ch, 9, (long)m.out_ch * 9, 9, 1);
      if (m.has_conv2) rc |= repack_frag(e, m.conv2_w, m.pf2, m.out_ch, m.in_ch, 1, m.in_ch, 1, 0);
      if (m.pf_id >= 0 && e->cfg.dtype == DS_BF16) {
        hipLaunchKernelGGL(repack_frag_kernel, dim3(cdiv((long)m.out_ch * m.out_ch, 256)), dim3(256), 0, 0, (const float*)nullptr,
                           (bf16_t*)(e->d_pack) + m.pf_id, m.out_ch, m.out_ch, 1, 0L, 0L, 0L);
        DS_LAUNCH_CHECK();
      }
      if (m.pf_id >= 0 && e->cfg.dtype == DS_F32 && e->split) {
        hipLaunchKernelGGL(repack_frag_split_kernel, dim3(cdiv((long)m.out_ch * m.out_ch, 256)), dim3(256), 0, 0, (const float*)nullptr,
                           (bf16_t*)((float*)(e->d_pack) + m.pf_id), m.out_ch, m.out_ch, 1, 0L, 0L, 0L);
        DS_LAUNCH_CHECK();
      }
      if (m.pf0a >= 0) {  // the halves of a cat(128, 128) block (channel offset 128 in the second)
        PRef w0b = m.conv0_w, w2b = m.conv2_w;
        w0b.off += 128L * 9;
        w2b.off += 128;
        rc |= repack_frag(e, m.conv0_w, m.pf0a, m.out_ch, 128, 9, (long)m.in_ch * 9, 9, 1);
        rc |= repack_frag(e, w0b, m.pf0b, m.out_ch, 128, 9, (long)m.in_ch * 9, 9, 1);
        rc |= repack_frag(e, m.conv2_w, m.pf2a, m.out_ch, 128, 1, m.in_ch, 1, 0);
        rc |= repack_weight(e, w2b, m.pk2b, m.out_ch, 128, 1, m.in_ch, 1, 0);
      }
      // Dense_0.weight [out][temb dim] -> columns [temb_off, temb_off + out) of the transposed concatenation
      // [temb dim][dense_total] (ds_launch_linear_t)
      rc |= ds_launch_dense_transpose(e->d_blob + m.dense_w.off, e->d_dense_w, m.out_ch, (int)(m.dense_w.numel / m.out_ch),
                                      e->arch.dense_total, m.temb_off, 0);
      DS_HIP(hipMemcpy(e->d_dense_b + m.temb_off, e->d_blob + m.dense_b.off, (size_t)m.dense_b.numel * 4,
                       hipMemcpyDeviceToDevice));
      break;
    case MK_ATTN:  // NIN.W is [in][out] (layers.py:678-689): packed as [out][in]
      // (the V projection is the A operand of its GEMM: it stays row-major)
      for (int i = 0; i < 4; ++i)
        rc |= repack_weight(e, m.nin_w[i], m.pk_nin[i], m.in_ch, m.in_ch, 1, 1, m.in_ch, 0, i != 2);
      // fused attention kernel: NIN.W is [in][out]; rows of the fragment-major copies of Wv / Wo = outputs ([out][in]); the
      // query and key projections are folded into one matrix and one bias vector
      for (int i = 2; i < 4; ++i) rc |= repack_frag(e, m.nin_w[i], m.pf_nin[i], m.in_ch, m.in_ch, 1, 1, m.in_ch, 0);
      if (m.pf_nin[0] >= 0 && e->cfg.dtype == DS_BF16 && e->d_attn_b) {
        hipLaunchKernelGGL(attn_fold_qk_kernel, dim3(cdiv((long)m.in_ch * m.in_ch, 256)), dim3(256), 0, 0, e->d_blob + m.nin_w[0].off,
                           e->d_blob + m.nin_w[1].off, e->d_blob + m.nin_b[0].off, (bf16_t*)(e->d_pack) + m.pf_nin[0],
                           e->d_attn_b + m.ab_off, m.in_ch);
        DS_LAUNCH_CHECK();
      }
      break;
    default: break;
  }
  return rc;
}

extern "C" int32_t diffsep_engine_create(const diffsep_model_config* cfg, const float* weights_host, int64_t n_floats,
                                         diffsep_engine** out) {
  DS_CHECK(cfg && weights_host && out, "engine_create: null argument");
  DS_CHECK(cfg->dtype == DS_F32 || cfg->dtype == DS_BF16 || cfg->dtype == DS_F32_SPLIT,
           "engine_create: dtype must be DIFFSEP_F32, DIFFSEP_BF16 or DIFFSEP_F32_SPLIT");
  diffsep_engine* e = new diffsep_engine();
  e->cfg = *cfg;
  if (cfg->dtype == DS_F32_SPLIT) { e->cfg.dtype = DS_F32; e->split = 1; }  // storage and every non-MFMA kernel: plain fp32
  // fragment-major weight copies only for the kernels this engine can dispatch to (an exact-fp32 engine: none)
  if (build_arch(e->cfg, e->arch, e->cfg.dtype == DS_BF16 ? 1 : (e->split ? 2 : 0))) { delete e; return 1; }
  const Arch& A = e->arch;
  if (n_floats != A.total) {
    ds_set_error("engine_create: weight blob has " + std::to_string(n_floats) + " floats, expected " +
                 std::to_string(A.total));
    delete e;
    return 1;
  }
  e->esz = e->cfg.dtype == DS_F32 ? 4 : 2;
  DS_HIP(hipMalloc((void**)&e->d_blob, (size_t)A.total * 4));
  DS_HIP(hipMemcpy(e->d_blob, weights_host, (size_t)A.total * 4, hipMemcpyHostToDevice));
  DS_HIP(hipMalloc((void**)&e->d_pack, (size_t)A.pack_total * e->esz + 256));
  DS_HIP(hipMemset(e->d_pack, 0, (size_t)A.pack_total * e->esz + 256));
  DS_HIP(hipMalloc((void**)&e->d_dense_w, (size_t)A.dense_total * 4 * cfg->nf * 4));
  DS_HIP(hipMalloc((void**)&e->d_dense_b, (size_t)A.dense_total * 4));
  if (A.attn_bias_total) DS_HIP(hipMalloc((void**)&e->d_attn_b, (size_t)A.attn_bias_total * 4));
  e->weight_bytes = (int64_t)A.total * 4 + (int64_t)A.pack_total * e->esz + (int64_t)A.dense_total * (4 * cfg->nf + 1) * 4;
  if (ds_build_stft_table(cfg->n_fft, &e->d_tab)) { delete e; return 1; }
  if (const char* sv = getenv("DIFFSEP_DBG_ALLOC")) e->dbg_alloc = atoi(sv) != 0;  // (read once, at creation)
  e->opts = ds_default_opts();
  DS_HIP(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
  DS_HIP(hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming));

  int rc = repack_weight(e, A.out_w, A.pk_out, A.chan_out, A.chan_in, 1, A.chan_in, 1, 0);
  for (const Module& m : A.mods) rc |= repack_module(e, m);
  if (rc) { delete e; return 1; }
  DS_HIP(hipDeviceSynchronize());
  *out = e;
  return 0;
}

extern "C" void diffsep_engine_destroy(diffsep_engine* e) {
  if (!e) return;
  drop_graph(e);
  hipFree(e->d_blob); hipFree(e->d_pack); hipFree(e->d_dense_w); hipFree(e->d_dense_b); hipFree(e->d_tab);
  if (e->d_attn_b) hipFree(e->d_attn_b);
  if (e->arena) hipFree(e->arena);
  if (e->own) hipStreamDestroy(e->own);
  if (e->ts_ev) hipEventDestroy(e->ts_ev);
  if (e->ts_pin) hipHostFree(e->ts_pin);
  if (e->ext_ev) hipEventDestroy(e->ext_ev);
  if (e->ext_pin) hipHostFree(e->ext_pin);
  if (e->ev_in) hipEventDestroy(e->ev_in);
  if (e->ev_out) hipEventDestroy(e->ev_out);
  delete e;
}
extern "C" int32_t diffsep_engine_reserve(diffsep_engine* e, int32_t B, int64_t T, void* stream) {
  DS_CHECK(e && B >= 1 && T >= 1, "reserve: bad argument");
  // size the workspace for a B x T batch now: plans of that size or smaller never reallocate afterwards
  // (hipFree / hipMalloc synchronise the whole device, i.e. every other stream's work)
  // (on the caller's stream — the null stream included: no private stream is created here, see StreamScope)
  hipStream_t st = (hipStream_t)stream;
  if (ensure_plan(e, B, T, st)) return 1;
  DS_HIP(hipStreamSynchronize(st));  // the workspace is zeroed before any other stream may use the plan
  return 0;
}
// Largest finite |value| and number of non-finite values of every activation tensor of the LAST eager forward (option
// "track_tensors" on): out[i] = {max |v|, non-finite count, H, C} for tensor i in allocation order.  Debug / test aid.
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ p, long n, int f32, unsigned* __restrict__ out) {
  float m = 0.f;
  unsigned bad = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = f32 ? reinterpret_cast<const float*>(p)[i] : h2f(reinterpret_cast<const bf16_t*>(p)[i]);
    if (v != v || fabsf(v) > 3.0e38f) ++bad; else m = fmaxf(m, fabsf(v));
  }
  atomicMax(out, __float_as_uint(m));
  if (bad) atomicAdd(out + 1, bad);
}
extern "C" int32_t diffsep_engine_debug_absmax(diffsep_engine* e, double* out, int32_t cap, int32_t* n) {
  DS_CHECK(e && n, "debug_absmax: null argument");
  *n = (int32_t)e->tracked.size();
  if (!out || cap <= 0) return 0;
  const int cnt = *n < cap ? *n : cap;
  unsigned* d = nullptr;
  DS_HIP(hipMalloc(&d, (size_t)cnt * 8 + 8));
  DS_HIP(hipMemset(d, 0, (size_t)cnt * 8 + 8));
  for (int i = 0; i < cnt; ++i) {
    const auto& t = e->tracked[i];
    long nb = (t.n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)nb), dim3(256), 0, nullptr, t.p, t.n, e->cfg.dtype == DS_F32 ? 1 : 0, d + 2 * i);
  }
  std::vector<unsigned> hbuf((size_t)cnt * 2);
  DS_HIP(hipDeviceSynchronize());
  DS_HIP(hipMemcpy(hbuf.data(), d, (size_t)cnt * 8, hipMemcpyDeviceToHost));
  hipFree(d);
  for (int i = 0; i < cnt; ++i) {
    float m;
    memcpy(&m, &hbuf[2 * i], 4);
    out[4 * i] = m; out[4 * i + 1] = hbuf[2 * i + 1]; out[4 * i + 2] = e->tracked[i].H; out[4 * i + 3] = e->tracked[i].C;
  }
  return 0;
}
extern "C" int32_t diffsep_engine_debug_arena(const diffsep_engine* e, void** base, int64_t* bytes, int64_t* fwd_base) {
  DS_CHECK(e && base && bytes && fwd_base, "debug_arena: null argument");
  *base = e->arena; *bytes = (int64_t)e->cap; *fwd_base = (int64_t)e->fwd_base;
  return 0;
}
extern "C" int64_t diffsep_engine_device_bytes(const diffsep_engine* e) { return e ? e->weight_bytes + (int64_t)e->cap : 0; }
extern "C" int32_t diffsep_engine_set_graph(diffsep_engine* e, int32_t enable) {
  DS_CHECK(e, "null engine");
  e->use_graph = enable;
  if (!enable) drop_graph(e);
  return 0;
}

extern "C" int32_t diffsep_engine_set_option(diffsep_engine* e, const char* name, int64_t value) {
  DS_CHECK(e && name, "engine_set_option: null argument");
  unsigned bit = 0;
  if (!strcmp(name, "graph_cache")) {
    DS_CHECK(value >= 1 && value <= 4096, "engine_set_option: graph_cache must be in [1, 4096]");
    e->graph_cap = (int)value;
  } else if (!strcmp(name, "ablate")) {
    e->ablate = (unsigned)value;
  } else if (!strcmp(name, "dbg_alloc")) {
    e->dbg_alloc = value != 0;
    return 0;  // (a log switch: no launch decision depends on it)
  } else if (!strcmp(name, "no_stft_fused")) {
    // stft.hip reads the PROCESS default (ds_default_opts): an engine-level value would be accepted and do nothing
    ds_set_error("engine_set_option: 'no_stft_fused' is a process-level option (diffsep_set_option / DIFFSEP_NO_STFT_FUSED)");
    return 1;
  } else if (!strcmp(name, "track_tensors")) {
    e->track_tensors = value != 0;
    e->tracked.clear();
    return 0;
  } else if (!opt_bit(name, &bit)) {
    e->opts = value ? (e->opts | bit) : (e->opts & ~bit);
  } else {
    ds_set_error(std::string("engine_set_option: unknown option '") + name + "'");
    return 1;
  }
  // the captured graphs froze the old launch decisions (and the cache may now be over its cap): forget them all — and the plan:
  // a dispatch switch may change what a forward allocates (tensors, GroupNorm accumulators), so the next call sizes it again
  DS_HIP(hipDeviceSynchronize());
  drop_graph(e);
  e->planB = -1;
  e->planT = -1;
  return 0;
}
extern "C" int64_t diffsep_engine_get_option(const diffsep_engine* e, const char* name) {
  if (!e || !name) return -1;
  unsigned bit = 0;
  if (!strcmp(name, "graph_cache")) return e->graph_cap;
  if (!strcmp(name, "graphs_cached")) return (int64_t)e->graphs.size();
  if (!strcmp(name, "ablate")) return e->ablate;
  if (!opt_bit(name, &bit)) return (e->opts & bit) ? 1 : 0;
  return -1;
}

// Per-launch timing of the MFMA contraction kernels inside the real launch sequence: between
// profile_begin and profile_end every conv/GEMM launch is bracketed by HIP events on its stream
// (graph replay is bypassed meanwhile).  Arrays have 10 entries (the last: the fused attention block): 3x3 {8x32xBN64, 8x32xBN32, 8x8xBN64},
// then the same three tiles for 1x1/GEMM, the weight-stationary 64 -> 64 3x3 kernel, the small-image 3x3 kernel, the
// register-weight 3x3 kernel.  flops = algorithmic 2*taps*Cin*Cout*H*W*B (unpadded).
extern "C" int32_t diffsep_engine_profile_begin(diffsep_engine* e) {
  DS_CHECK(e, "null engine");
  e->prof = true;
  e->prof_recs.clear();
  return 0;
}
static_assert(DS_NCLS == DIFFSEP_NUM_KERNEL_CLASSES, "header and engine agree on the number of kernel classes");
extern "C" int32_t diffsep_num_kernel_classes(void) { return DS_NCLS; }
extern "C" int32_t diffsep_engine_profile_end(diffsep_engine* e, double* flops, double* ms, int64_t* launches,
                                               double* bytes) {
  return diffsep_engine_profile_end_n(e, DS_NCLS, flops, ms, launches, bytes, nullptr);
}
extern "C" int32_t diffsep_engine_profile_end_n(diffsep_engine* e, int32_t n_classes, double* flops, double* ms,
                                                 int64_t* launches, double* bytes, int32_t* n_written) {
  DS_CHECK(e && flops && ms && launches && n_classes >= 0, "profile_end: null argument");
  DS_HIP(hipDeviceSynchronize());
  const int ncls = n_classes < DS_NCLS ? n_classes : DS_NCLS;
  if (n_written) *n_written = ncls;
  for (int i = 0; i < ncls; ++i) { flops[i] = 0; ms[i] = 0; launches[i] = 0; if (bytes) bytes[i] = 0; }
  e->prof_done.clear();
  for (auto& r : e->prof_recs) {
    float t = 0.f;
    hipEventElapsedTime(&t, r.a, r.b);
    r.ms = t;
    e->prof_done.push_back(r);
    if (r.cls >= 0 && r.cls < ncls) {  // (cls -1: the HBM-bound launches, reported through profile_records only)
      flops[r.cls] += r.flops;
      if (bytes) bytes[r.cls] += r.bytes;
      ms[r.cls] += t;
      launches[r.cls] += 1;
    }
    e->ev_pool.push_back(r.a);
    e->ev_pool.push_back(r.b);
  }
  e->prof_recs.clear();
  e->prof = false;
  return 0;
}

// The launches of the last profile_begin .. profile_end span one by one (call after profile_end): kernel instantiation
// with its template arguments, problem shape, algorithmic flops / bytes, duration.
extern "C" int32_t diffsep_engine_profile_records(diffsep_engine* e, diffsep_prof_record* out, int32_t cap, int32_t* n) {
  DS_CHECK(e && n, "profile_records: null argument");
  *n = (int32_t)e->prof_done.size();
  if (!out) return 0;
  for (int i = 0; i < *n && i < cap; ++i) {
    const auto& r = e->prof_done[i];
    diffsep_prof_record& o = out[i];
    memset(&o, 0, sizeof(o));
    snprintf(o.kernel, sizeof(o.kernel), "%s", r.kernel ? r.kernel : "");
    o.B = r.B; o.H = r.H; o.W = r.W; o.Cin = r.Cin; o.Cout = r.Cout; o.taps = r.taps; o.skip_cin = r.sCin; o.has_res = r.res;
    o.cls = r.cls; o.flops = r.flops; o.bytes = r.bytes; o.ms = r.ms;
  }
  return 0;
}

extern "C" int32_t diffsep_score_forward(diffsep_engine* e, const float* xt, const float* t, const float* mix,
                                         float* out, int32_t B, int64_t T, void* stream) {
  DS_CHECK(e && xt && t && mix && out, "score_forward: null argument");
  StreamScope sc_(e, stream);
  hipStream_t st = sc_.st;
  if (ensure_plan(e, B, T, st)) return 1;
  return score_forward_impl(e, xt, t, mix, out, B, T, st);
}

extern "C" int32_t diffsep_backbone_forward(diffsep_engine* e, const void* x, const float* t, void* y, int32_t B,
                                            int32_t W, void* stream) {
  DS_CHECK(e && x && t && y, "backbone_forward: null argument");
  DS_CHECK(W >= 64 && W % 64 == 0, "backbone_forward: W must be a positive multiple of 64");
  StreamScope sc_(e, stream);
  hipStream_t st = sc_.st;
  // plan sized through the equivalent signal length: F = W frames  <=>  T = (W-1)*hop - (n_fft-hop) + hop - 1
  const long T = (long)(W - 1) * e->cfg.hop - (e->cfg.n_fft - e->cfg.hop) + e->cfg.hop - 1;
  DS_CHECK(diffsep_padded_frames(&e->cfg, T) == W, "internal: width/length mapping");
  if (ensure_plan(e, B, T, st)) return 1;
  const int H = e->cfg.n_fft / 2 + 1;
  e->top = e->fwd_base;
  if (stats_begin(e, st)) return 1;
  Tn xin; xin.p = (void*)x; xin.C = xin.ld = e->arch.cpad_in; xin.H = H; xin.W = W;
  Tn x0 = e_tensor(e, B, H, W, e->arch.cpad_in);
  float* sc = e_f32(e, (size_t)B * e->arch.cpad_in);
  float* sh = e_f32(e, (size_t)B * e->arch.cpad_in);
  if (ds_launch_fill(sc, 2.f, (long)B * e->arch.cpad_in, st)) return 1;
  if (ds_launch_fill(sh, -1.f, (long)B * e->arch.cpad_in, st)) return 1;
  GnAff aff{sc, sh};
  if (gn_apply(e, xin, &aff, &x0, nullptr, B, 0, 0, st)) return 1;  // x = 2x - 1 (ncsnpp.py:347-349)
  Tn yo; yo.p = y; yo.C = yo.ld = e->arch.cpad_out; yo.H = H; yo.W = W;
  return net_forward(e, x0, t, yo, B, st);
}

// torch.linspace(start, end, n) in float32 (ATen RangeFactories: symmetric fill around the midpoint)
static void linspace_f32(float start, float end, int n, float* out) {
  if (n == 1) { out[0] = start; return; }
  const float step = (end - start) / (float)(n - 1);
  const int half = n / 2;
  for (int i = 0; i < n; ++i) out[i] = (i < half) ? (start + step * (float)i) : (end - step * (float)(n - 1 - i));
}

__global__ void bcast_rows_kernel(const float* __restrict__ v, float* __restrict__ out, int N, int B) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < N * B) out[i] = v[i / B];
}

static int run_nfe(diffsep_engine* e, int B, long T, hipStream_t st) {
  // one score evaluation on the resident state: (st_x, st_t, st_mix) -> st_score
  if (e->use_graph && e->warmed && !e->prof) {
    if (!e->graph_ok) {
      DS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      const int rc = score_forward_impl(e, e->st_x, e->st_t, e->st_mix, e->st_score, B, T, st);
      hipGraph_t g = nullptr;
      const hipError_t ce = hipStreamEndCapture(st, &g);
      if (rc || ce != hipSuccess || !g) {
        if (g) hipGraphDestroy(g);
        e->use_graph = 0;  // fall back to eager launches of the same kernels
        if (rc) return 1;
      } else {
        e->graph = g;
        DS_HIP(hipGraphInstantiate(&e->gexec, e->graph, nullptr, nullptr, 0));
        e->graph_ok = true;
        if ((int)e->graphs.size() >= e->graph_cap) {  // evict the least recently used plan's graph
          // (its last replay may still be running — on this stream or, if the engine was driven from another stream in an
          // earlier call, on that one: eviction is rare, wait for the device before destroying the executable)
          DS_HIP(hipDeviceSynchronize());
          while ((int)e->graphs.size() >= e->graph_cap) {
            auto lru = e->graphs.begin();
            for (auto it = e->graphs.begin(); it != e->graphs.end(); ++it)
              if (it->second.used < lru->second.used) lru = it;
            if (lru->second.x) hipGraphExecDestroy(lru->second.x);
            if (lru->second.g) hipGraphDestroy(lru->second.g);
            e->graphs.erase(lru);
          }
        }
        e->graphs[std::make_pair(B, (long)T)] = diffsep_engine::GraphRec{e->graph, e->gexec, ++e->graph_tick};
      }
    }
    if (e->graph_ok) {
      DS_HIP(hipGraphLaunch(e->gexec, st));
      return 0;
    }
  }
  const int rc = score_forward_impl(e, e->st_x, e->st_t, e->st_mix, e->st_score, B, T, st);
  e->warmed = true;  // the first eager pass also sets the kernels' LDS attributes (not capturable)
  return rc;
}

// zero the tail t >= lens[b] of [B][rows][T] rows (mixture of a mixed-length batch)
__global__ __launch_bounds__(256) void mask_tail_kernel(float* __restrict__ v, int rows, long T,
                                                        const int* __restrict__ lens) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T || t < lens[b]) return;
  for (int r = 0; r < rows; ++r) v[((long)b * rows + r) * T + t] = 0.f;
}

extern "C" int32_t diffsep_pc_sample_ex(diffsep_engine* e, const diffsep_sde_config* sde,
                                        const diffsep_sampler_config* smp, const diffsep_sampler_ext* ext,
                                        const float* mix_norm, float* out, int32_t B, int64_t T, const float* noise,
                                        uint64_t seed, const float* timesteps_host, int32_t* nfe_out, void* stream) {
  DS_CHECK(e && sde && smp && mix_norm && out, "pc_sample: null argument");
  DS_CHECK(sde->kind == DIFFSEP_SDE_MIX || sde->kind == DIFFSEP_SDE_PRIORMIX, "pc_sample: unknown SDE kind");
  DS_CHECK(sde->kind == DIFFSEP_SDE_MIX || sde->avg_len >= 1, "pc_sample: PriorMixSDE needs avg_len >= 1");
  DS_CHECK(sde->ndim == e->cfg.num_sources, "pc_sample: sde.ndim != num_sources");
  DS_CHECK(smp->N >= 1 && smp->N <= 4096, "pc_sample: N must be in [1,4096]");
  DS_CHECK(smp->predictor == DIFFSEP_PRED_REVERSE_DIFFUSION || smp->predictor == DIFFSEP_PRED_EULER_MARUYAMA ||
               smp->predictor == DIFFSEP_PRED_NONE,
           "pc_sample: predictor must be reverse_diffusion, euler_maruyama or none");
  DS_CHECK(smp->corrector == DIFFSEP_CORR_ALD2 || smp->corrector == DIFFSEP_CORR_NONE ||
               smp->corrector == DIFFSEP_CORR_ALD || smp->corrector == DIFFSEP_CORR_LANGEVIN,
           "pc_sample: corrector must be ald2, ald, langevin or none");
  DS_CHECK(smp->corrector != DIFFSEP_CORR_ALD || sde->kind == DIFFSEP_SDE_MIX,
           "pc_sample: the 'ald' corrector supports MixSDE only (sdes/correctors.py:64-67)");
  const int64_t* lengths = ext ? ext->lengths_host : nullptr;
  const uint64_t* seeds = ext ? ext->seeds_host : nullptr;
  diffsep_engine* tail = (ext && (ext->tail_steps > 0 || ext->head_steps > 0)) ? ext->tail_engine : nullptr;
  const int tail_steps = tail ? ext->tail_steps : 0;
  const int head_steps = tail ? ext->head_steps : 0;
  if (tail) {
    DS_CHECK(tail != e, "pc_sample: the tail engine must be a different engine");
    diffsep_model_config a = e->cfg, b2 = tail->cfg;
    a.dtype = b2.dtype = 0;
    DS_CHECK(memcmp(&a, &b2, sizeof(a)) == 0, "pc_sample: the tail engine must have the same architecture");
  }
  DS_CHECK(!lengths || smp->corrector != DIFFSEP_CORR_LANGEVIN,
           "pc_sample: the 'langevin' corrector couples the batch entries; it cannot run on a mixed-length batch");
  DS_CHECK(!seeds || !noise, "pc_sample: per-utterance seeds are for device noise (noise == NULL)");
  if (lengths) {
    const int Wp = diffsep_padded_frames(&e->cfg, T);
    for (int b = 0; b < B; ++b) {
      DS_CHECK(lengths[b] >= 1 && lengths[b] <= T, "pc_sample: utterance length outside [1, T]");
      DS_CHECK(diffsep_padded_frames(&e->cfg, lengths[b]) == Wp,
               "pc_sample: every utterance of a mixed-length batch must have the padded frame count of T");
    }
  }
  StreamScope sc_(e, stream);
  hipStream_t st = sc_.st;
  const int S = e->cfg.num_sources, N = smp->N;
  const int csteps = smp->corrector == DIFFSEP_CORR_NONE ? 0 : smp->corrector_steps;
  if (ensure_plan(e, B, T, st)) return 1;
  if (tail && ensure_plan(tail, B, T, st)) return 1;
  const size_t nst = (size_t)B * S * T;
  SdeP sp{sde->kind, sde->ndim, sde->d_lambda, sde->sigma_min, sde->sigma_max};
  // time steps -> device rows [N][B]
  std::vector<float> ts(N);
  if (timesteps_host) for (int i = 0; i < N; ++i) ts[i] = timesteps_host[i];
  else linspace_f32(1.0f, smp->eps, N, ts.data());
  if (e->ts_dev != ts || e->ts_B != B) {  // (same schedule as the last call: the device rows are already there)
    // through a pinned staging buffer, stream-ordered: a pageable hipMemcpyAsync + stream sync was measured waiting for
    // the work of OTHER streams (200 ms per new utterance length with four samplers in flight)
    const size_t nrow = (size_t)N * B;
    if (e->ts_ev_rec) DS_HIP(hipEventSynchronize(e->ts_ev));  // the previous upload has left the staging buffer
    if (nrow > e->ts_pin_cap) {
      if (e->ts_pin) DS_HIP(hipHostFree(e->ts_pin));
      e->ts_pin = nullptr;
      e->ts_pin_cap = 0;
      const size_t cap = nrow < 4096 ? 4096 : nrow;
      DS_HIP(hipHostMalloc((void**)&e->ts_pin, cap * sizeof(float), hipHostMallocDefault));
      e->ts_pin_cap = cap;
    }
    if (!e->ts_ev) DS_HIP(hipEventCreateWithFlags(&e->ts_ev, hipEventDisableTiming));
    for (int i = 0; i < N; ++i) for (int b = 0; b < B; ++b) e->ts_pin[(size_t)i * B + b] = ts[i];
    DS_HIP(hipMemcpyAsync(e->st_ts, e->ts_pin, nrow * 4, hipMemcpyHostToDevice, st));
    DS_HIP(hipEventRecord(e->ts_ev, st));
    e->ts_ev_rec = true;
    e->ts_dev = ts;
    e->ts_B = B;
  }
  DS_HIP(hipMemcpyAsync(e->st_mix, mix_norm, (size_t)B * T * 4, hipMemcpyDeviceToDevice, st));
  const int* lens = nullptr;
  if (lengths || seeds) {  // per-utterance lengths / seeds -> device (pinned staging, stream-ordered)
    const size_t need = (size_t)B * 16;
    if (e->ext_ev_rec) DS_HIP(hipEventSynchronize(e->ext_ev));
    if (need > e->ext_pin_cap) {
      if (e->ext_pin) DS_HIP(hipHostFree(e->ext_pin));
      e->ext_pin = nullptr;
      e->ext_pin_cap = 0;
      const size_t cap = need < 4096 ? 4096 : need;
      DS_HIP(hipHostMalloc((void**)&e->ext_pin, cap, hipHostMallocDefault));
      e->ext_pin_cap = cap;
    }
    if (!e->ext_ev) DS_HIP(hipEventCreateWithFlags(&e->ext_ev, hipEventDisableTiming));
    unsigned long long* ps = reinterpret_cast<unsigned long long*>(e->ext_pin);
    int* pl = reinterpret_cast<int*>(e->ext_pin + (size_t)B * 8);
    for (int b = 0; b < B; ++b) {
      ps[b] = seeds ? seeds[b] : seed + 0x9E3779B97F4A7C15ull * (unsigned long long)b;  // (b = 0: the B = 1 stream of `seed`)
      pl[b] = lengths ? (int)lengths[b] : (int)T;
    }
    DS_HIP(hipMemcpyAsync(e->st_seeds, ps, (size_t)B * 8, hipMemcpyHostToDevice, st));
    DS_HIP(hipMemcpyAsync(e->st_lens, pl, (size_t)B * 4, hipMemcpyHostToDevice, st));
    DS_HIP(hipEventRecord(e->ext_ev, st));
    e->ext_ev_rec = true;
    if (lengths) {
      lens = e->st_lens;
      hipLaunchKernelGGL(mask_tail_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, st, e->st_mix, 1, (long)T, lens);
      DS_LAUNCH_CHECK();
    }
  }
  const bool batch_rng = !noise && (seeds || lengths);

  long draw = 0;
  auto next_noise = [&](const float** z) -> int {
    if (noise) { *z = noise + (size_t)draw * nst; }
    else {
      if (batch_rng) {
        if (hbm_launch_prof(e, st, "randn (Philox4x32-10 + Box-Muller)", 4.0 * nst, B, 1, (int)T, S, [&]() {
              return ds_launch_randn_batch(e->st_noise, B, S, T, (const uint64_t*)e->st_seeds, e->st_lens, (uint64_t)draw, st);
            }))
          return 1;
      } else if (hbm_launch_prof(e, st, "randn (Philox4x32-10 + Box-Muller)", 4.0 * nst, B, 1, (int)T, S, [&]() {
                   return ds_launch_randn(e->st_noise, (long)nst, seed, (uint64_t)draw, st);
                 })) {
        return 1;
      }
      *z = e->st_noise;
    }
    ++draw;
    return 0;
  };
  const float* z = nullptr;
  if (next_noise(&z)) return 1;
  const float* smix = nullptr;
  if (sde->kind == DIFFSEP_SDE_PRIORMIX) {  // per-sample noise scale from the mixture envelope (sdes.py:477-489)
    if (ds_launch_sigma_mix(e->st_mix, e->st_smix, B, T, sde->avg_len, st)) return 1;
    smix = e->st_smix;
  }
  if (ds_launch_sde_prior(sp, e->st_mix, z, e->st_x, B, S, T, smix, st, lens)) return 1;
  DS_HIP(hipMemcpyAsync(e->st_xm, e->st_x, nst * 4, hipMemcpyDeviceToDevice, st));
  // one score evaluation of reverse step i: on this engine, or — in the last tail_steps steps — on the tail engine
  // (state and time step copied over, the score read from there)
  bool tail_ready = false;
  const float* score = e->st_score;
  auto eval_score = [&](int i) -> int {
    if (tail && (i >= N - tail_steps || i < head_steps)) {
      if (!tail_ready) {
        DS_HIP(hipMemcpyAsync(tail->st_mix, e->st_mix, (size_t)B * T * 4, hipMemcpyDeviceToDevice, st));
        tail_ready = true;
      }
      DS_HIP(hipMemcpyAsync(tail->st_x, e->st_x, nst * 4, hipMemcpyDeviceToDevice, st));
      DS_HIP(hipMemcpyAsync(tail->st_t, e->st_t, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
      score = tail->st_score;
      return run_nfe(tail, B, T, st);
    }
    score = e->st_score;
    return run_nfe(e, B, T, st);
  };
  int nfe = 0;
  for (int i = 0; i < N; ++i) {
    DS_HIP(hipMemcpyAsync(e->st_t, e->st_ts + (size_t)i * B, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
    for (int k = 0; k < csteps; ++k) {
      if (eval_score(i)) return 1;
      ++nfe;
      if (next_noise(&z)) return 1;
      if (smp->corrector == DIFFSEP_CORR_LANGEVIN) {
        if (ds_launch_langevin(smp->snr, e->st_x, score, z, e->st_x, e->st_xm, B, (long)S * T, e->st_lang, st))
          return 1;
      } else if (hbm_launch_prof(e, st, "sde_corrector (ald2 update)", 4.0 * nst * 5.0, B, 1, (int)T, S, [&]() {  // x, score, z in; x, x_mean out
                   return ds_launch_sde_corrector(sp, smp->snr, e->st_x, e->st_t, score, z, e->st_x, e->st_xm, B, S, T,
                                                  smix, smp->corrector == DIFFSEP_CORR_ALD ? 1 : 0, st, lens);
                 })) {
        return 1;
      }
    }
    if (smp->predictor != DIFFSEP_PRED_NONE) {
      // euler_maruyama (sdes/predictors.py:39-52) takes x + f*dt with the reverse drift f = drift - g^2 score and
      // noise g sqrt(dt): algebraically the reverse_diffusion step (dt = 1/N, G = g sqrt(dt)) — one kernel for both
      if (eval_score(i)) return 1;
      ++nfe;
      if (next_noise(&z)) return 1;
      if (hbm_launch_prof(e, st, "sde_predictor (reverse-diffusion update)", 4.0 * nst * 5.0, B, 1, (int)T, S, [&]() {
            return ds_launch_sde_predictor(sp, N, e->st_x, e->st_t, score, z, e->st_x, e->st_xm, B, S, T, smix, 0, st, lens);
          }))
        return 1;
    } else {
      DS_HIP(hipMemcpyAsync(e->st_xm, e->st_x, nst * 4, hipMemcpyDeviceToDevice, st));
    }
  }
  DS_HIP(hipMemcpyAsync(out, smp->denoise ? e->st_xm : e->st_x, nst * 4, hipMemcpyDeviceToDevice, st));
  if (nfe_out) *nfe_out = N * (csteps + 1);
  (void)nfe;
  return 0;
}

extern "C" int32_t diffsep_pc_sample(diffsep_engine* e, const diffsep_sde_config* sde, const diffsep_sampler_config* smp,
                                     const float* mix_norm, float* out, int32_t B, int64_t T, const float* noise,
                                     uint64_t seed, const float* timesteps_host, int32_t* nfe_out, void* stream) {
  return diffsep_pc_sample_ex(e, sde, smp, nullptr, mix_norm, out, B, T, noise, seed, timesteps_host, nfe_out, stream);
}

// ------------------------------------------------------------------ unit entry points
// The time embedding of NCSNpp.forward (ncsnpp.py:324-343): GaussianFourierProjection(log t) -> Linear -> SiLU -> Linear, with
// the kernels net_forward launches.  temb [B][4 nf]; workspace >= B * 6 nf floats.
extern "C" int32_t diffsep_time_embedding(const float* t, const float* fourier_w, const float* w1, const float* b1,
                                          const float* w2, const float* b2, float* temb, int32_t B, int32_t nf,
                                          void* workspace, int64_t workspace_bytes, void* stream) {
  DS_CHECK(t && fourier_w && w1 && b1 && w2 && b2 && temb && workspace, "time_embedding: null pointer");
  DS_CHECK(B >= 1 && nf >= 8 && nf % 8 == 0, "time_embedding: bad B / nf");
  DS_CHECK(workspace_bytes >= (int64_t)B * 6 * nf * 4, "time_embedding: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* emb = (float*)workspace;
  float* t1 = emb + (size_t)B * 2 * nf;
  if (ds_launch_fourier(t, fourier_w, emb, B, nf, st)) return 1;
  if (ds_launch_linear(emb, w1, b1, t1, B, 2 * nf, 4 * nf, 0, st)) return 1;
  return ds_launch_linear(t1, w2, b2, temb, B, 4 * nf, 4 * nf, 1, st);
}

extern "C" int32_t diffsep_upfirdn2d(const void* x, void* y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx,
                                     int32_t ldy, int32_t up, int32_t dtype, void* stream) {
  DS_CHECK(x && y, "upfirdn2d: null pointer");
  return ds_launch_gn_apply(x, ldx, nullptr, nullptr, C, nullptr, 0, y, ldy, B, H, W, 0, up ? 1 : 2, dtype,
                            (hipStream_t)stream);
}

extern "C" int32_t diffsep_groupnorm_act(const void* x, const float* gamma, const float* beta, void* y, void* xr,
                                         int32_t B, int32_t H, int32_t W, int32_t C, int32_t ldx, int32_t ldy,
                                         int32_t ldxr, int32_t groups, float eps, int32_t act, int32_t resample,
                                         int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  DS_CHECK(x && y && workspace, "groupnorm: null pointer");
  const long wsb = (ds_gn_workspace_bytes(B, H, W, C) + 255) & ~255L;
  DS_CHECK(workspace_bytes >= wsb + 2L * B * C * 4, "groupnorm: workspace too small");
  float* scale = (float*)((char*)workspace + wsb);
  float* shift = scale + (long)B * C;
  hipStream_t st = (hipStream_t)stream;
  if (ds_launch_gn_stats(x, ldx, nullptr, 0, C, B, H, W, C, groups, eps, gamma, beta, workspace, scale, shift, dtype, st))
    return 1;
  return ds_launch_gn_apply(x, ldx, scale, shift, C, y, ldy, xr, ldxr, B, H, W, act, resample, dtype, st);
}

extern "C" int32_t diffsep_conv2d(const void* x, const void* w, const float* bias, const float* bias_b, const void* res,
                                  void* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                                  int32_t ldx, int32_t ldr, int32_t ldy, float out_scale, int32_t dtype, void* stream) {
  DS_CHECK(ksize == 1 || ksize == 3, "conv2d: ksize must be 1 or 3");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.opts = ds_default_opts();
  a.x = x; a.x_bs = (long)H * W * ldx; a.ldx = ldx;
  a.w = w; a.w_bs = 0;
  a.bias = bias; a.bias_b = bias_b; a.bias_b_ld = Cout; a.bias_mode = 0;
  a.res = res; a.res_bs = (long)H * W * ldr; a.ldr = ldr;
  a.out_scale = out_scale;
  a.y = y; a.y_bs = (long)H * W * ldy; a.ldy = ldy;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.taps = ksize == 3 ? 9 : 1;
  a.dtype = dtype == DS_F32_SPLIT ? DS_F32 : dtype; a.split = dtype == DS_F32_SPLIT;
  return ds_launch_conv(a, (hipStream_t)stream);
}

extern "C" int32_t diffsep_groupnorm_stats(const void* x, const void* x2, int32_t C1, const float* gamma,
                                           const float* beta, float* scale, float* shift, int32_t B, int32_t H,
                                           int32_t W, int32_t C, int32_t ldx, int32_t ldx2, int32_t groups, float eps,
                                           int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  DS_CHECK(x && scale && shift && workspace, "groupnorm_stats: null pointer");
  DS_CHECK(workspace_bytes >= ds_gn_workspace_bytes(B, H, W, C), "groupnorm_stats: workspace too small");
  return ds_launch_gn_stats(x, ldx, x2, ldx2, x2 ? C1 : C, B, H, W, C, groups, eps, gamma, beta, workspace, scale, shift,
                            dtype, (hipStream_t)stream);
}

extern "C" int32_t diffsep_conv2d_fused(const void* x, const void* x2, int32_t C1, const float* gn_scale,
                                        const float* gn_shift, int32_t gn_act, const void* w, const float* bias,
                                        const float* bias_b, const void* res, void* y, int32_t B, int32_t H, int32_t W,
                                        int32_t Cin, int32_t Cout, int32_t ksize, int32_t ldx, int32_t ldx2,
                                        int32_t ldr, int32_t ldy, float out_scale, int32_t dtype, int64_t* stats,
                                        int32_t w_chunk, const int64_t* gn_acc1, const int64_t* gn_acc2,
                                        const float* gn_gamma, const float* gn_beta, int32_t gn_groups, void* stream) {
  DS_CHECK(ksize == 1 || ksize == 3, "conv2d: ksize must be 1 or 3");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.opts = ds_default_opts();
  a.stats_acc = (long long*)stats;
  if (gn_acc1) {
    DS_CHECK(gn_groups > 0 && Cin % gn_groups == 0, "conv2d: bad GroupNorm group count");
    a.gn_acc1 = (const long long*)gn_acc1; a.gn_acc2 = (const long long*)gn_acc2; a.gn_gamma = gn_gamma;
    a.gn_beta = gn_beta; a.gn_groups = gn_groups; a.gn_eps = 1e-6f;
    a.gn_inv_count = (float)(1.0 / ((double)H * W * (Cin / gn_groups)));
  }
  a.w_chunked = w_chunk;
  a.x = x; a.x_bs = (long)H * W * ldx; a.ldx = ldx;
  a.x2 = x2; a.x2_bs = (long)H * W * ldx2; a.ldx2 = ldx2; a.C1 = C1;
  a.gn_scale = gn_scale; a.gn_shift = gn_shift; a.gn_act = gn_act;
  a.w = w; a.w_bs = 0;
  a.bias = bias; a.bias_b = bias_b; a.bias_b_ld = Cout; a.bias_mode = 0;
  a.res = res; a.res_bs = (long)H * W * ldr; a.ldr = ldr;
  a.out_scale = out_scale;
  a.y = y; a.y_bs = (long)H * W * ldy; a.ldy = ldy;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.taps = ksize == 3 ? 9 : 1;
  a.dtype = dtype == DS_F32_SPLIT ? DS_F32 : dtype; a.split = dtype == DS_F32_SPLIT;
  return ds_launch_conv(a, (hipStream_t)stream);
}

// Unit entry of the streamed-weight 3x3 kernel (conv3x3_sw.hip), whatever the dispatch would have chosen for the shape: dense
// NHWC tensors, weights already in the fragment-major order of diffsep_frag_index (include/diffsep_hip.h).
extern "C" int32_t diffsep_conv3x3_streamed(const void* x, const void* x2, int32_t C1, const float* gn_scale,
                                            const float* gn_shift, const void* w_frag, const float* bias,
                                            const float* bias_b, const void* sx, const void* sx2, int32_t sC1,
                                            int32_t sCin, const void* sw_frag, void* y, int32_t B, int32_t H, int32_t W,
                                            int32_t Cin, int32_t Cout, float out_scale, int32_t dtype, int64_t* stats,
                                            const void* res, const void* ident_frag, void* stream) {
  DS_CHECK(x && w_frag && y, "conv3x3_streamed: null pointer");
  DS_CHECK(!res || (ident_frag && !sx), "conv3x3_streamed: a residual needs the identity copy and no skip");
  DS_CHECK(B > 0 && H > 0 && W > 0, "conv3x3_streamed: empty problem");
  DS_CHECK(!x2 || (C1 > 0 && C1 < Cin), "conv3x3_streamed: bad concat split");
  DS_CHECK(!sx || (sw_frag && sCin > 0 && (!sx2 || (sC1 > 0 && sC1 < sCin))), "conv3x3_streamed: bad skip operands");
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.opts = ds_default_opts();
  a.stats_acc = (long long*)stats;
  const int c1 = x2 ? C1 : Cin;
  a.x = x; a.x_bs = (long)H * W * c1; a.ldx = c1;
  a.x2 = x2; a.x2_bs = (long)H * W * (Cin - c1); a.ldx2 = Cin - c1; a.C1 = x2 ? C1 : 0;
  a.gn_scale = gn_scale; a.gn_shift = gn_shift; a.gn_act = gn_scale ? 1 : 0;
  a.w = w_frag; a.w_frag = w_frag; a.w_bs = 0;
  a.bias = bias; a.bias_b = bias_b; a.bias_b_ld = Cout; a.bias_mode = 0;
  if (sx) {
    const int s1 = sx2 ? sC1 : sCin;
    a.sx = sx; a.sx_bs = (long)H * W * s1; a.ldsx = s1;
    a.sx2 = sx2; a.sx2_bs = (long)H * W * (sCin - s1); a.ldsx2 = sCin - s1; a.sC1 = sx2 ? sC1 : 0; a.sCin = sCin;
    a.sw = sw_frag; a.sw_frag = sw_frag;
  }
  a.out_scale = out_scale;
  a.y = y; a.y_bs = (long)H * W * Cout; a.ldy = Cout;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.taps = 9;
  a.dtype = dtype == DS_F32_SPLIT ? DS_F32 : dtype; a.split = dtype == DS_F32_SPLIT;
  DS_CHECK((long)H * W * (Cin > Cout ? Cin : Cout) * 4 < 2147483647L, "conv3x3_streamed: image too large for 32-bit buffer offsets");
  a.res = res; a.res_bs = (long)H * W * Cout; a.ldr = Cout; a.ident_frag = ident_frag;
  if (a.split) {  // fp32 tensors, hi / lo fragment copies: conv3x3_sws.hip
    DS_CHECK(ds_conv_sws_supported(a), "conv3x3_streamed: shape outside the split kernel's instantiations (Cout = 64 / 128 / 256, Cin = 64 .. 256 "
                                       "by 64, W % 32 == 0, H % 8 == 0; skip / residual channels 64 .. 256 by 64 behind GroupNorm; raw input: Cin <= 128)");
    return ds_launch_conv_sws(a, (hipStream_t)stream);
  }
  DS_CHECK(ds_conv_sw_supported(a), "conv3x3_streamed: shape outside the kernel's instantiations (16-bit, Cout = 128 / 256, Cin = 64 .. 256 "
                                    "by 64, W % 32 == 0, H % 4 == 0; a skip needs GroupNorm and Cin = 128; raw input: Cin <= 128; Cout = 64: Cin = 192)");
  return ds_launch_conv_sw(a, (hipStream_t)stream);
}
extern "C" int64_t diffsep_frag_index(int32_t cout, int32_t tap, int32_t cin, int32_t taps, int32_t Cout) {
  return ds_rw_frag_index(cout, tap, cin, taps, Cout);
}
extern "C" int64_t diffsep_frag_index_split(int32_t cout, int32_t tap, int32_t cin, int32_t taps, int32_t Cout, int32_t plane) {
  return ds_sws_frag_index(cout, tap, cin, taps, Cout, plane);
}

extern "C" int32_t diffsep_conv2d_chunk(int32_t ksize, int32_t dtype) { return ds_conv_chunk(ksize == 3 ? 9 : 1, dtype); }

extern "C" int32_t diffsep_attention(const void* q, const void* k, const void* vt, void* o, int32_t B, int32_t L,
                                     int32_t C, int32_t ld, int32_t dtype, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
  DS_CHECK(q && k && vt && o && workspace, "attention: null pointer");
  DS_CHECK(ld == C, "attention: q/k must be dense [B,L,C] (ld == C)");
  const int split = dtype == DS_F32_SPLIT;
  if (split) dtype = DS_F32;
  const int Lp = rup8(L), esz = dtype == DS_F32 ? 4 : 2;
  const long one = (((long)B * L * Lp * esz) + 255) & ~255L;
  DS_CHECK(workspace_bytes >= 2 * one, "attention: workspace too small");
  return attention_core(q, k, vt, o, B, L, C, ld, ld, workspace, (char*)workspace + one, dtype, (hipStream_t)stream, split);
}

// ---- one ResnetBlockBigGANpp / AttnBlockpp through the ENGINE's block code (res_block / attn_block above: folded
// Conv_2, GroupNorm from the producer's accumulators, fused FIR resampling, MFMA attention), on caller-supplied
// parameters: the parity tests check the composition against the reference blocks in isolation (layerspp.py:291-323,
// 76-92).  A throw-away one-module engine is built per call (test path, not a hot path).
struct MiniEngine {
  diffsep_engine* e = nullptr;
  ~MiniEngine() { if (e) diffsep_engine_destroy(e); }
};
static int mini_engine_init(MiniEngine& me, int dtype, int temb_dim, const float* params_host, int64_t n_floats) {
  diffsep_engine* e = me.e;
  const Arch& A = e->arch;
  if (n_floats != A.total) {
    ds_set_error("block_forward: parameter blob has " + std::to_string(n_floats) + " floats, expected " +
                 std::to_string(A.total));
    return 1;
  }
  e->esz = dtype == DS_F32 ? 4 : 2;
  e->opts = ds_default_opts();
  DS_HIP(hipMalloc((void**)&e->d_blob, (size_t)A.total * 4));
  DS_HIP(hipMemcpy(e->d_blob, params_host, (size_t)A.total * 4, hipMemcpyHostToDevice));
  DS_HIP(hipMalloc((void**)&e->d_pack, (size_t)A.pack_total * e->esz + 256));
  DS_HIP(hipMemset(e->d_pack, 0, (size_t)A.pack_total * e->esz + 256));
  DS_HIP(hipMalloc((void**)&e->d_dense_w, (size_t)(A.dense_total + 1) * (temb_dim + 1) * 4));
  DS_HIP(hipMalloc((void**)&e->d_dense_b, (size_t)(A.dense_total + 1) * 4));
  if (A.attn_bias_total) DS_HIP(hipMalloc((void**)&e->d_attn_b, (size_t)A.attn_bias_total * 4));
  for (const Module& m : A.mods)
    if (repack_module(e, m)) return 1;
  DS_HIP(hipDeviceSynchronize());
  return 0;
}
template <typename F>
static int mini_engine_run(diffsep_engine* e, hipStream_t st, F&& body) {
  e->fwd_base = 0;
  e->dry = true;
  e->top = 0;
  if (stats_begin(e, st)) return 1;
  if (body()) { e->dry = false; return 1; }
  e->dry = false;
  const size_t need = e->top + e->stats_need + 8192;
  DS_HIP(hipMalloc((void**)&e->arena, need));
  e->cap = need;
  DS_HIP(hipMemsetAsync(e->arena, 0, need, st));
  e->top = 0;
  if (stats_begin(e, st)) return 1;
  if (body()) return 1;
  DS_HIP(hipStreamSynchronize(st));  // the arena is freed with the engine when the caller returns
  return 0;
}

extern "C" int32_t diffsep_resblock_forward(int32_t in_ch, int32_t out_ch, int32_t up, int32_t down, int32_t temb_dim,
                                            int32_t dtype, const float* params_host, int64_t n_floats, const void* x,
                                            const float* temb, void* y, int32_t B, int32_t H, int32_t W, void* stream) {
  DS_CHECK(params_host && x && temb && y, "resblock_forward: null pointer");
  DS_CHECK(dtype == DS_F32 || dtype == DS_BF16, "resblock_forward: bad dtype");
  DS_CHECK(in_ch % 8 == 0 && out_ch % 8 == 0 && in_ch >= 8 && out_ch >= 8, "resblock_forward: channels must be multiples of 8");
  DS_CHECK(temb_dim >= 4 && temb_dim % 4 == 0, "resblock_forward: temb_dim must be a multiple of 4");
  DS_CHECK(!(up && down) && B >= 1 && H >= 1 && W >= 1 && (!down || (H % 2 == 0 && W % 2 == 0)), "resblock_forward: bad shape");
  MiniEngine me;
  me.e = new diffsep_engine();
  diffsep_engine* e = me.e;
  memset(&e->cfg, 0, sizeof(e->cfg));
  e->cfg.dtype = dtype;
  e->cfg.nf = temb_dim / 4;
  {
    ArchBuilder b(e->arch);
    b.res(in_ch, out_ch, up != 0, down != 0, temb_dim);
  }
  if (mini_engine_init(me, dtype, temb_dim, params_host, n_floats)) return 1;
  const Module& m = e->arch.mods[0];
  hipStream_t st = (hipStream_t)stream;
  const int Ho = up ? 2 * H : (down ? H / 2 : H), Wo = up ? 2 * W : (down ? W / 2 : W);
  return mini_engine_run(e, st, [&]() -> int {
    float* proj = e_f32(e, (size_t)B * e->arch.dense_total);
    // Dense_0(act(temb))  layerspp.py:311-312
    if (!e->dry && ds_launch_linear_t(temb, e->d_dense_w, e->d_dense_b, proj, B, temb_dim, e->arch.dense_total, 1, st)) return 1;
    Tn xin;
    xin.p = const_cast<void*>(x); xin.C = xin.ld = in_ch; xin.H = H; xin.W = W;
    Tn out;
    if (res_block(e, m, xin, proj, B, out, st)) return 1;
    if (!e->dry)
      DS_HIP(hipMemcpyAsync(y, out.p, (size_t)B * Ho * Wo * out_ch * e->esz, hipMemcpyDeviceToDevice, st));
    return 0;
  });
}

extern "C" int32_t diffsep_attnblock_forward(int32_t channels, int32_t dtype, const float* params_host, int64_t n_floats,
                                             const void* x, void* y, int32_t B, int32_t H, int32_t W, void* stream) {
  DS_CHECK(params_host && x && y, "attnblock_forward: null pointer");
  DS_CHECK(dtype == DS_F32 || dtype == DS_BF16, "attnblock_forward: bad dtype");
  DS_CHECK(channels % 8 == 0 && channels >= 8 && B >= 1 && H >= 1 && W >= 1, "attnblock_forward: bad shape");
  MiniEngine me;
  me.e = new diffsep_engine();
  diffsep_engine* e = me.e;
  memset(&e->cfg, 0, sizeof(e->cfg));
  e->cfg.dtype = dtype;
  e->cfg.nf = 8;
  {
    ArchBuilder b(e->arch);
    b.attn(channels);
  }
  if (mini_engine_init(me, dtype, 4, params_host, n_floats)) return 1;
  const Module& m = e->arch.mods[0];
  hipStream_t st = (hipStream_t)stream;
  return mini_engine_run(e, st, [&]() -> int {
    Tn xin;
    xin.p = const_cast<void*>(x); xin.C = xin.ld = channels; xin.H = H; xin.W = W;
    Tn out;
    if (attn_block(e, m, xin, B, out, st)) return 1;
    if (!e->dry)
      DS_HIP(hipMemcpyAsync(y, out.p, (size_t)B * H * W * channels * e->esz, hipMemcpyDeviceToDevice, st));
    return 0;
  });
}

static float* g_tab = nullptr;
static int g_tab_n = 0;
static int unit_tab(int n_fft, float** tab) {
  if (g_tab_n != n_fft) {
    if (g_tab) hipFree(g_tab);
    g_tab = nullptr;
    g_tab_n = 0;
    if (ds_build_stft_table(n_fft, &g_tab)) return 1;
    g_tab_n = n_fft;
  }
  *tab = g_tab;
  return 0;
}

extern "C" int32_t diffsep_stft_pack(const float* xt, const float* mix, void* y, int32_t B, int32_t S, int64_t T,
                                     int32_t n_fft, int32_t hop, float exponent, float factor, int32_t W, int32_t Cpad,
                                     int32_t centered_shift, int32_t dtype, void* workspace, int64_t workspace_bytes,
                                     void* stream) {
  DS_CHECK(xt && mix && y && workspace, "stft_pack: null pointer");
  DS_CHECK(workspace_bytes >= ds_stft_workspace_bytes(B, S, T, n_fft, hop), "stft_pack: workspace too small");
  float* tab;
  if (unit_tab(n_fft, &tab)) return 1;
  return ds_launch_stft_pack(xt, mix, y, B, S, T, n_fft, hop, exponent, factor, W, Cpad, centered_shift, dtype, tab,
                             (float*)workspace, (hipStream_t)stream);
}

extern "C" int32_t diffsep_istft_unpack(const void* x, float* out, int32_t B, int32_t S, int64_t T, int32_t n_fft,
                                        int32_t hop, float exponent, float factor, int32_t W, int32_t Cpad,
                                        int32_t dtype, void* workspace, int64_t workspace_bytes, void* stream) {
  DS_CHECK(x && out && workspace, "istft_unpack: null pointer");
  DS_CHECK(workspace_bytes >= ds_istft_workspace_bytes(B, S, T, n_fft, hop), "istft_unpack: workspace too small");
  float* tab;
  if (unit_tab(n_fft, &tab)) return 1;
  return ds_launch_istft(x, out, B, S, T, n_fft, hop, exponent, factor, W, Cpad, dtype, tab, (float*)workspace,
                         (hipStream_t)stream);
}

static SdeP to_sdep(const diffsep_sde_config* s) { return SdeP{s->kind, s->ndim, s->d_lambda, s->sigma_min, s->sigma_max}; }

extern "C" int32_t diffsep_sde_sigma_mix(const float* mix, float* sigma_mix, int32_t B, int64_t T, int32_t avg_len,
                                         void* stream) {
  DS_CHECK(mix && sigma_mix, "sde_sigma_mix: null pointer");
  return ds_launch_sigma_mix(mix, sigma_mix, B, T, avg_len, (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_prior(const diffsep_sde_config* sde, const float* y, const float* z, float* x, int32_t B,
                                     int32_t S, int64_t T, const float* sigma_mix, void* stream) {
  DS_CHECK(sde && y && z && x, "sde_prior: null pointer");
  return ds_launch_sde_prior(to_sdep(sde), y, z, x, B, S, T, sigma_mix, (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_corrector_update(const diffsep_sde_config* sde, float snr, const float* x, const float* t,
                                                const float* score, const float* z, float* x_out, float* x_mean_out,
                                                int32_t B, int32_t S, int64_t T, const float* sigma_mix, int32_t variant,
                                                void* stream) {
  DS_CHECK(sde && x && t && score && x_out, "sde_corrector_update: null pointer");
  return ds_launch_sde_corrector(to_sdep(sde), snr, x, t, score, z, x_out, x_mean_out, B, S, T, sigma_mix, variant,
                                 (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_predictor_update(const diffsep_sde_config* sde, int32_t N, const float* x, const float* t,
                                                const float* score, const float* z, float* x_out, float* x_mean_out,
                                                int32_t B, int32_t S, int64_t T, const float* sigma_mix,
                                                int32_t probability_flow, void* stream) {
  DS_CHECK(sde && x && t && score && x_out, "sde_predictor_update: null pointer");
  return ds_launch_sde_predictor(to_sdep(sde), N, x, t, score, z, x_out, x_mean_out, B, S, T, sigma_mix,
                                 probability_flow, (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_coefficients(const diffsep_sde_config* sde, const float* x, const float* t,
                                            const float* sigma_mix, float* drift_out, float* diffusion_out, int32_t B,
                                            int32_t S, int64_t T, float f_scale, float g_scale, void* stream) {
  DS_CHECK(sde && x && t && drift_out && diffusion_out, "sde_coefficients: null pointer");
  return ds_launch_sde_coeff(to_sdep(sde), x, t, sigma_mix, drift_out, diffusion_out, B, S, T, f_scale, g_scale,
                             (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_mean(const diffsep_sde_config* sde, const float* x0, const float* t, float* mean_out,
                                    int32_t B, int32_t S, int64_t T, void* stream) {
  DS_CHECK(sde && x0 && t && mean_out, "sde_mean: null pointer");
  return ds_launch_sde_mean(to_sdep(sde), x0, t, mean_out, B, S, T, (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_std(const diffsep_sde_config* sde, const float* t, const float* sigma_mix, float* std_out,
                                   int32_t B, int32_t S, int64_t T, void* stream) {
  DS_CHECK(sde && t && std_out, "sde_std: null pointer");
  return ds_launch_sde_std(to_sdep(sde), t, sigma_mix, std_out, B, S, T, (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_mult_std(const float* std, const float* x, float* out, int32_t B, int32_t S, int64_t T,
                                        int32_t per_sample, void* stream) {
  DS_CHECK(std && x && out, "sde_mult_std: null pointer");
  return ds_launch_sde_mult_std(std, x, out, B, S, T, per_sample, (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_reverse_drift(const float* f, const float* G, const float* score, float* rev_f_out,
                                             int32_t B, int64_t n_per_batch, int32_t g_full, int32_t probability_flow,
                                             void* stream) {
  DS_CHECK(f && G && score && rev_f_out, "sde_reverse_drift: null pointer");
  return ds_launch_sde_reverse(f, G, score, rev_f_out, B, n_per_batch, g_full, probability_flow, (hipStream_t)stream);
}
extern "C" int32_t diffsep_sde_langevin_update(float snr, const float* x, const float* score, const float* z,
                                               float* x_out, float* x_mean_out, int32_t B, int64_t n_per_batch,
                                               void* workspace, int64_t workspace_bytes, void* stream) {
  DS_CHECK(x && score && z && x_out && workspace, "sde_langevin_update: null pointer");
  DS_CHECK(workspace_bytes >= 16 * (int64_t)B + 16, "sde_langevin_update: workspace too small");
  return ds_launch_langevin(snr, x, score, z, x_out, x_mean_out, B, n_per_batch, workspace, (hipStream_t)stream);
}
extern "C" int32_t diffsep_normalize_batch(const float* mix, float* mix_norm, float* mean, float* std, int32_t B,
                                           int64_t T, void* stream) {
  DS_CHECK(mix && mix_norm, "normalize_batch: null pointer");
  return ds_launch_normalize(mix, mix_norm, mean, std, B, T, (hipStream_t)stream);
}
extern "C" int32_t diffsep_scale_output(const float* mix, float* sep, int32_t B, int32_t S, int64_t T, void* stream) {
  DS_CHECK(mix && sep, "scale_output: null pointer");
  return ds_launch_scale_output(mix, sep, B, S, T, (hipStream_t)stream);
}
extern "C" int32_t diffsep_gram(const float* ref, const float* est, double* out, int32_t B, int32_t S, int64_t T,
                                void* stream) {
  DS_CHECK(ref && est && out, "gram: null pointer");
  return ds_launch_gram(ref, est, out, B, S, T, (hipStream_t)stream);
}
extern "C" int32_t diffsep_randn(float* out, int64_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  DS_CHECK(out, "randn: null pointer");
  return ds_launch_randn(out, n, seed, stream_id, (hipStream_t)stream);
}
extern "C" int32_t diffsep_randn_batch(float* out, int32_t B, int32_t S, int64_t T, const uint64_t* seeds,
                                       const int32_t* lengths, uint64_t stream_id, void* stream) {
  DS_CHECK(out && seeds && lengths && B >= 1 && S >= 1 && T >= 1, "randn_batch: bad argument");
  return ds_launch_randn_batch(out, B, S, T, seeds, lengths, stream_id, (hipStream_t)stream);
}
extern "C" int32_t diffsep_convert(const void* src, void* dst, int64_t n, int32_t sd, int32_t dd, void* stream) {
  DS_CHECK(src && dst, "convert: null pointer");
  return ds_launch_convert(src, dst, n, sd, dd, (hipStream_t)stream);
}
