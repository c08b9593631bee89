// HiFi-GAN / Fre-GAN generator forward as a chain of fused MFMA conv launches.
//
// Reference semantics:
//   HiFi-GAN  Generator.forward  models/vocoder/hifigan/models.py:134-150
//   Fre-GAN   FreGAN.forward     models/vocoder/fregan/generator.py:137-166
//
// Every leaky-relu, bias, residual add, the "sum of resblocks / num_kernels"
// average and the final tanh are folded into the producing / consuming conv
// kernel (conv1d.hip), so activations make exactly one HBM round trip per
// conv.  Activation layout is the reference's [B][C][T] (time contiguous).
#include <mutex>
#include "common.h"

namespace mb {

struct ConvSpec {
  int c_out, c_in, k, stride, pad, dil, transposed;
};

struct ConvW {
  ConvSpec s;
  DevBuf w, b;
};

static int get_padding(int k, int d) { return (k * d - d) / 2; }  // utils/util.py:60-61

// Conv list in ABI weight order (see mbhip.h).
static int gan_specs(const mb_gan_config* c, std::vector<ConvSpec>* out) {
  MB_REQUIRE(c, "gan: null config");
  MB_REQUIRE(c->num_upsamples >= 1 && c->num_upsamples <= MB_GAN_MAX_UPS, "gan: num_upsamples");
  MB_REQUIRE(c->num_kernels >= 1 && c->num_kernels <= MB_GAN_MAX_KERNELS, "gan: num_kernels");
  MB_REQUIRE(c->num_dilations >= 1 && c->num_dilations <= MB_GAN_MAX_DIL, "gan: num_dilations");
  const int uic = c->upsample_initial_channel;
  out->clear();
  out->push_back({uic, c->num_mels, 7, 1, 3, 1, 0});  // conv_pre
  for (int i = 0; i < c->num_upsamples; ++i) {
    const int u = c->upsample_rates[i], k = c->upsample_kernel_sizes[i];
    if (c->interp_ups) {  // InterpolationBlock(u) + Conv1d(k, padding=(k-1)//2)  models.py:107-112
      MB_REQUIRE(c->kind == MB_GAN_HIFIGAN, "gan: interp_ups is a HiFi-GAN option");
      MB_REQUIRE(u >= 1 && u <= 16 && k >= 1, "gan: upsample %d: rate %d kernel %d", i, u, k);
      out->push_back({uic >> (i + 1), uic >> i, k, 1, (k - 1) / 2, 1, 0});
      continue;
    }
    MB_REQUIRE(u >= 1 && u <= 8 && k % u == 0, "gan: upsample %d: rate %d kernel %d", i, u, k);
    MB_REQUIRE(k - 2 * (u / 2 + u % 2) + u % 2 == u, "gan: upsample %d is not an exact x%d", i, u);
    out->push_back({uic >> (i + 1), uic >> i, k, u, u / 2 + u % 2, 1, 1});
  }
  if (c->kind == MB_GAN_FREGAN) {
    const int lvl = c->num_upsamples - c->top_k;  // cond_level (generator.py:89)
    MB_REQUIRE(lvl >= 1, "fregan: top_k must be < num_upsamples");
    int kr = c->num_mels;
    for (int i = lvl; i < c->num_upsamples; ++i) {  // cond_up (generator.py:111-118)
      const int u = c->upsample_rates[i - 1], k = c->upsample_kernel_sizes[i - 1];
      out->push_back({uic >> i, kr, k, u, u / 2 + u % 2, 1, 1});
      kr = uic >> i;
    }
    for (int i = lvl + 1; i < c->num_upsamples; ++i)  // res_output (generator.py:103-110)
      out->push_back({uic >> (i + 1), uic >> i, 1, 1, 0, 1, 0});
  }
  MB_REQUIRE(c->resblock_type >= 0 && c->resblock_type <= 2, "gan: resblock_type %d", c->resblock_type);
  MB_REQUIRE(c->resblock_type != 2 || c->num_dilations >= 2, "gan: ResBlock2 needs two dilations per block");
  for (int i = 0; i < c->num_upsamples; ++i) {
    const int ch = uic >> (i + 1);
    for (int j = 0; j < c->num_kernels; ++j) {
      const int k = c->resblock_kernel_sizes[j];
      if (c->resblock_type == 2) {  // ResBlock2.convs[0..1] (models.py:54-61)
        for (int d = 0; d < 2; ++d) out->push_back({ch, ch, k, 1, get_padding(k, c->resblock_dilations[j][d]), c->resblock_dilations[j][d], 0});
        continue;
      }
      for (int d = 0; d < c->num_dilations; ++d) {
        const int dil = c->resblock_dilations[j][d];
        out->push_back({ch, ch, k, 1, get_padding(k, dil), dil, 0});
      }
      for (int d = 0; d < c->num_dilations; ++d) out->push_back({ch, ch, k, 1, get_padding(k, 1), 1, 0});
    }
  }
  out->push_back({1, uic >> c->num_upsamples, 7, 1, 3, 1, 0});  // conv_post
  return MB_OK;
}

__global__ void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t n4 = n >> 2;
  float4* y4 = reinterpret_cast<float4*>(y);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (size_t k = i; k < n4; k += stride) {
    float4 a = y4[k], b = x4[k];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    y4[k] = a;
  }
  for (size_t k = (n4 << 2) + i; k < n; k += stride) y[k] += x[k];
}

static int add_inplace(float* y, const float* x, size_t n, hipStream_t s) {
  if (!n) return MB_OK;
  int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(add_inplace_kernel, dim3(blocks), dim3(256), 0, s, y, x, n);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
// fp16 variant: n is a multiple of 8 (channel counts are multiples of 8 on the fp16 path)
__global__ void add_inplace_f16_kernel(_Float16* __restrict__ y, const _Float16* __restrict__ x, size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  h16x8_t* y8 = reinterpret_cast<h16x8_t*>(y);
  const h16x8_t* x8 = reinterpret_cast<const h16x8_t*>(x);
  for (size_t k = i; k < n8; k += stride) y8[k] = y8[k] + x8[k];
}

static int add_inplace_f16(void* y, const void* x, size_t n, hipStream_t s) {
  if (!n) return MB_OK;
  MB_REQUIRE(n % 8 == 0, "gan(f16): activation size %zu not a multiple of 8", n);
  int blocks = (int)std::min<size_t>((n / 8 + 255) / 256, 4096);
  hipLaunchKernelGGL(add_inplace_f16_kernel, dim3(blocks), dim3(256), 0, s, (_Float16*)y, (const _Float16*)x, n / 8);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

// x[b][c][t] += bias[b][c]  (fp32 channel-major)  /  x[b][t][c] += bias[b][c]  (fp16 time-major):
// the VITS decoder's `x + cond(g)` after conv_pre (vits.py:274-276), cond(g) being constant over time
__global__ void add_chan_bias_f32_kernel(float* __restrict__ x, const float* __restrict__ bias, int C, int T) {
  const int b = blockIdx.z, c = blockIdx.y;
  const float v = bias[(size_t)b * C + c];
  float* xr = x + ((size_t)b * C + c) * T;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) xr[t] += v;
}
__global__ void add_chan_bias_f16_kernel(_Float16* __restrict__ x, const float* __restrict__ bias, int C, int T) {
  const int b = blockIdx.y;
  const size_t n = (size_t)C * T;
  _Float16* xb = x + (size_t)b * n;
  const float* bb = bias + (size_t)b * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    xb[i] = (_Float16)((float)xb[i] + bb[i % C]);
}

// x[b][t][c] += bias[b][c]  (fp32 time-major)
__global__ void add_chan_bias_tm_kernel(float* __restrict__ x, const float* __restrict__ bias, int C, int T) {
  const int b = blockIdx.y;
  const size_t n = (size_t)C * T;
  float* xb = x + (size_t)b * n;
  const float* bb = bias + (size_t)b * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) xb[i] += bb[i % C];
}

}  // namespace mb

using namespace mb;

struct mb_gan {
  mb_gan_config cfg;
  int dtype = MB_F32;
  std::vector<ConvW> convs;  // ABI order
  // fp16 path: fused (convs1[d], convs2[d]) weight streams of the ResBlocks (resblock_f16.hip), indexed
  // ((stage * num_kernels + kernel) * num_dilations + d); empty buffer = pair not fusable -> two launches
  std::vector<DevBuf> pairs;
  // fp16 path, narrow stages (C <= 32): the whole ResBlock group of a stage as ONE launch (resblock_stage_f16.hip); per stage the
  // weight stream and the [kernel][unit][2][C] bias block; empty = the stage runs its units one by one
  std::vector<DevBuf> stage_w, stage_b;
  // fp16 path, wider stages (64 / 128 channels): single ResBlocks whose reach is small against the rows LDS holds (k = 3, 7 at 64
  // channels, k = 3 at 128) as ONE launch each, accumulating into the stage output; indexed (stage * num_kernels + kernel)
  std::vector<DevBuf> chain_w, chain_b;
  // fp32 path (round 4, resblock_stage_f32.hip): 32-channel stages as ONE launch per stage (s32_w / s32_b, indexed by stage); 64-channel
  // stages as one launch per ResBlock where the reach leaves >= 60 % of a 192-row window useful, else one launch per unit -- the
  // launches of ResBlock (stage, kernel) in order: first unit, unit count, weight stream, bias + unscale block
  struct S32Launch { int u0, nu; DevBuf w, b; };
  std::vector<DevBuf> s32_w, s32_b;
  std::vector<std::vector<S32Launch>> r32;  // [stage * num_kernels + kernel]
  // fp32 path (round 6, resblock_pair_split.hip): the ResBlock units on TIME-major fp32 tensors, one launch per (convs1[d], convs2[d]);
  // indexed as `pairs`; a stage whose units all have an image runs time-major (turned once in, once out), the others keep the plan above
  struct SPair { DevBuf w; float us1 = 0.f, us2 = 0.f; };
  std::vector<SPair> spairs;
  // ... and every other conv as a time-major split conv (conv_split_tm.hip), indexed as `convs` (ResBlock entries stay empty).  When
  // every conv of the generator has an image (tm_all) the whole forward runs time-major: the mel is turned once, nothing else is.
  struct TmConv {
    DevBuf w, bias; float us = 0.f; int c_in = 0, m = 0, k = 0, pad = 0, dil = 1, rep = 1;
    DevBuf w1; float bias1 = 0.f;  // a conv to ONE channel (conv_post) also as [k][c_in] fp32 for mb_conv_c1_tm: an HBM stream, no MFMA tile
  };
  std::vector<TmConv> tmc;
  bool tm_all = false;
  // tm_all: the parallel ResBlocks of a stage (xs = mean_j resblock_j(x): independent chains) run on streams of their own -- one launch
  // of 256 persistent workgroups rarely divides its tiles evenly (256 channels, 32 x 1600 rows: 576 tiles = 2.25 rounds), and the
  // chains' workgroups fill each other's last rounds.  side[j] carries chain j in stages of few rounds (a cross-queue edge costs
  // ~13 us: stages of many rounds stay on the caller's stream); ev_fork =
  // the stage input is ready, ev_last[j] = chain j's accumulate into the stage output is enqueued (chain j + 1's accumulate waits for
  // it: the sum keeps its order, results are bit for bit those of one stream).  Events belong to the handle: enqueue under `mu`.
  std::vector<hipStream_t> side;
  std::vector<hipEvent_t> ev_last;
  hipEvent_t ev_fork = nullptr;
  std::mutex mu;
  int hop;
  // indices into convs
  int i_pre, i_ups, i_cond, i_resout, i_rb, i_post;
};

extern "C" int mb_gan_num_weights(const mb_gan_config* cfg) {
  std::vector<ConvSpec> v;
  if (gan_specs(cfg, &v)) return MB_EINVAL;
  return (int)v.size() * 2;
}

extern "C" size_t mb_gan_weight_numel(const mb_gan_config* cfg, int index) {
  std::vector<ConvSpec> v;
  if (gan_specs(cfg, &v) || index < 0 || index >= (int)v.size() * 2) return 0;
  const ConvSpec& s = v[index / 2];
  return (index & 1) ? (size_t)s.c_out : (size_t)s.c_out * s.c_in * s.k;
}

extern "C" int mb_gan_create(const mb_gan_config* cfg, const float* const* h_weights, int n_weights,
                             mb_gan** out) {
  return mb_gan_create_ex(cfg, h_weights, n_weights, MB_F32, out);
}

extern "C" int mb_gan_dtype(const mb_gan* g) { return g ? g->dtype : MB_EINVAL; }

extern "C" int mb_gan_create_ex(const mb_gan_config* cfg, const float* const* h_weights, int n_weights,
                                int dtype, mb_gan** out) {
  MB_REQUIRE(out && h_weights, "gan_create: null pointer");
  MB_REQUIRE(dtype == MB_F32 || dtype == MB_F16, "gan_create: dtype %d", dtype);
  std::vector<ConvSpec> v;
  int rc = gan_specs(cfg, &v);
  if (rc) return rc;
  MB_REQUIRE(n_weights == (int)v.size() * 2, "gan_create: expected %d weight tensors, got %d",
             (int)v.size() * 2, n_weights);
  mb_gan* g = new mb_gan();
  g->cfg = *cfg;
  g->dtype = dtype;
  g->convs.resize(v.size());
  std::vector<float> packed;
  for (size_t i = 0; i < v.size(); ++i) {
    const ConvSpec& s = v[i];
    g->convs[i].s = s;
    if (dtype == MB_F16) {
      if (s.c_in % 8 != 0) {
        set_error("gan_create(f16): conv %zu has c_in=%d, the fp16 path needs multiples of 8", i, s.c_in);
        mb_gan_destroy(g);
        return MB_EINVAL;
      }
      const size_t nh = mb_conv1d_f16_packed_halves(s.c_out, s.c_in, s.k, s.stride);  // multiple of 512
      packed.assign(nh / 2, 0.f);
      rc = mb_conv1d_f16_pack(h_weights[2 * i], s.c_out, s.c_in, s.k, s.stride, s.transposed, s.pad,
                              reinterpret_cast<uint16_t*>(packed.data()));
    } else {
      packed.assign(mb_conv1d_packed_floats(s.c_out, s.c_in, s.k, s.stride), 0.f);
      rc = mb_conv1d_pack(h_weights[2 * i], s.c_out, s.c_in, s.k, s.stride, s.transposed, s.pad,
                          packed.data());
    }
    if (!rc) rc = g->convs[i].w.upload(packed.data(), packed.size());
    if (!rc) rc = g->convs[i].b.upload(h_weights[2 * i + 1], s.c_out);
    if (rc) { mb_gan_destroy(g); return rc; }
  }
  g->hop = 1;
  for (int i = 0; i < cfg->num_upsamples; ++i) g->hop *= cfg->upsample_rates[i];
  int idx = 0;
  g->i_pre = idx++;
  g->i_ups = idx; idx += cfg->num_upsamples;
  g->i_cond = g->i_resout = -1;
  if (cfg->kind == MB_GAN_FREGAN) {
    const int lvl = cfg->num_upsamples - cfg->top_k;
    g->i_cond = idx; idx += cfg->num_upsamples - lvl;
    g->i_resout = idx; idx += cfg->num_upsamples - lvl - 1;
  }
  const bool rb2 = cfg->resblock_type == 2;  // two single-conv units per block: none of the fused (convs1, convs2) plans applies
  g->i_rb = idx; idx += cfg->num_upsamples * cfg->num_kernels * (rb2 ? 2 : cfg->num_dilations * 2);
  g->i_post = idx;
  // MBHIP_GAN_FUSE = all (default) | nochain (no one-launch ResBlock chains) | units (fused units only: no stage / chain launches) |
  // none (one launch per conv): the fallbacks the parity tests compare the fused launches with
  const char* fenv = getenv("MBHIP_GAN_FUSE");
  const std::string fuse = fenv ? fenv : "all";
  if (!(fuse == "all" || fuse == "nochain" || fuse == "units" || fuse == "none" || fuse == "cm")) {
    set_error("MBHIP_GAN_FUSE: unknown value '%s' (all | nochain | units | none | cm)", fuse.c_str());
    mb_gan_destroy(g);
    return MB_EINVAL;
  }
  // (cm: the fp32 path's channel-major plan of rounds 4-5 -- resblock_stage_f32 launches + one launch per wide conv -- instead of the
  //  time-major split pairs; on the fp16 path it reads as "all")
  const bool no_fuse = fuse == "none" || rb2, no_stage = no_fuse || fuse == "units", no_chain = no_stage || fuse == "nochain";
  const bool f32_cm = fuse == "cm";
  if (dtype == MB_F32 && !no_fuse && !f32_cm) {
    const int nd = cfg->num_dilations;
    g->spairs.resize((size_t)cfg->num_upsamples * cfg->num_kernels * nd);
    std::vector<float> img;
    for (int i = 0; i < cfg->num_upsamples; ++i)
      for (int j = 0; j < cfg->num_kernels; ++j)
        for (int d = 0; d < nd; ++d) {
          const int base = g->i_rb + ((i * cfg->num_kernels + j) * nd) * 2;
          const ConvSpec& s1 = v[base + d];
          const ConvSpec& s2 = v[base + nd + d];
          if (s1.c_in != s1.c_out || s2.c_in != s1.c_in || s2.c_out != s1.c_in || s1.k != s2.k || s2.dil != 1 ||
              s1.transposed || s2.transposed || s1.pad != s1.dil * (s1.k - 1) / 2 || s2.pad != (s2.k - 1) / 2 ||
              !mb_resblock_pair_split_supported(s1.c_in, s1.k, s1.dil))
            continue;
          img.assign(mb_resblock_pair_split_packed_halves(s1.c_in, s1.k) / 2, 0.f);
          float us[2] = {0.f, 0.f};
          mb_gan::SPair& sp = g->spairs[(size_t)(i * cfg->num_kernels + j) * nd + d];
          rc = mb_resblock_pair_split_pack(h_weights[2 * (base + d)], h_weights[2 * (base + nd + d)], s1.c_in, s1.k,
                                           reinterpret_cast<uint16_t*>(img.data()), us);
          if (!rc) rc = sp.w.upload(img.data(), img.size());
          sp.us1 = us[0]; sp.us2 = us[1];
          if (rc) { mb_gan_destroy(g); return rc; }
        }
  }
  if (dtype == MB_F16 && !no_fuse) {
    const int nd = cfg->num_dilations;
    g->pairs.resize((size_t)cfg->num_upsamples * cfg->num_kernels * nd);
    std::vector<float> img;
    for (int i = 0; i < cfg->num_upsamples; ++i)
      for (int j = 0; j < cfg->num_kernels; ++j)
        for (int d = 0; d < nd; ++d) {
          const int base = g->i_rb + ((i * cfg->num_kernels + j) * nd) * 2;
          const ConvSpec& s1 = v[base + d];
          const ConvSpec& s2 = v[base + nd + d];
          if (s1.c_in != s1.c_out || s2.c_in != s1.c_in || s2.c_out != s1.c_in || s1.k != s2.k || s2.dil != 1 ||
              s1.transposed || s2.transposed || s1.pad != s1.dil * (s1.k - 1) / 2 || s2.pad != (s2.k - 1) / 2 ||
              !mb_resblock_pair_f16_supported(s1.c_in, s1.k, s1.dil))
            continue;
          img.assign(mb_resblock_pair_f16_packed_halves(s1.c_in, s1.k) / 2, 0.f);
          rc = mb_resblock_pair_f16_pack(h_weights[2 * (base + d)], h_weights[2 * (base + nd + d)], s1.c_in, s1.k,
                                         reinterpret_cast<uint16_t*>(img.data()));
          if (!rc) rc = g->pairs[(size_t)(i * cfg->num_kernels + j) * nd + d].upload(img.data(), img.size());
          if (rc) { mb_gan_destroy(g); return rc; }
        }
  }
  if (dtype == MB_F32 && (!g->spairs.empty() || (rb2 && fuse != "none" && !f32_cm)) && !cfg->interp_ups) {
    // conv_pre / ups / cond_up / res_output / conv_post as time-major split convs.  ConvTranspose1d(C -> C', 2 u taps, stride u, padding
    // u/2 + u%2) = a three-tap conv to u C' channels: W[r C' + co][ci][j] = w[ci][co][u (1 - j) + r + pad]  (conv_split_tm.hip)
    bool all = true;
    for (auto& sp : g->spairs) all = all && sp.w.p != nullptr;
    g->tmc.resize(v.size());
    std::vector<float> weff, beff, img;
    auto make_tm = [&](int idx, int rep) -> int {  // rep > 1: 1x1 conv behind a nearest-repeat x rep (Fre-GAN res_output)
      const ConvSpec& sp = v[idx];
      mb_gan::TmConv& t = g->tmc[idx];
      const float* w = h_weights[2 * idx];
      const float* b = h_weights[2 * idx + 1];
      if (sp.transposed) {
        const int u = sp.stride;
        if (sp.k != 2 * u || sp.dil != 1) return 1;
        t.c_in = sp.c_in; t.m = u * sp.c_out; t.k = 3; t.pad = 1; t.rep = u;
        weff.assign((size_t)t.m * t.c_in * 3, 0.f);
        for (int r = 0; r < u; ++r)
          for (int co = 0; co < sp.c_out; ++co)
            for (int ci = 0; ci < sp.c_in; ++ci)
              for (int j = 0; j < 3; ++j) {
                const int kk = u * (1 - j) + r + sp.pad;
                if (kk >= 0 && kk < sp.k) weff[((size_t)(r * sp.c_out + co) * t.c_in + ci) * 3 + j] = w[((size_t)ci * sp.c_out + co) * sp.k + kk];
              }
        beff.resize(t.m);
        for (int r = 0; r < u; ++r) memcpy(&beff[(size_t)r * sp.c_out], b, sp.c_out * sizeof(float));
      } else if (rep > 1) {
        if (sp.k != 1) return 1;
        t.c_in = sp.c_in; t.m = rep * sp.c_out; t.k = 1; t.pad = 0; t.rep = rep;
        weff.resize((size_t)t.m * t.c_in);
        beff.resize(t.m);
        for (int r = 0; r < rep; ++r) {
          memcpy(&weff[(size_t)r * sp.c_out * sp.c_in], w, (size_t)sp.c_out * sp.c_in * sizeof(float));
          memcpy(&beff[(size_t)r * sp.c_out], b, sp.c_out * sizeof(float));
        }
      } else {
        if (sp.stride != 1) return 1;
        t.c_in = sp.c_in; t.m = sp.c_out; t.k = sp.k; t.pad = sp.pad; t.dil = sp.dil; t.rep = 1;
        weff.assign(w, w + (size_t)sp.c_out * sp.c_in * sp.k);
        beff.assign(b, b + sp.c_out);
      }
      if (!mb_conv_split_tm_supported(t.m, t.c_in, t.k, t.dil)) return 1;
      img.assign(mb_conv_split_tm_packed_halves(t.m, t.c_in, t.k) / 2, 0.f);
      int r = mb_conv_split_tm_pack(weff.data(), t.m, t.c_in, t.k, reinterpret_cast<uint16_t*>(img.data()), &t.us);
      if (!r) r = t.w.upload(img.data(), img.size());
      if (!r) r = t.bias.upload(beff.data(), beff.size());
      if (!r && t.m == 1 && t.rep == 1 && t.c_in % 4 == 0 && 2 * t.pad == t.dil * (t.k - 1) && !diag_int("gan_post_mfma")) {  // (A/B: conv_post on the MFMA kernel)
        std::vector<float> w1((size_t)t.k * t.c_in);
        for (int ci = 0; ci < t.c_in; ++ci)
          for (int j = 0; j < t.k; ++j) w1[(size_t)j * t.c_in + ci] = weff[(size_t)ci * t.k + j];
        r = t.w1.upload(w1.data(), w1.size());
        t.bias1 = beff[0];
      }
      return r;
    };
    auto want = [&](int idx, int rep) {
      if (!all || rc) return;
      const int r = make_tm(idx, rep);
      if (r < 0) rc = r;
      else if (r) all = false;
    };
    want(g->i_pre, 1);
    for (int i = 0; i < cfg->num_upsamples; ++i) want(g->i_ups + i, 1);
    if (cfg->kind == MB_GAN_FREGAN) {
      const int lvl = cfg->num_upsamples - cfg->top_k;
      for (int i = lvl; i < cfg->num_upsamples; ++i) want(g->i_cond + (i - lvl), 1);
      for (int i = lvl + 1; i < cfg->num_upsamples; ++i) want(g->i_resout + (i - lvl - 1), cfg->upsample_rates[i]);
    }
    if (rb2)
      for (int q = 0; q < cfg->num_upsamples * cfg->num_kernels * 2; ++q) want(g->i_rb + q, 1);
    want(g->i_post, 1);
    if (rc) { mb_gan_destroy(g); return rc; }
    g->tm_all = all && !diag_int("gan_tm_pairs_only");  // A/B: the ResBlock units time-major, everything else channel-major (first form of the round)
    if (!g->tm_all) { for (auto& t : g->tmc) { t.w.release(); t.bias.release(); t.w1.release(); } g->tmc.clear(); }
    if (g->tm_all && cfg->num_kernels > 1 && !diag_int("gan_one_stream")) {  // (A/B: every launch on the caller's stream)
      hipError_t e = hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming);
      for (int j = 0; j < cfg->num_kernels && e == hipSuccess; ++j) {
        hipEvent_t ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) g->ev_last.push_back(ev);
      }
      // the chains borrow the library's pool streams (common.h: a handle with streams of its own oversubscribes the 4 hardware queues
      // as soon as a process holds a few handles); more ResBlocks than pool streams: the extra chains share
      for (int j = 0; j < cfg->num_kernels && e == hipSuccess; ++j) {
        hipStream_t st = nullptr;
        if (pool_stream(j % POOL_STREAMS, &st) != MB_OK) e = hipErrorUnknown;
        else g->side.push_back(st);
      }
      if (e != hipSuccess) { mb_gan_destroy(g); return hip_fail(e, "gan_create: branch streams", __FILE__, __LINE__); }
    }
  }
  if (dtype == MB_F16 && !no_stage && cfg->num_kernels <= 4 &&
      cfg->num_dilations <= 4) {
    const int nd = cfg->num_dilations, nk = cfg->num_kernels;
    g->stage_w.resize(cfg->num_upsamples);
    g->stage_b.resize(cfg->num_upsamples);
    std::vector<float> img, bias;
    for (int i = 0; i < cfg->num_upsamples; ++i) {
      const int ch = cfg->upsample_initial_channel >> (i + 1);
      int ks[4], dil[16];
      std::vector<const float*> w1((size_t)nk * nd), w2((size_t)nk * nd);
      bool ok = ch == 16 || ch == 32;
      bias.assign((size_t)nk * nd * 2 * ch, 0.f);
      for (int j = 0; j < nk && ok; ++j)
        for (int d = 0; d < nd && ok; ++d) {
          const int base = g->i_rb + ((i * nk + j) * nd) * 2;
          const ConvSpec& s1 = v[base + d];
          const ConvSpec& s2 = v[base + nd + d];
          ok = s1.c_in == ch && s1.c_out == ch && s2.c_in == ch && s2.c_out == ch && s1.k == s2.k && s2.dil == 1 && !s1.transposed &&
               !s2.transposed && s1.pad == s1.dil * (s1.k - 1) / 2 && s2.pad == (s2.k - 1) / 2 && (d == 0 || s1.k == ks[j]);
          ks[j] = s1.k; dil[j * nd + d] = s1.dil;
          w1[(size_t)j * nd + d] = h_weights[2 * (base + d)];
          w2[(size_t)j * nd + d] = h_weights[2 * (base + nd + d)];
          if (ok) {
            memcpy(&bias[(((size_t)j * nd + d) * 2) * ch], h_weights[2 * (base + d) + 1], ch * sizeof(float));
            memcpy(&bias[(((size_t)j * nd + d) * 2 + 1) * ch], h_weights[2 * (base + nd + d) + 1], ch * sizeof(float));
          }
        }
      if (!ok || !mb_resblock_stage_f16_supported(ch, nk, ks, nd, dil)) continue;
      img.assign(mb_resblock_stage_f16_packed_halves(ch, nk, ks, nd) / 2, 0.f);
      rc = mb_resblock_stage_f16_pack(w1.data(), w2.data(), ch, nk, ks, nd, reinterpret_cast<uint16_t*>(img.data()));
      if (!rc) rc = g->stage_w[i].upload(img.data(), img.size());
      if (!rc) rc = g->stage_b[i].upload(bias.data(), bias.size());
      if (rc) { mb_gan_destroy(g); return rc; }
    }
    if (!no_chain) {
      g->chain_w.resize((size_t)cfg->num_upsamples * nk);
      g->chain_b.resize((size_t)cfg->num_upsamples * nk);
      for (int i = 0; i < cfg->num_upsamples; ++i) {
        const int ch = cfg->upsample_initial_channel >> (i + 1);
        if (ch != 64 && ch != 128) continue;
        for (int j = 0; j < nk; ++j) {
          const int base = g->i_rb + ((i * nk + j) * nd) * 2;
          int kj = 0, dil[4];
          const float *w1[4], *w2[4];
          bool ok = !g->pairs.empty();
          bias.assign((size_t)nd * 2 * ch, 0.f);
          for (int d = 0; d < nd && ok; ++d) {
            const ConvSpec& s1 = v[base + d];
            ok = g->pairs[(size_t)(i * nk + j) * nd + d].p != nullptr && s1.c_in == ch && (d == 0 || s1.k == kj);  // (pairs: shapes checked above)
            kj = s1.k; dil[d] = s1.dil;
            w1[d] = h_weights[2 * (base + d)]; w2[d] = h_weights[2 * (base + nd + d)];
            if (ok) {
              memcpy(&bias[((size_t)d * 2) * ch], h_weights[2 * (base + d) + 1], ch * sizeof(float));
              memcpy(&bias[((size_t)d * 2 + 1) * ch], h_weights[2 * (base + nd + d) + 1], ch * sizeof(float));
            }
          }
          // useful rows per window row: measured on HiFi-GAN 32 x 200 (same box): no chain launches 3.667 ms, k = 3 at 64 channels
          // (0.91) 3.616, + k = 3 at 128 channels (0.81) 3.569, + k = 7 at 64 channels (0.72) 3.579 -- below ~0.75 the halo
          // recompute costs what the per-unit launches' tensor passes do
          // do.  k = 3 units are the least efficient per-unit launches (21 % of the matrix peak at 128 channels), so their chain
          // pays from 0.65: Fre-GAN 8 x 3000 with dilations (1, 3, 5, 7), k = 3 at 128 channels (0.69): 12.97 -> 12.78 ms same box.
          std::string ee;  // A/B (MBHIP_DIAG=gan_chain_eff=<x>): another threshold
          const bool have_ee = diag_str("gan_chain_eff", &ee);
          if (!ok || mb_resblock_stage_f16_efficiency(ch, 1, &kj, nd, dil) < (have_ee ? (float)atof(ee.c_str()) : (kj <= 3 ? 0.65f : 0.75f))) continue;
          img.assign(mb_resblock_stage_f16_packed_halves(ch, 1, &kj, nd) / 2, 0.f);
          rc = mb_resblock_stage_f16_pack(w1, w2, ch, 1, &kj, nd, reinterpret_cast<uint16_t*>(img.data()));
          if (!rc) rc = g->chain_w[(size_t)i * nk + j].upload(img.data(), img.size());
          if (!rc) rc = g->chain_b[(size_t)i * nk + j].upload(bias.data(), bias.size());
          if (rc) { mb_gan_destroy(g); return rc; }
        }
      }
    }
  }
  if (dtype == MB_F32 && !no_stage && cfg->num_kernels <= 4 && cfg->num_dilations <= 4) {
    // fp32 path: fused ResBlock groups on error-compensated operands (resblock_stage_f32.hip) -- the plan of MBHIP_GAN_FUSE=cm, and
    // of every stage the time-major pairs above do not cover
    const int nd = cfg->num_dilations, nk = cfg->num_kernels;
    g->s32_w.resize(cfg->num_upsamples);
    g->s32_b.resize(cfg->num_upsamples);
    g->r32.resize((size_t)cfg->num_upsamples * nk);
    // one launch's image: ResBlocks [j0, j0 + nkl) x units [u0, u0 + nul) of stage i
    auto make = [&](int i, int ch, int j0, int nkl, int u0, int nul, DevBuf* wout, DevBuf* bout) -> int {
      int ks[4], dil[16];
      std::vector<const float*> w1((size_t)nkl * nul), w2((size_t)nkl * nul);
      std::vector<float> bias((size_t)nkl * nul * 2 * (ch + 1), 0.f);
      for (int j = 0; j < nkl; ++j)
        for (int u = 0; u < nul; ++u) {
          const int base = g->i_rb + ((i * nk + j0 + j) * nd) * 2;
          const ConvSpec& s1 = v[base + u0 + u];
          const ConvSpec& s2 = v[base + nd + u0 + u];
          const bool ok = s1.c_in == ch && s1.c_out == ch && s2.c_in == ch && s2.c_out == ch && s1.k == s2.k && s2.dil == 1 && !s1.transposed &&
                          !s2.transposed && s1.pad == s1.dil * (s1.k - 1) / 2 && s2.pad == (s2.k - 1) / 2 && (u == 0 || s1.k == ks[j]);
          if (!ok) return 1;  // not a ResBlock1 unit: the per-conv launches run
          ks[j] = s1.k; dil[j * nul + u] = s1.dil;
          w1[(size_t)j * nul + u] = h_weights[2 * (base + u0 + u)];
          w2[(size_t)j * nul + u] = h_weights[2 * (base + nd + u0 + u)];
          memcpy(&bias[(((size_t)j * nul + u) * 2) * ch], h_weights[2 * (base + u0 + u) + 1], ch * sizeof(float));
          memcpy(&bias[(((size_t)j * nul + u) * 2 + 1) * ch], h_weights[2 * (base + nd + u0 + u) + 1], ch * sizeof(float));
        }
      if (!mb_resblock_stage_f32_supported(ch, nkl, ks, nul, dil)) return 1;
      std::vector<float> img(mb_resblock_stage_f32_packed_halves(ch, nkl, ks, nul) / 2, 0.f);
      int r = mb_resblock_stage_f32_pack(w1.data(), w2.data(), ch, nkl, ks, nul, reinterpret_cast<uint16_t*>(img.data()),
                                         &bias[(size_t)nkl * nul * 2 * ch]);
      if (!r) r = wout->upload(img.data(), img.size());
      if (!r) r = bout->upload(bias.data(), bias.size());
      return r ? r : 0;
    };
    for (int i = 0; i < cfg->num_upsamples && !rc; ++i) {
      const int ch = cfg->upsample_initial_channel >> (i + 1);
      // (32 channels: the whole group in ONE launch -- s32_w -- works, tests/test_resblock_stage_f32_gpu.py, but every chain then pays the
      //  widest chain's halo: 136 useful rows of 256; a launch per ResBlock / unit lets k = 3 keep 232 and k = 11's units 196-236)
      if (ch == 32 && diag_int("gan_s32_group")) {
        const int r = make(i, ch, 0, nk, 0, nd, &g->s32_w[i], &g->s32_b[i]);
        if (r < 0) rc = r;
        if (r) { g->s32_w[i].release(); g->s32_b[i].release(); }
      } else if (ch == 64 || ch == 32) {
        for (int j = 0; j < nk && !rc; ++j) {
          std::vector<mb_gan::S32Launch>& plan = g->r32[(size_t)i * nk + j];
          // whole ResBlock in one launch if >= 60 % of the window rows are useful, else unit by unit
          int kj = v[g->i_rb + ((i * nk + j) * nd) * 2].k, dl[4];
          for (int u = 0; u < nd; ++u) dl[u] = v[g->i_rb + ((i * nk + j) * nd) * 2 + u].dil;
          const bool whole = mb_resblock_stage_f32_efficiency(ch, 1, &kj, nd, dl) >= 0.6f;
          const int nl = whole ? 1 : nd;
          plan.resize(nl);
          for (int l = 0; l < nl && !rc; ++l) {
            plan[l].u0 = whole ? 0 : l; plan[l].nu = whole ? nd : 1;
            const int r = make(i, ch, j, 1, plan[l].u0, plan[l].nu, &plan[l].w, &plan[l].b);
            if (r < 0) rc = r;
            if (r) { for (auto& q : plan) { q.w.release(); q.b.release(); } plan.clear(); break; }
          }
        }
      }
    }
    if (rc) { mb_gan_destroy(g); return rc; }
  }
  *out = g;
  return MB_OK;
}

extern "C" void mb_gan_destroy(mb_gan* g) {
  if (!g) return;
  for (auto& c : g->convs) { c.w.release(); c.b.release(); }
  for (auto& p : g->pairs) p.release();
  for (auto& p : g->spairs) p.w.release();
  g->side.clear();  // (borrowed from the pool)
  for (auto ev : g->ev_last) (void)hipEventDestroy(ev);
  if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
  for (auto& t : g->tmc) { t.w.release(); t.bias.release(); t.w1.release(); }
  for (auto& p : g->stage_w) p.release();
  for (auto& p : g->stage_b) p.release();
  for (auto& p : g->s32_w) p.release();
  for (auto& p : g->s32_b) p.release();
  for (auto& pl : g->r32)
    for (auto& q : pl) { q.w.release(); q.b.release(); }
  for (auto& p : g->chain_w) p.release();
  for (auto& p : g->chain_b) p.release();
  delete g;
}

extern "C" int mb_gan_hop(const mb_gan* g) { return g ? g->hop : 0; }

// length after ups[i] for an input of t samples
static long long gan_up_len(const mb_gan_config& c, int i, long long t) {
  const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
  return c.interp_ups ? t * u + 2 * ((k - 1) / 2) - (k - 1) : t * u;
}

extern "C" long long mb_gan_out_samples(const mb_gan* g, int frames) {
  if (!g || frames <= 0) return 0;
  long long t = frames;
  for (int i = 0; i < g->cfg.num_upsamples; ++i) t = gan_up_len(g->cfg, i, t);
  return t > 0 ? t : 0;
}

// Largest [C][T] activation of any stage, per batch item, in floats.
static size_t gan_max_act(const mb_gan* g, int frames) {
  const mb_gan_config& c = g->cfg;
  size_t m = (size_t)c.upsample_initial_channel * frames;
  size_t t = frames;
  for (int i = 0; i < c.num_upsamples; ++i) {
    t *= c.upsample_rates[i];
    m = std::max(m, (size_t)(c.upsample_initial_channel >> (i + 1)) * t);
    m = std::max(m, (size_t)(c.upsample_initial_channel >> i) * t);  // fregan res_output / cond
  }
  return m;
}

extern "C" size_t mb_gan_workspace_bytes(const mb_gan* g, int batch, int frames) {
  if (!g || batch <= 0 || frames <= 0) return 0;
  const size_t esz = g->dtype == MB_F16 ? 2 : sizeof(float);
  const size_t per = align_up(gan_max_act(g, frames) * batch * esz, 256);
  const int nbuf = (g->cfg.kind == MB_GAN_FREGAN ? 8 : 4) + 2 * std::max(0, (int)g->side.size() - 1);  // + the chains' own ping-pong pairs (every chain runs on a side stream)
  // fp16: + the time-major fp16 copy of the mel
  const size_t melh = g->dtype == MB_F16 ? align_up((size_t)batch * frames * g->cfg.num_mels * 2, 256)
                                         : (g->tm_all ? align_up((size_t)batch * frames * g->cfg.num_mels * 4, 256) : 0);  // fp32: + its time-major copy
  return per * nbuf + melh + 256;
}

namespace {
struct Launcher {
  hipStream_t s;
  int batch;
  int dtype;
  int rc = MB_OK;
  int frames_max = 1;          // mel frames of the padded batch: a tensor of t stored rows has t / frames_max rows per mel frame
  const int* valid = nullptr;  // ragged batch: per-item mel frames (device); every launch masks by frames * (samples per frame so far)
  void add(void* y, const void* x, size_t n) {
    if (rc) return;
    rc = dtype == MB_F16 ? add_inplace_f16(y, x, n, s) : add_inplace((float*)y, (const float*)x, n, s);
  }
  void conv_f16(const ConvW& c, const void* x, int t_in, void* y, int in_act, float in_slope,
                const void* res, float out_scale, int accumulate, int out_act, int in_repeat, int y_f32) {
    mb_conv1d_f16_args a;
    memset(&a, 0, sizeof(a));
    const int t_eff = t_in * in_repeat;
    const int t_out = c.s.transposed ? t_eff * c.s.stride : t_eff + 2 * c.s.pad - c.s.dil * (c.s.k - 1);
    a.d_x = x; a.d_wpacked = c.w.p; a.d_bias = c.b.p; a.d_res = res; a.d_y = y;
    a.x_bstride = (long long)c.s.c_in * t_in;
    a.y_bstride = (long long)c.s.c_out * t_out;
    a.res_bstride = a.y_bstride;
    a.batch = batch; a.c_in = c.s.c_in; a.c_out = c.s.c_out; a.t_in = t_eff; a.t_out = t_out;
    a.ksize = c.s.k; a.dilation = c.s.dil; a.pad = c.s.pad; a.up = c.s.transposed ? c.s.stride : 1;
    a.in_act = in_act; a.in_slope = in_slope;
    a.out_act = out_act; a.out_scale = out_scale; a.accumulate = accumulate;
    a.in_repeat = in_repeat; a.y_f32 = y_f32;
    a.d_valid = valid; a.valid_mul = t_in / frames_max;
    rc = mb_conv1d_f16(&a, (mb_stream_t)s);
  }
  // fp16 fused ResBlock unit: y = (acc ? y : 0) + scale * (x + c2(lrelu(c1(lrelu(x)))))
  void pair_f16(const DevBuf& w, const ConvW& c1, const ConvW& c2, const void* x, int t, void* y, float slope,
                float out_scale, int accumulate) {
    if (rc) return;
    mb_resblock_pair_f16_args a;
    memset(&a, 0, sizeof(a));
    a.d_x = x; a.d_y = y; a.d_wpacked = w.p; a.d_b1 = c1.b.p; a.d_b2 = c2.b.p;
    a.batch = batch; a.channels = c1.s.c_in; a.t = t; a.ksize = c1.s.k; a.dilation = c1.s.dil;
    a.slope = slope; a.out_scale = out_scale; a.accumulate = accumulate;
    a.d_valid = valid; a.valid_mul = t / frames_max;
    rc = mb_resblock_pair_f16(&a, (mb_stream_t)s);
  }
  // fp32 fused ResBlock unit on time-major tensors: y = (acc ? y : 0) + scale * (x + c2(lrelu(c1(lrelu(x)))))
  void pair_split(const mb_gan::SPair& w, const ConvW& c1, const ConvW& c2, const void* x, int t, void* y, float slope,
                  float out_scale, int accumulate) {
    if (rc) return;
    mb_resblock_pair_split_args a;
    memset(&a, 0, sizeof(a));
    a.d_x = (const float*)x; a.d_y = (float*)y; a.d_wpacked = w.w.p; a.d_b1 = c1.b.p; a.d_b2 = c2.b.p;
    a.batch = batch; a.channels = c1.s.c_in; a.t = t; a.ksize = c1.s.k; a.dilation = c1.s.dil;
    a.slope = slope; a.out_scale = out_scale; a.unscale1 = w.us1; a.unscale2 = w.us2; a.accumulate = accumulate;
    a.d_valid = valid; a.valid_mul = t / frames_max;
    rc = mb_resblock_pair_split(&a, (mb_stream_t)s);
  }
  // time-major split conv (conv_split_tm.hip); t = input rows, the result has t rows of tc.m floats (= t * tc.rep rows of the conv's c_out)
  void conv_tm(const mb_gan::TmConv& tc, const void* x, int t, void* y, float in_slope, const void* res, int out_act,
               float out_scale = 1.f, int accumulate = 0) {
    if (rc) return;
    if (tc.w1.p && !res && !accumulate && out_scale == 1.f && (out_act == 0 || out_act == 2)) {  // one output channel: the streaming kernel
      rc = mb_conv_c1_tm((const float*)x, tc.w1.p, tc.bias1, (float*)y, batch, t, tc.c_in, tc.k, tc.dil, tc.pad, in_slope, out_act, valid,
                         t / frames_max, (mb_stream_t)s);
      return;
    }
    mb_conv_split_tm_args a;
    memset(&a, 0, sizeof(a));
    a.d_x = (const float*)x; a.d_y = (float*)y; a.d_wpacked = tc.w.p; a.d_bias = tc.bias.p; a.d_res = (const float*)res;
    a.batch = batch; a.t = t; a.c_in = tc.c_in; a.c_out = tc.m; a.ksize = tc.k; a.dilation = tc.dil; a.pad = tc.pad;
    a.in_slope = in_slope; a.unscale = tc.us; a.out_scale = out_scale; a.out_act = out_act; a.accumulate = accumulate;
    a.d_valid = valid; a.valid_mul = t / frames_max;
    rc = mb_conv_split_tm(&a, (mb_stream_t)s);
  }
  void to_tm(const void* x, void* y, int ch, int t) { if (!rc) rc = mb_f32_cm_to_tm((const float*)x, (float*)y, batch, ch, t, (mb_stream_t)s); }
  void to_cm(const void* x, void* y, int ch, int t) { if (!rc) rc = mb_f32_tm_to_cm((const float*)x, (float*)y, batch, ch, t, (mb_stream_t)s); }
  // y = conv(x) with fused pro/epilogue; lengths are per batch item.  `last` = conv_post (fp32 out).
  void conv(const ConvW& c, const void* xv, int t_in, void* yv, int in_act, float in_slope,
            const void* resv, float out_scale, int accumulate, int out_act, int in_repeat = 1,
            bool last = false) {
    if (rc) return;
    if (dtype == MB_F16) {
      conv_f16(c, xv, t_in, yv, in_act, in_slope, resv, out_scale, accumulate, out_act, in_repeat, last);
      return;
    }
    const float* x = (const float*)xv; float* y = (float*)yv; const float* res = (const float*)resv;
    mb_conv1d_args a;
    memset(&a, 0, sizeof(a));
    const int t_eff = t_in * in_repeat;
    const int t_out = c.s.transposed ? t_eff * c.s.stride : t_eff + 2 * c.s.pad - c.s.dil * (c.s.k - 1);
    a.d_x = x; a.d_wpacked = c.w.p; a.d_bias = c.b.p; a.d_res = res; a.d_y = y;
    a.x_bstride = (long long)c.s.c_in * t_in;
    a.y_bstride = (long long)c.s.c_out * t_out;
    a.res_bstride = a.y_bstride;
    a.batch = batch; a.c_in = c.s.c_in; a.c_out = c.s.c_out; a.t_in = t_eff; a.t_out = t_out;
    a.ksize = c.s.k; a.dilation = c.s.dil; a.pad = c.s.pad; a.up = c.s.transposed ? c.s.stride : 1;
    a.in_act = in_act; a.in_slope = in_slope; a.in_scale = 1.f;
    a.out_act = out_act; a.out_scale = out_scale; a.accumulate = accumulate;
    a.in_repeat = in_repeat;
    a.d_valid = valid; a.valid_mul = t_in / frames_max;
    rc = mb_conv1d(&a, (mb_stream_t)s);
  }
};
}  // namespace

extern "C" int mb_gan_forward(const mb_gan* g, const float* d_mel, int batch, int frames, float* d_wav,
                              void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  return mb_gan_forward_ex(g, d_mel, batch, frames, d_wav, nullptr, d_workspace, workspace_bytes, stream);
}

static int gan_forward_impl(const mb_gan* g, const float* d_mel, int batch, int frames, const int32_t* d_frames, float* d_wav,
                            const float* d_chan_bias, void* d_workspace, size_t workspace_bytes, mb_stream_t stream);

extern "C" int mb_gan_forward_ex(const mb_gan* g, const float* d_mel, int batch, int frames, float* d_wav,
                                 const float* d_chan_bias, void* d_workspace, size_t workspace_bytes,
                                 mb_stream_t stream) {
  return gan_forward_impl(g, d_mel, batch, frames, nullptr, d_wav, d_chan_bias, d_workspace, workspace_bytes, stream);
}

extern "C" int mb_gan_forward_ragged(const mb_gan* g, const float* d_mel, int batch, int frames, const int32_t* d_frames, float* d_wav,
                                     const float* d_chan_bias, void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  MB_REQUIRE(g && d_frames, "gan_forward_ragged: null pointer");
  MB_REQUIRE(!g->cfg.interp_ups, "gan_forward_ragged: interp_ups configs (24 kHz variant) have stage lengths that are not multiples "
                                 "of the frame count; run their utterances in equal-length batches");
  return gan_forward_impl(g, d_mel, batch, frames, d_frames, d_wav, d_chan_bias, d_workspace, workspace_bytes, stream);
}

// The fp32 generators entirely on time-major tensors (round 6): the mel is turned once, every conv is a conv_split_tm / resblock_pair_split
// launch, the waveform [T][1] comes out as it is stored.  Same dataflow as the general plan below.
static int gan_forward_tm(const mb_gan* g, const float* d_mel, int batch, int frames, const int32_t* d_frames, float* d_wav,
                          const float* d_chan_bias, void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  const mb_gan_config& c = g->cfg;
  const bool fre = c.kind == MB_GAN_FREGAN;
  const size_t per = gan_max_act(g, frames) * batch * sizeof(float);
  Arena ar(d_workspace, workspace_bytes);
  char* X = ar.take<char>(per);   // ups output = resblock input
  char* XS = ar.take<char>(per);  // stage output (mean of resblocks)
  char* XR = ar.take<char>(per);  // running x inside a resblock
  char* T = ar.take<char>(per);
  char *MELA = nullptr, *MELB = nullptr, *OUTA = nullptr, *OUTB = nullptr;
  if (fre) {
    MELA = ar.take<char>(per); MELB = ar.take<char>(per);
    OUTA = ar.take<char>(per); OUTB = ar.take<char>(per);
  }
  char* melt = ar.take<char>((size_t)batch * frames * c.num_mels * sizeof(float));
  const bool can_fork = !g->side.empty();
  std::vector<char*> XRj(c.num_kernels, XR), Tj(c.num_kernels, T);
  for (size_t j = 1; j < g->side.size(); ++j) { XRj[j] = ar.take<char>(per); Tj[j] = ar.take<char>(per); }
  const bool sum_fwd = diag_int("gan_sum_fwd") != 0;  // (A/B: the reference's order of the sum)
  const int fork_rounds = diag_int("gan_fork_rounds", 8);  // stages whose unit launches have fewer rounds of ~96-row tiles than this fork
  std::unique_lock<std::mutex> lock(const_cast<mb_gan*>(g)->mu, std::defer_lock);
  if (can_fork) lock.lock();
  hipStream_t main_s = (hipStream_t)stream;
  Launcher L{main_s, batch, MB_F32};
  L.valid = d_frames; L.frames_max = frames;
  L.to_tm(d_mel, melt, c.num_mels, frames);
  const float LRELU = 0.1f;
  const int lvl = fre ? c.num_upsamples - c.top_k : 1 << 30;
  const float inv_nk = 1.0f / (float)c.num_kernels;
  L.conv_tm(g->tmc[g->i_pre], melt, frames, XS, 1.f, nullptr, 0);  // conv_pre
  if (d_chan_bias && !L.rc) {  // VITS decoder: x = conv_pre(x) + cond(g)   vits.py:274-276
    const int C0 = c.upsample_initial_channel;
    hipLaunchKernelGGL(add_chan_bias_tm_kernel, dim3(std::min(cdiv(C0 * frames, 256), 1024), batch), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float*>(XS), d_chan_bias, C0, frames);
    MB_HIP(hipGetLastError());
  }
  int t = frames;
  const char* mel_cur = melt;
  int mel_t = frames;
  char* out_cur = nullptr;
  int out_t = 0;
  for (int i = 0; i < c.num_upsamples && !L.rc; ++i) {
    const int ch = c.upsample_initial_channel >> (i + 1);
    const int u = c.upsample_rates[i];
    char* pending_out = nullptr;
    if (fre && i >= lvl) {  // mel = cond_up[i-lvl](mel); x += mel (generator.py:142-144)
      char* mel_next = (mel_cur == MELA) ? MELB : MELA;
      const mb_gan::TmConv& cu = g->tmc[g->i_cond + (i - lvl)];
      L.conv_tm(cu, mel_cur, mel_t, mel_next, 1.f, nullptr, 0);
      mel_cur = mel_next; mel_t *= cu.rep;
      L.add(XS, mel_cur, (size_t)batch * (c.upsample_initial_channel >> i) * mel_t);
    }
    const bool ro_deferred = fre && i > lvl && out_cur;
    if (fre && i > lvl && !ro_deferred) {  // output = res_output[i-lvl-1](x) (generator.py:145-149): x is overwritten by this stage
      L.conv_tm(g->tmc[g->i_resout + (i - lvl - 1)], XS, t, OUTA, 1.f, nullptr, 0);
      pending_out = OUTA; out_t = t * u;
    }
    L.conv_tm(g->tmc[g->i_ups + i], XS, t, X, LRELU, nullptr, 0);  // x = ups[i](leaky_relu(x))
    t *= u;
    // xs = mean_j resblock_j(x): chain j ping-pongs XRj / Tj on its own stream, the mean accumulates in XS in the order of j
    const bool forked = can_fork && (long long)batch * t < (long long)fork_rounds * 96 * 256;
    if (forked && !L.rc) MB_HIP(hipEventRecord(g->ev_fork, main_s));
    // The mean is summed from the LAST chain to the first (the widest kernel first: r_{nk-1} + ... + r_0; the reference adds them in
    // index order, models.py:139-145 -- an fp32 rounding of difference): the chain with the longest launches is the critical path of a
    // forked stage, it opens the sum on the highest-priority stream and the cheap chains, which fill the gaps, close it.
    for (int o = 0; o < c.num_kernels && !L.rc; ++o) {
      const int j = sum_fwd ? o : c.num_kernels - 1 - o;
      L.s = forked ? g->side[o] : main_s;
      if (forked) MB_HIP(hipStreamWaitEvent(L.s, g->ev_fork, 0));
      auto before_last = [&]() -> int {  // the accumulate of the chain before this one is enqueued ahead of this one's
        if (forked && o > 0) MB_HIP(hipStreamWaitEvent(L.s, g->ev_last[o - 1], 0));
        return MB_OK;
      };
      if (c.resblock_type == 2) {  // ResBlock2 (models.py:63-68): x <- x + conv_d(lrelu(x)), twice
        const int b2 = g->i_rb + (i * c.num_kernels + j) * 2;
        L.conv_tm(g->tmc[b2], X, t, XRj[j], LRELU, X, 0);
        if (!L.rc) L.rc = before_last();
        L.conv_tm(g->tmc[b2 + 1], XRj[j], t, XS, LRELU, XRj[j], 0, inv_nk, o > 0);
      } else {
        const int base = g->i_rb + ((i * c.num_kernels + j) * c.num_dilations) * 2;
        const char* xr = X;
        for (int d = 0; d < c.num_dilations && !L.rc; ++d) {
          const bool last = d == c.num_dilations - 1;
          char* dst = last ? XS : ((d & 1) ? Tj[j] : XRj[j]);
          if (last) L.rc = before_last();
          L.pair_split(g->spairs[(size_t)(i * c.num_kernels + j) * c.num_dilations + d], g->convs[base + d],
                       g->convs[base + c.num_dilations + d], xr, t, dst, LRELU, last ? inv_nk : 1.f, last && o > 0);
          xr = dst;
        }
      }
      if (forked && !L.rc) MB_HIP(hipEventRecord(g->ev_last[o], L.s));
    }
    L.s = main_s;
    if (forked && !L.rc) MB_HIP(hipStreamWaitEvent(main_s, g->ev_last[c.num_kernels - 1], 0));  // (follows every chain's last launch)
    if (pending_out) {  // output = output + x (generator.py:158-159)
      L.add(pending_out, XS, (size_t)batch * ch * t);
      out_cur = pending_out;
    } else if (ro_deferred) {  // output = res_output[i-lvl-1](output) + x in one launch (x = this stage's result)
      char* dst = (out_cur == OUTA) ? OUTB : OUTA;
      L.conv_tm(g->tmc[g->i_resout + (i - lvl - 1)], out_cur, out_t, dst, 1.f, XS, 0);
      out_cur = dst; out_t *= u;
    }
  }
  const char* fin = (fre && out_cur) ? out_cur : XS;
  L.conv_tm(g->tmc[g->i_post], fin, t, d_wav, 0.01f, nullptr, 2);  // leaky_relu (default slope, models.py:146) -> conv_post -> tanh
  return L.rc;
}

static int gan_forward_impl(const mb_gan* g, const float* d_mel, int batch, int frames, const int32_t* d_frames, float* d_wav,
                            const float* d_chan_bias, void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  MB_REQUIRE(g && d_mel && d_wav, "gan_forward: null pointer");
  MB_REQUIRE(batch > 0 && frames > 0, "gan_forward: empty input (batch=%d frames=%d)", batch, frames);
  const size_t need = mb_gan_workspace_bytes(g, batch, frames);
  if (!d_workspace || workspace_bytes < need) {
    set_error("gan_forward: workspace %zu B < required %zu B", workspace_bytes, need);
    return MB_ENOMEM;
  }
  if (g->tm_all) return gan_forward_tm(g, d_mel, batch, frames, d_frames, d_wav, d_chan_bias, d_workspace, workspace_bytes, stream);
  const mb_gan_config& c = g->cfg;
  const bool fre = c.kind == MB_GAN_FREGAN;
  const bool f16 = g->dtype == MB_F16;
  const size_t esz = f16 ? 2 : sizeof(float);
  const size_t per = gan_max_act(g, frames) * batch * esz;  // bytes per activation buffer
  Arena ar(d_workspace, workspace_bytes);
  char* X = ar.take<char>(per);   // ups output = resblock input
  char* XS = ar.take<char>(per);  // stage output (mean of resblocks)
  char* XR = ar.take<char>(per);  // running x inside a resblock
  char* T = ar.take<char>(per);   // convs1 output
  char *MELA = nullptr, *MELB = nullptr, *OUTA = nullptr, *OUTB = nullptr;
  if (fre) {
    MELA = ar.take<char>(per); MELB = ar.take<char>(per);
    OUTA = ar.take<char>(per); OUTB = ar.take<char>(per);
  }
  Launcher L{(hipStream_t)stream, batch, g->dtype};
  L.valid = d_frames; L.frames_max = frames;  // ragged batch: every launch masks its input by frames[b] * (rows per frame)
  const void* mel_in = d_mel;
  if (f16) {  // [B][80][F] fp32 -> time-major fp16
    char* melh = ar.take<char>((size_t)batch * frames * c.num_mels * 2);
    L.rc = mb_f32_to_f16_tm(d_mel, melh, batch, c.num_mels, frames, stream);
    mel_in = melh;
  }
  const float LRELU = 0.1f;  // LRELU_SLOPE models.py:8
  const int lvl = fre ? c.num_upsamples - c.top_k : 1 << 30;
  const float inv_nk = 1.0f / (float)c.num_kernels;

  // conv_pre (models.py:135 / generator.py:139)
  L.conv(g->convs[g->i_pre], mel_in, frames, XS, 0, 0.f, nullptr, 1.f, 0, 0);
  if (d_chan_bias && !L.rc) {  // VITS decoder: x = conv_pre(x) + cond(g)   vits.py:274-276
    const int C0 = c.upsample_initial_channel;
    if (f16) hipLaunchKernelGGL(add_chan_bias_f16_kernel, dim3(std::min(cdiv(C0 * frames, 256), 1024), batch), dim3(256), 0,
                                (hipStream_t)stream, reinterpret_cast<_Float16*>(XS), d_chan_bias, C0, frames);
    else hipLaunchKernelGGL(add_chan_bias_f32_kernel, dim3(std::min(cdiv(frames, 256), 64), C0, batch), dim3(256), 0,
                            (hipStream_t)stream, reinterpret_cast<float*>(XS), d_chan_bias, C0, frames);
    MB_HIP(hipGetLastError());
  }
  int t = frames;                 // current length of XS
  const void* mel_cur = mel_in;   // fregan conditioning chain
  int mel_t = frames;
  char* out_cur = nullptr;        // fregan `output`
  int out_t = 0;
  for (int i = 0; i < c.num_upsamples && !L.rc; ++i) {
    const int ch = c.upsample_initial_channel >> (i + 1);
    char* pending_out = nullptr;  // res_output result waiting for "+ x"
    if (fre && i >= lvl) {
      // mel = cond_up[i-lvl](mel); x += mel (generator.py:142-144)
      char* mel_next = (mel_cur == MELA) ? MELB : MELA;
      const ConvW& cu = g->convs[g->i_cond + (i - lvl)];
      L.conv(cu, mel_cur, mel_t, mel_next, 0, 0.f, nullptr, 1.f, 0, 0);
      mel_cur = mel_next; mel_t *= cu.s.stride;
      L.add(XS, mel_cur, (size_t)batch * cu.s.c_out * mel_t);
    }
    const bool ro_deferred = fre && i > lvl && out_cur;  // res_output of `output`: runs behind the stage with "+ x" as its residual
    if (fre && i > lvl && !ro_deferred) {
      // output = res_output[i-lvl-1](x): nearest x u then 1x1 conv (generator.py:145-149); x is overwritten by this stage, so
      // this one runs here and "+ x" is a launch of its own below
      const ConvW& ro = g->convs[g->i_resout + (i - lvl - 1)];
      const int u = c.upsample_rates[i];
      char* dst = OUTA;
      L.conv(ro, XS, t, dst, 0, 0.f, nullptr, 1.f, 0, 0, u);
      pending_out = dst; out_t = t * u;
    }
    // x = ups[i](leaky_relu(x))
    const ConvW& up = g->convs[g->i_ups + i];
    // (24 kHz variant: interpolate(nearest, x u) is folded into the conv's read -- lrelu commutes with it)
    L.conv(up, XS, t, X, 1, LRELU, nullptr, 1.f, 0, 0, c.interp_ups ? c.upsample_rates[i] : 1);
    t = (int)gan_up_len(c, i, t);
    MB_REQUIRE(t > 0, "gan_forward: %d frames vanish in upsample stage %d", frames, i);
    // xs = mean_j resblock_j(x)
    bool tm_stage = !f16 && !g->spairs.empty();  // fp32 path, round 6: every unit of the stage as a time-major split pair
    for (int q = 0; q < c.num_kernels * c.num_dilations && tm_stage; ++q)
      tm_stage = g->spairs[(size_t)i * c.num_kernels * c.num_dilations + q].w.p != nullptr;
    if (tm_stage) {
      // X (channel-major, from ups[i]) -> XR time-major = the input of every ResBlock; chains ping-pong X / T; the mean accumulates
      // in XS (time-major) and is turned back into X, which becomes the stage's result
      L.to_tm(X, XR, ch, t);
      for (int j = 0; j < c.num_kernels; ++j) {
        const int base = g->i_rb + ((i * c.num_kernels + j) * c.num_dilations) * 2;
        const char* xr = XR;
        for (int d = 0; d < c.num_dilations; ++d) {
          const bool last = d == c.num_dilations - 1;
          char* dst = last ? XS : ((d & 1) ? T : X);
          L.pair_split(g->spairs[(size_t)(i * c.num_kernels + j) * c.num_dilations + d], g->convs[base + d],
                       g->convs[base + c.num_dilations + d], xr, t, dst, LRELU, last ? inv_nk : 1.f, last && j > 0);
          xr = dst;
        }
      }
      L.to_cm(XS, X, ch, t);
      std::swap(X, XS);
    } else if (f16 && !g->stage_w.empty() && g->stage_w[i].p) {  // narrow stage: every unit of every ResBlock in one launch, X -> XS
      mb_resblock_stage_f16_args a;
      memset(&a, 0, sizeof(a));
      a.d_x = X; a.d_y = XS; a.d_wpacked = g->stage_w[i].p; a.d_bias = g->stage_b[i].p;
      a.batch = batch; a.channels = ch; a.t = t; a.num_kernels = c.num_kernels; a.num_dilations = c.num_dilations;
      for (int j = 0; j < c.num_kernels; ++j) {
        const int base = g->i_rb + ((i * c.num_kernels + j) * c.num_dilations) * 2;
        a.ksize[j] = g->convs[base].s.k;
        for (int d = 0; d < c.num_dilations; ++d) a.dilation[j][d] = g->convs[base + d].s.dil;
      }
      a.slope = LRELU; a.out_scale = inv_nk;
      a.d_valid = L.valid; a.valid_mul = t / L.frames_max;
      if (!L.rc) L.rc = mb_resblock_stage_f16(&a, stream);
    } else if (!f16 && !g->s32_w.empty() && g->s32_w[i].p) {  // fp32 path, 32 channels: the whole group in one launch, X -> XS
      mb_resblock_stage_f16_args a;
      memset(&a, 0, sizeof(a));
      a.d_x = X; a.d_y = XS; a.d_wpacked = g->s32_w[i].p; a.d_bias = g->s32_b[i].p;
      a.batch = batch; a.channels = ch; a.t = t; a.num_kernels = c.num_kernels; a.num_dilations = c.num_dilations;
      for (int j = 0; j < c.num_kernels; ++j) {
        const int base = g->i_rb + ((i * c.num_kernels + j) * c.num_dilations) * 2;
        a.ksize[j] = g->convs[base].s.k;
        for (int d = 0; d < c.num_dilations; ++d) a.dilation[j][d] = g->convs[base + d].s.dil;
      }
      a.slope = LRELU; a.out_scale = inv_nk;
      a.d_valid = L.valid; a.valid_mul = t / L.frames_max;
      if (!L.rc) L.rc = mb_resblock_stage_f32(&a, stream);
    } else
    for (int j = 0; j < c.num_kernels; ++j) {
      if (c.resblock_type == 2) {  // ResBlock2 (models.py:63-68): x <- x + conv_d(lrelu(x)), twice; one launch per conv
        const int b2 = g->i_rb + (i * c.num_kernels + j) * 2;
        L.conv(g->convs[b2], X, t, XR, 1, LRELU, X, 1.f, 0, 0);
        L.conv(g->convs[b2 + 1], XR, t, XS, 1, LRELU, XR, inv_nk, j > 0, 0);
        continue;
      }
      const int base = g->i_rb + ((i * c.num_kernels + j) * c.num_dilations) * 2;
      const char* xr = X;
      if (!f16 && !g->r32.empty() && !g->r32[(size_t)i * c.num_kernels + j].empty()) {  // fp32 path, 64 channels: one launch per ResBlock / unit
        const std::vector<mb_gan::S32Launch>& plan = g->r32[(size_t)i * c.num_kernels + j];
        for (size_t l = 0; l < plan.size(); ++l) {
          const bool last = l + 1 == plan.size();
          char* dst = last ? XS : ((l & 1) ? T : XR);  // never in place: X -> XR -> T -> ... -> XS
          mb_resblock_stage_f16_args a;
          memset(&a, 0, sizeof(a));
          a.d_x = xr; a.d_y = dst; a.d_wpacked = plan[l].w.p; a.d_bias = plan[l].b.p;
          a.batch = batch; a.channels = ch; a.t = t; a.num_kernels = 1; a.num_dilations = plan[l].nu;
          a.ksize[0] = g->convs[base].s.k;
          for (int u = 0; u < plan[l].nu; ++u) a.dilation[0][u] = g->convs[base + plan[l].u0 + u].s.dil;
          a.slope = LRELU; a.out_scale = last ? inv_nk : 1.f; a.accumulate = last && j > 0;
          a.d_valid = L.valid; a.valid_mul = t / L.frames_max;
          if (!L.rc) L.rc = mb_resblock_stage_f32(&a, stream);
          xr = dst;
        }
        continue;
      }
      bool all_fused = f16 && !g->pairs.empty();
      for (int d = 0; d < c.num_dilations && all_fused; ++d)
        all_fused = g->pairs[(size_t)(i * c.num_kernels + j) * c.num_dilations + d].p != nullptr;
      if (all_fused && !g->chain_w.empty() && g->chain_w[(size_t)i * c.num_kernels + j].p) {  // the whole ResBlock in one launch
        mb_resblock_stage_f16_args a;
        memset(&a, 0, sizeof(a));
        a.d_x = X; a.d_y = XS; a.d_wpacked = g->chain_w[(size_t)i * c.num_kernels + j].p; a.d_bias = g->chain_b[(size_t)i * c.num_kernels + j].p;
        a.batch = batch; a.channels = ch; a.t = t; a.num_kernels = 1; a.num_dilations = c.num_dilations;
        a.ksize[0] = g->convs[base].s.k;
        for (int d = 0; d < c.num_dilations; ++d) a.dilation[0][d] = g->convs[base + d].s.dil;
        a.slope = LRELU; a.out_scale = inv_nk; a.accumulate = j > 0;
        a.d_valid = L.valid; a.valid_mul = t / L.frames_max;
        if (!L.rc) L.rc = mb_resblock_stage_f16(&a, stream);
        continue;
      }
      if (all_fused) {  // one launch per (convs1[d], convs2[d]); never in place: X -> XR -> T -> XR ... -> XS
        for (int d = 0; d < c.num_dilations; ++d) {
          const bool last = d == c.num_dilations - 1;
          char* dst = last ? XS : ((d & 1) ? T : XR);
          L.pair_f16(g->pairs[(size_t)(i * c.num_kernels + j) * c.num_dilations + d], g->convs[base + d],
                     g->convs[base + c.num_dilations + d], xr, t, dst, LRELU, last ? inv_nk : 1.f, last && j > 0);
          xr = dst;
        }
        continue;
      }
      for (int d = 0; d < c.num_dilations; ++d) {
        const ConvW& c1 = g->convs[base + d];
        const ConvW& c2 = g->convs[base + c.num_dilations + d];
        L.conv(c1, xr, t, T, 1, LRELU, nullptr, 1.f, 0, 0);
        const bool last = d == c.num_dilations - 1;
        if (last) L.conv(c2, T, t, XS, 1, LRELU, xr, inv_nk, j > 0, 0);
        else { L.conv(c2, T, t, XR, 1, LRELU, xr, 1.f, 0, 0); xr = XR; }
      }
    }
    if (pending_out) {  // output = output + x (generator.py:158-159)
      L.add(pending_out, XS, (size_t)batch * ch * t);
      out_cur = pending_out;
    } else if (ro_deferred) {  // output = res_output[i-lvl-1](output) + x in one launch (x = this stage's result)
      const ConvW& ro = g->convs[g->i_resout + (i - lvl - 1)];
      const int u = c.upsample_rates[i];
      char* dst = (out_cur == OUTA) ? OUTB : OUTA;
      L.conv(ro, out_cur, out_t, dst, 0, 0.f, XS, 1.f, 0, 0, u);
      out_cur = dst; out_t *= u;
    }
  }
  // x = leaky_relu(x) [default slope 0.01, models.py:146]; conv_post; tanh
  const char* fin = (fre && out_cur) ? out_cur : XS;
  L.conv(g->convs[g->i_post], fin, t, d_wav, 1, 0.01f, nullptr, 1.f, 0, 2, 1, true);
  return L.rc;
}
