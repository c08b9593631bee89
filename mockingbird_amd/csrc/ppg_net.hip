// ppg2mel: the one-shot networks either side of the decoder loop of MelDecoderMOLv2.inference
// (second half).
//
// Reference: models/ppg2mel/__init__.py
//   bnf_prenet / pitch_convs :50-98  Conv1d(k=1, no bias) -> LeakyReLU(0.1) -> InstanceNorm1d
//                                    -> Conv1d(k=2*d0, stride d0, pad d0/2) -> LeakyReLU -> InstanceNorm1d
//                                    -> Conv1d(k=2*d1, stride d1, pad d1/2) -> LeakyReLU -> InstanceNorm1d
//   inference :166-192               decoder_inputs = bnf_prenet(bnf) + pitch_convs(logf0_uv);
//                                    memory = reduce_proj(cat[decoder_inputs, F.normalize(spembs)]);
//                                    mel_postnet = mel + postnet(mel)
// models/ppg2mel/utils/cnn_postnet.py:8-52  five Conv1d(k=5) + BatchNorm1d (eval), tanh on all but the last.
//
// Every convolution runs on the fp32 MFMA implicit-GEMM primitive of conv1d.hip (the strided ones through its
// `down` field, LeakyReLU in its epilogue); eval-mode BatchNorm is folded into the conv weights on the host;
// InstanceNorm is a two-pass row kernel (one workgroup per (utterance, channel) row) that also performs the
// branch sum and writes straight into the [E + spk] concat buffer the final k=1 projection reads, which stores
// the decoder memory time-major.  fp32 throughout.
#include "common.h"

namespace mb {

// [B][T][C] -> [B][C][T], 32x32 LDS tiles
__global__ __launch_bounds__(256) void tc_to_ct_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int Cn) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* xb = x + (size_t)b * T * Cn;
  float* yb = y + (size_t)b * T * Cn;
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    tile[r][tx] = (t < T && c < Cn) ? xb[(size_t)t * Cn + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    if (t < T && c < Cn) yb[(size_t)c * T + t] = tile[tx][r];
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* s4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// torch.nn.InstanceNorm1d(affine=False, track_running_stats=False): y = (x - mean) / sqrt(var_biased + eps) per row.
// Row (b, c) of x [B][Cn][T] -> y + b*y_bstride + c*T (y may be a wider concat buffer); add: y += result.
__global__ __launch_bounds__(256) void instance_norm_kernel(const float* __restrict__ x, float* __restrict__ y, int Cn, int T,
                                                            long long y_bstride, float eps, int add) {
  __shared__ float s4[4];
  const int b = blockIdx.x / Cn, c = blockIdx.x % Cn;
  const float* xr = x + (size_t)blockIdx.x * T;
  float* yr = y + (size_t)b * y_bstride + (size_t)c * T;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) s += xr[t];
  const float mean = block_sum_256(s, s4) / (float)T;
  float q = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float d = xr[t] - mean;
    q += d * d;
  }
  const float var = block_sum_256(q, s4) / (float)T;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int t = threadIdx.x; t < T; t += 256) {
    const float v = (xr[t] - mean) * rstd;
    yr[t] = add ? yr[t] + v : v;
  }
}

// F.normalize(spembs) (x / max(||x||_2, 1e-12)) broadcast over time into rows [E, E+S) of the concat buffer
__global__ __launch_bounds__(256) void spk_rows_kernel(const float* __restrict__ spk, float* __restrict__ cat, int S, int E, int T) {
  __shared__ float s4[4];
  const int b = blockIdx.x;
  const float* sr = spk + (size_t)b * S;
  float q = 0.f;
  for (int i = threadIdx.x; i < S; i += 256) q += sr[i] * sr[i];
  const float inv = 1.0f / fmaxf(sqrtf(block_sum_256(q, s4)), 1e-12f);
  float* cb = cat + ((size_t)b * (E + S) + E) * T;
  for (size_t i = threadIdx.x; i < (size_t)S * T; i += 256) cb[i] = sr[i / T] * inv;
}

struct NetConv {
  DevBuf w, b;
  int c_in = 0, c_out = 0, k = 1, stride = 1, pad = 0;
  int load(const float* h_w, const float* h_b, int co, int ci, int k_, int stride_, int pad_) {
    c_in = ci; c_out = co; k = k_; stride = stride_; pad = pad_;
    std::vector<float> pk(mb_conv1d_packed_floats(co, ci, k_, 1));
    int rc = mb_conv1d_pack(h_w, co, ci, k_, 1, 0, pad_, pk.data());
    if (rc) return rc;
    rc = w.upload(pk.data(), pk.size());
    if (rc) return rc;
    return h_b ? b.upload(h_b, co) : MB_OK;
  }
  int t_out(int t) const { return (t + 2 * pad - (k - 1) - 1) / stride + 1; }
  void release() { w.release(); b.release(); }
};

struct Branch { NetConv c0, c1, c2; };

}  // namespace mb

using namespace mb;

struct mb_ppg2mel_net {
  mb_ppg2mel_net_config cfg;
  Branch bnf, pitch;
  NetConv reduce;
  std::vector<NetConv> post;
  void release() {
    for (Branch* br : {&bnf, &pitch}) { br->c0.release(); br->c1.release(); br->c2.release(); }
    reduce.release();
    for (auto& c : post) c.release();
  }
};

static int net_check(const mb_ppg2mel_net_config* c) {
  MB_REQUIRE(c, "ppg2mel_net: null config");
  MB_REQUIRE(c->bnf_dim >= 1 && c->spk_dim >= 1 && c->enc_dim >= 1 && c->num_mels >= 1, "ppg2mel_net: bad dims");
  MB_REQUIRE(c->down0 >= 1 && c->down0 <= 8 && c->down1 >= 1 && c->down1 <= 8, "ppg2mel_net: downsample rates %d,%d out of range",
             c->down0, c->down1);
  MB_REQUIRE(c->postnet_layers >= 2 && c->postnet_layers <= 16 && c->postnet_dim >= 1 && (c->postnet_ksize & 1),
             "ppg2mel_net: postnet needs >= 2 layers and an odd kernel");
  return MB_OK;
}

extern "C" int mb_ppg2mel_net_num_weights(const mb_ppg2mel_net_config* c) {
  if (net_check(c)) return -1;
  return 12 + 6 * c->postnet_layers;
}

extern "C" size_t mb_ppg2mel_net_weight_numel(const mb_ppg2mel_net_config* c, int i) {
  if (net_check(c) || i < 0 || i >= 12 + 6 * c->postnet_layers) return 0;
  const size_t E = c->enc_dim;
  if (i < 10) {
    const size_t cin = i < 5 ? c->bnf_dim : 2;
    switch (i % 5) {
      case 0: return E * cin;
      case 1: return E * E * 2 * c->down0;
      case 3: return E * E * 2 * c->down1;
      default: return E;
    }
  }
  if (i == 10) return E * (E + c->spk_dim);
  if (i == 11) return E;
  const int l = (i - 12) / 6, f = (i - 12) % 6, L = c->postnet_layers;
  const size_t ci = l == 0 ? c->num_mels : c->postnet_dim, co = l == L - 1 ? c->num_mels : c->postnet_dim;
  return f == 0 ? co * ci * c->postnet_ksize : co;
}

extern "C" int mb_ppg2mel_net_t_enc(const mb_ppg2mel_net_config* c, int t) {
  if (net_check(c)) return -1;
  const int t1 = (t + 2 * (c->down0 / 2) - (2 * c->down0 - 1) - 1) / c->down0 + 1;
  if (t1 < 1) return 0;
  const int t2 = (t1 + 2 * (c->down1 / 2) - (2 * c->down1 - 1) - 1) / c->down1 + 1;
  return t2 < 1 ? 0 : t2;
}

extern "C" int mb_ppg2mel_net_create(const mb_ppg2mel_net_config* c, const float* const* hw, int n, mb_ppg2mel_net** out) {
  int rc = net_check(c);
  if (rc) return rc;
  MB_REQUIRE(hw && out, "ppg2mel_net_create: null pointer");
  MB_REQUIRE(n == 12 + 6 * c->postnet_layers, "ppg2mel_net_create: expected %d weight tensors, got %d", 12 + 6 * c->postnet_layers, n);
  for (int i = 0; i < n; ++i) MB_REQUIRE(hw[i], "ppg2mel_net_create: weight %d is null", i);
  mb_ppg2mel_net* p = new mb_ppg2mel_net();
  p->cfg = *c;
  const int E = c->enc_dim;
#define NET_RC(x) do { rc = (x); if (rc) { p->release(); delete p; return rc; } } while (0)
  for (int br = 0; br < 2; ++br) {
    Branch& B = br ? p->pitch : p->bnf;
    const float* const* w = hw + 5 * br;
    NET_RC(B.c0.load(w[0], nullptr, E, br ? 2 : c->bnf_dim, 1, 1, 0));
    NET_RC(B.c1.load(w[1], w[2], E, E, 2 * c->down0, c->down0, c->down0 / 2));
    NET_RC(B.c2.load(w[3], w[4], E, E, 2 * c->down1, c->down1, c->down1 / 2));
  }
  NET_RC(p->reduce.load(hw[10], hw[11], E, E + c->spk_dim, 1, 1, 0));
  p->post.resize(c->postnet_layers);
  for (int l = 0; l < c->postnet_layers; ++l) {
    const float* const* w = hw + 12 + 6 * l;
    const int ci = l == 0 ? c->num_mels : c->postnet_dim, co = l == c->postnet_layers - 1 ? c->num_mels : c->postnet_dim;
    const int k = c->postnet_ksize;
    // eval-mode BatchNorm1d (eps 1e-5) folded into the conv: W' = W*g/sqrt(var+eps), b' = (b-mean)*g/sqrt(var+eps)+beta
    std::vector<float> wf((size_t)co * ci * k), bf(co);
    for (int o = 0; o < co; ++o) {
      const float sc = w[2][o] / sqrtf(w[5][o] + 1e-5f);
      for (size_t j = 0; j < (size_t)ci * k; ++j) wf[(size_t)o * ci * k + j] = w[0][(size_t)o * ci * k + j] * sc;
      bf[o] = (w[1][o] - w[4][o]) * sc + w[3][o];
    }
    NET_RC(p->post[l].load(wf.data(), bf.data(), co, ci, k, 1, (k - 1) / 2));
  }
#undef NET_RC
  *out = p;
  return MB_OK;
}

extern "C" void mb_ppg2mel_net_destroy(mb_ppg2mel_net* p) {
  if (!p) return;
  p->release();
  delete p;
}

extern "C" size_t mb_ppg2mel_net_workspace_bytes(const mb_ppg2mel_net* p, int batch, int t) {
  if (!p || batch <= 0 || t <= 0) return 0;
  const mb_ppg2mel_net_config& c = p->cfg;
  Arena a(nullptr, 0);
  const size_t B = batch, T = t;
  const size_t cmax = (size_t)(c.bnf_dim > c.num_mels ? c.bnf_dim : c.num_mels);
  a.take<float>(B * cmax * T);                               // transposed input
  const size_t wide = (size_t)(c.enc_dim > c.postnet_dim ? c.enc_dim : c.postnet_dim);
  a.take<float>(B * wide * T);                               // ping
  a.take<float>(B * wide * T);                               // pong
  a.take<float>(B * (size_t)(c.enc_dim + c.spk_dim) * T);    // concat (T_enc <= T)
  return a.off + 256;
}

static int net_conv(const NetConv& cv, const float* x, float* y, int B, int t_in, int out_act, float slope, const float* res,
                    int transpose_out, hipStream_t s) {
  mb_conv1d_args a;
  memset(&a, 0, sizeof(a));
  const int t_out = cv.t_out(t_in);
  a.d_x = x; a.d_wpacked = cv.w.p; a.d_bias = cv.b.p; a.d_res = res; a.d_y = y;
  a.x_bstride = (long long)cv.c_in * t_in; a.y_bstride = (long long)cv.c_out * t_out; a.res_bstride = a.y_bstride;
  a.batch = B; a.c_in = cv.c_in; a.c_out = cv.c_out; a.t_in = t_in; a.t_out = t_out;
  a.ksize = cv.k; a.dilation = 1; a.pad = cv.pad; a.up = 1; a.down = cv.stride;
  a.out_act = out_act; a.out_slope = slope; a.transpose_out = transpose_out;
  return mb_conv1d(&a, (mb_stream_t)s);
}

static void net_transpose(const float* x, float* y, int B, int T, int Cn, hipStream_t s) {
  hipLaunchKernelGGL(tc_to_ct_kernel, dim3(cdiv(T, 32), cdiv(Cn, 32), B), dim3(256), 0, s, x, y, T, Cn);
}

extern "C" int mb_ppg2mel_net_encode(const mb_ppg2mel_net* p, const float* d_bnf, const float* d_logf0_uv, const float* d_spk,
                                     int batch, int t, float* d_memory, void* d_ws, size_t ws_bytes, mb_stream_t stream) {
  MB_REQUIRE(p && d_bnf && d_logf0_uv && d_spk && d_memory && d_ws, "ppg2mel_net_encode: null pointer");
  MB_REQUIRE(batch >= 1 && t >= 1, "ppg2mel_net_encode: batch=%d t=%d", batch, t);
  const mb_ppg2mel_net_config& c = p->cfg;
  const int t_enc = mb_ppg2mel_net_t_enc(&c, t);
  MB_REQUIRE(t_enc >= 1, "ppg2mel_net_encode: %d frames are too few for the %dx%d downsampling", t, c.down0, c.down1);
  MB_REQUIRE(ws_bytes >= mb_ppg2mel_net_workspace_bytes(p, batch, t), "ppg2mel_net_encode: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Arena ar(d_ws, ws_bytes);
  const size_t B = batch, T = t;
  const size_t cmax = (size_t)(c.bnf_dim > c.num_mels ? c.bnf_dim : c.num_mels);
  float* xin = ar.take<float>(B * cmax * T);
  const size_t wide = (size_t)(c.enc_dim > c.postnet_dim ? c.enc_dim : c.postnet_dim);
  float* ping = ar.take<float>(B * wide * T);
  float* pong = ar.take<float>(B * wide * T);
  float* cat = ar.take<float>(B * (size_t)(c.enc_dim + c.spk_dim) * T);
  const int E = c.enc_dim, S = c.spk_dim;
  const long long cat_bs = (long long)(E + S) * t_enc;
  for (int br = 0; br < 2; ++br) {
    const Branch& Bn = br ? p->pitch : p->bnf;
    const int cin = br ? 2 : c.bnf_dim;
    net_transpose(br ? d_logf0_uv : d_bnf, xin, batch, t, cin, s);
    int rc = net_conv(Bn.c0, xin, ping, batch, t, 5, 0.1f, nullptr, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(instance_norm_kernel, dim3(batch * E), dim3(256), 0, s, ping, ping, E, t, (long long)E * t, 1e-5f, 0);
    const int t1 = Bn.c1.t_out(t);
    rc = net_conv(Bn.c1, ping, pong, batch, t, 5, 0.1f, nullptr, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(instance_norm_kernel, dim3(batch * E), dim3(256), 0, s, pong, pong, E, t1, (long long)E * t1, 1e-5f, 0);
    rc = net_conv(Bn.c2, pong, ping, batch, t1, 5, 0.1f, nullptr, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(instance_norm_kernel, dim3(batch * E), dim3(256), 0, s, ping, cat, E, t_enc, cat_bs, 1e-5f, br);
  }
  hipLaunchKernelGGL(spk_rows_kernel, dim3(batch), dim3(256), 0, s, d_spk, cat, S, E, t_enc);
  int rc = net_conv(p->reduce, cat, d_memory, batch, t_enc, 0, 0.f, nullptr, 1, s);
  if (rc) return rc;
  MB_HIP(hipGetLastError());
  return MB_OK;
}

extern "C" int mb_ppg2mel_net_postnet(const mb_ppg2mel_net* p, const float* d_mel, int batch, int t, float* d_out, void* d_ws,
                                      size_t ws_bytes, mb_stream_t stream) {
  MB_REQUIRE(p && d_mel && d_out && d_ws, "ppg2mel_net_postnet: null pointer");
  MB_REQUIRE(batch >= 1 && t >= 1, "ppg2mel_net_postnet: batch=%d t=%d", batch, t);
  MB_REQUIRE(ws_bytes >= mb_ppg2mel_net_workspace_bytes(p, batch, t), "ppg2mel_net_postnet: workspace too small");
  const mb_ppg2mel_net_config& c = p->cfg;
  hipStream_t s = (hipStream_t)stream;
  Arena ar(d_ws, ws_bytes);
  const size_t B = batch, T = t;
  const size_t cmax = (size_t)(c.bnf_dim > c.num_mels ? c.bnf_dim : c.num_mels);
  float* xin = ar.take<float>(B * cmax * T);
  const size_t wide = (size_t)(c.enc_dim > c.postnet_dim ? c.enc_dim : c.postnet_dim);
  float* ping = ar.take<float>(B * wide * T);
  float* pong = ar.take<float>(B * wide * T);
  net_transpose(d_mel, xin, batch, t, c.num_mels, s);
  const float* x = xin;
  const int L = c.postnet_layers;
  for (int l = 0; l < L; ++l) {
    const bool last = l == L - 1;
    float* y = last ? d_out : ((l & 1) ? pong : ping);
    // mel_outputs + postnet(mel_outputs): the residual is the time-major input itself, like the time-major store
    int rc = net_conv(p->post[l], x, y, batch, t, last ? 0 : 2, 0.f, last ? d_mel : nullptr, last ? 1 : 0, s);
    if (rc) return rc;
    x = y;
  }
  MB_HIP(hipGetLastError());
  return MB_OK;
}
