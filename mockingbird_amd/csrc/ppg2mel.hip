// ppg2mel voice-conversion decoder, inference loop.
//
// Reference: models/ppg2mel/rnn_decoder_mol.py
//   Decoder.inference :267-316, Decoder.inference_batched :318-374 (same loop body),
//   DecoderPrenet :10-22 (bias-free linears, dropout ON at inference), attend :187-198, decode :200-209;
// models/ppg2mel/utils/mol_attention.py  MOLAttention.forward :67-122 (discretised mixture of logistics).
//
// One decoder step = 8 dependent launches on the row-tile recurrent GEMM (rnn_body.h) plus two small
// kernels:
//   prenet fc0, fc1 (relu + dropout mask) -> attention LSTMCell [prenet, context, h] -> query layer 0 (relu)
//   -> mol_attention_kernel (query layer 2, mixture parameters, alpha over T_enc, context = alpha . memory)
//   -> decoder LSTMCell(s) [att_h, context, h] -> projection (+ stop row) -> ppg_finalize_kernel
// (frames / stop logits to the outputs, batch-wide stop rule).  The next step's prenet reads the last
// frame straight out of the projection buffer.  fp32 throughout.
#include <atomic>
#include "rnn.h"
#include "ppg_fast.h"
#include "ppg_resident.h"
#include "ppg_batch.h"

namespace mb {

struct MolK {
  const float* q;        // [B][Q] relu(query_layer.0(att_h))
  const float* w2;       // [3M][Q]
  const float* b2;       // [3M]
  const float* memory;   // [B][T][E]
  float* mu;             // [B][M] in/out
  float* context;        // [B][E]
  float* align_out;      // [B][max_steps][T]
  const int* skip_flag;
  int T, E, Q, M, step, max_steps;
  float eps;
};

__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // F.softplus(beta=1, threshold=20)

// One workgroup per utterance.  dynamic LDS: q[Q] | par[3M] (w, sigma, mu) | af[T+1] | alpha[T]
__global__ __launch_bounds__(256) void mol_attention_kernel(MolK a) {
  if (*a.skip_flag) return;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_q = sm;
  float* s_mp = s_q + a.Q;           // [3M] raw mixture parameters, then (w, sigma, mu_cur)
  float* s_af = s_mp + 3 * a.M + 1;  // [T+1]
  float* s_al = s_af + a.T + 1;      // [T]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < a.Q; i += 256) s_q[i] = a.q[(size_t)b * a.Q + i];
  __syncthreads();
  // mixture_params = query_layer.2(q)   :75  -- one wave-reduction per output
  for (int o = wave; o < 3 * a.M; o += 4) {
    float acc = 0.f;
    for (int k = lane; k < a.Q; k += 64) acc += a.w2[(size_t)o * a.Q + k] * s_q[k];
    acc = wave_sum(acc);
    if (lane == 0) s_mp[o] = acc + a.b2[o];
  }
  __syncthreads();
  if (tid == 0) {  // w = softmax(w_hat) + eps; sigma = softplus(sigma_hat) + eps; mu = mu_prev + softplus(Delta_hat)  :92-96
    float mx = -INFINITY;
    for (int m = 0; m < a.M; ++m) mx = fmaxf(mx, s_mp[m]);
    float se = 0.f;
    for (int m = 0; m < a.M; ++m) se += expf(s_mp[m] - mx);
    for (int m = 0; m < a.M; ++m) {
      const float w = expf(s_mp[m] - mx) / se + a.eps;
      const float sg = softplusf_(s_mp[a.M + m]) + a.eps;
      const float mu = a.mu[(size_t)b * a.M + m] + softplusf_(s_mp[2 * a.M + m]);
      a.mu[(size_t)b * a.M + m] = mu;
      s_mp[m] = w; s_mp[a.M + m] = sg; s_mp[2 * a.M + m] = mu;
    }
  }
  __syncthreads();
  // alpha_full[j] = sum_m w_m / (1 + sigmoid((mu_m - (j + 0.5)) / sigma_m)),  j = 0..T   :101-107
  for (int j = tid; j <= a.T; j += 256) {
    float s = 0.f;
    const float pos = (float)j + 0.5f;
    for (int m = 0; m < a.M; ++m) {
      const float z = (s_mp[2 * a.M + m] - pos) / s_mp[a.M + m];
      s += s_mp[m] * (1.f / (1.f + 1.f / (1.f + expf(-z))));
    }
    s_af[j] = s;
  }
  __syncthreads();
  float* al = a.align_out + ((size_t)b * a.max_steps + a.step) * a.T;
  for (int t = tid; t < a.T; t += 256) {  // alpha_t = diff; zeros -> eps   :108-109
    float v = s_af[t + 1] - s_af[t];
    if (v == 0.f) v = a.eps;
    s_al[t] = v;
    al[t] = v;
  }
  __syncthreads();
  // context = alpha . memory   :115
  const float* mem = a.memory + (size_t)b * a.T * a.E;
  for (int e = tid; e < a.E; e += 256) {
    float acc = 0.f;
    for (int t = 0; t < a.T; ++t) acc += s_al[t] * mem[(size_t)t * a.E + e];
    a.context[(size_t)b * a.E + e] = acc;
  }
}

struct PpgFinK {
  const float* y;   // [B][ldy]: r*num_mels frame values, then the stop logit
  float* mel_out;   // [B][max_steps][r*num_mels]
  float* stop_out;  // [B][max_steps] logits
  int* done; int* n_steps;
  int B, ldy, RM, step, max_steps, min_steps;
  float thr;
};

// one workgroup: scatter this step's frames / stop logits, batch-wide stop rule (:301-305, :349-354)
__global__ __launch_bounds__(256) void ppg_finalize_kernel(PpgFinK a) {
  if (*a.done) return;
  const int tid = threadIdx.x;
  __shared__ int s_below;
  if (tid == 0) s_below = 0;
  __syncthreads();
  for (int i = tid; i < a.B * a.RM; i += 256) {
    const int b = i / a.RM, q = i - b * a.RM;
    a.mel_out[((size_t)b * a.max_steps + a.step) * a.RM + q] = a.y[(size_t)b * a.ldy + q];
  }
  for (int b = tid; b < a.B; b += 256) {
    const float lg = a.y[(size_t)b * a.ldy + a.RM];
    a.stop_out[(size_t)b * a.max_steps + a.step] = lg;
    if (!(1.f / (1.f + expf(-lg)) > a.thr)) atomicAdd(&s_below, 1);
  }
  __syncthreads();
  if (tid == 0) {
    *a.n_steps = a.step + 1;
    if (s_below == 0 && a.step + 1 >= a.min_steps) *a.done = 1;
  }
}

}  // namespace mb

using namespace mb;

struct mb_ppg2mel {
  mb_ppg2mel_config cfg;
  std::vector<DevBuf> pre_w;  // prenet layers (row-tile packed)
  DevBuf zero_bias;           // the prenet linears are bias-free; a zero bias keeps the specialised instance
  DevBuf att_w, att_bih, att_bhh;
  DevBuf q0_w, q0_b, q2_w, q2_b;
  std::vector<DevBuf> dec_w, dec_bih, dec_bhh;
  DevBuf out_w, out_b;  // projection rows then the stop row
  // fast step (ppg_fast.h), production dims only
  bool fast = false;
  DevBuf f_att_p, f_att_c, f_att_h, f_att_b4, f_dec_x, f_dec_h, f_dec_b4, f_fc0_w, f_fc0_b;
  // resident loop at batch 1 (ppg_resident.h): per-workgroup LDS images of the same weights
  bool resident = false;
  DevBuf r_att, r_w1, r_dec, r_q0, r_out;
  // resident loop for a batch of 2..32 (ppg_batch.h): split fp16 images (A fragments of v_mfma_f32_16x16x32_f16) + 2^-s per role
  bool batch_resident = false;
  DevBuf b_att_w1, b_att_wx, b_att_wc, b_att_wh, b_q0, b_dec_wa, b_dec_wc, b_dec_wh, b_out_wh, b_out_wc;
  float b_us[5] = {1.f, 1.f, 1.f, 1.f, 1.f};  // prenet.1, attention LSTM, query_layer.0, decoder LSTM, output rows
  int batch_cus = -1;
  int last_batch_fallback = 0;  // 0 none, 1 lost hand-off, 2 operand range
  int resident_cus = -1;  // compute units a resident launch may count on (-1: not probed yet, 0: none)
  int* h_abort = nullptr;
  hipEvent_t ev_res = nullptr;
  struct GraphKey {
    const void *mem, *drop, *mel, *align, *stop, *ws; int B, T, max_steps, min_steps, G; float thr;
    bool operator==(const GraphKey& o) const {  // field by field (padding bytes are unspecified)
      return mem == o.mem && drop == o.drop && mel == o.mel && align == o.align && stop == o.stop && ws == o.ws && B == o.B && T == o.T &&
             max_steps == o.max_steps && min_steps == o.min_steps && G == o.G && thr == o.thr;
    }
  } gkey = {};
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int* h_flags = nullptr;
  hipEvent_t ev_flags[2] = {nullptr, nullptr}, ev_in = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  hipStream_t loop_stream = nullptr;
  int last_steps = 0, last_chain_steps = 0; bool timed = false, last_resident = false;
  void drop_graph() {
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
  }
};

// a resident launch (ppg_resident.h) once lost a hand-off on this device: stop defaulting to it there
static std::atomic<bool> g_ppg_resident_failed[64] = {};  // written by whichever host thread sees the abort word: atomic
static std::atomic<bool> g_ppg_batch_failed[64] = {};     // the batch kernel's own memo (160 + B workgroups against 217: a lost hand-off of one says nothing about the other)

static int ppg_shapes(const mb_ppg2mel_config* c, std::vector<size_t>* numel) {
  MB_REQUIRE(c, "ppg2mel: null config");
  MB_REQUIRE(c->n_prenet >= 1 && c->n_prenet <= 4 && c->num_decoder_rnn_layer >= 1 && c->num_decoder_rnn_layer <= 4,
             "ppg2mel: unsupported layer counts");
  MB_REQUIRE(c->enc_dim % 16 == 0 && c->attention_rnn_dim % 16 == 0 && c->decoder_rnn_dim % 16 == 0 && c->num_mels % 16 == 0,
             "ppg2mel: dims must be multiples of 16");
  for (int i = 0; i < c->n_prenet; ++i) MB_REQUIRE(c->prenet_dims[i] % 16 == 0, "ppg2mel: prenet dims must be multiples of 16");
  MB_REQUIRE(c->num_mixtures >= 1 && c->num_mixtures <= 16 && c->frames_per_step >= 1, "ppg2mel: bad mixture / frames_per_step");
  const size_t E = c->enc_dim, A = c->attention_rnn_dim, D = c->decoder_rnn_dim, nm = c->num_mels, r = c->frames_per_step;
  const size_t Q = 256, M = c->num_mixtures;
  numel->clear();
  size_t in = nm;
  for (int i = 0; i < c->n_prenet; ++i) { numel->push_back((size_t)c->prenet_dims[i] * in); in = c->prenet_dims[i]; }
  const size_t P = in;
  numel->push_back(4 * A * (P + E)); numel->push_back(4 * A * A); numel->push_back(4 * A); numel->push_back(4 * A);
  numel->push_back(Q * A); numel->push_back(Q); numel->push_back(3 * M * Q); numel->push_back(3 * M);
  for (int i = 0; i < c->num_decoder_rnn_layer; ++i) {
    const size_t kin = i == 0 ? E + A : D;
    numel->push_back(4 * D * kin); numel->push_back(4 * D * D); numel->push_back(4 * D); numel->push_back(4 * D);
  }
  const size_t kout = c->concat_context_to_last ? D + E : D;
  numel->push_back(nm * r * kout); numel->push_back(nm * r); numel->push_back(kout); numel->push_back(1);
  return MB_OK;
}

extern "C" int mb_ppg2mel_num_weights(const mb_ppg2mel_config* cfg) {
  std::vector<size_t> n;
  return ppg_shapes(cfg, &n) ? MB_EINVAL : (int)n.size();
}
extern "C" size_t mb_ppg2mel_weight_numel(const mb_ppg2mel_config* cfg, int index) {
  std::vector<size_t> n;
  if (ppg_shapes(cfg, &n) || index < 0 || index >= (int)n.size()) return 0;
  return n[index];
}

extern "C" void mb_ppg2mel_destroy(mb_ppg2mel* p) {
  if (!p) return;
  for (auto& b : p->pre_w) b.release();
  for (auto& b : p->dec_w) b.release();
  for (auto& b : p->dec_bih) b.release();
  for (auto& b : p->dec_bhh) b.release();
  DevBuf* bs[] = {&p->zero_bias, &p->att_w, &p->att_bih, &p->att_bhh, &p->q0_w, &p->q0_b, &p->q2_w, &p->q2_b, &p->out_w, &p->out_b,
                  &p->f_att_p, &p->f_att_c, &p->f_att_h, &p->f_att_b4, &p->f_dec_x, &p->f_dec_h, &p->f_dec_b4, &p->f_fc0_w, &p->f_fc0_b,
                  &p->r_att, &p->r_w1, &p->r_dec, &p->r_q0, &p->r_out,
                  &p->b_att_w1, &p->b_att_wx, &p->b_att_wc, &p->b_att_wh, &p->b_q0, &p->b_dec_wa, &p->b_dec_wc, &p->b_dec_wh, &p->b_out_wh, &p->b_out_wc};
  for (DevBuf* b : bs) b->release();
  p->drop_graph();
  if (p->h_flags) (void)hipHostFree(p->h_flags);
  if (p->h_abort) (void)hipHostFree(p->h_abort);
  hipEvent_t evs[] = {p->ev_flags[0], p->ev_flags[1], p->ev_in, p->ev_t0, p->ev_t1, p->ev_res};
  for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
  p->loop_stream = nullptr;  // (borrowed from the pool, common.h)
  delete p;
}

extern "C" int mb_ppg2mel_create(const mb_ppg2mel_config* cfg, const float* const* hw, int n_weights, mb_ppg2mel** out) {
  MB_REQUIRE(out && hw, "ppg2mel_create: null pointer");
  std::vector<size_t> shapes;
  int rc = ppg_shapes(cfg, &shapes);
  if (rc) return rc;
  MB_REQUIRE(n_weights == (int)shapes.size(), "ppg2mel_create: expected %d weight tensors, got %d", (int)shapes.size(), n_weights);
  mb_ppg2mel* p = new mb_ppg2mel();
  p->cfg = *cfg;
  const int E = cfg->enc_dim, A = cfg->attention_rnn_dim, D = cfg->decoder_rnn_dim, nm = cfg->num_mels, r = cfg->frames_per_step;
  const int Q = 256, M = cfg->num_mixtures;
  std::vector<float> rows, packed;
  int ix = 0;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  p->pre_w.resize(cfg->n_prenet);
  int in = nm, maxp = 0;
  for (int i = 0; i < cfg->n_prenet; ++i) {
    pack_rowtile(hw[ix++], cfg->prenet_dims[i], in, 4, &packed);
    RC(p->pre_w[i].upload(packed.data(), packed.size()));
    in = cfg->prenet_dims[i];
    maxp = std::max(maxp, in);
  }
  const int P = in;
  {
    std::vector<float> z(maxp, 0.f);
    RC(p->zero_bias.upload(z.data(), z.size()));
  }
  // attention_rnn = LSTMCell(P + E, A): input order [prenet, context] (attend :188)
  cell_rows(hw[ix], P + E, P + E, hw[ix + 1], A, A, 4, &rows);
  pack_rowtile(rows.data(), 4 * A, P + E + A, 4, &packed);
  RC(p->att_w.upload(packed.data(), packed.size())); RC(p->att_bih.upload(hw[ix + 2], 4 * A)); RC(p->att_bhh.upload(hw[ix + 3], 4 * A));
  ix += 4;
  pack_rowtile(hw[ix], Q, A, 4, &packed); RC(p->q0_w.upload(packed.data(), packed.size())); RC(p->q0_b.upload(hw[ix + 1], Q));
  RC(p->q2_w.upload(hw[ix + 2], (size_t)3 * M * Q)); RC(p->q2_b.upload(hw[ix + 3], 3 * M));
  ix += 4;
  p->dec_w.resize(cfg->num_decoder_rnn_layer); p->dec_bih.resize(cfg->num_decoder_rnn_layer); p->dec_bhh.resize(cfg->num_decoder_rnn_layer);
  for (int i = 0; i < cfg->num_decoder_rnn_layer; ++i) {
    const int kin = i == 0 ? A + E : D;  // layer 0 input order [attention_hidden, context] (attend :194-195)
    cell_rows(hw[ix], kin, kin, hw[ix + 1], D, D, 4, &rows);
    pack_rowtile(rows.data(), 4 * D, kin + D, 4, &packed);
    RC(p->dec_w[i].upload(packed.data(), packed.size())); RC(p->dec_bih[i].upload(hw[ix + 2], 4 * D)); RC(p->dec_bhh[i].upload(hw[ix + 3], 4 * D));
    ix += 4;
  }
  {  // linear_projection rows then the stop_layer row: one launch produces both (:287-288)
    const int kout = cfg->concat_context_to_last ? D + E : D;
    std::vector<float> w((size_t)(nm * r + 1) * kout), b(nm * r + 1);
    memcpy(w.data(), hw[ix], sizeof(float) * (size_t)nm * r * kout);
    memcpy(b.data(), hw[ix + 1], sizeof(float) * nm * r);
    memcpy(w.data() + (size_t)nm * r * kout, hw[ix + 2], sizeof(float) * kout);
    b[nm * r] = hw[ix + 3][0];
    pack_rowtile(w.data(), nm * r + 1, kout, 4, &packed);
    RC(p->out_w.upload(packed.data(), packed.size())); RC(p->out_b.upload(b.data(), b.size()));
    ix += 4;
  }
  p->fast = cfg->n_prenet == 2 && cfg->prenet_dims[0] == 256 && cfg->prenet_dims[1] == 128 && E == 256 && A == 512 && D == 512 &&
            cfg->num_decoder_rnn_layer == 1 && cfg->concat_context_to_last && (nm * r) % 16 == 0 && nm % 4 == 0 && 3 * M <= 16;  // (ppg_mol_kernel: 16 parameter rows)
  if (p->fast && !rc) {
    // weight list: 0-1 prenet, 2-5 attention_rnn (w_ih [4A][P+E], w_hh, b_ih, b_hh), 6-9 query layers, 10-13 decoder rnn, 14-17 projection / stop
    const int Pn = cfg->prenet_dims[1];
    const float *a_wih = hw[2], *a_whh = hw[3], *a_bih = hw[4], *a_bhh = hw[5];
    const float *d_wih = hw[10], *d_whh = hw[11], *d_bih = hw[12], *d_bhh = hw[13];
    const float *w_proj = hw[14], *b_proj = hw[15];
    cell_rows(a_wih, Pn, Pn + E, a_whh, 0, A, 4, &rows); pack_rowtile(rows.data(), 4 * A, Pn, 4, &packed); RC(p->f_att_p.upload(packed.data(), packed.size()));
    cell_rows(a_wih + Pn, E, Pn + E, a_whh, 0, A, 4, &rows); pack_rowtile(rows.data(), 4 * A, E, 4, &packed); RC(p->f_att_c.upload(packed.data(), packed.size()));
    cell_rows(a_whh, A, A, a_whh, 0, A, 4, &rows); pack_rowtile(rows.data(), 4 * A, A, 4, &packed); RC(p->f_att_h.upload(packed.data(), packed.size()));
    cell_rows(d_wih, A + E, A + E, d_whh, 0, D, 4, &rows); pack_rowtile(rows.data(), 4 * D, A + E, 4, &packed); RC(p->f_dec_x.upload(packed.data(), packed.size()));
    cell_rows(d_whh, D, D, d_whh, 0, D, 4, &rows); pack_rowtile(rows.data(), 4 * D, D, 4, &packed); RC(p->f_dec_h.upload(packed.data(), packed.size()));
    std::vector<float> b4((size_t)A * 4);
    for (int j = 0; j < A; ++j) for (int g = 0; g < 4; ++g) b4[(size_t)j * 4 + g] = a_bih[g * A + j] + a_bhh[g * A + j];
    RC(p->f_att_b4.upload(b4.data(), b4.size()));
    b4.assign((size_t)D * 4, 0.f);
    for (int j = 0; j < D; ++j) for (int g = 0; g < 4; ++g) b4[(size_t)j * 4 + g] = d_bih[g * D + j] + d_bhh[g * D + j];
    RC(p->f_dec_b4.upload(b4.data(), b4.size()));
    {  // prenet.0 (bias-free, [P0][nm]) folded through the projection's LAST frame rows (r-1)*nm + m: W' = W0 . Wp_last, b' = W0 . bp_last
      const int P0 = cfg->prenet_dims[0], kout = D + E;
      std::vector<float> wf((size_t)P0 * kout), bf(P0);
      std::vector<double> acc(kout);
      for (int o = 0; o < P0; ++o) {
        std::fill(acc.begin(), acc.end(), 0.0);
        double ab = 0.0;
        for (int m = 0; m < nm; ++m) {
          const double f = hw[0][(size_t)o * nm + m];
          const float* pr = w_proj + ((size_t)(r - 1) * nm + m) * kout;
          for (int k2 = 0; k2 < kout; ++k2) acc[k2] += f * (double)pr[k2];
          ab += f * (double)b_proj[(r - 1) * nm + m];
        }
        for (int k2 = 0; k2 < kout; ++k2) wf[(size_t)o * kout + k2] = (float)acc[k2];
        bf[o] = (float)ab;
      }
      pack_rowtile(wf.data(), P0, kout, 4, &packed); RC(p->f_fc0_w.upload(packed.data(), packed.size()));
      RC(p->f_fc0_b.upload(bf.data(), bf.size()));
      // resident loop (ppg_resident.h): float4 chunk c of thread t at (c * threads + t) * 4, slice sl = t & 15 owns k = (c * 16 + sl) * 4 .. + 3
      const int RMr = nm * r, nmel = RMr / 16;
      p->resident = 17 + nmel <= 32;
      if (p->resident) {
        const float *w_stop = hw[16], *q0w = hw[6], *w1 = hw[1];
        std::vector<float> img;
        img.assign((size_t)PR_ATT * PR_IMG_ATT, 0.f);
        for (int g = 0; g < PR_ATT; ++g)
          for (int c = 0; c < 14; ++c)
            for (int t = 0; t < 512; ++t) {
              const int rl = t >> 4, sl = t & 15, row = (rl & 3) * A + g * 8 + (rl >> 2);
              for (int e = 0; e < 4; ++e) {
                float v;
                if (c < 2) v = a_wih[(size_t)row * (Pn + E) + (c * 16 + sl) * 4 + e];
                else if (c < 6) v = a_wih[(size_t)row * (Pn + E) + Pn + ((c - 2) * 16 + sl) * 4 + e];
                else v = a_whh[(size_t)row * A + ((c - 6) * 16 + sl) * 4 + e];
                img[(size_t)g * PR_IMG_ATT + ((size_t)c * 512 + t) * 4 + e] = v;
              }
            }
        RC(p->r_att.upload(img.data(), img.size()));
        img.assign((size_t)PR_IMG_W1, 0.f);
        for (int c = 0; c < 16; ++c)
          for (int t = 0; t < 512; ++t)
            for (int e = 0; e < 4; ++e) img[((size_t)c * 512 + t) * 4 + e] = w1[(size_t)(t >> 2) * P0 + (c * 4 + (t & 3)) * 4 + e];
        RC(p->r_w1.upload(img.data(), img.size()));
        img.assign((size_t)PR_DEC * PR_IMG_DEC, 0.f);
        for (int d = 0; d < PR_DEC; ++d)
          for (int t = 0; t < 256; ++t) {
            const int rl = t >> 4, sl = t & 15, row = (rl & 3) * D + d * 4 + (rl >> 2);
            for (int c = 0; c < 12; ++c)
              for (int e = 0; e < 4; ++e)
                img[(size_t)d * PR_IMG_DEC + ((size_t)c * 256 + t) * 4 + e] = d_wih[(size_t)row * (A + E) + (c * 16 + sl) * 4 + e];
            for (int c = 0; c < 8; ++c)
              for (int e = 0; e < 4; ++e)
                img[(size_t)d * PR_IMG_DEC + ((size_t)(12 + c) * 256 + t) * 4 + e] = d_whh[(size_t)row * D + (c * 16 + sl) * 4 + e];
          }
        RC(p->r_dec.upload(img.data(), img.size()));
        img.assign((size_t)PR_Q0 * PR_IMG_Q0, 0.f);
        for (int j = 0; j < PR_Q0; ++j)
          for (int c = 0; c < 8; ++c)
            for (int t = 0; t < 512; ++t)
              for (int e = 0; e < 4; ++e)
                img[(size_t)j * PR_IMG_Q0 + ((size_t)c * 512 + t) * 4 + e] = q0w[(size_t)(j * 32 + (t >> 4)) * A + (c * 16 + (t & 15)) * 4 + e];
        RC(p->r_q0.upload(img.data(), img.size()));
        img.assign((size_t)PR_OUT * PR_IMG_OUT, 0.f);
        for (int j = 0; j < PR_OUT; ++j)
          for (int t = 0; t < 512; ++t) {
            const int lr = t >> 4, sl = t & 15;
            const float* src = lr == 0 ? w_stop : lr <= 16 ? wf.data() + (size_t)(j * 16 + lr - 1) * kout
                                                 : lr < 17 + nmel ? w_proj + (size_t)(j * nmel + lr - 17) * kout : nullptr;
            if (!src) continue;
            for (int c = 0; c < 12; ++c)
              for (int e = 0; e < 4; ++e) img[(size_t)j * PR_IMG_OUT + ((size_t)c * 512 + t) * 4 + e] = src[(c * 16 + sl) * 4 + e];
          }
        RC(p->r_out.upload(img.data(), img.size()));
      }
      // resident loop for 2..32 utterances (ppg_batch.h): the same matrices as split fp16 A fragments, row tiles in the roles' order
      if (RMr % 16 == 0 && RMr / 16 <= 16 && M <= 5) {
        auto cols = [](const float* w, int rows_n, int ld, int c0, int nc, auto row_of) {  // rows_n rows (tile order via row_of) x columns [c0, c0 + nc)
          std::vector<float> o((size_t)rows_n * nc);
          for (int r2 = 0; r2 < rows_n; ++r2) {
            const int src = row_of(r2);
            if (src < 0) { std::fill(o.begin() + (size_t)r2 * nc, o.begin() + (size_t)(r2 + 1) * nc, 0.f); continue; }
            memcpy(&o[(size_t)r2 * nc], w + (size_t)src * ld + c0, sizeof(float) * nc);
          }
          return o;
        };
        auto natural = [](int r2) { return r2; };
        auto lstm_a = [A](int r2) { return (r2 & 3) * A + (r2 >> 2); };  // unit-major tile rows: row 4 u + gate <- gate-major source row gate A + u
        auto lstm_d = [D](int r2) { return (r2 & 3) * D + (r2 >> 2); };
        auto up = [&](const std::vector<float>& m, int n_tiles, int K, int nwv, int sexp, DevBuf* dst) {
          std::vector<unsigned short> img16;
          pb_pack(m, n_tiles, K, nwv, sexp, &img16);
          return dst->upload(reinterpret_cast<const float*>(img16.data()), img16.size() / 2);
        };
        const float *q0w = hw[6], *w1 = hw[1], *w_stop = hw[16];
        {  // prenet.1 [128][256]
          std::vector<float> m = cols(w1, Pn, P0, 0, P0, natural);
          const int e = pb_scale_exp({&m});
          RC(up(m, Pn / 16, P0, 8, e, &p->b_att_w1)); p->b_us[0] = std::ldexp(1.f, -e);
        }
        {  // attention LSTM: [prenet 128 | context 256] of w_ih, w_hh 512; one scale for the three parts of a gate sum
          std::vector<float> mx = cols(a_wih, 4 * A, Pn + E, 0, Pn, lstm_a), mc = cols(a_wih, 4 * A, Pn + E, Pn, E, lstm_a), mh = cols(a_whh, 4 * A, A, 0, A, lstm_a);
          const int e = pb_scale_exp({&mx, &mc, &mh});
          RC(up(mx, A / 4, Pn, 4, e, &p->b_att_wx)); RC(up(mc, A / 4, E, 8, e, &p->b_att_wc)); RC(up(mh, A / 4, A, 8, e, &p->b_att_wh));
          p->b_us[1] = std::ldexp(1.f, -e);
        }
        {  // query_layer.0 [256][512]
          std::vector<float> m = cols(q0w, Q, A, 0, A, natural);
          const int e = pb_scale_exp({&m});
          RC(up(m, Q / 16, A, 8, e, &p->b_q0)); p->b_us[2] = std::ldexp(1.f, -e);
        }
        {  // decoder LSTM: [attention_hidden 512 | context 256] of w_ih, w_hh 512
          std::vector<float> ma = cols(d_wih, 4 * D, A + E, 0, A, lstm_d), mc = cols(d_wih, 4 * D, A + E, A, E, lstm_d), mh = cols(d_whh, 4 * D, D, 0, D, lstm_d);
          const int e = pb_scale_exp({&ma, &mc, &mh});
          RC(up(ma, D / 4, A, 8, e, &p->b_dec_wa)); RC(up(mc, D / 4, E, 8, e, &p->b_dec_wc)); RC(up(mh, D / 4, D, 8, e, &p->b_dec_wh));
          p->b_us[3] = std::ldexp(1.f, -e);
        }
        {  // output rows over [h 512 | context 256]: 16 tiles of prenet.0', RM / 16 projection tiles, the stop row (row 0 of its tile)
          const int n_tiles = P0 / 16 + RMr / 16 + 1;
          std::vector<float> all((size_t)n_tiles * 16 * kout, 0.f);
          memcpy(all.data(), wf.data(), sizeof(float) * (size_t)P0 * kout);
          memcpy(all.data() + (size_t)P0 * kout, w_proj, sizeof(float) * (size_t)RMr * kout);
          memcpy(all.data() + (size_t)(P0 + RMr) * kout, w_stop, sizeof(float) * kout);
          std::vector<float> mh = cols(all.data(), n_tiles * 16, kout, 0, D, natural), mc = cols(all.data(), n_tiles * 16, kout, D, E, natural);
          const int e = pb_scale_exp({&mh, &mc});
          RC(up(mh, n_tiles, D, 8, e, &p->b_out_wh)); RC(up(mc, n_tiles, E, 8, e, &p->b_out_wc));
          p->b_us[4] = std::ldexp(1.f, -e);
        }
        p->batch_resident = !rc;
      }
    }
    if (!rc && (hipHostMalloc((void**)&p->h_flags, sizeof(int) * 16) != hipSuccess ||
                pool_stream(0, &p->loop_stream) != MB_OK ||
                hipEventCreateWithFlags(&p->ev_in, hipEventDisableTiming) != hipSuccess ||
                hipEventCreate(&p->ev_t0) != hipSuccess || hipEventCreate(&p->ev_t1) != hipSuccess ||
                hipEventCreateWithFlags(&p->ev_flags[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&p->ev_flags[1], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&p->ev_res, hipEventDisableTiming) != hipSuccess ||
                hipHostMalloc((void**)&p->h_abort, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess)) {
      set_error("ppg2mel_create: stream / events / pinned flags");
      rc = MB_EHIP;
    }
  }
#undef RC
  if (rc) { mb_ppg2mel_destroy(p); return rc; }
  *out = p;
  return MB_OK;
}

namespace {
struct PpgLayout {
  float *pbuf[4], *att_h, *att_c, *ctx, *q, *mu, *dec_h[4], *dec_c[4], *y, *zero;
  // fast step: FM activations (p0, p1, att_h, q, ctx, dec_h), CM1 cells, CM4 gate parts
  float *f_p0, *f_p1, *f_ah, *f_q, *f_ctx, *f_dh, *f_ac, *f_dc, *f_prec, *f_preh, *f_pred;
  size_t f_bytes;
  int* flags;
  unsigned long long* px;  // resident loop (ppg_resident.h): granule exchange area + abort word + diagnostics marks
  size_t bytes;
  int ldy;
};
void ppg_layout(const mb_ppg2mel* p, int B, void* base, PpgLayout* L) {
  const mb_ppg2mel_config& c = p->cfg;
  Arena ar(base, (size_t)-1);
  for (int i = 0; i < c.n_prenet; ++i) L->pbuf[i] = ar.take<float>((size_t)B * c.prenet_dims[i]);
  L->att_h = ar.take<float>((size_t)2 * B * c.attention_rnn_dim); L->att_c = ar.take<float>((size_t)2 * B * c.attention_rnn_dim);
  L->ctx = ar.take<float>((size_t)B * c.enc_dim);
  L->q = ar.take<float>((size_t)B * 256);
  L->mu = ar.take<float>((size_t)B * c.num_mixtures);
  for (int i = 0; i < c.num_decoder_rnn_layer; ++i) {
    L->dec_h[i] = ar.take<float>((size_t)2 * B * c.decoder_rnn_dim); L->dec_c[i] = ar.take<float>((size_t)2 * B * c.decoder_rnn_dim);
  }
  L->ldy = (c.num_mels * c.frames_per_step + 1 + 3) & ~3;  // rows of 16-byte multiples: the next prenet reads frames in place
  L->y = ar.take<float>((size_t)B * L->ldy);
  L->zero = ar.take<float>((size_t)B * L->ldy);
  {
    const int nta = (B + 15) / 16, A = c.attention_rnn_dim, D = c.decoder_rnn_dim;
    L->f_p0 = ar.take<float>(fm_floats(c.prenet_dims[0], nta));
    const size_t start = ar.off - fm_floats(c.prenet_dims[0], nta) * sizeof(float);
    L->f_p1 = ar.take<float>(fm_floats(c.prenet_dims[c.n_prenet - 1], nta));
    L->f_ah = ar.take<float>(fm_floats(A, nta)); L->f_q = ar.take<float>(fm_floats(256, nta));
    L->f_ctx = ar.take<float>(fm_floats(c.enc_dim, nta)); L->f_dh = ar.take<float>(fm_floats(D, nta));
    L->f_ac = ar.take<float>(cm_items(A, nta)); L->f_dc = ar.take<float>(cm_items(D, nta));
    L->f_prec = ar.take<float>(4 * cm_items(A, nta)); L->f_preh = ar.take<float>(4 * cm_items(A, nta));
    L->f_pred = ar.take<float>(4 * cm_items(D, nta));
    L->f_bytes = ar.off - start;
  }
  L->flags = ar.take<int>(16);
  L->px = ar.take<unsigned long long>(std::max(pr_exchange_bytes(), pb_exchange_bytes()) / 8);
  L->bytes = ar.off + 256;
}
}  // namespace

extern "C" size_t mb_ppg2mel_workspace_bytes(const mb_ppg2mel* p, int batch) {
  if (!p || batch <= 0) return 0;
  PpgLayout L;
  ppg_layout(p, batch, nullptr, &L);
  return L.bytes;
}

// ---- fast step (ppg_fast.h): 6 launches per step, hipGraph replays of 16 steps, stop flag polled one replay behind ----
static int ppg_fast_loop_body(mb_ppg2mel* p, const PpgLayout& L, const float* d_memory, int B, int T, int max_steps, int min_steps,
                              float thr, const float* d_dropout, uint64_t seed, float* d_mel, float* d_align, float* d_stop,
                              void* d_workspace, hipStream_t s, int* steps_out) {
  const mb_ppg2mel_config& c = p->cfg;
  const int E = c.enc_dim, A = c.attention_rnn_dim, D = c.decoder_rnn_dim, nm = c.num_mels, r = c.frames_per_step;
  const int RM = nm * r, Q = 256, M = c.num_mixtures, P0 = c.prenet_dims[0], P1 = c.prenet_dims[1];
  const int nta = cdiv(B, 16), gy = nta >= 2 ? cdiv(nta, 2) : nta;
  int* flags = L.flags;
  MB_HIP(hipMemsetAsync(L.f_p0, 0, L.f_bytes, s));  // zero states, zero go frame -> p0 = 0, gate parts = 0 (DecoderPrenet is bias-free)
  MB_HIP(hipMemcpyAsync(flags + TF_SEED, &seed, sizeof(seed), hipMemcpyHostToDevice, s));
  DropK dk;
  dk.thresh = 0x80000000u; dk.scale = 2.f; dk.enabled = 1; dk.it_add = 0; dk.it_limit = max_steps;  // F.dropout(p = 0.5, training = True)   DecoderPrenet :18-21
  const size_t lds_mol = std::max(sizeof(float) * (((size_t)Q + 3 * M + 1 + 2 * T + 1 + 3) & ~(size_t)3) + 8 * 64 * 16,
                                  sizeof(float) * (size_t)(nta >= 2 ? FmRed<2, 1>::floats : FmRed<1, 1>::floats));
  MB_REQUIRE(lds_mol <= 160 * 1024, "ppg2mel_decode: memory too long for the attention window in LDS (T=%d)", T);
  if (lds_mol > 48 * 1024) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppg_mol_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mol));
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppg_mol_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mol));
  }
  auto step = [&](int it_off) -> int {
    const dim3 blk(512);
#define PF_LAUNCH(KERNEL, GX, ...)                                                            \
    do {                                                                                       \
      if (nta >= 2) hipLaunchKernelGGL((KERNEL<2>), dim3(GX, gy), blk, 0, s, __VA_ARGS__);     \
      else hipLaunchKernelGGL((KERNEL<1>), dim3(GX, gy), blk, 0, s, __VA_ARGS__);              \
    } while (0)
    PfFc1K f1;
    f1.w = p->pre_w[1].p; f1.p0 = L.f_p0; f1.p1 = L.f_p1; f1.nta = nta; f1.B = B; f1.it_off = it_off; f1.flags = flags;
    f1.drop = dk; f1.drop.layer = 1; f1.drop.ld = P1; f1.drop.it_stride = (long long)B * P1;
    f1.drop.mask = d_dropout ? d_dropout + (size_t)max_steps * B * P0 : nullptr;
    PF_LAUNCH(ppg_fc1_kernel, P1 / 16, f1);
    PfLstmK la;
    la.w = p->f_att_p.p; la.x0 = L.f_p1; la.x1 = L.f_p1; la.pre_a = reinterpret_cast<const float4*>(L.f_prec);
    la.pre_b = reinterpret_cast<const float4*>(L.f_preh); la.b4 = reinterpret_cast<const float4*>(p->f_att_b4.p);
    la.h = L.f_ah; la.c = L.f_ac; la.nta = nta; la.flags = flags;
    if (nta >= 2) hipLaunchKernelGGL((ppg_lstm_kernel<2, 1, 1>), dim3(A / 4, gy), blk, 0, s, la);
    else hipLaunchKernelGGL((ppg_lstm_kernel<1, 1, 1>), dim3(A / 4, gy), blk, 0, s, la);
    PfQ0K q0;
    q0.w = p->q0_w.p; q0.bias = p->q0_b.p; q0.att_h = L.f_ah; q0.q = L.f_q; q0.n_q = Q / 16; q0.nta = nta; q0.flags = flags;
    q0.hh.w = p->f_att_h.p; q0.hh.x = L.f_ah; q0.hh.out = reinterpret_cast<float4*>(L.f_preh); q0.hh.n_tiles = A / 4;
    PF_LAUNCH(ppg_q0_kernel, Q / 16 + A / 4, q0);
    PfMolK mk;
    mk.q = L.f_q; mk.w2 = p->q2_w.p; mk.b2 = p->q2_b.p; mk.memory = d_memory; mk.mu = L.mu; mk.ctx = L.f_ctx; mk.align_out = d_align;
    mk.T = T; mk.E = E; mk.Q = Q; mk.M = M; mk.nta = nta; mk.it_off = it_off; mk.max_steps = max_steps; mk.eps = 1e-5f; mk.flags = flags;
    PfPreK dh;
    dh.w = p->f_dec_h.p; dh.x = L.f_dh; dh.out = reinterpret_cast<float4*>(L.f_pred); dh.n_tiles = D / 4;
    if (nta >= 2) hipLaunchKernelGGL(ppg_mol_kernel<2>, dim3(B + (D / 4) * gy), blk, lds_mol, s, mk, dh, B, gy);
    else hipLaunchKernelGGL(ppg_mol_kernel<1>, dim3(B + (D / 4) * gy), blk, lds_mol, s, mk, dh, B, gy);
    PfLstmK ld;
    ld.w = p->f_dec_x.p; ld.x0 = L.f_ah; ld.x1 = L.f_ctx; ld.pre_a = reinterpret_cast<const float4*>(L.f_pred); ld.pre_b = nullptr;
    ld.b4 = reinterpret_cast<const float4*>(p->f_dec_b4.p); ld.h = L.f_dh; ld.c = L.f_dc; ld.nta = nta; ld.flags = flags;
    if (nta >= 2) hipLaunchKernelGGL((ppg_lstm_kernel<2, 6, 4>), dim3(D / 4, gy), blk, 0, s, ld);
    else hipLaunchKernelGGL((ppg_lstm_kernel<1, 6, 4>), dim3(D / 4, gy), blk, 0, s, ld);
    PfOutK ok;
    ok.w_out = p->out_w.p; ok.b_out = p->out_b.p; ok.w_fc0 = p->f_fc0_w.p; ok.b_fc0 = p->f_fc0_b.p; ok.h = L.f_dh; ok.ctx = L.f_ctx;
    ok.p0 = L.f_p0; ok.mel_out = d_mel; ok.stop_out = d_stop;
    ok.cpart.w = p->f_att_c.p; ok.cpart.x = L.f_ctx; ok.cpart.out = reinterpret_cast<float4*>(L.f_prec); ok.cpart.n_tiles = A / 4;
    ok.nta = nta; ok.B = B; ok.n_proj = RM / 16; ok.n_fc0 = P0 / 16; ok.RM = RM; ok.max_steps = max_steps; ok.min_steps = min_steps;
    ok.it_off = it_off; ok.thr = thr; ok.flags = flags;
    ok.drop = dk; ok.drop.layer = 0; ok.drop.it_add = 1; ok.drop.ld = P0; ok.drop.it_stride = (long long)B * P0; ok.drop.mask = d_dropout;
    PF_LAUNCH(ppg_out_kernel, RM / 16 + P0 / 16 + 1 + A / 4, ok);
#undef PF_LAUNCH
    MB_HIP(hipGetLastError());
    return MB_OK;
  };
  MB_HIP(hipEventRecord(p->ev_t0, s));
  int G = 16;
  if (const char* ge = getenv("MBHIP_GRAPH_STEPS")) G = std::max(1, atoi(ge));
  const bool use_graph = getenv("MBHIP_NO_GRAPH") == nullptr && max_steps >= G;
  int done_steps = 0, eager_steps = 0, rc = MB_OK;
  bool stopped = false;
  if (use_graph) {
    mb_ppg2mel::GraphKey key = {d_memory, d_dropout, d_mel, d_align, d_stop, d_workspace, B, T, max_steps, min_steps, G, thr};
    if (!p->graph_exec || !(key == p->gkey)) {
      p->drop_graph();
      MB_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      for (int i = 0; i < G && !rc; ++i) rc = step(i);
      hipLaunchKernelGGL(ppg_bump_kernel, dim3(1), dim3(1), 0, s, flags, G);
      hipError_t e = hipStreamEndCapture(s, &p->graph);
      if (rc) { p->drop_graph(); return rc; }
      if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture", __FILE__, __LINE__);
      e = hipGraphInstantiate(&p->graph_exec, p->graph, nullptr, nullptr, 0);
      if (e != hipSuccess) { p->drop_graph(); return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__); }
      p->gkey = key;
    }
    const int reps = max_steps / G;
    for (int rep = 0; rep < reps; ++rep) {
      MB_HIP(hipGraphLaunch(p->graph_exec, s));
      MB_HIP(hipMemcpyAsync(p->h_flags + 8 * (rep & 1), flags, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
      MB_HIP(hipEventRecord(p->ev_flags[rep & 1], s));
      done_steps += G;
      if (rep > 0) {
        MB_HIP(hipEventSynchronize(p->ev_flags[(rep - 1) & 1]));
        if (p->h_flags[8 * ((rep - 1) & 1) + TF_DONE]) { stopped = true; break; }
      }
    }
  }
  for (int st = done_steps; st < max_steps && !stopped; ++st) {
    if ((rc = step(st - done_steps))) return rc;
    ++eager_steps;
    if (((st - done_steps) & 15) == 15) {
      MB_HIP(hipMemcpyAsync(p->h_flags, flags, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
      MB_HIP(hipStreamSynchronize(s));
      if (p->h_flags[TF_DONE]) stopped = true;
    }
  }
  MB_HIP(hipEventRecord(p->ev_t1, s));
  MB_HIP(hipMemcpyAsync(p->h_flags, flags, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
  MB_HIP(hipStreamSynchronize(s));
  *steps_out = p->h_flags[TF_NFRAMES];
  p->last_steps = *steps_out; p->timed = true; p->last_chain_steps = done_steps + eager_steps;
  return MB_OK;
}

extern "C" int mb_ppg2mel_last_loop_ms(const mb_ppg2mel* p, float* ms, int* steps) {
  MB_REQUIRE(p && ms, "ppg2mel_last_loop_ms: null pointer");
  if (!p->timed) { set_error("ppg2mel_last_loop_ms: no decode on the fast loop yet"); return MB_ESTATE; }
  MB_HIP(hipEventElapsedTime(ms, p->ev_t0, p->ev_t1));
  if (steps) *steps = p->last_steps;
  return MB_OK;
}

extern "C" int mb_ppg2mel_last_loop_launches(const mb_ppg2mel* p, int* launches) {
  MB_REQUIRE(p && launches, "ppg2mel_last_loop_launches: null pointer");
  if (!p->timed) { set_error("ppg2mel_last_loop_launches: no decode on the fast loop yet"); return MB_ESTATE; }
  *launches = p->last_resident ? 1 : 6 * p->last_chain_steps;
  return MB_OK;
}

extern "C" int mb_ppg2mel_decode(const mb_ppg2mel* p, const float* d_memory, int batch, int t_enc, int max_steps,
                                 int min_steps, float stop_threshold, const float* d_dropout, uint64_t seed,
                                 float* d_mel, float* d_align, float* d_stop, int* h_n_steps, void* d_workspace,
                                 size_t workspace_bytes, mb_stream_t stream) {
  MB_REQUIRE(p && d_memory && d_mel && d_align && d_stop && h_n_steps, "ppg2mel_decode: null pointer");
  MB_REQUIRE(batch > 0 && t_enc > 0 && max_steps > 0, "ppg2mel_decode: empty input");
  MB_REQUIRE(batch <= 256, "ppg2mel_decode: batch %d > 256", batch);
  PpgLayout L;
  ppg_layout(p, batch, d_workspace, &L);
  if (!d_workspace || workspace_bytes < L.bytes) {
    set_error("ppg2mel_decode: workspace %zu B < required %zu B", workspace_bytes, L.bytes);
    return MB_ENOMEM;
  }
  const mb_ppg2mel_config& c = p->cfg;
  const int B = batch, T = t_enc, E = c.enc_dim, A = c.attention_rnn_dim, D = c.decoder_rnn_dim, nm = c.num_mels, r = c.frames_per_step;
  const int RM = nm * r, Q = 256, M = c.num_mixtures, NL = c.num_decoder_rnn_layer;
  const int P = c.prenet_dims[c.n_prenet - 1];
  hipStream_t s = (hipStream_t)stream;
  const size_t lds_mol = sizeof(float) * ((size_t)Q + 3 * M + 1 + (T + 1) + T + 8);
  MB_REQUIRE(lds_mol <= 64 * 1024, "ppg2mel_decode: memory too long for the attention window in LDS (T=%d)", T);
  // zero initial states (initialize_decoder_states :118-148, init_states :58-66, go frame :111-115)
  MB_HIP(hipMemsetAsync(L.att_h, 0, sizeof(float) * 2 * B * A, s)); MB_HIP(hipMemsetAsync(L.att_c, 0, sizeof(float) * 2 * B * A, s));
  for (int i = 0; i < NL; ++i) {
    MB_HIP(hipMemsetAsync(L.dec_h[i], 0, sizeof(float) * 2 * B * D, s)); MB_HIP(hipMemsetAsync(L.dec_c[i], 0, sizeof(float) * 2 * B * D, s));
  }
  MB_HIP(hipMemsetAsync(L.ctx, 0, sizeof(float) * B * E, s));
  MB_HIP(hipMemsetAsync(L.mu, 0, sizeof(float) * B * M, s));
  MB_HIP(hipMemsetAsync(L.zero, 0, sizeof(float) * B * L.ldy, s));
  MB_HIP(hipMemsetAsync(L.flags, 0, sizeof(int) * 8, s));
  MB_HIP(hipMemsetAsync(d_mel, 0, sizeof(float) * (size_t)B * max_steps * RM, s));
  MB_HIP(hipMemsetAsync(d_align, 0, sizeof(float) * (size_t)B * max_steps * T, s));
  MB_HIP(hipMemsetAsync(d_stop, 0, sizeof(float) * (size_t)B * max_steps, s));
  const char* fenv = getenv("MBHIP_PPG_FAST");
  if (p->fast && !(fenv && atoi(fenv) == 0)) {  // production dims: ppg_fast.h (MBHIP_PPG_FAST=0 keeps the general loop)
    mb_ppg2mel* pm = const_cast<mb_ppg2mel*>(p);
    MB_HIP(hipEventRecord(pm->ev_in, s));
    MB_HIP(hipStreamWaitEvent(pm->loop_stream, pm->ev_in, 0));
    // One utterance: the whole loop as ONE resident launch (ppg_resident.h; MBHIP_PPG_RESIDENT=0 keeps the 6-launch step,
    // =1 retries on a device where a resident launch lost a hand-off before).  Needs its 217 workgroups co-resident, one
    // per compute unit: checked against the device; a launch that still loses a hand-off drains itself (abort word = 1)
    // and the launch chain below redoes the utterance.
    const char* renv = getenv("MBHIP_PPG_RESIDENT");
    bool resident = pm->resident && B == 1 && T <= PR_T_MAX && !(renv && atoi(renv) == 0);
    int dev = 0;
    // 2..32 utterances: the whole loop as ONE resident launch on MFMA tiles (ppg_batch.h): 160 role workgroups + one per utterance,
    // co-resident; the same switch, the same fallbacks (lost hand-off -> abort word 1, operand range -> range word: the chain reruns the batch)
    pm->last_batch_fallback = 0;
    bool batch_res = pm->batch_resident && B >= 2 && B <= PB_BMAX && T <= PB_T_MAX && !(renv && atoi(renv) == 0);
    if (batch_res) {
      MB_HIP(hipGetDevice(&dev));
      if (pm->batch_cus < 0) {
        hipDeviceProp_t prop;
        MB_HIP(hipGetDeviceProperties(&prop, dev));
        MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppg_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PB_LDS_BYTES));
        int nb = 0;
        MB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(ppg_batch_kernel), 512, PB_LDS_BYTES));
        pm->batch_cus = nb >= 1 ? prop.multiProcessorCount : 0;
      }
      if (pm->batch_cus < PB_G_MOL + B) batch_res = false;
      if (dev >= 0 && dev < 64 && g_ppg_batch_failed[dev] && !renv) batch_res = false;
    }
    if (batch_res) {
      hipStream_t ls = pm->loop_stream;
      int* abort_word = reinterpret_cast<int*>(L.px + (size_t)PB_NG * 2 * PBX_PER);
      MB_HIP(hipMemsetAsync(L.px, 0, pb_exchange_bytes(), ls));
      const bool test_abort = diag_int("abort_pr") != 0;
      if (test_abort) MB_HIP(hipMemsetAsync(abort_word, 1, 1, ls));
      MB_HIP(hipMemcpyAsync(L.flags + TF_SEED, &seed, sizeof(seed), hipMemcpyHostToDevice, ls));
      const int P0 = c.prenet_dims[0], P1 = c.prenet_dims[1];
      PbK k;
      k.att_w1 = {reinterpret_cast<const uint4*>(pm->b_att_w1.p), pm->b_us[0]};
      k.att_wx = {reinterpret_cast<const uint4*>(pm->b_att_wx.p), pm->b_us[1]};
      k.att_wc = {reinterpret_cast<const uint4*>(pm->b_att_wc.p), pm->b_us[1]};
      k.att_wh = {reinterpret_cast<const uint4*>(pm->b_att_wh.p), pm->b_us[1]};
      k.q0_w = {reinterpret_cast<const uint4*>(pm->b_q0.p), pm->b_us[2]};
      k.dec_wa = {reinterpret_cast<const uint4*>(pm->b_dec_wa.p), pm->b_us[3]};
      k.dec_wc = {reinterpret_cast<const uint4*>(pm->b_dec_wc.p), pm->b_us[3]};
      k.dec_wh = {reinterpret_cast<const uint4*>(pm->b_dec_wh.p), pm->b_us[3]};
      k.out_wh = {reinterpret_cast<const uint4*>(pm->b_out_wh.p), pm->b_us[4]};
      k.out_wc = {reinterpret_cast<const uint4*>(pm->b_out_wc.p), pm->b_us[4]};
      k.att_b4 = reinterpret_cast<const float4*>(pm->f_att_b4.p); k.dec_b4 = reinterpret_cast<const float4*>(pm->f_dec_b4.p);
      k.q0_b = pm->q0_b.p; k.out_b = pm->out_b.p; k.fc0_b = pm->f_fc0_b.p; k.w2 = pm->q2_w.p; k.b2 = pm->q2_b.p;
      k.memory = d_memory; k.mel_out = d_mel; k.align_out = d_align; k.stop_out = d_stop;
      DropK dk;
      dk.thresh = 0x80000000u; dk.scale = 2.f; dk.enabled = 1; dk.it_add = 0; dk.it_limit = max_steps;  // F.dropout(p = 0.5, training = True)   DecoderPrenet :18-21
      k.drop1 = dk; k.drop1.layer = 1; k.drop1.ld = P1; k.drop1.it_stride = (long long)B * P1;
      k.drop1.mask = d_dropout ? d_dropout + (size_t)max_steps * B * P0 : nullptr;
      k.drop0 = dk; k.drop0.layer = 0; k.drop0.it_add = 1; k.drop0.ld = P0; k.drop0.it_stride = (long long)B * P0; k.drop0.mask = d_dropout;
      k.ex = L.px; k.abort_word = abort_word; k.range_word = abort_word + 1; k.flags = L.flags;
      k.B = B; k.T = T; k.M = M; k.RM = RM; k.S = max_steps; k.min_steps = min_steps; k.thr = stop_threshold; k.eps = 1e-5f;
      const int ng = (B <= PB_GC && !(B >= 2 && diag_int("pb_groups") == 2)) ? 1 : 2;  // (A/B: MBHIP_DIAG=pb_groups=2 splits a small batch too)
      k.gn0[0] = 0; k.gn0[1] = ng == 1 ? B : (B + 1) / 2; k.gn0[2] = B;
      MB_HIP(hipEventRecord(pm->ev_t0, ls));
      hipLaunchKernelGGL(ppg_batch_kernel, dim3(PB_G_MOL + B), dim3(512), PB_LDS_BYTES, ls, k);
      MB_HIP(hipGetLastError());
      MB_HIP(hipEventRecord(pm->ev_t1, ls));
      MB_HIP(hipMemcpyAsync(pm->h_abort, abort_word, 2 * sizeof(int), hipMemcpyDeviceToHost, ls));
      MB_HIP(hipMemcpyAsync(pm->h_flags, L.flags, sizeof(int) * 8, hipMemcpyDeviceToHost, ls));
      MB_HIP(hipEventRecord(pm->ev_res, ls));
      MB_HIP(hipEventSynchronize(pm->ev_res));
      if (pm->h_abort[0] != 1 && pm->h_abort[1] == 0) {
        *h_n_steps = pm->h_flags[TF_NFRAMES];
        pm->last_steps = *h_n_steps; pm->timed = true; pm->last_resident = true;
        return MB_OK;
      }
      if (pm->h_abort[0] == 1) {
        static bool warned_b = false;
        if (!warned_b) {
          fprintf(stderr, "[mbhip] ppg2mel: the batch resident kernel could not keep its workgroups co-resident; using the launch chain\n");
          warned_b = true;
        }
        if (dev >= 0 && dev < 64 && !test_abort) g_ppg_batch_failed[dev] = true;
        pm->last_batch_fallback = 1;
      } else {
        static bool warned_r = false;
        if (!warned_r) {
          fprintf(stderr, "[mbhip] ppg2mel: an activation left the operand-pair kernel's range (|x| > 65504 or NaN); using the fp32 launch chain\n");
          warned_r = true;
        }
        pm->last_batch_fallback = 2;
      }
      MB_HIP(hipMemsetAsync(L.mu, 0, sizeof(float) * B * M, ls));
      MB_HIP(hipMemsetAsync(L.flags, 0, sizeof(int) * 8, ls));
      MB_HIP(hipMemsetAsync(d_mel, 0, sizeof(float) * (size_t)B * max_steps * RM, ls));
      MB_HIP(hipMemsetAsync(d_align, 0, sizeof(float) * (size_t)B * max_steps * T, ls));
      MB_HIP(hipMemsetAsync(d_stop, 0, sizeof(float) * (size_t)B * max_steps, ls));
    }
    if (resident) {
      MB_HIP(hipGetDevice(&dev));
      if (pm->resident_cus < 0) {
        hipDeviceProp_t prop;
        MB_HIP(hipGetDeviceProperties(&prop, dev));
        MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ppg_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PR_LDS_BYTES));
        int nb = 0;
        MB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(ppg_resident_kernel), 512, PR_LDS_BYTES));
        pm->resident_cus = nb >= 1 ? prop.multiProcessorCount : 0;
      }
      if (pm->resident_cus < PR_WGS) resident = false;
      if (dev >= 0 && dev < 64 && g_ppg_resident_failed[dev] && !renv) resident = false;
    }
    if (resident) {
      hipStream_t ls = pm->loop_stream;
      int* abort_word = reinterpret_cast<int*>(L.px + (size_t)2 * PRX_PER);
      MB_HIP(hipMemsetAsync(L.px, 0, pr_exchange_bytes(), ls));
      const bool test_abort = diag_int("abort_pr") != 0;  // tests only (MBHIP_DIAG=abort_pr): the chain takes over
      if (test_abort) MB_HIP(hipMemsetAsync(abort_word, 1, 1, ls));
      std::string rtrace_file;  // diagnostics (MBHIP_DIAG=pr_trace=<file>): dump the wall-clock marks of the kernel to this file
      const char* rtrace = diag_str("pr_trace", &rtrace_file) ? rtrace_file.c_str() : nullptr;
      PrK k;
      k.img_att = pm->r_att.p; k.img_w1 = pm->r_w1.p; k.img_dec = pm->r_dec.p; k.img_q0 = pm->r_q0.p; k.img_out = pm->r_out.p;
      k.att_b4 = reinterpret_cast<const float4*>(pm->f_att_b4.p); k.dec_b4 = reinterpret_cast<const float4*>(pm->f_dec_b4.p);
      k.q0_b = pm->q0_b.p; k.out_b = pm->out_b.p; k.fc0_b = pm->f_fc0_b.p; k.w2 = pm->q2_w.p; k.b2 = pm->q2_b.p;
      k.memory = d_memory; k.mel_out = d_mel; k.align_out = d_align; k.stop_out = d_stop; k.drop_mask = d_dropout;
      k.ex = L.px; k.abort_word = abort_word; k.flags = L.flags; k.seed = seed;
      k.T = T; k.M = M; k.RM = RM; k.S = max_steps; k.min_steps = min_steps; k.thr = stop_threshold; k.eps = 1e-5f;
      k.variant = diag_int("pr_variant", 0);
      k.trace = rtrace ? reinterpret_cast<unsigned long long*>(abort_word) + 32 : nullptr;
      MB_HIP(hipEventRecord(pm->ev_t0, ls));
      hipLaunchKernelGGL(ppg_resident_kernel, dim3(PR_WGS), dim3(512), PR_LDS_BYTES, ls, k);
      MB_HIP(hipGetLastError());
      MB_HIP(hipEventRecord(pm->ev_t1, ls));
      MB_HIP(hipMemcpyAsync(pm->h_abort, abort_word, sizeof(int), hipMemcpyDeviceToHost, ls));
      MB_HIP(hipMemcpyAsync(pm->h_flags, L.flags, sizeof(int) * 8, hipMemcpyDeviceToHost, ls));
      MB_HIP(hipEventRecord(pm->ev_res, ls));
      MB_HIP(hipEventSynchronize(pm->ev_res));
      if (*pm->h_abort != 1) {
        if (rtrace) {
          unsigned long long marks[1000];
          MB_HIP(hipMemcpy(marks, k.trace, sizeof(marks), hipMemcpyDeviceToHost));
          if (FILE* f = fopen(rtrace, "wb")) { fwrite(marks, sizeof(marks), 1, f); fclose(f); }
        }
        *h_n_steps = pm->h_flags[TF_NFRAMES];
        pm->last_steps = *h_n_steps; pm->timed = true; pm->last_resident = true;
        return MB_OK;
      }
      static bool warned = false;
      if (!warned) {
        fprintf(stderr, "[mbhip] ppg2mel: resident kernel could not keep its workgroups co-resident; using the launch chain\n");
        warned = true;
      }
      if (dev >= 0 && dev < 64 && !test_abort) g_ppg_resident_failed[dev] = true;
      // whatever the drained launch left behind: outputs and flags back to zero, then the chain from step 0
      MB_HIP(hipMemsetAsync(L.mu, 0, sizeof(float) * B * M, ls));
      MB_HIP(hipMemsetAsync(L.flags, 0, sizeof(int) * 8, ls));
      MB_HIP(hipMemsetAsync(d_mel, 0, sizeof(float) * (size_t)B * max_steps * RM, ls));
      MB_HIP(hipMemsetAsync(d_align, 0, sizeof(float) * (size_t)B * max_steps * T, ls));
      MB_HIP(hipMemsetAsync(d_stop, 0, sizeof(float) * (size_t)B * max_steps, ls));
    }
    pm->last_resident = false;
    const int rcf = ppg_fast_loop_body(pm, L, d_memory, B, T, max_steps, min_steps, stop_threshold, d_dropout, seed, d_mel, d_align,
                                       d_stop, d_workspace, pm->loop_stream, h_n_steps);
    if (rcf) (void)hipStreamSynchronize(pm->loop_stream);
    return rcf;
  }
  int* done = L.flags;
  int* n_steps = L.flags + 1;
  // dropout masks: layer l at d_dropout + sum_{k<l} max_steps*B*dims[k], step-major inside
  size_t mask_base[4] = {0, 0, 0, 0};
  for (int i = 1; i < c.n_prenet; ++i) mask_base[i] = mask_base[i - 1] + (size_t)max_steps * B * c.prenet_dims[i - 1];

  int steps = 0;
  for (int st = 0; st < max_steps; ++st) {
    const int pp = st & 1;
    float* ah_p = L.att_h + (size_t)pp * B * A; float* ah_n = L.att_h + (size_t)(pp ^ 1) * B * A;
    float* ac_p = L.att_c + (size_t)pp * B * A; float* ac_n = L.att_c + (size_t)(pp ^ 1) * B * A;
    RnnK k;
    int rc;
    // decoder_input = prenet(last frame of the previous step | go frame)   :283, :307
    const float* xin = st == 0 ? L.zero : L.y + (size_t)(r - 1) * nm;
    int kin = nm;
    for (int i = 0; i < c.n_prenet; ++i) {
      memset(&k, 0, sizeof(k));
      k.w = p->pre_w[i].p; k.nseg = 1; k.nkb_total = kin / 16;
      k.seg[0] = {i == 0 ? xin : L.pbuf[i - 1], i == 0 ? L.ldy : c.prenet_dims[i - 1], kin / 16, 0};
      k.N = B; k.units = c.prenet_dims[i]; k.biasX = p->zero_bias.p; k.y = L.pbuf[i]; k.ldy = c.prenet_dims[i]; k.act = 1;
      k.mask_scale = 2.f;
      k.mask = d_dropout ? d_dropout + mask_base[i] + (size_t)st * B * c.prenet_dims[i] : nullptr;
      k.drop_on = d_dropout ? 0 : 1; k.drop_thresh = 0x80000000u; k.drop_seed = seed; k.drop_iter = st; k.drop_layer = i; k.skip_flag = done;
      if ((rc = rnn_launch(EPI_LINEAR, k, s))) return rc;
      kin = c.prenet_dims[i];
    }
    // attention_hidden, attention_cell = attention_rnn([decoder_input, attention_context], ...)   :188-190
    memset(&k, 0, sizeof(k));
    k.w = p->att_w.p; k.nseg = 3; k.nkb_total = (P + E + A) / 16;
    k.seg[0] = {L.pbuf[c.n_prenet - 1], P, P / 16, 0}; k.seg[1] = {L.ctx, E, E / 16, 0}; k.seg[2] = {ah_p, A, A / 16, 1};
    k.N = B; k.units = A; k.biasX = p->att_bih.p; k.biasH = p->att_bhh.p; k.c_prev = ac_p; k.h_out = ah_n; k.c_out = ac_n; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LSTM, k, s))) return rc;
    // MOLAttention: q = relu(query_layer.0(att_h)), then the attention kernel   mol_attention.py:75-122
    memset(&k, 0, sizeof(k));
    k.w = p->q0_w.p; k.nseg = 1; k.nkb_total = A / 16; k.seg[0] = {ah_n, A, A / 16, 0};
    k.N = B; k.units = Q; k.biasX = p->q0_b.p; k.y = L.q; k.ldy = Q; k.act = 1; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LINEAR, k, s))) return rc;
    MolK mk;
    mk.q = L.q; mk.w2 = p->q2_w.p; mk.b2 = p->q2_b.p; mk.memory = d_memory; mk.mu = L.mu; mk.context = L.ctx;
    mk.align_out = d_align; mk.skip_flag = done; mk.T = T; mk.E = E; mk.Q = Q; mk.M = M; mk.step = st; mk.max_steps = max_steps;
    mk.eps = 1e-5f;
    hipLaunchKernelGGL(mol_attention_kernel, dim3(B), dim3(256), lds_mol, s, mk);
    MB_HIP(hipGetLastError());
    // decoder LSTM stack   :200-209
    const float* dprev = nullptr;
    for (int i = 0; i < NL; ++i) {
      float* hp_ = L.dec_h[i] + (size_t)pp * B * D; float* hn_ = L.dec_h[i] + (size_t)(pp ^ 1) * B * D;
      float* cp_ = L.dec_c[i] + (size_t)pp * B * D; float* cn_ = L.dec_c[i] + (size_t)(pp ^ 1) * B * D;
      memset(&k, 0, sizeof(k));
      k.w = p->dec_w[i].p;
      if (i == 0) {
        k.nseg = 3; k.nkb_total = (A + E + D) / 16;
        k.seg[0] = {ah_n, A, A / 16, 0}; k.seg[1] = {L.ctx, E, E / 16, 0}; k.seg[2] = {hp_, D, D / 16, 1};
      } else {
        k.nseg = 2; k.nkb_total = 2 * D / 16;
        k.seg[0] = {dprev, D, D / 16, 0}; k.seg[1] = {hp_, D, D / 16, 1};
      }
      k.N = B; k.units = D; k.biasX = p->dec_bih[i].p; k.biasH = p->dec_bhh[i].p; k.c_prev = cp_; k.h_out = hn_; k.c_out = cn_; k.skip_flag = done;
      if ((rc = rnn_launch(EPI_LSTM, k, s))) return rc;
      dprev = hn_;
    }
    // mel_output = linear_projection([h, context]); stop_output = stop_layer(...)   :281-288
    memset(&k, 0, sizeof(k));
    k.w = p->out_w.p;
    if (c.concat_context_to_last) {
      k.nseg = 2; k.nkb_total = (D + E) / 16; k.seg[0] = {dprev, D, D / 16, 0}; k.seg[1] = {L.ctx, E, E / 16, 0};
    } else {
      k.nseg = 1; k.nkb_total = D / 16; k.seg[0] = {dprev, D, D / 16, 0};
    }
    k.N = B; k.units = RM + 1; k.biasX = p->out_b.p; k.y = L.y; k.ldy = L.ldy; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LINEAR, k, s))) return rc;
    PpgFinK fk;
    fk.y = L.y; fk.mel_out = d_mel; fk.stop_out = d_stop; fk.done = done; fk.n_steps = n_steps;
    fk.B = B; fk.ldy = L.ldy; fk.RM = RM; fk.step = st; fk.max_steps = max_steps; fk.min_steps = min_steps; fk.thr = stop_threshold;
    hipLaunchKernelGGL(ppg_finalize_kernel, dim3(1), dim3(256), 0, s, fk);
    MB_HIP(hipGetLastError());
    if ((st & 15) == 15 || st == max_steps - 1) {  // poll the stop flag (skipped launches in between are no-ops)
      int hf[2] = {0, 0};
      MB_HIP(hipMemcpyAsync(hf, L.flags, sizeof(hf), hipMemcpyDeviceToHost, s));
      MB_HIP(hipStreamSynchronize(s));
      steps = hf[1];
      if (hf[0]) break;
    }
  }
  *h_n_steps = steps;
  return MB_OK;
}
