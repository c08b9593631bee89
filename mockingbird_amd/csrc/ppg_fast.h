// ppg2mel decoder step, production shape (prenet 256/128, attention LSTM 512, decoder LSTM 512 x 1 layer, memory 256,
// 5 mixtures, r = 2, context concatenated to the projection input): six launches per step on fragment-major
// activations (fm_gemm.h), hipGraph-replayed.  Reference: models/ppg2mel/rnn_decoder_mol.py:267-316 (loop),
// :187-209 (attend / decode), DecoderPrenet :10-22; utils/mol_attention.py:67-122.
//
// The first-generation step (ppg2mel.hip: 8 eager launches of rnn_rowtile_body, 46 us at batch 1) was bound by the
// host's launch rate and, per launch, by uncoalesced operand loads.  Here
//   1  fc1           p1 = dropout(relu(prenet.1 . p0))                                   (p0 left by launch 6 of the previous step)
//   2  attention LSTMCell on its PRENET columns only (K = 128): the context and hidden parts of its gates depend on the
//      previous step alone and arrive as CM4 quads from launches 6 / 3 of the previous step
//   3  q = relu(query_layer.0 . att_h)   beside   W_hh_att . att_h  for the next step
//   4  MoL attention + context (one workgroup per utterance)   beside   W_hh_dec . h_dec for THIS step's decoder LSTM
//   5  decoder LSTMCell on [att_h | context] (K = 768), hidden part from launch 4
//   6  projection -> mel frames   beside   prenet.0 folded through the projection's last frame (exact algebra, the
//      prenet is bias-free: W0 . (Wp x + bp)), dropout of step + 1   beside   the stop logit + batch-wide stop rule
//      beside   W_ih_att[:, context] . context for the next step
// Step index, seed and flags live in device memory (TF_* words), so one captured graph serves every step.
#pragma once
#include "fm_gemm.h"

namespace mb {

// gate functions on the hardware exp2 / reciprocal (1 ulp), as ppg_resident.h / gru_scan.h / taco_fast.h use them
__device__ __forceinline__ float pf_sigmoid(const float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float pf_tanh(const float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }


__device__ __forceinline__ float pf_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // F.softplus(beta=1, threshold=20)

// LSTM-tile-order product -> CM4 gate quads (a hidden / context part computed one launch or one step ahead)
struct PfPreK { const float* w; const float* x; float4* out; int n_tiles; };
template <int NT, int PW>
__device__ __forceinline__ void pf_pre_job(const PfPreK& a, const int mt, const int nt0, const int nta, const int done, float* red) {
  float sx[4], sh[4];
  if (!fm_gemm<NT, PW, PW, 4, 1>(a.w, mt, a.x, a.x, nta, nt0, red, sx, sh)) return;
  const int lane = threadIdx.x & 63, nt = nt0 + (threadIdx.x >> 6);
  if (nt >= nta || done) return;
  a.out[((size_t)mt * nta + nt) * 64 + lane] = make_float4(sx[0], sx[1], sx[2], sx[3]);
}

// ---------------------------------------------------------------------------------------------- 1: prenet layer 1
struct PfFc1K { const float* w; const float* p0; float* p1; int nta, B, it_off; const int* flags; DropK drop; };
template <int NT>
__global__ __launch_bounds__(512) void ppg_fc1_kernel(PfFc1K a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int mt = blockIdx.x, nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE], it = a.flags[TF_ITER] + a.it_off;
  float sx[4], sh[4];
  if (!fm_gemm<NT, 2, 2, 4, 1>(a.w, mt, a.p0, a.p0, a.nta, nt0, red, sx, sh)) return;
  const int lane = threadIdx.x & 63, nt = nt0 + (threadIdx.x >> 6), du = lane >> 4, n = nt * 16 + (lane & 15);
  if (nt >= a.nta || done) return;
  float v[4] = {sx[0], sx[1], sx[2], sx[3]};  // bias-free linears (DecoderPrenet :14-16)
  relu_drop_quad(a.drop, a.flags, it, n < a.B ? n : a.B - 1, mt * 16 + du * 4, v);
  reinterpret_cast<float4*>(a.p1)[((size_t)mt * a.nta + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------------- 2 / 5: LSTMCells
// gates = W_x . x (on the chain) + pre_a + pre_b (CM4, computed ahead) + (b_ih + b_hh);  torch gate order (i, f, g, o)
struct PfLstmK {
  const float* w; const float* x0; const float* x1;  // W_x tiles; K segments (FM)
  const float4* pre_a; const float4* pre_b;            // pre_b may be null
  const float4* b4; float* h; float* c;                // h FM (written), c CM1 (in place)
  int nta; const int* flags;
};
template <int NT, int PW, int PS>
__global__ __launch_bounds__(512) void ppg_lstm_kernel(PfLstmK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int mt = blockIdx.x, nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  const int ntE = (nt0 + (wv < NT ? wv : 0) < a.nta) ? nt0 + (wv < NT ? wv : 0) : a.nta - 1;
  const size_t cm = ((size_t)mt * a.nta + ntE) * 64 + lane;
  const float4 bq = a.b4[mt * 4 + du];
  const float4 pa = a.pre_a[cm];
  const float4 pb = a.pre_b ? a.pre_b[cm] : make_float4(0.f, 0.f, 0.f, 0.f);
  float* cp = a.c + cm;
  const float cprev = *cp;
  float sx[4], sh[4];
  if (!fm_gemm<NT, PW, PS, 4, 1>(a.w, mt, a.x0, a.x1, a.nta, nt0, red, sx, sh)) return;
  if (nt0 + wv >= a.nta || done) return;
  const float gi = pf_sigmoid(((sx[0] + pa.x) + pb.x) + bq.x);
  const float gf = pf_sigmoid(((sx[1] + pa.y) + pb.y) + bq.y);
  const float gg = pf_tanh(((sx[2] + pa.z) + pb.z) + bq.z);
  const float go = pf_sigmoid(((sx[3] + pa.w) + pb.w) + bq.w);
  const float cy = gf * cprev + gi * gg;
  *cp = cy;
  a.h[((size_t)(mt >> 2) * a.nta + ntE) * 256 + (mt & 3) * 64 + i * 4 + du] = go * pf_tanh(cy);
}

// ---------------------------------------------------------------------------------------------- 3: query layer 0 (+ att hh)
struct PfQ0K { const float* w; const float* bias; const float* att_h; float* q; int n_q, nta; const int* flags; PfPreK hh; };
template <int NT>
__global__ __launch_bounds__(512) void ppg_q0_kernel(PfQ0K a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int nt0 = blockIdx.y * NT, done = a.flags[TF_DONE];
  if ((int)blockIdx.x >= a.n_q) { pf_pre_job<NT, 4>(a.hh, blockIdx.x - a.n_q, nt0, a.nta, done, red); return; }
  const int mt = blockIdx.x;
  float sx[4], sh[4];
  if (!fm_gemm<NT, 4, 4, 4, 1>(a.w, mt, a.att_h, a.att_h, a.nta, nt0, red, sx, sh)) return;
  const int lane = threadIdx.x & 63, nt = nt0 + (threadIdx.x >> 6), du = lane >> 4;
  if (nt >= a.nta || done) return;
  const float4 bq = *reinterpret_cast<const float4*>(a.bias + mt * 16 + du * 4);
  reinterpret_cast<float4*>(a.q)[((size_t)mt * a.nta + nt) * 64 + lane] =
      make_float4(fmaxf(sx[0] + bq.x, 0.f), fmaxf(sx[1] + bq.y, 0.f), fmaxf(sx[2] + bq.z, 0.f), fmaxf(sx[3] + bq.w, 0.f));
}

// ---------------------------------------------------------------------------------------------- 4: MoL attention (+ dec hh)
// One workgroup (8 waves) per utterance: mixture parameters = query_layer.2 . q (wave reductions), the discretised
// mixture-of-logistics window over T_enc (mol_attention.py:92-109), context = alpha . memory with the memory rows split
// over the waves (16-byte loads, 8 rows in flight per wave).  E = 256 (one float4 per lane).
struct PfMolK {
  const float* q;       // FM [Q]
  const float* w2; const float* b2;  // [3M][Q], [3M]
  const float* memory;  // [B][T][E]
  float* mu;            // [B][M] in/out
  float* ctx;           // FM [E]
  float* align_out;     // [B][max_steps][T]
  int T, E, Q, M, nta, it_off, max_steps; float eps;
  const int* flags;
};
__device__ __forceinline__ void pf_mol_body(const PfMolK& a, const int b, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int done = a.flags[TF_DONE];  // (no early exit: the stop flag only predicates the stores, nothing waits for it)
  const int step = a.flags[TF_ITER] + a.it_off;
  float* s_q = sm;                   // [Q]
  float* s_mp = s_q + a.Q;           // [3M] raw mixture parameters, then (w, sigma, mu)
  float* s_af = s_mp + 3 * a.M + 1;  // [T + 1]
  float* s_al = s_af + a.T + 1;      // [T]
  float4* s_part = reinterpret_cast<float4*>(sm + ((a.Q + 3 * a.M + 1 + 2 * a.T + 1 + 3) & ~3));  // [8][64]
  // Everything that does not depend on the query is requested first: this wave's memory rows (wave w owns rows w, w + 8,
  // ...: up to 32 of them = T_enc <= 256 in registers, the rest in a tail loop) and its two rows of query_layer.2.
  // They arrive while the mixture parameters are worked out (first generation: 4 dependent load->FMA rounds, 6 us).
  const float* mem = a.memory + (size_t)b * a.T * a.E + lane * 4;
  constexpr int NR = 32;
  float4 mv[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int t = wave + 8 * j;
    mv[j] = t < a.T ? *reinterpret_cast<const float4*>(mem + (size_t)t * a.E) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float w2v[2][4];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int o = wave + 8 * rr;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int k = lane + 64 * cc;
      w2v[rr][cc] = (o < 3 * a.M && k < a.Q) ? a.w2[(size_t)o * a.Q + k] : 0.f;
    }
  }
  const float b2a = (wave < 3 * a.M) ? a.b2[wave] : 0.f, b2b = (wave + 8 < 3 * a.M) ? a.b2[wave + 8] : 0.f;
  const float mu_prev = (tid < a.M) ? a.mu[(size_t)b * a.M + tid] : 0.f;
  for (int k = tid; k < a.Q; k += 512) s_q[k] = a.q[fm_index(a.nta, b, k)];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {  // mixture_params = query_layer.2(q)   :75   (3M <= 16 outputs, Q <= 256)
    const int o = wave + 8 * rr;
    float acc = 0.f;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) { const int k = lane + 64 * cc; if (k < a.Q) acc += w2v[rr][cc] * s_q[k]; }
    acc = wave_sum(acc);
    if (lane == 0 && o < 3 * a.M) s_mp[o] = acc + (rr ? b2b : b2a);
  }
  __syncthreads();
  if (wave == 0) {  // w = softmax(w_hat) + eps; sigma = softplus(sigma_hat) + eps; mu = mu_prev + softplus(Delta_hat)  :92-96
    const bool live = lane < a.M;  // lane m owns mixture m: the M chains of transcendentals run side by side
    const float wh = live ? s_mp[lane] : -INFINITY, sh_ = live ? s_mp[a.M + lane] : 0.f, dh = live ? s_mp[2 * a.M + lane] : 0.f;
    const float mx = wave_max(wh);
    const float ew = live ? expf(wh - mx) : 0.f;
    float se = 0.f;
    for (int m = 0; m < a.M; ++m) se += __shfl(ew, m, 64);  // ascending m, as the sequential sum
    if (live) {
      const float w = ew / se + a.eps;
      const float sg = pf_softplus(sh_) + a.eps;
      const float mu = mu_prev + pf_softplus(dh);
      if (!done) a.mu[(size_t)b * a.M + lane] = mu;
      s_mp[lane] = w; s_mp[a.M + lane] = sg; s_mp[2 * a.M + lane] = mu;
    }
  }
  __syncthreads();
  for (int j = tid; j <= a.T; j += 512) {  // alpha_full[j] = sum_m w_m / (1 + sigmoid((mu_m - (j + 0.5)) / sigma_m))   :101-107
    float s = 0.f;
    const float pos = (float)j + 0.5f;
    for (int m = 0; m < a.M; ++m) {
      const float z = (s_mp[2 * a.M + m] - pos) / s_mp[a.M + m];
      s += s_mp[m] * (1.f / (1.f + 1.f / (1.f + expf(-z))));
    }
    s_af[j] = s;
  }
  __syncthreads();
  float* al = a.align_out + ((size_t)b * a.max_steps + step) * a.T;
  for (int t = tid; t < a.T; t += 512) {  // alpha_t = diff; zeros -> eps   :108-109
    float v = s_af[t + 1] - s_af[t];
    if (v == 0.f) v = a.eps;
    s_al[t] = v;
    if (!done) al[t] = v;
  }
  __syncthreads();
  // context = alpha . memory   :115   (E = 256: lane l owns columns 4 l .. 4 l + 3)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int t = wave + 8 * j;
    const float sc = t < a.T ? s_al[t] : 0.f;
    acc.x += sc * mv[j].x; acc.y += sc * mv[j].y; acc.z += sc * mv[j].z; acc.w += sc * mv[j].w;
  }
  for (int t0 = wave + 8 * NR; t0 < a.T; t0 += 64) {  // T_enc > 256: the rest, 8 rows in flight per wave
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = t0 + 8 * j;
      v[j] = t < a.T ? *reinterpret_cast<const float4*>(mem + (size_t)t * a.E) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = t0 + 8 * j;
      const float sc = t < a.T ? s_al[t] : 0.f;
      acc.x += sc * v[j].x; acc.y += sc * v[j].y; acc.z += sc * v[j].z; acc.w += sc * v[j].w;
    }
  }
  s_part[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && !done) {
    float4 r = s_part[lane];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) { const float4 o = s_part[w8 * 64 + lane]; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
    const int p = lane * 4;  // FM float4 slot of columns [p, p + 4) of utterance b
    reinterpret_cast<float4*>(a.ctx)[((size_t)(p >> 4) * a.nta + (b >> 4)) * 64 + ((p >> 2) & 3) * 16 + (b & 15)] = r;
  }
}
template <int NT>
__global__ __launch_bounds__(512) void ppg_mol_kernel(PfMolK a, PfPreK hh, int B, int gy) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // max(attention window, FmRed<NT,1>)
  const int id = blockIdx.x;
  if (id < B) { pf_mol_body(a, id, sm); return; }
  const int j = id - B, mt = j / gy;
  pf_pre_job<NT, 4>(hh, mt, (j - mt * gy) * NT, a.nta, a.flags[TF_DONE], sm);
}

// ---------------------------------------------------------------------------------------------- 6: projection (+ fc0' + stop + att ctx part)
struct PfOutK {
  const float* w_out; const float* b_out;  // projection rows then the stop row: tile n_proj holds the stop row as its row 0
  const float* w_fc0; const float* b_fc0;  // prenet.0 folded through the projection's last frame
  const float* h; const float* ctx;        // FM [D], FM [E]
  float* p0; float* mel_out; float* stop_out;
  PfPreK cpart;                            // W_ih_att[:, context columns] . context for the next step
  int nta, B, n_proj, n_fc0, RM, max_steps, min_steps, it_off; float thr; int* flags; DropK drop;
};
template <int NT>
__global__ __launch_bounds__(512) void ppg_out_kernel(PfOutK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE], it = a.flags[TF_ITER] + a.it_off;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  const int bx = blockIdx.x;
  float sx[4], sh[4];
  if (bx < a.n_proj) {  // mel_output = linear_projection([h, context])   :281-287
    if (!fm_gemm<NT, 6, 4, 4, 1>(a.w_out, bx, a.h, a.ctx, a.nta, nt0, red, sx, sh)) return;
    const int nt = nt0 + wv, n = nt * 16 + i, row0 = bx * 16 + du * 4;
    if (nt >= a.nta || n >= a.B || done) return;
    const float4 bq = *reinterpret_cast<const float4*>(a.b_out + row0);
    *reinterpret_cast<float4*>(a.mel_out + ((size_t)n * a.max_steps + it) * a.RM + row0) =
        make_float4(sx[0] + bq.x, sx[1] + bq.y, sx[2] + bq.z, sx[3] + bq.w);
    return;
  }
  if (bx < a.n_proj + a.n_fc0) {  // next step's prenet layer 0 from the same operands
    const int mt = bx - a.n_proj;
    if (!fm_gemm<NT, 6, 4, 4, 1>(a.w_fc0, mt, a.h, a.ctx, a.nta, nt0, red, sx, sh)) return;
    const int nt = nt0 + wv, n = nt * 16 + i, row0 = mt * 16 + du * 4;
    if (nt >= a.nta || done) return;
    const float4 bq = *reinterpret_cast<const float4*>(a.b_fc0 + row0);
    float v[4] = {sx[0] + bq.x, sx[1] + bq.y, sx[2] + bq.z, sx[3] + bq.w};
    relu_drop_quad(a.drop, a.flags, it, n < a.B ? n : a.B - 1, row0, v);
    reinterpret_cast<float4*>(a.p0)[((size_t)mt * a.nta + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    return;
  }
  if (bx > a.n_proj + a.n_fc0) {  // attention LSTM, context part of the next step's gates
    pf_pre_job<NT, 2>(a.cpart, bx - (a.n_proj + a.n_fc0 + 1), nt0, a.nta, done, red);
    return;
  }
  // stop_output = stop_layer([h, context]) (:288) + batch-wide stop rule (:301-305, :349-354)
  if (!fm_gemm<NT, 6, 4, 4, 1>(a.w_out, a.n_proj, a.h, a.ctx, a.nta, nt0, red, sx, sh)) return;
  const int nt = nt0 + wv, n = nt * 16 + i;
  if (nt >= a.nta || done) return;
  int below = 0;
  if (du == 0 && n < a.B) {
    const float lg = sx[0] + a.b_out[a.RM];
    a.stop_out[(size_t)n * a.max_steps + it] = lg;
    below = !(1.f / (1.f + expf(-lg)) > a.thr);
  }
  const unsigned long long vote = __ballot(below);
  if (lane == 0) {
    if (vote) {
      int old = atomicAdd(a.flags + TF_NOTREADY, __popcll(vote));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");
    }
    if (atomicAdd(a.flags + TF_ARRIVE, 1) == a.nta - 1) {  // last column tile decides for the batch
      const int not_ready = __hip_atomic_load(a.flags + TF_NOTREADY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.flags[TF_NFRAMES] = it + 1;  // steps produced
      if (not_ready == 0 && it + 1 >= a.min_steps) a.flags[TF_DONE] = 1;
      a.flags[TF_ARRIVE] = 0; a.flags[TF_NOTREADY] = 0;
    }
  }
}

__global__ void ppg_bump_kernel(int* flags, int n) { flags[TF_ITER] += n; }

}  // namespace mb
