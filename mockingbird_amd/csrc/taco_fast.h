// Tacotron decoder iteration, production shape (decoder 128, memory 1024, LSTM 1024): the seven launches of
// one iteration (tacotron.py:71-138) on activations kept in MFMA-fragment order.
//
// What the first-generation loop (rnn_rowtile_body, 9 launches per iteration) paid for, per
// profiles/r01_tacotron_generate_kernel_stats.csv and r01_pmc_tacotron.json:
//   * every B-operand load was a 16-byte piece of a [column][K] row: 16 half-used cache lines per wave
//     instruction, the activations re-read by every row tile through that path (2 x the weight bytes for the
//     batch-32 LSTMs: 33.5 MB in 15.3 us = 2.2 TB/s, 47 % of the wave cycles stalled on issue);
//   * cell outputs were written back as 4-byte stores 4 KB apart;
//   * two launches (prenet fc1, finalize) and the context / hidden-state parts of the attention GRU sat on the
//     dependent chain although their inputs are known a launch (or an iteration) earlier.
// Here:
//   * FM ("fragment-major") activations: element (column n, feature k) of a K-vector set lives at
//       ((k/16 * NTA + n/16) * 64 + (k/4 % 4) * 16 + n % 16) * 4 + k % 4          (NTA = column tiles)
//     so the B fragment of a (k-block, column tile) is ONE contiguous 1 KB read -- the same shape as the packed A
//     fragments -- and the LINEAR / GRU / LSTM epilogues, whose lanes own (row quad | unit, column), store
//     contiguous 1 KB / 256 B pieces of the next launch's operand.
//   * CM4 / CM1 ("cell-major") per-(unit, column) quads / scalars at (unit/4 * NTA + n/16) * 64 + lane: what one
//     epilogue lane writes is what the consuming epilogue lane reads (gate pre-activations, LSTM cell state).
//   * every weight and activation fragment of a wave (K/128 k-blocks) is requested before the first MFMA; the two
//     column tiles of a workgroup alternate on the matrix pipe, so consecutive MFMAs are independent.
//   * chain: fc2 | attention GRU on its prenet columns only | LSA + context | rnn_input beside the NEXT iteration's
//     GRU context/hidden pre-activations | LSTM1 | LSTM2 | mel_proj beside the next iteration's prenet fc1 (folded
//     through mel_proj: fc1 . mel_proj[last frame], exact algebra) and the stop token + batch-wide stop rule.
//     7 launches, captured in a hipGraph (iteration index, seed and flags live in device memory).
//   * round 5 (tacotron.hip taco_front_kernel, fm_gemm.h fm_gemm16): launches 1..3 as ROLES of one launch with tagged-granule
//     hand-offs, rnn_input folded into the attention role through a memory projected once per call (texts <= 128 symbols), the K >= 1024
//     tile products of what remains -- LSTM 1 | LSTM 2 | mel_proj + fc1' + stop, and the hidden-half riders -- on the fp16 matrix pipe
//     with split operands: 4 launches per iteration (taco_pick_form chooses among 4 / 5 / 7); the kernels below are the jobs of all forms.
// Reference arithmetic: models/synthesizer/models/tacotron.py:71-138, sublayer/pre_net.py:21-26, torch GRUCell /
// LSTMCell gate orders as in rnn_body.h.
#pragma once
#include "fm_gemm.h"
#include "granule.h"

namespace mb {

// Gate functions of the production loop on the hardware exp2 / reciprocal (1 ulp each; gru_scan.h's forms).  Round 3 measured them
// (48.6 -> 47.5 us per iteration) and dropped them together with the WaveRNN ones because the MOL chain / resident pair stopped being
// bit-identical; the Tacotron loops have no bit-identical partner (fast vs general loop: 2e-4), so the production loop takes them.
__device__ __forceinline__ float tf_sigmoid(const float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tf_tanh(const float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }


// ------------------------------------------------------------------------------------------------ fc2
// p2 = dropout(relu(fc2 . p1 + b2))   (pre_net.py:24-26).  K = 2D = 256 -> PW = 2.
struct TfFcK {
  const float* w; const float* bias; const float* xin; float* yout;
  int nta, B, it_off; const int* flags; DropK drop; unsigned long long* trace;
};
template <int NT>
__global__ __launch_bounds__(512) void taco_fc2_kernel(TfFcK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int mt = blockIdx.x, nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE], it = a.flags[TF_ITER] + a.it_off;
  float sx[4], sh[4];
  const bool pick = blockIdx.x == 3 && blockIdx.y == 0;
  tf_mark(a.trace, TS_FC2, 0, pick);
  if (!fm_gemm<NT, 2, 2, 4, 1>(a.w, mt, a.xin, a.xin, a.nta, nt0, red, sx, sh, a.trace, TS_FC2, pick)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = nt0 + wave, du = lane >> 4, n = nt * 16 + (lane & 15), row0 = mt * 16 + du * 4;
  if (nt >= a.nta || done) return;
  const float4 bq = *reinterpret_cast<const float4*>(a.bias + row0);
  float v[4] = {sx[0] + bq.x, sx[1] + bq.y, sx[2] + bq.z, sx[3] + bq.w};
  relu_drop_quad(a.drop, a.flags, it, n < a.B ? n : a.B - 1, row0, v);
  reinterpret_cast<float4*>(a.yout)[((size_t)mt * a.nta + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
  tf_mark_end(a.trace, TS_FC2, 4, pick);
}

// ------------------------------------------------------------------------------------------------ attention GRU
// attn_hidden = GRUCell([context, prenet_out], attn_hidden)  (tacotron.py:97-98) on its prenet columns only:
// W_ih[:, :P] . context + b_ih and W_hh . attn_hidden + b_hh were left as CM4 quads by the previous rnn_input launch.
struct TfGruK {
  const float* w; const float* xin; const float4* xpre; const float4* hpre; float* ah;  // ah FM [D], updated in place
  int nta, B; const int* flags; unsigned long long* trace;
  // folded form (taco_front_kernel without an rnn_input launch): the role multiplies W_hh . attn_hidden itself, in front of its wait
  // (the k-blocks P / 16 .. of w_pre's tiles, rnn_input job 1's products in their order), and attn_hidden ping-pongs between two FM
  // buffers (the other tiles' epilogues would otherwise write into the fragments this one is still reading)
  const float* w_pre = nullptr; const float4* bhh4 = nullptr; float* ah_out = nullptr;
};
template <int NT>
__global__ __launch_bounds__(512) void taco_gru_kernel(TfGruK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int mt = blockIdx.x, nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntE = (nt0 + (wv < NT ? wv : 0) < a.nta) ? nt0 + (wv < NT ? wv : 0) : a.nta - 1;
  const size_t cm = ((size_t)mt * a.nta + ntE) * 64 + lane;
  const float4 xp = a.xpre[cm], hp = a.hpre[cm];  // requested before the fragments are waited for
  const int du = lane >> 4, i = lane & 15;
  float* hpt = a.ah + ((size_t)(mt >> 2) * a.nta + ntE) * 256 + (mt & 3) * 64 + i * 4 + du;
  const float hprev = *hpt;
  float sx[4], sh[4];
  const bool pick = blockIdx.x == 3 && blockIdx.y == 0;
  tf_mark(a.trace, TS_GRU, 0, pick);
  if (!fm_gemm<NT, 2, 2, 3, 1>(a.w, mt, a.xin, a.xin, a.nta, nt0, red, sx, sh, a.trace, TS_GRU, pick)) return;
  if (nt0 + wv >= a.nta || done) return;
  // torch GRUCell, gate order (r, z, n)
  const float rg = tf_sigmoid((sx[0] + xp.x) + hp.x);
  const float zg = tf_sigmoid((sx[1] + xp.y) + hp.y);
  const float ng = tf_tanh((sx[2] + xp.z) + rg * hp.z);
  *hpt = ng + zg * (hprev - ng);
  tf_mark_end(a.trace, TS_GRU, 4, pick);
}

// ------------------------------------------------------------------------------------------------ LSTM hidden halves
// hh job: hpre = W_hh . h (LSTM tile order: the 16 rows of tile mt are the 4 gates of units 4 mt .. 4 mt + 3), one
// CM4 quad per (unit, column).  Rides as extra workgroups behind the jobs of a latency-bound launch.
// w16: the split fp16 image for the F16 form (fm_gemm16).  (No pointer behind `unscale`: a {float, pointer} tail of this struct inside a
// kernel argument is loaded as one vector, parked in a private array and promoted to LDS -- and a kernel with a promoted array reads
// its workgroup size from the dispatch packet in host memory at the start of EVERY workgroup: 80 ns per workgroup, 20 us per launch.)
struct TfHhK { const float* w; const float* h; float4* hpre; int n_tiles, tile0; const uint4* w16 = nullptr; float unscale = 1.f; };  // row tiles [tile0, tile0 + n_tiles)
// (F16 is a compile-time choice: the fp32 kernels stay the code they were)
template <int NT, bool F16 = false>
__device__ __forceinline__ void fm_hh_job(const TfHhK& a, const int mt_local, const int nt0, const int nta, const int done, float* red, int* lost = nullptr) {
  const int mt = a.tile0 + mt_local;
  float sx[4], sh[4];
  if constexpr (F16) {
    if (!fm_gemm16<NT, 4, 4, 1>(a.w16, mt, a.h, 64, a.h, 0, nta, nt0, red, a.unscale, sx, sh)) return;
  } else {
    if (!fm_gemm<NT, 8, 8, 4, 1>(a.w, mt, a.h, a.h, nta, nt0, red, sx, sh)) return;
  }
  const int lane = threadIdx.x & 63, nt = nt0 + (threadIdx.x >> 6);
  if (nt >= nta || done) return;
  if (F16) fm_range_check(sx, lost);
  a.hpre[((size_t)mt * nta + nt) * 64 + lane] = make_float4(sx[0], sx[1], sx[2], sx[3]);
}
// Two row tiles per workgroup against ONE set of activation fragments (taco_front_kernel: every workgroup of that launch has a compute
// unit to itself, so the tile COUNT is what the launch's tail is made of; the 128 KB of h fragments are read once per pair).  Per tile
// the products, their order and the reduction are fm_gemm's: same bits as two fm_hh_job calls.  red: 2 * FmRed<NT, 1>::floats.
template <int NT>
__device__ __forceinline__ void fm_hh_pair_job(const TfHhK& a, const int pair, const int nta, const int done, float* red) {
  constexpr int PW = 8, BLK = 4 * 4 * 16, NKB = 8 * PW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4, u = i >> 2, tau = i & 3;
  const int mtl0 = 2 * pair, mtl1 = (2 * pair + 1 < a.n_tiles) ? 2 * pair + 1 : 2 * pair;  // odd count: the last pair repeats its tile
  const float* wl0 = a.w + (size_t)(a.tile0 + mtl0) * NKB * BLK + ((u * 4 + tau) * 4 + kq) * 4;
  const float* wl1 = a.w + (size_t)(a.tile0 + mtl1) * NKB * BLK + ((u * 4 + tau) * 4 + kq) * 4;
  int ntc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ntc[nt] = nt < nta ? nt : nta - 1;
  float4 w0[PW], w1[PW], b[PW][NT];
  const float4* sp = reinterpret_cast<const float4*>(a.h);
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int kb = wave + 8 * p;
    w0[p] = *reinterpret_cast<const float4*>(wl0 + (size_t)kb * BLK);
    w1[p] = *reinterpret_cast<const float4*>(wl1 + (size_t)kb * BLK);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[p][nt] = sp[((size_t)kb * nta + ntc[nt]) * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc0[NT], acc1[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { acc0[nt] = {0.f, 0.f, 0.f, 0.f}; acc1[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int p = 0; p < PW; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float a0 = c == 0 ? w0[p].x : c == 1 ? w0[p].y : c == 2 ? w0[p].z : w0[p].w;
      const float a1 = c == 0 ? w1[p].x : c == 1 ? w1[p].y : c == 2 ? w1[p].z : w1[p].w;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = c == 0 ? b[p][nt].x : c == 1 ? b[p][nt].y : c == 2 ? b[p][nt].z : b[p][nt].w;
        acc0[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc0[nt], 0, 0, 0);
        acc1[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc1[nt], 0, 0, 0);
      }
    }
  float4* red4 = reinterpret_cast<float4*>(red);  // [2 tiles][8][NT][64]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    red4[(wave * NT + nt) * 64 + lane] = make_float4(acc0[nt][0], acc0[nt][1], acc0[nt][2], acc0[nt][3]);
    red4[((8 + wave) * NT + nt) * 64 + lane] = make_float4(acc1[nt][0], acc1[nt][1], acc1[nt][2], acc1[nt][3]);
  }
  __syncthreads();
  if (wave >= 2 * NT) return;
  const int ts = wave / NT, nt = wave - ts * NT;
  float sx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const float4 v = red4[((ts * 8 + w8) * NT + nt) * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
  }
  if (nt >= nta || done || (ts == 1 && mtl1 == mtl0)) return;
  a.hpre[((size_t)(a.tile0 + (ts ? mtl1 : mtl0)) * nta + nt) * 64 + lane] = make_float4(sx[0], sx[1], sx[2], sx[3]);
}
// stand-alone form (general LSA kernel in use, or diagnostics)
template <int NT>
__global__ __launch_bounds__(512) void taco_hh_kernel(TfHhK a, int nta, const int* flags) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  fm_hh_job<NT>(a, blockIdx.x, blockIdx.y * NT, nta, flags[TF_DONE], red);
}

// ------------------------------------------------------------------------------------------------ rnn_input (+ GRU pre)
// job 0 (blockIdx.x < n_rin): x = rnn_input([context, attn_hidden])  (tacotron.py:108-109) -> FM
// job 1: the NEXT iteration's attention-GRU pre-activations from the same operands:
//        xpre = W_ih[:, :P] . context + b_ih,  hpre = W_hh . attn_hidden + b_hh   (CM4: r, z, n, -)
// job 2 (one tile, one live row): the context half of the stop token's logit, stop_proj[:, H:] . context
//        (tacotron.py:133-135); the mel launch adds the x half once the LSTMs are through
struct TfRinK {
  const float* w_rin; const float* b_rin; const float* w_pre; const float4* bih4; const float4* bhh4;
  const float* w_stopc; float* stop_part;  // job 2: stop_proj's context columns . context -> [nta*16] partial logits
  TfHhK hh;  // job 3: second half of the row tiles of W_hh1 . h1 (the first half rode in the previous mel launch)
  const float* ctx; const float* ah; float* x; float4* xpre; float4* hpre;
  int nta, n_rin; const int* flags; unsigned long long* trace;
  // split fp16 images of the three chain jobs (fm_gemm16; K = P + D padded to 1280) -- null: the fp32 pipe
  const uint4* rin16 = nullptr; const uint4* pre16 = nullptr; const uint4* stopc16 = nullptr;
  float us_rin = 1.f, us_pre = 1.f, us_stopc = 1.f; int* lost = nullptr;
};
template <int NT, bool F16 = false>
__global__ __launch_bounds__(512) void taco_rin_kernel(TfRinK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 2>::floats];
  const int nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4;
  float sx[4], sh[4];
  if ((int)blockIdx.x < a.n_rin) {
    const int mt = blockIdx.x;
    const bool pick = blockIdx.x == 3 && blockIdx.y == 0;
    tf_mark(a.trace, TS_RIN, 0, pick);
    if constexpr (F16) {
      if (!fm_gemm16<NT, 5, 4, 1>(a.rin16, mt, a.ctx, 64, a.ah, 8, a.nta, nt0, red, a.us_rin, sx, sh, a.trace, TS_RIN, pick)) return;
    } else {
      if (!fm_gemm<NT, 9, 8, 4, 1>(a.w_rin, mt, a.ctx, a.ah, a.nta, nt0, red, sx, sh, a.trace, TS_RIN, pick)) return;
    }
    const int nt = nt0 + wv;
    if (nt >= a.nta || done) return;
    if (F16) fm_range_check(sx, a.lost);
    const float4 bq = *reinterpret_cast<const float4*>(a.b_rin + mt * 16 + du * 4);
    reinterpret_cast<float4*>(a.x)[((size_t)mt * a.nta + nt) * 64 + lane] =
        make_float4(sx[0] + bq.x, sx[1] + bq.y, sx[2] + bq.z, sx[3] + bq.w);
    tf_mark_end(a.trace, TS_RIN, 4, pick);
    return;
  }
  const int mt = blockIdx.x - a.n_rin;
  if (mt > 32) {
    fm_hh_job<NT, F16>(a.hh, mt - 33, nt0, a.nta, done, red, a.lost);
    if (a.trace && threadIdx.x == 0) atomicMax(a.trace + TS_RIN * 16 + 13, (unsigned long long)wall_clock64());  // last rider
    return;
  }
  if (mt == 32) {  // stop token, context half
    if constexpr (F16) {
      if (!fm_gemm16<NT, 5, 4, 1>(a.stopc16, 0, a.ctx, 64, a.ah, 8, a.nta, nt0, red, a.us_stopc, sx, sh)) return;
    } else {
      if (!fm_gemm<NT, 9, 8, 4, 1>(a.w_stopc, 0, a.ctx, a.ah, a.nta, nt0, red, sx, sh)) return;
    }
    if (F16) fm_range_check(sx, a.lost);  // (every fm_gemm16 result is checked where it is made: no reliance on a sibling tile)
    const int nt = nt0 + wv;
    if (nt >= a.nta || done || du != 0) return;
    a.stop_part[nt * 16 + (lane & 15)] = sx[0];
    if (a.trace && lane == 0) atomicMax(a.trace + TS_RIN * 16 + 11, (unsigned long long)wall_clock64());  // stop tile
    return;
  }
  if constexpr (F16) {
    if (!fm_gemm16<NT, 5, 4, 2>(a.pre16, mt, a.ctx, 64, a.ah, 8, a.nta, nt0, red, a.us_pre, sx, sh)) return;
  } else {
    if (!fm_gemm<NT, 9, 8, 3, 2>(a.w_pre, mt, a.ctx, a.ah, a.nta, nt0, red, sx, sh)) return;
  }
  const int nt = nt0 + wv;
  if (nt >= a.nta || done) return;
  if (F16) { fm_range_check(sx, a.lost); fm_range_check(sh, a.lost); }
  const float4 bi = a.bih4[mt * 4 + du], bh = a.bhh4[mt * 4 + du];
  const size_t cm = ((size_t)mt * a.nta + nt) * 64 + lane;
  a.xpre[cm] = make_float4(sx[0] + bi.x, sx[1] + bi.y, sx[2] + bi.z, 0.f);
  a.hpre[cm] = make_float4(sh[0] + bh.x, sh[1] + bh.y, sh[2] + bh.z, 0.f);
  if (a.trace && lane == 0) atomicMax(a.trace + TS_RIN * 16 + 12, (unsigned long long)wall_clock64());  // last GRU-pre tile
}

// ------------------------------------------------------------------------------------------------ residual LSTM
// h, c = LSTMCell(x, (h, c)); x = x + h   (tacotron.py:112-125, eval branch).  Only the input half W_ih . x (64
// k-blocks) is multiplied on the dependent chain: the hidden half W_hh . h depends on the PREVIOUS iteration's state
// alone and was left as CM4 gate quads by an hh job (fm_hh_job) riding in an earlier, latency-bound launch.
struct TfLstmK {
  const float* w; const float4* b4;  // W_ih tiles; b_ih + b_hh per unit: (i, f, g, o)
  const float4* hpre;                // W_hh . h_prev (CM4)
  const float* x; float* h_out; float* c; float* x_out;  // FM, FM, CM1 (in place), FM
  int nta; const int* flags; unsigned long long* trace; int trace_slot;
  const uint4* w16 = nullptr; float unscale = 1.f; int* lost = nullptr;  // F16: W_ih as a split fp16 image (fm_gemm16)
};
template <int NT, bool F16 = false>
__global__ __launch_bounds__(512) void taco_lstm_kernel(TfLstmK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int mt = blockIdx.x, nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  const int ntE = (nt0 + (wv < NT ? wv : 0) < a.nta) ? nt0 + (wv < NT ? wv : 0) : a.nta - 1;
  const float4 bq = a.b4[mt * 4 + du];
  const float4 hq = a.hpre[((size_t)mt * a.nta + ntE) * 64 + lane];
  float* cp = a.c + ((size_t)mt * a.nta + ntE) * 64 + lane;
  const float cprev = *cp;
  const size_t fo = ((size_t)(mt >> 2) * a.nta + ntE) * 256 + (mt & 3) * 64 + i * 4 + du;
  const float xr = a.x[fo];
  float sx[4], sh[4];
  const bool pick = blockIdx.x == 100 && blockIdx.y == 0;
  tf_mark(a.trace, a.trace_slot, 0, pick);
  if (F16) {
    if (!fm_gemm16<NT, 4, 4, 1>(a.w16, mt, a.x, 64, a.x, 0, a.nta, nt0, red, a.unscale, sx, sh, a.trace, a.trace_slot, pick)) return;
  } else if (!fm_gemm<NT, 8, 8, 4, 1>(a.w, mt, a.x, a.x, a.nta, nt0, red, sx, sh, a.trace, a.trace_slot, pick)) return;
  if (nt0 + wv >= a.nta || done) return;
  if (F16) fm_range_check(sx, a.lost);
  // torch LSTMCell, gate order (i, f, g, o)
  const float gi = tf_sigmoid((sx[0] + hq.x) + bq.x);
  const float gf = tf_sigmoid((sx[1] + hq.y) + bq.y);
  const float gg = tf_tanh((sx[2] + hq.z) + bq.z);
  const float go = tf_sigmoid((sx[3] + hq.w) + bq.w);
  const float cy = gf * cprev + gi * gg;
  const float hy = go * tf_tanh(cy);
  *cp = cy;
  a.h_out[fo] = hy;
  a.x_out[fo] = xr + hy;
  tf_mark_end(a.trace, a.trace_slot, 4, pick);
}

// (Both LSTM cells in ONE launch with a fence + counter hand-off was measured in round 3 and lost 2x -- 100.7 against 48.7 us per
//  iteration, profiles/r03_taco_lstm_merged_ab.txt: an L2 write-back / invalidate per wave costs far more than the launch boundary it
//  replaces.  The kernel is no longer part of the library; the source is kept under tools/rejected/taco_lstm2_kernel.h.)

// ------------------------------------------------------------------------------------------------ mel_proj (+ fc1' + stop)
// job 0 (n_mel tiles): mels = mel_proj(x)[:, :, :r] (tacotron.py:128-129: only the r live frames' rows are kept,
//        frame-major) scattered straight into mel_out[b][m][t0 + j]
// job 1 (16 tiles):    the NEXT iteration's prenet layer 1: relu(fc1 . mel[last frame] + b1) with fc1 folded through
//        mel_proj (W' = fc1 . mel_proj[last frame rows]), dropout of iteration it + 1
// job 2 (1 tile):      stop = sigmoid(stop_proj([x, context])) (:133-136) -- x half here, context half from the rnn_input
//        launch -- and the batch-wide stop rule (:275)
// job 3 (hh.n_tiles row tiles): W_hh1 . h1 for the next iteration (fm_hh_job)
// job 4 (hh2.n_tiles row tiles): W_hh2 . h2 for the next iteration, the tiles taco_front_kernel has no room for
struct TfMelK {
  const float* w_mel; const float* w_fc1; const float* b_fc1; const float* w_stop; const float* b_stop;
  const float* x2; const float* stop_part; float* p1; float* mel_out; float* stop_out;
  TfHhK hh;  // job 3: hidden half of the NEXT iteration's first LSTM (h1 of this iteration is final)
  TfHhK hh2; // job 4 (fused front only): the first row tiles of the next iteration's W_hh2 . h2 (h2 is final as well)
  int nta, B, n_mel, M, r, max_steps, it_off; float min_stop_token; int* flags; DropK drop; unsigned long long* trace;
  // split fp16 images of the three chain jobs (fm_gemm16, K = H) -- null: the fp32 pipe
  const uint4* mel16 = nullptr; const uint4* fc116 = nullptr; const uint4* stop16 = nullptr;
  float us_mel = 1.f, us_fc1 = 1.f, us_stop = 1.f;
};
template <int NT, bool F16 = false>
__global__ __launch_bounds__(512) void taco_mel_kernel(TfMelK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int nt0 = blockIdx.y * NT;
  const int done = a.flags[TF_DONE], it = a.flags[TF_ITER] + a.it_off;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  const int bx = blockIdx.x;
  float sx[4], sh[4];
  if (bx < a.n_mel) {
    const bool pick = bx == 3 && blockIdx.y == 0;
    tf_mark(a.trace, TS_MEL, 0, pick);
    if constexpr (F16) {
      if (!fm_gemm16<NT, 4, 4, 1>(a.mel16, bx, a.x2, 64, a.x2, 0, a.nta, nt0, red, a.us_mel, sx, sh, a.trace, TS_MEL, pick)) return;
    } else {
      if (!fm_gemm<NT, 8, 8, 4, 1>(a.w_mel, bx, a.x2, a.x2, a.nta, nt0, red, sx, sh, a.trace, TS_MEL, pick)) return;
    }
    const int nt = nt0 + wv, n = nt * 16 + i, t0 = it * a.r;
    if (nt >= a.nta || n >= a.B || done) return;
    if (F16) fm_range_check(sx, a.flags + TF_LOST);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = bx * 16 + du * 4 + q;
      if (row < a.r * a.M) {
        const int j = row / a.M, m = row - j * a.M;
        if (t0 + j < a.max_steps) a.mel_out[((size_t)n * a.M + m) * a.max_steps + t0 + j] = sx[q];
      }
    }
    tf_mark_end(a.trace, TS_MEL, 4, pick);
    return;
  }
  if (bx < a.n_mel + 16) {
    const int mt = bx - a.n_mel;
    const bool pick = mt == 3 && blockIdx.y == 0;
    tf_mark(a.trace, TS_MEL_FC1, 0, pick);
    if constexpr (F16) {
      if (!fm_gemm16<NT, 4, 4, 1>(a.fc116, mt, a.x2, 64, a.x2, 0, a.nta, nt0, red, a.us_fc1, sx, sh, a.trace, TS_MEL_FC1, pick)) return;
    } else {
      if (!fm_gemm<NT, 8, 8, 4, 1>(a.w_fc1, mt, a.x2, a.x2, a.nta, nt0, red, sx, sh, a.trace, TS_MEL_FC1, pick)) return;
    }
    if (F16) fm_range_check(sx, a.flags + TF_LOST);
    const int nt = nt0 + wv, n = nt * 16 + i, row0 = mt * 16 + du * 4;
    if (nt >= a.nta || done) return;
    const float4 bq = *reinterpret_cast<const float4*>(a.b_fc1 + row0);
    float v[4] = {sx[0] + bq.x, sx[1] + bq.y, sx[2] + bq.z, sx[3] + bq.w};
    relu_drop_quad(a.drop, a.flags, it, n < a.B ? n : a.B - 1, row0, v);
    reinterpret_cast<float4*>(a.p1)[((size_t)mt * a.nta + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    tf_mark_end(a.trace, TS_MEL_FC1, 4, pick);
    return;
  }
  if (bx > a.n_mel + 16) {
    const int j = bx - (a.n_mel + 17);
    if (j < a.hh.n_tiles) fm_hh_job<NT, F16>(a.hh, j, nt0, a.nta, done, red, a.flags + TF_LOST);
    else fm_hh_job<NT, F16>(a.hh2, j - a.hh.n_tiles, nt0, a.nta, done, red, a.flags + TF_LOST);
    if (a.trace && threadIdx.x == 0) atomicMax(a.trace + TS_MEL * 16 + 13, (unsigned long long)wall_clock64());  // last rider
    return;
  }
  // stop tile: one live row (row 0) over K = x2; the context half of the logit comes from the rnn_input launch
  const bool pickS = blockIdx.y == 0;
  tf_mark(a.trace, TS_MEL_STOP, 0, pickS);
  const int ntS = (nt0 + (wv < NT ? wv : 0) < a.nta) ? nt0 + (wv < NT ? wv : 0) : a.nta - 1;
  const float spart = a.stop_part[ntS * 16 + i];
  if constexpr (F16) {
    if (!fm_gemm16<NT, 4, 4, 1>(a.stop16, 0, a.x2, 64, a.x2, 0, a.nta, nt0, red, a.us_stop, sx, sh, a.trace, TS_MEL_STOP, pickS)) return;
  } else {
    if (!fm_gemm<NT, 8, 8, 4, 1>(a.w_stop, 0, a.x2, a.x2, a.nta, nt0, red, sx, sh, a.trace, TS_MEL_STOP, pickS)) return;
  }
  if (F16) fm_range_check(sx, a.flags + TF_LOST);
  const int nt = nt0 + wv, n = nt * 16 + i;
  if (nt >= a.nta || done) return;
  int below = 0;
  if (du == 0 && n < a.B) {
    const float sgm = 1.0f / (1.0f + expf(-((sx[0] + spart) + a.b_stop[0])));
    a.stop_out[n] = sgm;
    below = !(sgm * 10.f > a.min_stop_token);  // (stop * 10 > min_stop_token).all()
  }
  const unsigned long long vote = __ballot(below);
  if (lane == 0) {
    if (vote) {  // utterances not ready to stop: performed before this tile's arrival ticket
      int old = atomicAdd(a.flags + TF_NOTREADY, __popcll(vote));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");
    }
    if (atomicAdd(a.flags + TF_ARRIVE, 1) == a.nta - 1) {  // last column tile to arrive decides for the batch
      const int not_ready = __hip_atomic_load(a.flags + TF_NOTREADY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int t0 = it * a.r;
      a.flags[TF_NFRAMES] = min(t0 + a.r, a.max_steps);
      if (not_ready == 0 && t0 > 10) a.flags[TF_DONE] = 1;  // ... and t > 10
      a.flags[TF_ARRIVE] = 0; a.flags[TF_NOTREADY] = 0;       // visible to the next iteration (kernel boundary)
    }
  }
  tf_mark_end(a.trace, TS_MEL_STOP, 4, pickS);
}

// p1 of iteration 0: the <GO> frame is all zeros (tacotron.py:261), so fc1's output is relu(b1), then dropout(it = 0)
struct TfP1K { const float* b_fc1; float* p1; int nta, B, rows; const int* flags; DropK drop; };
__global__ __launch_bounds__(64) void taco_p1_init_kernel(TfP1K a) {
  const int mt = blockIdx.x, nt = blockIdx.y, lane = threadIdx.x, du = lane >> 4, n = nt * 16 + (lane & 15), row0 = mt * 16 + du * 4;
  const float4 bq = *reinterpret_cast<const float4*>(a.b_fc1 + row0);
  float v[4] = {bq.x, bq.y, bq.z, bq.w};
  relu_drop_quad(a.drop, a.flags, a.flags[TF_ITER], n < a.B ? n : a.B - 1, row0, v);
  reinterpret_cast<float4*>(a.p1)[((size_t)mt * a.nta + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void taco_bump_kernel(int* flags, int n) { flags[TF_ITER] += n; }

}  // namespace mb
