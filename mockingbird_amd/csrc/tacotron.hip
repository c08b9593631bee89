// Tacotron (SV2TTS) decoder loop + CBHG postnet.
//
// Reference: models/synthesizer/models/tacotron.py
//   Decoder.forward :71-138 (one iteration), loop + stop rule :264-275,
//   postnet + post_proj :281-283; sublayer/lsa.py:21-42; sublayer/pre_net.py:12-27;
//   sublayer/cbhg.py:40-78; common/batch_norm_conv.py:11-14; common/highway_network.py:12-17.
//
// Per decoder iteration (r mel frames), 9 launches on one stream:
//   prenet fc1, fc2 (LINEAR + always-on dropout)        rnn.hip
//   attention GRUCell over [context | prenet]            rnn.hip (EPI_GRU)
//   location-sensitive attention + context               lsa_kernel (one workgroup / utterance,
//                                                        cumulative window, energies, softmax in LDS)
//   rnn_input Linear, 2 x residual LSTMCell              rnn.hip (EPI_LINEAR / EPI_LSTM)
//   mel_proj (only the r*80 of 1600 rows the reference keeps, tacotron.py:128-129)
//   finalize: stop token, stop rule on device, frame scatter
// The batch-wide stop rule (:275) is evaluated on the device; once it fires every later launch is
// a no-op (skip flag) and the host learns the frame count at its next poll.
// CBHG runs on the MFMA conv kernel (conv1d.hip) with ReLU->BatchNorm, max-pool and highway
// gates fused; the bidirectional GRU is a scan of EPI_GRU launches over precomputed W_ih.x.
#include "taco_fast.h"
#include "gru_scan.h"

namespace mb {

// ---------------------------------------------------------------- LSA + context
struct LsaK {
  const float* query;     // attn_h' [B][D]
  const float* mem_proj;  // [B][T][D]
  const float* memory;    // [B][T][P]
  const int* chars;       // [B][T]
  const float* cum_in;    // [B][T] cumulative attention (read)
  float* cum_out;         // [B][T] cumulative attention (written; ping-pong, no intra-launch race)
  const float* conv_w;    // [Fl][Kl]
  const float* conv_b;    // [Fl]
  const float* Lw;        // [D][Fl]
  const float* Ww;        // [D][D]
  const float* Wb;        // [D]
  const float* vw;        // [D]
  float* context;         // [B][P] out
  float* attn_out;        // [B][n_iter_max][T] (row `iter`)
  int T, D, P, Fl, Kl, iter, n_iter_max, psplit;
  const int* skip_flag;
  // fast path (lsa_fast_kernel): conv and L folded into one D x Kl tap matrix, W transposed
  const float* Mt;   // [Kl][D]  M = L . conv_w   (location features -> processed location in one 31-tap conv)
  const float* c0;   // [D]      L . conv_b
  const float* Wt;   // [D(k)][D(d)] W^T
  // the same operands as 16-byte rows per thread (one dwordx4 load where the first generation issued four dword loads:
  // the attention launch was bound by the vector-memory INSTRUCTION count, 118 per lane, profiles/r02_taco_trace_v1.json):
  const float4* Wq4;   // [4 quarters][8][D]: W[d][32 q + 4 k4 .. +3]
  const float4* Mf4;   // [8 d-tiles][2][64 lanes]: MFMA B fragments of the folded tap matrix, M[16 dt + i][4 g + kq], g = 4 gh + c
  const float4* mpf4;  // [B][T tiles][8 d-tiles][64 lanes]: mem_proj[b][16 tile + 4 rq + 0..3][16 dt + i] (D-fragment order),
                       // built per decode call (lsa_pack_memproj_kernel)
  // fast decoder loop (taco_fast.h): query / context are FM buffers with fm_nta column tiles (0: plain [B][D] / [B][P]),
  // and the iteration index is iter + *iter_base (device word, bumped once per graph replay)
  int fm_nta; const int* iter_base;
  unsigned long long* trace;  // diagnostics (taco_fast.h tf_mark)
  // fused front (taco_front_kernel): the query arrives inside the launch as {value, iteration + 1} granules [B][D] written by the
  // attention-GRU workgroups; `lost` = the word a timed-out wait raises (flags + TF_LOST)
  const unsigned long long* q_gran = nullptr; int* lost = nullptr;
  unsigned long long* e_gran = nullptr;  // [B][128] energies exchanged between the four workgroups of an utterance (T <= 128)
  int dma_early = 0;  // fused launch: waves 2..7 queue the memory rows' LDS-DMA in front of the query wait (waves 0 / 1 poll)
  // folded form (taco_front_kernel, no rnn_input launch): `memory` holds the PROJECTED memory rows [B][T][mem_ld] = W . memory_t with
  // W = [rnn_input's context columns (H rows) | attention GRU's W_ih context columns as (r, z, n, 0) quads per unit (4 D rows) | stop_proj's
  // context columns (1 row)], so that sum_t score_t row_t IS rnn_input's context part / the next GRU pre-activation / the stop logit's
  // context part (the context vector itself is no longer formed).  `context` then receives x = rnn_input([context, attn_hidden]).
  int mem_ld = 0;              // floats per memory row (0: P)
  int fold = 0;
  const float4* rin_a4 = nullptr;  // [psplit][D][64]: rnn_input's attn_hidden columns, W[p0 + 4 lane .. +3][P + k]
  const float* rin_b = nullptr;    // [H]
  const float4* bih4 = nullptr;    // [D] (b_r, b_z, b_n, 0) of the attention GRU's W_ih
  float4* xpre_out = nullptr;      // CM4 [D / 4][nta][64]: next iteration's W_ih[:, :P] . context + b_ih
  float* stop_part = nullptr;      // [nta * 16]: stop_proj's context half
};
__device__ __forceinline__ size_t lsa_qidx(const LsaK& a, int b, int k) { return a.fm_nta ? fm_index(a.fm_nta, b, k) : (size_t)b * a.D + k; }
// float4 slot of context columns [p, p+4) of utterance b (p % 4 == 0)
__device__ __forceinline__ float4* lsa_ctx4(const LsaK& a, int b, int p) {
  return a.fm_nta ? reinterpret_cast<float4*>(a.context) + ((size_t)(p >> 4) * a.fm_nta + (b >> 4)) * 64 + ((p >> 2) & 3) * 16 + (b & 15)
                  : reinterpret_cast<float4*>(a.context + (size_t)b * a.P + p);
}

// One workgroup (8 waves) per (utterance, quarter of the context columns).  The attention
// window -- cumulative alignment (zero padded), location features, energies, scores -- lives in
// LDS; the scores are recomputed by each of the `psplit` column groups (cheap) so the T x P
// context product, the only part that streams real data (T*P*4 B per utterance from L2),
// is spread over psplit CUs with 16 B/lane loads, 8 rows in flight per wave.
// dynamic LDS: cum[T+Kl-1] | pq[D] | LwT[Fl][D] | cw[Fl*Kl] | loc[T][Fl] | u[T] | red[64] | part[8][64*4]
__global__ __launch_bounds__(512) void lsa_kernel(LsaK a) {
  if (a.skip_flag && *a.skip_flag) return;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, pg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = a.T, D = a.D, Fl = a.Fl, Kl = a.Kl, half = (Kl - 1) / 2;
  float* cum = sm;  // zero padded by `half` on both sides
  float* pq = cum + ((T + Kl - 1 + 3) & ~3);
  float* LwT = pq + D;
  float* cw = LwT + (size_t)Fl * D;
  float* loc = cw + ((Fl * Kl + 3) & ~3);
  float* u = loc + (size_t)T * Fl;
  float* red = u + ((T + 3) & ~3);
  float* part = red + 64;
  const float* cg = a.cum_in + (size_t)b * T;
  for (int i = tid; i < T + Kl - 1; i += 512) {
    const int t = i - half;
    cum[i] = (t >= 0 && t < T) ? cg[t] : 0.f;
  }
  for (int i = tid; i < Fl * D; i += 512) { const int d = i / Fl, f = i - d * Fl; LwT[f * D + d] = a.Lw[i]; }
  for (int i = tid; i < Fl * Kl; i += 512) cw[i] = a.conv_w[i];
  // processed_query = W(query) (lsa.py:25): thread (d, quarter of k), partials through LDS
  {
    const int nq = 512 / D > 0 ? 512 / D : 1;  // k-parts per output (4 for D=128)
    const int d = tid % D, kp = tid / D;
    float acc = 0.f;
    if (kp < nq) {
      const float* wr = a.Ww + (size_t)d * D;
      const int k0 = kp * (D / nq), k1 = (kp == nq - 1) ? D : k0 + D / nq;
      for (int k = k0; k < k1; ++k) acc += wr[k] * a.query[lsa_qidx(a, b, k)];
      part[kp * D + d] = acc;
    }
    __syncthreads();
    if (tid < D) {
      float sacc = 0.f;
      for (int j = 0; j < nq; ++j) sacc += part[j * D + tid];
      pq[tid] = sacc + a.Wb[tid];
    }
  }
  __syncthreads();
  // location features: conv1d(1 -> Fl, k = Kl, same padding) over the cumulative attention (lsa.py:27-28)
  for (int i = tid; i < T * Fl; i += 512) {
    const int t = i / Fl, f = i - t * Fl;
    const float* wf = cw + f * Kl;
    float acc = 0.f;
    for (int j = 0; j < Kl; ++j) acc += wf[j] * cum[t + j];
    loc[i] = acc + a.conv_b[f];
  }
  __syncthreads();
  // u[t] = v . tanh(pq + mem_proj[t] + L(loc[t])) (lsa.py:28-31): one wave per t, lanes over d
  const float* mp = a.mem_proj + (size_t)b * T * D;
  for (int t = wave; t < T; t += 8) {
    float pacc = 0.f;
    const float* lt = loc + (size_t)t * Fl;
    for (int d = lane; d < D; d += 64) {
      float pl = 0.f;
      for (int f = 0; f < Fl; ++f) pl += LwT[f * D + d] * lt[f];
      pacc += a.vw[d] * tanhf((pq[d] + mp[(size_t)t * D + d]) + pl);
    }
    pacc = wave_sum(pacc);
    if (lane == 0) u[t] = a.chars[(size_t)b * T + t] != 0 ? pacc : pacc * 0.f;  // u * (chars != 0) (lsa.py:34)
  }
  __syncthreads();
  // softmax over T (lsa.py:38)
  float m = -INFINITY;
  for (int t = tid; t < T; t += 512) m = fmaxf(m, u[t]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = red[0];
  for (int w = 1; w < 8; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  float ssum = 0.f;
  for (int t = tid; t < T; t += 512) { const float e = expf(u[t] - m); u[t] = e; ssum += e; }
  ssum = wave_sum(ssum);
  if (lane == 0) red[wave] = ssum;
  __syncthreads();
  ssum = 0.f;
  for (int w = 0; w < 8; ++w) ssum += red[w];
  const int iter = a.iter + (a.iter_base ? *a.iter_base : 0);
  float* ao = (a.attn_out && pg == 0) ? a.attn_out + ((size_t)b * a.n_iter_max + iter) * T : nullptr;
  for (int t = tid; t < T; t += 512) {
    const float sc = u[t] / ssum;
    u[t] = sc;
    if (pg == 0) a.cum_out[(size_t)b * T + t] = cum[t + half] + sc;  // cumulative += attention (lsa.py:40)
    if (ao) ao[t] = sc;
  }
  __syncthreads();
  // context = scores @ encoder_seq (tacotron.py:104) for this group's columns [p0, p0 + pw)
  const int pw = a.P / a.psplit, p0 = pg * pw;
  const float* mem = a.memory + (size_t)b * T * a.P + p0;
  for (int c4 = lane * 4; c4 < pw; c4 += 256) {  // 64 lanes x float4 = 256 columns per sweep
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int t = wave;
    for (; t + 56 < T; t += 64) {  // 8 independent rows in flight
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(mem + (size_t)(t + 8 * j) * a.P + c4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sc = u[t + 8 * j];
        acc.x += sc * v[j].x; acc.y += sc * v[j].y; acc.z += sc * v[j].z; acc.w += sc * v[j].w;
      }
    }
    for (; t < T; t += 8) {
      const float4 v = *reinterpret_cast<const float4*>(mem + (size_t)t * a.P + c4);
      const float sc = u[t];
      acc.x += sc * v.x; acc.y += sc * v.y; acc.z += sc * v.z; acc.w += sc * v.w;
    }
    reinterpret_cast<float4*>(part)[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0) {
      float4 r = reinterpret_cast<float4*>(part)[lane];
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        const float4 o = reinterpret_cast<float4*>(part)[w * 64 + lane];
        r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
      }
      *lsa_ctx4(a, b, p0 + c4) = r;
    }
    __syncthreads();
  }
}

// Latency-first LSA for the production shape (D = 128, context column group of 256, Kl <= 31, T <= 4*TJ).
// History of this launch at B = 32, T = 111 (profiles/r02_taco_trace_*.json): 19.6 us as a VALU kernel with one
// dword load per operand word (118 vector-memory instructions per lane: bound by the instruction count), 12.1 us with
// 16-byte operand rows, a packed-FMA location conv and the memory rows brought in by LDS-DMA -- of which 6.9 us were
// the energy phase, VALU-issue-bound.  Now:
//   * every global load -- query, cumulative attention, W rows, tap-matrix and processed-memory fragments -- is issued
//     before the first wait, 16 bytes per lane and instruction;
//   * conv1d(1->32,k=31) followed by L (32->128) is one 31-tap conv with the folded matrix M = L.conv_w, run on the
//     MATRIX pipe: loc[t][d] = sum_j cum[t+j-15] M[d][j] is a [T x 32] x [32 x 128] product whose A operand is a Toeplitz
//     view of the zero-padded cumulative attention in LDS.  Wave w owns positions 16 w .. 16 w + 15 (and 16 (w+8) ..
//     for T > 128): 8 d-tiles x 8 k-steps of v_mfma_f32_16x16x4_f32 (exact fp32, same products as the FMA form), its D
//     fragment hands every lane loc for 4 positions x 8 d, so tanh / v-weighting / the reduction over d (in-lane over
//     the d-tiles, one 16-lane DPP row sum across them) need no cross-wave partials;
//   * the workgroup's 256 context columns of every memory row arrive by LDS-DMA (global_load_lds: no registers)
//     while the energies are computed (T <= 128; for longer texts they are loaded to registers after the energies);
//   * 5 workgroup barriers in total.
// tanh is evaluated as 1 - 2 rcp(exp(2x)+1) (absolute error ~1e-7, the parity bar on attention is 1e-4).
// FUSED (taco_front_kernel): the query arrives inside the launch, ~4 us after its start.  Everything that does not depend on it runs
// in front of the wait -- the window staging, the location term's MFMAs (accumulators kept) -- so that processed query, tanh /
// v-weighting, softmax and context are what is left behind it.  (The LDS-DMA of the memory rows stays behind B2: queued early it
// competes with the W_hh2 tiles of the same launch and, loads returning in order, holds the query poll back by a microsecond.)  Same operations on the same operands in the
// same order per accumulator: the two forms give the same bits.
template <int TJ, bool FUSED = false, bool FOLD = false>
__device__ __forceinline__ void lsa_fast_body(const LsaK& a, const int b, const int pg, float* s_mem = nullptr) {
  constexpr int D = 128, TMAX = 4 * TJ, TM = TMAX / 8, PW = 256, NTILE = TMAX / 16, NPASS = (NTILE + 7) / 8;
  // ES (fused launch, T <= 128): the four workgroups of an utterance (one per 256 context columns) need the same T energies -- 32 tanh
  // per lane, bound by the transcendental rate (3.5 of the 6.6 us behind the query).  Each computes the energies of ITS two position
  // tiles only, wave w taking d-tiles 2 (w & 3), +1 of tile 2 pg + (w >> 2); the tanh values meet in LDS, waves 0 / 1 run the
  // v-weighted sum over d in the order of the one-workgroup form (same bits) and publish the 32 energies as {value, iteration + 1}
  // granules; the other 96 come from the three sibling workgroups the same way.
  constexpr bool ES = FUSED && TJ == 32;
  constexpr int ND = ES ? 2 : 8;  // d-tiles per wave
  __shared__ __attribute__((aligned(16))) float4 s_th[ES ? 2 : 1][ES ? 8 : 1][ES ? 64 : 1];
  __shared__ __attribute__((aligned(16))) float s_cum[TMAX + 64];   // zero padded by `half` on both sides
  __shared__ __attribute__((aligned(16))) float s_q[D];
  __shared__ __attribute__((aligned(16))) float s_pq[4][D];
  __shared__ __attribute__((aligned(16))) float s_vc[2][D];         // v, c0
  __shared__ __attribute__((aligned(16))) float s_e[TMAX];          // energies
  __shared__ __attribute__((aligned(16))) float s_u[TMAX];          // scores
  __shared__ __attribute__((aligned(16))) float4 s_part[8][64];
  // folded form: partial sums of the next GRU pre-activation (32 unit quads per wave); ES: in the tanh exchange buffer, dead by then
  __shared__ __attribute__((aligned(16))) float4 s_part2s[(FUSED && FOLD && !ES) ? 8 : 1][64];
  float4 (*s_part2)[64] = ES ? reinterpret_cast<float4 (*)[64]>(&s_th[0][0][0]) : s_part2s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = tid & (D - 1), tq = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int T = a.T, P = a.P, half = (a.Kl - 1) / 2;
  int skip = 0;
  if (a.skip_flag) skip = *a.skip_flag;
  const bool pick = b == 5 && pg == 1;
  tf_mark(a.trace, TS_LSA, 0, pick);

  // ---- phase 0: every global load ----
  float qv = (tid < D && !FUSED) ? a.query[lsa_qidx(a, b, tid)] : 0.f;  // fresh (attention GRU output)
  const int iter = a.iter + (a.iter_base ? *a.iter_base : 0);
  const float* cg = a.cum_in + (size_t)b * T;
  float cpre[(TMAX + 64 + 511) / 512];
#pragma unroll
  for (int m = 0; m < (TMAX + 64 + 511) / 512; ++m) {
    const int i2 = tid + 512 * m, t = i2 - half;
    cpre[m] = (i2 < T + 2 * half && t >= 0 && t < T) ? cg[t] : 0.f;
  }
  float4 wq4[8];  // W[d][k] for k in this thread's quarter
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) wq4[k4] = a.Wq4[(size_t)(tq * 8 + k4) * D + d];
  // B fragments of the tap matrix: lane (i = lane & 15, kq = lane >> 4), d-tile dt, k-step g: M[16 dt + i][4 g + kq]
  const int dt0 = ES ? 2 * (wave & 3) : 0;  // this wave's d-tiles: [dt0, dt0 + ND)
  float4 mf[ND][2];
#pragma unroll
  for (int dl = 0; dl < ND; ++dl)
#pragma unroll
    for (int gh = 0; gh < 2; ++gh) mf[dl][gh] = a.Mf4[(size_t)((dt0 + dl) * 2 + gh) * 64 + lane];
  // processed memory in D-fragment order: lane (i, rq), tile, d-tile: mem_proj[16 tile + 4 rq + 0..3][16 dt + i]
  float4 mpf[NPASS][ND];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int tile = ES ? 2 * pg + (wave >> 2) : (wave + 8 * ps < NTILE ? wave + 8 * ps : NTILE - 1);
#pragma unroll
    for (int dl = 0; dl < ND; ++dl) mpf[ps][dl] = a.mpf4[(((size_t)b * NTILE + tile) * 8 + dt0 + dl) * 64 + lane];
  }
  const float wb = a.Wb[d];
  const float vc = (tid < 2 * D) ? (tid < D ? a.vw[tid] : a.c0[tid - D]) : 0.f;
  // chars of this thread's softmax positions (mask)
  int chv[(TMAX + 63) / 64];
  if (FUSED) {
#pragma unroll
    for (int m = 0; m < (TMAX + 63) / 64; ++m) {
      const int t = lane + 64 * m;
      chv[m] = (t < T) ? a.chars[(size_t)b * T + t] : 0;
    }
  }
  const int p0 = pg * PW;
  const int ld = a.mem_ld ? a.mem_ld : P;
  const float* mem = a.memory + (size_t)b * T * ld + p0 + lane * 4;
  const bool dma = TJ == 32 && s_mem != nullptr;
  constexpr bool fold = FUSED && FOLD;
  // folded form: wave 0 runs the softmax alone, so the rows of the score-weighted sums belong to waves 1..7 (their operands arrive
  // while they wait for it) -- wave 0 would meet its own loads at the barrier behind the softmax
  constexpr int TMF = fold ? (TMAX + 6) / 7 : TM;  // rows per wave
  // GRU pre-activation rows: waves 2..7 (the ones without a poll in the energy exchange: loads return in order), requested right behind
  // that exchange's first barrier -- the hop and the softmax are their time to arrive (requested in front of the query wait instead:
  // 44 more registers held for 8 us, the same 32.5 us per iteration); lanes 0..31 take the even rows of the wave, 32..63 the odd ones
  constexpr int TM6 = fold ? (TMAX + 5) / 6 : 1, TMH = (TM6 + 1) / 2;
  static_assert(!fold || ES, "the folded form is the LDS-window form (T <= 128)");
  float4 pm2v[fold ? TMH : 1];
  // rnn_input's attn_hidden columns for this wave's 16 k: requested with everything else, multiplied as soon as the query is staged
  float4 ra4[fold ? 16 : 1];
  if (fold) {
#pragma unroll
    for (int k = 0; k < 16; ++k) ra4[k] = a.rin_a4[((size_t)pg * D + wave * 16 + k) * 64 + lane];
  }

  // folded form: the stop logit's context part of this thread's softmax positions, rnn_input's bias for this lane's four columns
  float pm3[(TMAX + 63) / 64];
  float4 brin = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fold) {
#pragma unroll
    for (int m = 0; m < (TMAX + 63) / 64; ++m) {
      const int t = lane + 64 * m;
      pm3[m] = (t < T && pg == 0 && wave == 0) ? a.memory[((size_t)b * T + t) * ld + P + 4 * D] : 0.f;
    }
    if (wave == 0) brin = *reinterpret_cast<const float4*>(a.rin_b + p0 + lane * 4);
  }
  // staging
  if (!FUSED && tid < D) s_q[tid] = qv;
  if (tid < 2 * D) s_vc[tid >> 7][tid & (D - 1)] = vc;
#pragma unroll
  for (int m = 0; m < (TMAX + 64 + 511) / 512; ++m) {
    const int i2 = tid + 512 * m;
    if (i2 < TMAX + 64) s_cum[i2] = cpre[m];
  }
  f32x4 accs[FUSED ? NPASS : 1][ND];
  if (FUSED) {
    __syncthreads();  // B1a: window staged
    if (ES && dma && a.dma_early) {
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        asm volatile("" : "+v"(mf[dt][0].x), "+v"(mf[dt][0].y), "+v"(mf[dt][0].z), "+v"(mf[dt][0].w));
        asm volatile("" : "+v"(mf[dt][1].x), "+v"(mf[dt][1].y), "+v"(mf[dt][1].z), "+v"(mf[dt][1].w));
        asm volatile("" : "+v"(mpf[0][dt].x), "+v"(mpf[0][dt].y), "+v"(mpf[0][dt].z), "+v"(mpf[0][dt].w));
      }
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) asm volatile("" : "+v"(wq4[k4].x), "+v"(wq4[k4].y), "+v"(wq4[k4].z), "+v"(wq4[k4].w));
#pragma unroll
      for (int m = 0; m < (TMAX + 63) / 64; ++m) asm volatile("" : "+v"(chv[m]));
      if (wave >= 2) {
#pragma unroll
        for (int j = 0; j < (TMAX + 5) / 6; ++j) {
          const int t = (wave - 2) + 6 * j;  // wave-uniform
          if (t < T)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mem + (size_t)t * ld),
                                             (__attribute__((address_space(3))) void*)(s_mem + t * PW), 16, 0, 0);
        }
      }
    }
    // location term on the matrix pipe (phase 2's MFMAs, same order per accumulator)
    {
      const int i = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int tile = ES ? 2 * pg + (wave >> 2) : wave + 8 * ps;
#pragma unroll
        for (int dl = 0; dl < ND; ++dl) accs[ps][dl] = {0.f, 0.f, 0.f, 0.f};
        if (tile * 16 < T) {
          float av[8];
#pragma unroll
          for (int g = 0; g < 8; ++g) av[g] = s_cum[tile * 16 + i + 4 * g + kq];
#pragma unroll
          for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int dl = 0; dl < ND; ++dl) {
              const float4 m4 = mf[dl][g >> 2];
              const float bv = (g & 3) == 0 ? m4.x : (g & 3) == 1 ? m4.y : (g & 3) == 2 ? m4.z : m4.w;
              accs[ps][dl] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g], bv, accs[ps][dl], 0, 0, 0);
            }
        }
      }
    }
    if (tid < D && !skip) {  // now wait for this utterance's attention-GRU output (every thread its own granule: ONE watching lane per
      // workgroup + a barrier + one read each was measured -- 32.64 against 32.63 us per iteration, 30.4 against 30.0 at batch 16)
      qv = __uint_as_float((unsigned)wp_wait2(a.q_gran + (size_t)b * D + tid, (unsigned)iter + 1u, a.lost));
    }
    if (tid < D) s_q[tid] = qv;
    if (a.trace && tid == 0 && pg * 32 + b < 128 && a.fm_nta) a.trace[TS_WG + 2 * (pg * a.fm_nta * 16 + b)] = (unsigned long long)wall_clock64();  // (fused launch: l = pg B + b; approximated for B = 16 nta)
  }
  tf_mark(a.trace, TS_LSA, 1, pick);
  __syncthreads();  // B1
  tf_mark(a.trace, TS_LSA, 2, pick);
  // ---- phase 1: processed query partials (lsa.py:25); the bias rides in partial 0 ----
  {
    float acc = tq == 0 ? wb : 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
      const float4 q4 = *reinterpret_cast<const float4*>(&s_q[tq * 32 + k4 * 4]);  // same address for the whole wave: broadcast
      acc += wq4[k4].x * q4.x; acc += wq4[k4].y * q4.y; acc += wq4[k4].z * q4.z; acc += wq4[k4].w * q4.w;
    }
    s_pq[tq][d] = acc;
  }
  float4 acc_ra = make_float4(0.f, 0.f, 0.f, 0.f);  // folded form: this wave's k slice of W_rin[:, P:] . attn_hidden for the lane's four columns
  if (fold) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float qk = s_q[wave * 16 + k];
      acc_ra.x += ra4[k].x * qk; acc_ra.y += ra4[k].y * qk; acc_ra.z += ra4[k].z * qk; acc_ra.w += ra4[k].w * qk;
    }
  }
  // chars of this thread's softmax positions (mask), requested early as well
  if (!FUSED) {
#pragma unroll
    for (int m = 0; m < (TMAX + 63) / 64; ++m) {
      const int t = lane + 64 * m;
      chv[m] = (t < T) ? a.chars[(size_t)b * T + t] : 0;
    }
  }
  if (dma) {
    // Every ordinary load above must have LANDED before the DMA is queued: hipcc waits vmcnt(0) at the next use of a
    // loaded value while an LDS-DMA is in flight, which would put the 128 KB in front of the energy phase.  Touching
    // the operands here makes that wait happen now (they were requested microseconds ago), with nothing else outstanding.
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      asm volatile("" : "+v"(mf[dt][0].x), "+v"(mf[dt][0].y), "+v"(mf[dt][0].z), "+v"(mf[dt][0].w));
      asm volatile("" : "+v"(mf[dt][1].x), "+v"(mf[dt][1].y), "+v"(mf[dt][1].z), "+v"(mf[dt][1].w));
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) asm volatile("" : "+v"(mpf[ps][dt].x), "+v"(mpf[ps][dt].y), "+v"(mpf[ps][dt].z), "+v"(mpf[ps][dt].w));
    }
#pragma unroll
    for (int m = 0; m < (TMAX + 63) / 64; ++m) asm volatile("" : "+v"(chv[m]));
  }
  __syncthreads();  // B2
  tf_mark(a.trace, TS_LSA, 3, pick);
  if (dma && !(ES && a.dma_early)) {  // queued behind B2: a barrier drains the DMA queue (hipcc emits vmcnt(0) in front of s_barrier)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int t = wave + 8 * j;  // wave-uniform: row t lands at s_mem[t][0..255], lane l -> floats [4 l, 4 l + 4)
      if (t < T)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mem + (size_t)t * ld),
                                         (__attribute__((address_space(3))) void*)(s_mem + t * PW), 16, 0, 0);
    }
  }
  // ---- phase 2+3: location term on the matrix pipe, energies, reduction over d ----
  if constexpr (ES) {
    const int i = lane & 15, kq = lane >> 4, tl = wave >> 2, tile = 2 * pg + tl;
    if (tile * 16 < T) {  // wave-uniform
      float4 th4[2];
#pragma unroll
      for (int dl = 0; dl < 2; ++dl) {
        const int dd = (dt0 + dl) * 16 + i;
        const float pqd = (s_pq[0][dd] + s_pq[1][dd]) + (s_pq[2][dd] + s_pq[3][dd]), c0d = s_vc[1][dd];
        const float mpr[4] = {mpf[0][dl].x, mpf[0][dl].y, mpf[0][dl].z, mpf[0][dl].w};
        float th[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = (pqd + mpr[r]) + (accs[0][dl][r] + c0d);
          th[r] = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f);
        }
        th4[dl] = make_float4(th[0], th[1], th[2], th[3]);
      }
      s_th[tl][dt0][lane] = th4[0];
      s_th[tl][dt0 + 1][lane] = th4[1];
    }
    __syncthreads();  // B2a: the tanh values of this workgroup's two tiles
    if (fold && wave >= 2) {  // next GRU pre-activation rows of this workgroup's 32 units: lane (l & 31) holds unit 32 pg + (l & 31)'s (r, z, n, 0) quad
      const float* p2 = a.memory + (size_t)b * T * ld + P + 128 * pg + 4 * (lane & 31);
#pragma unroll
      for (int j = 0; j < TMH; ++j) {
        const int rr = 2 * j + (lane >> 5), t = (wave - 2) + 6 * rr;
        pm2v[j] = (t < T && rr < TM6) ? *reinterpret_cast<const float4*>(p2 + (size_t)t * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (wave < 2) {
      const int tg = 2 * pg + wave;
      const unsigned tag = (unsigned)iter + 1u;
      if (tg * 16 < T) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const float vvd = s_vc[0][dt * 16 + i];
          const float4 t4 = s_th[wave][dt][lane];
          const float th[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] += vvd * th[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float es = row16_sum(e[r]);
          const int t = tg * 16 + 4 * kq + r;
          if (i == 0 && t < T) {
            s_e[t] = es;
            if (!skip) wp_put(a.e_gran + (size_t)b * TMAX + t, es, tag);
          }
        }
      }
      // the sibling workgroups' 96 energies
      if (tid < T && (tid >> 5) != pg && !skip) {
        s_e[tid] = __uint_as_float((unsigned)wp_wait2(a.e_gran + (size_t)b * TMAX + tid, tag, a.lost));
      }
    }
  } else {
    const int i = lane & 15, kq = lane >> 4;
    float pqv[8], vv[8], c0v[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
      const int dd = dt * 16 + i;
      pqv[dt] = (s_pq[0][dd] + s_pq[1][dd]) + (s_pq[2][dd] + s_pq[3][dd]);
      vv[dt] = s_vc[0][dd]; c0v[dt] = s_vc[1][dd];
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int tile = wave + 8 * ps;
      if (tile * 16 < T) {  // wave-uniform
        // A fragments: cum_pad[16 tile + i + 4 g + kq] (s_cum is padded by `half`: index t + j is cum[t + j - half])
        float av[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) av[g] = s_cum[tile * 16 + i + 4 * g + kq];
        f32x4 acc[8];
        if (FUSED) {
#pragma unroll
          for (int dt = 0; dt < 8; ++dt) acc[dt] = accs[ps][dt];
        } else {
#pragma unroll
          for (int dt = 0; dt < 8; ++dt) acc[dt] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int g = 0; g < 8; ++g)
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {  // consecutive MFMAs on different accumulators
              const float4 m4 = mf[dt][g >> 2];
              const float bv = (g & 3) == 0 ? m4.x : (g & 3) == 1 ? m4.y : (g & 3) == 2 ? m4.z : m4.w;
              acc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g], bv, acc[dt], 0, 0, 0);
            }
        }
        // D fragment: acc[dt][r] = loc[16 tile + 4 kq + r][16 dt + i]
        float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
          const float mpr[4] = {mpf[ps][dt].x, mpf[ps][dt].y, mpf[ps][dt].z, mpf[ps][dt].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = (pqv[dt] + mpr[r]) + (acc[dt][r] + c0v[dt]);
            const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f);  // tanh, 1 ulp reciprocal
            e[r] += vv[dt] * th;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float es = row16_sum(e[r]);  // over the 16 d of a tile column group: 4 VALU ops
          const int t = tile * 16 + 4 * kq + r;
          if (i == 0 && t < T) s_e[t] = es;
        }
      }
    }
  }
  __syncthreads();  // B3
  tf_mark(a.trace, TS_LSA, 5, pick);
  // context rows (stable data) of the register path (no LDS window): requested now, they arrive while wave 0 runs the softmax
  float4 memv[TM];
  if (!dma && !fold) {
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int t = wave + 8 * j;
      memv[j] = (t < T) ? *reinterpret_cast<const float4*>(mem + (size_t)t * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // ---- phase 4: mask, softmax over T (lsa.py:34-38) by wave 0, cumulative update ----
  if (wave == 0) {
    float uv[(TMAX + 63) / 64];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < (TMAX + 63) / 64; ++q) {
      const int t = lane + 64 * q;
      float u = -INFINITY;
      if (t < T) { u = s_e[t]; u = chv[q] != 0 ? u : u * 0.f; }  // u * (chars != 0)
      uv[q] = u;
      m = fmaxf(m, u);
    }
    m = wave64_max(m);
    float ssum = 0.f;
#pragma unroll
    for (int q = 0; q < (TMAX + 63) / 64; ++q) {
      const int t = lane + 64 * q;
      uv[q] = (t < T) ? expf(uv[q] - m) : 0.f;
      ssum += uv[q];
    }
    ssum = wave64_sum(ssum);
    float* ao = (a.attn_out && pg == 0) ? a.attn_out + ((size_t)b * a.n_iter_max + iter) * T : nullptr;
    float sp = 0.f;
#pragma unroll
    for (int q = 0; q < (TMAX + 63) / 64; ++q) {
      const int t = lane + 64 * q;
      if (t < T) {
        const float sc = uv[q] / ssum;
        s_u[t] = sc;
        if (fold) sp += sc * pm3[q];
        if (pg == 0 && !skip) {
          a.cum_out[(size_t)b * T + t] = s_cum[t + half] + sc;  // cumulative += attention (lsa.py:40)
          if (ao) ao[t] = sc;
        }
      }
    }
    if (fold && pg == 0) {  // stop_proj's context half (tacotron.py:133-135): sum_t score_t (w_stop[H:] . memory_t)
      sp = wave64_sum(sp);
      if (lane == 0 && !skip) a.stop_part[b] = sp;
    }
  }
  __syncthreads();  // B4
  tf_mark(a.trace, TS_LSA, 6, pick);
  // ---- phase 5: context = scores @ encoder_seq (tacotron.py:104), this group's 256 columns ----
  if constexpr (fold) {
    float4 acc = acc_ra, acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < TMF; ++j) {
        const int t = (wave - 1) + 7 * j;
        const float sc = (t < T) ? s_u[t] : 0.f;
        float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) mv = *reinterpret_cast<const float4*>(s_mem + t * PW + lane * 4);
        acc.x += sc * mv.x; acc.y += sc * mv.y; acc.z += sc * mv.z; acc.w += sc * mv.w;
      }
      if (wave >= 2) {
#pragma unroll
        for (int j = 0; j < TMH; ++j) {
          const int rr = 2 * j + (lane >> 5), t = (wave - 2) + 6 * rr;
          const float sc = (t < T && rr < TM6) ? s_u[t] : 0.f;
          acc2.x += sc * pm2v[j].x; acc2.y += sc * pm2v[j].y; acc2.z += sc * pm2v[j].z; acc2.w += sc * pm2v[j].w;
        }
      }
    }
    s_part2[wave][lane] = acc2;
    s_part[wave][lane] = acc;
  } else {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int t = wave + 8 * j;
      const float sc = (t < T) ? s_u[t] : 0.f;
      float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (dma) { if (t < T) mv = *reinterpret_cast<const float4*>(s_mem + t * PW + lane * 4); }
      else mv = memv[j];
      acc.x += sc * mv.x; acc.y += sc * mv.y; acc.z += sc * mv.z; acc.w += sc * mv.w;
    }
    s_part[wave][lane] = acc;
  }
  __syncthreads();  // B5
  tf_mark(a.trace, TS_LSA, 7, pick);
  if (wave == 0 && !skip) {
    float4 r = s_part[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const float4 o = s_part[w][lane];
      r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    if (fold) { r.x += brin.x; r.y += brin.y; r.z += brin.z; r.w += brin.w; }  // x = rnn_input([context, attn_hidden])
    *lsa_ctx4(a, b, p0 + lane * 4) = r;
  }
  if (fold && wave == 1 && lane < 32 && !skip) {  // W_ih[:, :P] . context + b_ih of unit 32 pg + lane: the next iteration's GRU reads it
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 2; w < 8; ++w) {  // (waves 0 / 1 hold no rows)
      const float4 o = s_part2[w][lane], o2 = s_part2[w][lane + 32];
      r.x += o.x + o2.x; r.y += o.y + o2.y; r.z += o.z + o2.z; r.w += o.w + o2.w;
    }
    const int u = 32 * pg + lane;
    const float4 bi = a.bih4[u];
    a.xpre_out[((size_t)(u >> 2) * a.fm_nta + (b >> 4)) * 64 + (u & 3) * 16 + (b & 15)] = make_float4(r.x + bi.x, r.y + bi.y, r.z + bi.z, 0.f);
  }
  tf_mark_end(a.trace, TS_LSA, 8, pick);
}

// mem_proj [B][T][D] -> the order in which the lanes of lsa_fast_body hold the location term (MFMA D fragments):
// float4 (b, tile, dt, lane = rq * 16 + i) = mem_proj[b][16 tile + 4 rq + 0..3][16 dt + i], zero beyond T.  Once per decode call.
__global__ void lsa_pack_memproj_kernel(const float* __restrict__ mp, float4* __restrict__ out, int B, int T, int ntile) {
  constexpr int D = 128;
  const size_t n = (size_t)B * ntile * 8 * 64;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63), dt = (int)((idx >> 6) & 7);
    const int tile = (int)((idx >> 9) % ntile), b = (int)((idx >> 9) / ntile);
    const int i = lane & 15, rq = lane >> 4;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = tile * 16 + rq * 4 + r;
      v[r] = t < T ? mp[((size_t)b * T + t) * D + dt * 16 + i] : 0.f;
    }
    out[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int TJ>
__global__ __launch_bounds__(512) void lsa_fast_kernel(LsaK a) {
  __shared__ __attribute__((aligned(16))) float s_big[TJ == 32 ? 32 * 4 * 256 : 4];
  lsa_fast_body<TJ>(a, blockIdx.x, blockIdx.y, TJ == 32 ? s_big : nullptr);
}
// The attention launch occupies B * psplit of the 256 CUs for ~10 us of dependent latencies.  The hidden half of the
// decoder's SECOND LSTM, W_hh2 . h2 (16.8 MB of weights), depends only on the previous iteration's state: it rides here
// as extra workgroups on the idle CUs (the attention workgroups come first in dispatch order) and leaves the chain.
template <int TJ, int NT>
__global__ __launch_bounds__(512) void lsa_hh_kernel(LsaK a, TfHhK hh, int n_lsa, int B, int gy, int nta) {
  // one LDS block serves both jobs: the attention's 128 KB memory window (TJ = 32) / the hh job's reduction buffer
  __shared__ __attribute__((aligned(16))) float s_big[TJ == 32 ? 32 * 4 * 256 : FmRed<NT, 1>::floats];
  float* red = s_big;
  const int id = blockIdx.x;
  if (id < n_lsa) { lsa_fast_body<TJ>(a, id % B, id / B, TJ == 32 ? s_big : nullptr); return; }
  const int j = id - n_lsa, mt = j / gy;
  fm_hh_job<NT>(hh, mt, (j - mt * gy) * NT, nta, a.skip_flag ? *a.skip_flag : 0, red);
}

constexpr int TACO_PM_LD = 1664;  // floats per projected-memory row: H + 4 D + 1 = 1537 rows of W, padded to 13 x 128 output channels
// [B][T][C] -> [B][C][T] (the attention memory as the 1 x 1 conv's channel-major input), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void btc_to_bct_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (t < T && c < C) ? x[((size_t)b * T + t) * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    if (c < C && t < T) y[((size_t)b * C + c) * T + t] = tile[tx][ty + 8 * i];
  }
}

// ---------------------------------------------------------------- fused front of the fast decoder loop
// Launches 1..3 of an iteration (prenet fc2 | attention GRU | attention + context, 4.6 + 4.7 + 11.7 us in
// profiles/r05_bench_kernel_stats.csv) carry 2.5 + 1.5 us of work in front of the attention: each of the two small launches is a launch
// boundary plus a cold start of its weight loads.  taco_front_kernel runs the three as ROLES of one launch -- workgroups [0, 16) fc2
// row tiles, [16, 48) attention-GRU unit tiles, then the B x psplit attention workgroups, then the W_hh2 . h2 tiles as before -- with
// two in-launch hand-offs on tagged granules (granule.h: one 8-byte {value, iteration + 1} store per element, no fence, the tag is the
// flag; the buffers are zeroed per decode call, every iteration rewrites all of them, so one buffer serves every iteration):
//   p2g  prenet output in the attention GRU's B-fragment order, ((k-block * nta + column tile) * 4 + r) * 64 + lane: the fc2 epilogue
//        lane that owns rows 4 du + r of column i writes what lane (kq = du, i) of the GRU's MFMA reads;
//   ahg  attention-GRU output [column][D], read by thread d of every attention workgroup of that utterance.
// The roles lower in the chain have every other operand -- weights, gate pre-activations, the attention window, the processed memory --
// requested or landed when the tag arrives.  The arithmetic is the three kernels' own (fm_gemm's MFMA and reduction order): the fused
// and the three-launch forms give the same bits.  Needs the 16 + 32 + B psplit producers/consumers co-resident (they are the lowest
// block indices of a launch that starts on an idle device); a wait older than 0.2 s raises flags[TF_LOST] and the host reruns the
// call with three launches.
struct TfFrontX {
  unsigned long long* p2g; unsigned long long* ahg; int* lost;
  int n_fc2, n_gru, watch;
  int hh_pairs;  // the hh2 workgroups take two row tiles each (fm_hh_pair_job)
};
template <int NT>
__device__ __forceinline__ void front_fc2_job(const TfFcK& a, const TfFrontX& x, const int mt, float* red) {
  if (a.flags[TF_DONE]) return;
  const int it = a.flags[TF_ITER] + a.it_off;
  float sx[4], sh[4];
  const bool pick = mt == 3;
  tf_mark(a.trace, TS_FC2, 0, pick);
  const int lane = threadIdx.x & 63, nt = threadIdx.x >> 6;
  const int du = lane >> 4, n = nt * 16 + (lane & 15), row0 = mt * 16 + du * 4;
  const float4 bq = *reinterpret_cast<const float4*>(a.bias + row0);  // (requested with the fragments, not behind the reduction)
  float keep[4] = {1.f, 1.f, 1.f, 1.f};  // the dropout draw does not depend on the sums either
  if (nt < a.nta) drop_quad_factors(a.drop, a.flags, it, n < a.B ? n : a.B - 1, row0, keep);
  if (!fm_gemm<NT, 2, 2, 4, 1>(a.w, mt, a.xin, a.xin, a.nta, 0, red, sx, sh, a.trace, TS_FC2, pick)) return;
  if (nt >= a.nta) return;
  float v[4] = {sx[0] + bq.x, sx[1] + bq.y, sx[2] + bq.z, sx[3] + bq.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f) * keep[r];
  unsigned long long* gp = x.p2g + ((size_t)(mt * a.nta + nt) * 4) * 64 + lane;
#pragma unroll
  for (int r = 0; r < 4; ++r) wp_put(gp + r * 64, v[r], (unsigned)it + 1u);
  tf_mark_end(a.trace, TS_FC2, 4, pick);
}
template <int NT>
__device__ __forceinline__ void front_gru_job(const TfGruK& a, const TfFrontX& x, const int mt, const int it, float* red) {
  constexpr int RL = 3, BLK = 4 * RL * 16, PW = 2;
  if (a.flags[TF_DONE]) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4, du = kq;
  const bool pick = mt == 3;
  tf_mark(a.trace, TS_GRU, 0, pick);
  // weights, gate pre-activations, previous state: requested before the wait
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;
  const float* wl = a.w + (size_t)mt * (8 * PW) * BLK + ((u * RL + tau) * 4 + kq) * 4;
  float4 wa[PW];
#pragma unroll
  for (int p = 0; p < PW; ++p) wa[p] = *reinterpret_cast<const float4*>(wl + (size_t)(wave + 8 * p) * BLK);
  const int ntE = (wave < NT && wave < a.nta) ? wave : a.nta - 1;
  const size_t cm = ((size_t)mt * a.nta + ntE) * 64 + lane;
  const bool own_h = a.w_pre != nullptr;
  const float4 xp = a.xpre[cm];
  float4 hp = own_h ? a.bhh4[mt * 4 + du] : a.hpre[cm];
  const size_t ho = ((size_t)(mt >> 2) * a.nta + ntE) * 256 + (mt & 3) * 64 + i * 4 + du;
  float* hpt = (own_h ? a.ah_out : a.ah) + ho;
  const float hprev = a.ah[ho];
  int ntc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) ntc[nt] = nt < a.nta ? nt : a.nta - 1;
  // folded form: W_hh . attn_hidden(t-1) for this tile, k-block `wave` of the hidden part (fm_gemm<NT, 9, 8, 3, 2>'s accH)
  f32x4 accH[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) accH[nt] = {0.f, 0.f, 0.f, 0.f};
  if (own_h) {
    const float4 wh = *reinterpret_cast<const float4*>(a.w_pre + (size_t)mt * 72 * BLK + (size_t)(64 + wave) * BLK + ((u * RL + tau) * 4 + kq) * 4);
    float4 bh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bh[nt] = reinterpret_cast<const float4*>(a.ah)[((size_t)wave * a.nta + ntc[nt]) * 64 + lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float av = c == 0 ? wh.x : c == 1 ? wh.y : c == 2 ? wh.z : wh.w;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float bv = c == 0 ? bh[nt].x : c == 1 ? bh[nt].y : c == 2 ? bh[nt].z : bh[nt].w;
        accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accH[nt], 0, 0, 0);
      }
    }
  }
  const unsigned tag = (unsigned)it + 1u;
  // watch: lane j < PW * NT of every wave polls the last granule of one of the wave's fragments, then one sweep (re-read while stale)
  if (x.watch) {
    const int j = lane < PW * NT ? lane : 0, p = j / NT, nt = j - p * NT;
    const unsigned long long* wp = x.p2g + ((size_t)((wave + 8 * p) * a.nta + (nt < a.nta ? nt : a.nta - 1)) * 4 + 3) * 64 + 63;
    unsigned long long t0 = 0, cur = wp_get(wp);
    for (int tries = 0;; ++tries) {  // two polls in flight (granule.h wp_wait2)
      const unsigned long long nxt = wp_get(wp);
      const bool fresh = (unsigned)(cur >> 32) == tag;
      if (__builtin_amdgcn_ballot_w64(!fresh) == 0ull) break;
      if ((tries & 1023) == 1023 && wp_lost(tries, t0, x.lost)) break;
      cur = nxt;
    }
  }
  tf_mark(a.trace, TS_GRU, 1, pick);
  unsigned long long g[PW][NT][4];
  {
    unsigned long long t0 = 0;
    for (int tries = 0;; ++tries) {
      unsigned stale = 0u;
#pragma unroll
      for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const unsigned long long* gp = x.p2g + ((size_t)((wave + 8 * p) * a.nta + ntc[nt]) * 4) * 64 + lane;
#pragma unroll
          for (int r = 0; r < 4; ++r) g[p][nt][r] = wp_get(gp + r * 64);
        }
#pragma unroll
      for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) stale |= (unsigned)(g[p][nt][r] >> 32) ^ tag;
      if (stale == 0u) break;
      if ((tries & 1023) == 1023 && wp_lost(tries, t0, x.lost)) break;
      __builtin_amdgcn_s_sleep(1);
    }
  }
  tf_mark(a.trace, TS_GRU, 2, pick);
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < PW; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float av = c == 0 ? wa[p].x : c == 1 ? wa[p].y : c == 2 ? wa[p].z : wa[p].w;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, __uint_as_float((unsigned)g[p][nt][c]), acc[nt], 0, 0, 0);
    }
  float4* red4 = reinterpret_cast<float4*>(red);  // [8][NT][2][64]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    red4[((wave * NT + nt) * 2 + 0) * 64 + lane] = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
    if (own_h) red4[((wave * NT + nt) * 2 + 1) * 64 + lane] = make_float4(accH[nt][0], accH[nt][1], accH[nt][2], accH[nt][3]);
  }
  __syncthreads();
  tf_mark(a.trace, TS_GRU, 3, pick);
  if (wave >= NT || wave >= a.nta) return;
  float sx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const float4 v = red4[((w8 * NT + wave) * 2 + 0) * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
  }
  if (own_h) {
    float sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) {
      const float4 v = red4[((w8 * NT + wave) * 2 + 1) * 64 + lane];
      sh[0] += v.x; sh[1] += v.y; sh[2] += v.z; sh[3] += v.w;
    }
    hp = make_float4(sh[0] + hp.x, sh[1] + hp.y, sh[2] + hp.z, 0.f);
  }
  // torch GRUCell, gate order (r, z, n)  (taco_gru_kernel's epilogue)
  const float rg = tf_sigmoid((sx[0] + xp.x) + hp.x);
  const float zg = tf_sigmoid((sx[1] + xp.y) + hp.y);
  const float ng = tf_tanh((sx[2] + xp.z) + rg * hp.z);
  const float hn = ng + zg * (hprev - ng);
  wp_put(x.ahg + (size_t)(wave * 16 + i) * 128 + mt * 4 + du, hn, tag);  // the attention workgroups wait for this one
  *hpt = hn;                                                               // rnn_input / next iteration: after the launch
  tf_mark_end(a.trace, TS_GRU, 4, pick);
}
template <int TJ, int NT, bool F16, bool FOLD>
__global__ __launch_bounds__(512) void taco_front_kernel(TfFcK fk, TfGruK gk, LsaK a, TfHhK hh, TfHhK hhb, TfFrontX x, int n_lsa, int B, int gy, int nta) {
  __shared__ __attribute__((aligned(16))) float s_big[TJ == 32 ? 32 * 4 * 256 : 2 * FmRed<NT, 1>::floats];
  const int id = blockIdx.x;
  if (id < x.n_fc2) { front_fc2_job<NT>(fk, x, id, s_big); return; }
  if (id < x.n_fc2 + x.n_gru) { front_gru_job<NT>(gk, x, id - x.n_fc2, fk.flags[TF_ITER] + fk.it_off, s_big); return; }
  const int l = id - x.n_fc2 - x.n_gru;
  if (l < n_lsa) {
    lsa_fast_body<TJ, true, FOLD>(a, l % B, l / B, TJ == 32 ? s_big : nullptr);
    if (a.trace && threadIdx.x == 0) {
      atomicMax(a.trace + TS_LSA * 16 + 13, (unsigned long long)wall_clock64());  // last attention workgroup
      if (l < 128) a.trace[TS_WG + 2 * l + 1] = (unsigned long long)wall_clock64();
    }
    return;
  }
  const int j = l - n_lsa;  // (gy == 1: the fused launch serves at most NT column tiles)
  if (x.hh_pairs) fm_hh_pair_job<NT>(hh, j, nta, a.skip_flag ? *a.skip_flag : 0, s_big);
  else if (j < hh.n_tiles) fm_hh_job<NT, F16>(hh, j, 0, nta, a.skip_flag ? *a.skip_flag : 0, s_big, x.lost);
  else fm_hh_job<NT, F16>(hhb, j - hh.n_tiles, 0, nta, a.skip_flag ? *a.skip_flag : 0, s_big, x.lost);  // folded form: the W_hh1 tiles the mel launch has no room for
  if (a.trace && threadIdx.x == 0) {
    atomicMax(a.trace + TS_FC2 * 16 + 13, (unsigned long long)wall_clock64());  // last hh2 tile
    if (j == 0) a.trace[TS_FC2 * 16 + 12] = (unsigned long long)wall_clock64();  // first hh2 tile done
  }
}

// ---------------------------------------------------------------- finalize
struct FinK {
  const float* x2;       // [B][H]
  const float* context;  // [B][P]
  const float* stop_w;   // [H + P]
  const float* stop_b;   // [1]
  const float* melstep;  // [B][r*M] frame-major
  float* mel_out;        // [B][M][max_steps]
  float* stop_out;       // [B] scratch
  int* done;             // skip flag for later launches
  int* n_frames;         // frames produced so far
  int* arrive;           // block arrival counter (zeroed per iteration by the last block)
  int B, H, P, M, r, max_steps, t0;
  float min_stop_token;
};

// One workgroup per utterance (a single CU cannot pull the whole batch's 256 KB of fresh state fast
// enough: that variant measured 14.5 us).  The batch-wide stop rule is decided by the last workgroup to
// arrive, through two device-scope atomics -- a count of utterances still below the threshold, then the
// arrival ticket -- with no __threadfence(): nothing but those atomics is exchanged (the fenced
// "last block" protocol this replaces spent ~7 of its 9.9 us in the fences).
__global__ __launch_bounds__(256) void finalize_kernel(FinK a) {
  if (*a.done) return;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float red[4];
  // stop = sigmoid(stop_proj([x, context]))  (tacotron.py:133-136)
  float acc = 0.f;
  for (int k = tid; k < a.H; k += 256) acc += a.stop_w[k] * a.x2[(size_t)b * a.H + k];
  for (int k = tid; k < a.P; k += 256) acc += a.stop_w[a.H + k] * a.context[(size_t)b * a.P + k];
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  // mel frames of this iteration -> mel_out[b][m][t0 + j]
  for (int i = tid; i < a.r * a.M; i += 256) {
    const int j = i / a.M, m = i - j * a.M;
    if (a.t0 + j < a.max_steps) a.mel_out[((size_t)b * a.M + m) * a.max_steps + a.t0 + j] = a.melstep[(size_t)b * a.r * a.M + i];
  }
  __syncthreads();
  if (tid == 0) {
    const float sgm = 1.0f / (1.0f + expf(-((red[0] + red[1]) + (red[2] + red[3]) + a.stop_b[0])));
    a.stop_out[b] = sgm;
    // batch-wide stop rule  (stop*10 > min_stop_token).all() and t > 10  (tacotron.py:275)
    if (!(sgm * 10.f > a.min_stop_token)) {
      int old = atomicAdd(a.arrive + 1, 1);  // utterances not ready to stop; performed before the ticket below
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(old) : : "memory");
    }
    if (atomicAdd(a.arrive, 1) == a.B - 1) {  // last arrival
      const int not_ready = __hip_atomic_load(a.arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *a.n_frames = min(a.t0 + a.r, a.max_steps);
      if (not_ready == 0 && a.t0 > 10) *a.done = 1;
      a.arrive[0] = 0; a.arrive[1] = 0;  // visible to the next iteration's launch (kernel boundary)
    }
  }
}

}  // namespace mb

using namespace mb;

namespace {
struct ConvL {  // conv (+ folded/attached BN) for conv1d.hip
  int c_out = 0, c_in = 0, k = 1, pad = 0;
  DevBuf w, b, ps, pt;  // packed weight, bias, post scale/shift (BN after ReLU)
  void release() { w.release(); b.release(); ps.release(); pt.release(); }
};

// BatchNormConv (conv -> [relu] -> BN).  With relu the BN stays a separate scale/shift applied after
// the activation; without relu it is folded into the conv (batch_norm_conv.py:11-14).
int make_bnconv(ConvL* c, const float* w, int c_out, int c_in, int k, const float* const* bn, bool relu) {
  c->c_out = c_out; c->c_in = c_in; c->k = k; c->pad = k / 2;
  std::vector<float> scale(c_out), shift(c_out);
  for (int co = 0; co < c_out; ++co) {
    const double sc = (double)bn[0][co] / std::sqrt((double)bn[3][co] + 1e-5);
    scale[co] = (float)sc;
    shift[co] = (float)((double)bn[1][co] - (double)bn[2][co] * sc);
  }
  std::vector<float> wf(w, w + (size_t)c_out * c_in * k);
  int rc = MB_OK;
  if (!relu) {
    for (int co = 0; co < c_out; ++co)
      for (size_t i = 0; i < (size_t)c_in * k; ++i) wf[(size_t)co * c_in * k + i] *= scale[co];
    rc = c->b.upload(shift.data(), c_out);
  } else {
    rc = c->ps.upload(scale.data(), c_out);
    if (!rc) rc = c->pt.upload(shift.data(), c_out);
  }
  std::vector<float> packed(mb_conv1d_packed_floats(c_out, c_in, k, 1));
  if (!rc) rc = mb_conv1d_pack(wf.data(), c_out, c_in, k, 1, 0, c->pad, packed.data());
  if (!rc) rc = c->w.upload(packed.data(), packed.size());
  return rc;
}

int make_linear_conv(ConvL* c, const float* w, int c_out, int c_in, const float* bias) {
  c->c_out = c_out; c->c_in = c_in; c->k = 1; c->pad = 0;
  std::vector<float> packed(mb_conv1d_packed_floats(c_out, c_in, 1, 1));
  int rc = mb_conv1d_pack(w, c_out, c_in, 1, 1, 0, 0, packed.data());
  if (!rc) rc = c->w.upload(packed.data(), packed.size());
  if (!rc && bias) rc = c->b.upload(bias, c_out);
  return rc;
}

// One conv as a time-major split conv (conv_split_tm.hip): round 6, the CBHG postnet on [B][F][C] tensors
struct TmL {
  DevBuf w, b, ps, pt;
  float us = 0.f;
  int c_in = 0, m = 0, k = 1, pad = 0;
  void release() { w.release(); b.release(); ps.release(); pt.release(); }
};
int make_tml(TmL* t, const float* w_eff, int m, int c_in, int k, int pad, const float* bias, const float* ps, const float* pt) {
  t->c_in = c_in; t->m = m; t->k = k; t->pad = pad;
  if (!mb_conv_split_tm_supported(m, c_in, k, 1)) return 1;
  std::vector<float> img(mb_conv_split_tm_packed_halves(m, c_in, k) / 2, 0.f);
  int rc = mb_conv_split_tm_pack(w_eff, m, c_in, k, reinterpret_cast<uint16_t*>(img.data()), &t->us);
  if (!rc) rc = t->w.upload(img.data(), img.size());
  if (!rc && bias) rc = t->b.upload(bias, m);
  if (!rc && ps) rc = t->ps.upload(ps, m);
  if (!rc && pt) rc = t->pt.upload(pt, m);
  return rc;
}
struct CbhgTm {
  bool ok = false;
  TmL bank, proj1, proj2, pre, ih_f, ih_b;
  std::vector<TmL> hwf;  // per highway layer ONE conv to 2 ch channels: (W1 | W2) stacked, the combine is mb_highway_tm
  void release() {
    bank.release(); proj1.release(); proj2.release(); pre.release(); ih_f.release(); ih_b.release();
    for (auto& c : hwf) c.release();
    ok = false;
  }
};

// CBHG (sublayer/cbhg.py:6-84): conv bank K -> maxpool -> 2 projections -> +residual -> [pre_highway]
// -> highways -> bidirectional GRU.  Shared by the postnet (in 80, ch 512) and the text encoder
// (in 256, ch 256).  Activations stay channel-major [B][C][T] end to end.
struct Cbhg {
  int cin = 0, ch = 0, K = 0, nh = 0, proj_out = 0;
  std::vector<ConvL> bank, hw1, hw2;
  ConvL proj1, proj2, pre_highway, gru_ih_f, gru_ih_b;
  DevBuf gru_hh_f, gru_hh_b, gru_bhh_f, gru_bhh_b;
  DevBuf gru_raw_f, gru_raw_b;  // torch weight_hh as is: the resident scan (gru_scan.h) splits it into its registers
  int gru_sexp[2] = {0, 0};     // 2^s of that split, per direction
  bool has_pre = false;
  CbhgTm tm;                    // the same convs as time-major split convs (make_cbhg_tm; the postnet)
  void release() {
    tm.release();
    for (auto& c : bank) c.release();
    for (auto& c : hw1) c.release();
    for (auto& c : hw2) c.release();
    proj1.release(); proj2.release(); pre_highway.release(); gru_ih_f.release(); gru_ih_b.release();
    gru_hh_f.release(); gru_hh_b.release(); gru_bhh_f.release(); gru_bhh_b.release();
    gru_raw_f.release(); gru_raw_b.release();
  }
};

void cbhg_shapes(std::vector<size_t>* n, size_t cin, size_t ch, size_t p0, size_t p1, int K, int nh) {
  auto bn = [&](size_t k) { for (int i = 0; i < 4; ++i) n->push_back(k); };
  for (int k = 1; k <= K; ++k) { n->push_back(ch * cin * k); bn(ch); }
  n->push_back(p0 * (ch * K) * 3); bn(p0);
  n->push_back(p1 * p0 * 3); bn(p1);
  if (p1 != ch) n->push_back(ch * p1);  // pre_highway (cbhg.py:26-30)
  for (int i = 0; i < nh; ++i) { n->push_back(ch * ch); n->push_back(ch); n->push_back(ch * ch); n->push_back(ch); }
  for (int d = 0; d < 2; ++d) { n->push_back(3 * (ch / 2) * ch); n->push_back(3 * (ch / 2) * (ch / 2)); n->push_back(3 * (ch / 2)); n->push_back(3 * (ch / 2)); }
}

int make_cbhg(Cbhg* c, const float* const* hw, int* pix, int cin, int ch, int p0, int p1, int K, int nh) {
  int ix = *pix, rc = MB_OK;
  c->cin = cin; c->ch = ch; c->K = K; c->nh = nh; c->proj_out = p1;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  c->bank.resize(K);
  for (int k = 1; k <= K; ++k) { RC(make_bnconv(&c->bank[k - 1], hw[ix], ch, cin, k, hw + ix + 1, true)); ix += 5; }
  RC(make_bnconv(&c->proj1, hw[ix], p0, ch * K, 3, hw + ix + 1, true)); ix += 5;
  RC(make_bnconv(&c->proj2, hw[ix], p1, p0, 3, hw + ix + 1, false)); ix += 5;
  c->has_pre = p1 != ch;
  if (c->has_pre) { RC(make_linear_conv(&c->pre_highway, hw[ix], ch, p1, nullptr)); ix += 1; }
  c->hw1.resize(nh); c->hw2.resize(nh);
  for (int i = 0; i < nh; ++i) {
    RC(make_linear_conv(&c->hw1[i], hw[ix], ch, ch, hw[ix + 1]));
    RC(make_linear_conv(&c->hw2[i], hw[ix + 2], ch, ch, hw[ix + 3]));
    ix += 4;
  }
  const int Hg = ch / 2;
  std::vector<float> rows, packed;
  for (int d = 0; d < 2; ++d) {
    RC(make_linear_conv(d ? &c->gru_ih_b : &c->gru_ih_f, hw[ix], 3 * Hg, ch, hw[ix + 2]));  // W_ih.x + b_ih for all t
    cell_rows(hw[ix + 1], 0, 0, hw[ix + 1], Hg, Hg, 3, &rows);                                // hidden part only
    pack_rowtile(rows.data(), 3 * Hg, Hg, 3, &packed);
    RC((d ? c->gru_hh_b : c->gru_hh_f).upload(packed.data(), packed.size()));
    RC((d ? c->gru_bhh_b : c->gru_bhh_f).upload(hw[ix + 3], 3 * Hg));
    RC((d ? c->gru_raw_b : c->gru_raw_f).upload(hw[ix + 1], (size_t)3 * Hg * Hg));
    c->gru_sexp[d] = gru_scan_scale_exp(hw[ix + 1], (size_t)3 * Hg * Hg);
    ix += 4;
  }
#undef RC
  *pix = ix;
  return rc;
}

// The time-major images of a CBHG's convs (same weight list as make_cbhg).  The conv bank's K convs (kernel sizes 1..K, padding k/2,
// output cut to T: cbhg.py:53-59) become ONE conv to K ch channels over KT = K | 1 centred taps: conv k's tap j sits at j - k/2 + KT/2.
int make_cbhg_tm(Cbhg* c, const float* const* hw, int ix, int cin, int ch, int p0, int p1, int K, int nh) {
  CbhgTm& m = c->tm;
  int rc = MB_OK;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  auto bn_fold = [&](const float* const* bn, int n, std::vector<float>* scale, std::vector<float>* shift, size_t at) {
    for (int co = 0; co < n; ++co) {
      const double sc = (double)bn[0][co] / std::sqrt((double)bn[3][co] + 1e-5);
      (*scale)[at + co] = (float)sc;
      (*shift)[at + co] = (float)((double)bn[1][co] - (double)bn[2][co] * sc);
    }
  };
  const int KT = K | 1, P = KT / 2;
  {
    std::vector<float> w((size_t)K * ch * cin * KT, 0.f), sc((size_t)K * ch), sh((size_t)K * ch);
    for (int k = 1; k <= K; ++k) {
      const float* wk = hw[ix];
      for (int co = 0; co < ch; ++co)
        for (int ci = 0; ci < cin; ++ci)
          for (int j = 0; j < k; ++j) w[(((size_t)(k - 1) * ch + co) * cin + ci) * KT + (j - k / 2 + P)] = wk[((size_t)co * cin + ci) * k + j];
      bn_fold(hw + ix + 1, ch, &sc, &sh, (size_t)(k - 1) * ch);
      ix += 5;
    }
    RC(make_tml(&m.bank, w.data(), K * ch, cin, KT, P, nullptr, sc.data(), sh.data()));
  }
  {
    std::vector<float> sc(p0), sh(p0);
    bn_fold(hw + ix + 1, p0, &sc, &sh, 0);
    RC(make_tml(&m.proj1, hw[ix], p0, ch * K, 3, 1, nullptr, sc.data(), sh.data()));
    ix += 5;
  }
  {
    std::vector<float> sc(p1), sh(p1), wf(hw[ix], hw[ix] + (size_t)p1 * p0 * 3);
    bn_fold(hw + ix + 1, p1, &sc, &sh, 0);
    for (int co = 0; co < p1; ++co)
      for (size_t i = 0; i < (size_t)p0 * 3; ++i) wf[(size_t)co * p0 * 3 + i] *= sc[co];
    RC(make_tml(&m.proj2, wf.data(), p1, p0, 3, 1, sh.data(), nullptr, nullptr));
    ix += 5;
  }
  if (p1 != ch) { RC(make_tml(&m.pre, hw[ix], ch, p1, 1, 0, nullptr, nullptr, nullptr)); ix += 1; }
  m.hwf.resize(nh);
  for (int i = 0; i < nh; ++i) {
    std::vector<float> w2((size_t)2 * ch * ch), b2((size_t)2 * ch);
    memcpy(w2.data(), hw[ix], (size_t)ch * ch * sizeof(float)); memcpy(w2.data() + (size_t)ch * ch, hw[ix + 2], (size_t)ch * ch * sizeof(float));
    memcpy(b2.data(), hw[ix + 1], ch * sizeof(float)); memcpy(b2.data() + ch, hw[ix + 3], ch * sizeof(float));
    RC(make_tml(&m.hwf[i], w2.data(), 2 * ch, ch, 1, 0, b2.data(), nullptr, nullptr));
    ix += 4;
  }
  if (!rc && (ch % 8 || p0 % 8 || (ch * K) % 8 || ch <= 32 || p0 <= 32)) rc = 1;  // the split tensors between the launches need wide rows
  const int Hg = ch / 2;
  RC(make_tml(&m.ih_f, hw[ix], 3 * Hg, ch, 1, 0, hw[ix + 2], nullptr, nullptr)); ix += 4;
  RC(make_tml(&m.ih_b, hw[ix], 3 * Hg, ch, 1, 0, hw[ix + 2], nullptr, nullptr)); ix += 4;
#undef RC
  if (rc > 0) { m.release(); return MB_OK; }  // a shape conv_split_tm has no instance for: the channel-major convs run
  m.ok = rc == MB_OK;
  return rc;
}

struct CbhgWs { float *bank, *pj1, *pj2, *hwa, *hwb, *gate, *ihf, *ihb, *gh, *seq, *seq_tm, *xt, *bank2, *sp1, *spa, *spb, *hg; unsigned long long* gsx; };
constexpr size_t GSX_WORDS = (size_t)2 * 2 * 32 * 256 + 32;  // granules of the resident GRU scan (gru_scan.h) + its abort word

void cbhg_take(Arena& ar, const Cbhg& c, size_t B, size_t F, CbhgWs* w) {
  const size_t ch = c.ch;
  w->bank = ar.take<float>(B * ch * c.K * F);
  w->pj1 = ar.take<float>(B * std::max<size_t>(ch, c.proj1.c_out) * F);
  w->pj2 = ar.take<float>(B * std::max<size_t>(c.proj_out, 1) * F);
  w->hwa = ar.take<float>(B * ch * F); w->hwb = ar.take<float>(B * ch * F); w->gate = ar.take<float>(B * ch * F);
  w->ihf = ar.take<float>(B * F * 3 * (ch / 2)); w->ihb = ar.take<float>(B * F * 3 * (ch / 2));
  w->gh = ar.take<float>(4 * B * (ch / 2));
  w->seq = ar.take<float>(B * ch * F);
  w->gsx = ar.take<unsigned long long>(GSX_WORDS);
  w->seq_tm = ar.take<float>(B * ch * F);
  w->xt = w->bank2 = w->sp1 = w->spa = w->spb = w->hg = nullptr;
  if (c.tm.ok) {  // the time-major front: the turned input, the pooled conv bank / projection / highway outputs as split tensors
    w->xt = ar.take<float>(B * (size_t)c.cin * F);
    w->bank2 = ar.take<float>(B * ch * c.K * F);   // (a split tensor has the bytes of the fp32 one)
    w->sp1 = ar.take<float>(B * std::max<size_t>(ch, c.proj1.c_out) * F);
    w->spa = ar.take<float>(B * ch * F); w->spb = ar.take<float>(B * ch * F);
    w->hg = ar.take<float>(B * 2 * ch * F);
  }
}

int run_conv(const ConvL& c, const float* x, int batch, int t, float* y, long long y_bstride, int in_act,
             int out_act, const float* res, const float* gate, int transpose_out, hipStream_t s);

int cbhg_scan(const Cbhg& c, int B, int F, const CbhgWs& L, hipStream_t s, bool want_cm, bool* tm_valid);

// one conv_split_tm launch of the time-major front (t rows per item)
static int run_tml(const TmL& c, const float* x, int B, int t, float* y, int out_act, const float* res, const float* gate, hipStream_t s,
                   long long x_bstride = 0, int x_row_stride = 0, bool x_split = false, float* ysplit = nullptr) {
  mb_conv_split_tm_args a;
  memset(&a, 0, sizeof(a));
  a.d_x = x; a.d_y = y; a.d_wpacked = c.w.p; a.d_bias = c.b.p; a.d_res = res; a.d_gate = gate;
  a.d_post_scale = c.ps.p; a.d_post_shift = c.pt.p;
  a.batch = B; a.t = t; a.c_in = c.c_in; a.c_out = c.m; a.ksize = c.k; a.dilation = 1; a.pad = c.pad;
  a.in_slope = 1.f; a.unscale = c.us; a.out_scale = 1.f; a.out_act = out_act;
  a.x_bstride = x_bstride; a.x_row_stride = x_row_stride; a.x_split = x_split ? 1 : 0; a.d_ysplit = ysplit;
  return mb_conv_split_tm(&a, (mb_stream_t)s);
}

// Round 6: the CBHG on time-major tensors -- x [B][cin][F] is turned once, the conv bank is ONE launch, every
// conv a conv_split_tm launch (BatchNorm / ReLU / residual / highway in its write-out), the GRU tables come out time-major as the
// scan reads them.  -> L.seq_tm [F][B][ch] (*tm_valid) or, when the resident scan is not available, L.seq [B][ch][F].
int cbhg_forward_tm(const Cbhg& c, const float* x, int B, int F, const CbhgWs& L, hipStream_t s, bool* tm_valid, bool want_cm = false) {
  int rc = MB_OK;
  const int C = c.ch;
  const CbhgTm& m = c.tm;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  // Tensors a later conv multiplies are ALSO written as split tensors (fp16 hi | scaled lo rows: the bytes of the fp32 tensor) by their
  // producer, so that the consumer stages them by copy: a 512 -> 512 pointwise conv has 36 MFMAs per wave between two chunk
  // barriers, and the support waves' fp32 -> split conversion of the chunk (every element once per channel group and consumer)
  // took four times that (first form of this function: 152 us per highway conv, the whole front no faster than channel-major).
  RC(mb_f32_cm_to_tm(x, L.xt, B, c.cin, F, (mb_stream_t)s));
  RC(run_tml(m.bank, L.xt, B, F, L.bank, 1, nullptr, nullptr, s));                       // conv -> ReLU -> BN, all K kernels (cbhg.py:53-59)
  RC(mb_maxpool2_tm(L.bank, nullptr, L.bank2, B, F, C * c.K, (mb_stream_t)s));           // maxpool(2,1,1)[:F] (cbhg.py:61-62) -> split
  RC(run_tml(m.proj1, L.bank2, B, F, L.pj1, 1, nullptr, nullptr, s, 0, 0, true, L.sp1)); // conv_project1 -> ReLU -> BN (cbhg.py:65)
  RC(run_tml(m.proj2, L.sp1, B, F, L.pj2, 0, L.xt, nullptr, s, 0, 0, true, c.has_pre ? nullptr : L.spa));  // conv_project2 (BN folded) + residual (cbhg.py:66-69)
  const float* hin = L.pj2;
  float* scur = L.spa;
  if (c.has_pre) { RC(run_tml(m.pre, L.pj2, B, F, L.hwa, 0, nullptr, nullptr, s, 0, 0, false, L.spa)); hin = L.hwa; }
  for (int i = 0; i < c.nh; ++i) {  // highway_network.py:12-17: (W1 x + b1 | W2 x + b2) in one launch, then the combine
    float* dst = (hin == L.hwa) ? L.hwb : L.hwa;
    float* snext = (scur == L.spa) ? L.spb : L.spa;
    RC(run_tml(m.hwf[i], scur, B, F, L.hg, 0, nullptr, nullptr, s, 0, 0, true, nullptr));
    RC(mb_highway_tm(L.hg, hin, dst, snext, (long long)B * F, C, (mb_stream_t)s));
    hin = dst; scur = snext;
  }
  RC(run_tml(m.ih_f, scur, B, F, L.ihf, 0, nullptr, nullptr, s, 0, 0, true, nullptr)); // W_ih x + b_ih for every t (cbhg.py:76-77)
  RC(run_tml(m.ih_b, scur, B, F, L.ihb, 0, nullptr, nullptr, s, 0, 0, true, nullptr));
#undef RC
  if (rc) return rc;
  return cbhg_scan(c, B, F, L, s, want_cm, tm_valid);
}

// x [B][cin][F] -> ws.seq [B][ch][F] (forward half in channels [0, ch/2), backward in [ch/2, ch))
int cbhg_forward(const Cbhg& c, const float* x, int B, int F, const CbhgWs& L, hipStream_t s) {
  int rc = MB_OK;
  const int C = c.ch, Hg = C / 2;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  const long long bank_bs = (long long)C * c.K * F;
  for (int k = 0; k < c.K; ++k)  // conv -> ReLU -> BN, concatenated on the channel axis (cbhg.py:53-59)
    RC(run_conv(c.bank[k], x, B, F, L.bank + (size_t)k * C * F, bank_bs, 0, 1, nullptr, nullptr, 0, s));
  {  // maxpool(2,1,1)[:F] (cbhg.py:61-62) fused into conv_project1's input staging
    mb_conv1d_args a;
    memset(&a, 0, sizeof(a));
    const ConvL& cv = c.proj1;
    a.d_x = L.bank; a.d_wpacked = cv.w.p; a.d_y = L.pj1; a.d_post_scale = cv.ps.p; a.d_post_shift = cv.pt.p;
    a.x_bstride = bank_bs; a.y_bstride = (long long)cv.c_out * F; a.batch = B; a.c_in = cv.c_in; a.c_out = cv.c_out;
    a.t_in = F; a.t_out = F; a.ksize = 3; a.dilation = 1; a.pad = 1; a.up = 1; a.in_act = 2; a.out_act = 1;
    RC(mb_conv1d(&a, (mb_stream_t)s));
  }
  RC(run_conv(c.proj2, L.pj1, B, F, L.pj2, 0, 0, 0, x, nullptr, 0, s));  // BN folded, + residual (cbhg.py:66-69)
  float* hx = L.hwa; float* hy = L.hwb;
  const float* hin = L.pj2;
  if (c.has_pre) { RC(run_conv(c.pre_highway, L.pj2, B, F, L.hwa, 0, 0, 0, nullptr, nullptr, 0, s)); hin = L.hwa; hx = L.hwa; hy = L.hwb; }
  for (int i = 0; i < c.nh; ++i) {  // highway_network.py:12-17
    float* dst = (hin == L.hwa) ? L.hwb : L.hwa;
    RC(run_conv(c.hw2[i], hin, B, F, L.gate, 0, 0, 3, nullptr, nullptr, 0, s));   // g = sigmoid(W2 x)
    RC(run_conv(c.hw1[i], hin, B, F, dst, 0, 0, 4, hin, L.gate, 0, s));           // g*relu(W1 x) + (1-g)*x
    hin = dst;
  }
  (void)hx; (void)hy;
  // bidirectional GRU (cbhg.py:76-77): W_ih.x + b_ih for every t as one GEMM per direction (time-major table)
  RC(run_conv(c.gru_ih_f, hin, B, F, L.ihf, (long long)F * 3 * Hg, 0, 0, nullptr, nullptr, 1, s));
  RC(run_conv(c.gru_ih_b, hin, B, F, L.ihb, (long long)F * 3 * Hg, 0, 0, nullptr, nullptr, 1, s));
#undef RC
  if (rc) return rc;
  return cbhg_scan(c, B, F, L, s, true, nullptr);
}

// The bidirectional GRU scan over the [B][F][3 Hg] tables L.ihf / L.ihb -> L.seq_tm [F][B][ch] (resident launch; turned into L.seq
// [B][ch][F] when want_cm) or, on the launch-per-step fallback, L.seq only.  *tm_valid (may be null) = L.seq_tm holds the sequence.
int cbhg_scan(const Cbhg& c, int B, int F, const CbhgWs& L, hipStream_t s, bool want_cm, bool* tm_valid) {
  int rc = MB_OK;
  const int C = c.ch, Hg = C / 2;
  if (tm_valid) *tm_valid = false;
  // the scan: one resident launch for both directions (gru_scan.h); a lost hand-off (never seen: 4..8 workgroups) or
  // MBHIP_GRU_SCAN=0 or a shape it has no instance for -> one launch per step
  bool scanned = false;
  {
    const char* e = getenv("MBHIP_GRU_SCAN");
    const bool forced = e && atoi(e) == 1;  // MBHIP_GRU_SCAN=1: try the resident launch even on a device remembered as unsuitable
    if (!rc && !(e && atoi(e) == 0) && gru_scan_shape_ok(B, Hg) && c.gru_raw_f.p && c.gru_raw_b.p && (forced || !gru_scan_device_failed())) {
      GruScanK k;
      memset(&k, 0, sizeof(k));
      k.whh[0] = c.gru_raw_f.p; k.whh[1] = c.gru_raw_b.p; k.bhh[0] = c.gru_bhh_f.p; k.bhh[1] = c.gru_bhh_b.p;
      k.ih[0] = L.ihf; k.ih[1] = L.ihb; k.seq_tm = L.seq_tm; k.ex = L.gsx; k.abort_word = reinterpret_cast<int*>(L.gsx + GSX_WORDS - 32);
      k.B = B; k.F = F; k.Hg = Hg;
      for (int d = 0; d < 2; ++d) {
        k.wscale[d] = std::ldexp(1.f, c.gru_sexp[d]);
        k.unscale[d] = std::ldexp(1.f, -c.gru_sexp[d] - 10);
      }
      k.dbg = diag_int("gs_dbg", 0);
      MB_HIP(hipMemsetAsync(L.gsx, 0, GSX_WORDS * sizeof(unsigned long long), s));
      const bool test_abort = diag_int("abort_gru_scan") != 0;  // tests (MBHIP_DIAG=abort_gru_scan): the fallback path
      if (test_abort) {
        const int one = 1;
        MB_HIP(hipMemcpyAsync(k.abort_word, &one, sizeof(int), hipMemcpyHostToDevice, s));
      }
      int lrc = gru_scan_launch(k, s);  // a launch that fails (e.g. the LDS attribute on an odd device) is not an error of the
      if (!lrc) {                        // encode: the launch-per-step scan below computes the same sequence
        if (want_cm) {
          hipLaunchKernelGGL(gru_scan_transpose_kernel, dim3(cdiv(C, 32), cdiv(F, 32), B), dim3(256), 0, s, L.seq_tm, L.seq, F, B, C);
          MB_HIP(hipGetLastError());
        }
      } else (void)hipGetLastError();
      int aborted = 0;
      if (!lrc) {
        MB_HIP(hipMemcpyAsync(&aborted, k.abort_word, sizeof(int), hipMemcpyDeviceToHost, s));
        MB_HIP(hipStreamSynchronize(s));
      }
      scanned = !lrc && !aborted;
      if (scanned && tm_valid) *tm_valid = true;
      if (!scanned && !test_abort) gru_scan_mark_failed();
    }
  }
  if (!rc && !scanned) {
    MB_HIP(hipMemsetAsync(L.gh, 0, sizeof(float) * 4 * B * Hg, s));
    for (int st = 0; st < F && !rc; ++st) {
      RnnK kd[2];
      for (int d = 0; d < 2; ++d) {
        const int tt = d ? F - 1 - st : st;
        float* hp = L.gh + ((size_t)d * 2 + (st & 1)) * B * Hg;
        float* hn = L.gh + ((size_t)d * 2 + ((st & 1) ^ 1)) * B * Hg;
        RnnK& k = kd[d];
        memset(&k, 0, sizeof(k));
        k.w = d ? c.gru_hh_b.p : c.gru_hh_f.p; k.nseg = 1; k.nkb_total = Hg / 16; k.seg[0] = {hp, Hg, Hg / 16, 1};
        k.N = B; k.units = Hg; k.biasH = d ? c.gru_bhh_b.p : c.gru_bhh_f.p;
        k.pre_table = d ? L.ihb : L.ihf; k.pre_stride = 3 * Hg; k.pre_base_row = tt; k.pre_n_stride = F;
        k.h_prev = hp; k.h_out = hn;
        k.seq_out = L.seq; k.seq_n_stride = (long long)C * F; k.seq_j_stride = F; k.seq_off = (long long)d * Hg * F + tt;
      }
      rc = rnn_launch_dual_gru(kd[0], kd[1], s);  // both directions of step st in one launch
    }
  }
  return rc;
}

// ---- text-encoder helpers (tacotron.py:31-44, 171-197, 255) ----
// xe[b][c][t] = embedding[chars[b][t]][c]   (channel-major for the conv kernel)
__global__ void embed_gather_kernel(const int* __restrict__ chars, const float* __restrict__ emb, float* __restrict__ xe,
                                    int T, int E, int num_chars) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E * T; i += gridDim.x * blockDim.x) {
    const int c = i / T, t = i - c * T;
    int id = chars[(size_t)b * T + t];
    id = id < 0 ? 0 : (id >= num_chars ? num_chars - 1 : id);
    xe[(size_t)b * E * T + i] = emb[(size_t)id * E + c];
  }
}

// always-on PreNet dropout (pre_net.py:23,26) on y [B][C][T]; masks (if injected) are [B][T][C]
__global__ void dropout_cm_kernel(float* __restrict__ y, const float* __restrict__ mask, int C, int T,
                                  unsigned long long seed, int layer, unsigned thresh, float scale) {
  const int b = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C * T; i += gridDim.x * blockDim.x) {
    const int c = i / T, t = i - c * T;
    float keep;
    if (mask) keep = mask[((size_t)b * T + t) * C + c];
    else {
      uint32_t r[4];
      philox4x32((uint32_t)(b * T + t), (uint32_t)(c >> 2), (uint32_t)layer, 0x454e4344u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
      keep = (r[c & 3] >= thresh) ? 1.f : 0.f;
    }
    y[(size_t)b * C * T + i] *= keep * scale;
  }
}

// Global style token, inference branches of tacotron.py:243-251 + global_style_token.py:93-145.
//  style_idx in [0, tokens): query = 0 -> one key -> softmax over a single score = 1 -> value row.
//  otherwise: q = W_query [ref_h ; speaker] (ref_h = ReferenceEncoder(zeros), folded into qconst at
//  load), scores_h = q_h . K_h^T / sqrt(d_k), softmax over the tokens, out_h = scores_h V_h.
// K = W_key tanh(embed), V = W_value tanh(embed) are checkpoint constants, WqS = W_query[:, E/2:]^T.
__global__ __launch_bounds__(256) void gst_style_kernel(const float* __restrict__ spk, const float* __restrict__ qconst,
                                                        const float* __restrict__ WqS, const float* __restrict__ Kt,
                                                        const float* __restrict__ Vt, float* __restrict__ out, int S, int E,
                                                        int tokens, int heads, int style_idx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // spk[S] | q[E] | sc[heads*tokens]
  const int b = blockIdx.x, tid = threadIdx.x;
  if (style_idx >= 0 && style_idx < tokens) {
    for (int d = tid; d < E; d += 256) out[(size_t)b * E + d] = Vt[(size_t)style_idx * E + d];
    return;
  }
  float* sp = sm; float* q = sm + S; float* sc = q + E;
  for (int k = tid; k < S; k += 256) sp[k] = spk[(size_t)b * S + k];
  __syncthreads();
  for (int d = tid; d < E; d += 256) {
    float acc = 0.f;
    for (int k = 0; k < S; ++k) acc += WqS[(size_t)k * E + d] * sp[k];
    q[d] = qconst[d] + acc;
  }
  __syncthreads();
  const int dk = E / heads;
  const float inv = 1.0f / sqrtf((float)dk);
  for (int t = tid; t < heads * tokens; t += 256) {
    const int h = t / tokens, j = t - h * tokens;
    float acc = 0.f;
    for (int dd = 0; dd < dk; ++dd) acc += q[h * dk + dd] * Kt[(size_t)j * E + h * dk + dd];
    sc[t] = acc * inv;
  }
  __syncthreads();
  if (tid < heads) {
    float m = -INFINITY;
    for (int j = 0; j < tokens; ++j) m = fmaxf(m, sc[tid * tokens + j]);
    float ssum = 0.f;
    for (int j = 0; j < tokens; ++j) { const float e = expf(sc[tid * tokens + j] - m); sc[tid * tokens + j] = e; ssum += e; }
    for (int j = 0; j < tokens; ++j) sc[tid * tokens + j] /= ssum;
  }
  __syncthreads();
  for (int d = tid; d < E; d += 256) {
    const int h = d / dk;
    float acc = 0.f;
    for (int j = 0; j < tokens; ++j) acc += sc[h * tokens + j] * Vt[(size_t)j * E + d];
    out[(size_t)b * E + d] = acc;
  }
}

// memory[b][t] = [enc_seq[b][:, t] | speaker[b] | style[b]]; projres[b][t][d] = Wp[d][Ce:] . [speaker; style]
__global__ __launch_bounds__(256) void assemble_memory_kernel(const float* __restrict__ seq, const float* __restrict__ spk,
                                                              const float* __restrict__ style, int style_batch,
                                                              const float* __restrict__ Wp, float* __restrict__ memory,
                                                              float* __restrict__ projres, int T, int Ce, int S, int E, int D) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // tail[S+E] | pb[D]
  const int b = blockIdx.x, tid = threadIdx.x, P = Ce + S + E;
  float* tail = sm;
  float* pb = sm + S + E;
  const float* st = style + (size_t)(style_batch > 1 ? b : 0) * E;
  for (int i = tid; i < S; i += 256) tail[i] = spk[(size_t)b * S + i];
  for (int i = tid; i < E; i += 256) tail[S + i] = st[i];
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    const float* wr = Wp + (size_t)d * P + Ce;
    float acc = 0.f;
    for (int k = 0; k < S + E; ++k) acc += wr[k] * tail[k];
    pb[d] = acc;
  }
  __syncthreads();
  for (int i = tid; i < T * P; i += 256) {
    const int t = i / P, p = i - t * P;
    memory[((size_t)b * T + t) * P + p] = p < Ce ? seq[((size_t)b * Ce + p) * T + t] : tail[p - Ce];
  }
  for (int i = tid; i < T * D; i += 256) projres[(size_t)b * T * D + i] = pb[i % D];
}
}  // namespace

struct mb_taco {
  mb_taco_config cfg;
  // decoder
  DevBuf pre1_w, pre1_b, pre2_w, pre2_b;
  DevBuf lsa_conv_w, lsa_conv_b, lsa_L, lsa_W, lsa_Wb, lsa_v;
  DevBuf lsa_Mt, lsa_c0, lsa_Wt;  // folded / transposed copies for lsa_fast_kernel
  DevBuf lsa_Wq4, lsa_Mq4;        // ... as 16-byte rows per thread
  DevBuf attn_w, attn_bih, attn_bhh;
  DevBuf rin_w, rin_b;
  DevBuf l1_w, l1_bih, l1_bhh, l2_w, l2_bih, l2_bhh;
  DevBuf l1_wx, l1_whh, l2_wx, l2_whh;  // split-hidden decoder: input halves (LSTM tile order), hidden halves (plain row tiles)
  DevBuf mel_w, stop_w, stop_b;
  // postnet
  Cbhg post;
  ConvL post_proj;
  TmL post_proj_tm;
  // text encoder (optional: cfg.has_encoder)
  DevBuf emb;
  ConvL enc_fc1, enc_fc2, enc_proj;
  DevBuf enc_proj_full;  // encoder_proj.weight [D][P] (speaker/style columns used by assemble_memory)
  Cbhg enc;
  // global style tokens: checkpoint constants folded at load (see gst_style_kernel)
  DevBuf gst_qconst, gst_WqS, gst_K, gst_V;
  // fast decoder loop (taco_fast.h), packed when the checkpoint has the production dims
  bool fast = false;
  int n_cus = -1; bool front_failed = false; int last_front = 0;  // fused front of the fast loop (taco_front_kernel)
  // split fp16 images (pack_rowtile16) of the fast loop's big tiles, for the 5-launch form's products on the fp16 matrix pipe
  struct Img16 { DevBuf w; float unscale = 1.f; };
  Img16 i_l1x, i_l2x, i_l1hh, i_l2hh, i_rin, i_pre, i_stopc, i_mel, i_fc1, i_stop;
  bool last_f16 = false;
  // folded form of the fast loop (4 launches per iteration): the 1 x 1 conv that projects the attention memory once per decode call
  // (rows: rnn_input's context columns | attention GRU W_ih context columns as unit quads | stop_proj's context columns), and
  // rnn_input's attn_hidden columns in the attention workgroups' lane order
  ConvL pm_conv; DevBuf rin_a4; int last_form = 7;
  DevBuf f_gru_w, f_pre_w, f_bih4, f_bhh4, f_l1_b4, f_l2_b4, f_fc1_w, f_stop_w, f_stopc_w, f_l1_hh, f_l2_hh;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // loop timing (mb_taco_last_loop_ms)
  hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;  // postnet timing (mb_taco_last_postnet_ms)
  mutable bool post_timed = false;
  int last_iters = 0; bool timed = false;
  // captured iterations of the fast loop: reused while the call arguments do not change (handle is single-threaded)
  struct GraphKey {
    const void *mem, *memp, *chars, *drop, *mel, *attn, *ws; int B, T, max_steps, G; float mst; int variant;  // variant: run-time switches
    bool operator==(const GraphKey& o) const {  // field by field: the struct has tail padding, memcmp would read it
      return mem == o.mem && memp == o.memp && chars == o.chars && drop == o.drop && mel == o.mel && attn == o.attn && ws == o.ws &&
             B == o.B && T == o.T && max_steps == o.max_steps && G == o.G && mst == o.mst && variant == o.variant;
    }
  } gkey = {};
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int* h_flags = nullptr;  // pinned: [2][8] flag snapshots
  hipEvent_t ev_flags[2] = {nullptr, nullptr};
  unsigned long long* d_trace = nullptr;  // MBHIP_DIAG=taco_trace=<file>
  hipStream_t loop_stream = nullptr;  // the loop runs (and is captured) on its own stream: the caller's may be the
  hipEvent_t ev_in = nullptr;         // legacy default stream, which cannot be captured
  void drop_graph() {
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
  }
};

// ReferenceEncoder(zeros) -> ref_h, then the style-token constants (double precision on the host:
// this is weight preprocessing like BatchNorm folding, not part of the per-request path).
// global_style_token.py:31-76 (6 x [Conv2d 3x3 s2 p1, BN, ReLU], GRU) on the all-zero input the
// inference branch feeds (tacotron.py:250); :79-145 (STL + multi-head attention).
// Split fp16 image of row tiles for fm_gemm16 (fm_gemm.h): rows as pack_rowtile takes them (tile mt = rows [mt * 4 RL, +4 RL), unit-major),
// K padded to a multiple of 256 with zero columns, values scaled by 2^sexp (chosen so that max |w| 2^s ~ 2^14: both halves fp16 normals).
// out: [tile][g = 8 s + w][hi | lo][64 lanes][8 halves]; lane (i, kq): features 4 kq .. + 3 of k-blocks w + 16 s and w + 16 s + 8.
static int pack_rowtile16(const float* rows, int n_live_rows, int K, int RL, mb_taco::Img16* img) {
  const int per_tile = 4 * RL, n_mt = (n_live_rows + per_tile - 1) / per_tile;
  const int Kp = (K + 255) / 256 * 256, PW2 = Kp / 256;
  float wmax = 0.f;
  for (size_t i = 0; i < (size_t)n_live_rows * K; ++i) wmax = std::max(wmax, std::fabs(rows[i]));
  int e2 = 0, sexp = 0;
  if (wmax > 0.f && std::isfinite(wmax)) { (void)std::frexp(wmax, &e2); sexp = std::max(-24, std::min(40, 14 - e2)); }
  const float scale = std::ldexp(1.f, sexp);
  std::vector<unsigned short> out((size_t)n_mt * PW2 * 8 * 2 * 64 * 8, 0);
  for (int mt = 0; mt < n_mt; ++mt)
    for (int s = 0; s < PW2; ++s)
      for (int w = 0; w < 8; ++w)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, u = i >> 2, tau = i & 3;
          const int row = mt * per_tile + u * RL + tau;
          const size_t base = ((((size_t)mt * PW2 * 8 + (s * 8 + w)) * 2) * 64 + lane) * 8;
          for (int e = 0; e < 8; ++e) {
            const int kb = (e < 4) ? w + 16 * s : w + 16 * s + 8, k = kb * 16 + kq * 4 + (e & 3);
            const float v = (tau < RL && row < n_live_rows && k < K) ? rows[(size_t)row * K + k] * scale : 0.f;
            const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
            memcpy(&out[base + e], &hi, 2);
            memcpy(&out[base + (size_t)64 * 8 + e], &lo, 2);
          }
        }
  img->unscale = std::ldexp(1.f, -sexp);
  return img->w.upload(reinterpret_cast<const float*>(out.data()), out.size() / 2);
}

static int fold_gst(mb_taco* t, const float* const* hw, int* pix) {
  const mb_taco_config& c = t->cfg;
  int ix = *pix;
  const int E = c.style_dims, Eh = E / 2, S = c.speaker_dims, dk = E / c.gst_heads, NTK = c.gst_tokens;
  int Hh = S / c.gst_width, Wd = c.gst_width, fin = 1;
  std::vector<double> x((size_t)Hh * Wd, 0.0), y;
  for (int l = 0; l < c.gst_n_convs; ++l) {
    const int fo = c.gst_filters[l], Ho = (Hh - 1) / 2 + 1, Wo = (Wd - 1) / 2 + 1;
    const float *cw = hw[ix], *cb = hw[ix + 1], *bw = hw[ix + 2], *bb = hw[ix + 3], *bm = hw[ix + 4], *bv = hw[ix + 5];
    ix += 6;
    y.assign((size_t)fo * Ho * Wo, 0.0);
    for (int co = 0; co < fo; ++co) {
      const double sc = (double)bw[co] / std::sqrt((double)bv[co] + 1e-5);
      for (int oy = 0; oy < Ho; ++oy)
        for (int ox = 0; ox < Wo; ++ox) {
          double acc = cb[co];
          for (int ci = 0; ci < fin; ++ci)
            for (int ky = 0; ky < 3; ++ky) {
              const int iy = oy * 2 - 1 + ky;
              if (iy < 0 || iy >= Hh) continue;
              for (int kx = 0; kx < 3; ++kx) {
                const int jx = ox * 2 - 1 + kx;
                if (jx < 0 || jx >= Wd) continue;
                acc += (double)cw[(((size_t)co * fin + ci) * 3 + ky) * 3 + kx] * x[((size_t)ci * Hh + iy) * Wd + jx];
              }
            }
          const double v = (acc - (double)bm[co]) * sc + (double)bb[co];
          y[((size_t)co * Ho + oy) * Wo + ox] = v > 0.0 ? v : 0.0;
        }
    }
    x.swap(y); fin = fo; Hh = Ho; Wd = Wo;
  }
  // [N, C, T', W'] -> [N, T', C*W'] -> GRU(h0 = 0), last hidden state
  const float *wih = hw[ix], *whh = hw[ix + 1], *bih = hw[ix + 2], *bhh = hw[ix + 3];
  ix += 4;
  const int gin = fin * Wd;
  std::vector<double> h(Eh, 0.0), hn(Eh), gi(3 * Eh), gh(3 * Eh);
  for (int tt = 0; tt < Hh; ++tt) {
    for (int r = 0; r < 3 * Eh; ++r) {
      double a = bih[r], b = bhh[r];
      for (int ch = 0; ch < fin; ++ch)
        for (int xx = 0; xx < Wd; ++xx) a += (double)wih[(size_t)r * gin + ch * Wd + xx] * x[((size_t)ch * Hh + tt) * Wd + xx];
      for (int k = 0; k < Eh; ++k) b += (double)whh[(size_t)r * Eh + k] * h[k];
      gi[r] = a; gh[r] = b;
    }
    for (int j = 0; j < Eh; ++j) {
      const double rg = 1.0 / (1.0 + std::exp(-(gi[j] + gh[j])));
      const double zg = 1.0 / (1.0 + std::exp(-(gi[Eh + j] + gh[Eh + j])));
      const double ng = std::tanh(gi[2 * Eh + j] + rg * gh[2 * Eh + j]);
      hn[j] = (1.0 - zg) * ng + zg * h[j];
    }
    h = hn;
  }
  const float *emb = hw[ix], *Wq = hw[ix + 1], *Wk = hw[ix + 2], *Wv = hw[ix + 3];
  ix += 4;
  const int dq = Eh + S;
  std::vector<float> qconst(E), WqS((size_t)S * E), Kt((size_t)NTK * E), Vt((size_t)NTK * E);
  for (int d = 0; d < E; ++d) {
    double a = 0.0;
    for (int k = 0; k < Eh; ++k) a += (double)Wq[(size_t)d * dq + k] * h[k];
    qconst[d] = (float)a;
    for (int k = 0; k < S; ++k) WqS[(size_t)k * E + d] = Wq[(size_t)d * dq + Eh + k];
    for (int j = 0; j < NTK; ++j) {
      double ak = 0.0, av = 0.0;
      for (int q = 0; q < dk; ++q) {
        const double key = std::tanh((double)emb[(size_t)j * dk + q]);
        ak += (double)Wk[(size_t)d * dk + q] * key;
        av += (double)Wv[(size_t)d * dk + q] * key;
      }
      Kt[(size_t)j * E + d] = (float)ak;
      Vt[(size_t)j * E + d] = (float)av;
    }
  }
  int rc = t->gst_qconst.upload(qconst.data(), qconst.size());
  if (!rc) rc = t->gst_WqS.upload(WqS.data(), WqS.size());
  if (!rc) rc = t->gst_K.upload(Kt.data(), Kt.size());
  if (!rc) rc = t->gst_V.upload(Vt.data(), Vt.size());
  *pix = ix;
  return rc;
}

static int taco_shapes(const mb_taco_config* c, std::vector<size_t>* n) {
  MB_REQUIRE(c, "taco: null config");
  MB_REQUIRE(c->n_mels % 16 == 0 && c->project_dims % 16 == 0 && c->decoder_dims % 16 == 0 && c->lstm_dims % 16 == 0,
             "taco: n_mels/project_dims/decoder_dims/lstm_dims must be multiples of 16");
  MB_REQUIRE(c->r >= 1 && c->r <= c->max_r, "taco: r=%d out of range", c->r);
  MB_REQUIRE(c->dropout < 1.f, "taco: dropout probability %g must be < 1 (0 = the reference's 0.5, negative = off)", (double)c->dropout);
  MB_REQUIRE(c->postnet_dims % 32 == 0 && c->postnet_K >= 1 && c->postnet_K <= 16, "taco: postnet dims");
  const size_t M = c->n_mels, P = c->project_dims, D = c->decoder_dims, H = c->lstm_dims, C = c->postnet_dims;
  n->clear();
  n->push_back(2 * D * M); n->push_back(2 * D); n->push_back(2 * D * 2 * D); n->push_back(2 * D);  // prenet
  n->push_back((size_t)c->lsa_filters * c->lsa_kernel); n->push_back(c->lsa_filters);               // attn_net.conv
  n->push_back(D * c->lsa_filters); n->push_back(D * D); n->push_back(D); n->push_back(D);          // L, W.w, W.b, v
  n->push_back(3 * D * (P + 2 * D)); n->push_back(3 * D * D); n->push_back(3 * D); n->push_back(3 * D);  // attn_rnn
  n->push_back(H * (P + D)); n->push_back(H);                                                        // rnn_input
  for (int i = 0; i < 2; ++i) { n->push_back(4 * H * H); n->push_back(4 * H * H); n->push_back(4 * H); n->push_back(4 * H); }
  n->push_back(M * c->max_r * H);                                                                    // mel_proj
  n->push_back(H + P); n->push_back(1);                                                              // stop_proj
  cbhg_shapes(n, M, C, C, M, c->postnet_K, c->num_highways);                                         // postnet
  n->push_back(M * C);                                                                               // post_proj
  if (c->has_encoder) {
    MB_REQUIRE(c->encoder_dims + c->speaker_dims + c->style_dims == c->project_dims,
               "taco: encoder_dims + speaker_dims + style_dims != project_dims");
    MB_REQUIRE(c->encoder_dims % 32 == 0 && c->embed_dims % 8 == 0 && c->num_chars > 0, "taco: encoder dims");
    const size_t Ce = c->encoder_dims, Em = c->embed_dims;
    n->push_back((size_t)c->num_chars * Em);
    n->push_back(Ce * Em); n->push_back(Ce); n->push_back(Ce * Ce); n->push_back(Ce);               // encoder.pre_net
    cbhg_shapes(n, Ce, Ce, Ce, Ce, c->encoder_K, c->num_highways);                                   // encoder.cbhg
    n->push_back(D * P);                                                                             // encoder_proj
    if (c->has_gst) {
      MB_REQUIRE(c->gst_tokens > 0 && c->gst_heads > 0 && c->style_dims % (2 * c->gst_heads) == 0 && c->gst_n_convs >= 1 &&
                 c->gst_n_convs <= 8 && c->gst_width > 0 && c->speaker_dims % c->gst_width == 0, "taco: bad GST config");
      const size_t E = c->style_dims, dk = E / c->gst_heads;
      size_t fin = 1, wd = c->gst_width;
      for (int i = 0; i < c->gst_n_convs; ++i) {
        const size_t fo = c->gst_filters[i];
        n->push_back(fo * fin * 9); n->push_back(fo);
        for (int j = 0; j < 4; ++j) n->push_back(fo);
        fin = fo; wd = (wd - 1) / 2 + 1;
      }
      n->push_back(3 * (E / 2) * fin * wd); n->push_back(3 * (E / 2) * (E / 2)); n->push_back(3 * (E / 2)); n->push_back(3 * (E / 2));
      n->push_back((size_t)c->gst_tokens * dk);
      n->push_back(E * (E / 2 + c->speaker_dims)); n->push_back(E * dk); n->push_back(E * dk);
    }
  }
  return MB_OK;
}

extern "C" int mb_taco_num_weights(const mb_taco_config* cfg) {
  std::vector<size_t> v;
  if (taco_shapes(cfg, &v)) return MB_EINVAL;
  return (int)v.size();
}
extern "C" size_t mb_taco_weight_numel(const mb_taco_config* cfg, int index) {
  std::vector<size_t> v;
  if (taco_shapes(cfg, &v) || index < 0 || index >= (int)v.size()) return 0;
  return v[index];
}

extern "C" int mb_taco_create(const mb_taco_config* cfg, const float* const* hw, int n_weights, mb_taco** out) {
  MB_REQUIRE(out && hw, "taco_create: null pointer");
  std::vector<size_t> shapes;
  int rc = taco_shapes(cfg, &shapes);
  if (rc) return rc;
  MB_REQUIRE(n_weights == (int)shapes.size(), "taco_create: expected %d weight tensors, got %d", (int)shapes.size(), n_weights);
  mb_taco* t = new mb_taco();
  t->cfg = *cfg;
  // dropout: 0 (a zero-initialised config) = the reference's always-on 0.5 (pre_net.py:23,26); negative = disabled
  t->cfg.dropout = cfg->dropout == 0.f ? 0.5f : (cfg->dropout < 0.f ? 0.f : cfg->dropout);
  const int M = cfg->n_mels, P = cfg->project_dims, D = cfg->decoder_dims, H = cfg->lstm_dims, C = cfg->postnet_dims;
  std::vector<float> rows, packed;
  int ix = 0;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  // prenet
  pack_rowtile(hw[ix], 2 * D, M, 4, &packed); RC(t->pre1_w.upload(packed.data(), packed.size())); RC(t->pre1_b.upload(hw[ix + 1], 2 * D));
  pack_rowtile(hw[ix + 2], 2 * D, 2 * D, 4, &packed); RC(t->pre2_w.upload(packed.data(), packed.size())); RC(t->pre2_b.upload(hw[ix + 3], 2 * D));
  ix += 4;
  // LSA
  RC(t->lsa_conv_w.upload(hw[ix], (size_t)cfg->lsa_filters * cfg->lsa_kernel)); RC(t->lsa_conv_b.upload(hw[ix + 1], cfg->lsa_filters));
  RC(t->lsa_L.upload(hw[ix + 2], (size_t)D * cfg->lsa_filters)); RC(t->lsa_W.upload(hw[ix + 3], (size_t)D * D));
  RC(t->lsa_Wb.upload(hw[ix + 4], D)); RC(t->lsa_v.upload(hw[ix + 5], D));
  {  // M = L . conv_w ([D][Kl], stored [Kl][D]), c0 = L . conv_b, W^T
    const int Fl = cfg->lsa_filters, Kl = cfg->lsa_kernel;
    const float *cw = hw[ix], *cb = hw[ix + 1], *Lw = hw[ix + 2], *Ww = hw[ix + 3];
    std::vector<float> Mt((size_t)Kl * D), c0(D), Wt((size_t)D * D);
    for (int dd = 0; dd < D; ++dd) {
      double acc0 = 0.0;
      for (int f = 0; f < Fl; ++f) acc0 += (double)Lw[(size_t)dd * Fl + f] * (double)cb[f];
      c0[dd] = (float)acc0;
      for (int j = 0; j < Kl; ++j) {
        double acc = 0.0;
        for (int f = 0; f < Fl; ++f) acc += (double)Lw[(size_t)dd * Fl + f] * (double)cw[(size_t)f * Kl + j];
        Mt[(size_t)j * D + dd] = (float)acc;
      }
      for (int k2 = 0; k2 < D; ++k2) Wt[(size_t)k2 * D + dd] = Ww[(size_t)dd * D + k2];
    }
    RC(t->lsa_Mt.upload(Mt.data(), Mt.size())); RC(t->lsa_c0.upload(c0.data(), c0.size())); RC(t->lsa_Wt.upload(Wt.data(), Wt.size()));
    if (D % 32 == 0 && Kl <= 32) {  // rows of 4: Wq4[(q*8 + k4)*D + d] = W[d][q*(D/4) + 4 k4 ..], Mq4[j4*D + d] = M[d][4 j4 ..]
      const int KQ = D / 4;  // k range of a quarter (32 for D = 128)
      std::vector<float> Wq((size_t)D * D, 0.f), Mq((size_t)8 * 2 * 64 * 4, 0.f);
      for (int dd = 0; dd < D; ++dd) {
        for (int k2 = 0; k2 < D; ++k2) {
          const int q = k2 / KQ, k4 = (k2 % KQ) / 4, c4 = k2 % 4;
          Wq[(((size_t)q * (KQ / 4) + k4) * D + dd) * 4 + c4] = Ww[(size_t)dd * D + k2];
        }
        // MFMA B fragment of tap j = 4 g + kq for d = 16 dt + i: float4 (dt, gh = g / 4) of lane (kq * 16 + i), component g % 4
        for (int j = 0; j < Kl; ++j) {
          const int g = j / 4, kq = j % 4, dt = dd / 16, i2 = dd % 16;
          Mq[(((size_t)dt * 2 + g / 4) * 64 + kq * 16 + i2) * 4 + (g % 4)] = Mt[(size_t)j * D + dd];
        }
      }
      RC(t->lsa_Wq4.upload(Wq.data(), Wq.size())); RC(t->lsa_Mq4.upload(Mq.data(), Mq.size()));
    }
  }
  ix += 6;
  // attn_rnn GRUCell(P + 2D -> D)
  cell_rows(hw[ix], P + 2 * D, P + 2 * D, hw[ix + 1], D, D, 3, &rows);
  pack_rowtile(rows.data(), 3 * D, P + 3 * D, 3, &packed);
  RC(t->attn_w.upload(packed.data(), packed.size())); RC(t->attn_bih.upload(hw[ix + 2], 3 * D)); RC(t->attn_bhh.upload(hw[ix + 3], 3 * D));
  ix += 4;
  // rnn_input
  pack_rowtile(hw[ix], H, P + D, 4, &packed); RC(t->rin_w.upload(packed.data(), packed.size())); RC(t->rin_b.upload(hw[ix + 1], H));
  ix += 2;
  // res_rnn1 / res_rnn2 LSTMCell(H -> H)
  cell_rows(hw[ix], H, H, hw[ix + 1], H, H, 4, &rows); pack_rowtile(rows.data(), 4 * H, 2 * H, 4, &packed);
  RC(t->l1_w.upload(packed.data(), packed.size())); RC(t->l1_bih.upload(hw[ix + 2], 4 * H)); RC(t->l1_bhh.upload(hw[ix + 3], 4 * H));
  cell_rows(hw[ix], H, H, hw[ix + 1], 0, H, 4, &rows); pack_rowtile(rows.data(), 4 * H, H, 4, &packed);
  RC(t->l1_wx.upload(packed.data(), packed.size()));
  pack_rowtile(hw[ix + 1], 4 * H, H, 4, &packed); RC(t->l1_whh.upload(packed.data(), packed.size()));
  ix += 4;
  cell_rows(hw[ix], H, H, hw[ix + 1], H, H, 4, &rows); pack_rowtile(rows.data(), 4 * H, 2 * H, 4, &packed);
  RC(t->l2_w.upload(packed.data(), packed.size())); RC(t->l2_bih.upload(hw[ix + 2], 4 * H)); RC(t->l2_bhh.upload(hw[ix + 3], 4 * H));
  cell_rows(hw[ix], H, H, hw[ix + 1], 0, H, 4, &rows); pack_rowtile(rows.data(), 4 * H, H, 4, &packed);
  RC(t->l2_wx.upload(packed.data(), packed.size()));
  pack_rowtile(hw[ix + 1], 4 * H, H, 4, &packed); RC(t->l2_whh.upload(packed.data(), packed.size()));
  ix += 4;
  // mel_proj: keep rows m*max_r + j for j < r, ordered frame-major (j, m)  (tacotron.py:128-129)
  {
    std::vector<float> sel((size_t)cfg->r * M * H);
    for (int j = 0; j < cfg->r; ++j)
      for (int m = 0; m < M; ++m)
        memcpy(&sel[((size_t)j * M + m) * H], hw[ix] + ((size_t)m * cfg->max_r + j) * H, sizeof(float) * H);
    pack_rowtile(sel.data(), cfg->r * M, H, 4, &packed);
    RC(t->mel_w.upload(packed.data(), packed.size()));
    ix += 1;
  }
  RC(t->stop_w.upload(hw[ix], H + P)); RC(t->stop_b.upload(hw[ix + 1], 1));
  ix += 2;
  // ---- fast decoder loop (taco_fast.h): production dims only; every k extent a multiple of 128 ----
  t->fast = D == 128 && P == 1024 && H == 1024 && M % 16 == 0 && (cfg->r * M) % 16 == 0 && cfg->lsa_kernel <= 31 && (cfg->lsa_kernel & 1);
  if (t->fast && !rc) {
    // weight list offsets (taco_shapes order): 0-3 prenet, 4-9 LSA, 10-13 attn_rnn, 14-15 rnn_input, 16-19 / 20-23 LSTMs, 24 mel_proj, 25-26 stop
    const float *fc1_w = hw[0], *a_wih = hw[10], *a_whh = hw[11], *a_bih = hw[12], *a_bhh = hw[13];
    const float *mel_w = hw[24], *stop_w = hw[25];
    // attention GRU on its prenet columns W_ih[:, P:P+2D] (GRU tile order, K = 2D)
    cell_rows(a_wih + P, 2 * D, P + 2 * D, a_whh, 0, D, 3, &rows);
    pack_rowtile(rows.data(), 3 * D, 2 * D, 3, &packed); RC(t->f_gru_w.upload(packed.data(), packed.size()));
    // ... and its context / hidden parts over K = [context | attn_hidden], two accumulators
    cell_rows(a_wih, P, P + 2 * D, a_whh, D, D, 3, &rows);
    pack_rowtile(rows.data(), 3 * D, P + D, 3, &packed); RC(t->f_pre_w.upload(packed.data(), packed.size()));
    RC(pack_rowtile16(rows.data(), 3 * D, P + D, 3, &t->i_pre));
    RC(pack_rowtile16(hw[14], H, P + D, 4, &t->i_rin));  // rnn_input.weight [H][P + D]
    std::vector<float> q4((size_t)D * 4), h4((size_t)D * 4);
    for (int j = 0; j < D; ++j)
      for (int g = 0; g < 4; ++g) {
        q4[(size_t)j * 4 + g] = g < 3 ? a_bih[g * D + j] : 0.f;
        h4[(size_t)j * 4 + g] = g < 3 ? a_bhh[g * D + j] : 0.f;
      }
    RC(t->f_bih4.upload(q4.data(), q4.size())); RC(t->f_bhh4.upload(h4.data(), h4.size()));
    for (int l = 0; l < 2; ++l) {  // LSTM biases b_ih + b_hh per unit (i, f, g, o)
      const float *bi = hw[16 + 4 * l + 2], *bh = hw[16 + 4 * l + 3];
      std::vector<float> b4((size_t)H * 4);
      for (int j = 0; j < H; ++j)
        for (int g = 0; g < 4; ++g) b4[(size_t)j * 4 + g] = bi[g * H + j] + bh[g * H + j];
      RC((l ? t->f_l2_b4 : t->f_l1_b4).upload(b4.data(), b4.size()));
    }
    {  // prenet fc1 folded through mel_proj's last live frame: W' = fc1 . mel_proj[rows m*max_r + (r-1)]  ([2D][H])
      std::vector<double> acc(H);
      std::vector<float> wf((size_t)2 * D * H);
      for (int o = 0; o < 2 * D; ++o) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int m = 0; m < M; ++m) {
          const double f = fc1_w[(size_t)o * M + m];
          const float* mr = mel_w + ((size_t)m * cfg->max_r + (cfg->r - 1)) * H;
          for (int k = 0; k < H; ++k) acc[k] += f * (double)mr[k];
        }
        for (int k = 0; k < H; ++k) wf[(size_t)o * H + k] = (float)acc[k];
      }
      pack_rowtile(wf.data(), 2 * D, H, 4, &packed); RC(t->f_fc1_w.upload(packed.data(), packed.size()));
      RC(pack_rowtile16(wf.data(), 2 * D, H, 4, &t->i_fc1));
    }
    // stop_proj as two one-live-row tiles: x half (K = H) for the mel launch, context half (K = [P | 0 x D]) for rnn_input's
    pack_rowtile(stop_w, 1, H, 4, &packed); RC(t->f_stop_w.upload(packed.data(), packed.size()));
    RC(pack_rowtile16(stop_w, 1, H, 4, &t->i_stop));
    {  // mel_proj's live rows, frame-major (as packed for mel_w above)
      std::vector<float> sel((size_t)cfg->r * M * H);
      for (int j = 0; j < cfg->r; ++j)
        for (int m = 0; m < M; ++m) memcpy(&sel[((size_t)j * M + m) * H], mel_w + ((size_t)m * cfg->max_r + j) * H, sizeof(float) * H);
      RC(pack_rowtile16(sel.data(), cfg->r * M, H, 4, &t->i_mel));
    }
    {
      std::vector<float> sc((size_t)P + D, 0.f);
      memcpy(sc.data(), stop_w + H, sizeof(float) * P);
      pack_rowtile(sc.data(), 1, P + D, 4, &packed); RC(t->f_stopc_w.upload(packed.data(), packed.size()));
      RC(pack_rowtile16(sc.data(), 1, P + D, 4, &t->i_stopc));
    }
    {  // folded form: projected-memory weights [TACO_PM_LD][P] and rnn_input's attn_hidden columns [4][D][64] x 4
      const float* rin = hw[14];  // rnn_input.weight [H][P + D], input order [context, attn_hidden] (tacotron.py:108-109)
      std::vector<float> wpm((size_t)TACO_PM_LD * P, 0.f);
      for (int r = 0; r < H; ++r) memcpy(&wpm[(size_t)r * P], rin + (size_t)r * (P + D), sizeof(float) * P);
      for (int u = 0; u < D; ++u)
        for (int g = 0; g < 3; ++g) memcpy(&wpm[(size_t)(H + 4 * u + g) * P], a_wih + (size_t)(g * D + u) * (P + 2 * D), sizeof(float) * P);
      memcpy(&wpm[(size_t)(H + 4 * D) * P], stop_w + H, sizeof(float) * P);
      RC(make_linear_conv(&t->pm_conv, wpm.data(), TACO_PM_LD, P, nullptr));
      std::vector<float> ra((size_t)4 * D * 64 * 4);
      for (int pg = 0; pg < 4; ++pg)
        for (int k = 0; k < D; ++k)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) ra[(((size_t)pg * D + k) * 64 + l) * 4 + j] = rin[(size_t)(256 * pg + 4 * l + j) * (P + D) + P + k];
      RC(t->rin_a4.upload(ra.data(), ra.size()));
    }
    for (int l = 0; l < 2; ++l) {  // hidden halves W_hh in LSTM tile order (K = H)
      cell_rows(hw[16 + 4 * l + 1], H, H, hw[16 + 4 * l + 1], 0, H, 4, &rows);
      pack_rowtile(rows.data(), 4 * H, H, 4, &packed);
      RC((l ? t->f_l2_hh : t->f_l1_hh).upload(packed.data(), packed.size()));
      RC(pack_rowtile16(rows.data(), 4 * H, H, 4, l ? &t->i_l2hh : &t->i_l1hh));
      cell_rows(hw[16 + 4 * l], H, H, hw[16 + 4 * l + 1], 0, H, 4, &rows);  // input halves W_ih (LSTM tile order)
      RC(pack_rowtile16(rows.data(), 4 * H, H, 4, l ? &t->i_l2x : &t->i_l1x));
    }
    if (!rc && (hipHostMalloc((void**)&t->h_flags, sizeof(int) * 16) != hipSuccess ||
                pool_stream(0, &t->loop_stream) != MB_OK ||
                hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming) != hipSuccess ||
                hipEventCreate(&t->ev_t0) != hipSuccess || hipEventCreate(&t->ev_t1) != hipSuccess ||
                hipEventCreate(&t->ev_p0) != hipSuccess || hipEventCreate(&t->ev_p1) != hipSuccess ||
                hipEventCreateWithFlags(&t->ev_flags[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&t->ev_flags[1], hipEventDisableTiming) != hipSuccess)) {
      set_error("taco_create: pinned flag buffer / events");
      rc = MB_EHIP;
    }
  }
  // postnet CBHG + post_proj
  const int ix_post = ix;
  RC(make_cbhg(&t->post, hw, &ix, M, C, C, M, cfg->postnet_K, cfg->num_highways));
  RC(make_linear_conv(&t->post_proj, hw[ix], M, C, nullptr));
  if (!rc && M % 4 == 0 && !diag_int("taco_post_cm")) {  // round 6: the postnet's convs as time-major split convs too (A/B: MBHIP_DIAG=taco_post_cm at create)
    RC(make_cbhg_tm(&t->post, hw, ix_post, M, C, C, M, cfg->postnet_K, cfg->num_highways));
    if (!rc && t->post.tm.ok) {
      const int r = make_tml(&t->post_proj_tm, hw[ix], M, C, 1, 0, nullptr, nullptr, nullptr);
      if (r < 0) rc = r;
      if (r) { t->post.tm.release(); t->post_proj_tm.release(); }
    }
  }
  ix += 1;
  if (cfg->has_encoder && !rc) {
    const int Ce = cfg->encoder_dims, Em = cfg->embed_dims;
    RC(t->emb.upload(hw[ix], (size_t)cfg->num_chars * Em)); ix += 1;
    RC(make_linear_conv(&t->enc_fc1, hw[ix], Ce, Em, hw[ix + 1]));
    RC(make_linear_conv(&t->enc_fc2, hw[ix + 2], Ce, Ce, hw[ix + 3]));
    ix += 4;
    const int ix_enc = ix;
    RC(make_cbhg(&t->enc, hw, &ix, Ce, Ce, Ce, Ce, cfg->encoder_K, cfg->num_highways));
    if (!rc && Ce % 4 == 0 && !diag_int("taco_post_cm"))  // the encoder's CBHG on the time-major split convs as well (5 + 8 launches of 37 us -> 1 + 4)
      RC(make_cbhg_tm(&t->enc, hw, ix_enc, Ce, Ce, Ce, Ce, cfg->encoder_K, cfg->num_highways));
    {  // encoder_proj: columns [0, Ce) as a 1x1 conv over the encoder sequence; the rest per utterance
      std::vector<float> we((size_t)D * Ce);
      for (int d = 0; d < D; ++d) memcpy(&we[(size_t)d * Ce], hw[ix] + (size_t)d * P, sizeof(float) * Ce);
      RC(make_linear_conv(&t->enc_proj, we.data(), D, Ce, nullptr));
      RC(t->enc_proj_full.upload(hw[ix], (size_t)D * P));
      ix += 1;
    }
    if (cfg->has_gst && !rc) rc = fold_gst(t, hw, &ix);
  }
#undef RC
  if (rc) { mb_taco_destroy(t); return rc; }
  *out = t;
  return MB_OK;
}

extern "C" void mb_taco_destroy(mb_taco* t) {
  if (!t) return;
  DevBuf* bs[] = {&t->pre1_w, &t->pre1_b, &t->pre2_w, &t->pre2_b, &t->lsa_conv_w, &t->lsa_conv_b, &t->lsa_L, &t->lsa_W,
                  &t->lsa_Wb, &t->lsa_v, &t->attn_w, &t->attn_bih, &t->attn_bhh, &t->rin_w, &t->rin_b, &t->l1_w,
                  &t->l1_bih, &t->l1_bhh, &t->l2_w, &t->l2_bih, &t->l2_bhh, &t->l1_wx, &t->l1_whh, &t->l2_wx, &t->l2_whh, &t->mel_w, &t->stop_w, &t->stop_b,
                  &t->emb, &t->enc_proj_full, &t->lsa_Mt, &t->lsa_c0, &t->lsa_Wt, &t->gst_qconst, &t->gst_WqS, &t->gst_K, &t->gst_V,
                  &t->f_gru_w, &t->f_pre_w, &t->f_bih4, &t->f_bhh4, &t->f_l1_b4, &t->f_l2_b4, &t->f_fc1_w, &t->f_stop_w, &t->f_stopc_w, &t->f_l1_hh, &t->f_l2_hh, &t->lsa_Wq4, &t->lsa_Mq4, &t->rin_a4};
  for (DevBuf* b : bs) b->release();
  t->drop_graph();
  if (t->h_flags) (void)hipHostFree(t->h_flags);
  for (int e = 0; e < 2; ++e) if (t->ev_flags[e]) (void)hipEventDestroy(t->ev_flags[e]);
  if (t->d_trace) (void)hipFree(t->d_trace);
  if (t->ev_in) (void)hipEventDestroy(t->ev_in);
  if (t->ev_t0) (void)hipEventDestroy(t->ev_t0);
  if (t->ev_t1) (void)hipEventDestroy(t->ev_t1);
  if (t->ev_p0) (void)hipEventDestroy(t->ev_p0);
  if (t->ev_p1) (void)hipEventDestroy(t->ev_p1);
  t->loop_stream = nullptr;  // (borrowed from the pool, common.h)
  t->post.release(); t->post_proj.release(); t->post_proj_tm.release(); t->enc.release();
  t->enc_fc1.release(); t->enc_fc2.release(); t->enc_proj.release();
  for (mb_taco::Img16* im : {&t->i_l1x, &t->i_l2x, &t->i_l1hh, &t->i_l2hh, &t->i_rin, &t->i_pre, &t->i_stopc, &t->i_mel, &t->i_fc1, &t->i_stop}) im->w.release();
  t->pm_conv.release();
  delete t;
}

namespace {
struct TacoLayout {
  float *p1, *p2, *attn_h, *context, *x, *x1, *x2, *h1, *c1, *h2, *c2, *melstep, *cumulative, *stop;
  // fast loop (taco_fast.h): FM activations, CM cell state / gate pre-activations
  float *f_p1, *f_p2, *f_ah, *f_ctx, *f_x, *f_x1, *f_x2, *f_h1, *f_h2, *f_c1, *f_c2, *f_xpre, *f_hpre, *f_hp1, *f_hp2, *f_stop_part;
  float *f_ah2;       // folded form: the second attn_hidden buffer (ping-pong)
  float *memT, *pm;   // folded form: the memory channel-major [B][P][T], the projected memory [B][T][TACO_PM_LD]
  unsigned long long *f_p2g, *f_ahg, *f_eg;  // fused front (taco_front_kernel): tagged granules of p2 / attn_hidden / energies
  size_t f_state_bytes;  // the region above, zeroed per call
  float* mpq4;  // mem_proj in MFMA D-fragment order for lsa_fast_body: [B][4 TJ / 16][8][64] float4 = B * 4 TJ * D floats, TJ = 32 | 48
  int* flags;  // [0] done, [1] n_frames, [2] arrive, [3] utterances below the stop threshold, [4] iteration base, [6..7] seed
  // postnet
  float *melc, *linc;
  CbhgWs cb;
  size_t bytes;
};
}  // namespace

static void taco_layout(const mb_taco* t, int B, int T, int max_steps, void* base, TacoLayout* L) {
  const mb_taco_config& c = t->cfg;
  const size_t D = c.decoder_dims, P = c.project_dims, H = c.lstm_dims, M = c.n_mels, C = c.postnet_dims, F = max_steps;
  Arena ar(base, (size_t)-1);
  L->p1 = ar.take<float>(B * 2 * D); L->p2 = ar.take<float>(B * 2 * D);
  L->attn_h = ar.take<float>(2 * B * D);
  L->context = ar.take<float>(2 * B * P);
  L->x = ar.take<float>(B * H); L->x1 = ar.take<float>(B * H); L->x2 = ar.take<float>(B * H);
  L->h1 = ar.take<float>(2 * B * H); L->c1 = ar.take<float>(2 * B * H);
  L->h2 = ar.take<float>(2 * B * H); L->c2 = ar.take<float>(2 * B * H);
  {
    const int nta = (B + 15) / 16;
    L->f_p1 = ar.take<float>(fm_floats(2 * D, nta));
    const size_t start = ar.off - fm_floats(2 * D, nta) * sizeof(float);
    L->f_p2 = ar.take<float>(fm_floats(2 * D, nta)); L->f_ah = ar.take<float>(fm_floats(D, nta));
    L->f_ctx = ar.take<float>(fm_floats(P, nta));
    L->f_x = ar.take<float>(fm_floats(H, nta)); L->f_x1 = ar.take<float>(fm_floats(H, nta)); L->f_x2 = ar.take<float>(fm_floats(H, nta));
    L->f_h1 = ar.take<float>(fm_floats(H, nta)); L->f_h2 = ar.take<float>(fm_floats(H, nta));
    L->f_hp1 = ar.take<float>(4 * cm_items(H, nta)); L->f_hp2 = ar.take<float>(4 * cm_items(H, nta));
    L->f_stop_part = ar.take<float>((size_t)nta * 16);
    L->f_c1 = ar.take<float>(cm_items(H, nta)); L->f_c2 = ar.take<float>(cm_items(H, nta));
    L->f_xpre = ar.take<float>(4 * cm_items(D, nta)); L->f_hpre = ar.take<float>(4 * cm_items(D, nta));
    L->f_p2g = ar.take<unsigned long long>(fm_floats(2 * D, nta)); L->f_ahg = ar.take<unsigned long long>((size_t)nta * 16 * D);
    L->f_eg = ar.take<unsigned long long>((size_t)nta * 16 * 128);
    L->f_ah2 = ar.take<float>(fm_floats(D, nta));
    L->f_state_bytes = ar.off - start;
  }
  L->melstep = ar.take<float>((size_t)B * c.r * M);
  L->cumulative = ar.take<float>((size_t)2 * B * T);
  L->stop = ar.take<float>(B);
  L->mpq4 = ar.take<float>(T <= 192 ? (size_t)B * 4 * (T <= 128 ? 32 : 48) * 128 : 1);
  {
    const bool foldable = t->fast && T <= 192 && B <= 32;  // (the shapes taco_front_kernel serves)
    L->memT = ar.take<float>(foldable ? (size_t)B * P * T : 1);
    L->pm = ar.take<float>(foldable ? (size_t)B * T * 1664 : 1);
  }
  L->flags = ar.take<int>(16);
  L->melc = ar.take<float>(B * M * F);
  L->linc = ar.take<float>(B * M * F);
  cbhg_take(ar, t->post, B, F, &L->cb);
  (void)C;
  L->bytes = ar.off + 256;
}

extern "C" size_t mb_taco_workspace_bytes(const mb_taco* t, int batch, int t_text, int max_steps) {
  if (!t || batch <= 0 || t_text <= 0 || max_steps <= 0) return 0;
  TacoLayout L;
  taco_layout(t, batch, t_text, max_steps, nullptr, &L);
  return L.bytes;
}

namespace {
int run_conv(const ConvL& c, const float* x, int batch, int t, float* y, long long y_bstride, int in_act,
             int out_act, const float* res, const float* gate, int transpose_out, hipStream_t s) {
  mb_conv1d_args a;
  memset(&a, 0, sizeof(a));
  a.d_x = x; a.d_wpacked = c.w.p; a.d_bias = c.b.p; a.d_res = res; a.d_y = y; a.d_gate = gate;
  a.d_post_scale = c.ps.p; a.d_post_shift = c.pt.p;
  a.x_bstride = (long long)c.c_in * t; a.y_bstride = y_bstride ? y_bstride : (long long)c.c_out * t;
  a.res_bstride = a.y_bstride;
  a.batch = batch; a.c_in = c.c_in; a.c_out = c.c_out; a.t_in = t; a.t_out = t;
  a.ksize = c.k; a.dilation = 1; a.pad = c.pad; a.up = 1;
  a.in_act = in_act; a.out_act = out_act; a.transpose_out = transpose_out;
  return mb_conv1d(&a, (mb_stream_t)s);
}
}  // namespace

// THE selection of the production-dims decoder loop's form (exported as mb_taco_loop_form for the host-logic test):
//   batch, t_text   the call;  lsa_fast: the fast attention kernel serves it (production dims, t_text <= 192, no MBHIP_LSA_GENERIC)
//   n_cus           compute units of the device;  images: the fp16 images / projection weights exist (production handles: yes)
//   front_failed    a hand-off of the fused launch timed out on this handle before
//   front_sw / f16_sw / fold_sw   MBHIP_DIAG taco_front / taco_f16 / taco_fold: -1 = unset, 0, 1
// -> launches per iteration: 7 (one launch per stage, fp32 pipe) | 5 (taco_front_kernel) | 4 (+ rnn_input folded into the attention role);
//    *f16 = the K >= 1024 tile products run on the fp16 matrix pipe (fm_gemm16).
// | t_text     | batch <= 16                  | 17..32                       | > 32 (or too few compute units, or front_failed) |
// | <= 128     | 4 launches, fp16 products    | 4 launches, fp16 products    | 7 launches, fp32                                 |
// | 129..192   | 5 launches, fp32 products    | 5 launches, fp16 products    | 7 launches, fp32                                 |
// | > 192      | 7 launches, fp32 (general attention kernel)                                                                    |
static int taco_pick_form(int batch, int t_text, int lsa_fast, int n_cus, int images, int front_failed, int front_sw, int f16_sw, int fold_sw, int* f16_out) {
  const int nta = (batch + 15) / 16;
  const bool front = lsa_fast && batch >= 1 && batch <= 32 && 16 + 32 + 4 * batch <= n_cus && !front_failed && front_sw != 0;
  const bool foldable = front && t_text <= 128 && images;
  const bool f16 = front && images && (f16_sw < 0 ? (nta >= 2 || (foldable && fold_sw != 0)) : f16_sw != 0);
  const bool fold = foldable && (fold_sw < 0 ? f16 : fold_sw != 0);
  if (f16_out) *f16_out = f16 ? 1 : 0;
  return fold ? 4 : front ? 5 : 7;
}
extern "C" int mb_taco_loop_form(int batch, int t_text, int lsa_fast, int n_cus, int images, int front_failed, int front_sw, int f16_sw, int fold_sw, int* f16_out) {
  return taco_pick_form(batch, t_text, lsa_fast, n_cus, images, front_failed, front_sw, f16_sw, fold_sw, f16_out);
}

// ---- fast decoder loop (taco_fast.h): 7 launches per iteration, hipGraph-captured, stop flag polled one replay behind ----
static int taco_fast_loop_body(mb_taco* t, const TacoLayout& L, const float* d_memory, const float* d_memory_proj, const int32_t* d_chars,
                               int B, int T, int max_steps, float min_stop_token, const float* d_dropout, uint64_t seed, float* d_mel,
                               float* d_attn, bool lsa_fast, size_t lds_lsa, int psplit, void* d_workspace, hipStream_t s, int* frames_out,
                               bool front, int* lost_out) {
  const mb_taco_config& c = t->cfg;
  const int D = c.decoder_dims, P = c.project_dims, H = c.lstm_dims, M = c.n_mels, r = c.r;
  const int nta = cdiv(B, 16), n_iter_max = cdiv(max_steps, r);
  const bool drop_enabled = c.dropout > 0.f;
  DropK dk;
  dk.mask = drop_enabled ? d_dropout : nullptr; dk.ld = 2 * D; dk.layer = 0; dk.it_add = 0;
  dk.it_stride = (long long)2 * B * 2 * D;  // masks [n_iter][2 layers][B][2D]
  dk.it_limit = n_iter_max;
  dk.thresh = drop_enabled ? (unsigned)std::min(4294967295.0, (double)c.dropout * 4294967296.0) : 0u;
  dk.scale = drop_enabled ? 1.f / (1.f - c.dropout) : 1.f; dk.enabled = drop_enabled ? 1 : 0;
  int* flags = L.flags;
  std::string trace_file;  // diagnostics (MBHIP_DIAG=taco_trace=<file>)
  const char* trace_path = diag_str("taco_trace", &trace_file) ? trace_file.c_str() : nullptr;
  if (trace_path && !t->d_trace) MB_HIP(hipMalloc((void**)&t->d_trace, sizeof(unsigned long long) * TS_WORDS));
  unsigned long long* tr = trace_path ? t->d_trace : nullptr;
  if (tr) MB_HIP(hipMemsetAsync(tr, 0, sizeof(unsigned long long) * TS_WORDS, s));
  MB_HIP(hipMemsetAsync(L.f_p1, 0, L.f_state_bytes, s));
  MB_HIP(hipMemcpyAsync(flags + TF_SEED, &seed, sizeof(seed), hipMemcpyHostToDevice, s));  // pageable source: staged before return
  // folded form (4 launches per iteration): rnn_input, the next GRU pre-activation and the stop logit's context half are linear in the
  // context = sum_t score_t memory_t, so with the memory rows projected ONCE per call (a 1 x 1 conv on the split-fp16 path,
  // conv1d.hip) the attention workgroups produce them directly and the rnn_input launch is gone
  // (taco_pick_form.  T <= 128: the attention's LDS window form; by default only together with the fp16-pipe riders: the rnn_input
  //  launch is also where 128 hidden-half tiles rode, and only the short riders fit the front / mel launches without a second round)
  int f16_i = 0;
  const int form = !front ? 7 : taco_pick_form(B, T, 1, t->n_cus, t->i_l1x.w.p && t->pm_conv.w.p ? 1 : 0, 0, 1, diag_int("taco_f16", -1), diag_int("taco_fold", -1), &f16_i);
  const bool f16 = front && f16_i != 0, fold = form == 4;
  if (fold) {
    hipLaunchKernelGGL(btc_to_bct_kernel, dim3(cdiv(P, 32), cdiv(T, 32), B), dim3(256), 0, s, d_memory, L.memT, T, P);
    MB_HIP(hipGetLastError());
    const int rcv = run_conv(t->pm_conv, L.memT, B, T, L.pm, (long long)T * TACO_PM_LD, 0, 0, nullptr, nullptr, 1, s);
    if (rcv) return rcv;
  }
  {
    TfP1K pk;
    pk.b_fc1 = t->pre1_b.p; pk.p1 = L.f_p1; pk.nta = nta; pk.B = B; pk.rows = 2 * D; pk.flags = flags; pk.drop = dk;
    hipLaunchKernelGGL(taco_p1_init_kernel, dim3(2 * D / 16, nta), dim3(64), 0, s, pk);
  }
  // attention-GRU pre-activations of iteration 0: context = 0, attn_hidden = 0 -> the biases
  {
    TfRinK rk;
    rk.w_rin = t->rin_w.p; rk.b_rin = t->rin_b.p; rk.w_pre = t->f_pre_w.p;
    rk.bih4 = reinterpret_cast<const float4*>(t->f_bih4.p); rk.bhh4 = reinterpret_cast<const float4*>(t->f_bhh4.p);
    rk.ctx = L.f_ctx; rk.ah = L.f_ah; rk.x = L.f_x; rk.xpre = reinterpret_cast<float4*>(L.f_xpre); rk.hpre = reinterpret_cast<float4*>(L.f_hpre);
    rk.nta = nta; rk.n_rin = H / 16; rk.flags = flags; rk.trace = nullptr;
    rk.w_stopc = t->f_stopc_w.p; rk.stop_part = L.f_stop_part; rk.hh = TfHhK{nullptr, nullptr, nullptr, 0, 0};
    if (nta >= 2) hipLaunchKernelGGL(taco_rin_kernel<2>, dim3(H / 16 + D / 4, cdiv(nta, 2)), dim3(512), 0, s, rk);
    else hipLaunchKernelGGL(taco_rin_kernel<1>, dim3(H / 16 + D / 4, nta), dim3(512), 0, s, rk);
  }
  MB_HIP(hipGetLastError());

  // fused front: the attention workgroups hold 128 KB of LDS each, so every workgroup of that launch has a compute unit to itself and
  // the W_hh2 . h2 tiles behind them come in rounds of (256 - 48 - B psplit): the first hh2_mel row tiles ride in the previous
  // iteration's mel launch instead (h2 is final there), the rest stay (sweep: profiles/r05_taco_front_ab.json)
  // (f16, decided above: the K >= 1024 tile products -- LSTM input halves, rnn_input, mel / fc1' / stop rows, the hidden-half riders -- on
  //  the fp16 matrix pipe from split images, fm_gemm16: fp32-grade, not the 7-launch loop's bits; a value beyond fp16's range raises
  //  flags[TF_LOST] -> the call reruns on the exact loop.  On its own it pays with two column tiles only -- with one the fp32 products
  //  are half as many and the conversions cost what is saved, 31.6 against 32.1 us at batch 16 -- but the folded form needs its short riders)
  const int front_free = t->n_cus - (2 * D / 16 + D / 4 + B * psplit);  // compute units the hh2 tiles of the fused launch start on
  // (96: the mel launch stays within one round of 256 workgroups; the fp16-pipe riders are short enough for the front launch to keep all)
  const int hh2_auto = f16 ? 0 : std::min(std::max(H / 4 - 2 * front_free, 0), 96);
  const int hh2_mel = front ? std::min(std::max(diag_int("taco_hh2_mel", hh2_auto), 0), H / 4) : 0;
  auto img16 = [&](TfHhK& h, const mb_taco::Img16& im) {
    if (f16) { h.w16 = reinterpret_cast<const uint4*>(im.w.p); h.unscale = im.unscale; }
  };
  // folded form: W_hh1 row tiles [0, m1) ride in the mel launch (one round with its 27 chain tiles), the rest behind the W_hh2 tiles of the front launch
  const int m1 = fold ? std::min(std::max(diag_int("taco_m1", 200), 0), H / 4) : 0;
  int hh1_split = H / 8;  // row tiles of W_hh1 . h1 taken by the mel launch (the rest: next rnn_input launch)
  // (an hh1-split sweep, round 2: flat between 128 and 224 rows -- the switch is gone, the value stays)
  auto iteration = [&](int pp, int it_off) -> int {
    // (the LSTM state needs no ping-pong: h is read only by the hh jobs, which have finished before the next LSTM launch)
    const dim3 blk(512);
    const int gy = nta >= 2 ? cdiv(nta, 2) : nta;
#define TF_LAUNCH16(KERNEL, GX, ARG)                                                         \
    do {                                                                                     \
      if (nta >= 2) hipLaunchKernelGGL((KERNEL<2, true>), dim3(GX, gy), blk, 0, s, ARG);     \
      else hipLaunchKernelGGL((KERNEL<1, true>), dim3(GX, gy), blk, 0, s, ARG);              \
    } while (0)
#define TF_LAUNCH(KERNEL, GX, ARG)                                                     \
    do {                                                                               \
      if (nta >= 2) hipLaunchKernelGGL(KERNEL<2>, dim3(GX, gy), blk, 0, s, ARG);       \
      else hipLaunchKernelGGL(KERNEL<1>, dim3(GX, gy), blk, 0, s, ARG);                \
    } while (0)
    // 1. prenet layer 2 (layer 1 was left by the previous iteration's mel launch / the prologue)
    TfFcK fk;
    fk.w = t->pre2_w.p; fk.bias = t->pre2_b.p; fk.xin = L.f_p1; fk.yout = L.f_p2; fk.nta = nta; fk.B = B; fk.it_off = it_off;
    fk.flags = flags; fk.drop = dk; fk.drop.layer = 1; fk.trace = tr;
    if (fk.drop.mask) fk.drop.mask += (size_t)B * 2 * D;  // layer 1
    if (!front) TF_LAUNCH(taco_fc2_kernel, 2 * D / 16, fk);
    // 2. attention GRU on the prenet columns
    TfGruK gk;
    gk.w = t->f_gru_w.p; gk.xin = L.f_p2; gk.xpre = reinterpret_cast<const float4*>(L.f_xpre);
    gk.hpre = reinterpret_cast<const float4*>(L.f_hpre); gk.ah = L.f_ah; gk.nta = nta; gk.B = B; gk.flags = flags; gk.trace = tr;
    if (fold) {  // attn_hidden ping-pongs (pp: this iteration's input buffer); the role multiplies W_hh . attn_hidden itself
      gk.ah = pp ? L.f_ah2 : L.f_ah; gk.ah_out = pp ? L.f_ah : L.f_ah2;
      gk.w_pre = t->f_pre_w.p; gk.bhh4 = reinterpret_cast<const float4*>(t->f_bhh4.p);
    }
    if (!front) TF_LAUNCH(taco_gru_kernel, D / 4, gk);
    // 3. location-sensitive attention + context
    LsaK lk;
    lk.query = L.f_ah; lk.mem_proj = d_memory_proj; lk.memory = d_memory; lk.chars = d_chars;
    lk.cum_in = L.cumulative + (size_t)pp * B * T; lk.cum_out = L.cumulative + (size_t)(pp ^ 1) * B * T; lk.psplit = psplit;
    lk.conv_w = t->lsa_conv_w.p; lk.conv_b = t->lsa_conv_b.p; lk.Lw = t->lsa_L.p; lk.Ww = t->lsa_W.p; lk.Wb = t->lsa_Wb.p;
    lk.vw = t->lsa_v.p; lk.context = L.f_ctx; lk.attn_out = d_attn; lk.T = T; lk.D = D; lk.P = P; lk.Fl = c.lsa_filters;
    lk.Kl = c.lsa_kernel; lk.iter = it_off; lk.n_iter_max = n_iter_max; lk.skip_flag = flags + TF_DONE;
    lk.Mt = t->lsa_Mt.p; lk.c0 = t->lsa_c0.p; lk.Wt = t->lsa_Wt.p; lk.fm_nta = nta; lk.iter_base = flags + TF_ITER; lk.trace = tr;
    lk.Wq4 = reinterpret_cast<const float4*>(t->lsa_Wq4.p); lk.Mf4 = reinterpret_cast<const float4*>(t->lsa_Mq4.p);
    lk.mpf4 = reinterpret_cast<const float4*>(L.mpq4);
    // ... with the hidden half of THIS iteration's second LSTM (W_hh2 . h2 of the previous iteration) on the idle CUs
    TfHhK hh2;
    hh2.w = t->f_l2_hh.p; hh2.h = L.f_h2; hh2.hpre = reinterpret_cast<float4*>(L.f_hp2); hh2.n_tiles = H / 4; hh2.tile0 = 0;
    img16(hh2, t->i_l2hh);
    const int n_lsa = B * psplit;
    if (front) {  // 1..3 as one launch (taco_front_kernel): fc2 tiles | GRU tiles | attention workgroups | hh2 tiles
      TfFrontX fx;
      fx.p2g = L.f_p2g; fx.ahg = L.f_ahg; fx.lost = flags + TF_LOST; fx.n_fc2 = 2 * D / 16; fx.n_gru = D / 4; fx.watch = diag_int("taco_gru_watch", 1);
      fx.hh_pairs = f16 ? 0 : diag_int("taco_hh_pairs", 1);  // (fp16-pipe riders are short enough to come one tile per workgroup)
      lk.dma_early = diag_int("taco_dma_early", nta >= 2 ? 1 : 0);  // (one column tile: 29.2 against 30.0 us per iteration with the DMA behind B2)
      lk.q_gran = L.f_ahg; lk.lost = flags + TF_LOST; lk.e_gran = L.f_eg;
      if (fold) {
        lk.fold = 1; lk.memory = L.pm; lk.mem_ld = TACO_PM_LD; lk.context = L.f_x;
        lk.rin_a4 = reinterpret_cast<const float4*>(t->rin_a4.p); lk.rin_b = t->rin_b.p;
        lk.bih4 = reinterpret_cast<const float4*>(t->f_bih4.p); lk.xpre_out = reinterpret_cast<float4*>(L.f_xpre); lk.stop_part = L.f_stop_part;
      }
      hh2.tile0 = fold ? 0 : hh2_mel; hh2.n_tiles = H / 4 - hh2.tile0;
      TfHhK hh1b;  // folded form: W_hh1 row tiles [m1, H / 4) (h1 of the previous iteration: final since its LSTM-1 launch)
      hh1b.w = t->f_l1_hh.p; hh1b.h = L.f_h1; hh1b.hpre = reinterpret_cast<float4*>(L.f_hp1); hh1b.tile0 = m1; hh1b.n_tiles = fold ? H / 4 - m1 : 0;
      img16(hh1b, t->i_l1hh);
      if (fold) fx.hh_pairs = 0;
      const dim3 g1(fx.n_fc2 + fx.n_gru + n_lsa + (fx.hh_pairs ? cdiv(hh2.n_tiles, 2) : hh2.n_tiles + hh1b.n_tiles));
#define TF_FRONT(TJ_, NT_, FOLD_)                                                                                                      \
      do {                                                                                                                              \
        if (f16) hipLaunchKernelGGL((taco_front_kernel<TJ_, NT_, true, FOLD_>), g1, blk, 0, s, fk, gk, lk, hh2, hh1b, fx, n_lsa, B, gy, nta);  \
        else hipLaunchKernelGGL((taco_front_kernel<TJ_, NT_, false, FOLD_>), g1, blk, 0, s, fk, gk, lk, hh2, hh1b, fx, n_lsa, B, gy, nta);     \
      } while (0)
      if (T <= 128 && nta >= 2) { if (fold) TF_FRONT(32, 2, true); else TF_FRONT(32, 2, false); }
      else if (T <= 128) { if (fold) TF_FRONT(32, 1, true); else TF_FRONT(32, 1, false); }
      else if (nta >= 2) TF_FRONT(48, 2, false);
      else TF_FRONT(48, 1, false);
#undef TF_FRONT
    } else if (lsa_fast) {
      const dim3 g1(n_lsa + (H / 4) * gy);
      if (T <= 128 && nta >= 2) hipLaunchKernelGGL((lsa_hh_kernel<32, 2>), g1, blk, 0, s, lk, hh2, n_lsa, B, gy, nta);
      else if (T <= 128) hipLaunchKernelGGL((lsa_hh_kernel<32, 1>), g1, blk, 0, s, lk, hh2, n_lsa, B, gy, nta);
      else if (nta >= 2) hipLaunchKernelGGL((lsa_hh_kernel<48, 2>), g1, blk, 0, s, lk, hh2, n_lsa, B, gy, nta);
      else hipLaunchKernelGGL((lsa_hh_kernel<48, 1>), g1, blk, 0, s, lk, hh2, n_lsa, B, gy, nta);
    } else {  // long texts: the general attention kernel, hidden half as its own launch in front of it
      if (nta >= 2) hipLaunchKernelGGL(taco_hh_kernel<2>, dim3(H / 4, gy), blk, 0, s, hh2, nta, (const int*)flags);
      else hipLaunchKernelGGL(taco_hh_kernel<1>, dim3(H / 4, gy), blk, 0, s, hh2, nta, (const int*)flags);
      hipLaunchKernelGGL(lsa_kernel, dim3(B, psplit), dim3(512), lds_lsa, s, lk);
    }
    // 4. rnn_input beside the next iteration's attention-GRU pre-activations and the stop token's context half
    TfRinK rk;
    rk.w_rin = t->rin_w.p; rk.b_rin = t->rin_b.p; rk.w_pre = t->f_pre_w.p;
    rk.bih4 = reinterpret_cast<const float4*>(t->f_bih4.p); rk.bhh4 = reinterpret_cast<const float4*>(t->f_bhh4.p);
    rk.ctx = L.f_ctx; rk.ah = L.f_ah; rk.x = L.f_x; rk.xpre = reinterpret_cast<float4*>(L.f_xpre); rk.hpre = reinterpret_cast<float4*>(L.f_hpre);
    rk.nta = nta; rk.n_rin = H / 16; rk.flags = flags; rk.trace = tr;
    rk.w_stopc = t->f_stopc_w.p; rk.stop_part = L.f_stop_part;
    // hidden half of THIS iteration's first LSTM: row tiles [hh1_split, H/4) here, [0, hh1_split) rode in the previous
    // iteration's mel launch (both launches leave most CUs idle; one launch taking all 256 tiles slowed its chain jobs)
    rk.hh.w = t->f_l1_hh.p; rk.hh.h = L.f_h1; rk.hh.hpre = reinterpret_cast<float4*>(L.f_hp1);
    rk.hh.tile0 = hh1_split; rk.hh.n_tiles = H / 4 - hh1_split;
    if (!fold) {
      img16(rk.hh, t->i_l1hh);
      if (f16) {
        rk.rin16 = reinterpret_cast<const uint4*>(t->i_rin.w.p); rk.us_rin = t->i_rin.unscale;
        rk.pre16 = reinterpret_cast<const uint4*>(t->i_pre.w.p); rk.us_pre = t->i_pre.unscale;
        rk.stopc16 = reinterpret_cast<const uint4*>(t->i_stopc.w.p); rk.us_stopc = t->i_stopc.unscale; rk.lost = flags + TF_LOST;
      }
      if (f16) TF_LAUNCH16(taco_rin_kernel, H / 16 + D / 4 + 1 + rk.hh.n_tiles, rk);
      else TF_LAUNCH(taco_rin_kernel, H / 16 + D / 4 + 1 + rk.hh.n_tiles, rk);
    }
    // 5./6. residual LSTMs
    TfLstmK lk1;
    lk1.w = t->l1_wx.p; lk1.b4 = reinterpret_cast<const float4*>(t->f_l1_b4.p); lk1.x = L.f_x; lk1.h_out = L.f_h1;
    lk1.hpre = reinterpret_cast<const float4*>(L.f_hp1);
    lk1.c = L.f_c1; lk1.x_out = L.f_x1; lk1.nta = nta; lk1.flags = flags; lk1.trace = tr; lk1.trace_slot = TS_LSTM1;
    TfLstmK lk2 = lk1;
    lk2.w = t->l2_wx.p; lk2.b4 = reinterpret_cast<const float4*>(t->f_l2_b4.p); lk2.x = L.f_x1; lk2.h_out = L.f_h2;
    lk2.hpre = reinterpret_cast<const float4*>(L.f_hp2);
    lk2.c = L.f_c2; lk2.x_out = L.f_x2; lk2.trace_slot = TS_LSTM2;
    if (f16) {
      lk1.w16 = reinterpret_cast<const uint4*>(t->i_l1x.w.p); lk1.unscale = t->i_l1x.unscale; lk1.lost = flags + TF_LOST;
      lk2.w16 = reinterpret_cast<const uint4*>(t->i_l2x.w.p); lk2.unscale = t->i_l2x.unscale; lk2.lost = flags + TF_LOST;
      TF_LAUNCH16(taco_lstm_kernel, H / 4, lk1);
      TF_LAUNCH16(taco_lstm_kernel, H / 4, lk2);
    } else {
      TF_LAUNCH(taco_lstm_kernel, H / 4, lk1);
      TF_LAUNCH(taco_lstm_kernel, H / 4, lk2);
    }
    // 7. mel frames, next prenet layer 1, stop token + stop rule
    TfMelK mk;
    mk.w_mel = t->mel_w.p; mk.w_fc1 = t->f_fc1_w.p; mk.b_fc1 = t->pre1_b.p; mk.w_stop = t->f_stop_w.p; mk.b_stop = t->stop_b.p;
    mk.x2 = L.f_x2; mk.stop_part = L.f_stop_part; mk.p1 = L.f_p1; mk.mel_out = d_mel; mk.stop_out = L.stop;
    mk.hh.w = t->f_l1_hh.p; mk.hh.h = L.f_h1; mk.hh.hpre = reinterpret_cast<float4*>(L.f_hp1); mk.hh.n_tiles = hh1_split; mk.hh.tile0 = 0;
    img16(mk.hh, t->i_l1hh);
    if (f16) {
      mk.mel16 = reinterpret_cast<const uint4*>(t->i_mel.w.p); mk.us_mel = t->i_mel.unscale;
      mk.fc116 = reinterpret_cast<const uint4*>(t->i_fc1.w.p); mk.us_fc1 = t->i_fc1.unscale;
      mk.stop16 = reinterpret_cast<const uint4*>(t->i_stop.w.p); mk.us_stop = t->i_stop.unscale;
    }
    mk.nta = nta; mk.B = B; mk.n_mel = r * M / 16; mk.M = M; mk.r = r; mk.max_steps = max_steps; mk.it_off = it_off;
    mk.min_stop_token = min_stop_token; mk.flags = flags; mk.drop = dk; mk.drop.layer = 0; mk.drop.it_add = 1; mk.trace = tr;
    mk.hh2 = hh2; mk.hh2.tile0 = 0; mk.hh2.n_tiles = hh2_mel;
    if (fold) { mk.hh.tile0 = 0; mk.hh.n_tiles = m1; mk.hh2.n_tiles = 0; }  // W_hh1 row tiles [0, m1); every W_hh2 tile rides in the front launch
    const int mel_riders = fold ? m1 : hh1_split + hh2_mel;
    if (f16) TF_LAUNCH16(taco_mel_kernel, r * M / 16 + 2 * D / 16 + 1 + mel_riders, mk);
    else TF_LAUNCH(taco_mel_kernel, r * M / 16 + 2 * D / 16 + 1 + mel_riders, mk);
#undef TF_LAUNCH
#undef TF_LAUNCH16
    MB_HIP(hipGetLastError());
    return MB_OK;
  };

  MB_HIP(hipEventRecord(t->ev_t0, s));
  int G = 16;  // iterations per graph replay (even: the LSTM state / cumulative-attention parity returns to 0)
  if (const char* ge = getenv("MBHIP_GRAPH_STEPS")) G = atoi(ge) & ~1;  // (steps of this loop = decoder iterations)
  const bool use_graph = getenv("MBHIP_NO_GRAPH") == nullptr && G >= 2 && n_iter_max >= G;
  int it_done = 0, rc = MB_OK;
  bool stopped = false;
  if (use_graph) {
    const unsigned variant = !front ? 0u : 1u | (unsigned)hh2_mel << 1 | (unsigned)diag_int("taco_gru_watch", 1) << 10 | (unsigned)diag_int("taco_dma_early", nta >= 2 ? 1 : 0) << 11 |
                                          (unsigned)diag_int("taco_hh_pairs", 1) << 12 | (fold ? 1u : 0u) << 13 | (f16 ? 1u : 0u) << 14 | (unsigned)m1 << 15;
    mb_taco::GraphKey key = {d_memory, d_memory_proj, d_chars, d_dropout, d_mel, d_attn, d_workspace, B, T, max_steps, G, min_stop_token, (int)variant};
    if (!t->graph_exec || !(key == t->gkey)) {
      t->drop_graph();
      MB_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      for (int i = 0; i < G && !rc; ++i) rc = iteration(i & 1, i);
      hipLaunchKernelGGL(taco_bump_kernel, dim3(1), dim3(1), 0, s, flags, G);
      hipError_t e = hipStreamEndCapture(s, &t->graph);
      if (rc) { t->drop_graph(); return rc; }
      if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture", __FILE__, __LINE__);
      e = hipGraphInstantiate(&t->graph_exec, t->graph, nullptr, nullptr, 0);
      if (e != hipSuccess) { t->drop_graph(); return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__); }
      t->gkey = key;
    }
    const int reps = n_iter_max / G;
    for (int rep = 0; rep < reps; ++rep) {
      MB_HIP(hipGraphLaunch(t->graph_exec, s));
      MB_HIP(hipMemcpyAsync(t->h_flags + 8 * (rep & 1), flags, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
      MB_HIP(hipEventRecord(t->ev_flags[rep & 1], s));
      it_done += G;
      if (rep > 0) {  // look at the flags of the replay before: the device never waits for the host
        MB_HIP(hipEventSynchronize(t->ev_flags[(rep - 1) & 1]));
        if (t->h_flags[8 * ((rep - 1) & 1) + TF_DONE] || t->h_flags[8 * ((rep - 1) & 1) + TF_LOST]) { stopped = true; break; }
      }
    }
  }
  for (int it = it_done; it < n_iter_max && !stopped; ++it) {  // eager tail (and short runs): the offset carries the iteration
    if ((rc = iteration(it & 1, it - it_done))) return rc;
    if (((it - it_done) & 15) == 15) {
      MB_HIP(hipMemcpyAsync(t->h_flags, flags, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
      MB_HIP(hipStreamSynchronize(s));
      if (t->h_flags[TF_DONE]) stopped = true;
    }
  }
  MB_HIP(hipEventRecord(t->ev_t1, s));
  MB_HIP(hipMemcpyAsync(t->h_flags, flags, sizeof(int) * 8, hipMemcpyDeviceToHost, s));
  MB_HIP(hipStreamSynchronize(s));
  *frames_out = t->h_flags[TF_NFRAMES];
  *lost_out = t->h_flags[TF_LOST];  // 0 | 1 (a hand-off timed out) | 2 (an operand left fp16's range)
  t->last_f16 = f16;
  t->last_form = fold ? 4 : front ? 5 : 7;
  t->last_iters = cdiv(*frames_out, r); t->timed = true;
  if (tr) {
    std::vector<unsigned long long> host((size_t)TS_WORDS);
    MB_HIP(hipMemcpy(host.data(), tr, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "wb")) { fwrite(host.data(), sizeof(unsigned long long), host.size(), f); fclose(f); }
  }
  return MB_OK;
}

static int taco_fast_loop(mb_taco* t, const TacoLayout& L, const float* d_memory, const float* d_memory_proj, const int32_t* d_chars,
                          int B, int T, int max_steps, float min_stop_token, const float* d_dropout, uint64_t seed, float* d_mel,
                          float* d_attn, bool lsa_fast, size_t lds_lsa, int psplit, void* d_workspace, hipStream_t caller, int* frames_out,
                          bool front, int* lost_out) {
  hipStream_t ls = t->loop_stream;
  MB_HIP(hipEventRecord(t->ev_in, caller));  // the caller's memsets / producers of memory first
  MB_HIP(hipStreamWaitEvent(ls, t->ev_in, 0));
  const int rc = taco_fast_loop_body(t, L, d_memory, d_memory_proj, d_chars, B, T, max_steps, min_stop_token, d_dropout, seed, d_mel,
                                     d_attn, lsa_fast, lds_lsa, psplit, d_workspace, ls, frames_out, front, lost_out);
  // success: the body ended with a host synchronisation of the loop stream (frame count), so whatever the caller
  // enqueues next is ordered behind the loop.  Failure: drain what was enqueued before handing the buffers back.
  if (rc) (void)hipStreamSynchronize(ls);
  return rc;
}

extern "C" int mb_taco_decode(const mb_taco* t, const float* d_memory, const float* d_memory_proj,
                              const int32_t* d_chars, int batch, int t_text, int max_steps, float min_stop_token,
                              const float* d_dropout, uint64_t seed, float* d_mel, float* d_linear, float* d_attn,
                              int* h_n_frames, void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  MB_REQUIRE(t && d_memory && d_memory_proj && d_chars && d_mel && d_linear && h_n_frames, "taco_decode: null pointer");
  MB_REQUIRE(batch > 0 && t_text > 0 && max_steps > 0, "taco_decode: empty input");
  TacoLayout L;
  taco_layout(t, batch, t_text, max_steps, d_workspace, &L);
  if (!d_workspace || workspace_bytes < L.bytes) {
    set_error("taco_decode: workspace %zu B < required %zu B", workspace_bytes, L.bytes);
    return MB_ENOMEM;
  }
  const mb_taco_config& c = t->cfg;
  const int B = batch, T = t_text, D = c.decoder_dims, P = c.project_dims, H = c.lstm_dims, M = c.n_mels, r = c.r;
  hipStream_t s = (hipStream_t)stream;
  const int n_iter_max = cdiv(max_steps, r);
  const int psplit = (P % 1024 == 0) ? 4 : 1;  // context column groups per utterance (pw must be a multiple of 256)
  MB_REQUIRE((P / psplit) % 256 == 0 && D <= 512 && 512 % D == 0, "taco_decode: unsupported project_dims/decoder_dims for the LSA kernel");
  const size_t lds_lsa = sizeof(float) * ((size_t)((T + c.lsa_kernel - 1 + 3) & ~3) + D + (size_t)c.lsa_filters * D +
                                          ((c.lsa_filters * c.lsa_kernel + 3) & ~3) + (size_t)T * c.lsa_filters + ((T + 3) & ~3) + 64 + 8 * 64 * 4);
  const bool lsa_fast = D == 128 && P / psplit == 256 && c.lsa_kernel <= 31 && (c.lsa_kernel & 1) && T <= 192 &&
                        getenv("MBHIP_LSA_GENERIC") == nullptr;
  if (!lsa_fast) {  // the general kernel keeps its location window in dynamic LDS: up to the 160 KB a gfx950 CU has
    MB_REQUIRE(lds_lsa <= 160 * 1024, "taco_decode: text too long for the LSA window in LDS (T=%d needs %zu B of 163840)", T, lds_lsa);
    if (lds_lsa > 48 * 1024)
      MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lsa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lsa));
  }

  if (lsa_fast) {  // processed memory as per-thread position quads (once per call)
    const int ntile = (T <= 128 ? 128 : 192) / 16;
    hipLaunchKernelGGL(lsa_pack_memproj_kernel, dim3(std::min(cdiv(B * ntile * 512, 256), 2048)), dim3(256), 0, s, d_memory_proj,
                       reinterpret_cast<float4*>(L.mpq4), B, T, ntile);
    MB_HIP(hipGetLastError());
  }
  // zero initial states (tacotron.py:219-230,261; lsa.py:15-19)
  auto zero_state = [&]() -> int {
    MB_HIP(hipMemsetAsync(L.attn_h, 0, sizeof(float) * 2 * B * D, s));
    MB_HIP(hipMemsetAsync(L.context, 0, sizeof(float) * 2 * B * P, s));
    MB_HIP(hipMemsetAsync(L.h1, 0, sizeof(float) * 2 * B * H, s)); MB_HIP(hipMemsetAsync(L.c1, 0, sizeof(float) * 2 * B * H, s));
    MB_HIP(hipMemsetAsync(L.h2, 0, sizeof(float) * 2 * B * H, s)); MB_HIP(hipMemsetAsync(L.c2, 0, sizeof(float) * 2 * B * H, s));
    MB_HIP(hipMemsetAsync(L.melstep, 0, sizeof(float) * B * r * M, s));  // <GO> frame
    MB_HIP(hipMemsetAsync(L.cumulative, 0, sizeof(float) * 2 * B * T, s));
    MB_HIP(hipMemsetAsync(L.flags, 0, sizeof(int) * 16, s));
    MB_HIP(hipMemsetAsync(d_mel, 0, sizeof(float) * (size_t)B * M * max_steps, s));
    if (d_attn) MB_HIP(hipMemsetAsync(d_attn, 0, sizeof(float) * (size_t)B * n_iter_max * T, s));
    return MB_OK;
  };
  { const int rz = zero_state(); if (rz) return rz; }
  int* done = L.flags;
  int* n_frames = L.flags + 1;
  int* arrive = L.flags + 2;

  // always-on PreNet dropout (pre_net.py:23,26) with the checkpoint's probability p: keep iff draw >= p * 2^32
  const bool drop_enabled = c.dropout > 0.f;
  const float drop_scale = drop_enabled ? 1.f / (1.f - c.dropout) : 1.f;
  const unsigned drop_thresh = drop_enabled ? (unsigned)std::min(4294967295.0, (double)c.dropout * 4294967296.0) : 0u;
  int frames = 0;
  // production dims: the FM-layout loop of taco_fast.h (MBHIP_TACO_FAST=0 forces the general loop below, which also
  // serves every other checkpoint shape)
  const char* fenv = getenv("MBHIP_TACO_FAST");
  const bool use_fast = t->fast && !(fenv && atoi(fenv) == 0);
  if (use_fast) {
    // launches 1..3 of an iteration as one (taco_front_kernel) when its 48 + B psplit chained workgroups fit the device side by side
    mb_taco* tm = const_cast<mb_taco*>(t);
    if (tm->n_cus < 0) {
      int dev = 0, ncu = 0;
      MB_HIP(hipGetDevice(&dev));
      MB_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
      // a compute-unit count alone does not say that an attention workgroup (128 KB of LDS) fits a compute unit of THIS device or
      // partition: ask the occupancy of the fused launch's instances once per handle (as wavernn.hip does for its resident
      // kernels) -- a device that cannot hold one workgroup per compute unit runs the 7-launch loop from the start
      int nb = 1, nb_min = 1 << 30;
#define TF_OCC(...)                                                                                                                     \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&taco_front_kernel<__VA_ARGS__>), 512, 0) != hipSuccess) nb = 0; \
      nb_min = std::min(nb_min, nb);
      TF_OCC(32, 2, true, true) TF_OCC(32, 1, true, true) TF_OCC(32, 2, false, false) TF_OCC(48, 2, true, false) TF_OCC(48, 2, false, false) TF_OCC(48, 1, false, false)
#undef TF_OCC
      (void)hipGetLastError();
      tm->n_cus = nb_min >= 1 ? ncu : 0;
    }
    bool front = taco_pick_form(B, T, lsa_fast ? 1 : 0, tm->n_cus, 1, tm->front_failed ? 1 : 0, diag_int("taco_front", -1), 0, 0, nullptr) != 7;
    for (;;) {
      int lost = 0;
      if (front && diag_int("taco_front_lost")) {  // tests: every wait of the fused launch bails out at its first clock check
        const int one = 1;
        MB_HIP(hipMemcpyAsync(L.flags + TF_LOST, &one, sizeof(int), hipMemcpyHostToDevice, s));
      }
      const int rcf = taco_fast_loop(tm, L, d_memory, d_memory_proj, d_chars, B, T, max_steps, min_stop_token,
                                     d_dropout, seed, d_mel, d_attn, lsa_fast, lds_lsa, psplit, d_workspace, s, &frames, front, &lost);
      if (rcf) return rcf;
      tm->last_front = front ? 1 : 0;
      if (!lost) break;
      // a hand-off inside the fused launch timed out (its workgroups were not co-resident: remembered, the handle stays on the 7-launch
      // loop) or an operand of the fp16-pipe products left fp16's range (this call only): the call again on the exact loop
      MB_REQUIRE(front, "taco_decode: lost hand-off flag without a fused launch");
      if (lost == 1 && !diag_int("taco_front_lost")) tm->front_failed = true;
      front = false;
      const int rz = zero_state();
      if (rz) return rz;
    }
  } else
  for (int it = 0; it < n_iter_max; ++it) {
    const int pp = it & 1;
    float* ah_p = L.attn_h + (size_t)pp * B * D; float* ah_n = L.attn_h + (size_t)(pp ^ 1) * B * D;
    float* cx_p = L.context + (size_t)pp * B * P; float* cx_n = L.context + (size_t)(pp ^ 1) * B * P;
    float* h1p = L.h1 + (size_t)pp * B * H; float* h1n = L.h1 + (size_t)(pp ^ 1) * B * H;
    float* c1p = L.c1 + (size_t)pp * B * H; float* c1n = L.c1 + (size_t)(pp ^ 1) * B * H;
    float* h2p = L.h2 + (size_t)pp * B * H; float* h2n = L.h2 + (size_t)(pp ^ 1) * B * H;
    float* c2p = L.c2 + (size_t)pp * B * H; float* c2n = L.c2 + (size_t)(pp ^ 1) * B * H;
    RnnK k;
    int rc;
    // prenet (pre_net.py:21-26): input = last frame of the previous iteration (tacotron.py:268)
    memset(&k, 0, sizeof(k));
    k.w = t->pre1_w.p; k.nseg = 1; k.nkb_total = M / 16; k.seg[0] = {L.melstep + (size_t)(r - 1) * M, r * M, M / 16, 0};
    k.N = B; k.units = 2 * D; k.biasX = t->pre1_b.p; k.y = L.p1; k.ldy = 2 * D; k.act = 1; k.mask_scale = drop_scale; k.drop_thresh = drop_thresh;
    k.mask = (d_dropout && drop_enabled) ? d_dropout + ((size_t)it * 2 + 0) * B * 2 * D : nullptr;
    k.drop_on = (d_dropout || !drop_enabled) ? 0 : 1; k.drop_seed = seed; k.drop_iter = it; k.drop_layer = 0; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LINEAR, k, s))) return rc;
    memset(&k, 0, sizeof(k));
    k.w = t->pre2_w.p; k.nseg = 1; k.nkb_total = 2 * D / 16; k.seg[0] = {L.p1, 2 * D, 2 * D / 16, 0};
    k.N = B; k.units = 2 * D; k.biasX = t->pre2_b.p; k.y = L.p2; k.ldy = 2 * D; k.act = 1; k.mask_scale = drop_scale; k.drop_thresh = drop_thresh;
    k.mask = (d_dropout && drop_enabled) ? d_dropout + ((size_t)it * 2 + 1) * B * 2 * D : nullptr;
    k.drop_on = (d_dropout || !drop_enabled) ? 0 : 1; k.drop_seed = seed; k.drop_iter = it; k.drop_layer = 1; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LINEAR, k, s))) return rc;
    // attn_hidden = attn_rnn([context, prenet_out], attn_hidden)  (tacotron.py:97-98)
    memset(&k, 0, sizeof(k));
    k.w = t->attn_w.p; k.nseg = 3; k.nkb_total = (P + 3 * D) / 16;
    k.seg[0] = {cx_p, P, P / 16, 0}; k.seg[1] = {L.p2, 2 * D, 2 * D / 16, 0}; k.seg[2] = {ah_p, D, D / 16, 1};
    k.N = B; k.units = D; k.biasX = t->attn_bih.p; k.biasH = t->attn_bhh.p; k.h_prev = ah_p; k.h_out = ah_n; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_GRU, k, s))) return rc;
    // scores = attn_net(...); context = scores @ encoder_seq  (tacotron.py:101-105)
    LsaK lk;
    lk.query = ah_n; lk.mem_proj = d_memory_proj; lk.memory = d_memory; lk.chars = d_chars;
    lk.cum_in = L.cumulative + (size_t)pp * B * T; lk.cum_out = L.cumulative + (size_t)(pp ^ 1) * B * T; lk.psplit = psplit;
    lk.conv_w = t->lsa_conv_w.p; lk.conv_b = t->lsa_conv_b.p; lk.Lw = t->lsa_L.p; lk.Ww = t->lsa_W.p; lk.Wb = t->lsa_Wb.p;
    lk.vw = t->lsa_v.p; lk.context = cx_n; lk.attn_out = d_attn; lk.T = T; lk.D = D; lk.P = P; lk.Fl = c.lsa_filters;
    lk.Kl = c.lsa_kernel; lk.iter = it; lk.n_iter_max = n_iter_max; lk.skip_flag = done;
    lk.Mt = t->lsa_Mt.p; lk.c0 = t->lsa_c0.p; lk.Wt = t->lsa_Wt.p; lk.fm_nta = 0; lk.iter_base = nullptr; lk.trace = nullptr;
    lk.Wq4 = reinterpret_cast<const float4*>(t->lsa_Wq4.p); lk.Mf4 = reinterpret_cast<const float4*>(t->lsa_Mq4.p);
    lk.mpf4 = reinterpret_cast<const float4*>(L.mpq4);
    if (lsa_fast && T <= 128) hipLaunchKernelGGL(lsa_fast_kernel<32>, dim3(B, psplit), dim3(512), 0, s, lk);
    else if (lsa_fast) hipLaunchKernelGGL(lsa_fast_kernel<48>, dim3(B, psplit), dim3(512), 0, s, lk);
    else hipLaunchKernelGGL(lsa_kernel, dim3(B, psplit), dim3(512), lds_lsa, s, lk);
    MB_HIP(hipGetLastError());
    // x = rnn_input([context, attn_hidden])  (tacotron.py:108-109)
    memset(&k, 0, sizeof(k));
    k.w = t->rin_w.p; k.nseg = 2; k.nkb_total = (P + D) / 16; k.seg[0] = {cx_n, P, P / 16, 0}; k.seg[1] = {ah_n, D, D / 16, 0};
    k.N = B; k.units = H; k.biasX = t->rin_b.p; k.y = L.x; k.ldy = H; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LINEAR, k, s))) return rc;
    // residual LSTMs (tacotron.py:112-125, eval branch)
    memset(&k, 0, sizeof(k));
    k.w = t->l1_w.p; k.nseg = 2; k.nkb_total = 2 * H / 16; k.seg[0] = {L.x, H, H / 16, 0}; k.seg[1] = {h1p, H, H / 16, 1};
    k.N = B; k.units = H; k.biasX = t->l1_bih.p; k.biasH = t->l1_bhh.p; k.c_prev = c1p; k.x_res = L.x;
    k.h_out = h1n; k.c_out = c1n; k.x_out = L.x1; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LSTM, k, s))) return rc;
    memset(&k, 0, sizeof(k));
    k.w = t->l2_w.p; k.nseg = 2; k.nkb_total = 2 * H / 16; k.seg[0] = {L.x1, H, H / 16, 0}; k.seg[1] = {h2p, H, H / 16, 1};
    k.N = B; k.units = H; k.biasX = t->l2_bih.p; k.biasH = t->l2_bhh.p; k.c_prev = c2p; k.x_res = L.x1;
    k.h_out = h2n; k.c_out = c2n; k.x_out = L.x2; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LSTM, k, s))) return rc;
    // mels = mel_proj(x)[:, :, :r]  (tacotron.py:128-129)
    memset(&k, 0, sizeof(k));
    k.w = t->mel_w.p; k.nseg = 1; k.nkb_total = H / 16; k.seg[0] = {L.x2, H, H / 16, 0};
    k.N = B; k.units = r * M; k.y = L.melstep; k.ldy = r * M; k.skip_flag = done;
    if ((rc = rnn_launch(EPI_LINEAR, k, s))) return rc;
    // stop token + stop rule + frame scatter
    FinK fk;
    fk.x2 = L.x2; fk.context = cx_n; fk.stop_w = t->stop_w.p; fk.stop_b = t->stop_b.p; fk.melstep = L.melstep;
    fk.mel_out = d_mel; fk.stop_out = L.stop; fk.done = done; fk.n_frames = n_frames; fk.arrive = arrive;
    fk.B = B; fk.H = H; fk.P = P; fk.M = M; fk.r = r; fk.max_steps = max_steps; fk.t0 = it * r; fk.min_stop_token = min_stop_token;
    hipLaunchKernelGGL(finalize_kernel, dim3(B), dim3(256), 0, s, fk);
    MB_HIP(hipGetLastError());
    // poll the stop flag every 16 iterations (the skipped launches in between are no-ops)
    if ((it & 15) == 15 || it == n_iter_max - 1) {
      int hflags[2] = {0, 0};
      MB_HIP(hipMemcpyAsync(hflags, L.flags, sizeof(hflags), hipMemcpyDeviceToHost, s));
      MB_HIP(hipStreamSynchronize(s));
      frames = hflags[1];
      if (hflags[0]) break;
    }
  }
  *h_n_frames = frames;
  const int F = frames;
  if (F <= 0) return MB_OK;

  // ---- postnet: CBHG(mel_outputs) -> post_proj  (tacotron.py:281-283, cbhg.py:40-78) ----
  MB_HIP(hipMemcpy2DAsync(L.melc, sizeof(float) * F, d_mel, sizeof(float) * max_steps, sizeof(float) * F, (size_t)B * M,
                          hipMemcpyDeviceToDevice, s));
  int rc = MB_OK;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  if (t->ev_p0) MB_HIP(hipEventRecord(t->ev_p0, s));
  if (t->post.tm.ok && L.cb.xt) {
    bool seq_tm = false;
    RC(cbhg_forward_tm(t->post, L.melc, B, F, L.cb, s, &seq_tm));
    if (seq_tm) {  // post_proj reads the scan's [F][B][C] output as B items of F rows (row stride B C), the result is turned back once
      const int C = t->post.ch;
      RC(run_tml(t->post_proj_tm, L.cb.seq_tm, B, F, L.cb.hwa, 0, nullptr, nullptr, s, C, B * C));
      RC(mb_f32_tm_to_cm(L.cb.hwa, L.linc, B, M, F, (mb_stream_t)s));
    } else {
      RC(run_conv(t->post_proj, L.cb.seq, B, F, L.linc, 0, 0, 0, nullptr, nullptr, 0, s));
    }
  } else {
    RC(cbhg_forward(t->post, L.melc, B, F, L.cb, s));
    RC(run_conv(t->post_proj, L.cb.seq, B, F, L.linc, 0, 0, 0, nullptr, nullptr, 0, s));
  }
  if (t->ev_p1) { MB_HIP(hipEventRecord(t->ev_p1, s)); t->post_timed = true; }
  if (!rc) {
    MB_HIP(hipMemsetAsync(d_linear, 0, sizeof(float) * (size_t)B * M * max_steps, s));
    MB_HIP(hipMemcpy2DAsync(d_linear, sizeof(float) * max_steps, L.linc, sizeof(float) * F, sizeof(float) * F, (size_t)B * M,
                            hipMemcpyDeviceToDevice, s));
  }
#undef RC
  return rc;
}

namespace {
struct EncLayout { float *xe, *p1, *p2, *projres, *style; CbhgWs cb; size_t bytes; };
void enc_layout(const mb_taco* t, int B, int T, void* base, EncLayout* L) {
  const mb_taco_config& c = t->cfg;
  Arena ar(base, (size_t)-1);
  L->xe = ar.take<float>((size_t)B * c.embed_dims * T);
  L->p1 = ar.take<float>((size_t)B * c.encoder_dims * T);
  L->p2 = ar.take<float>((size_t)B * c.encoder_dims * T);
  L->projres = ar.take<float>((size_t)B * T * c.decoder_dims);
  L->style = ar.take<float>((size_t)B * std::max(c.style_dims, 1));
  cbhg_take(ar, t->enc, B, T, &L->cb);
  L->bytes = ar.off + 256;
}
}  // namespace

extern "C" int mb_taco_last_loop_f16(const mb_taco* t) {
  if (!t || !t->timed) return -1;
  return t->last_f16 ? 1 : 0;
}

extern "C" int mb_taco_last_loop_form(const mb_taco* t) {
  if (!t || !t->timed) return -1;
  return t->last_form;
}

extern "C" int mb_taco_last_postnet_ms(const mb_taco* t, float* ms) {
  MB_REQUIRE(t && ms, "taco_last_postnet_ms: null pointer");
  if (!t->post_timed) { set_error("taco_last_postnet_ms: no decode with a postnet pass yet"); return MB_ESTATE; }
  MB_HIP(hipEventSynchronize(t->ev_p1));
  MB_HIP(hipEventElapsedTime(ms, t->ev_p0, t->ev_p1));
  return MB_OK;
}

extern "C" int mb_taco_last_loop_ms(const mb_taco* t, float* ms, int* iterations) {
  MB_REQUIRE(t && ms, "taco_last_loop_ms: null pointer");
  if (!t->timed) { set_error("taco_last_loop_ms: no decode on the fast loop yet"); return MB_ESTATE; }
  MB_HIP(hipEventElapsedTime(ms, t->ev_t0, t->ev_t1));
  if (iterations) *iterations = t->last_iters;
  return MB_OK;
}

extern "C" size_t mb_taco_encode_workspace_bytes(const mb_taco* t, int batch, int t_text) {
  if (!t || !t->cfg.has_encoder || batch <= 0 || t_text <= 0) return 0;
  EncLayout L;
  enc_layout(t, batch, t_text, nullptr, &L);
  return L.bytes;
}

extern "C" int mb_taco_encode(const mb_taco* t, const int32_t* d_chars, const float* d_speaker, int style_idx,
                              int batch, int t_text, const float* d_dropout, uint64_t seed,
                              float* d_memory, float* d_memory_proj, void* d_workspace, size_t workspace_bytes,
                              mb_stream_t stream) {
  MB_REQUIRE(t && d_chars && d_speaker && d_memory && d_memory_proj, "taco_encode: null pointer");
  MB_REQUIRE(t->cfg.has_encoder, "taco_encode: handle was created without encoder weights");
  MB_REQUIRE(batch > 0 && t_text > 0, "taco_encode: bad shape");
  MB_REQUIRE(t->cfg.style_dims == 0 || t->cfg.has_gst, "taco_encode: style_dims > 0 but the handle has no GST weights");
  EncLayout L;
  enc_layout(t, batch, t_text, d_workspace, &L);
  if (!d_workspace || workspace_bytes < L.bytes) {
    set_error("taco_encode: workspace %zu B < required %zu B", workspace_bytes, L.bytes);
    return MB_ENOMEM;
  }
  const mb_taco_config& c = t->cfg;
  const int B = batch, T = t_text, Ce = c.encoder_dims, Em = c.embed_dims, D = c.decoder_dims;
  hipStream_t s = (hipStream_t)stream;
  int rc = MB_OK;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  const bool e_drop = c.dropout > 0.f;
  const float e_scale = e_drop ? 1.f / (1.f - c.dropout) : 1.f;
  const unsigned e_thresh = e_drop ? (unsigned)std::min(4294967295.0, (double)c.dropout * 4294967296.0) : 0u;
  // x = embedding(texts); x = pre_net(x)  (tacotron.py:41-42, pre_net.py:21-26)
  hipLaunchKernelGGL(embed_gather_kernel, dim3(std::min(cdiv(Em * T, 256), 1024), B), dim3(256), 0, s, (const int*)d_chars,
                     t->emb.p, L.xe, T, Em, c.num_chars);
  RC(run_conv(t->enc_fc1, L.xe, B, T, L.p1, 0, 0, 1, nullptr, nullptr, 0, s));
  hipLaunchKernelGGL(dropout_cm_kernel, dim3(std::min(cdiv(Ce * T, 256), 1024), B), dim3(256), 0, s, L.p1,
                     d_dropout ? d_dropout : nullptr, Ce, T, (unsigned long long)seed, 0, e_thresh, e_scale);
  RC(run_conv(t->enc_fc2, L.p1, B, T, L.p2, 0, 0, 1, nullptr, nullptr, 0, s));
  hipLaunchKernelGGL(dropout_cm_kernel, dim3(std::min(cdiv(Ce * T, 256), 1024), B), dim3(256), 0, s, L.p2,
                     d_dropout ? d_dropout + (size_t)B * T * Ce : nullptr, Ce, T, (unsigned long long)seed, 1, e_thresh, e_scale);
  // x = cbhg(x)  (tacotron.py:43-44)
  if (t->enc.tm.ok) RC(cbhg_forward_tm(t->enc, L.p2, B, T, L.cb, s, nullptr, true));
  else RC(cbhg_forward(t->enc, L.p2, B, T, L.cb, s));
  // speaker + style concat (tacotron.py:171-197, 253) and encoder_proj (:255)
  if (!rc) {
    const size_t lds = sizeof(float) * (c.speaker_dims + c.style_dims + D);
    int style_batch = 1;
    if (c.has_gst) {
      const bool token = style_idx >= 0 && style_idx < c.gst_tokens;
      style_batch = token ? 1 : B;
      const size_t lg = sizeof(float) * (c.speaker_dims + c.style_dims + c.gst_heads * c.gst_tokens);
      hipLaunchKernelGGL(gst_style_kernel, dim3(style_batch), dim3(256), lg, s, d_speaker, t->gst_qconst.p, t->gst_WqS.p,
                         t->gst_K.p, t->gst_V.p, L.style, c.speaker_dims, c.style_dims, c.gst_tokens, c.gst_heads, style_idx);
    }
    hipLaunchKernelGGL(assemble_memory_kernel, dim3(B), dim3(256), lds, s, L.cb.seq, d_speaker, L.style, style_batch,
                       t->enc_proj_full.p, d_memory, L.projres, T, Ce, c.speaker_dims, c.style_dims, D);
    MB_HIP(hipGetLastError());
  }
  RC(run_conv(t->enc_proj, L.cb.seq, B, T, d_memory_proj, (long long)T * D, 0, 0, L.projres, nullptr, 1, s));
#undef RC
  return rc;
}
