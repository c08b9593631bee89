// Skinny recurrent GEMM  Y[N][rows] = epilogue( W[rows][K] . X[N][K]^T )  for
// the two autoregressive loops (WaveRNN sample loop, Tacotron decoder loop).
//
// Shape regime: N = folds / batch (1..32+), K = 512..2048, rows = 512..4096,
// re-launched every time step with the same weights -> bound by streaming W
// from L2 / Infinity Cache / HBM and by launch latency, not by FLOPs.
//
// Mapping (one workgroup = NW waves, one 16-row MFMA tile, 16 columns):
//   * rows are permuted on the host so that MFMA row i = unit*4 + gate: the
//     v_mfma_f32_16x16x4_f32 D fragment then gives every lane all gates of ONE
//     (hidden unit, column) pair -> the GRU / LSTM cell update is lane-local.
//   * the NW waves split K (16-wide blocks); each wave issues one coalesced
//     float4 weight load (A fragment) and one float4 activation load (B
//     fragment, activations are [N][K] so 4 consecutive k are contiguous) per
//     block and 4 MFMAs; partials meet in LDS (one ds_write_b128 per part).
//   * K is a concatenation of up to 4 activation segments, each tagged as the
//     cell's input part (X) or hidden part (H); GRU keeps X/H sums apart
//     because n = tanh(i_n + r*h_n) needs them separately.
//   * GRU rows carry 3 live gates per unit; the 4th MFMA row is never loaded.
#pragma once
#include "common.h"

namespace mb {

enum { EPI_LINEAR = 0, EPI_GRU = 1, EPI_LSTM = 2 };

struct RnnSeg {
  const float* p;  // [N][ld]
  int ld;
  int nkb;   // 16-wide k blocks in this segment
  int part;  // 0 = X (input), 1 = H (hidden)
};

struct RnnK {
  const float* w;  // packed by pack_rowtile
  RnnSeg seg[4];
  int nseg, nkb_total;
  int N;      // live columns
  int units;  // hidden units (GRU/LSTM) or output rows (LINEAR)
  const float* biasX;  // GRU/LSTM: [gates*units] torch gate-major order; LINEAR: [rows] (may be null)
  const float* biasH;  // GRU/LSTM hidden bias (may be null)
  const float* pre_table;  // optional additive X-part rows: pre_table[pre_idx[n]*pre_stride + gate*units + unit]
  const int* pre_idx;      // null -> row pre_base_row + n*pre_n_stride
  int pre_stride;
  int pre_base_row, pre_n_stride;
  const float* h_prev; const float* c_prev; const float* x_res;  // [N][units]
  const float* h_pre;  // GRU, optional: precomputed hidden-part pre-activations [N][3*units] gate-major (W_hh.h + b_hh);
                       // the K segments then hold the input part only and biasH must be null
  float* h_out; float* c_out; float* x_out;                      // [N][units]
  float* y; int ldy; int act;  // LINEAR: y[n*ldy + row]; act 0 none, 1 relu, 2 sigmoid, 3 tanh
  const float* mask; float mask_scale;  // LINEAR: optional y *= mask[n*ldy+row]*mask_scale (dropout)
  // Table row computed in-kernel from a step index (WaveRNN fold geometry, fatchord_version.py:334-336):
  //   s = *fr_base + fr_off;  pos = (fr_n_off + n)*fr_fold_stride + s;
  //   row = pos < fr_total_len ? pos / fr_hop : fr_frames          (used when fr_base != null)
  // *fr_base changes once per graph replay, fr_off is baked per launch: the loop kernels never read a
  // word the previous launch has just written except the activations themselves.
  const int* fr_base; int fr_off, fr_n_off, fr_fold_stride, fr_total_len, fr_hop, fr_frames;
  // Several utterances in one loop (mb_wavernn_generate_batch): per-fold descriptors replace the arithmetic
  // above.  fr_desc[(fr_n_off + n) * 8 + ..] = {pos0, total_len_u, first row of the utterance in the U tables (WfCond), frame_row_base, frames_u,
  // local fold index, seed lo, seed hi}:  pos = pos0 + s (zero conditioning from total_len_u on);
  // frame-table row = frame_row_base + (pos < total_len_u ? pos / fr_hop : frames_u); the Gumbel noise of fold n
  // is Philox(seed_u; s, local fold, class/4) -- exactly what utterance u alone with seed_u would draw.
  const int* fr_desc;
  const int* skip_flag;  // if non-null and *skip_flag != 0 the launch is a no-op (decoder stop rule)
  // optional strided copy of h_out into a sequence tensor: seq_out[n*seq_n_stride + j*seq_j_stride + seq_off]
  float* seq_out; long long seq_n_stride, seq_j_stride, seq_off;
  // LINEAR dropout when mask == null and drop_seed_on: keep = philox(seed, drop_iter, drop_layer; n,row) < 0.5
  // (keep iff the 32-bit draw >= drop_thresh = p * 2^32; the kept value is scaled by mask_scale = 1/(1-p))
  int drop_on; unsigned long long drop_seed; int drop_iter, drop_layer; unsigned int drop_thresh;
  // ---- WaveRNN fused sampling (production path, no injected noise) ----
  // aff_slot != null (GRU): the cell input is x0[n] = aff_table[ipos_n] + x_n * aff_vec  (I(x) split,
  // wavernn.hip header) with ipos_n from the fr_* geometry (row fr_total_len = zero-conditioning row) and
  // x_n decoded from the argmax word aff_slot[n] the previous step's fc3 launch left (0 = no sample yet
  // -> x = 0).  K segment 0 then reads table row ipos_n for column n and the epilogue adds
  // x_n * aff_gate (aff_gate = W_ih . aff_vec, [3*units] gate-major); x_res is rebuilt the same way.
  // Workgroups with blockIdx.x == 0 also store x_n to aff_samples[(fr_n_off+n)*aff_S + s-1].
  const unsigned long long* aff_slot; const float* aff_table; const float* aff_vec; const float* aff_gate;
  int aff_ld, aff_C, aff_S;
  float* aff_samples; volatile int* aff_progress;
  // gum_slot != null (LINEAR, rows = classes): instead of (only) storing logits, draw Exp(1) noise from
  // Philox(gum_seed; s, fold, class/4) and atomicMax the packed (logit - log E, class) into gum_slot[n]:
  // argmax_c p_c/E_c == argmax_c (l_c - log E_c)  (torch.multinomial's rule).
  unsigned long long* gum_slot; unsigned long long gum_seed;
  // zero_slot != null: workgroup (0,0) clears zero_slot[0..N) (the argmax words of the NEXT step)
  unsigned long long* zero_slot;
  // diagnostics (MBHIP_DIAG=trace_file=<file>): per-workgroup (start, end) wall_clock64 ticks, TRACE_SLOTS pairs
  unsigned long long* trace;
  // wide batches (rnn_ts3_body.h): the same matrix as fp16 hi / lo A fragments of v_mfma_f32_16x16x32_f16
  // ([row tile][k-step of 32][hi | lo][lane][8], wavernn_pipe16.h wq16_pack) scaled by 2^s, and 2^-s; null -> the fp32 forms
  const void* w16; float w16_unscale;
  int* range_word;  // rnn_ts3_body: raised when a staged activation is beyond the operand pairs' range (|x| > 65504 or NaN); may be null
  int dbg;  // diagnostics (MBHIP_DIAG=ts3_dbg=<bits>, results are WRONG on purpose): 1 = rnn_ts3_body skips its k loop, 2 = skips its epilogues
};

// order-preserving float -> uint key and the packed (key, lowest-class-wins) argmax word
__host__ __device__ __forceinline__ unsigned int float_key(float f) {
  const unsigned int b = __builtin_bit_cast(unsigned int, f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ unsigned long long pack_argmax(float score, int cls) {
  return ((unsigned long long)float_key(score) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)cls);
}
__host__ __device__ __forceinline__ int argmax_class(unsigned long long w) { return (int)(0xFFFFFFFFu - (unsigned)w); }

static constexpr int TRACE_SLOTS = 512;  // workgroups recorded per launch
#ifdef MB_TRACE_MARKS
// diagnostics build only (tools/build_diag.sh): shader-clock marks inside workgroup 0 / wave 0
#define MB_MARK(tr, k, waitall)                                                                     \
  do {                                                                                              \
    if (waitall) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
    if ((tr) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)                             \
      (tr)[2 * 496 + (k)] = (unsigned long long)clock64();                                          \
  } while (0)
#else
#define MB_MARK(tr, k, waitall) do { } while (0)
#endif
__device__ __forceinline__ void trace_begin(unsigned long long* tr) {
  if (tr && threadIdx.x == 0) {
    const unsigned b = blockIdx.x + blockIdx.y * gridDim.x;
    if (b < TRACE_SLOTS) tr[2 * b] = (unsigned long long)wall_clock64();
    if (b == 0) tr[2 * (TRACE_SLOTS - 1)] = (unsigned long long)clock64();  // shader-clock ticks (DVFS probe)
  }
}
__device__ __forceinline__ void trace_end(unsigned long long* tr) {
  if (tr && (threadIdx.x & 63) == 0) {
    const unsigned b = blockIdx.x + blockIdx.y * gridDim.x;
    if (b < TRACE_SLOTS) atomicMax(tr + 2 * b + 1, (unsigned long long)wall_clock64());
    if (b == 0 && threadIdx.x == 0) tr[2 * (TRACE_SLOTS - 1) + 1] = (unsigned long long)clock64();
  }
}

// rows: live rows x K (K multiple of 16), tile-ordered: rows of tile mt are
// [mt*4*RL, (mt+1)*4*RL) in (unit, gate) order.  RL = live gates per unit (3 GRU, 4 else).
void pack_rowtile(const float* rows, int n_live_rows, int K, int RL, std::vector<float>* out);

// Build tile-ordered rows for a GRU/LSTM cell from torch weight_ih [G*H][Kx], weight_hh [G*H][H].
void cell_rows(const float* w_ih, int kx, int ldx, const float* w_hh, int kh, int H, int G,
               std::vector<float>* rows);

// Per-FRAME conditioning tables of the WaveRNN loop (round 3: they replace the per-position Ipre / T1 tables, 8.2 KB per output
// sample -> 8 KB per mel frame).  I([x, m_t, a1_t]) is affine in the conditioning; the upsampled mel m_t (UpsampleNetwork: three
// stretch + box-filter stages, fatchord_version.py:60-85) is a fixed linear combination of 5 neighbouring mel frames with weights
// that depend on t mod hop only (Kw[hop][5], computed at create time by pushing an impulse through the stages); a1_t is constant
// over a frame.  So for position t = hop f + p:
//   Ipre[t] = AI[f] + sum_o Kw[p][o] UI[f + o - 2]     UI[f] = W_I[:, 1:1+feat] mel_f,   AI[f] = W_I[:, 1+feat:] a1_f + b_I
//   T1[t]   = AT[f] + sum_o Kw[p][o] UT[f + o - 2]     (the same with W_ih1 folded in, gate-major rows)
// evaluated as ONE fmaf chain (aux/bias term first, then o = 0..4) by every consumer: all loop variants see the same bits.
// Rows: U* [frames + 9] (2 zero rows, the frames, 7 zero rows: fold_with_overlap's zero padding and the sequence ends),
//       A* [frames + 1] (row `frames` = the zero-conditioning row: the bias alone).
struct WfCond {
  const float* UT; const float* AT; const float* UI; const float* AI; const float* Kw;
  int hop, frames;
};
// (T1[pos][r, z, n], Ipre[pos]) of unit j; pos >= total_len = zero conditioning.  u_row0 / a_row0: first row of this utterance's
// block in concatenated tables (batch loop).
// the same from (frame f, phase p) of a live position (resident kernels advance them step by step instead of dividing by the hop)
__device__ __forceinline__ float4 wf_cond_row4_fp(const WfCond& c, const unsigned f_live, const unsigned p_live, const bool live, const int j, const int H,
                                                  const int frames, const long long u_row0 = 0, const long long a_row0 = 0) {
  const unsigned f = live ? f_live : (unsigned)frames + 4u;   // dead: five zero rows
  const unsigned p = live ? p_live : 0u;
  const long long fa = a_row0 + (live ? (long long)f : (long long)frames);
  const float* kw = c.Kw + p * 5;
  const float* at = c.AT + fa * 3 * H + j;
  float4 acc = make_float4(at[0], at[H], at[2 * H], c.AI[fa * H + j]);
  const float* ut = c.UT + (u_row0 + f) * 3 * H + j;   // row f = frame f - 2
  const float* ui = c.UI + (u_row0 + f) * H + j;
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    const float k = kw[o];
    acc.x = fmaf(k, ut[(size_t)o * 3 * H], acc.x);
    acc.y = fmaf(k, ut[(size_t)o * 3 * H + H], acc.y);
    acc.z = fmaf(k, ut[(size_t)o * 3 * H + 2 * H], acc.z);
    acc.w = fmaf(k, ui[(size_t)o * H], acc.w);
  }
  return acc;
}
// ... and in two halves: the 29 loads now, the fmaf chain later (a resident workgroup has other work to put in between); same
// expression, same bits
struct WfCondRaw { float at[3], ai, kw[5], ut[5][3], ui[5]; };
__device__ __forceinline__ void wf_cond_load(const WfCond& c, const unsigned f_live, const unsigned p_live, const bool live, const int j, const int H,
                                             const int frames, WfCondRaw& r) {
  const unsigned f = live ? f_live : (unsigned)frames + 4u;
  const unsigned p = live ? p_live : 0u;
  const long long fa = live ? (long long)f : (long long)frames;
  const float* kw = c.Kw + p * 5;
  const float* at = c.AT + fa * 3 * H + j;
  r.at[0] = at[0]; r.at[1] = at[H]; r.at[2] = at[2 * H]; r.ai = c.AI[fa * H + j];
  const float* ut = c.UT + (long long)f * 3 * H + j;
  const float* ui = c.UI + (long long)f * H + j;
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    r.kw[o] = kw[o];
    r.ut[o][0] = ut[(size_t)o * 3 * H]; r.ut[o][1] = ut[(size_t)o * 3 * H + H]; r.ut[o][2] = ut[(size_t)o * 3 * H + 2 * H];
    r.ui[o] = ui[(size_t)o * H];
  }
}
__device__ __forceinline__ float4 wf_cond_fma(const WfCondRaw& r) {
  float4 acc = make_float4(r.at[0], r.at[1], r.at[2], r.ai);
#pragma unroll
  for (int o = 0; o < 5; ++o) {
    acc.x = fmaf(r.kw[o], r.ut[o][0], acc.x);
    acc.y = fmaf(r.kw[o], r.ut[o][1], acc.y);
    acc.z = fmaf(r.kw[o], r.ut[o][2], acc.z);
    acc.w = fmaf(r.kw[o], r.ui[o], acc.w);
  }
  return acc;
}
__device__ __forceinline__ float4 wf_cond_row4(const WfCond& c, const unsigned pos, const unsigned total_len, const int j, const int H,
                                               const int frames, const long long u_row0 = 0, const long long a_row0 = 0) {
  const bool live = pos < total_len;
  const unsigned f = live ? pos / (unsigned)c.hop : 0u;
  return wf_cond_row4_fp(c, f, live ? pos - f * (unsigned)c.hop : 0u, live, j, H, frames, u_row0, a_row0);
}
// Ipre[pos][j] alone (the exact 6-launch chain feeds I(..) itself to rnn1)
__device__ __forceinline__ float wf_cond_ipre(const WfCond& c, const unsigned pos, const unsigned total_len, const int j, const int H,
                                              const int frames) {
  const bool live = pos < total_len;
  const unsigned f = live ? pos / (unsigned)c.hop : (unsigned)frames + 4u;
  const unsigned p = live ? pos - f * (unsigned)c.hop : 0u;
  const float* kw = c.Kw + p * 5;
  float acc = c.AI[(size_t)(live ? f : (unsigned)frames) * H + j];
  const float* ui = c.UI + (size_t)f * H + j;
#pragma unroll
  for (int o = 0; o < 5; ++o) acc = fmaf(kw[o], ui[(size_t)o * H], acc);
  return acc;
}

// WaveRNN split-hidden chain, rnn1 of a step as an ELEMENTWISE job (wavernn.hip header):
//   h1 = GRUCell(I([x, m_t, a1_t]), h1); x1 = I(..) + h1   (fatchord_version.py:195-198) with
//   i_g = T1[pos][g] + x * g1[g]      T1 = W_ih1.(W_I[:,1:].[m;a1] + b_I) + b_ih1,  g1 = W_ih1.W_I[:,0]
//   h_g = P1[n][g]                    W_hh1.h1 + b_hh1, left by the previous step's fc1 launch
// x is decoded from the argmax word of the previous step's fc3 launch (0 = no sample yet -> x = 0).
struct Fin1K {
  const unsigned long long* slot;  // [nl]
  WfCond cond; const float* P1; const float* g1; const float* wI0; const float* h_prev;
  float* h_out; float* x_out; float* samples; volatile int* progress;
  const int* step_base; int step_off, n_off, nl, R, C, S, fold_stride, total_len;
  const int* desc;  // optional per-fold descriptors (RnnK::fr_desc layout)
  unsigned long long* trace;
};
int rnn_launch(int epi, const RnnK& k, hipStream_t s);
// Forward and backward step of a bidirectional GRU scan in one launch (falls back to two launches for
// shapes outside the instantiated scan instance)
int rnn_launch_dual_gru(const RnnK& k0, const RnnK& k1, hipStream_t s);
// gru1-finish launch (the elementwise rnn1 of the split-hidden chains)
int rnn_launch_finish(const Fin1K& f, hipStream_t s);
// Two LINEAR jobs in one launch (job 0 on the critical path gets the first workgroups); see rnn.hip.
int rnn_launch_dual_linear(const RnnK& k0, const RnnK& k1, hipStream_t s);
// MBHIP_RNN_WIDE is unset or one of ts3 | ts2 | ts, optionally :1..3 (callers reject anything else up front)
bool rnn_wide_switch_valid();

}  // namespace mb
