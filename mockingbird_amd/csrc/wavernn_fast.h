// WaveRNN sample loop, production shape (rnn 512, fc 512, up to 64 fold columns of ONE utterance): the five
// launches of the split-hidden chain (wavernn.hip header) on fragment-major activations (taco_fast.h):
//   A  gru1 finish   elementwise rnn1 from the argmax word, T1 / Ipre table rows and the hidden half P1
//   B  rnn2          GRU on its input half (K = 512), hidden half P2, per-frame table G2
//   C  fc1 || hh1    relu(fc1 . x + F1[frame])  beside  P1 = W_hh1 . h1 + b_hh1 for the next step
//   D  fc2 || hh2    relu(fc2 . x + F2[frame])  beside  P2 = W_hh2 . h2 + b_hh2
//   E  fc3 + Gumbel-argmax sampler
// Same arithmetic, the same k split (k-block kb belongs to wave kb mod 8, partials added in wave order) and the same
// expression order as the rnn_rowtile_body instances they replace, so the sample stream is bit-identical to that chain
// (and to the per-utterance streams of mb_wavernn_generate_batch, which stays on those instances).  What changes is how
// the operands move: B fragments are contiguous 1 KB reads of FM buffers instead of 16-byte pieces of [fold][K] rows,
// every fragment of a wave is in flight before its first MFMA, a workgroup covers both fold-column tiles with one
// weight fetch, and the cell outputs are stored as contiguous pieces of the next launch's operand.
// Reference: models/vocoder/wavernn/models/fatchord_version.py:190-228.
#pragma once
#include "fm_gemm.h"

namespace mb {

struct WfGeom {  // fold geometry of one utterance (fatchord_version.py:334-336) + step counter
  const int* step_base; int step_off;  // step s = *step_base + step_off (the word changes once per graph replay)
  int fold_stride, total_len, hop, frames;
  int nta, N;  // column tiles, live folds
};
// conditioning-sequence position of fold n at step s (clamped to the zero-conditioning row) and its per-frame table row
__device__ __forceinline__ unsigned wf_pos(const WfGeom& g, int n, int s) {
  const unsigned pos = (unsigned)n * (unsigned)g.fold_stride + (unsigned)s;
  return pos > (unsigned)g.total_len ? (unsigned)g.total_len : pos;
}
__device__ __forceinline__ int wf_frame_row(const WfGeom& g, int n, int s) {
  const unsigned pos = (unsigned)n * (unsigned)g.fold_stride + (unsigned)s;
  return pos < (unsigned)g.total_len ? (int)(pos / (unsigned)g.hop) : g.frames;
}

// The T1 / Ipre rows of a step do not depend on the sample: they are rebuilt from the per-frame tables (rnn.h WfCond: 24 scattered
// reads + 20 FMAs per (unit, column)) into a dense CM4 block one launch ahead, by a job that rides in the FINISH launch of the
// previous step (the shortest launch of the chain: elementwise work only).
struct WfStageK { WfCond cond; float4* Tq; int R, step_add; };
__device__ __forceinline__ void wf_stage_rows(const WfStageK& a, const WfGeom& g, const int wg_index) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = wg_index * (blockDim.x >> 6) + wave;  // (row tile of 4 units, column tile)
  const int mt = item / g.nta, nt = item - mt * g.nta;
  if (mt >= a.R / 4) return;
  const int du = lane >> 4, i = lane & 15, j = mt * 4 + du, H = a.R;
  const int n_raw = nt * 16 + i, n = n_raw < g.N ? n_raw : g.N - 1;
  const int s = *g.step_base + g.step_off + a.step_add;
  a.Tq[((size_t)mt * g.nta + nt) * 64 + lane] = wf_cond_row4(a.cond, wf_pos(g, n, s), (unsigned)g.total_len, j, H, g.frames);
}
__global__ __launch_bounds__(512) void wf_stage_kernel(WfStageK a, WfGeom g) { wf_stage_rows(a, g, blockIdx.x); }

// ---------------------------------------------------------------------------------------------- A: gru1 finish
//   h1 = GRUCell(I([x, m_t, a1_t]), h1); x1 = I(..) + h1   (fatchord_version.py:195-198) with
//   i_g = T1[pos][g] + x * g1[g],  h_g = P1 (CM4),  x decoded from the previous step's argmax word.
struct WfFinK {
  WfGeom g;
  const unsigned long long* slot;  // [N]
  const float4* Tq;  // (T1[pos][r,z,n], Ipre[pos]) of THIS step per (unit, column), staged a launch earlier (wf_stage_rows)
  const float4* P1; const float* g1; const float* wI0;
  float* h1; float* x1;            // FM (h1 updated in place)
  float* samples; volatile int* progress;
  int R, C, S;
  int mol;  // MOL mode: the slot holds the sample itself (wf_fc3_mol_kernel), not a packed argmax
  WfStageK stage; int n_fin;  // blockIdx.x >= n_fin (blockIdx.y == 0): stage the NEXT step's table rows into stage.Tq (other buffer)
};
__global__ __launch_bounds__(256) void wf_finish_kernel(WfFinK a) {
  if ((int)blockIdx.x >= a.n_fin) {
    if (blockIdx.y == 0) wf_stage_rows(a.stage, a.g, blockIdx.x - a.n_fin);
    return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = blockIdx.x * 4 + wave, nt = blockIdx.y;
  const int du = lane >> 4, i = lane & 15, j = mt * 4 + du, H = a.R;
  const int n_raw = nt * 16 + i, n = n_raw < a.g.N ? n_raw : a.g.N - 1;  // clamped: loads legal, nothing stored for dead columns
  const int s = *a.g.step_base + a.g.step_off;
  const float4 hq = a.P1[((size_t)mt * a.g.nta + nt) * 64 + lane];
  const size_t fo = ((size_t)(mt >> 2) * a.g.nta + nt) * 256 + (mt & 3) * 64 + i * 4 + du;
  const float hp = a.h1[fo];
  const float4 tq = a.Tq[((size_t)mt * a.g.nta + nt) * 64 + lane];
  const float tr = tq.x, tz = tq.y, tn = tq.z, ip = tq.w;
  const float gr = a.g1[j], gz = a.g1[H + j], gn = a.g1[2 * H + j], w0 = a.wI0[j];
  const unsigned long long slot = a.slot[n];
  if (n_raw >= a.g.N) return;
  const float x = !slot ? 0.f : a.mol ? __uint_as_float((unsigned)slot) : 2.f * (float)argmax_class(slot) / ((float)a.C - 1.f) - 1.f;
  // torch GRUCell, gate order (r, z, n)
  const float rg = sigmoidf_((tr + x * gr) + hq.x);
  const float zg = sigmoidf_((tz + x * gz) + hq.y);
  const float ng = tanhf((tn + x * gn) + rg * hq.z);
  const float hy = ng + zg * (hp - ng);
  a.h1[fo] = hy;
  a.x1[fo] = (ip + x * w0) + hy;
  if (j == 0 && s > 0) {  // previous step's sample -> output tensor
    a.samples[(size_t)n * a.S + (s - 1)] = x;
    if (a.progress && n == 0 && (s - 1) % 100 == 0) *a.progress = s;
  }
}

// ---------------------------------------------------------------------------------------------- B: rnn2
//   h2 = GRUCell([x, a2_t], h2); x = x + h2   (:199-202): input half on x (K = R), aux/bias part from the per-frame
//   table G2, hidden half P2 (CM4).  Workgroup (0, 0) also clears the argmax words the NEXT step's fc3 will fill.
struct WfRnn2K {
  WfGeom g;
  const float* w; const float* x1; const float4* P2; const float* G2;  // G2 [frames + 1][3R] gate-major
  float* h2; float* x2;  // FM (h2 in place)
  unsigned long long* zero_slot;
  int R;
};
template <int NT>
__global__ __launch_bounds__(512) void wf_rnn2_kernel(WfRnn2K a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int mt = blockIdx.x, nt0 = blockIdx.y * NT, H = a.R;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15, j = mt * 4 + du;
  const int ntE = (nt0 + (wv < NT ? wv : 0) < a.g.nta) ? nt0 + (wv < NT ? wv : 0) : a.g.nta - 1;
  const int n_raw = ntE * 16 + i, n = n_raw < a.g.N ? n_raw : a.g.N - 1;
  const int s = *a.g.step_base + a.g.step_off;
  const float4 hq = a.P2[((size_t)mt * a.g.nta + ntE) * 64 + lane];
  const size_t fo = ((size_t)(mt >> 2) * a.g.nta + ntE) * 256 + (mt & 3) * 64 + i * 4 + du;
  const float hp = a.h2[fo], xr = a.x1[fo];
  const float* gp = a.G2 + (size_t)wf_frame_row(a.g, n, s) * 3 * H + j;
  const float pr = gp[0], pz = gp[H], pn = gp[2 * H];
  float sx[4], sh[4];
  if (!fm_gemm<NT, 4, 4, 3, 1>(a.w, mt, a.x1, a.x1, a.g.nta, nt0, red, sx, sh)) return;
  if (mt == 0 && du == 0 && nt0 + wv < a.g.nta && n_raw < a.g.N) a.zero_slot[n_raw] = 0ull;
  if (nt0 + wv >= a.g.nta || n_raw >= a.g.N) return;
  const float rg = sigmoidf_((sx[0] + pr) + hq.x);
  const float zg = sigmoidf_((sx[1] + pz) + hq.y);
  const float ng = tanhf((sx[2] + pn) + rg * hq.z);
  const float hy = ng + zg * (hp - ng);
  a.h2[fo] = hy;
  a.x2[fo] = xr + hy;
}

// ---------------------------------------------------------------------------------------------- C / D: fc || hidden half
//   job 0 (blockIdx.x < n_fc): y = relu(fc . x + F[frame])   (:203-207; aux columns and bias folded into the table F)
//   job 1: P = W_hh . h + b_hh of the NEXT step's GRU, GRU tile order -> CM4 (r, z, n, -)
struct WfFcHhK {
  WfGeom g;
  const float* w_fc; const float* xin; const float* F; float* y; int n_fc, FC;  // F [frames + 1][FC]
  const float* w_hh; const float* h; const float4* bhh4; float4* P;
};
template <int NT>
__global__ __launch_bounds__(512) void wf_fc_hh_kernel(WfFcHhK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int nt0 = blockIdx.y * NT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  float sx[4], sh[4];
  if ((int)blockIdx.x < a.n_fc) {
    const int mt = blockIdx.x;
    const int ntE = (nt0 + (wv < NT ? wv : 0) < a.g.nta) ? nt0 + (wv < NT ? wv : 0) : a.g.nta - 1;
    const int n_raw = ntE * 16 + i, n = n_raw < a.g.N ? n_raw : a.g.N - 1;
    const int s = *a.g.step_base + a.g.step_off;
    const float4 pre = *reinterpret_cast<const float4*>(a.F + (size_t)wf_frame_row(a.g, n, s) * a.FC + mt * 16 + du * 4);
    if (!fm_gemm<NT, 4, 4, 4, 1>(a.w_fc, mt, a.xin, a.xin, a.g.nta, nt0, red, sx, sh)) return;
    if (nt0 + wv >= a.g.nta) return;
    reinterpret_cast<float4*>(a.y)[((size_t)mt * a.g.nta + nt0 + wv) * 64 + lane] =
        make_float4(fmaxf(sx[0] + pre.x, 0.f), fmaxf(sx[1] + pre.y, 0.f), fmaxf(sx[2] + pre.z, 0.f), fmaxf(sx[3] + pre.w, 0.f));
    return;
  }
  const int mt = blockIdx.x - a.n_fc;
  const float4 bq = a.bhh4[mt * 4 + du];
  if (!fm_gemm<NT, 4, 4, 3, 1>(a.w_hh, mt, a.h, a.h, a.g.nta, nt0, red, sx, sh)) return;
  if (nt0 + wv >= a.g.nta) return;
  a.P[((size_t)mt * a.g.nta + nt0 + wv) * 64 + lane] = make_float4(sx[0] + bq.x, sx[1] + bq.y, sx[2] + bq.z, 0.f);
}

// ---------------------------------------------------------------------------------------------- E: fc3 + sampler
//   logits = fc3(x) (:209); Categorical(softmax(logits)).sample() (:222-226) as Gumbel-argmax: draw Exp(1) noise from
//   Philox(seed; step, fold, class / 4) and atomicMax the packed (logit - log E, class) into slot[fold]
//   (argmax_c p_c / E_c == argmax_c (l_c - log E_c): torch.multinomial's rule).
struct WfFc3K {
  WfGeom g;
  const float* w; const float* bias; const float* xin; unsigned long long* slot; unsigned long long seed; int C;
};
template <int NT>
__global__ __launch_bounds__(512) void wf_fc3_kernel(WfFc3K a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  const int mt = blockIdx.x, nt0 = blockIdx.y * NT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  const int s = *a.g.step_base + a.g.step_off;
  const float4 bq = *reinterpret_cast<const float4*>(a.bias + mt * 16 + du * 4);
  float sx[4], sh[4];
  if (!fm_gemm<NT, 4, 4, 4, 1>(a.w, mt, a.xin, a.xin, a.g.nta, nt0, red, sx, sh)) return;
  const int nt = nt0 + wv, n = nt * 16 + i;
  if (nt >= a.g.nta || n >= a.g.N) return;  // (whole 16-lane groups drop out together: the shuffles below stay within live groups' columns)
  uint32_t gr[4];
  philox4x32((uint32_t)s, (uint32_t)n, (uint32_t)((mt * 16 + du * 4) >> 2), 0x57415645u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), gr);
  const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
  float best = -INFINITY;
  int bcls = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = mt * 16 + du * 4 + r;
    const float v = sx[r] + bv[r];
    const float gmb = v - logf(-logf(u32_to_unit(gr[r])));
    if (gmb > best) { best = gmb; bcls = row; }  // ascending rows: first maximum kept
  }
  // the 4 row quads of this column sit in lanes l, l+16, l+32, l+48
  unsigned long long pk = pack_argmax(best, bcls);
  const unsigned long long o1 = __shfl_xor(pk, 16, 64);
  pk = o1 > pk ? o1 : pk;
  const unsigned long long o2 = __shfl_xor(pk, 32, 64);
  pk = o2 > pk ? o2 : pk;
  if (du == 0) atomicMax(a.slot + n, pk);
}

// ---------------------------------------------------------------------------------------------- E (MOL mode): fc3 + mixture-of-logistics sampler
//   fc3 has 3 * nr_mix = 30 outputs (mixture logits | means | log scales; fatchord_version.py:95-98, :213-220): ONE workgroup
//   per column-tile group multiplies both 16-row tiles, leaves the 30 values of every column in LDS, and one lane per column
//   runs sample_from_discretized_mix_logistic (models/vocoder/distribution.py:87-123) with the draws and the expressions of
//   wavernn_sample_mol_kernel (wavernn.hip) -- the stand-alone sampler of the 6-launch path, which it equals bit for bit.
//   slot[fold] = marker << 32 | float bits of the sample (the finish launch feeds it back: mol = 1).
struct WfFc3MolK {
  WfGeom g;
  const float* w; const float* bias; const float* xin; unsigned long long* slot; unsigned long long seed; int C, nr_mix;
};
template <int NT>
__global__ __launch_bounds__(512) void wf_fc3_mol_kernel(WfFc3MolK a) {
  __shared__ __attribute__((aligned(16))) float red[FmRed<NT, 1>::floats];
  __shared__ float lg[NT * 16][33];
  const int nt0 = blockIdx.y * NT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, du = lane >> 4, i = lane & 15;
  const int s = *a.g.step_base + a.g.step_off;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    float sx[4], sh[4];
    if (fm_gemm<NT, 4, 4, 4, 1>(a.w, mt, a.xin, a.xin, a.g.nta, nt0, red, sx, sh)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + du * 4 + r;
        lg[wv * 16 + i][row] = sx[r] + (row < a.C ? a.bias[row] : 0.f);
      }
    }
    __syncthreads();  // red is free for the second tile; lg complete after it
  }
  if ((int)threadIdx.x >= NT * 16) return;
  const int nt = nt0 + ((int)threadIdx.x >> 4), n = nt * 16 + ((int)threadIdx.x & 15);
  if (nt >= a.g.nta || n >= a.g.N) return;
  const float* l = lg[threadIdx.x];
  const int M = a.nr_mix;
  float best = -INFINITY, uu = 0.5f;
  int bidx = 0;
  for (int q = 0; q <= M / 4; ++q) {  // draws 0 .. M: M mixture-indicator uniforms, then the logistic one (word m & 3 of call m >> 2)
    uint32_t r[4];
    philox4x32((uint32_t)s, (uint32_t)n, (uint32_t)q, 0x4d4f4c21u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = q * 4 + e;
      const float u = 1e-5f + (1.0f - 2e-5f) * u32_to_unit(r[e]);  // uniform_(1e-5, 1 - 1e-5)
      if (m < M) {
        const float v = l[m] - logf(-logf(u));
        if (v > best) { best = v; bidx = m; }  // first maximum on ties
      } else if (m == M) uu = u;
    }
  }
  const float mean = l[M + bidx];
  const float ls = fmaxf(l[2 * M + bidx], -32.23619130191664f);  // log(1e-14)
  float x = mean + expf(ls) * (logf(uu) - logf(1.f - uu));
  x = fminf(fmaxf(x, -1.f), 1.f);
  a.slot[n] = (0x80000000ull << 32) | (unsigned long long)__float_as_uint(x);
}

}  // namespace mb
