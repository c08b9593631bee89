// WaveRNN (fatchord) generate(): conditioning networks + autoregressive sample loop.
//
// Reference: models/vocoder/wavernn/models/fatchord_version.py
//   MelResNet/UpsampleNetwork :27-85, WaveRNN.generate :153-257 (loop body :190-234),
//   fold_with_overlap :288-338 (done here by index arithmetic, never materialised).
//
// Loop restructuring (all exact algebra, fp32 throughout):
//   * I([x_prev, m_t, a1_t]) = W_I[:,0]*x_prev + (W_I[:,1:].[m_t;a1_t] + b_I): the
//     second term does not depend on the recurrence.  The upsampled mel m_t is a fixed 5-tap
//     combination of neighbouring mel frames and a1_t is constant over a frame, so the term is
//     rebuilt per position from per-FRAME tables (rnn.h WfCond / wf_cond_row4: 8 KB per mel frame
//     instead of a per-position table of 8.2 KB per output sample).
//   * the aux columns of rnn2 / fc1 / fc2 (a2,a3,a4 are constant over the 200
//     samples of a mel frame) become per-FRAME tables G2pre/F1pre/F2pre with
//     the biases folded in.  One extra all-zero conditioning row stands for
//     the zero padding fold_with_overlap appends (:325-327).
//   * what remains per step is 5 dependent skinny GEMMs (rnn.hip) + the
//     sampler; the steps are captured in a hipGraph and replayed, the step
//     index lives in device memory so the graph is parameter free.
//   * production ("split-hidden") chain: a GRU's hidden half W_hh.h + b_hh depends only on the
//     PREVIOUS step's state, so it is computed beside fc1 / fc2 of the previous step in the same
//     launches (rnn_dual_linear_kernel: extra workgroups on CUs the 64-workgroup fc launch leaves
//     idle) and leaves the dependent chain.  rnn1's input half is W_ih1.(Ipre[pos] + x*W_I[:,0]) =
//     T1[pos] + x*(W_ih1.W_I[:,0]) with T1 one more per-position table, so once the sample x is
//     known rnn1 is ELEMENTWISE (wavernn_gru1_finish_kernel) and rnn2 only multiplies its input
//     half (K = 512 instead of 1024).  Still 5 launches per step, about half the bytes on the chain.
#include <atomic>
#include "rnn.h"
#include "wavernn_fast.h"
#include "wavernn_persist.h"
#include "wavernn_pipe16.h"

namespace mb {

// ---- sampler ----------------------------------------------------------------
struct SampK {
  const float* logits;  // [n][C] (lane-local)
  const float* noise;   // [S][N_total][C] Exp(1) draws or null
  unsigned long long seed;
  const float* forced;  // [N_total][S] or null
  float* samples;       // [N_total][S]
  float* logits_out;    // [S][N_total][C] or null
  const int* step_base; // step index of the first launch of the current graph replay (device word)
  int step_off;         // + offset baked into this launch
  int n_off, N_total;   // this lane covers folds [n_off, n_off + gridDim.x)
  int C, S, R;
  int fold_stride, total_len, hop, frames;
  WfCond cond;          // per-frame conditioning tables (rnn.h): Ipre[pos] is rebuilt from them
  const float* wI0;     // [R]
  float* x0;            // [n][R] (lane-local)
  volatile int* progress;
  unsigned long long* trace;
};

// x0 / table row for lane-local fold n at step s1 with fed-back sample xfb (:192-195 + fold indexing :334-336)
__device__ __forceinline__ void prep_step(const SampK& a, int n, int s1, float xfb, int tid, int nthreads) {
  const unsigned pos = (unsigned)(a.n_off + n) * (unsigned)a.fold_stride + (unsigned)s1;
  for (int j = tid; j < a.R; j += nthreads)
    a.x0[(size_t)n * a.R + j] = wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, j, a.R, a.frames) + xfb * a.wI0[j];
}

__global__ __launch_bounds__(128) void wavernn_init_kernel(SampK a) {
  prep_step(a, blockIdx.x, 0, 0.f, threadIdx.x, blockDim.x);
}

// MBHIP_WAVERNN_CHAIN=classic only: its rnn1 launch reads the Ipre rows as a GEMM operand -> materialise them ([total_len + 1][R])
__global__ void wavernn_materialize_ipre_kernel(WfCond c, int total_len, int R, float* out) {
  const size_t total = (size_t)(total_len + 1) * R;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    out[idx] = wf_cond_ipre(c, (unsigned)(idx / R), (unsigned)total_len, (int)(idx % R), R, c.frames);
}

// fused-sampling path: the sample of the LAST step is still only an argmax word; decode it.
__global__ void wavernn_flush_kernel(const unsigned long long* slot, float* samples, int N, int S, int C, int mol) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) samples[(size_t)n * S + (S - 1)] = !slot[n] ? 0.f : mol ? __uint_as_float((unsigned)slot[n]) : 2.f * (float)argmax_class(slot[n]) / ((float)C - 1.f) - 1.f;
}

// step_base += n (last node of every graph replay / after every eager step)
__global__ void wavernn_bump_kernel(int* step_base, int n) { *step_base += n; }

// softmax -> Categorical.sample() -> 2k/(C-1)-1   (:222-228).  torch.multinomial(p, 1) on the
// CPU path is argmax(p / Exp(1) noise) (verified bit-exact), restated here with
// the noise either injected (parity) or drawn from Philox (production; one call per 4 classes).
//
// ONE wavefront per fold: C/64 classes per lane (C = 512 -> two float4), so max / sum / argmax are
// pure wave-shuffle reductions -- no LDS, no barrier.  Latency-first like rnn.hip: the logits, the
// step counter, the next step's Ipre row and W_I[:,0] are all requested before the first wait; the
// fed-back sample only scales W_I[:,0] at the very end.
template <int C4>  // float4 chunks of logits per lane (C = 256 * C4)
__global__ __launch_bounds__(64) void wavernn_sample_kernel(SampK a) {
  trace_begin(a.trace);
  const int n = blockIdx.x, lane = threadIdx.x;
  const int gn = a.n_off + n;
  const float* lg = a.logits + (size_t)n * a.C;
  float4 v[C4];
#pragma unroll
  for (int q = 0; q < C4; ++q) v[q] = *reinterpret_cast<const float4*>(lg + (q * 64 + lane) * 4);
  const int s = *a.step_base + a.step_off;
  // next step's input row: position known from s alone (fold indexing :334-336)
  const unsigned pos = (unsigned)gn * (unsigned)a.fold_stride + (unsigned)(s + 1);
  constexpr int R4MAX = 4;  // R <= 1024
  float4 ipv[R4MAX], w0v[R4MAX];
#pragma unroll
  for (int q = 0; q < R4MAX; ++q) {
    const int j = (q * 64 + lane) * 4;
    const int jc = j < a.R ? j : 0;
    ipv[q] = make_float4(wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc, a.R, a.frames), wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc + 1, a.R, a.frames),
                         wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc + 2, a.R, a.frames), wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc + 3, a.R, a.frames));
    w0v[q] = *reinterpret_cast<const float4*>(a.wI0 + jc);
  }
  const size_t nb = ((size_t)s * a.N_total + gn) * a.C;
  float4 ev[C4];
  if (a.noise) {
#pragma unroll
    for (int q = 0; q < C4; ++q) ev[q] = *reinterpret_cast<const float4*>(a.noise + nb + (q * 64 + lane) * 4);
  }
  const float forced = a.forced ? a.forced[(size_t)gn * a.S + s] : 0.f;

  float m = -INFINITY;
#pragma unroll
  for (int q = 0; q < C4; ++q) m = fmaxf(fmaxf(m, fmaxf(v[q].x, v[q].y)), fmaxf(v[q].z, v[q].w));
  m = wave_max(m);
  float ex[C4][4];
  float sum = 0.f;
#pragma unroll
  for (int q = 0; q < C4; ++q) {
    ex[q][0] = expf(v[q].x - m); ex[q][1] = expf(v[q].y - m); ex[q][2] = expf(v[q].z - m); ex[q][3] = expf(v[q].w - m);
    sum += (ex[q][0] + ex[q][1]) + (ex[q][2] + ex[q][3]);
  }
  sum = wave_sum(sum);
  float best = -1.f;
  int bidx = 0x7fffffff;
#pragma unroll
  for (int q = 0; q < C4; ++q) {
    const int c = (q * 64 + lane) * 4;
    float e4[4];
    if (a.noise) { e4[0] = ev[q].x; e4[1] = ev[q].y; e4[2] = ev[q].z; e4[3] = ev[q].w; }
    else {
      uint32_t r[4];
      philox4x32((uint32_t)s, (uint32_t)gn, (uint32_t)(c >> 2), 0x57415645u, (uint32_t)a.seed,
                 (uint32_t)(a.seed >> 32), r);
#pragma unroll
      for (int i = 0; i < 4; ++i) e4[i] = -logf(u32_to_unit(r[i]));
    }
    if (a.logits_out) *reinterpret_cast<float4*>(a.logits_out + nb + c) = v[q];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float qv = (ex[q][i] / sum) / e4[i];
      if (qv > best) { best = qv; bidx = c + i; }  // ascending c per lane: first max kept
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
  const float x = 2.f * (float)bidx / ((float)a.C - 1.f) - 1.f;
  if (lane == 0) {
    a.samples[(size_t)gn * a.S + s] = x;
    if (a.progress && gn == 0 && (s % 100 == 0 || s == a.S - 1)) *a.progress = s + 1;
  }
  const float xfb = a.forced ? forced : x;
  if (s + 1 < a.S) {
#pragma unroll
    for (int q = 0; q < R4MAX; ++q) {
      const int j = (q * 64 + lane) * 4;
      if (j < a.R) {
        float4 o;
        o.x = ipv[q].x + xfb * w0v[q].x; o.y = ipv[q].y + xfb * w0v[q].y;
        o.z = ipv[q].z + xfb * w0v[q].z; o.w = ipv[q].w + xfb * w0v[q].w;
        *reinterpret_cast<float4*>(a.x0 + (size_t)n * a.R + j) = o;
      }
    }
  }
  trace_end(a.trace);
}

// MOL mode (fatchord_version.py:213-220 -> models/vocoder/distribution.py:87-123, B = 1, T = folds): the 3 * nr_mix fc3
// outputs are (mixture logits | means | log scales); pick the mixture by Gumbel-argmax over logits - log(-log u_m),
// sample x = mean + exp(max(log_scale, log 1e-14)) * (log u - log(1 - u)), clamp to [-1, 1].  One wavefront per fold,
// lane c holds output c.  Noise: injected uniforms [S][N_total][nr_mix + 1] (the reference's two uniform_(1e-5, 1 - 1e-5)
// draws of a step: nr_mix indicator draws, then the logistic draw) or Philox.
__global__ __launch_bounds__(64) void wavernn_sample_mol_kernel(SampK a, int nr_mix) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int gn = a.n_off + n;
  const float lg = lane < a.C ? a.logits[(size_t)n * a.C + lane] : 0.f;
  const int s = *a.step_base + a.step_off;
  const unsigned pos = (unsigned)gn * (unsigned)a.fold_stride + (unsigned)(s + 1);
  constexpr int R4MAX = 4;  // R <= 1024
  float4 ipv[R4MAX], w0v[R4MAX];
#pragma unroll
  for (int q = 0; q < R4MAX; ++q) {
    const int j = (q * 64 + lane) * 4;
    const int jc = j < a.R ? j : 0;
    ipv[q] = make_float4(wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc, a.R, a.frames), wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc + 1, a.R, a.frames),
                         wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc + 2, a.R, a.frames), wf_cond_ipre(a.cond, pos, (unsigned)a.total_len, jc + 3, a.R, a.frames));
    w0v[q] = *reinterpret_cast<const float4*>(a.wI0 + jc);
  }
  float u;
  if (a.noise) u = lane <= nr_mix ? a.noise[((size_t)s * a.N_total + gn) * (nr_mix + 1) + lane] : 0.5f;
  else {
    uint32_t r[4];
    philox4x32((uint32_t)s, (uint32_t)gn, (uint32_t)(lane >> 2), 0x4d4f4c21u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), r);
    u = 1e-5f + (1.0f - 2e-5f) * u32_to_unit(r[lane & 3]);  // uniform_(1e-5, 1 - 1e-5)
  }
  const float forced = a.forced ? a.forced[(size_t)gn * a.S + s] : 0.f;
  if (a.logits_out && lane < a.C) a.logits_out[((size_t)s * a.N_total + gn) * a.C + lane] = lg;
  // mixture indicator: argmax_m logit_m - log(-log u_m), first maximum on ties
  float best = lane < nr_mix ? lg - logf(-logf(u)) : -INFINITY;
  int bidx = lane < nr_mix ? lane : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bidx, o, 64);
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
  }
  const float mean = __shfl(lg, nr_mix + bidx, 64);
  const float ls = fmaxf(__shfl(lg, 2 * nr_mix + bidx, 64), -32.23619130191664f);  // log(1e-14)
  const float uu = __shfl(u, nr_mix, 64);
  float x = mean + expf(ls) * (logf(uu) - logf(1.f - uu));
  x = fminf(fmaxf(x, -1.f), 1.f);
  if (lane == 0) {
    a.samples[(size_t)gn * a.S + s] = x;
    if (a.progress && gn == 0 && (s % 100 == 0 || s == a.S - 1)) *a.progress = s + 1;
  }
  const float xfb = a.forced ? forced : x;
  if (s + 1 < a.S) {
#pragma unroll
    for (int q = 0; q < R4MAX; ++q) {
      const int j = (q * 64 + lane) * 4;
      if (j < a.R) {
        float4 o;
        o.x = ipv[q].x + xfb * w0v[q].x; o.y = ipv[q].y + xfb * w0v[q].y;
        o.z = ipv[q].z + xfb * w0v[q].z; o.w = ipv[q].w + xfb * w0v[q].w;
        *reinterpret_cast<float4*>(a.x0 + (size_t)n * a.R + j) = o;
      }
    }
  }
}

}  // namespace mb

using namespace mb;

namespace {
struct CondConv {  // conv with BN folded, packed for conv1d.hip
  int c_out, c_in, k, pad;
  DevBuf w, b;
};
}  // namespace

struct mb_wavernn {
  mb_wavernn_config cfg;
  int hop, aux_dims, n_classes;
  // conditioning
  CondConv conv_in, conv_out;
  std::vector<CondConv> res1, res2;
  // tables
  CondConv t_g2, t_f1, t_f2;  // 1x1 convs producing the per-frame tables G2pre / F1pre / F2pre
  // per-frame tables of the position-dependent terms (rnn.h WfCond): mel part (no bias) and aux part (+ bias) of Ipre and of T1
  CondConv t_UI, t_AI, t_UT, t_AT;
  DevBuf kw;  // [hop][5] upsampling weights of frame offsets -2 .. +2
  // loop weights
  DevBuf wI0, g1I0, w_rnn1, w_rnn2, w_fc1, w_fc2, w_fc3;
  DevBuf b_ih1, b_hh1, b_hh2, b_fc3;
  // split-hidden chain: T1 table conv, rnn2 input half (GRU tile order, K = R), hidden halves as plain
  // row-tile linears (rows in torch gate-major order)
  DevBuf w_rnn2x, w_hh1, w_hh2;
  // fast chain (wavernn_fast.h): hidden halves in GRU tile order, their biases as (r, z, n, -) per unit
  DevBuf f_hh1t, f_hh2t, f_bhh1q, f_bhh2q;
  // resident kernel on 22-bit operand pairs (wavernn_pipe16.h): fp16 hi / lo A fragments of the six on-chip matrices (uint16 pairs
  // stored in float-typed buffers) and their 2^-s
  DevBuf q_rnn2, q_hh2, q_hh1, q_fc1, q_fc2, q_fc3;
  DevBuf q_hh1l, q_hh2l;  // the hidden halves as plain gate-major linears (the batch loop's hh jobs, rnn_ts3_body.h)
  float q_us[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  // Folds are independent sequences: they are dealt to up to MAX_LANES "lanes", each with its own
  // stream + graph, so the per-kernel dependency latency of one lane overlaps with the others.
  static constexpr int MAX_LANES = 8;
  int resident_cus = -1;                      // compute units a resident launch may count on (-1: not probed yet, 0: none)
  int* d_range = nullptr;                     // device word the wide-batch fp16 GEMMs (rnn_ts3_body.h) raise for an activation beyond their range
  int* h_range = nullptr;                     // its pinned host copy
  int* h_abort = nullptr;                     // pinned host copy of the resident launch's abort word [0] and range word [1]
  int last_path = MB_WRN_PATH_CHAIN;          // what computed the samples of the last generate call (mb_wavernn_last_path)
  int last_fallback = MB_WRN_FALLBACK_NONE;   // and whether a resident launch was discarded on the way
  hipStream_t loop_stream = nullptr;          // lane 0 (also runs the conditioning networks)
  hipStream_t lane_stream[MAX_LANES] = {};
  hipEvent_t lane_ev[MAX_LANES] = {};
  hipEvent_t ev_in = nullptr, ev_out = nullptr, ev_t0 = nullptr, ev_t1 = nullptr, ev_cond = nullptr;
  hipGraph_t graph[MAX_LANES] = {};           // kept until the next call / destroy so generate stays async
  hipGraphExec_t graph_exec[MAX_LANES] = {};
  int last_launches = 0, last_lanes = 1;
  bool timed = false;
  int bench_which = 0, bench_iters = 0;  // set by mb_wavernn_bench_kernel
  bool bench_chain = false;              // mb_wavernn_bench_kernel times the launch chain: no resident launch meanwhile
  void drop_graph() {
    for (int l = 0; l < MAX_LANES; ++l) {
      if (graph_exec[l]) { (void)hipStreamSynchronize(lane_stream[l]); (void)hipGraphExecDestroy(graph_exec[l]); graph_exec[l] = nullptr; }
      if (graph[l]) { (void)hipGraphDestroy(graph[l]); graph[l] = nullptr; }
    }
  }
};

// a resident launch (wavernn_persist.h / wavernn_pipe.h) once lost a hand-off on this device: stop defaulting to them there
static std::atomic<bool> g_resident_failed[64] = {};  // written by whichever host thread sees the abort word: atomic

// ---- which form of the sample loop a call runs: ONE function (the host code below and tests/test_host_logic.py both go through it) ----
//   columns      fold columns of the call (1 = batched=False)
//   mode         0 RAW, 1 MOL
//   production   1: production-dims model (rnn 512 / fc 512, classes <= 512), sampling on the device, no logits dump / teacher forcing /
//                trace / per-kernel bench -- the only calls a resident launch serves
//   have_q16     the fp16 hi / lo weight images of wavernn_pipe16.h exist (K = 512 models)
//   resident_cus co-resident 512-thread workgroups the device offers (its CU count if the kernels fit one per unit, else 0)
//   dev_failed   a resident launch lost a hand-off on this device before
//   resident     MBHIP_WAVERNN_RESIDENT: -1 unset (auto), 0 = no resident launch, 1 = resident launch wherever legal (also on a device
//                that failed before), 2 ("exact") = like 1 with the exact fp32 kernel (wavernn_pipe.h) instead of wavernn_pipe16.h
// | columns | RAW                                   | MOL                                   |
// | 1       | wf_pipe16_kernel; exact: wf_persist1  | wf_pipe16_kernel; exact: chain        |
// | 2..32   | wf_pipe16_kernel; exact: wf_pipe      | wf_pipe16_kernel; exact: wf_pipe      |
// | 33..64  | wf_pipe16_kernel; exact: chain        | wf_pipe16_kernel; exact: chain        |
// | 65..96  | wf_pipe16_kernel; exact: chain        | launch chain                          |
// | > 96    | launch chain (mb_wavernn_generate_batch's wide GEMMs serve several utterances)   |
// and the launch chain whenever production == 0, the device offers fewer units than the kernel has workgroups (224 / 192), resident
// = 0, or the device failed before and the switch does not ask explicitly.
int wavernn_pick_path(int columns, int mode, int production, int have_q16, int resident_cus, int dev_failed, int resident) {
  if (!production || columns < 1 || resident == 0) return MB_WRN_PATH_CHAIN;
  if (dev_failed && resident < 0) return MB_WRN_PATH_CHAIN;
  const bool q16 = have_q16 && resident != 2;  // (both sampler modes: the MOL F3 role exists on both resident kernels)
  if (columns >= 2) {
    if (columns > (q16 ? WQ_GMAX : WQ_G) * WQ_GC) return MB_WRN_PATH_CHAIN;
    if (mode != 0 && columns > 64) return MB_WRN_PATH_CHAIN;  // the MOL launch chain's fused sampler stops at 64 columns, and with it `production`
    if (resident_cus < WQ_WGS) return MB_WRN_PATH_CHAIN;
    return q16 ? MB_WRN_PATH_PIPE16 : MB_WRN_PATH_PIPE;
  }
  // one column (batched=False): the operand-pair kernel serves it as one group of one column (round 5: 8.4 us per step against 9.4 on
  // wf_persist1_kernel, and MOL models leave the launch chain); "exact" keeps the fmaf-chain kernel (RAW only)
  if (q16 && resident_cus >= WQ_WGS) return MB_WRN_PATH_PIPE16;
  if (mode != 0) return MB_WRN_PATH_CHAIN;
  if (resident_cus < WP_ON + WP_OFF) return MB_WRN_PATH_CHAIN;
  return MB_WRN_PATH_PERSIST1;
}
extern "C" int mb_wavernn_loop_path(int columns, int mode, int production, int have_q16, int resident_cus, int dev_failed, int resident) {
  return wavernn_pick_path(columns, mode, production, have_q16, resident_cus, dev_failed, resident);
}
// MBHIP_WAVERNN_RESIDENT as wavernn_pick_path's `resident`
static int wavernn_resident_env() {  // -2 = a value that is none of auto | 0 | 1 | exact
  const char* e = getenv("MBHIP_WAVERNN_RESIDENT");
  if (!e || strcmp(e, "auto") == 0) return -1;
  if (strcmp(e, "exact") == 0) return 2;
  if (strcmp(e, "0") == 0) return 0;
  if (strcmp(e, "1") == 0) return 1;
  return -2;
}

static int wavernn_shapes(const mb_wavernn_config* c, std::vector<size_t>* numel) {
  MB_REQUIRE(c, "wavernn: null config");
  MB_REQUIRE(c->mode == 0 || c->mode == 1, "wavernn: mode must be 0 (RAW) or 1 (MOL)");
  MB_REQUIRE(c->n_upsample >= 1 && c->n_upsample <= 4, "wavernn: n_upsample");
  MB_REQUIRE(c->rnn_dims % 16 == 0 && c->fc_dims % 16 == 0, "wavernn: rnn_dims/fc_dims must be multiples of 16");
  MB_REQUIRE(c->res_out_dims % 4 == 0, "wavernn: res_out_dims %% 4");
  MB_REQUIRE(c->mode == 1 || (c->bits >= 8 && c->bits <= 10), "wavernn: bits=%d unsupported by the one-wave sampler (8..10)", c->bits);
  MB_REQUIRE(c->rnn_dims <= 1024, "wavernn: rnn_dims=%d > 1024", c->rnn_dims);
  const size_t R = c->rnn_dims, FC = c->fc_dims, A = c->res_out_dims / 4, CD = c->compute_dims;
  const size_t C = c->mode == 1 ? 30 : (size_t)1 << c->bits;  // fatchord_version.py:95-98
  numel->clear();
  auto bn = [&](size_t n) { for (int i = 0; i < 4; ++i) numel->push_back(n); };
  numel->push_back(CD * c->feat_dims * (2 * c->pad + 1));  // conv_in
  bn(CD);
  for (int i = 0; i < c->res_blocks; ++i) {
    numel->push_back(CD * CD); bn(CD);  // conv1, batch_norm1
    numel->push_back(CD * CD); bn(CD);  // conv2, batch_norm2
  }
  numel->push_back((size_t)c->res_out_dims * CD);  // conv_out.weight
  numel->push_back(c->res_out_dims);               // conv_out.bias
  for (int i = 0; i < c->n_upsample; ++i) numel->push_back(2 * c->upsample_factors[i] + 1);
  numel->push_back(R * (c->feat_dims + A + 1)); numel->push_back(R);                    // I
  numel->push_back(3 * R * R); numel->push_back(3 * R * R); numel->push_back(3 * R); numel->push_back(3 * R);  // rnn1
  numel->push_back(3 * R * (R + A)); numel->push_back(3 * R * R); numel->push_back(3 * R); numel->push_back(3 * R);  // rnn2
  numel->push_back(FC * (R + A)); numel->push_back(FC);   // fc1
  numel->push_back(FC * (FC + A)); numel->push_back(FC);  // fc2
  numel->push_back(C * FC); numel->push_back(C);          // fc3
  return MB_OK;
}

extern "C" int mb_wavernn_num_weights(const mb_wavernn_config* cfg) {
  std::vector<size_t> v;
  if (wavernn_shapes(cfg, &v)) return MB_EINVAL;
  return (int)v.size();
}
extern "C" size_t mb_wavernn_weight_numel(const mb_wavernn_config* cfg, int index) {
  std::vector<size_t> v;
  if (wavernn_shapes(cfg, &v) || index < 0 || index >= (int)v.size()) return 0;
  return v[index];
}

// split image of a tile-ordered matrix for wavernn_pipe16.h (K = 512 only: the resident kernels are production-dims kernels)
static int upload_q16(const float* rows, int n_live_rows, int K, int RL, DevBuf* dst, float* unscale, int min_tiles = 0) {
  if (K != 512) return MB_OK;
  const int sexp = wq16_scale_exp(rows, (size_t)n_live_rows * K);
  std::vector<unsigned short> img;
  wq16_pack(rows, n_live_rows, K, RL, sexp, &img, min_tiles);
  *unscale = std::ldexp(1.f, -sexp);
  return dst->upload(reinterpret_cast<const float*>(img.data()), img.size() / 2);
}

// conv weight [c_out][c_in][k] (+ optional eval BatchNorm folded in) -> packed + bias
static int make_cond_conv(CondConv* cc, const float* w, int c_out, int c_in, int k, int pad,
                          const float* bias, const float* const* bn /* w,b,mean,var or null */) {
  cc->c_out = c_out; cc->c_in = c_in; cc->k = k; cc->pad = pad;
  std::vector<float> wf((size_t)c_out * c_in * k), bf(c_out, 0.f);
  for (int co = 0; co < c_out; ++co) {
    double scale = 1.0, shift = bias ? bias[co] : 0.0;
    if (bn) {  // y = (conv - mean)/sqrt(var+eps)*w + b ; eps = 1e-5 (nn.BatchNorm1d default)
      scale = (double)bn[0][co] / std::sqrt((double)bn[3][co] + 1e-5);
      shift = (double)bn[1][co] + (shift - (double)bn[2][co]) * scale;
    }
    for (size_t i = 0; i < (size_t)c_in * k; ++i)
      wf[(size_t)co * c_in * k + i] = (float)((double)w[(size_t)co * c_in * k + i] * scale);
    bf[co] = (float)shift;
  }
  std::vector<float> packed(mb_conv1d_packed_floats(c_out, c_in, k, 1));
  int rc = mb_conv1d_pack(wf.data(), c_out, c_in, k, 1, 0, pad, packed.data());
  if (!rc) rc = cc->w.upload(packed.data(), packed.size());
  if (!rc) rc = cc->b.upload(bf.data(), bf.size());
  return rc;
}

// sub-matrix columns [c0, c0+nc) of a row-major [rows][ld] matrix
static std::vector<float> col_slice(const float* w, int rows, int ld, int c0, int nc) {
  std::vector<float> o((size_t)rows * nc);
  for (int r = 0; r < rows; ++r) memcpy(&o[(size_t)r * nc], w + (size_t)r * ld + c0, sizeof(float) * nc);
  return o;
}

extern "C" int mb_wavernn_create(const mb_wavernn_config* cfg, const float* const* hw, int n_weights,
                                 mb_wavernn** out) {
  MB_REQUIRE(out && hw, "wavernn_create: null pointer");
  std::vector<size_t> shapes;
  int rc = wavernn_shapes(cfg, &shapes);
  if (rc) return rc;
  MB_REQUIRE(n_weights == (int)shapes.size(), "wavernn_create: expected %d weight tensors, got %d",
             (int)shapes.size(), n_weights);
  mb_wavernn* w = new mb_wavernn();
  w->cfg = *cfg;
  const int R = cfg->rnn_dims, FC = cfg->fc_dims, A = cfg->res_out_dims / 4, CD = cfg->compute_dims;
  const int C = cfg->mode == 1 ? 30 : 1 << cfg->bits, FEAT = cfg->feat_dims;
  w->aux_dims = A; w->n_classes = C;
  w->hop = 1;
  for (int i = 0; i < cfg->n_upsample; ++i) w->hop *= cfg->upsample_factors[i];
  int ix = 0;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  // MelResNet :27-44
  RC(make_cond_conv(&w->conv_in, hw[ix], CD, FEAT, 2 * cfg->pad + 1, cfg->pad, nullptr, hw + ix + 1));
  ix += 5;
  w->res1.resize(cfg->res_blocks); w->res2.resize(cfg->res_blocks);
  for (int i = 0; i < cfg->res_blocks; ++i) {
    RC(make_cond_conv(&w->res1[i], hw[ix], CD, CD, 1, 0, nullptr, hw + ix + 1)); ix += 5;
    RC(make_cond_conv(&w->res2[i], hw[ix], CD, CD, 1, 0, nullptr, hw + ix + 1)); ix += 5;
  }
  RC(make_cond_conv(&w->conv_out, hw[ix], cfg->res_out_dims, CD, 1, 0, hw[ix + 1], nullptr)); ix += 2;
  ix += cfg->n_upsample;  // the upsampling filter taps: folded into the Kw table below
  // I :108,195
  const float* WI = hw[ix]; const float* bI = hw[ix + 1]; ix += 2;
  const int KI = FEAT + A + 1;
  {
    std::vector<float> c0 = col_slice(WI, R, KI, 0, 1);
    RC(w->wI0.upload(c0.data(), c0.size()));
    std::vector<float> wm = col_slice(WI, R, KI, 1, FEAT), wa = col_slice(WI, R, KI, 1 + FEAT, A);
    RC(make_cond_conv(&w->t_UI, wm.data(), R, FEAT, 1, 0, nullptr, nullptr));   // UI[f] = W_I[:, 1:1+feat] . mel_f
    RC(make_cond_conv(&w->t_AI, wa.data(), R, A, 1, 0, bI, nullptr));           // AI[f] = W_I[:, 1+feat:] . a1_f + b_I
  }
  {  // Kw[p][o]: weight of mel frame f + o - 2 in the upsampled mel at position hop f + p -- an impulse pushed through the
     // stretch + box-filter stages (UpsampleNetwork :69-74 with this checkpoint's filter taps) in double precision
    const int NF = 9, c0 = 4;
    std::vector<double> cur(NF, 0.0), nxt;
    cur[c0] = 1.0;
    for (int i = 0; i < cfg->n_upsample; ++i) {
      const int sc = cfg->upsample_factors[i];
      const int tv = (int)cur.size() * sc;
      nxt.assign(tv, 0.0);
      for (int t = 0; t < tv; ++t) {
        double acc = 0.0;
        for (int j = 0; j <= 2 * sc; ++j) {
          const int uu = t + j - sc;
          if (uu >= 0 && uu < tv) acc += (double)hw[ix - cfg->n_upsample - 2 + i][j] * cur[uu / sc];
        }
        nxt[t] = acc;
      }
      cur.swap(nxt);
    }
    const int hop = w->hop;
    std::vector<float> kw((size_t)hop * 5);
    double outside = 0.0;
    for (int f = 0; f < NF; ++f)
      for (int p = 0; p < hop; ++p) {
        const int o = c0 + 2 - f;  // frame f of the response = offset o of the table
        if (o >= 0 && o < 5) kw[(size_t)p * 5 + o] = (float)cur[(size_t)f * hop + p];
        else outside = std::max(outside, std::fabs(cur[(size_t)f * hop + p]));
      }
    if (outside > 0.0) {  // (upsample factors whose filters reach further than two frames: not a fatchord configuration)
      set_error("wavernn_create: the upsampling filters reach beyond +-2 mel frames (%g outside)", outside);
      rc = MB_EINVAL;
    }
    RC(w->kw.upload(kw.data(), kw.size()));
  }
  std::vector<float> rows, packed;
  // rnn1 :109
  {
    const float *wih = hw[ix], *whh = hw[ix + 1], *bih = hw[ix + 2], *bhh = hw[ix + 3]; ix += 4;
    cell_rows(wih, R, R, whh, R, R, 3, &rows);
    pack_rowtile(rows.data(), 3 * R, 2 * R, 3, &packed);
    RC(w->w_rnn1.upload(packed.data(), packed.size()));
    {  // W_ih1 . W_I[:,0] (fused-sampling path: the fed-back sample enters the gates through this vector)
      std::vector<float> g(3 * R);
      for (int r = 0; r < 3 * R; ++r) {
        double acc = 0.0;
        for (int k2 = 0; k2 < R; ++k2) acc += (double)wih[(size_t)r * R + k2] * (double)WI[(size_t)k2 * KI];
        g[r] = (float)acc;
      }
      RC(w->g1I0.upload(g.data(), g.size()));
    }
    RC(w->b_ih1.upload(bih, 3 * R)); RC(w->b_hh1.upload(bhh, 3 * R));
    {  // T1 = (W_ih1 . W_I[:,1:]) . [m; a1] + (W_ih1 . b_I + b_ih1), as per-frame tables: mel columns (UT) and aux columns + bias (AT)
      const int KC = FEAT + A;
      std::vector<float> m((size_t)3 * R * KC), mb(3 * R);
      std::vector<double> acc(KC);
      for (int r = 0; r < 3 * R; ++r) {
        std::fill(acc.begin(), acc.end(), 0.0);
        double ab = (double)bih[r];
        for (int k2 = 0; k2 < R; ++k2) {
          const double wv = (double)wih[(size_t)r * R + k2];
          const float* wi = WI + (size_t)k2 * KI + 1;
          for (int q = 0; q < KC; ++q) acc[q] += wv * (double)wi[q];
          ab += wv * (double)bI[k2];
        }
        for (int q = 0; q < KC; ++q) m[(size_t)r * KC + q] = (float)acc[q];
        mb[r] = (float)ab;
      }
      std::vector<float> mm = col_slice(m.data(), 3 * R, KC, 0, FEAT), ma = col_slice(m.data(), 3 * R, KC, FEAT, A);
      RC(make_cond_conv(&w->t_UT, mm.data(), 3 * R, FEAT, 1, 0, nullptr, nullptr));
      RC(make_cond_conv(&w->t_AT, ma.data(), 3 * R, A, 1, 0, mb.data(), nullptr));
    }
    pack_rowtile(whh, 3 * R, R, 4, &packed);
    RC(w->w_hh1.upload(packed.data(), packed.size()));
    RC(upload_q16(whh, 3 * R, R, 4, &w->q_hh1l, &w->q_us[6]));
    cell_rows(whh, R, R, whh, 0, R, 3, &rows); pack_rowtile(rows.data(), 3 * R, R, 3, &packed);
    RC(w->f_hh1t.upload(packed.data(), packed.size()));
    RC(upload_q16(rows.data(), 3 * R, R, 3, &w->q_hh1, &w->q_us[2]));
    std::vector<float> bq((size_t)R * 4);
    for (int j = 0; j < R; ++j) for (int g = 0; g < 4; ++g) bq[(size_t)j * 4 + g] = g < 3 ? bhh[g * R + j] : 0.f;
    RC(w->f_bhh1q.upload(bq.data(), bq.size()));
  }
  // rnn2 :110 (input = [x, a2])
  {
    const float *wih = hw[ix], *whh = hw[ix + 1], *bih = hw[ix + 2], *bhh = hw[ix + 3]; ix += 4;
    cell_rows(wih, R, R + A, whh, R, R, 3, &rows);
    pack_rowtile(rows.data(), 3 * R, 2 * R, 3, &packed);
    RC(w->w_rnn2.upload(packed.data(), packed.size()));
    RC(w->b_hh2.upload(bhh, 3 * R));
    cell_rows(wih, R, R + A, whh, 0, R, 3, &rows);
    pack_rowtile(rows.data(), 3 * R, R, 3, &packed);
    RC(w->w_rnn2x.upload(packed.data(), packed.size()));
    RC(upload_q16(rows.data(), 3 * R, R, 3, &w->q_rnn2, &w->q_us[0]));
    pack_rowtile(whh, 3 * R, R, 4, &packed);
    RC(w->w_hh2.upload(packed.data(), packed.size()));
    RC(upload_q16(whh, 3 * R, R, 4, &w->q_hh2l, &w->q_us[7]));
    cell_rows(whh, R, R, whh, 0, R, 3, &rows); pack_rowtile(rows.data(), 3 * R, R, 3, &packed);
    RC(w->f_hh2t.upload(packed.data(), packed.size()));
    RC(upload_q16(rows.data(), 3 * R, R, 3, &w->q_hh2, &w->q_us[1]));
    std::vector<float> bq((size_t)R * 4);
    for (int j = 0; j < R; ++j) for (int g = 0; g < 4; ++g) bq[(size_t)j * 4 + g] = g < 3 ? bhh[g * R + j] : 0.f;
    RC(w->f_bhh2q.upload(bq.data(), bq.size()));
    std::vector<float> wa = col_slice(wih, 3 * R, R + A, R, A);
    RC(make_cond_conv(&w->t_g2, wa.data(), 3 * R, A, 1, 0, bih, nullptr));
  }
  // fc1 / fc2 / fc3 :111-113
  {
    const float *w1 = hw[ix], *b1 = hw[ix + 1], *w2 = hw[ix + 2], *b2 = hw[ix + 3], *w3 = hw[ix + 4], *b3 = hw[ix + 5];
    ix += 6;
    std::vector<float> m = col_slice(w1, FC, R + A, 0, R);
    pack_rowtile(m.data(), FC, R, 4, &packed); RC(w->w_fc1.upload(packed.data(), packed.size()));
    RC(upload_q16(m.data(), FC, R, 4, &w->q_fc1, &w->q_us[3]));
    std::vector<float> a1 = col_slice(w1, FC, R + A, R, A);
    RC(make_cond_conv(&w->t_f1, a1.data(), FC, A, 1, 0, b1, nullptr));
    m = col_slice(w2, FC, FC + A, 0, FC);
    pack_rowtile(m.data(), FC, FC, 4, &packed); RC(w->w_fc2.upload(packed.data(), packed.size()));
    RC(upload_q16(m.data(), FC, FC, 4, &w->q_fc2, &w->q_us[4]));
    std::vector<float> a2 = col_slice(w2, FC, FC + A, FC, A);
    RC(make_cond_conv(&w->t_f2, a2.data(), FC, A, 1, 0, b2, nullptr));
    pack_rowtile(w3, C, FC, 4, &packed); RC(w->w_fc3.upload(packed.data(), packed.size()));
    RC(upload_q16(w3, C, FC, 4, &w->q_fc3, &w->q_us[5], cfg->mode == 1 ? 2 : 0));  // MOL: the F3 role reads two row tiles whatever nr_mix is
    RC(w->b_fc3.upload(b3, C));
  }
#undef RC
  if (!rc) rc = pool_stream(0, &w->lane_stream[0]);  // (one lane: the loop runs on the library's pool stream 0, common.h)
  if (!rc && hipEventCreateWithFlags(&w->lane_ev[0], hipEventDisableTiming) != hipSuccess) rc = MB_EHIP;
  w->loop_stream = w->lane_stream[0];
  if (!rc && hipEventCreateWithFlags(&w->ev_cond, hipEventDisableTiming) != hipSuccess) rc = MB_EHIP;
  if (!rc) {
    if (hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&w->ev_out, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&w->ev_t0) != hipSuccess || hipEventCreate(&w->ev_t1) != hipSuccess)
      rc = MB_EHIP;
  }
  // the range word of the wide-batch operand-pair GEMMs and its pinned copy: both here, once (allocated lazily on a const
  // handle, a failed second allocation left a null host word behind a non-null device word)
  if (!rc && (hipMalloc((void**)&w->d_range, sizeof(int)) != hipSuccess ||
              hipHostMalloc((void**)&w->h_range, sizeof(int), hipHostMallocDefault) != hipSuccess)) rc = MB_EHIP;
  if (rc) { if (rc == MB_EHIP) set_error("wavernn_create: stream / event / range-word creation failed"); mb_wavernn_destroy(w); return rc; }
  *out = w;
  return MB_OK;
}

extern "C" void mb_wavernn_destroy(mb_wavernn* w) {
  if (!w) return;
  auto rel = [](CondConv& c) { c.w.release(); c.b.release(); };
  rel(w->conv_in); rel(w->conv_out); rel(w->t_UI); rel(w->t_AI); rel(w->t_UT); rel(w->t_AT); rel(w->t_g2); rel(w->t_f1); rel(w->t_f2);
  for (auto& c : w->res1) rel(c);
  for (auto& c : w->res2) rel(c);
  DevBuf* bs[] = {&w->wI0, &w->g1I0, &w->w_rnn1, &w->w_rnn2, &w->w_fc1, &w->w_fc2, &w->w_fc3,
                  &w->b_ih1, &w->b_hh1, &w->b_hh2, &w->b_fc3, &w->w_rnn2x, &w->w_hh1, &w->w_hh2,
                  &w->f_hh1t, &w->f_hh2t, &w->f_bhh1q, &w->f_bhh2q, &w->kw,
                  &w->q_rnn2, &w->q_hh2, &w->q_hh1, &w->q_fc1, &w->q_fc2, &w->q_fc3, &w->q_hh1l, &w->q_hh2l};
  for (DevBuf* b : bs) b->release();
  w->drop_graph();
  if (w->h_abort) (void)hipHostFree(w->h_abort);
  if (w->h_range) (void)hipHostFree(w->h_range);
  if (w->d_range) (void)hipFree(w->d_range);
  if (w->ev_in) (void)hipEventDestroy(w->ev_in);
  if (w->ev_out) (void)hipEventDestroy(w->ev_out);
  if (w->ev_t0) (void)hipEventDestroy(w->ev_t0);
  if (w->ev_t1) (void)hipEventDestroy(w->ev_t1);
  if (w->ev_cond) (void)hipEventDestroy(w->ev_cond);
  for (int l = 0; l < mb_wavernn::MAX_LANES; ++l) {
    if (w->lane_ev[l]) (void)hipEventDestroy(w->lane_ev[l]);
    w->lane_stream[l] = nullptr;  // (borrowed from the pool)
  }
  delete w;
}

namespace {
struct WrnLayout {
  float *r0, *r1, *r2, *aux, *UT, *AT, *UI, *AI, *Ipre, *G2, *F1, *F2;  // Ipre: MBHIP_WAVERNN_CHAIN=classic only
  float *x0, *x1, *x2, *y1, *y2, *logits, *h1, *h2;
  float *P1, *P2;  // split-hidden chain: hidden-half pre-activations
  float *f_x1, *f_x2, *f_y1, *f_y2, *f_h1, *f_h2, *f_P1, *f_P2, *f_Tq;  // fast chain: FM activations / state, CM4 hidden halves, staged table rows
  size_t f_bytes;
  int* step; unsigned long long* slots;
  unsigned long long* px;  // persistent kernel (wavernn_persist.h): granule exchange area + abort word
  size_t bytes;
};
// python-style floor division
inline long long floordiv(long long a, long long b) { long long q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --q; return q; }
}  // namespace

// MBHIP_WAVERNN_CHAIN selects the launch chain behind (or instead of) the resident kernels -- a comma list of
//   fast     (default) the split-hidden chain on fragment-major activations (wavernn_fast.h)
//   split    the split-hidden chain on the rnn_rowtile_body instances (round 1)
//   classic  the 5-launch chain with both GRU halves on the dependent path (no T1 table: 6 KB less workspace per conditioning position)
//   nofuse   stand-alone sampler launch (6 launches per step) under any of them
// The exact / teacher-forced / logits-dump test paths build on split / classic + nofuse whatever the variable says.
static bool wavernn_chain_has(const char* word) {
  const char* e = getenv("MBHIP_WAVERNN_CHAIN");
  if (!e) return false;
  const size_t wl = strlen(word);
  for (const char* p = e; *p;) {
    const char* end = strchr(p, ',');
    const size_t len = end ? (size_t)(end - p) : strlen(p);
    if (len == wl && strncmp(p, word, wl) == 0) return true;
    p += len + (end ? 1 : 0);
  }
  return false;
}
static bool wavernn_split_chain() { return !wavernn_chain_has("classic"); }
// every word of MBHIP_WAVERNN_CHAIN is one of the four (a misspelt switch must not silently run the default)
static bool wavernn_chain_valid() {
  const char* e = getenv("MBHIP_WAVERNN_CHAIN");
  if (!e) return true;
  for (const char* p = e; *p;) {
    const char* end = strchr(p, ',');
    const size_t len = end ? (size_t)(end - p) : strlen(p);
    const std::string word(p, len);
    if (word != "fast" && word != "split" && word != "classic" && word != "nofuse") return false;
    p += len + (end ? 1 : 0);
  }
  return true;
}

static void wavernn_layout(const mb_wavernn* w, const mb_wavernn_plan* p, void* base, WrnLayout* L) {
  const mb_wavernn_config& c = w->cfg;
  const size_t F = p->frames, T = p->total_len, N = p->n_folds, R = c.rnn_dims, FC = c.fc_dims;
  const size_t CD = c.compute_dims;
  Arena ar(base, (size_t)-1);
  L->r0 = ar.take<float>(CD * F); L->r1 = ar.take<float>(CD * F); L->r2 = ar.take<float>(CD * F);
  L->aux = ar.take<float>((size_t)c.res_out_dims * F);
  // per-frame tables of the position-dependent conditioning terms (rnn.h WfCond): 2 zero rows, F frames, 7 zero rows / F + 1 rows
  L->UT = ar.take<float>((F + 9) * 3 * R); L->UI = ar.take<float>((F + 9) * R);
  L->AT = ar.take<float>((F + 1) * 3 * R); L->AI = ar.take<float>((F + 1) * R);
  L->Ipre = ar.take<float>(wavernn_split_chain() ? 1 : (T + 1) * R);
  L->G2 = ar.take<float>((F + 1) * 3 * R);
  L->F1 = ar.take<float>((F + 1) * FC);
  L->F2 = ar.take<float>((F + 1) * FC);
  L->x0 = ar.take<float>(N * R); L->x1 = ar.take<float>(N * R); L->x2 = ar.take<float>(N * R);
  L->y1 = ar.take<float>(N * FC); L->y2 = ar.take<float>(N * FC);
  L->logits = ar.take<float>(N * w->n_classes);
  L->h1 = ar.take<float>(2 * N * R); L->h2 = ar.take<float>(2 * N * R);
  L->P1 = ar.take<float>(N * 3 * R); L->P2 = ar.take<float>(N * 3 * R);
  {
    const int nta = (int)((N + 15) / 16);
    L->f_x1 = ar.take<float>(fm_floats((int)R, nta));
    const size_t start = ar.off - fm_floats((int)R, nta) * sizeof(float);
    L->f_x2 = ar.take<float>(fm_floats((int)R, nta)); L->f_y1 = ar.take<float>(fm_floats((int)FC, nta)); L->f_y2 = ar.take<float>(fm_floats((int)FC, nta));
    L->f_h1 = ar.take<float>(fm_floats((int)R, nta)); L->f_h2 = ar.take<float>(fm_floats((int)R, nta));
    L->f_P1 = ar.take<float>(4 * cm_items((int)R, nta)); L->f_P2 = ar.take<float>(4 * cm_items((int)R, nta));
    L->f_Tq = ar.take<float>(2 * 4 * cm_items((int)R, nta));  // ping-pong: the finish launch reads one, stages the next step's into the other
    L->f_bytes = ar.off - start;
  }
  L->step = ar.take<int>(16);
  L->slots = ar.take<unsigned long long>(2 * N);
  L->px = ar.take<unsigned long long>(std::max(wp_exchange_bytes(), wq_exchange_bytes()) / 8);
  L->bytes = ar.off + 256;
}

extern "C" int mb_wavernn_plan_generate(const mb_wavernn* w, int frames, int batched, int target,
                                        int overlap, mb_wavernn_plan* plan) {
  MB_REQUIRE(w && plan, "wavernn_plan: null pointer");
  MB_REQUIRE(frames >= 1, "wavernn_plan: empty mel (frames=%d)", frames);
  MB_REQUIRE(w->cfg.n_upsample <= 3, "wavernn_plan: >3 upsample stages not supported");
  plan->frames = frames;
  const long long total = (long long)frames * w->hop;
  MB_REQUIRE(total < (1ll << 31) - 1, "wavernn_plan: mel too long");
  plan->total_len = (int)total;
  if (batched) {
    MB_REQUIRE(target > 0 && overlap >= 0, "wavernn_plan: target/overlap");
    // fold_with_overlap :313-322
    long long nf = floordiv(total - overlap, target + overlap);
    const long long ext = nf * (overlap + target) + overlap;
    if (total - ext != 0) nf += 1;
    MB_REQUIRE(nf >= 1, "wavernn_plan: mel too short for batched generation (target=%d overlap=%d)", target, overlap);
    plan->n_folds = (int)nf;
    plan->seq_len = target + 2 * overlap;
    plan->fold_stride = target + overlap;
  } else {
    plan->n_folds = 1; plan->seq_len = plan->total_len; plan->fold_stride = 0;
  }
  WrnLayout L;
  wavernn_layout(w, plan, nullptr, &L);
  plan->workspace_bytes = L.bytes;
  return MB_OK;
}

static int run_cond_conv(const CondConv& cc, const float* x, int t, float* y, const float* res,
                         int out_act, int transpose_out, hipStream_t s) {
  mb_conv1d_args a;
  memset(&a, 0, sizeof(a));
  a.d_x = x; a.d_wpacked = cc.w.p; a.d_bias = cc.b.p; a.d_res = res; a.d_y = y;
  a.batch = 1; a.c_in = cc.c_in; a.c_out = cc.c_out; a.t_in = t; a.t_out = t;
  a.ksize = cc.k; a.dilation = 1; a.pad = cc.pad; a.up = 1;
  a.out_act = out_act; a.transpose_out = transpose_out;
  return mb_conv1d(&a, (mb_stream_t)s);
}

// Error exits of the two generate entry points: work may already be queued on the handle's own lane streams, which
// the caller's stream is not ordered after until the final join.  Drain them so that the caller may free or reuse
// the workspace / mel / sample buffers once the error is returned.  (A handle is single-threaded: it owns mutable
// graph, event and stream state; concurrent calls need one handle each.)
static int wavernn_join_on_error(mb_wavernn* w, int rc) {
  if (rc && w)
    for (int l = 0; l < mb_wavernn::MAX_LANES; ++l)
      if (w->lane_stream[l]) (void)hipStreamSynchronize(w->lane_stream[l]);
  return rc;
}

static int wavernn_generate_impl(const mb_wavernn* wc, const mb_wavernn_plan* plan, const float* d_mel,
                                 const float* d_noise, uint64_t seed, float* d_samples,
                                 float* d_logits_out, const float* d_forced, int* h_progress,
                                 void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  mb_wavernn* w = const_cast<mb_wavernn*>(wc);
  MB_REQUIRE(w && plan && d_mel && d_samples, "wavernn_generate: null pointer");
  WrnLayout L;
  wavernn_layout(w, plan, d_workspace, &L);
  if (!d_workspace || workspace_bytes < L.bytes) {
    set_error("wavernn_generate: workspace %zu B < required %zu B", workspace_bytes, L.bytes);
    return MB_ENOMEM;
  }
  const mb_wavernn_config& c = w->cfg;
  const int F = plan->frames, T = plan->total_len, N = plan->n_folds, S = plan->seq_len;
  const int R = c.rnn_dims, FC = c.fc_dims, A = w->aux_dims, C = w->n_classes;
  hipStream_t cs = (hipStream_t)stream, s = w->loop_stream;
  MB_HIP(hipEventRecord(w->ev_in, cs));
  MB_HIP(hipStreamWaitEvent(s, w->ev_in, 0));
  int rc = MB_OK;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  // ---- aux = MelResNet(pad(mel)) :37-44 (zero padding == conv padding) ----
  RC(run_cond_conv(w->conv_in, d_mel, F, L.r0, nullptr, 1, 0, s));
  float* cur = L.r0; float* oth = L.r1;
  for (int i = 0; i < c.res_blocks; ++i) {  // ResBlock :17-24
    RC(run_cond_conv(w->res1[i], cur, F, L.r2, nullptr, 1, 0, s));
    RC(run_cond_conv(w->res2[i], L.r2, F, oth, cur, 0, 0, s));
    std::swap(cur, oth);
  }
  RC(run_cond_conv(w->conv_out, cur, F, L.aux, nullptr, 0, 0, s));
  // ---- the position-dependent conditioning terms as per-FRAME tables (rnn.h WfCond; the upsampled mel :78-85 is never built) ----
  WfCond cond;
  cond.UT = L.UT; cond.AT = L.AT; cond.UI = L.UI; cond.AI = L.AI; cond.Kw = w->kw.p; cond.hop = w->hop; cond.frames = F;
  if (!rc) {
    MB_HIP(hipMemsetAsync(L.UT, 0, sizeof(float) * (size_t)(F + 9) * 3 * R, s));
    MB_HIP(hipMemsetAsync(L.UI, 0, sizeof(float) * (size_t)(F + 9) * R, s));
  }
  RC(run_cond_conv(w->t_UT, d_mel, F, L.UT + (size_t)2 * 3 * R, nullptr, 0, 1, s));
  RC(run_cond_conv(w->t_UI, d_mel, F, L.UI + (size_t)2 * R, nullptr, 0, 1, s));
  RC(run_cond_conv(w->t_AT, L.aux, F, L.AT, nullptr, 0, 1, s));   // a1 = aux rows [0, A)
  RC(run_cond_conv(w->t_AI, L.aux, F, L.AI, nullptr, 0, 1, s));
  if (!rc) {  // zero-conditioning rows = the biases
    MB_HIP(hipMemcpyAsync(L.AT + (size_t)F * 3 * R, w->t_AT.b.p, sizeof(float) * 3 * R, hipMemcpyDeviceToDevice, s));
    MB_HIP(hipMemcpyAsync(L.AI + (size_t)F * R, w->t_AI.b.p, sizeof(float) * R, hipMemcpyDeviceToDevice, s));
  }
  // Production path (no injected noise / teacher forcing / logits dump): the sampler is fused into the
  // fc3 launch and the next step's input is rebuilt from the argmax word -> 5 launches per step.
  // MOL mode: the fused form exists on the fragment-major chain only (wf_fc3_mol_kernel: fc3 + the mixture sampler in one launch,
  // the sample itself in the slot word); every other configuration of a MOL model keeps the exact 6-launch chain.
  MB_REQUIRE(wavernn_chain_valid(), "MBHIP_WAVERNN_CHAIN: unknown word in '%s' (fast | split | classic, optionally ,nofuse)", getenv("MBHIP_WAVERNN_CHAIN"));
  const bool fast_ok = wavernn_split_chain() && !wavernn_chain_has("split");
  const bool mol_fast = c.mode == 1 && R == 512 && FC == 512 && C <= 32 && N <= 64 && fast_ok;
  const bool fused = !d_noise && !d_forced && !d_logits_out && !wavernn_chain_has("nofuse") && (c.mode == 0 || mol_fast);
  const bool split = fused && wavernn_split_chain();
  // Production shape (rnn 512 / fc 512, <= 64 fold columns): the same chain on fragment-major activations
  // (wavernn_fast.h), bit-identical sample stream.  MBHIP_WAVERNN_CHAIN=split keeps the rnn_rowtile_body instances.
  // (fc3 + the next step's rnn1 in one launch, two column tiles per row-tile workgroup and several stream "lanes" of columns were
  //  measured in round 1 -- 26.4 / 29.1 against 25.7 / 26.0 us per step, lanes do not overlap -- and left the library in round 4.)
  const bool fastk = split && R == 512 && FC == 512 && (C % 16 == 0 || c.mode == 1) && N <= 64 && fast_ok;
  const int nta = cdiv(N, 16);
  const int fnt = nta < 2 ? 1 : 2;  // fold-column tiles per workgroup of the fast chain
  // ---- per-frame tables of the aux columns (time-major) + the zero-conditioning row = bias ----
  RC(run_cond_conv(w->t_g2, L.aux + (size_t)1 * A * F, F, L.G2, nullptr, 0, 1, s));
  RC(run_cond_conv(w->t_f1, L.aux + (size_t)2 * A * F, F, L.F1, nullptr, 0, 1, s));
  RC(run_cond_conv(w->t_f2, L.aux + (size_t)3 * A * F, F, L.F2, nullptr, 0, 1, s));
  if (!rc) {
    if (fused && !split) {  // classic chain: its rnn1 launch reads the I(..) rows as a GEMM operand
      hipLaunchKernelGGL(wavernn_materialize_ipre_kernel, dim3(2048), dim3(256), 0, s, cond, T, R, L.Ipre);
      MB_HIP(hipGetLastError());
    }
    MB_HIP(hipMemcpyAsync(L.G2 + (size_t)F * 3 * R, w->t_g2.b.p, sizeof(float) * 3 * R, hipMemcpyDeviceToDevice, s));
    MB_HIP(hipMemcpyAsync(L.F1 + (size_t)F * FC, w->t_f1.b.p, sizeof(float) * FC, hipMemcpyDeviceToDevice, s));
    MB_HIP(hipMemcpyAsync(L.F2 + (size_t)F * FC, w->t_f2.b.p, sizeof(float) * FC, hipMemcpyDeviceToDevice, s));
    MB_HIP(hipMemsetAsync(L.h1, 0, sizeof(float) * 2 * N * R, s));
    MB_HIP(hipMemsetAsync(L.h2, 0, sizeof(float) * 2 * N * R, s));
    MB_HIP(hipMemsetAsync(L.step, 0, sizeof(int) * 16, s));
    MB_HIP(hipMemsetAsync(L.slots, 0, sizeof(unsigned long long) * 2 * N, s));
  }
  WfGeom wg;
  wg.step_base = L.step; wg.step_off = 0; wg.fold_stride = plan->fold_stride; wg.total_len = T; wg.hop = w->hop; wg.frames = F;
  wg.nta = nta; wg.N = N;
  auto fc_hh = [&](int g, int n_fc, int soff, hipStream_t st) {  // launch C (g = 0) / D (g = 1); n_fc = 0: hidden halves only
    WfFcHhK k;
    k.g = wg; k.g.step_off = soff;
    k.w_fc = g ? w->w_fc2.p : w->w_fc1.p; k.xin = g ? L.f_y1 : L.f_x2; k.F = g ? L.F2 : L.F1; k.y = g ? L.f_y2 : L.f_y1;
    k.n_fc = n_fc; k.FC = FC;
    k.w_hh = g ? w->f_hh2t.p : w->f_hh1t.p; k.h = g ? L.f_h2 : L.f_h1;
    k.bhh4 = reinterpret_cast<const float4*>(g ? w->f_bhh2q.p : w->f_bhh1q.p); k.P = reinterpret_cast<float4*>(g ? L.f_P2 : L.f_P1);
    const dim3 grid(n_fc + R / 4, cdiv(nta, fnt));
    if (fnt == 2) hipLaunchKernelGGL(wf_fc_hh_kernel<2>, grid, dim3(512), 0, st, k);
    else hipLaunchKernelGGL(wf_fc_hh_kernel<1>, grid, dim3(512), 0, st, k);
  };
  // Resident forms of the loop: ONE launch for the whole utterance, every weight tile in LDS, granule hand-offs between
  // the layers, same sample stream as the chain.
  //   * wf_pipe16_kernel (wavernn_pipe16.h, RAW and MOL models) / wf_pipe_kernel (wavernn_pipe.h, MBHIP_WAVERNN_RESIDENT=exact or no fp16 images), 2..96 / 2..32 fold
  //     columns: role-specialised workgroups, column groups in flight.
  //   * wf_persist1_kernel (wavernn_persist.h): one column (batched=False).  MBHIP_WAVERNN_RESIDENT=0 keeps the chain for both.
  //   The choice is wavernn_pick_path's table (above).
  // Both need their workgroups co-resident, one per compute unit: checked here against the device (CU count, occupancy
  // of the kernel, a per-device "it failed before" flag); a launch that still loses a hand-off (another process holds
  // compute units) times out after 0.2 s, raises its abort word and the chain below computes the utterance instead; so does a
  // wf_pipe16_kernel launch that met a value beyond its operand range (range word).  The chain's stream is wf_pipe_kernel's and
  // wf_persist1_kernel's bit for bit, but NOT wf_pipe16_kernel's (same noise, fp32-grade instead of fp32 sums: the two differ at
  // near-ties) -- mb_wavernn_last_path says which form produced the samples of a call.
  // NOTE: a resident launch makes this call host-blocking (the abort word has to be looked at before returning).
  std::string trace_file;  // diagnostics (MBHIP_DIAG=trace_file=<path>, wf_which=<bits>): the chain, launch by launch
  const char* trace_path = diag_str("trace_file", &trace_file) ? trace_file.c_str() : nullptr;
  const int dbg_which = diag_int("wf_which", -1);
  // (the resident kernels go beyond the fast chain's 64 columns: up to WQ_GMAX groups of 16; a MOL model's F3 role reads two fc3 row tiles)
  const bool resident_dims = split && R == 512 && FC == 512 && (C % 16 == 0 || c.mode == 1) && fast_ok && (c.mode == 0 || C > 16);
  const bool resident_ok = resident_dims && C <= 512 && !w->bench_which && !w->bench_chain && !trace_path && dbg_which < 0;
  int path = MB_WRN_PATH_CHAIN;
  w->last_path = MB_WRN_PATH_CHAIN; w->last_fallback = MB_WRN_FALLBACK_NONE;
  if (resident_ok && !rc) {
    int dev = 0;
    MB_HIP(hipGetDevice(&dev));
    if (w->resident_cus < 0) {  // once per handle = per device
      hipDeviceProp_t prop;
      MB_HIP(hipGetDeviceProperties(&prop, dev));
      MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wf_persist1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WP1_LDS_BYTES));
      MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wf_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WQ_LDS_BYTES));
      int nb1 = 0, nb2 = 0, nb3 = 1 << 30;
      MB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, reinterpret_cast<const void*>(wf_persist1_kernel), 512, WP1_LDS_BYTES));
      MB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, reinterpret_cast<const void*>(wf_pipe_kernel), 512, WQ_LDS_BYTES));
      const void* p16[7] = {reinterpret_cast<const void*>(wf_pipe16_kernel<false, 2, false>), reinterpret_cast<const void*>(wf_pipe16_kernel<false, 4, false>),
                            reinterpret_cast<const void*>(wf_pipe16_kernel<false, 6, false>), reinterpret_cast<const void*>(wf_pipe16_kernel<true, 2, false>),
                            reinterpret_cast<const void*>(wf_pipe16_kernel<true, 4, false>), reinterpret_cast<const void*>(wf_pipe16_kernel<true, 6, false>),
                            reinterpret_cast<const void*>(wf_pipe16_kernel<false, 2, true>)};
      for (const void* f : p16) {
        int nb = 0;
        MB_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WQ16_LDS_BYTES));
        MB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, 512, WQ16_LDS_BYTES));
        nb3 = std::min(nb3, nb);
      }
      w->resident_cus = (nb1 >= 1 && nb2 >= 1 && nb3 >= 1) ? prop.multiProcessorCount : 0;
      if (!w->h_abort) MB_HIP(hipHostMalloc((void**)&w->h_abort, 2 * sizeof(int), hipHostMallocDefault));
    }
    const bool dev_failed = dev >= 0 && dev < 64 && g_resident_failed[dev];
    const int resident_sw = wavernn_resident_env();
    MB_REQUIRE(resident_sw != -2, "MBHIP_WAVERNN_RESIDENT: unknown value '%s' (auto | 0 | 1 | exact)", getenv("MBHIP_WAVERNN_RESIDENT"));
    path = wavernn_pick_path(N, c.mode, 1, w->q_fc3.p && w->q_hh1.p ? 1 : 0, w->resident_cus, dev_failed ? 1 : 0, resident_sw);
  }
  const bool pipe = path == MB_WRN_PATH_PIPE || path == MB_WRN_PATH_PIPE16, persist = path == MB_WRN_PATH_PERSIST1;
  const bool q16 = path == MB_WRN_PATH_PIPE16;
  if ((pipe || persist) && !rc) {
    const size_t ex_bytes = pipe ? wq_exchange_bytes() : wp_exchange_bytes();
    int* abort_word = reinterpret_cast<int*>(L.px + (pipe ? (size_t)WQ_GMAX * 2 * WQX_PER : (size_t)2 * WPX_PER_PARITY));
    int* range_word = abort_word + 1;
    MB_HIP(hipMemsetAsync(L.px, 0, ex_bytes, s));
    const bool test_abort = diag_int("abort_wp") != 0;  // tests only (MBHIP_DIAG=abort_wp): the launch finds its abort word raised, the chain takes over
    if (test_abort) MB_HIP(hipMemsetAsync(abort_word, 1, 1, s));
    std::string wtrace_file;  // diagnostics (MBHIP_DIAG=wp_trace=<file>): dump the wall-clock marks of the kernel to this file
    const char* wtrace = diag_str("wp_trace", &wtrace_file) ? wtrace_file.c_str() : nullptr;
    unsigned long long* trace = wtrace ? reinterpret_cast<unsigned long long*>(abort_word) + 32 : nullptr;
    MB_HIP(hipEventRecord(w->ev_t0, s));
    int n_groups = 2;
    if (pipe) {
      WqK qk;
      qk.w_rnn2 = w->w_rnn2x.p; qk.w_hh2 = w->f_hh2t.p; qk.w_hh1 = w->f_hh1t.p; qk.w_fc1 = w->w_fc1.p; qk.w_fc2 = w->w_fc2.p; qk.w_fc3 = w->w_fc3.p;
      qk.bhh1q = reinterpret_cast<const float4*>(w->f_bhh1q.p); qk.bhh2q = reinterpret_cast<const float4*>(w->f_bhh2q.p);
      qk.b_fc3 = w->b_fc3.p; qk.g1 = w->g1I0.p; qk.wI0 = w->wI0.p;
      qk.cond = cond; qk.G2 = L.G2; qk.F1 = L.F1; qk.F2 = L.F2;
      qk.g = wg; qk.ex = L.px; qk.abort_word = abort_word;
      qk.samples = d_samples; qk.progress = h_progress; qk.seed = seed; qk.R = R; qk.FC = FC; qk.C = C; qk.S = S; qk.N = N;
      qk.mol = c.mode == 1 ? 1 : 0; qk.nr_mix = C / 3;
      {  // column groups: two up to 32 columns, ceil(N / 16) beyond (pipe16 only), sizes as even as possible
        int ng = N <= 2 * WQ_GC ? 2 : (N + WQ_GC - 1) / WQ_GC;
        if (const int want = diag_int("wq_groups", 0)) {  // A/B (MBHIP_DIAG=wq_groups=<n>): one group (no pipelining) / more groups than needed
          if (want == 1 && N <= WQ_GC) ng = 1;
          else if (q16 && want >= ng && want <= WQ_GMAX) ng = want;
        }
        n_groups = ng;
        for (int g = 0; g <= WQ_GMAX; ++g) qk.gn0[g] = g >= ng ? N : (int)((long long)N * g / ng + ((long long)N * g % ng ? 1 : 0));  // ceil(N g / ng)
        if (ng == 2) qk.gn0[1] = (N + 1) / 2;
      }
      qk.flags = diag_int("wq_flags", 17);  // A/B switches of wavernn_pipe.h (1 = padded rows, 16 = weights in registers)
      qk.trace = trace;
      if (q16) {
        Wq16K k16;
        k16.q = qk;
        k16.h_rnn2 = reinterpret_cast<const uint4*>(w->q_rnn2.p); k16.h_hh2 = reinterpret_cast<const uint4*>(w->q_hh2.p);
        k16.h_hh1 = reinterpret_cast<const uint4*>(w->q_hh1.p); k16.h_fc1 = reinterpret_cast<const uint4*>(w->q_fc1.p);
        k16.h_fc2 = reinterpret_cast<const uint4*>(w->q_fc2.p); k16.h_fc3 = reinterpret_cast<const uint4*>(w->q_fc3.p);
        k16.us_rnn2 = w->q_us[0]; k16.us_hh2 = w->q_us[1]; k16.us_hh1 = w->q_us[2]; k16.us_fc1 = w->q_us[3]; k16.us_fc2 = w->q_us[4]; k16.us_fc3 = w->q_us[5];
        k16.range_word = range_word;
        // the instance with the fewest group slots that holds the launch's groups (per-group state is register arrays of that size)
#define MB_WQ16(MOL_, NG_, TR_) hipLaunchKernelGGL((wf_pipe16_kernel<MOL_, NG_, TR_>), dim3(WQ_WGS), dim3(512), WQ16_LDS_BYTES, s, k16)
        if (c.mode == 1) { if (n_groups <= 2) MB_WQ16(true, 2, false); else if (n_groups <= 4) MB_WQ16(true, 4, false); else MB_WQ16(true, 6, false); }
        else if (n_groups <= 2) { if (trace) MB_WQ16(false, 2, true); else MB_WQ16(false, 2, false); }  // (marks: the RAW two-group instance only)
        else if (n_groups <= 4) MB_WQ16(false, 4, false);
        else MB_WQ16(false, 6, false);
#undef MB_WQ16
      } else
      hipLaunchKernelGGL(wf_pipe_kernel, dim3(WQ_WGS), dim3(512), WQ_LDS_BYTES, s, qk);
    } else {
      WpK pk;
      pk.w_rnn2 = w->w_rnn2x.p; pk.w_fc1 = w->w_fc1.p; pk.w_fc2 = w->w_fc2.p; pk.w_fc3 = w->w_fc3.p; pk.w_hh1 = w->f_hh1t.p; pk.w_hh2 = w->f_hh2t.p;
      pk.bhh1q = reinterpret_cast<const float4*>(w->f_bhh1q.p); pk.bhh2q = reinterpret_cast<const float4*>(w->f_bhh2q.p);
      pk.b_fc3 = w->b_fc3.p; pk.g1 = w->g1I0.p; pk.wI0 = w->wI0.p;
      pk.cond = cond; pk.G2 = L.G2; pk.F1 = L.F1; pk.F2 = L.F2;
      pk.g = wg; pk.ex = L.px; pk.abort_word = abort_word;
      pk.samples = d_samples; pk.progress = h_progress; pk.seed = seed; pk.R = R; pk.FC = FC; pk.C = C; pk.S = S; pk.N = N;
      pk.trace = trace;
      hipLaunchKernelGGL(wf_persist1_kernel, dim3(WP_ON + WP_OFF), dim3(512), WP1_LDS_BYTES, s, pk);  // (one column)
    }
    MB_HIP(hipGetLastError());
    MB_HIP(hipEventRecord(w->ev_t1, s));
    w->last_launches = 1; w->last_lanes = 1; w->timed = true;
    // the abort word is the only way a broken hand-off shows: one D2H copy into pinned memory behind the launch, one wait
    MB_HIP(hipMemcpyAsync(w->h_abort, abort_word, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    MB_HIP(hipEventRecord(w->ev_out, s));
    MB_HIP(hipStreamWaitEvent(cs, w->ev_out, 0));
    MB_HIP(hipEventSynchronize(w->ev_out));
    const int aborted = w->h_abort[0], out_of_range = q16 ? w->h_abort[1] : 0;
    if (wtrace && !aborted) {
      unsigned long long marks[1024];  // [role 5][step 4][mark 16], then from word 512 every workgroup's publish time at step 1001
      MB_HIP(hipMemcpy(marks, trace, sizeof(marks), hipMemcpyDeviceToHost));
      if (FILE* f = fopen(wtrace, "wb")) { fwrite(marks, sizeof(marks), 1, f); fclose(f); }
    }
    if (!aborted && !out_of_range) { w->last_path = path; return MB_OK; }
    if (aborted) {
      // A hand-off never arrived within the time limit: the workgroups were not all resident (something else holds
      // compute units -- another stream or process) and the launch drained itself.  The chain below computes the utterance
      // without needing co-residency (the exact kernels' stream bit for bit; wf_pipe16_kernel's up to near-ties: same noise,
      // fp32 sums); say so once and stop defaulting to resident launches here.
      static bool warned = false;
      if (!warned) {
        fprintf(stderr, "[mbhip] wavernn: resident kernel could not keep its workgroups co-resident; using the launch chain\n");
        warned = true;
      }
      int dev = 0;
      if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !diag_int("abort_wp")) g_resident_failed[dev] = true;
      w->last_fallback = MB_WRN_FALLBACK_ABORT;
    } else {
      // wf_pipe16_kernel published a value beyond fp16's range (or a NaN): its operand pairs cannot carry this utterance.  The
      // launch's samples are discarded and the fp32 chain computes it; nothing is remembered (the next utterance tries again).
      static bool warned_range = false;
      if (!warned_range) {
        fprintf(stderr, "[mbhip] wavernn: an activation left the operand-pair kernel's range (|x| > 65504 or NaN); using the fp32 launch chain\n");
        warned_range = true;
      }
      w->last_fallback = MB_WRN_FALLBACK_RANGE;
    }
  }
  if (fastk && !rc) {  // zero state; P = W_hh.0 + b_hh for the first step by the loop's own hidden-half jobs
    MB_HIP(hipMemsetAsync(L.f_x1, 0, L.f_bytes, s));
    fc_hh(0, 0, 0, s);
    fc_hh(1, 0, 0, s);
    {  // table rows of step 0
      WfStageK sk;
      sk.cond = cond; sk.Tq = reinterpret_cast<float4*>(L.f_Tq); sk.R = R; sk.step_add = 0;
      hipLaunchKernelGGL(wf_stage_kernel, dim3(cdiv((R / 4) * nta, 8)), dim3(512), 0, s, sk, wg);  // -> buffer 0 (step parity 0)
    }
    MB_HIP(hipGetLastError());
  } else
  if (split && !rc) {  // P = W_hh.0 + b_hh for the first step, by the same launch the loop uses
    for (int g = 0; g < 2 && !rc; ++g) {
      RnnK k;
      memset(&k, 0, sizeof(k));
      k.w = g ? w->w_hh2.p : w->w_hh1.p; k.nseg = 1; k.nkb_total = R / 16;
      k.seg[0] = {g ? L.h2 : L.h1, R, R / 16, 0};
      k.N = N; k.units = 3 * R; k.biasX = g ? w->b_hh2.p : w->b_hh1.p; k.y = g ? L.P2 : L.P1; k.ldy = 3 * R;
      rc = rnn_launch(EPI_LINEAR, k, s);
    }
  }
  if (rc) return rc;

  // ---- one lane = one stream for all fold columns.  (Several lanes of contiguous fold ranges were measured in round 1,
  // profiles/r01_wavernn_lane_sweep.json: they do NOT overlap -- kernel submission serialises on the host / CP at ~3-4 us per
  // launch; the switch is gone, the loops below keep the lane index.) ----
  const int lanes = 1;
  int lane_n0[mb_wavernn::MAX_LANES + 1];
  for (int l = 0; l <= lanes; ++l) lane_n0[l] = (int)((long long)N * l / lanes);
  MB_HIP(hipEventRecord(w->ev_cond, s));  // conditioning tables + zeroed state are ready

  auto make_sk = [&](int l) {
    const int n0 = lane_n0[l];
    SampK sk;
    sk.logits = L.logits + (size_t)n0 * C; sk.noise = d_noise; sk.seed = seed; sk.forced = d_forced;
    sk.samples = d_samples; sk.logits_out = d_logits_out; sk.step_base = L.step + l; sk.step_off = 0; sk.n_off = n0; sk.N_total = N;
    sk.C = C; sk.S = S; sk.R = R; sk.fold_stride = plan->fold_stride; sk.total_len = T; sk.hop = w->hop; sk.frames = F;
    sk.cond = cond; sk.wI0 = w->wI0.p; sk.x0 = L.x0 + (size_t)n0 * R;
    sk.progress = h_progress; sk.trace = nullptr;
    return sk;
  };

  // one time step of lane l = 5 GEMM launches + sampler; pp = parity of the step (state ping-pong)
  auto step = [&](int l, int pp, int soff, int which = 0x3f, unsigned long long* tr = nullptr) -> int {
    const int n0 = lane_n0[l], nl = lane_n0[l + 1] - n0;
    hipStream_t ls = w->lane_stream[l];
    float* x0 = L.x0 + (size_t)n0 * R; float* x1 = L.x1 + (size_t)n0 * R; float* x2 = L.x2 + (size_t)n0 * R;
    float* y1 = L.y1 + (size_t)n0 * FC; float* y2 = L.y2 + (size_t)n0 * FC; float* lgt = L.logits + (size_t)n0 * C;
    float* h1p = L.h1 + ((size_t)pp * N + n0) * R; float* h1n = L.h1 + ((size_t)(pp ^ 1) * N + n0) * R;
    float* h2p = L.h2 + ((size_t)pp * N + n0) * R; float* h2n = L.h2 + ((size_t)(pp ^ 1) * N + n0) * R;
    auto frame_rows = [&](RnnK& k) {  // per-frame table row of fold n at this step, computed in-kernel
      k.fr_base = L.step + l; k.fr_off = soff; k.fr_n_off = n0; k.fr_fold_stride = plan->fold_stride;
      k.fr_total_len = T; k.fr_hop = w->hop; k.fr_frames = F;
    };
    RnnK k;
    int r = MB_OK;
    // h1 = rnn1(x, h1); x = x + h1   :196-198
    unsigned long long* slot_prev = L.slots + (size_t)(pp ^ 1) * N + n0;  // written by fc3 of step s-1
    unsigned long long* slot_cur = L.slots + (size_t)pp * N + n0;         // written by fc3 of this step
    if (fastk) {  // wavernn_fast.h: A | B | C | D | E on FM activations (one lane: n0 = 0, nl = N)
      // diagnostics only (tools/pmc_wavernn_r02.sh): run just the launches whose bit is set -- rocprofv3 counter mode
      // cannot survive the interleaved chain, one launch type at a time it can.  Results are garbage.
      if (dbg_which >= 0) which &= dbg_which;
      if (which & 1) {
        WfFinK f;
        f.g = wg; f.g.step_off = soff; f.slot = slot_prev; f.P1 = reinterpret_cast<const float4*>(L.f_P1);
        const size_t tqn = cm_items(R, nta);  // float4 items per buffer
        f.Tq = reinterpret_cast<const float4*>(L.f_Tq) + (size_t)pp * tqn;
        f.stage.cond = cond; f.stage.Tq = reinterpret_cast<float4*>(L.f_Tq) + (size_t)(pp ^ 1) * tqn; f.stage.R = R; f.stage.step_add = 1;
        f.n_fin = R / 16;
        f.g1 = w->g1I0.p; f.wI0 = w->wI0.p; f.h1 = L.f_h1; f.x1 = L.f_x1; f.samples = d_samples; f.progress = h_progress;
        f.R = R; f.C = C; f.S = S; f.mol = c.mode == 1 ? 1 : 0;
        hipLaunchKernelGGL(wf_finish_kernel, dim3(R / 16 + cdiv((R / 4) * nta, 4), nta), dim3(256), 0, ls, f);
      }
      if (which & 2) {
        WfRnn2K k2;
        k2.g = wg; k2.g.step_off = soff; k2.w = w->w_rnn2x.p; k2.x1 = L.f_x1; k2.P2 = reinterpret_cast<const float4*>(L.f_P2); k2.G2 = L.G2;
        k2.h2 = L.f_h2; k2.x2 = L.f_x2; k2.zero_slot = slot_prev; k2.R = R;
        if (fnt == 2) hipLaunchKernelGGL(wf_rnn2_kernel<2>, dim3(R / 4, cdiv(nta, 2)), dim3(512), 0, ls, k2);
        else hipLaunchKernelGGL(wf_rnn2_kernel<1>, dim3(R / 4, nta), dim3(512), 0, ls, k2);
      }
      if (which & 4) fc_hh(0, FC / 16, soff, ls);
      if (which & 8) fc_hh(1, FC / 16, soff, ls);
      if ((which & 16) && c.mode == 1) {
        WfFc3MolK km;
        km.g = wg; km.g.step_off = soff; km.w = w->w_fc3.p; km.bias = w->b_fc3.p; km.xin = L.f_y2; km.slot = slot_cur; km.seed = seed; km.C = C; km.nr_mix = C / 3;
        if (fnt == 2) hipLaunchKernelGGL(wf_fc3_mol_kernel<2>, dim3(1, cdiv(nta, 2)), dim3(512), 0, ls, km);
        else hipLaunchKernelGGL(wf_fc3_mol_kernel<1>, dim3(1, nta), dim3(512), 0, ls, km);
      } else if (which & 16) {
        WfFc3K k3;
        k3.g = wg; k3.g.step_off = soff; k3.w = w->w_fc3.p; k3.bias = w->b_fc3.p; k3.xin = L.f_y2; k3.slot = slot_cur; k3.seed = seed; k3.C = C;
        if (fnt == 2) hipLaunchKernelGGL(wf_fc3_kernel<2>, dim3(C / 16, cdiv(nta, 2)), dim3(512), 0, ls, k3);
        else hipLaunchKernelGGL(wf_fc3_kernel<1>, dim3(C / 16, nta), dim3(512), 0, ls, k3);
      }
      MB_HIP(hipGetLastError());
      return MB_OK;
    }
    if (split) {
      float* P1 = L.P1 + (size_t)n0 * 3 * R; float* P2 = L.P2 + (size_t)n0 * 3 * R;
      // (A) rnn1, elementwise: every product is precomputed (T1 table, P1 from the previous step).
      auto fin = [&](int parity, int off) {
        Fin1K f;
        memset(&f, 0, sizeof(f));
        f.slot = L.slots + (size_t)(parity ^ 1) * N + n0;
        f.cond = cond; f.P1 = P1; f.g1 = w->g1I0.p; f.wI0 = w->wI0.p;
        f.h_prev = L.h1 + ((size_t)parity * N + n0) * R; f.h_out = L.h1 + ((size_t)(parity ^ 1) * N + n0) * R;
        f.x_out = x1; f.samples = d_samples; f.progress = h_progress;
        f.step_base = L.step + l; f.step_off = off; f.n_off = n0; f.nl = nl; f.R = R; f.C = C; f.S = S;
        f.fold_stride = plan->fold_stride; f.total_len = T;
        return f;
      };
      if (which & 1) {
        Fin1K f = fin(pp, soff);
        f.trace = tr ? tr + 0 : nullptr;
        if ((r = rnn_launch_finish(f, ls))) return r;
      }
      // (B) rnn2 on its input half; hidden half = P2 (left by the previous step's fc2 launch)
      memset(&k, 0, sizeof(k));
      k.w = w->w_rnn2x.p; k.nseg = 1; k.nkb_total = R / 16; k.seg[0] = {x1, R, R / 16, 0};
      k.N = nl; k.units = R; k.h_pre = P2;
      k.pre_table = L.G2; frame_rows(k); k.pre_stride = 3 * R;
      k.h_prev = h2p; k.x_res = x1; k.h_out = h2n; k.x_out = x2;
      k.zero_slot = slot_prev;  // free for fc3 of step s+1: its only reader (A) has run
      k.trace = tr ? tr + 2 * TRACE_SLOTS : nullptr;
      if ((which & 2) && (r = rnn_launch(EPI_GRU, k, ls))) return r;
      // (C) fc1 || P1 = W_hh1.h1 + b_hh1 for the next step; (D) fc2 || P2 = W_hh2.h2 + b_hh2
      for (int g = 0; g < 2; ++g) {
        RnnK k1;
        memset(&k, 0, sizeof(k)); memset(&k1, 0, sizeof(k1));
        k.w = g ? w->w_fc2.p : w->w_fc1.p; k.nseg = 1; k.nkb_total = R / 16;
        k.seg[0] = {g ? y1 : x2, g ? FC : R, (g ? FC : R) / 16, 0};
        k.nkb_total = k.seg[0].nkb;
        k.N = nl; k.units = FC; k.pre_table = g ? L.F2 : L.F1; frame_rows(k); k.pre_stride = FC;
        k.y = g ? y2 : y1; k.ldy = FC; k.act = 1;
        k.trace = tr ? tr + (4 + 2 * g) * TRACE_SLOTS : nullptr;
        k1.w = g ? w->w_hh2.p : w->w_hh1.p; k1.nseg = 1; k1.nkb_total = R / 16;
        k1.seg[0] = {g ? h2n : h1n, R, R / 16, 0};
        k1.N = nl; k1.units = 3 * R; k1.biasX = g ? w->b_hh2.p : w->b_hh1.p; k1.y = g ? P2 : P1; k1.ldy = 3 * R;
        if ((which & (4 << g)) && (r = rnn_launch_dual_linear(k, k1, ls))) return r;
      }
      // (E) logits = fc3(x) with the Gumbel-argmax sampler in the epilogue   :209,222-228
      memset(&k, 0, sizeof(k));
      k.w = w->w_fc3.p; k.nseg = 1; k.nkb_total = FC / 16; k.seg[0] = {y2, FC, FC / 16, 0};
      k.N = nl; k.units = C; k.biasX = w->b_fc3.p; k.ldy = C;
      frame_rows(k);
      k.gum_slot = slot_cur; k.gum_seed = seed;
      k.trace = tr ? tr + 8 * TRACE_SLOTS : nullptr;
      if ((which & 16) && (r = rnn_launch(EPI_LINEAR, k, ls))) return r;
      return MB_OK;
    }
    memset(&k, 0, sizeof(k));
    k.w = w->w_rnn1.p; k.nseg = 2; k.nkb_total = 2 * R / 16;
    k.seg[0] = {x0, R, R / 16, 0}; k.seg[1] = {h1p, R, R / 16, 1};
    k.N = nl; k.units = R; k.biasX = w->b_ih1.p; k.biasH = w->b_hh1.p;
    k.h_prev = h1p; k.x_res = x0; k.h_out = h1n; k.x_out = x1;
    if (fused) {  // x0 = Ipre[pos] + x_{s-1} * W_I[:,0] rebuilt inside the launch from the argmax word
      frame_rows(k);
      k.x_res = nullptr;
      k.aff_slot = slot_prev; k.aff_table = L.Ipre; k.aff_vec = w->wI0.p; k.aff_gate = w->g1I0.p; k.aff_ld = R; k.aff_C = C; k.aff_S = S;
      k.aff_samples = d_samples; k.aff_progress = h_progress;
    }
    k.trace = tr ? tr + 0 : nullptr;
    if ((which & 1) && (r = rnn_launch(EPI_GRU, k, ls))) return r;
    // h2 = rnn2([x, a2], h2); x = x + h2   :199-202
    memset(&k, 0, sizeof(k));
    k.w = w->w_rnn2.p; k.nseg = 2; k.nkb_total = 2 * R / 16;
    k.seg[0] = {x1, R, R / 16, 0}; k.seg[1] = {h2p, R, R / 16, 1};
    k.N = nl; k.units = R; k.biasH = w->b_hh2.p;
    k.pre_table = L.G2; frame_rows(k); k.pre_stride = 3 * R;
    k.h_prev = h2p; k.x_res = x1; k.h_out = h2n; k.x_out = x2;
    if (fused) k.zero_slot = slot_prev;  // free for fc3 of step s+1 once rnn1 of this step has read it
    k.trace = tr ? tr + 2 * TRACE_SLOTS : nullptr;
    if ((which & 2) && (r = rnn_launch(EPI_GRU, k, ls))) return r;
    // x = relu(fc1([x, a3]))   :203-204
    memset(&k, 0, sizeof(k));
    k.w = w->w_fc1.p; k.nseg = 1; k.nkb_total = R / 16; k.seg[0] = {x2, R, R / 16, 0};
    k.N = nl; k.units = FC; k.pre_table = L.F1; frame_rows(k); k.pre_stride = FC;
    k.y = y1; k.ldy = FC; k.act = 1;
    k.trace = tr ? tr + 4 * TRACE_SLOTS : nullptr;
    if ((which & 4) && (r = rnn_launch(EPI_LINEAR, k, ls))) return r;
    // x = relu(fc2([x, a4]))   :206-207
    memset(&k, 0, sizeof(k));
    k.w = w->w_fc2.p; k.nseg = 1; k.nkb_total = FC / 16; k.seg[0] = {y1, FC, FC / 16, 0};
    k.N = nl; k.units = FC; k.pre_table = L.F2; frame_rows(k); k.pre_stride = FC;
    k.y = y2; k.ldy = FC; k.act = 1;
    k.trace = tr ? tr + 6 * TRACE_SLOTS : nullptr;
    if ((which & 8) && (r = rnn_launch(EPI_LINEAR, k, ls))) return r;
    // logits = fc3(x)   :209
    memset(&k, 0, sizeof(k));
    k.w = w->w_fc3.p; k.nseg = 1; k.nkb_total = FC / 16; k.seg[0] = {y2, FC, FC / 16, 0};
    k.N = nl; k.units = C; k.biasX = w->b_fc3.p; k.y = lgt; k.ldy = C;
    if (fused) {  // Gumbel-argmax sampler in the epilogue: logits never leave the launch
      frame_rows(k);
      k.y = nullptr; k.gum_slot = slot_cur; k.gum_seed = seed;
    }
    k.trace = tr ? tr + 8 * TRACE_SLOTS : nullptr;
    if ((which & 16) && (r = rnn_launch(EPI_LINEAR, k, ls))) return r;
    if ((which & 32) && !fused) {
      SampK sk = make_sk(l);
      sk.step_off = soff;
      sk.trace = tr ? tr + 10 * TRACE_SLOTS : nullptr;
      if (c.mode == 1) hipLaunchKernelGGL(wavernn_sample_mol_kernel, dim3(nl), dim3(64), 0, ls, sk, C / 3);
      else if (C == 512) hipLaunchKernelGGL(wavernn_sample_kernel<2>, dim3(nl), dim3(64), 0, ls, sk);
      else if (C == 256) hipLaunchKernelGGL(wavernn_sample_kernel<1>, dim3(nl), dim3(64), 0, ls, sk);
      else hipLaunchKernelGGL(wavernn_sample_kernel<4>, dim3(nl), dim3(64), 0, ls, sk);
      MB_HIP(hipGetLastError());
    }
    return MB_OK;
  };

  for (int l = 0; l < lanes; ++l) {
    if (l > 0) MB_HIP(hipStreamWaitEvent(w->lane_stream[l], w->ev_cond, 0));
    if (!fused) hipLaunchKernelGGL(wavernn_init_kernel, dim3(lane_n0[l + 1] - lane_n0[l]), dim3(128), 0, w->lane_stream[l], make_sk(l));
    MB_HIP(hipGetLastError());
  }

  // diagnostics: MBHIP_DIAG=trace_file=<path> records per-kernel first-wave-start / last-store-end device
  // timestamps (wall_clock64, 100 MHz) of the LAST graph replay and dumps them after a sync.
  unsigned long long* d_trace = nullptr;
  int trace_steps = 0;
  MB_HIP(hipEventRecord(w->ev_t0, s));
  const bool use_graph = getenv("MBHIP_NO_GRAPH") == nullptr && S >= 64;
  int done = 0;
  if (use_graph) {
    int G = 128;  // steps per graph (even: state parity returns to 0)
    const char* ge = getenv("MBHIP_GRAPH_STEPS");
    if (ge && atoi(ge) >= 2) G = atoi(ge) & ~1;
    while (G > S) G >>= 1;
    G &= ~1;
    if (G >= 2) {
      w->drop_graph();
      for (int l = 0; l < lanes; ++l) {
        hipStream_t ls = w->lane_stream[l];
        if (trace_path && l == 0 && lanes == 1) {
          trace_steps = G;
          if (hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 12 * TRACE_SLOTS * G) != hipSuccess) { d_trace = nullptr; trace_steps = 0; }
        }
        MB_HIP(hipStreamBeginCapture(ls, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < G && !rc; ++i)
          rc = step(l, i & 1, i, 0x3f & ~w->bench_which, d_trace ? d_trace + (size_t)12 * TRACE_SLOTS * i : nullptr);
        hipLaunchKernelGGL(wavernn_bump_kernel, dim3(1), dim3(1), 0, ls, L.step + l, G);  // step_base += G per replay
        hipError_t e = hipStreamEndCapture(ls, &w->graph[l]);
        if (rc) { w->drop_graph(); return rc; }
        if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture", __FILE__, __LINE__);
        e = hipGraphInstantiate(&w->graph_exec[l], w->graph[l], nullptr, nullptr, 0);
        if (e != hipSuccess) { w->drop_graph(); return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__); }
      }
      const int reps = S / G;
      for (int r = 0; r < reps; ++r) {
        if (d_trace && r == reps - 1) {  // (start, end) pairs: start = ~0 for atomicMin, end = 0 for atomicMax
          std::vector<unsigned long long> init((size_t)12 * TRACE_SLOTS * trace_steps);
          for (size_t i = 0; i < init.size(); ++i) init[i] = (i & 1) ? 0ull : ~0ull;
          MB_HIP(hipStreamSynchronize(w->lane_stream[0]));
          MB_HIP(hipMemcpy(d_trace, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        }
        for (int l = 0; l < lanes; ++l) MB_HIP(hipGraphLaunch(w->graph_exec[l], w->lane_stream[l]));
      }
      done = reps * G;
    }
  }
  for (int i = done; i < S && !rc; ++i)  // eager tail: step_base stays at `done`, the offset carries the step
    for (int l = 0; l < lanes && !rc; ++l) rc = step(l, i & 1, i - done, 0x3f & ~w->bench_which);
  if (rc) return rc;
  if (fused) {
    for (int l = 0; l < lanes; ++l) {
      const int n0 = lane_n0[l], nl = lane_n0[l + 1] - n0;
      hipLaunchKernelGGL(wavernn_flush_kernel, dim3(cdiv(nl, 64)), dim3(64), 0, w->lane_stream[l],
                         L.slots + (size_t)((S - 1) & 1) * N + n0, d_samples + (size_t)n0 * S, nl, S, C, c.mode == 1 ? 1 : 0);
    }
    MB_HIP(hipGetLastError());
  }
  for (int l = 1; l < lanes; ++l) {  // join the lanes on lane 0
    MB_HIP(hipEventRecord(w->lane_ev[l], w->lane_stream[l]));
    MB_HIP(hipStreamWaitEvent(s, w->lane_ev[l], 0));
  }
  MB_HIP(hipEventRecord(w->ev_t1, s));
  if (d_trace) {
    std::vector<unsigned long long> host((size_t)12 * TRACE_SLOTS * trace_steps);
    MB_HIP(hipStreamSynchronize(s));
    MB_HIP(hipMemcpy(host.data(), d_trace, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(d_trace);
    if (FILE* f = fopen(trace_path, "wb")) { fwrite(host.data(), sizeof(unsigned long long), host.size(), f); fclose(f); }
  }
  w->last_launches = (fused ? 5 : 6) * S * lanes;
  w->last_lanes = lanes;
  w->timed = true;
  MB_HIP(hipEventRecord(w->ev_out, s));
  MB_HIP(hipStreamWaitEvent(cs, w->ev_out, 0));
#undef RC
  return MB_OK;
}

// ---------------------------------------------------------------------------------------------
// Several utterances in ONE sample loop (additive API; the reference vocodes one utterance per call).
// The folds of all utterances become the columns of the same 5 launches per step, so the launch
// latency that bounds a single utterance (23 columns) is amortised over n_utt x folds columns.
// Production chain only (split-hidden, fused Philox sampler).  Fold n carries a descriptor
// (RnnK::fr_desc) with its utterance's table offsets, so utterances may have different lengths;
// utterance u draws exactly the noise it would draw alone with seed h_seeds[u].
// ---------------------------------------------------------------------------------------------
namespace {
struct WrnBatchLayout {
  float *r0, *r1, *r2, *aux;                             // per-utterance conditioning scratch (max frames)
  float *UT, *UI, *AT, *AI, *G2, *F1, *F2;               // concatenated per-frame tables: U* frames + 9 rows per utterance, the others frames + 1
  float *x1, *x2, *y1, *y2, *h1, *h2, *P1, *P2;
  int* step; unsigned long long* slots; int* desc;
  size_t bytes;
};
void wavernn_batch_layout(const mb_wavernn* w, int n_utt, const int* frames, int n_folds, void* base, WrnBatchLayout* L) {
  const mb_wavernn_config& c = w->cfg;
  const size_t R = c.rnn_dims, FC = c.fc_dims, CD = c.compute_dims, N = n_folds;
  size_t fmax = 0, u_rows = 0, frame_rows = 0;
  for (int u = 0; u < n_utt; ++u) {
    fmax = std::max(fmax, (size_t)frames[u]);
    u_rows += (size_t)frames[u] + 9;
    frame_rows += (size_t)frames[u] + 1;
  }
  Arena ar(base, (size_t)-1);
  L->r0 = ar.take<float>(CD * fmax); L->r1 = ar.take<float>(CD * fmax); L->r2 = ar.take<float>(CD * fmax);
  L->aux = ar.take<float>((size_t)c.res_out_dims * fmax);
  L->UT = ar.take<float>(u_rows * 3 * R); L->UI = ar.take<float>(u_rows * R);
  L->AT = ar.take<float>(frame_rows * 3 * R); L->AI = ar.take<float>(frame_rows * R);
  L->G2 = ar.take<float>(frame_rows * 3 * R); L->F1 = ar.take<float>(frame_rows * FC); L->F2 = ar.take<float>(frame_rows * FC);
  L->x1 = ar.take<float>(N * R); L->x2 = ar.take<float>(N * R); L->y1 = ar.take<float>(N * FC); L->y2 = ar.take<float>(N * FC);
  L->h1 = ar.take<float>(2 * N * R); L->h2 = ar.take<float>(2 * N * R);
  L->P1 = ar.take<float>(N * 3 * R); L->P2 = ar.take<float>(N * 3 * R);
  L->step = ar.take<int>(16);
  L->slots = ar.take<unsigned long long>(2 * N);
  L->desc = ar.take<int>(N * 8);
  L->bytes = ar.off + 256;
}
int batch_folds(const mb_wavernn* w, int frames, int target, int overlap, int* nf) {
  const long long total = (long long)frames * w->hop;
  long long n = floordiv(total - overlap, target + overlap);  // fold_with_overlap :313-322
  if (total - (n * (overlap + target) + overlap) != 0) n += 1;
  MB_REQUIRE(n >= 1, "wavernn_batch: mel of %d frames too short for batched generation", frames);
  *nf = (int)n;
  return MB_OK;
}
}  // namespace

extern "C" int mb_wavernn_plan_generate_batch(const mb_wavernn* w, int n_utt, const int* h_frames, int target, int overlap,
                                              mb_wavernn_batch_plan* plan, int* h_fold_offsets) {
  MB_REQUIRE(w && h_frames && plan && h_fold_offsets && n_utt >= 1, "wavernn_plan_batch: bad arguments");
  MB_REQUIRE(target > 0 && overlap >= 0 && w->cfg.n_upsample <= 3, "wavernn_plan_batch: target/overlap");
  int n = 0;
  for (int u = 0; u < n_utt; ++u) {
    MB_REQUIRE(h_frames[u] >= 1, "wavernn_plan_batch: utterance %d is empty", u);
    int nf = 0;
    int rc = batch_folds(w, h_frames[u], target, overlap, &nf);
    if (rc) return rc;
    h_fold_offsets[u] = n;
    n += nf;
  }
  h_fold_offsets[n_utt] = n;
  plan->n_utt = n_utt; plan->n_folds = n; plan->seq_len = target + 2 * overlap; plan->fold_stride = target + overlap;
  WrnBatchLayout L;
  wavernn_batch_layout(w, n_utt, h_frames, n, nullptr, &L);
  plan->workspace_bytes = L.bytes;
  return MB_OK;
}

extern "C" int mb_wavernn_generate(const mb_wavernn* wc, const mb_wavernn_plan* plan, const float* d_mel,
                                   const float* d_noise, uint64_t seed, float* d_samples,
                                   float* d_logits_out, const float* d_forced, int* h_progress,
                                   void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  return wavernn_join_on_error(const_cast<mb_wavernn*>(wc),
                               wavernn_generate_impl(wc, plan, d_mel, d_noise, seed, d_samples, d_logits_out, d_forced,
                                                     h_progress, d_workspace, workspace_bytes, stream));
}

static int wavernn_generate_batch_impl(const mb_wavernn* wc, const mb_wavernn_batch_plan* plan, const int* h_frames,
                                       const float* const* h_d_mels, const uint64_t* h_seeds, float* d_samples,
                                       void* d_workspace, size_t workspace_bytes, mb_stream_t stream, const bool allow_w16) {
  mb_wavernn* w = const_cast<mb_wavernn*>(wc);
  MB_REQUIRE(w && plan && h_frames && h_d_mels && h_seeds && d_samples, "wavernn_generate_batch: null pointer");
  MB_REQUIRE(rnn_wide_switch_valid(), "MBHIP_RNN_WIDE: unknown value '%s' (ts3 | ts2 | ts, optionally :1..3)", getenv("MBHIP_RNN_WIDE"));
  MB_REQUIRE(w->cfg.mode == 0, "wavernn_generate_batch: RAW mode only (the shared loop is the fused-sampler chain); run MOL utterances one by one");
  const int n_utt = plan->n_utt, N = plan->n_folds, S = plan->seq_len;
  WrnBatchLayout L;
  wavernn_batch_layout(w, n_utt, h_frames, N, d_workspace, &L);
  if (!d_workspace || workspace_bytes < L.bytes) {
    set_error("wavernn_generate_batch: workspace %zu B < required %zu B", workspace_bytes, L.bytes);
    return MB_ENOMEM;
  }
  const mb_wavernn_config& c = w->cfg;
  const int R = c.rnn_dims, FC = c.fc_dims, A = w->aux_dims, C = w->n_classes;
  hipStream_t cs = (hipStream_t)stream, s = w->loop_stream;
  MB_HIP(hipEventRecord(w->ev_in, cs));
  MB_HIP(hipStreamWaitEvent(s, w->ev_in, 0));
  int rc = MB_OK;
#define RC(x) do { if (!rc) rc = (x); } while (0)
  // ---- conditioning networks + tables, one utterance after the other into the concatenated tables ----
  {
    size_t ur = 0;
    for (int u = 0; u < n_utt; ++u) ur += (size_t)h_frames[u] + 9;
    MB_HIP(hipMemsetAsync(L.UT, 0, sizeof(float) * ur * 3 * R, s));
    MB_HIP(hipMemsetAsync(L.UI, 0, sizeof(float) * ur * R, s));
  }
  std::vector<int> desc((size_t)N * 8);
  size_t u_row = 0, frame_row = 0;
  int fold = 0;
  for (int u = 0; u < n_utt && !rc; ++u) {
    const int F = h_frames[u], T = F * w->hop;
    const float* d_mel = h_d_mels[u];
    MB_REQUIRE(d_mel, "wavernn_generate_batch: mel %d is null", u);
    MB_REQUIRE(u_row + F + 9 < ((size_t)1 << 30), "wavernn_generate_batch: batch too long for 32-bit table rows");
    RC(run_cond_conv(w->conv_in, d_mel, F, L.r0, nullptr, 1, 0, s));  // MelResNet :37-44
    float* cur = L.r0; float* oth = L.r1;
    for (int i = 0; i < c.res_blocks; ++i) {
      RC(run_cond_conv(w->res1[i], cur, F, L.r2, nullptr, 1, 0, s));
      RC(run_cond_conv(w->res2[i], L.r2, F, oth, cur, 0, 0, s));
      std::swap(cur, oth);
    }
    RC(run_cond_conv(w->conv_out, cur, F, L.aux, nullptr, 0, 0, s));
    float* UT = L.UT + u_row * 3 * R; float* UI = L.UI + u_row * R;
    float* AT = L.AT + frame_row * 3 * R; float* AI = L.AI + frame_row * R;
    float* G2 = L.G2 + frame_row * 3 * R; float* F1 = L.F1 + frame_row * FC; float* F2 = L.F2 + frame_row * FC;
    RC(run_cond_conv(w->t_UT, d_mel, F, UT + (size_t)2 * 3 * R, nullptr, 0, 1, s));   // rows 0..1 and F+2..F+8 stay zero
    RC(run_cond_conv(w->t_UI, d_mel, F, UI + (size_t)2 * R, nullptr, 0, 1, s));
    RC(run_cond_conv(w->t_AT, L.aux, F, AT, nullptr, 0, 1, s));
    RC(run_cond_conv(w->t_AI, L.aux, F, AI, nullptr, 0, 1, s));
    RC(run_cond_conv(w->t_g2, L.aux + (size_t)1 * A * F, F, G2, nullptr, 0, 1, s));
    RC(run_cond_conv(w->t_f1, L.aux + (size_t)2 * A * F, F, F1, nullptr, 0, 1, s));
    RC(run_cond_conv(w->t_f2, L.aux + (size_t)3 * A * F, F, F2, nullptr, 0, 1, s));
    if (!rc) {  // zero-conditioning rows = the biases
      MB_HIP(hipMemcpyAsync(AT + (size_t)F * 3 * R, w->t_AT.b.p, sizeof(float) * 3 * R, hipMemcpyDeviceToDevice, s));
      MB_HIP(hipMemcpyAsync(AI + (size_t)F * R, w->t_AI.b.p, sizeof(float) * R, hipMemcpyDeviceToDevice, s));
      MB_HIP(hipMemcpyAsync(G2 + (size_t)F * 3 * R, w->t_g2.b.p, sizeof(float) * 3 * R, hipMemcpyDeviceToDevice, s));
      MB_HIP(hipMemcpyAsync(F1 + (size_t)F * FC, w->t_f1.b.p, sizeof(float) * FC, hipMemcpyDeviceToDevice, s));
      MB_HIP(hipMemcpyAsync(F2 + (size_t)F * FC, w->t_f2.b.p, sizeof(float) * FC, hipMemcpyDeviceToDevice, s));
    }
    int nf = 0;
    RC(batch_folds(w, F, plan->fold_stride - (S - plan->fold_stride), S - plan->fold_stride, &nf));
    for (int f = 0; f < nf && !rc; ++f, ++fold) {
      MB_REQUIRE(fold < N, "wavernn_generate_batch: plan does not match the frame counts");
      int* d = &desc[(size_t)fold * 8];
      d[0] = f * plan->fold_stride; d[1] = T; d[2] = (int)u_row; d[3] = (int)frame_row; d[4] = F; d[5] = f;
      d[6] = (int)(uint32_t)h_seeds[u]; d[7] = (int)(uint32_t)(h_seeds[u] >> 32);
    }
    u_row += (size_t)F + 9; frame_row += (size_t)F + 1;
  }
  if (!rc && fold != N) { set_error("wavernn_generate_batch: plan has %d folds, frames give %d", N, fold); rc = MB_EINVAL; }
  if (rc) return rc;
  MB_HIP(hipMemcpyAsync(L.desc, desc.data(), sizeof(int) * desc.size(), hipMemcpyHostToDevice, s));
  MB_HIP(hipStreamSynchronize(s));  // `desc` is a host temporary
  MB_HIP(hipMemsetAsync(L.h1, 0, sizeof(float) * 2 * N * R, s));
  MB_HIP(hipMemsetAsync(L.h2, 0, sizeof(float) * 2 * N * R, s));
  MB_HIP(hipMemsetAsync(L.step, 0, sizeof(int) * 16, s));
  MB_HIP(hipMemsetAsync(L.slots, 0, sizeof(unsigned long long) * 2 * N, s));
  for (int g = 0; g < 2 && !rc; ++g) {  // P = W_hh.0 + b_hh for the first step
    RnnK k;
    memset(&k, 0, sizeof(k));
    k.w = g ? w->w_hh2.p : w->w_hh1.p; k.nseg = 1; k.nkb_total = R / 16; k.seg[0] = {g ? L.h2 : L.h1, R, R / 16, 0};
    k.N = N; k.units = 3 * R; k.biasX = g ? w->b_hh2.p : w->b_hh1.p; k.y = g ? L.P2 : L.P1; k.ldy = 3 * R;
    rc = rnn_launch(EPI_LINEAR, k, s);
  }
  if (rc) return rc;

  WfCond bcond;
  bcond.UT = L.UT; bcond.AT = L.AT; bcond.UI = L.UI; bcond.AI = L.AI; bcond.Kw = w->kw.p; bcond.hop = w->hop; bcond.frames = 0;
  // one time step = the split-hidden chain of mb_wavernn_generate with per-fold descriptors
  auto step = [&](int pp, int soff) -> int {
    float* h1p = L.h1 + (size_t)pp * N * R; float* h1n = L.h1 + (size_t)(pp ^ 1) * N * R;
    float* h2p = L.h2 + (size_t)pp * N * R; float* h2n = L.h2 + (size_t)(pp ^ 1) * N * R;
    unsigned long long* slot_prev = L.slots + (size_t)(pp ^ 1) * N;
    unsigned long long* slot_cur = L.slots + (size_t)pp * N;
    auto frame_rows = [&](RnnK& k) {
      k.fr_base = L.step; k.fr_off = soff; k.fr_n_off = 0; k.fr_fold_stride = plan->fold_stride;
      k.fr_total_len = 0; k.fr_hop = w->hop; k.fr_frames = 0; k.fr_desc = L.desc;
    };
    int r;
    Fin1K f;
    memset(&f, 0, sizeof(f));
    f.slot = slot_prev; f.cond = bcond; f.P1 = L.P1; f.g1 = w->g1I0.p; f.wI0 = w->wI0.p;
    f.h_prev = h1p; f.h_out = h1n; f.x_out = L.x1; f.samples = d_samples; f.progress = nullptr;
    f.step_base = L.step; f.step_off = soff; f.n_off = 0; f.nl = N; f.R = R; f.C = C; f.S = S;
    f.fold_stride = plan->fold_stride; f.total_len = 0; f.desc = L.desc;
    if ((r = rnn_launch_finish(f, s))) return r;
    RnnK k;
    memset(&k, 0, sizeof(k));
    k.w = w->w_rnn2x.p; k.nseg = 1; k.nkb_total = R / 16; k.seg[0] = {L.x1, R, R / 16, 0};
    k.w16 = allow_w16 ? w->q_rnn2.p : nullptr; k.w16_unscale = w->q_us[0]; k.range_word = w->d_range;
    k.N = N; k.units = R; k.h_pre = L.P2; k.pre_table = L.G2; frame_rows(k); k.pre_stride = 3 * R;
    k.h_prev = h2p; k.x_res = L.x1; k.h_out = h2n; k.x_out = L.x2; k.zero_slot = slot_prev;
    if ((r = rnn_launch(EPI_GRU, k, s))) return r;
    for (int g = 0; g < 2; ++g) {
      RnnK k1;
      memset(&k, 0, sizeof(k)); memset(&k1, 0, sizeof(k1));
      k.w = g ? w->w_fc2.p : w->w_fc1.p; k.nseg = 1;
      k.w16 = !allow_w16 ? nullptr : g ? w->q_fc2.p : w->q_fc1.p; k.w16_unscale = w->q_us[g ? 4 : 3]; k.range_word = w->d_range;
      k.seg[0] = {g ? L.y1 : L.x2, g ? FC : R, (g ? FC : R) / 16, 0};
      k.nkb_total = k.seg[0].nkb;
      k.N = N; k.units = FC; k.pre_table = g ? L.F2 : L.F1; frame_rows(k); k.pre_stride = FC;
      k.y = g ? L.y2 : L.y1; k.ldy = FC; k.act = 1;
      k1.w = g ? w->w_hh2.p : w->w_hh1.p; k1.nseg = 1; k1.nkb_total = R / 16; k1.seg[0] = {g ? h2n : h1n, R, R / 16, 0};
      k1.w16 = !allow_w16 ? nullptr : g ? w->q_hh2l.p : w->q_hh1l.p; k1.w16_unscale = w->q_us[g ? 7 : 6];  // (|h| < 1: no range word)
      k1.N = N; k1.units = 3 * R; k1.biasX = g ? w->b_hh2.p : w->b_hh1.p; k1.y = g ? L.P2 : L.P1; k1.ldy = 3 * R;
      if ((r = rnn_launch_dual_linear(k, k1, s))) return r;
    }
    memset(&k, 0, sizeof(k));
    k.w = w->w_fc3.p; k.nseg = 1; k.nkb_total = FC / 16; k.seg[0] = {L.y2, FC, FC / 16, 0};
    k.w16 = allow_w16 ? w->q_fc3.p : nullptr; k.w16_unscale = w->q_us[5]; k.range_word = w->d_range;
    k.N = N; k.units = C; k.biasX = w->b_fc3.p; k.ldy = C; frame_rows(k);
    k.gum_slot = slot_cur; k.gum_seed = 0;
    return rnn_launch(EPI_LINEAR, k, s);
  };
  MB_HIP(hipEventRecord(w->ev_t0, s));
  const bool use_graph = getenv("MBHIP_NO_GRAPH") == nullptr && S >= 64;
  int done = 0;
  if (use_graph) {
    int G = 128;
    while (G > S) G >>= 1;
    G &= ~1;
    if (G >= 2) {
      w->drop_graph();
      MB_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
      for (int i = 0; i < G && !rc; ++i) rc = step(i & 1, i);
      hipLaunchKernelGGL(wavernn_bump_kernel, dim3(1), dim3(1), 0, s, L.step, G);
      hipError_t e = hipStreamEndCapture(s, &w->graph[0]);
      if (rc) { w->drop_graph(); return rc; }
      if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture", __FILE__, __LINE__);
      e = hipGraphInstantiate(&w->graph_exec[0], w->graph[0], nullptr, nullptr, 0);
      if (e != hipSuccess) { w->drop_graph(); return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__); }
      const int reps = S / G;
      for (int r2 = 0; r2 < reps; ++r2) MB_HIP(hipGraphLaunch(w->graph_exec[0], s));
      done = reps * G;
    }
  }
  for (int i = done; i < S && !rc; ++i) rc = step(i & 1, i - done);
  if (rc) return rc;
  hipLaunchKernelGGL(wavernn_flush_kernel, dim3(cdiv(N, 64)), dim3(64), 0, s, L.slots + (size_t)((S - 1) & 1) * N, d_samples, N, S, C, 0);
  MB_HIP(hipGetLastError());
  MB_HIP(hipEventRecord(w->ev_t1, s));
  w->last_launches = 5 * S; w->last_lanes = 1; w->timed = true;
  MB_HIP(hipEventRecord(w->ev_out, s));
  MB_HIP(hipStreamWaitEvent(cs, w->ev_out, 0));
#undef RC
  return MB_OK;
}

extern "C" int mb_wavernn_generate_batch(const mb_wavernn* wc, const mb_wavernn_batch_plan* plan, const int* h_frames,
                                         const float* const* h_d_mels, const uint64_t* h_seeds, float* d_samples,
                                         void* d_workspace, size_t workspace_bytes, mb_stream_t stream) {
  mb_wavernn* w = const_cast<mb_wavernn*>(wc);
  MB_REQUIRE(w && plan, "wavernn_generate_batch: null pointer");
  MB_REQUIRE(w->d_range && w->h_range, "wavernn_generate_batch: handle without its range word");
  w->last_path = MB_WRN_PATH_CHAIN; w->last_fallback = MB_WRN_FALLBACK_NONE;
  // > 64 columns: the GEMMs of the loop are rnn_ts3_body's operand pairs (|x| <= 65504).  Their range word is read behind the loop --
  // the call is HOST-BLOCKING there -- and a raised word reruns the whole loop on the fp32 instances (rnn_ts2_body.h): no silent clamp.
  const bool pairs = plan->n_folds > 64;
  if (pairs) MB_HIP(hipMemsetAsync(w->d_range, 0, sizeof(int), w->loop_stream));  // (ordered before the loop: same stream)
  int rc = wavernn_join_on_error(w, wavernn_generate_batch_impl(wc, plan, h_frames, h_d_mels, h_seeds, d_samples, d_workspace,
                                                                workspace_bytes, stream, true));
  if (rc || !pairs) return rc;
  MB_HIP(hipMemcpyAsync(w->h_range, w->d_range, sizeof(int), hipMemcpyDeviceToHost, w->loop_stream));
  MB_HIP(hipStreamSynchronize(w->loop_stream));
  if (!*w->h_range) return MB_OK;
  static bool warned = false;
  if (!warned) {
    fprintf(stderr, "[mbhip] wavernn batch: an activation left the operand-pair GEMMs' range (|x| > 65504 or NaN); rerunning on the fp32 instances\n");
    warned = true;
  }
  w->last_fallback = MB_WRN_FALLBACK_RANGE;
  return wavernn_join_on_error(w, wavernn_generate_batch_impl(wc, plan, h_frames, h_d_mels, h_seeds, d_samples, d_workspace,
                                                              workspace_bytes, stream, false));
}

// Test hook (tests/test_wavernn_gpu.py): the Exp(1) draws E[step][fold][class] = -log u that every fused Gumbel-argmax
// sampler of the production paths consumes -- wf_fc3_kernel, the gum epilogue of rnn_body.h / rnn_ts2_body.h,
// wf_persist(1)_kernel: Philox(counter = (step, fold, class / 4, 'WAVE'), key = seed), word class % 4 -- so that the
// oracle can run sample_loop(noise = E) against the DEFAULT generate(seed) path (argmax_c l_c - log E_c ==
// argmax_c softmax(l)_c / E_c, torch.multinomial's rule, fatchord_version.py:222-226).
__global__ void wavernn_debug_noise_kernel(unsigned long long seed, int step0, int steps, int folds, int C4, float4* out) {
  const size_t total = (size_t)steps * folds * C4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(idx % C4);
    const size_t r = idx / C4;
    const int n = (int)(r % folds), s = step0 + (int)(r / folds);
    uint32_t g[4];
    mb::philox4x32((uint32_t)s, (uint32_t)n, (uint32_t)q, 0x57415645u, (uint32_t)seed, (uint32_t)(seed >> 32), g);
    out[idx] = make_float4(-logf(mb::u32_to_unit(g[0])), -logf(mb::u32_to_unit(g[1])), -logf(mb::u32_to_unit(g[2])), -logf(mb::u32_to_unit(g[3])));
  }
}
extern "C" int mb_wavernn_debug_noise(uint64_t seed, int step0, int steps, int folds, int n_classes, float* d_out, mb_stream_t stream) {
  MB_REQUIRE(d_out && steps >= 1 && folds >= 1 && step0 >= 0 && n_classes >= 4 && n_classes % 4 == 0, "wavernn_debug_noise: bad arguments");
  const size_t total = (size_t)steps * folds * (n_classes / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(wavernn_debug_noise_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed, step0, steps, folds,
                     n_classes / 4, reinterpret_cast<float4*>(d_out));
  MB_HIP(hipGetLastError());
  return MB_OK;
}

// the same hook for MOL mode: the uniform_(1e-5, 1 - 1e-5) draws [step][fold][nr_mix + 1] (nr_mix mixture-indicator draws, then
// the logistic draw) of wavernn_sample_mol_kernel / wf_fc3_mol_kernel: Philox(counter = (step, fold, m / 4, 'MOL!'), key = seed), word m % 4
__global__ void wavernn_debug_noise_mol_kernel(unsigned long long seed, int step0, int steps, int folds, int nr_mix, float* out) {
  const int W = nr_mix + 1;
  const size_t total = (size_t)steps * folds * W;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx % W);
    const size_t r = idx / W;
    const int n = (int)(r % folds), s = step0 + (int)(r / folds);
    uint32_t g[4];
    mb::philox4x32((uint32_t)s, (uint32_t)n, (uint32_t)(m >> 2), 0x4d4f4c21u, (uint32_t)seed, (uint32_t)(seed >> 32), g);
    out[idx] = 1e-5f + (1.0f - 2e-5f) * mb::u32_to_unit(g[m & 3]);
  }
}
extern "C" int mb_wavernn_debug_noise_mol(uint64_t seed, int step0, int steps, int folds, int nr_mix, float* d_out, mb_stream_t stream) {
  MB_REQUIRE(d_out && steps >= 1 && folds >= 1 && step0 >= 0 && nr_mix >= 1 && nr_mix <= 63, "wavernn_debug_noise_mol: bad arguments");
  const size_t total = (size_t)steps * folds * (nr_mix + 1);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(wavernn_debug_noise_mol_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed, step0, steps, folds,
                     nr_mix, d_out);
  MB_HIP(hipGetLastError());
  return MB_OK;
}
extern "C" int mb_wavernn_last_path(const mb_wavernn* w, int* path, int* fallback) {
  MB_REQUIRE(w, "wavernn_last_path: null handle");
  if (path) *path = w->last_path;
  if (fallback) *fallback = w->last_fallback;
  return MB_OK;
}
extern "C" int mb_wavernn_last_loop_ms(const mb_wavernn* w, float* ms, int* launches) {
  MB_REQUIRE(w && ms, "wavernn_last_loop_ms: null pointer");
  if (!w->timed) { set_error("wavernn_last_loop_ms: no generate call yet"); return MB_ESTATE; }
  MB_HIP(hipEventSynchronize(w->ev_t1));
  MB_HIP(hipEventElapsedTime(ms, w->ev_t0, w->ev_t1));
  if (launches) *launches = w->last_launches;
  return MB_OK;
}

extern "C" int mb_wavernn_bench_kernel(mb_wavernn* w, const mb_wavernn_plan* plan, const float* d_mel,
                                       float* d_samples, void* d_workspace, size_t workspace_bytes,
                                       int which, int iters, float* avg_us, double* algorithmic_bytes,
                                       mb_stream_t stream) {
  // In-situ marginal duration: time the real (graph-replayed) sample loop with HIP events on its
  // stream, with and without kernel `which`; kernels of the chain run back to back, so the
  // difference per step is that kernel's launch-to-launch duration (what rocprofv3 reports).
  MB_REQUIRE(w && plan && avg_us && which >= 0 && which < 5, "wavernn_bench_kernel: which must be 0..4");
  (void)iters;
  float ms_full = 0.f, ms_wo = 0.f;
  int rc = MB_OK;
  w->bench_chain = true;
  for (int pass = 0; pass < 2 && !rc; ++pass) {
    w->bench_which = pass ? (1 << which) : 0;
    rc = mb_wavernn_generate(w, plan, d_mel, nullptr, 0, d_samples, nullptr, nullptr, nullptr, d_workspace,
                             workspace_bytes, stream);
    if (!rc) rc = mb_wavernn_last_loop_ms(w, pass ? &ms_wo : &ms_full, nullptr);
  }
  w->bench_which = 0;
  w->bench_chain = false;
  if (rc) return rc;
  const float ms = ms_full - ms_wo;
  iters = plan->seq_len;
  *avg_us = ms * 1000.f / iters;
  if (algorithmic_bytes) {
    const double R = w->cfg.rnn_dims, FC = w->cfg.fc_dims, C = w->n_classes, N = plan->n_folds;
    double b = 0;
    if (wavernn_split_chain()) {
      switch (which) {  // split-hidden chain (A..E of mb_wavernn_generate), fp32
        case 0: b = N * 3 * R * 2 + N * R * 2 + 4 * R + 2 * N * R; break;                     // P1, T1 rows, Ipre rows, h1 | g1, wI0 | h1', x1
        case 1: b = 3 * R * R + N * R + N * 3 * R + N * 3 * R + N * R + 2 * N * R; break;     // W_ih2 | x1, P2, G2 rows, h2 | h2', x2
        case 2: b = FC * R + 3 * R * R + N * R + 2 * N * FC + N * R + 3 * R + N * 3 * R; break;   // fc1 + W_hh1 | x2, F1 rows, y1 | h1, b_hh1, P1
        case 3: b = FC * FC + 3 * R * R + N * FC + 2 * N * FC + N * R + 3 * R + N * 3 * R; break; // fc2 + W_hh2
        default: b = C * FC + N * FC + N * C + C; break;
      }
    } else
    switch (which) {  // weights once + activation vectors in/out + table rows, fp32
      case 0: b = 3 * R * 2 * R + 2 * N * R + 2 * N * R + 6 * R; break;
      case 1: b = 3 * R * 2 * R + 2 * N * R + 2 * N * R + 3 * R + N * 3 * R; break;
      case 2: b = FC * R + N * R + N * FC + N * FC; break;
      case 3: b = FC * FC + N * FC + N * FC + N * FC; break;
      case 4: b = C * FC + N * FC + N * C + C; break;
      default: b = N * C + N * R * 2 + R; break;
    }
    *algorithmic_bytes = b * 4.0;
  }
  return MB_OK;
}
