// Row-tile MFMA recurrent GEMM kernels (see rnn.h for the design).
//
// What bounds these launches on MI355X (profiles/r01_wavernn_step_timeline.md): each one is a few us of
// dependent latencies, not bandwidth --
//   * ~1.5 us launch-to-launch gap, ~0.5 us for the first instruction fetch of a launch (the
//     instruction cache is cold at every dispatch; sequential fetch then keeps up, but every taken
//     branch into a line that was not prefetched pays the miss again),
//   * one ~0.9 us round trip for data the PREVIOUS launch wrote (it comes from another XCD's L2 via
//     the fabric), ~0.2 us for data that is stable across launches (weights, tables, biases),
//   * the fp32 MFMA chain itself (256 FLOP/clk/CU).
// Hence: (1) the kernel is specialised at compile time on the feature set the caller uses (F), so a
// launch executes straight-line code with no dead branches; (2) every load is issued before the
// first wait; (3) nothing the previous launch wrote is read except the activations themselves.
#include "rnn_ts3_body.h"

namespace mb {

template <int EPI, int NT, int UB, unsigned F>
__global__ __launch_bounds__(512) void rnn_rowtile_kernel(RnnDev d) {
  rnn_rowtile_body<EPI, NT, UB, F>(d, blockIdx.x, blockIdx.y);
}

// ---- gru1-finish job (Fin1K, rnn.h): one thread per (fold, unit), all loads issued before first use ----
__device__ __forceinline__ void gru1_finish_body(const Fin1K& a, const int block, const int nthreads) {
  const int idx = block * nthreads + threadIdx.x;
  const int H = a.R;
  int n = idx / H;
  const int j = idx - n * H;
  const bool live = n < a.nl;
  if (!live) n = a.nl - 1;  // clamped: loads legal, nothing stored (keeps every thread at the barrier below)
  const int s = *a.step_base + a.step_off;
  const float* p1 = a.P1 + (size_t)n * 3 * H + j;
  const float hr = p1[0], hz = p1[H], hn = p1[2 * H];
  const float hp = a.h_prev[(size_t)n * H + j];
  float4 tq;
  if (a.desc) {  // several utterances: (pos0 of the fold, total_len_u, first U row, first A row, frames_u, ...)
    const int4 d0 = *reinterpret_cast<const int4*>(a.desc + (size_t)(a.n_off + n) * 8);
    const int frames_u = a.desc[(size_t)(a.n_off + n) * 8 + 4];
    tq = wf_cond_row4(a.cond, (unsigned)(d0.x + s), (unsigned)d0.y, j, H, frames_u, d0.z, d0.w);
  } else {
    tq = wf_cond_row4(a.cond, (unsigned)(a.n_off + n) * (unsigned)a.fold_stride + (unsigned)s, (unsigned)a.total_len, j, H, a.cond.frames);
  }
  const float tr = tq.x, tz = tq.y, tn = tq.z, ip = tq.w;
  const float gr = a.g1[j], gz = a.g1[H + j], gn = a.g1[2 * H + j], w0 = a.wI0[j];
  const unsigned long long slot = a.slot[n];
  if (!live) return;
  const float x = slot ? 2.f * (float)argmax_class(slot) / ((float)a.C - 1.f) - 1.f : 0.f;
  // torch GRUCell, gate order (r, z, n)
  const float rg = sigmoidf_((tr + x * gr) + hr);
  const float zg = sigmoidf_((tz + x * gz) + hz);
  const float ng = tanhf((tn + x * gn) + rg * hn);
  const float hy = ng + zg * (hp - ng);
  a.h_out[(size_t)n * H + j] = hy;
  a.x_out[(size_t)n * H + j] = (ip + x * w0) + hy;
  if (j == 0 && s > 0) {  // previous step's sample -> output tensor
    a.samples[(size_t)(a.n_off + n) * a.S + (s - 1)] = x;
    if (a.progress && a.n_off + n == 0 && (s - 1) % 100 == 0) *a.progress = s;
  }
}

__global__ __launch_bounds__(256) void wavernn_gru1_finish_kernel(Fin1K a) {
  trace_begin(a.trace);
  gru1_finish_body(a, blockIdx.x, 256);
  trace_end(a.trace);
}

// (fc3 + the NEXT step's gru1-finish in one launch through an arrival counter was measured in rounds 1 and 4 -- 26.4 against 25.7 us per
//  step at 23 columns, 73-94 against 62 at 736 -- and is not part of the library: DESIGN 4i, tools/rejected/.)

// Two independent LINEAR jobs in ONE launch: workgroups with blockIdx.x < nx0 run job 0 (the one on
// the critical path: they are dispatched first), the rest run job 1 on the CUs job 0 leaves idle.
// The WaveRNN loop uses it to compute the hidden halves W_hh.h + b_hh of the NEXT step's GRUs beside
// fc1 / fc2 (wavernn.hip), which takes them off the dependent chain.
template <int UB0, unsigned F0, int UB1, unsigned F1>
__global__ __launch_bounds__(512) void rnn_dual_linear_kernel(RnnDev d0, RnnDev d1, int nx0) {
  if ((int)blockIdx.x < nx0) rnn_rowtile_body<EPI_LINEAR, 1, UB0, F0>(d0, blockIdx.x, blockIdx.y);
  else rnn_rowtile_body<EPI_LINEAR, 1, UB1, F1>(d1, blockIdx.x - nx0, blockIdx.y);
}
// (17..64 fold columns with two 16-column tiles per workgroup -- one weight fetch for both -- was measured in round 1 and not adopted:
//  29.1 us per step against 26.0; the loop is latency-bound, not bound by the weight stream.  Removed from the library in round 4.)

// Tile-split launches (rnn_body.h, TS): wide batches (several utterances per WaveRNN loop)
template <int EPI, unsigned F>
__global__ __launch_bounds__(512) void rnn_ts_kernel(RnnDev d) {
  rnn_rowtile_body<EPI, 1, 8, F, true>(d, blockIdx.x, blockIdx.y);
}
template <unsigned F0, unsigned F1>
__global__ __launch_bounds__(512) void rnn_dual_linear_ts_kernel(RnnDev d0, RnnDev d1, int nx0) {
  if ((int)blockIdx.x < nx0) rnn_rowtile_body<EPI_LINEAR, 1, 8, F0, true>(d0, blockIdx.x, blockIdx.y);
  else rnn_rowtile_body<EPI_LINEAR, 1, 8, F1, true>(d1, blockIdx.x - nx0, blockIdx.y);
}
#ifndef MB_TS2_FC3_MT
#define MB_TS2_FC3_MT 2  // wave tile of the fc3 + sampler launch (32 row tiles only; 1x1, 1x2, 1x3 measured no better)
#define MB_TS2_FC3_NT 1
#endif
// Register-tiled wide form (rnn_ts2_body.h): 4 waves, each MT row tiles x NT column tiles, one workgroup per CU
template <int EPI, unsigned F, int MT, int NT>
__global__ __launch_bounds__(256) void rnn_ts2_kernel(RnnDev d) {
  rnn_ts2_body<EPI, F, MT, NT>(d, blockIdx.x, blockIdx.y);
}
template <unsigned F0, unsigned F1, int MT, int NT>
__global__ __launch_bounds__(256) void rnn_dual_linear_ts2_kernel(RnnDev d0, RnnDev d1, int nx0) {
  if ((int)blockIdx.x < nx0) rnn_ts2_body<EPI_LINEAR, F0, MT, NT>(d0, blockIdx.x, blockIdx.y);
  else rnn_ts2_body<EPI_LINEAR, F1, MT, NT>(d1, blockIdx.x - nx0, blockIdx.y);
}
// The same wave tiling on the fp16 matrix pipe (rnn_ts3_body.h: error-compensated products, activations split once per workgroup in LDS)
template <int EPI, unsigned F, int MT, int NT>
__global__ __launch_bounds__(256) void rnn_ts3_kernel(RnnDev d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ts3_lds[];
  rnn_ts3_body<EPI, F, MT, NT>(d, blockIdx.x, blockIdx.y, reinterpret_cast<t3h*>(ts3_lds));
}
template <unsigned F0, unsigned F1, int MT, int NT>
__global__ __launch_bounds__(256) void rnn_dual_linear_ts3_kernel(RnnDev d0, RnnDev d1, int nx0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ts3_lds[];
  if ((int)blockIdx.x < nx0) rnn_ts3_body<EPI_LINEAR, F0, MT, NT>(d0, blockIdx.x, blockIdx.y, reinterpret_cast<t3h*>(ts3_lds));
  else rnn_ts3_body<EPI_LINEAR, F1, MT, NT>(d1, blockIdx.x - nx0, blockIdx.y, reinterpret_cast<t3h*>(ts3_lds));
}
// Wide batches (> 64 fold columns): MBHIP_RNN_WIDE = ts3 (default: fp16 pipe, rnn_ts3_body.h) | ts2 (fp32 register-tiled form,
// rnn_ts2_body.h) | ts (first wide form, rnn_body.h TS), optionally ":<column tiles per wave>" (1..3) for the parity tests
static int rnn_wide_form(int* nt_forced = nullptr) {  // 3 / 2 / 1
  if (nt_forced) *nt_forced = 0;
  const char* e = getenv("MBHIP_RNN_WIDE");
  if (!e) return 3;
  if (nt_forced) if (const char* c = strchr(e, ':')) *nt_forced = atoi(c + 1);
  return strncmp(e, "ts2", 3) == 0 ? 2 : (strncmp(e, "ts3", 3) == 0 ? 3 : (strncmp(e, "ts", 2) == 0 ? 1 : 3));
}
static bool rnn_ts3_enabled() { return rnn_wide_form() == 3; }
bool rnn_wide_switch_valid() {
  const char* e = getenv("MBHIP_RNN_WIDE");
  if (!e) return true;
  std::string form(e), nt;
  const size_t c = form.find(':');
  if (c != std::string::npos) { nt = form.substr(c + 1); form = form.substr(0, c); }
  if (form != "ts3" && form != "ts2" && form != "ts") return false;
  return c == std::string::npos || nt == "1" || nt == "2" || nt == "3";
}
// Column tiles per wave of the register-tiled wide form, 0 = use the first wide form (rnn_body.h TS).
// 128 row tiles in pieces of 2 make 16 workgroup rows, so the piece must be narrow enough for >= 256 workgroups:
// 3 column tiles from 44 column tiles up (736 columns: exactly one piece per SIMD), 2 from 28, 1 from 14; narrower
// batches keep the 8-wave one-tile-per-wave form, which cuts the same work into four times as many workgroups.
// MBHIP_RNN_WIDE=ts forces that form, "ts3:2" / "ts2:1" force a piece width (parity tests, A/B runs).
static int rnn_ts2_nt(int N) {
  int nt_forced = 0;
  if (rnn_wide_form(&nt_forced) == 1) return 0;
  if (nt_forced >= 1 && nt_forced <= 3) return nt_forced;
  const int ct = cdiv(N, 16);
  return ct >= 44 ? 3 : ct >= 28 ? 2 : ct >= 14 ? 1 : 0;
}
static bool rnn_ts_enabled(int N) { return N > 64; }

// Two independent GRU steps in ONE launch (the forward and backward directions of a bidirectional scan,
// cbhg.py:76-77): blockIdx.x < nx0 -> job 0, else job 1.  Halves the launch count of the CBHG scans.
template <int UB, unsigned F>
__global__ __launch_bounds__(512) void rnn_dual_gru_kernel(RnnDev d0, RnnDev d1, int nx0) {
  if ((int)blockIdx.x < nx0) rnn_rowtile_body<EPI_GRU, 1, UB, F>(d0, blockIdx.x, blockIdx.y);
  else rnn_rowtile_body<EPI_GRU, 1, UB, F>(d1, blockIdx.x - nx0, blockIdx.y);
}

void pack_rowtile(const float* rows, int n_live_rows, int K, int RL, std::vector<float>* out) {
  const int per_tile = 4 * RL;
  const int n_mt = (n_live_rows + per_tile - 1) / per_tile;
  const int nkb = K / 16;
  out->assign((size_t)n_mt * nkb * per_tile * 16, 0.f);
  for (int mt = 0; mt < n_mt; ++mt)
    for (int kb = 0; kb < nkb; ++kb)
      for (int r = 0; r < per_tile; ++r) {
        const int row = mt * per_tile + r;
        if (row >= n_live_rows) continue;
        float* dst = out->data() + (((size_t)mt * nkb + kb) * per_tile + r) * 16;
        const float* src = rows + (size_t)row * K + kb * 16;
        for (int q = 0; q < 16; ++q) dst[q] = src[q];
      }
}

void cell_rows(const float* w_ih, int kx, int ldx, const float* w_hh, int kh, int H, int G,
               std::vector<float>* rows) {
  const int K = kx + kh;
  rows->assign((size_t)H * G * K, 0.f);
  for (int j = 0; j < H; ++j)
    for (int g = 0; g < G; ++g) {
      float* dst = rows->data() + ((size_t)j * G + g) * K;
      memcpy(dst, w_ih + (size_t)(g * H + j) * ldx, sizeof(float) * kx);
      memcpy(dst + kx, w_hh + (size_t)(g * H + j) * kh, sizeof(float) * kh);
    }
}

// Specialised instances: the exact (epilogue, column tiles, batch depth, feature set) tuples the two
// autoregressive loops and the CBHG scan launch.  Anything else runs the RF_GENERIC instance.
#define ACT(n) ((unsigned)(n) << RF_ACT_SHIFT)
#define MB_RNN_INSTANCES(X)                                                                               \
  /* WaveRNN: rnn1 (exact / fused), rnn2 (+ slot clear), fc1|fc2, fc3 (exact / fused) */                  \
  X(EPI_GRU, 1, 8, RF_BIASX | RF_BIASH | RF_XRES | RF_XOUT | RF_MULTISEG)                                 \
  X(EPI_GRU, 1, 8, RF_BIASX | RF_BIASH | RF_FRAME | RF_AFFINE | RF_XOUT | RF_MULTISEG)                    \
  X(EPI_GRU, 1, 8, RF_PRE | RF_FRAME | RF_BIASH | RF_XRES | RF_XOUT | RF_MULTISEG)                        \
  X(EPI_GRU, 1, 8, RF_PRE | RF_FRAME | RF_BIASH | RF_XRES | RF_XOUT | RF_MULTISEG | RF_ZERO)              \
  X(EPI_LINEAR, 1, 4, RF_PRE | RF_FRAME | ACT(1))                                                         \
  X(EPI_LINEAR, 1, 4, RF_BIASX)                                                                           \
  X(EPI_LINEAR, 1, 4, RF_BIASX | RF_FRAME | RF_GUMBEL)                                                    \
  /* WaveRNN split-hidden chain: rnn2 on its input half only (hidden half precomputed) */                 \
  X(EPI_GRU, 1, 4, RF_PRE | RF_FRAME | RF_HPRE | RF_XRES | RF_XOUT | RF_ZERO)                             \
  /* ... and with per-fold descriptors (several utterances per loop) */                                   \
  X(EPI_GRU, 1, 4, RF_PRE | RF_FRAME | RF_HPRE | RF_XRES | RF_XOUT | RF_ZERO | RF_FOLDTAB)                \
  X(EPI_LINEAR, 1, 4, RF_BIASX | RF_FRAME | RF_GUMBEL | RF_FOLDTAB)                                       \
  /* Tacotron decoder: prenet fc1/fc2 (mask / on-device dropout), attention GRU, rnn_input, LSTMs, mel */ \
  X(EPI_LINEAR, 1, 2, RF_BIASX | RF_SKIP | RF_MASK | ACT(1))                                              \
  X(EPI_LINEAR, 1, 2, RF_BIASX | RF_SKIP | RF_DROP | ACT(1))                                              \
  X(EPI_GRU, 1, 8, RF_BIASX | RF_BIASH | RF_SKIP | RF_MULTISEG)                                           \
  X(EPI_LINEAR, 1, 8, RF_BIASX | RF_SKIP | RF_MULTISEG)                                                   \
  X(EPI_LSTM, 2, 4, RF_BIASX | RF_BIASH | RF_XRES | RF_XOUT | RF_SKIP | RF_MULTISEG)                      \
  X(EPI_LSTM, 1, 8, RF_BIASX | RF_BIASH | RF_XRES | RF_XOUT | RF_SKIP | RF_MULTISEG)                      \
  /* decoder LSTMs on their input half (hidden half precomputed beside the attention launch) */            \
  X(EPI_LSTM, 2, 4, RF_BIASX | RF_HPRE | RF_XRES | RF_XOUT | RF_SKIP)                                     \
  X(EPI_LSTM, 1, 8, RF_BIASX | RF_HPRE | RF_XRES | RF_XOUT | RF_SKIP)                                     \
  X(EPI_LINEAR, 1, 8, RF_SKIP)                                                                            \
  /* ppg2mel decoder: attention / decoder LSTMCells (no residual), projection + stop rows, query layer */ \
  X(EPI_LSTM, 1, 4, RF_BIASX | RF_BIASH | RF_SKIP | RF_MULTISEG)                                          \
  X(EPI_LSTM, 1, 8, RF_BIASX | RF_BIASH | RF_SKIP | RF_MULTISEG)                                          \
  X(EPI_LINEAR, 1, 4, RF_BIASX | RF_SKIP | RF_MULTISEG)                                                   \
  X(EPI_LINEAR, 1, 4, RF_BIASX | RF_SKIP | ACT(1))                                                        \
  /* CBHG bidirectional GRU scan */                                                                       \
  X(EPI_GRU, 1, 2, RF_PRE | RF_BIASH | RF_SEQ)

int rnn_launch_dual_linear(const RnnK& k0, const RnnK& k1, hipStream_t s) {
  constexpr int NW = 8;
  MB_REQUIRE(k0.N >= 1 && k0.N == k1.N && k0.nseg == 1 && k1.nseg == 1, "rnn_launch_dual: bad shape");
  RnnDev d0, d1;
  int rc = make_rnn_dev(k0, &d0);
  if (!rc) rc = make_rnn_dev(k1, &d1);
  if (rc) return rc;
  const int nx0 = cdiv(k0.units, 16), nx1 = cdiv(k1.units, 16);
  const unsigned f0 = rnn_features(EPI_LINEAR, k0), f1 = rnn_features(EPI_LINEAR, k1);
  constexpr unsigned F0 = RF_PRE | RF_FRAME | (1u << RF_ACT_SHIFT), F1 = RF_BIASX;
  const int pw0 = cdiv(k0.nkb_total, NW), pw1 = cdiv(k1.nkb_total, NW);
  MB_REQUIRE((f0 == F0 || f0 == (F0 | RF_FOLDTAB)) && f1 == F1 && pw0 == 4 && pw1 == 4,
             "rnn_launch_dual: only the (relu table linear, biased linear) K=512 pair is instantiated (features %x/%x)", f0, f1);
  const int nt2 = rnn_ts_enabled(k0.N) ? rnn_ts2_nt(k0.N) : 0;
  if (f0 == (F0 | RF_FOLDTAB) && nt2 && k0.nkb_total % 8 == 0 && k1.nkb_total % 8 == 0) {
    constexpr int MT = 2;  // workgroup = 8 row tiles x nt2 column tiles (rnn_ts2_body.h)
    dim3 g2(cdiv(nx0, MT * TS2_WAVES) + cdiv(nx1, MT * TS2_WAVES), cdiv(cdiv(k0.N, 16), nt2));
    MB_REQUIRE(k0.nkb_total == k1.nkb_total, "rnn_launch_dual(ts2): jobs must share K");
    const int nxw = cdiv(nx0, MT * TS2_WAVES);
    if (k0.w16 && k1.w16 && rnn_ts3_enabled()) {  // fp16 matrix pipe, error-compensated (rnn_ts3_body.h)
      d0.k.dbg = d1.k.dbg = diag_int("ts3_dbg", 0);  // diagnostics of rnn_ts3_body.h (wrong results on purpose); read on this path only
      if (nt2 == 3) hipLaunchKernelGGL((rnn_dual_linear_ts3_kernel<F0 | RF_FOLDTAB, F1, MT, 3>), g2, dim3(256), ts3_lds_bytes<3>(), s, d0, d1, nxw);
      else if (nt2 == 2) hipLaunchKernelGGL((rnn_dual_linear_ts3_kernel<F0 | RF_FOLDTAB, F1, MT, 2>), g2, dim3(256), ts3_lds_bytes<2>(), s, d0, d1, nxw);
      else hipLaunchKernelGGL((rnn_dual_linear_ts3_kernel<F0 | RF_FOLDTAB, F1, MT, 1>), g2, dim3(256), ts3_lds_bytes<1>(), s, d0, d1, nxw);
      MB_HIP(hipGetLastError());
      return MB_OK;
    }
    if (nt2 == 3) hipLaunchKernelGGL((rnn_dual_linear_ts2_kernel<F0 | RF_FOLDTAB, F1, MT, 3>), g2, dim3(256), 0, s, d0, d1, nxw);
    else if (nt2 == 2) hipLaunchKernelGGL((rnn_dual_linear_ts2_kernel<F0 | RF_FOLDTAB, F1, MT, 2>), g2, dim3(256), 0, s, d0, d1, nxw);
    else hipLaunchKernelGGL((rnn_dual_linear_ts2_kernel<F0 | RF_FOLDTAB, F1, MT, 1>), g2, dim3(256), 0, s, d0, d1, nxw);
    MB_HIP(hipGetLastError());
    return MB_OK;
  }
  if (f0 == (F0 | RF_FOLDTAB) && rnn_ts_enabled(k0.N) && k0.nkb_total % 8 == 0 && k1.nkb_total % 8 == 0) {  // wide batch: tile-split form, 2 row tiles x 4 column tiles per workgroup
    dim3 gts(cdiv(nx0, 2) + cdiv(nx1, 2), cdiv(cdiv(k0.N, 16), 4));
    MB_REQUIRE(k0.nkb_total == k1.nkb_total, "rnn_launch_dual(ts): jobs must share K");
    hipLaunchKernelGGL((rnn_dual_linear_ts_kernel<F0 | RF_FOLDTAB, F1>), gts, dim3(NW * 64), 0, s, d0, d1, cdiv(nx0, 2));
    MB_HIP(hipGetLastError());
    return MB_OK;
  }
  dim3 grid(nx0 + nx1, cdiv(k0.N, 16));
  if (f0 == F0) hipLaunchKernelGGL((rnn_dual_linear_kernel<4, F0, 4, F1>), grid, dim3(NW * 64), 0, s, d0, d1, nx0);
  else hipLaunchKernelGGL((rnn_dual_linear_kernel<4, F0 | RF_FOLDTAB, 4, F1>), grid, dim3(NW * 64), 0, s, d0, d1, nx0);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

int rnn_launch_dual_gru(const RnnK& k0, const RnnK& k1, hipStream_t s) {
  constexpr int NW = 8;
  constexpr unsigned F = RF_PRE | RF_BIASH | RF_SEQ;
  MB_REQUIRE(k0.N == k1.N && k0.units == k1.units && k0.nkb_total == k1.nkb_total && k0.nseg == 1 && k1.nseg == 1,
             "rnn_launch_dual_gru: the two directions must have the same shape");
  const bool inst = rnn_features(EPI_GRU, k0) == F && rnn_features(EPI_GRU, k1) == F && cdiv(k0.nkb_total, NW) <= 2;
  if (!inst || getenv("MBHIP_RNN_GENERIC")) {  // not the instantiated scan shape: two plain launches
    int rc = rnn_launch(EPI_GRU, k0, s);
    return rc ? rc : rnn_launch(EPI_GRU, k1, s);
  }
  RnnDev d0, d1;
  int rc = make_rnn_dev(k0, &d0);
  if (!rc) rc = make_rnn_dev(k1, &d1);
  if (rc) return rc;
  const int nx = cdiv(k0.units, 4);
  dim3 grid(2 * nx, cdiv(k0.N, 16));
  hipLaunchKernelGGL((rnn_dual_gru_kernel<2, F>), grid, dim3(NW * 64), 0, s, d0, d1, nx);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

int rnn_launch_finish(const Fin1K& f, hipStream_t s) {
  MB_REQUIRE(f.nl >= 1 && f.R >= 1, "rnn_launch_finish: bad shape");
  hipLaunchKernelGGL(wavernn_gru1_finish_kernel, dim3(cdiv(f.nl * f.R, 256)), dim3(256), 0, s, f);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

int rnn_launch(int epi, const RnnK& k, hipStream_t s) {
  MB_REQUIRE(k.N >= 1 && k.units >= 1 && k.nseg >= 1 && k.nseg <= 4, "rnn_launch: bad shape");
  MB_REQUIRE(!k.aff_slot || (k.fr_base && k.nseg >= 1 && epi == EPI_GRU), "rnn_launch: rebuilt segment needs the fold geometry");
  MB_REQUIRE(!k.gum_slot || (k.fr_base && epi == EPI_LINEAR && k.units % 4 == 0), "rnn_launch: fused sampler needs the step index");
  MB_REQUIRE(!k.h_pre || (epi != EPI_LINEAR && !k.biasH), "rnn_launch: h_pre is a GRU/LSTM feature and already holds b_hh");
  constexpr int NW = 8;
  const int n_mt = (epi == EPI_LINEAR) ? cdiv(k.units, 16) : cdiv(k.units, 4);
  // More than 16 columns AND a weight matrix big enough to be bandwidth-bound (the batch-32
  // Tacotron LSTMs, 33.5 MB): one workgroup covers two 16-column tiles so the weights are
  // streamed once per 32 columns.  Small matrices (WaveRNN, <= 6.6 MB) are latency-bound and run
  // faster with twice the workgroups.
  const int rl = (epi == EPI_GRU) ? 3 : 4;
  const size_t wbytes = (size_t)n_mt * k.nkb_total * 4 * rl * 16 * sizeof(float);
  const unsigned feat = rnn_features(epi, k);
  const int nt = (k.N > 16 && wbytes >= ((size_t)12 << 20)) ? 2 : 1;
  RnnDev d;
  int rcd = make_rnn_dev(k, &d);
  if (rcd) return rcd;
  const int per_wave = cdiv(k.nkb_total, NW);
  const int ub = nt == 2 ? (per_wave >= 4 ? 4 : 2) : (per_wave >= 8 ? 8 : (per_wave >= 4 ? 4 : 2));
  dim3 grid(n_mt, cdiv(k.N, 16 * nt));
  bool done = false;
  if (rnn_ts_enabled(k.N) && k.nseg == 1 && k.nkb_total % 8 == 0) {  // wide batch: tile-split instances (whole batches of 8 k-blocks)
    constexpr unsigned FG = RF_PRE | RF_FRAME | RF_HPRE | RF_XRES | RF_XOUT | RF_ZERO | RF_FOLDTAB;
    constexpr unsigned FL = RF_BIASX | RF_FRAME | RF_GUMBEL | RF_FOLDTAB;
    dim3 gts(cdiv(n_mt, 2), cdiv(cdiv(k.N, 16), 4));
    if (const int nt2 = rnn_ts2_nt(k.N)) {
      const bool ts3 = k.w16 && rnn_ts3_enabled();  // fp16 matrix pipe, error-compensated (rnn_ts3_body.h)
      if (ts3) d.k.dbg = diag_int("ts3_dbg", 0);  // diagnostics of rnn_ts3_body.h (wrong results on purpose); read on this path only
      if (epi == EPI_GRU && feat == FG && ts3) {
        dim3 g2(cdiv(n_mt, 2 * TS2_WAVES), cdiv(cdiv(k.N, 16), nt2));
        if (nt2 == 3) hipLaunchKernelGGL((rnn_ts3_kernel<EPI_GRU, FG, 2, 3>), g2, dim3(256), ts3_lds_bytes<3>(), s, d);
        else if (nt2 == 2) hipLaunchKernelGGL((rnn_ts3_kernel<EPI_GRU, FG, 2, 2>), g2, dim3(256), ts3_lds_bytes<2>(), s, d);
        else hipLaunchKernelGGL((rnn_ts3_kernel<EPI_GRU, FG, 2, 1>), g2, dim3(256), ts3_lds_bytes<1>(), s, d);
        done = true;
      } else if (epi == EPI_LINEAR && feat == FL && ts3) {
        dim3 g2(cdiv(n_mt, MB_TS2_FC3_MT * TS2_WAVES), cdiv(cdiv(k.N, 16), MB_TS2_FC3_NT));
        hipLaunchKernelGGL((rnn_ts3_kernel<EPI_LINEAR, FL, MB_TS2_FC3_MT, MB_TS2_FC3_NT>), g2, dim3(256), ts3_lds_bytes<MB_TS2_FC3_NT>(), s, d); done = true;
      } else
      if (epi == EPI_GRU && feat == FG) {  // rnn2: 128 row tiles -> 2 x nt2 tiles per wave
        dim3 g2(cdiv(n_mt, 2 * TS2_WAVES), cdiv(cdiv(k.N, 16), nt2));
        if (nt2 == 3) hipLaunchKernelGGL((rnn_ts2_kernel<EPI_GRU, FG, 2, 3>), g2, dim3(256), 0, s, d);
        else if (nt2 == 2) hipLaunchKernelGGL((rnn_ts2_kernel<EPI_GRU, FG, 2, 2>), g2, dim3(256), 0, s, d);
        else hipLaunchKernelGGL((rnn_ts2_kernel<EPI_GRU, FG, 2, 1>), g2, dim3(256), 0, s, d);
        done = true;
      } else if (epi == EPI_LINEAR && feat == FL) {  // fc3 + sampler: 32 row tiles only -> 2 x 1 tiles per wave at every width
        dim3 g2(cdiv(n_mt, MB_TS2_FC3_MT * TS2_WAVES), cdiv(cdiv(k.N, 16), MB_TS2_FC3_NT));
        hipLaunchKernelGGL((rnn_ts2_kernel<EPI_LINEAR, FL, MB_TS2_FC3_MT, MB_TS2_FC3_NT>), g2, dim3(256), 0, s, d); done = true;
      }
    }
    if (done) {}
    else if (epi == EPI_GRU && feat == FG) { hipLaunchKernelGGL((rnn_ts_kernel<EPI_GRU, FG>), gts, dim3(NW * 64), 0, s, d); done = true; }
    else if (epi == EPI_LINEAR && feat == FL) { hipLaunchKernelGGL((rnn_ts_kernel<EPI_LINEAR, FL>), gts, dim3(NW * 64), 0, s, d); done = true; }
  }
#define MB_TRY(EPI_, NT_, UB_, F_)                                                                         \
  if (!done && epi == (EPI_) && nt == (NT_) && ub == (UB_) && feat == (unsigned)(F_)) {                    \
    hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_, NT_, UB_, (unsigned)(F_)>), grid, dim3(NW * 64), 0, s, d); \
    done = true;                                                                                           \
  }
  if (!getenv("MBHIP_RNN_GENERIC")) { MB_RNN_INSTANCES(MB_TRY) }
#undef MB_TRY
#define MB_RNN(EPI_, NT_, UB_) hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_, NT_, UB_, RF_GENERIC>), grid, dim3(NW * 64), 0, s, d)
#define MB_RNN_E(EPI_)                                                                    \
  do {                                                                                    \
    if (nt == 2) { if (ub == 4) MB_RNN(EPI_, 2, 4); else MB_RNN(EPI_, 2, 2); }            \
    else if (ub == 8) MB_RNN(EPI_, 1, 8);                                                 \
    else if (ub == 4) MB_RNN(EPI_, 1, 4);                                                 \
    else MB_RNN(EPI_, 1, 2);                                                              \
  } while (0)
  if (!done) {
    if (epi == EPI_LINEAR) MB_RNN_E(EPI_LINEAR);
    else if (epi == EPI_GRU) MB_RNN_E(EPI_GRU);
    else MB_RNN_E(EPI_LSTM);
  }
#undef MB_RNN_E
#undef MB_RNN
  MB_HIP(hipGetLastError());
  return MB_OK;
}

}  // namespace mb
