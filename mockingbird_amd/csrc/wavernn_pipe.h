// WaveRNN sample loop for 2..32 fold columns (BASELINE configs[1]: 23 folds) as ONE persistent launch: the layers of the
// step are ROLES of resident workgroups with their weight tiles in LDS, and the folds -- independent sequences -- are cut
// into two column GROUPS that travel through the roles half a period apart, so that the hand-off latency of one group is
// covered by the arithmetic of the other.  (The 5-launch chain of wavernn_fast.h pays a kernel boundary + ramp per layer,
// 4.3 us x 5; two streams do not overlap on this chip, profiles/r01_wavernn_lane_sweep.json.  wavernn_persist.h is the
// few-column ancestor: every on-chain workgroup there recomputes the whole rnn1 finish and fetches all of P1 per column,
// which is what stops it at 4 columns.)
//
//   role  workgroups  LDS weights                                   per (step, group)
//   R1    64          W_hh1 GRU tiles 2b, 2b+1 (8 units)            keys(s-1) -> x | rnn1 finish for its 8 units (P1, h1, table rows
//                                                                   are lane-private) -> publishes x1, h1 | gathers h1, P1(s+1) = W_hh1.h1 + b
//   R2    64          W_ih2[x] and W_hh2 tiles 2b, 2b+1             gathers x1 | GRU on its 8 units (P2 lane-private) -> publishes x2, h2
//                                                                   | gathers h2, P2(s+1) = W_hh2.h2 + b
//   F1    32          fc1 tile                                      gathers x2 -> relu(fc1 . x2 + F1[frame]) -> publishes y1
//   F2    32          fc2 tile                                      gathers y1 -> publishes y2
//   F3    32          fc3 tile                                      gathers y2 -> logits + Gumbel-argmax over its 16 classes -> publishes keys
// (fatchord_version.py:190-228; split-hidden algebra of wavernn.hip's header).  Hidden halves never travel: the
// workgroup that owns a unit's W_hh rows also owns its state, its gate pre-activations and its elementwise update.
//
// Hand-offs are the 8-byte {value, step tag} granules of wavernn_persist.h (one relaxed agent-scope store / load each,
// the tag is the flag), dense [feature][column of the group], double-buffered by tag parity per group.  Why two
// parities suffice (tag t is overwritten by tag t+2): whoever writes tag t+2 of a vector has consumed keys(t+1), which
// needed every tile of y2(t+1) <- y1(t+1) <- x2(t+1) <- x1(t+1), i.e. every workgroup of every role has finished step
// t+1 and with it its reads of tag t (h1 / h2: P(t+1) is needed by the owner's own step t+1, so its gather of h(t) is
// over as well).  Every workgroup walks the (step, group) items in the same order and an item only depends on earlier
// items or on the same item at an earlier role: no cycle, no deadlock as long as the 224 workgroups are resident
// (checked on the host; every spin still has a wall-clock bail-out -> abort word -> the launch chain runs instead).
//
// Arithmetic: per tile the MFMA sequence and the wave-order reduction of fm_gemm<1, 4, 4, RL, 1>, epilogue expressions of
// wavernn_fast.h, the same Philox words: the sample stream is bit-identical to the chain's (tests/test_env_switches_gpu.py,
// and against the oracle in tests/test_wavernn_gpu.py).
#pragma once
#include "wavernn_persist.h"

namespace mb {

constexpr int WQ_R1 = 64, WQ_R2 = 64, WQ_F = 32;
constexpr int WQ_WGS = WQ_R1 + WQ_R2 + 3 * WQ_F;  // 224 resident workgroups, one per compute unit
constexpr int WQ_G = 2;                           // column groups in flight (this kernel: 2..32 columns)
constexpr int WQ_GMAX = 6;                        // wavernn_pipe16.h serves up to six groups (33..96 columns: a 3000-frame utterance is 68 folds)
constexpr int WQ_GC = 16;                         // columns per group (one MFMA column tile)
constexpr int WQ_DEFAULT_ON = 1;                  // default for 2..32 columns (MBHIP_WAVERNN_RESIDENT overrides)

// exchange area per (group, parity), in granules
enum { WQX_X1 = 0, WQX_X2 = 8192, WQX_H1 = 16384, WQX_H2 = 24576, WQX_Y1 = 32768, WQX_Y2 = 40960, WQX_KEY = 49152, WQX_PER = 50176 };
inline size_t wq_exchange_bytes() { return (size_t)WQ_GMAX * 2 * WQX_PER * 8 + 256 + 8192; }  // + abort word + diagnostics marks

struct WqK {
  const float* w_rnn2; const float* w_hh2; const float* w_hh1; const float* w_fc1; const float* w_fc2; const float* w_fc3;
  const float4* bhh1q; const float4* bhh2q; const float* b_fc3; const float* g1; const float* wI0;
  WfCond cond; const float* G2; const float* F1; const float* F2;
  WfGeom g;
  unsigned long long* ex; int* abort_word;
  float* samples; volatile int* progress;
  unsigned long long seed; int R, FC, C, S, N;
  int gn0[WQ_GMAX + 1];       // group g owns fold columns [gn0[g], gn0[g + 1]); wf_pipe_kernel looks at the first WQ_G groups only
  int mol, nr_mix;            // MOL mode (fatchord_version.py:213-220): fc3 has 3 nr_mix rows, F3 is ONE workgroup that samples the
                              // mixture of logistics itself (wf_fc3_mol_kernel's draws) and hands the SAMPLE to R1
  int flags;                  // A/B switches (MBHIP_DIAG=wq_flags=<bits>, default 17): 1 = exchange rows padded to 16 columns, 2 = R2's residual x1 by a global load,
                              // 16 = weight fragments held in registers for the whole utterance (LDS reads per product otherwise: +0.3 us per step)
  unsigned long long* trace;  // diagnostics (MBHIP_DIAG=wp_trace=<file>): wall-clock marks of one workgroup per role, steps 1000..1003
};

// wp_gather with the row stride LD apart from the live column count N
template <int SLEEP>
__device__ __forceinline__ bool wq_gather(const unsigned long long* vec, const unsigned tag, const int N, const int LD, float4 (&b)[4], int* abort_word,
                                          unsigned long long* mk = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
  // ONE watching lane, then a barrier, then one sweep (four staggered watchers + an LDS flag instead: 14.2 vs 13.9 us per step)
  wp_watch<SLEEP>(vec + (size_t)511 * LD + (N - 1), tag, abort_word);
  if (mk && threadIdx.x == 0) *mk = (unsigned long long)wall_clock64();  // diagnostics: the hand-off was noticed
#pragma unroll
  for (int p = 0; p < 4; ++p) b[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i >= N) return true;
  const unsigned long long* base = vec + ((size_t)(wave * 16 + kq * 4) * LD + i);
  unsigned long long v[16];
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) v[p * 4 + c] = wp_get(base + ((size_t)p * 128 + c) * LD);
    {  // all tags in one xor / or chain (a chain of && compiled to nested exec-mask branches, wavernn_pipe16.h)
      unsigned stale_ = 0u;
#pragma unroll
      for (int q = 0; q < 16; ++q) stale_ |= (unsigned)(v[q] >> 32) ^ tag;
      ok = ok && stale_ == 0u;
    }
    if (ok) break;
    if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p)
    b[p] = make_float4(__uint_as_float((unsigned)v[p * 4]), __uint_as_float((unsigned)v[p * 4 + 1]), __uint_as_float((unsigned)v[p * 4 + 2]),
                       __uint_as_float((unsigned)v[p * 4 + 3]));
  return true;
}

// wp_gemm / wp_gemm2 on A fragments the caller holds (registers for the whole utterance with flags & 16, re-read from LDS per
// product otherwise): the same MFMA sequence and wave-order reduction.
template <int RL>
__device__ __forceinline__ void wq_load_a(const float* lw, float4 (&a)[4]) {
  constexpr int BLK = 4 * RL * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;
  const float* wl = lw + ((u * RL + tau) * 4 + kq) * 4;
#pragma unroll
  for (int p = 0; p < 4; ++p) a[p] = *reinterpret_cast<const float4*>(wl + (size_t)(wave + 8 * p) * BLK);
}
__device__ __forceinline__ bool wq_gemm1(const float4 (&a)[4], const float4 (&b)[4], float* red, float (&sx)[4], unsigned long long* mk = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p].x, b[p].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p].y, b[p].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p].z, b[p].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p].w, b[p].w, acc, 0, 0, 0);
  }
  float4* red4 = reinterpret_cast<float4*>(red);
  red4[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (mk && threadIdx.x == 0) *mk = (unsigned long long)wall_clock64();
  if (wave != 0) return false;
#pragma unroll
  for (int g = 0; g < 4; ++g) sx[g] = 0.f;
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const float4 v = red4[w8 * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
  }
  return true;
}
__device__ __forceinline__ bool wq_gemm2(const float4 (&a0)[4], const float4 (&a1)[4], const float4 (&b)[4], float* red, float (&sx)[4],
                                         unsigned long long* mk = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[p].x, b[p].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[p].x, b[p].x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[p].y, b[p].y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[p].y, b[p].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[p].z, b[p].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[p].z, b[p].z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[p].w, b[p].w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[p].w, b[p].w, acc1, 0, 0, 0);
  }
  float4* red4 = reinterpret_cast<float4*>(red);  // [2 tiles][8 waves][64]
  red4[wave * 64 + lane] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
  red4[512 + wave * 64 + lane] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
  __syncthreads();
  if (mk && threadIdx.x == 0) *mk = (unsigned long long)wall_clock64();
  if (wave >= 2) return false;
#pragma unroll
  for (int g = 0; g < 4; ++g) sx[g] = 0.f;
#pragma unroll
  for (int w8 = 0; w8 < 8; ++w8) {
    const float4 v = red4[wave * 512 + w8 * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
  }
  return true;
}

// dynamic LDS (floats): [weights: up to 4 GRU tiles] [red: two 4096-float buffers, alternating] [keys / samples]
constexpr int WQ_LDS_W = 4 * 6144, WQ_LDS_RED = 2 * 4096;
constexpr size_t WQ_LDS_BYTES = (size_t)(WQ_LDS_W + WQ_LDS_RED) * 4 + 2 * WQ_GC * 8 + 8 * 16 * 4 + 64;

__global__ __launch_bounds__(512) void wf_pipe_kernel(WqK a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lw = lds;
  float* red = lds + WQ_LDS_W;
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(red + WQ_LDS_RED);  // [GC] max key of the step
  float* s_x = reinterpret_cast<float*>(s_key + WQ_GC);                                 // [GC] decoded sample
  if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (tests: the fallback path)
  const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int du = lane >> 4, i = lane & 15;
  const int H = a.R, S = a.S;
  const int n_t3 = a.mol ? 1 : a.C / 16;  // fc3 workgroups (MOL: one, holding both row tiles of the <= 32 mixture parameters)
  int rb = 0;  // red buffer of the next GEMM
  auto EX = [&](int what, int g, unsigned tag) { return a.ex + ((size_t)g * 2 + (tag & 1)) * WQX_PER + what; };
#define WQ_MARK(role, k)                                                                                   \
  do {                                                                                                     \
    if (a.trace && tid == 0 && mark_wg && g == 0 && s >= 1000 && s < 1004)                                 \
      a.trace[((role) * 4 + (s - 1000)) * 16 + (k)] = (unsigned long long)wall_clock64();                  \
  } while (0)

#define WQ_MK(role, k) ((a.trace && mark_wg && g == 0 && s >= 1000 && s < 1004) ? a.trace + (((role) * 4 + (s - 1000)) * 16 + (k)) : nullptr)
  const bool aregs = (a.flags & 16) != 0;   // weight fragments live in registers for the whole utterance

  if (blk < WQ_R1) {
    // ---------------------------------------------------------------------------------------------- R1: rnn1
    const bool mark_wg = blk == 0;
    wp_copy_tile(lw, a.w_hh1 + (size_t)(2 * blk) * 6144, 12288);
    if (tid < WQ_GC) { s_key[tid] = 0ull; s_x[tid] = 0.f; }
    const int ju = (2 * blk + (wave & 1)) * 4 + du;  // unit of an epilogue lane (waves 0 / 1)
    const float4 bq = a.bhh1q[ju];
    const float gr = a.g1[ju], gz = a.g1[H + ju], gn = a.g1[2 * H + ju], w0 = a.wI0[ju];
    float h1[WQ_G], P1[WQ_G][3], tq[WQ_G][4];
#pragma unroll
    for (int g = 0; g < WQ_G; ++g) {
      h1[g] = 0.f; P1[g][0] = bq.x; P1[g][1] = bq.y; P1[g][2] = bq.z;  // W_hh . 0 + b_hh
      const int Ng = a.gn0[g + 1] - a.gn0[g];
      const int ncl = a.gn0[g] + (i < Ng ? i : (Ng > 0 ? Ng - 1 : 0));
      const float4 t4 = wf_cond_row4(a.cond, wf_pos(a.g, ncl, 0), (unsigned)a.g.total_len, ju, H, a.g.frames);
      tq[g][0] = t4.x; tq[g][1] = t4.y; tq[g][2] = t4.z; tq[g][3] = t4.w;
    }
    __syncthreads();
    float4 A0[4], A1[4];
    if (aregs) { wq_load_a<3>(lw, A0); wq_load_a<3>(lw + 6144, A1); }
    for (int s = 0; s <= S; ++s) {
      const unsigned tag_prev = (unsigned)s, tag = (unsigned)s + 1;
#pragma unroll
      for (int g = 0; g < WQ_G; ++g) {
        const int n0 = a.gn0[g], Ng = a.gn0[g + 1] - n0, LD = (a.flags & 1) ? WQ_GC : Ng;
        if (Ng <= 0) continue;
        WQ_MARK(0, 0);
        // ---- keys of step s-1 -> sample x of every column of the group ----
        if (s > 0 && a.mol) {  // MOL: the key word IS the sample (one granule per column, written by the one F3 workgroup)
          if (tid < Ng) {
            unsigned xv[1];
            if (!wp_wait<1>(EX(WQX_KEY, g, tag_prev) + tid, 1, tag_prev, xv, a.abort_word)) return;
            const float x = __uint_as_float(xv[0]);
            s_x[tid] = x;
            if (blk == 0) {
              a.samples[(size_t)(n0 + tid) * S + (s - 1)] = x;
              if (a.progress && n0 + tid == 0 && (s - 1) % 100 == 0) *a.progress = s;
            }
          }
          __syncthreads();
          WQ_MARK(0, 1);
        } else if (s > 0) {
          const unsigned long long* K = EX(WQX_KEY, g, tag_prev);  // key halves of a tile: rows of LD granules (whole lines per store)
          wp_watch<1>(K + (size_t)((n_t3 - 1) * 2 + 1) * LD + (Ng - 1), tag_prev, a.abort_word);
          if (tid < 32 * Ng && (tid & 31) < n_t3) {
            const int tile = tid & 31, n = tid >> 5;
            unsigned kv[2];
            if (!wp_wait<2>(K + (size_t)tile * 2 * LD + n, LD, tag_prev, kv, a.abort_word)) return;
            atomicMax(&s_key[n], ((unsigned long long)kv[0] << 32) | (unsigned long long)kv[1]);
          }
          __syncthreads();
          WQ_MARK(0, 1);
          if (tid < Ng) {
            const unsigned long long slot = s_key[tid];
            const float x = slot ? 2.f * (float)argmax_class(slot) / ((float)a.C - 1.f) - 1.f : 0.f;
            s_x[tid] = x;
            s_key[tid] = 0ull;
            if (blk == 0) {
              a.samples[(size_t)(n0 + tid) * S + (s - 1)] = x;
              if (a.progress && n0 + tid == 0 && (s - 1) % 100 == 0) *a.progress = s;
            }
          }
          __syncthreads();
        }
        if (s == S) continue;
        // ---- rnn1 finish for (unit ju, column i): wf_finish_kernel's expressions ----
        if (wave < 2 && i < Ng) {
          const float x = s_x[i];
          const float rg = sigmoidf_((tq[g][0] + x * gr) + P1[g][0]);
          const float zg = sigmoidf_((tq[g][1] + x * gz) + P1[g][1]);
          const float ng = tanhf((tq[g][2] + x * gn) + rg * P1[g][2]);
          const float hy = ng + zg * (h1[g] - ng);
          h1[g] = hy;
          wp_put(EX(WQX_X1, g, tag) + (size_t)ju * LD + i, (tq[g][3] + x * w0) + hy, tag);
          wp_put(EX(WQX_H1, g, tag) + (size_t)ju * LD + i, hy, tag);
        }
        WQ_MARK(0, 2);
        if (s + 1 >= S) continue;
        // ---- next step's table rows (a whole step to arrive) ----
        if (wave < 2) {
          const float4 t4 = wf_cond_row4(a.cond, wf_pos(a.g, n0 + (i < Ng ? i : Ng - 1), s + 1), (unsigned)a.g.total_len, ju, H, a.g.frames);
          tq[g][0] = t4.x; tq[g][1] = t4.y; tq[g][2] = t4.z; tq[g][3] = t4.w;
        }
        // ---- hidden half of the next step: P1 = W_hh1 . h1 + b_hh1, kept by the lane that will use it ----
        float4 b[4];
        if (!wq_gather<2>(EX(WQX_H1, g, tag), tag, Ng, LD, b, a.abort_word, WQ_MK(0, 6))) return;
        WQ_MARK(0, 3);
        float sx[4];
        if (!aregs) { wq_load_a<3>(lw, A0); wq_load_a<3>(lw + 6144, A1); }
        const bool epi = wq_gemm2(A0, A1, b, red + rb * 4096, sx);
        rb ^= 1;
        if (epi) { P1[g][0] = sx[0] + bq.x; P1[g][1] = sx[1] + bq.y; P1[g][2] = sx[2] + bq.z; }
        WQ_MARK(0, 4);
      }
    }
    return;
  }

  if (blk < WQ_R1 + WQ_R2) {
    // ---------------------------------------------------------------------------------------------- R2: rnn2
    const int b2 = blk - WQ_R1;
    const bool mark_wg = b2 == 0;
    wp_copy_tile(lw, a.w_rnn2 + (size_t)(2 * b2) * 6144, 12288);
    wp_copy_tile(lw + 12288, a.w_hh2 + (size_t)(2 * b2) * 6144, 12288);
    const int ju = (2 * b2 + (wave & 1)) * 4 + du;
    const int xr_wave = (b2 >> 1) & 7, xr_p = b2 >> 4, xr_kq0 = (b2 & 1) * 2;  // where units 8 b2 .. 8 b2 + 7 sit in the B fragments
    float* s_xr = s_x + WQ_GC;  // [2 tiles][4 units][16 columns]
    const float4 bq = a.bhh2q[ju];
    float h2[WQ_G], P2[WQ_G][3], g2v[WQ_G][3];
    int g2_row[WQ_G];
#pragma unroll
    for (int g = 0; g < WQ_G; ++g) { h2[g] = 0.f; P2[g][0] = bq.x; P2[g][1] = bq.y; P2[g][2] = bq.z; g2_row[g] = -1; g2v[g][0] = g2v[g][1] = g2v[g][2] = 0.f; }
    __syncthreads();
    float4 A0[4], A1[4], A2[4], A3[4];
    if (aregs) { wq_load_a<3>(lw, A0); wq_load_a<3>(lw + 6144, A1); wq_load_a<3>(lw + 12288, A2); wq_load_a<3>(lw + 12288 + 6144, A3); }
    for (int s = 0; s < S; ++s) {
      const unsigned tag = (unsigned)s + 1;
#pragma unroll
      for (int g = 0; g < WQ_G; ++g) {
        const int n0 = a.gn0[g], Ng = a.gn0[g + 1] - n0, LD = (a.flags & 1) ? WQ_GC : Ng;
        if (Ng <= 0) continue;
        const int frow = wf_frame_row(a.g, n0 + (i < Ng ? i : Ng - 1), s);
        if (wave < 2 && frow != g2_row[g]) {  // the per-frame rows change once per hop: kept in registers in between
          const float* gp = a.G2 + (size_t)frow * 3 * H + ju;
          g2v[g][0] = gp[0]; g2v[g][1] = gp[H]; g2v[g][2] = gp[2 * H];
          g2_row[g] = frow;
        }
        WQ_MARK(1, 0);
        float4 b[4];
        if (!wq_gather<1>(EX(WQX_X1, g, tag), tag, Ng, LD, b, a.abort_word, WQ_MK(1, 6))) return;
        WQ_MARK(1, 1);
        // the residual x1 of this workgroup's own 8 units sits in the fragments just gathered (one k-block, wave xr_wave,
        // register xr_p, lanes kq = xr_kq0 + tile): handed to the epilogue lanes through LDS behind the GEMM's own barrier
        if (wave == xr_wave && (lane >> 4) >= xr_kq0 && (lane >> 4) < xr_kq0 + 2) {
          float* dst = s_xr + (((lane >> 4) - xr_kq0) * 4) * 16 + i;
#pragma unroll
          for (int p = 0; p < 4; ++p)  // (static register index: a run-time b[xr_p] would put the fragments in scratch)
            if (p == xr_p) { dst[0] = b[p].x; dst[16] = b[p].y; dst[32] = b[p].z; dst[48] = b[p].w; }
        }
        float sx[4];
        if (!aregs) { wq_load_a<3>(lw, A0); wq_load_a<3>(lw + 6144, A1); }
        const bool epi = wq_gemm2(A0, A1, b, red + rb * 4096, sx, WQ_MK(1, 7));
        rb ^= 1;
        WQ_MARK(1, 5);
        if (epi && i < Ng) {
          float xr = s_xr[((wave & 1) * 4 + du) * 16 + i];
          if (a.flags & 2) {
            unsigned xu[1];
            if (!wp_wait<1>(EX(WQX_X1, g, tag) + (size_t)ju * LD + i, 1, tag, xu, a.abort_word)) return;
            xr = __uint_as_float(xu[0]);
          }
          const float rg = sigmoidf_((sx[0] + g2v[g][0]) + P2[g][0]);
          const float zg = sigmoidf_((sx[1] + g2v[g][1]) + P2[g][1]);
          const float ng = tanhf((sx[2] + g2v[g][2]) + rg * P2[g][2]);
          const float hy = ng + zg * (h2[g] - ng);
          h2[g] = hy;
          wp_put(EX(WQX_X2, g, tag) + (size_t)ju * LD + i, xr + hy, tag);
          wp_put(EX(WQX_H2, g, tag) + (size_t)ju * LD + i, hy, tag);
        }
        WQ_MARK(1, 2);
        if (s + 1 >= S) continue;
        if (!wq_gather<2>(EX(WQX_H2, g, tag), tag, Ng, LD, b, a.abort_word)) return;
        WQ_MARK(1, 3);
        if (!aregs) { wq_load_a<3>(lw + 12288, A2); wq_load_a<3>(lw + 12288 + 6144, A3); }
        const bool epi2 = wq_gemm2(A2, A3, b, red + rb * 4096, sx);
        rb ^= 1;
        if (epi2) { P2[g][0] = sx[0] + bq.x; P2[g][1] = sx[1] + bq.y; P2[g][2] = sx[2] + bq.z; }
        WQ_MARK(1, 4);
      }
    }
    return;
  }

  // ------------------------------------------------------------------------------------------------ F1 / F2 / F3
  const int fr = (blk - WQ_R1 - WQ_R2) / WQ_F, ft = (blk - WQ_R1 - WQ_R2) % WQ_F;  // role 0 / 1 / 2, row tile
  const bool mark_wg = ft == 0;
  if (fr == 2 && ft >= n_t3) return;
  const bool f3mol = fr == 2 && a.mol;
  wp_copy_tile(lw, (fr == 0 ? a.w_fc1 : fr == 1 ? a.w_fc2 : a.w_fc3) + (size_t)ft * 8192, f3mol ? 16384 : 8192);
  const float4 b3q = fr == 2 && !a.mol ? *reinterpret_cast<const float4*>(a.b_fc3 + ft * 16 + du * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float bmol[4] = {0.f, 0.f, 0.f, 0.f};  // MOL: bias of rows wave * 16 + du * 4 + r (waves 0 / 1 = the two row tiles)
  if (f3mol && wave < 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + du * 4 + r;
      bmol[r] = row < a.C ? a.b_fc3[row] : 0.f;
    }
  }
  float4 fpre[WQ_G];
  int f_row[WQ_G];
#pragma unroll
  for (int g = 0; g < WQ_G; ++g) { fpre[g] = make_float4(0.f, 0.f, 0.f, 0.f); f_row[g] = -1; }
  const int src = fr == 0 ? WQX_X2 : fr == 1 ? WQX_Y1 : WQX_Y2;
  __syncthreads();
  float4 A0[4];
  if (aregs && !f3mol) wq_load_a<4>(lw, A0);
  for (int s = 0; s < S; ++s) {
    const unsigned tag = (unsigned)s + 1;
#pragma unroll
    for (int g = 0; g < WQ_G; ++g) {
      const int n0 = a.gn0[g], Ng = a.gn0[g + 1] - n0, LD = (a.flags & 1) ? WQ_GC : Ng;
      if (Ng <= 0) continue;
      const int ncl = n0 + (i < Ng ? i : Ng - 1);
      float lgn[4] = {0.f, 0.f, 0.f, 0.f};
      if (fr < 2) {
        const int frow = wf_frame_row(a.g, ncl, s);
        if (wave == 0 && frow != f_row[g]) {
          fpre[g] = *reinterpret_cast<const float4*>((fr == 0 ? a.F1 : a.F2) + (size_t)frow * a.FC + ft * 16 + du * 4);
          f_row[g] = frow;
        }
      } else if (wave == 0 && !a.mol) {  // the step's Gumbel noise does not depend on the data: drawn before the wait
        uint32_t grn[4];
        philox4x32((uint32_t)s, (uint32_t)ncl, (uint32_t)((ft * 16 + du * 4) >> 2), 0x57415645u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), grn);
#pragma unroll
        for (int r = 0; r < 4; ++r) lgn[r] = logf(-logf(u32_to_unit(grn[r])));
      }
      WQ_MARK(2 + fr, 0);
      float4 b[4];
      if (!wq_gather<1>(EX(src, g, tag), tag, Ng, LD, b, a.abort_word, WQ_MK(2 + fr, 6))) return;
      WQ_MARK(2 + fr, 1);
      float sx[4];
      if (f3mol) {
        // ---- MOL: both row tiles against the gathered y2 (the per-tile sums of fm_gemm), the mixture parameters of the group's
        //      columns through LDS (the red half the NEXT product will use: free until the barrier of its gather), then
        //      wf_fc3_mol_kernel's sampler per column -- same Philox words, same expressions -- and the sample as the key ----
        const bool epi2 = wp_gemm2<4>(lw, 8192, b, red + rb * 4096, sx);
        rb ^= 1;
        float* lg = red + rb * 4096;  // [16 columns][33]
        if (epi2) {
#pragma unroll
          for (int r = 0; r < 4; ++r) lg[i * 33 + wave * 16 + du * 4 + r] = sx[r] + bmol[r];
        }
        __syncthreads();
        WQ_MARK(2 + fr, 3);
        if (tid < Ng) {
          const float* l = lg + tid * 33;
          const int M = a.nr_mix, n = n0 + tid;
          float best = -INFINITY, uu = 0.5f;
          int bidx = 0;
          for (int q = 0; q <= M / 4; ++q) {  // draws 0 .. M: M mixture-indicator uniforms, then the logistic one
            uint32_t rr[4];
            philox4x32((uint32_t)s, (uint32_t)n, (uint32_t)q, 0x4d4f4c21u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), rr);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int m = q * 4 + e;
              const float u = 1e-5f + (1.0f - 2e-5f) * u32_to_unit(rr[e]);  // uniform_(1e-5, 1 - 1e-5)
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
          wp_put(EX(WQX_KEY, g, tag) + tid, x, tag);
        }
        WQ_MARK(2 + fr, 2);
        continue;
      }
      if (!aregs) wq_load_a<4>(lw, A0);
      const bool epi = wq_gemm1(A0, b, red + rb * 4096, sx, WQ_MK(2 + fr, 7));
      rb ^= 1;
      WQ_MARK(2 + fr, 3);
      if (!epi) continue;
      if (fr < 2) {
        if (i < Ng) {
          unsigned long long* Y = EX(fr == 0 ? WQX_Y1 : WQX_Y2, g, tag) + (size_t)(ft * 16 + du * 4) * LD + i;
          wp_put(Y, fmaxf(sx[0] + fpre[g].x, 0.f), tag);
          wp_put(Y + LD, fmaxf(sx[1] + fpre[g].y, 0.f), tag);
          wp_put(Y + 2 * LD, fmaxf(sx[2] + fpre[g].z, 0.f), tag);
          wp_put(Y + 3 * LD, fmaxf(sx[3] + fpre[g].w, 0.f), tag);
        }
      } else {  // wf_fc3_kernel's sampler; lanes of dead columns take part in the shuffles only
        const float bv[4] = {b3q.x, b3q.y, b3q.z, b3q.w};
        float best = -INFINITY;
        int bcls = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = ft * 16 + du * 4 + r;
          const float v = sx[r] + bv[r];
          const float gmb = v - lgn[r];
          if (gmb > best) { best = gmb; bcls = row; }
        }
        unsigned long long pk = pack_argmax(best, bcls);
        const unsigned long long o1 = __shfl_xor(pk, 16, 64);
        pk = o1 > pk ? o1 : pk;
        const unsigned long long o2 = __shfl_xor(pk, 32, 64);
        pk = o2 > pk ? o2 : pk;
        if (du == 0 && i < Ng) {
          unsigned long long* K = EX(WQX_KEY, g, tag) + (size_t)ft * 2 * LD + i;
          wp_put_u(K, (unsigned)(pk >> 32), tag);
          wp_put_u(K + LD, (unsigned)pk, tag);
        }
      }
      WQ_MARK(2 + fr, 2);
    }
  }
#undef WQ_MARK
#undef WQ_MK
}

}  // namespace mb
