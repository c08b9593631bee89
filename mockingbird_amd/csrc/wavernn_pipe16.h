// Round 4 / 5: the resident pipelined WaveRNN kernel of wavernn_pipe.h with its exchange vectors and its products on 22-bit
// operand pairs -- the benchmarked path for 2..96 fold columns (BASELINE configs[1]: 23 folds of a RAW-mode model; MOL-mode models run it too).
//
// What round 3's wall-clock marks said about wf_pipe_kernel (profiles/r04_wavernn_pipe_marks.json, 14.3 us per step on that box):
// an edge is NOT latency -- a producer's store is noticed by the watching lane 0.3 us later -- it is the SWEEP: every consumer
// workgroup pulls 512 features x 16 columns x 8-byte granules = 64 KB through its compute unit's 64 B/clk path to the L2
// (1.1 us until the first wave has its fragments, 1.4 us until the slowest), five times per step; and a stage is 0.43-0.85 us
// of v_mfma_f32_16x16x4_f32 (one tile per unit at the 1/16-rate fp32 pipe).  Levers that do not touch those two were measured
// and lost (wq_flags sweep of the same session: whole-line stores +0.0, 16-byte sweep loads +1.7, two / four staggered
// polls of the watching lane +0.5 / +0.7, keys read by the finish lanes +3.4 us per step; weight fragments in registers -0.3).
// So this kernel halves the bytes of every sweep and takes the products off the fp32 pipe:
//   * a value crosses workgroups as fp16 hi + fp16 lo, x = xh + 2^-11 xl with xl = fp16((x - xh) 2^11) -- conv1d.hip's scaled
//     residual (round 5; round 4 stored the residual unscaled: an fp16 subnormal below |x| = 0.125, i.e. 14 bits at |x| = 1e-3,
//     which is what relu(fc1 ..) / relu(fc2 ..) of a real checkpoint may well be): each operand keeps max(2^-21 |x|, 2^-35);
//   * an 8-byte granule carries TWO features as {hi pair | lo pair}: word 0 = (xh, xh') IS one register of the hi B fragment,
//     word 1 = (xl, xl') one register of the lo fragment -- no byte permutes in the sweep (round 4 interleaved hi / lo per feature:
//     16 v_perm + 8 v_and per lane and sweep), and the publishing lane needs two packed conversions.  The step tag is ONE bit,
//     the least significant bit of the first low half (a granule is rewritten every second step and read in between:
//     tbit(t) = ((t + 1) >> 1) & 1 alternates per parity buffer and is 1 for the first write, memory starts as 0).  The consumer
//     leaves it in place: whatever it is, the low half is off by at most one unit of ITS last place = 2^-21 |x|, which is what
//     clearing the bit costs as well;
//   * the weights are split on the host, w 2^s = wh + wl (s per matrix: max |w| 2^s in [2^13, 2^14)), into the A fragments of
//     v_mfma_f32_16x16x32_f16 and live in REGISTERS for the whole utterance (16 VGPRs per tile, no LDS tile at all); a product is
//     (wl.xh + wh.xh) + 2^-11 (wh.xl) in two fp32 accumulators -- 6 MFMAs of 16 cycles per wave and tile instead of 16 of 32.
// Range: |x| <= 65504 (fp16's largest value).  x1 = I(..) + h1, x2 = x1 + h2 and the two relu outputs are not bounded by the
// architecture; a publishing lane that sees a value beyond the range (or a NaN) raises the RANGE word, the host discards the launch's
// samples and the exact fp32 launch chain computes the utterance (mb_wavernn_last_path reports it) -- no silent clamp.
// The sample stream is NOT bit-identical to the launch chain's / wf_pipe_kernel's (VERDICT r03 item 3 allows that); it is held to the
// oracle itself: tests/test_wavernn_gpu.py::test_production_* replay the reference loop body on the device's own history with the
// exported noise, since round 5 over ALL steps of the benchmarked call.  MBHIP_WAVERNN_RESIDENT=exact selects the exact kernel
// (wavernn_pipe.h, the A/B partner).
// Roles, item order, deadlock argument, bail-outs: wavernn_pipe.h's, unchanged.  Column groups: two for 2..32 columns as there; up to
// SIX for 33..96 columns (fold_with_overlap has no limit, fatchord_version.py:288-338: a 3000-mel-frame utterance -- BASELINE
// configs[4]'s length -- is 68 folds) -- a workgroup serves the groups in turn, the period stays the trip of ONE group while its
// items fit.  The group count is a template parameter (2 / 4 / 6): per-group state lives in registers, and the six-group form of
// the loop spilled 117 scalar registers in the two-group case that the headline runs.
#pragma once
#include "wavernn_pipe.h"
#include "pair_granule.h"

namespace mb {

struct Wq16K {
  WqK q;                                      // everything wf_pipe_kernel takes (its fp32 weight images are unused here)
  const uint4* h_rnn2; const uint4* h_hh2; const uint4* h_hh1; const uint4* h_fc1; const uint4* h_fc2; const uint4* h_fc3;  // split images
  float us_rnn2, us_hh2, us_hh1, us_fc1, us_fc2, us_fc3;  // 2^-s of each matrix
  int* range_word;                            // raised by a publishing lane that sees |x| > 65504 or a NaN
};

// Host: tile-ordered rows (pack_rowtile's input: rows of tile mt are [mt * 4 RL, (mt + 1) * 4 RL) in (unit, gate) order, K = 512)
// -> per tile [wave 8][k-step 2][hi | lo][lane 64][8 halves]: lane (m = lane & 15, kb = lane >> 4) holds row (m >> 2) * RL +
// min(m & 3, RL - 1) of the tile (the dead 4th GRU row re-reads gate 2, as the fp32 image does), k = wave * 64 + step * 32 + kb * 8 + e.
inline int wq16_scale_exp(const float* rows, size_t n) {
  float wmax = 0.f;
  for (size_t i = 0; i < n; ++i) wmax = std::max(wmax, std::fabs(rows[i]));
  if (!(wmax > 0.f) || !std::isfinite(wmax)) return 0;
  int e2;
  (void)std::frexp(wmax, &e2);
  return std::max(-24, std::min(40, 14 - e2));
}
// min_tiles: images a kernel reads a fixed number of tiles of (MOL fc3: two) are padded with zero tiles
inline void wq16_pack(const float* rows, int n_live_rows, int K, int RL, int sexp, std::vector<unsigned short>* out, int min_tiles = 0) {
  const int per_tile = 4 * RL, n_mt = std::max((n_live_rows + per_tile - 1) / per_tile, min_tiles);
  const float scale = std::ldexp(1.f, sexp);
  out->assign((size_t)n_mt * 8 * 2 * 2 * 64 * 8, 0);
  for (int mt = 0; mt < n_mt; ++mt)
    for (int w = 0; w < 8; ++w)
      for (int st = 0; st < 2; ++st)
        for (int lane = 0; lane < 64; ++lane) {
          const int m = lane & 15, kb = lane >> 4;
          const int row = mt * per_tile + (m >> 2) * RL + std::min(m & 3, RL - 1);
          for (int e = 0; e < 8; ++e) {
            const int k = w * 64 + st * 32 + kb * 8 + e;
            float v = 0.f;
            if (row < n_live_rows && k < K) v = rows[(size_t)row * K + k] * scale;
            const wh16 hi = (wh16)v;
            const wh16 lo = (wh16)(v - (float)hi);
            const size_t base = ((((size_t)mt * 8 + w) * 2 + st) * 2) * 64 * 8;
            unsigned short hb, lb;
            memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
            (*out)[base + (size_t)lane * 8 + e] = hb;
            (*out)[base + 64 * 8 + (size_t)lane * 8 + e] = lb;
          }
        }
}

#ifndef WQ16_RETRY_SLEEP
#define WQ16_RETRY_SLEEP 1  // s_sleep between two sweeps of a lane that found a stale granule (A/B)
#endif
// B fragments (hi and lo, two k-steps of 32) of this lane from an exchange vector [feature pair 256][16 columns]: lane (column
// i, kb) of wave w needs features w * 64 + st * 32 + kb * 8 + 0..7 = pairs w * 32 + st * 16 + kb * 4 + 0..3 -- eight 8-byte
// loads per lane (wf_pipe_kernel: sixteen), 32 KB per workgroup and sweep; word 0 / word 1 of granule j ARE register j of the
// hi / lo fragment.  One watching lane, a barrier, one sweep.  lane_off = (wave * 32 + kb * 4) * WQ_GC + min(i, N - 1), in granules:
// the lanes of dead columns (i >= N) read the last live column again -- the same cache lines, no zero-fill, no divergence; what
// they compute is a copy of that column and is never published.
template <int SLEEP>
__device__ __forceinline__ bool wq16_gather(const unsigned long long* vec, const unsigned lane_off, const unsigned tag, const int N, wh16x8 (&bh)[2], wh16x8 (&bl)[2],
                                            int* abort_word, unsigned long long* mk = nullptr) {
  const unsigned tb = wq16_tbit(tag);
  // SLEEP == 2 marks the hidden-state sweeps of R1 / R2 (h1, h2: published about a sweep's round trip before the workgroup asks for
  // them, and off the chain): no watching lane there, every lane polls its own granules -- the watch was a second round trip for
  // data that is almost always in place, and it kept the two busiest roles away from the chain's items (8.75 -> 8.53 us per step)
  if (threadIdx.x == 0 && SLEEP != 2) {
    const unsigned long long* p = vec + (size_t)255 * WQ_GC + (N - 1);
    unsigned long long t0 = 0;
    for (int tries = 0; !wq16_fresh(wp_get(p), tb); ++tries) {
      if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) break;
      __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
  __syncthreads();
  if (mk && threadIdx.x == 0) *mk = (unsigned long long)wall_clock64();
  const unsigned long long* base = vec + lane_off;
  unsigned long long v[8];
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[st * 4 + j] = wp_get(base + (size_t)(st * 16 + j) * WQ_GC);
    // all eight tags in one chain of three-input bit operations: stale |= word1 ^ expected (truth table 0xf6 = a | (b ^ c))
    unsigned stale = 0u;
#pragma unroll
    for (int q = 0; q < 8; ++q) stale = __builtin_amdgcn_bitop3_b32(stale, (unsigned)(v[q] >> 32), tb, 0xf6);
    if ((stale & 1u) == 0u) break;
    if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) return false;
    __builtin_amdgcn_s_sleep(WQ16_RETRY_SLEEP);
  }
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    wq_u4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = (unsigned)v[st * 4 + j]; l[j] = (unsigned)(v[st * 4 + j] >> 32); }
    bh[st] = __builtin_bit_cast(wh16x8, h);
    bl[st] = __builtin_bit_cast(wh16x8, l);
  }
  return true;
}

// A fragments of one tile for this lane: [k-step][hi | lo]
struct Wq16A { wh16x8 h[2], l[2]; };
__device__ __forceinline__ void wq16_load_a(const uint4* img, const int tile, Wq16A& A) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint4* p = img + ((size_t)(tile * 8 + wave) * 4) * 64 + lane;
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    A.h[st] = __builtin_bit_cast(wh16x8, p[(st * 2) * 64]);
    A.l[st] = __builtin_bit_cast(wh16x8, p[(st * 2 + 1) * 64]);
  }
}
// (wl.xh + wh.xh) + 2^-11 (wh.xl'): two independent accumulator chains
__device__ __forceinline__ f32x4 wq16_mma(const Wq16A& A, const wh16x8 (&bh)[2], const wh16x8 (&bl)[2]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.l[st], bh[st], acc, 0, 0, 0);
    acl = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.h[st], bl[st], acl, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.h[st], bh[st], acc, 0, 0, 0);
  }
  return acc + acl * WQ16_LO_UNSCALE;
}
// one tile: the 8 waves' K slices through LDS, wave 0 gets the sums (x 2^-s)
__device__ __forceinline__ bool wq16_gemm1(const Wq16A& A, const wh16x8 (&bh)[2], const wh16x8 (&bl)[2], float* red, const float unscale, float (&sx)[4],
                                           unsigned long long* mk = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f32x4 acc = wq16_mma(A, bh, bl);
  float4* red4 = reinterpret_cast<float4*>(red);
  red4[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (mk && threadIdx.x == 0) *mk = (unsigned long long)wall_clock64();
  if (wave != 0) return false;
  {
    const float4 v = red4[lane];
    sx[0] = v.x; sx[1] = v.y; sx[2] = v.z; sx[3] = v.w;
  }
#pragma unroll
  for (int w8 = 1; w8 < 8; ++w8) {
    const float4 v = red4[w8 * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) sx[g] *= unscale;
  return true;
}
// two tiles against the same B fragments, one barrier; wave tt (0 / 1) gets tile tt's sums
__device__ __forceinline__ bool wq16_gemm2(const Wq16A& A0, const Wq16A& A1, const wh16x8 (&bh)[2], const wh16x8 (&bl)[2], float* red, const float unscale,
                                           float (&sx)[4], unsigned long long* mk = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const f32x4 acc0 = wq16_mma(A0, bh, bl), acc1 = wq16_mma(A1, bh, bl);
  float4* red4 = reinterpret_cast<float4*>(red);  // [2 tiles][8 waves][64]
  red4[wave * 64 + lane] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
  red4[512 + wave * 64 + lane] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
  __syncthreads();
  if (mk && threadIdx.x == 0) *mk = (unsigned long long)wall_clock64();
  if (wave >= 2) return false;
  {
    const float4 v = red4[wave * 512 + lane];
    sx[0] = v.x; sx[1] = v.y; sx[2] = v.z; sx[3] = v.w;
  }
#pragma unroll
  for (int w8 = 1; w8 < 8; ++w8) {
    const float4 v = red4[wave * 512 + w8 * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) sx[g] *= unscale;
  return true;
}

// conditioning-sequence position of a lane's fold column, advanced one step at a time (wf_pos / wf_frame_row divide by the hop
// at every item: ~ 25 instructions per lane, group and role)
struct WqPos {
  unsigned pos, f, p;
  __device__ __forceinline__ void init(const unsigned pos0, const unsigned hop) { pos = pos0; f = pos0 / hop; p = pos0 - f * hop; }
  __device__ __forceinline__ void step(const unsigned hop) { ++pos; if (++p == hop) { p = 0u; ++f; } }
  __device__ __forceinline__ int frame_row(const unsigned total_len, const int frames) const { return pos < total_len ? (int)f : frames; }  // = wf_frame_row
};

// LDS (floats): [red: two 4096-float buffers, alternating] [keys / samples / residual hand-over]
constexpr size_t WQ16_LDS_BYTES = (size_t)WQ_LDS_RED * 4 + WQ_GMAX * WQ_GC * 8 + WQ_GMAX * WQ_GC * 4 + 8 * 16 * 4 + 64;

// MOL: the sampler mode is a compile-time property (a run-time `a.mol` next to the RAW path cost the headline 0.27 us per step: one
// more fragment set live in the F roles, a branch per item).  NG: column groups the instance serves (its per-group state is
// register arrays of that size; groups the launch does not use have Ng = 0).  TRACE: the diagnostics marks (MBHIP_DIAG=wp_trace=<file>)
// exist in the <false, 2, true> instance only -- each mark is an exec-mask branch on the critical path of every item otherwise.
#ifndef WQ16_CRIT_SLEEP
#define WQ16_CRIT_SLEEP 1   // s_sleep between the polls of the watching lane on the chain's edges (A/B: 0 / 2)
#endif
template <bool MOL, int NG, bool TRACE>
__global__ __launch_bounds__(512) void wf_pipe16_kernel(Wq16K k16) {
  static_assert(NG >= 1 && NG <= WQ_GMAX, "column groups");
  const WqK& a = k16.q;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds;
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(red + WQ_LDS_RED);  // [group][GC] max key of the step
  float* s_x = reinterpret_cast<float*>(s_key + WQ_GMAX * WQ_GC);                          // [group][GC] (MOL) decoded samples
  float* s_xr = s_x + WQ_GMAX * WQ_GC;                                                     // R2: [8 units][16 columns] residual hand-over
  int* s_stale = reinterpret_cast<int*>(s_xr + 8 * 16);                                    // R1: a key lane found a stale granule in its direct fetch
  if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (tests: the fallback path)
  const int blk = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int du = lane >> 4, i = lane & 15;
  const int H = a.R, S = a.S;
  const int n_t3 = MOL ? 1 : a.C / 16;  // fc3 workgroups (MOL: one, holding both row tiles of the <= 32 mixture parameters)
  constexpr int LD = WQ_GC;   // rows of 16 columns: one 128-byte line per feature pair (and per key half)
  const unsigned hop = (unsigned)a.g.hop, total_len = (unsigned)a.g.total_len;
  unsigned g_off[NG];         // this lane's first granule of a sweep of group g (dead columns: the last live one)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int Ng = a.gn0[g + 1] - a.gn0[g];
    g_off[g] = (unsigned)((wave * 32 + (lane >> 4) * 4) * WQ_GC + (i < Ng ? i : (Ng > 0 ? Ng - 1 : 0)));
  }
  int rb = 0;                 // red buffer of the next GEMM
  float rmax = 0.f;           // range bookkeeping of this lane's publications (wq16_track*)
  auto EX = [&](int what, int g, unsigned tag) { return a.ex + ((size_t)g * 2 + (tag & 1)) * WQX_PER + what; };
#define WQ_MARK(role, k)                                                                                   \
  do {                                                                                                     \
    if (TRACE && a.trace && tid == 0 && mark_wg && g == 0 && s >= 1000 && s < 1004)                        \
      a.trace[((role) * 4 + (s - 1000)) * 16 + (k)] = (unsigned long long)wall_clock64();                  \
  } while (0)
#define WQ_MK(role, k) ((TRACE && a.trace && mark_wg && g == 0 && s >= 1000 && s < 1004) ? a.trace + (((role) * 4 + (s - 1000)) * 16 + (k)) : nullptr)

  if (blk < WQ_R1) {
    // ---------------------------------------------------------------------------------------------- R1: rnn1
    const bool mark_wg = blk == 0;
    if (tid < WQ_GMAX * WQ_GC) s_key[tid] = 0ull;
    if (tid == 0) *s_stale = 0;
    Wq16A A0, A1;
    wq16_load_a(k16.h_hh1, 2 * blk, A0);
    wq16_load_a(k16.h_hh1, 2 * blk + 1, A1);
    const int ju = (2 * blk + (wave & 1)) * 4 + du;  // unit of an epilogue lane (waves 0 / 1)
    const unsigned p_off = (unsigned)((ju >> 1) * LD + i);  // its granule in x1 / h1 (units 2j, 2j + 1 share one)
    const float4 bq = a.bhh1q[ju];
    const float gr = a.g1[ju], gz = a.g1[H + ju], gn = a.g1[2 * H + ju], w0 = a.wI0[ju];
    float h1[NG], P1[NG][3], tq[NG][4];
    WqPos pq[NG];  // position of the NEXT table row to rebuild
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      h1[g] = 0.f; P1[g][0] = bq.x; P1[g][1] = bq.y; P1[g][2] = bq.z;  // W_hh . 0 + b_hh
      const int Ng = a.gn0[g + 1] - a.gn0[g];
      const int ncl = a.gn0[g] + (i < Ng ? i : (Ng > 0 ? Ng - 1 : 0));
      pq[g].init((unsigned)ncl * (unsigned)a.g.fold_stride, hop);
      const float4 t4 = wf_cond_row4_fp(a.cond, pq[g].f, pq[g].p, pq[g].pos < total_len, ju, H, a.g.frames);
      pq[g].step(hop);
      tq[g][0] = t4.x; tq[g][1] = t4.y; tq[g][2] = t4.z; tq[g][3] = t4.w;
    }
    __syncthreads();
    for (int s = 0; s <= S; ++s) {
      const unsigned tag_prev = (unsigned)s, tag = (unsigned)s + 1;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int n0 = a.gn0[g], Ng = a.gn0[g + 1] - n0;
        if (Ng <= 0) continue;
        WQ_MARK(0, 0);
        // ---- keys of step s-1 -> sample x of every column of the group (classic {value, 32-bit tag} granules): 32 tiles x Ng lanes
        //      fetch and max them into LDS, ONE barrier, and every finish lane decodes its column's sample itself; the slots are cleared
        //      behind the next barrier of this item (the h1 gather's), which every reader has passed by then ----
        float x = 0.f;
        if (MOL && s > 0) {  // MOL (fatchord_version.py:213-220): the key word IS the sample -- one classic granule per column from the one F3 workgroup
          if (tid < Ng) {
            unsigned xv[1];
            if (!wp_wait<1>(EX(WQX_KEY, g, tag_prev) + tid, 1, tag_prev, xv, a.abort_word)) { wq16_range_report(k16.range_word, rmax); return; }
            s_x[g * WQ_GC + tid] = __uint_as_float(xv[0]);
            if (blk == 0) {
              a.samples[(size_t)(n0 + tid) * S + (s - 1)] = __uint_as_float(xv[0]);
              if (a.progress && n0 + tid == 0 && (s - 1) % 100 == 0) *a.progress = s;
            }
          }
          __syncthreads();
          WQ_MARK(0, 1);
          if (wave < 2) x = i < Ng ? s_x[g * WQ_GC + i] : 0.f;
        } else if (s > 0) {
          const unsigned long long* K = EX(WQX_KEY, g, tag_prev);
          // ONE direct fetch first: when this workgroup is late (it serves the other group's item while the keys arrive) they are all
          // there and the watch would be a second round trip for nothing; a lane that finds a stale granule says so and the
          // workgroup falls back to watch + fetch (every lane polling by itself while the keys are NOT there yet cost 0.7 us per
          // step in round 3)
          const bool key_lane = tid < 32 * Ng && (tid & 31) < n_t3;
          const int tile = tid & 31, n = tid >> 5;
          bool have = false;
#ifdef WQ16_AB_KEYS_WATCH_ONLY
          if (tid == 0) *s_stale = 1;
#else
          if (key_lane) {
            const unsigned long long k0 = wp_get(K + (size_t)tile * 2 * LD + n), k1 = wp_get(K + (size_t)tile * 2 * LD + LD + n);
            have = (unsigned)(k0 >> 32) == tag_prev && (unsigned)(k1 >> 32) == tag_prev;
            if (have) atomicMax(&s_key[g * WQ_GC + n], ((unsigned long long)(unsigned)k0 << 32) | (unsigned long long)(unsigned)k1);
            else *s_stale = 1;
          }
#endif
          __syncthreads();
          if (*s_stale) {  // (uniform; cleared behind the h1 gather's barrier with the key slots)
            wp_watch<1>(K + (size_t)((n_t3 - 1) * 2 + 1) * LD + (Ng - 1), tag_prev, a.abort_word);
            if (key_lane && !have) {
              unsigned kv[2];
              if (!wp_wait<2>(K + (size_t)tile * 2 * LD + n, LD, tag_prev, kv, a.abort_word)) { wq16_range_report(k16.range_word, rmax); return; }
              atomicMax(&s_key[g * WQ_GC + n], ((unsigned long long)kv[0] << 32) | (unsigned long long)kv[1]);
            }
            __syncthreads();
          }
          WQ_MARK(0, 1);
          if (wave < 2) {
            const unsigned long long slot = s_key[g * WQ_GC + i];
            x = (slot && i < Ng) ? 2.f * (float)argmax_class(slot) / ((float)a.C - 1.f) - 1.f : 0.f;
            if (blk == 0 && wave == 0 && du == 0 && i < Ng) {
              a.samples[(size_t)(n0 + i) * S + (s - 1)] = x;
              if (a.progress && n0 + i == 0 && (s - 1) % 100 == 0) *a.progress = s;
            }
          }
        }
        if (s == S) continue;
        // ---- rnn1 finish for (unit ju, column i): wf_finish_kernel's expressions; units 2j, 2j + 1 (lanes 16 apart) share a granule ----
        if (wave < 2) {
          const float rg = wq16_sigmoid((tq[g][0] + x * gr) + P1[g][0]);
          const float zg = wq16_sigmoid((tq[g][1] + x * gz) + P1[g][1]);
          const float ng = wq16_tanh((tq[g][2] + x * gn) + rg * P1[g][2]);
          const float hy = ng + zg * (h1[g] - ng);
          h1[g] = hy;
          const float x1 = (tq[g][3] + x * w0) + hy;
          const float x1o = __shfl_xor(x1, 16, 64), hyo = __shfl_xor(hy, 16, 64);
          if (!(du & 1) && i < Ng) {
            const unsigned tb = wq16_tbit(tag);
            wq16_put(EX(WQX_X1, g, tag) + p_off, x1, x1o, tb);
            wq16_put(EX(WQX_H1, g, tag) + p_off, hy, hyo, tb);   // |h| < 1: in range by construction
            rmax = wq16_track2(rmax, x1, x1o);
          }
        }
        WQ_MARK(0, 2);
        if (TRACE && a.trace && tid == 0 && g == 0 && s == 1001) a.trace[512 + blk] = (unsigned long long)wall_clock64();  // every workgroup's publish time
        if (s + 1 >= S) {  // last step: no hidden half to prepare; the key slots are still cleared behind a barrier
          __syncthreads();
          if (tid < WQ_GC) s_key[g * WQ_GC + tid] = 0ull;
          if (tid == 0) *s_stale = 0;
          continue;
        }
        // ---- next step's table rows: the 29 loads are REQUESTED here and combined at the end of the item (round 5's marks: R1 is the
        //      busiest role -- its item, not the ring, set the period -- and waves 0 / 1, the watching lane among them, sat in
        //      these loads' round trip between the publication and the h1 sweep) ----
        WfCondRaw craw;
        if (wave < 2) wf_cond_load(a.cond, pq[g].f, pq[g].p, pq[g].pos < total_len, ju, H, a.g.frames, craw);
        pq[g].step(hop);
        WQ_MARK(0, 5);
        // ---- hidden half of the next step: P1 = W_hh1 . h1 + b_hh1, kept by the lane that will use it ----
        wh16x8 bh[2], bl[2];
        if (!wq16_gather<2>(EX(WQX_H1, g, tag), g_off[g], tag, Ng, bh, bl, a.abort_word, WQ_MK(0, 6))) { wq16_range_report(k16.range_word, rmax); return; }
        if (tid < WQ_GC) s_key[g * WQ_GC + tid] = 0ull;  // every finish lane has read the step's keys (the gather's barrier is behind us)
        if (tid == 0) *s_stale = 0;
        WQ_MARK(0, 3);
        float sx[4];
        const bool epi = wq16_gemm2(A0, A1, bh, bl, red + rb * 4096, k16.us_hh1, sx);
        rb ^= 1;
        if (epi) { P1[g][0] = sx[0] + bq.x; P1[g][1] = sx[1] + bq.y; P1[g][2] = sx[2] + bq.z; }
        WQ_MARK(0, 4);
        // ---- ... and the fmaf chains of the rows requested above (their loads have long arrived) ----
        if (wave < 2) {
          const float4 t4 = wf_cond_fma(craw);
          tq[g][0] = t4.x; tq[g][1] = t4.y; tq[g][2] = t4.z; tq[g][3] = t4.w;
        }
      }
    }
    wq16_range_report(k16.range_word, rmax);
    return;
  }

  if (blk < WQ_R1 + WQ_R2) {
    // ---------------------------------------------------------------------------------------------- R2: rnn2
    const int b2 = blk - WQ_R1;
    const bool mark_wg = b2 == 0;
    Wq16A A0, A1, A2, A3;
    wq16_load_a(k16.h_rnn2, 2 * b2, A0);
    wq16_load_a(k16.h_rnn2, 2 * b2 + 1, A1);
    wq16_load_a(k16.h_hh2, 2 * b2, A2);
    wq16_load_a(k16.h_hh2, 2 * b2 + 1, A3);
    const int ju = (2 * b2 + (wave & 1)) * 4 + du;
    const unsigned p_off = (unsigned)((ju >> 1) * LD + i);
    // this workgroup's own units 8 b2 .. 8 b2 + 7 (the residual x1 of its epilogue) are the eight halves of ONE fragment word group:
    // features 8 b2 + e = wave xr_wave, k-step xr_st, lanes kb = xr_kb
    const int xr_wave = b2 >> 3, xr_st = (b2 >> 2) & 1, xr_kb = b2 & 3;
    const float4 bq = a.bhh2q[ju];
    float h2[NG], P2[NG][3], g2v[NG][3];
    int g2_row[NG];
    WqPos pq[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      h2[g] = 0.f; P2[g][0] = bq.x; P2[g][1] = bq.y; P2[g][2] = bq.z; g2_row[g] = -1; g2v[g][0] = g2v[g][1] = g2v[g][2] = 0.f;
      const int Ng = a.gn0[g + 1] - a.gn0[g];
      pq[g].init((unsigned)(a.gn0[g] + (i < Ng ? i : (Ng > 0 ? Ng - 1 : 0))) * (unsigned)a.g.fold_stride, hop);
    }
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const unsigned tag = (unsigned)s + 1;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int n0 = a.gn0[g], Ng = a.gn0[g + 1] - n0;
        if (Ng <= 0) continue;
        const int frow = pq[g].frame_row(total_len, a.g.frames);
        pq[g].step(hop);
        if (wave < 2 && frow != g2_row[g]) {  // the per-frame rows change once per hop: kept in registers in between
          const float* gp = a.G2 + (size_t)frow * 3 * H + ju;
          g2v[g][0] = gp[0]; g2v[g][1] = gp[H]; g2v[g][2] = gp[2 * H];
          g2_row[g] = frow;
        }
        WQ_MARK(1, 0);
        wh16x8 bh[2], bl[2];
        if (!wq16_gather<WQ16_CRIT_SLEEP>(EX(WQX_X1, g, tag), g_off[g], tag, Ng, bh, bl, a.abort_word, WQ_MK(1, 6))) { wq16_range_report(k16.range_word, rmax); return; }
        WQ_MARK(1, 1);
        if (wave == xr_wave && (lane >> 4) == xr_kb) {  // residual of the own units: hi + 2^-11 lo, through LDS behind the GEMM's own barrier
          const wh16x8 xh = xr_st ? bh[1] : bh[0], xl = xr_st ? bl[1] : bl[0];
#pragma unroll
          for (int e = 0; e < 8; ++e) s_xr[e * 16 + i] = (float)xh[e] + (float)xl[e] * WQ16_LO_UNSCALE;
        }
        float sx[4];
        const bool epi = wq16_gemm2(A0, A1, bh, bl, red + rb * 4096, k16.us_rnn2, sx, WQ_MK(1, 7));
        rb ^= 1;
        WQ_MARK(1, 5);
        if (epi) {
          const float xr = s_xr[((wave & 1) * 4 + du) * 16 + i];
          const float rg = wq16_sigmoid((sx[0] + g2v[g][0]) + P2[g][0]);
          const float zg = wq16_sigmoid((sx[1] + g2v[g][1]) + P2[g][1]);
          const float ng = wq16_tanh((sx[2] + g2v[g][2]) + rg * P2[g][2]);
          const float hy = ng + zg * (h2[g] - ng);
          h2[g] = hy;
          const float x2 = xr + hy;
          const float x2o = __shfl_xor(x2, 16, 64), hyo = __shfl_xor(hy, 16, 64);
          if (!(du & 1) && i < Ng) {
            const unsigned tb = wq16_tbit(tag);
            wq16_put(EX(WQX_X2, g, tag) + p_off, x2, x2o, tb);
            wq16_put(EX(WQX_H2, g, tag) + p_off, hy, hyo, tb);
            rmax = wq16_track2(rmax, x2, x2o);
          }
        }
        WQ_MARK(1, 2);
        if (TRACE && a.trace && tid == 0 && g == 0 && s == 1001) a.trace[512 + blk] = (unsigned long long)wall_clock64();
        if (s + 1 >= S) continue;
        if (!wq16_gather<2>(EX(WQX_H2, g, tag), g_off[g], tag, Ng, bh, bl, a.abort_word)) { wq16_range_report(k16.range_word, rmax); return; }
        WQ_MARK(1, 3);
        const bool epi2 = wq16_gemm2(A2, A3, bh, bl, red + rb * 4096, k16.us_hh2, sx);
        rb ^= 1;
        if (epi2) { P2[g][0] = sx[0] + bq.x; P2[g][1] = sx[1] + bq.y; P2[g][2] = sx[2] + bq.z; }
        WQ_MARK(1, 4);
      }
    }
    wq16_range_report(k16.range_word, rmax);
    return;
  }

  // ------------------------------------------------------------------------------------------------ F1 / F2 / F3
  const int fr = (blk - WQ_R1 - WQ_R2) / WQ_F, ft = (blk - WQ_R1 - WQ_R2) % WQ_F;  // role 0 / 1 / 2, row tile
  const bool mark_wg = ft == 0;
  if (fr == 2 && ft >= n_t3) return;
  const bool f3mol = MOL && fr == 2;
  Wq16A A0, A1;
  wq16_load_a(fr == 0 ? k16.h_fc1 : fr == 1 ? k16.h_fc2 : k16.h_fc3, ft, A0);
  if (MOL) wq16_load_a(k16.h_fc3, f3mol ? 1 : 0, A1);  // (MOL: the second row tile of the mixture parameters; the image holds two tiles)
  else A1 = A0;
  const float us = fr == 0 ? k16.us_fc1 : fr == 1 ? k16.us_fc2 : k16.us_fc3;
  const float4 b3q = fr == 2 && !MOL ? *reinterpret_cast<const float4*>(a.b_fc3 + ft * 16 + du * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float bmol[4] = {0.f, 0.f, 0.f, 0.f};  // MOL: bias of rows wave * 16 + du * 4 + r (waves 0 / 1 = the two row tiles)
  if (f3mol && wave < 2) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + du * 4 + r;
      bmol[r] = row < a.C ? a.b_fc3[row] : 0.f;
    }
  }
  const unsigned y_off = (unsigned)((ft * 8 + du * 2) * LD + i);  // rows ft * 16 + du * 4 + 0..3 = feature pairs ft * 8 + du * 2 + 0, 1
  float4 fpre[NG];
  int f_row[NG];
  WqPos pq[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    fpre[g] = make_float4(0.f, 0.f, 0.f, 0.f); f_row[g] = -1;
    const int Ng = a.gn0[g + 1] - a.gn0[g];
    pq[g].init((unsigned)(a.gn0[g] + (i < Ng ? i : (Ng > 0 ? Ng - 1 : 0))) * (unsigned)a.g.fold_stride, hop);
  }
  const int src = fr == 0 ? WQX_X2 : fr == 1 ? WQX_Y1 : WQX_Y2;
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    const unsigned tag = (unsigned)s + 1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int n0 = a.gn0[g], Ng = a.gn0[g + 1] - n0;
      if (Ng <= 0) continue;
      const int ncl = n0 + (i < Ng ? i : Ng - 1);
      float lgn[4] = {0.f, 0.f, 0.f, 0.f};
      if (fr < 2) {
        const int frow = pq[g].frame_row(total_len, a.g.frames);
        pq[g].step(hop);
        if (wave == 0 && frow != f_row[g]) {
          fpre[g] = *reinterpret_cast<const float4*>((fr == 0 ? a.F1 : a.F2) + (size_t)frow * a.FC + ft * 16 + du * 4);
          f_row[g] = frow;
        }
      } else if (wave == 0 && !MOL) {  // the step's Gumbel noise does not depend on the data: drawn before the wait
        uint32_t grn[4];
        philox4x32((uint32_t)s, (uint32_t)ncl, (uint32_t)((ft * 16 + du * 4) >> 2), 0x57415645u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), grn);
#pragma unroll
        for (int r = 0; r < 4; ++r) lgn[r] = logf(-logf(u32_to_unit(grn[r])));
      }
      WQ_MARK(2 + fr, 0);
      wh16x8 bh[2], bl[2];
      if (!wq16_gather<WQ16_CRIT_SLEEP>(EX(src, g, tag), g_off[g], tag, Ng, bh, bl, a.abort_word, WQ_MK(2 + fr, 6))) { wq16_range_report(k16.range_word, rmax); return; }
      WQ_MARK(2 + fr, 1);
      float sx[4];
      if (f3mol) {
        // ---- MOL: both row tiles against the gathered y2, the mixture parameters of the group's columns through LDS (the red half the
        //      NEXT product will use: free until the barrier of its gather), then wf_fc3_mol_kernel's sampler per column -- same
        //      Philox words, same expressions -- and the sample itself as the key (wavernn_pipe.h's F3, on this kernel's products) ----
        const bool epi2 = wq16_gemm2(A0, A1, bh, bl, red + rb * 4096, us, sx);
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
      const bool epi = wq16_gemm1(A0, bh, bl, red + rb * 4096, us, sx, WQ_MK(2 + fr, 7));
      rb ^= 1;
      WQ_MARK(2 + fr, 3);
      if (!epi) continue;
      if (fr < 2) {
        if (i < Ng) {
          unsigned long long* Y = EX(fr == 0 ? WQX_Y1 : WQX_Y2, g, tag) + y_off;
          const unsigned tb = wq16_tbit(tag);
          const float y0 = fmaxf(sx[0] + fpre[g].x, 0.f), y1 = fmaxf(sx[1] + fpre[g].y, 0.f);
          const float y2 = fmaxf(sx[2] + fpre[g].z, 0.f), y3 = fmaxf(sx[3] + fpre[g].w, 0.f);
          wq16_put(Y, y0, y1, tb);
          wq16_put(Y + LD, y2, y3, tb);
          rmax = wq16_track4(rmax, y0, y1, y2, y3);
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
      if (TRACE && a.trace && tid == 0 && g == 0 && s == 1001) a.trace[512 + blk] = (unsigned long long)wall_clock64();
    }
  }
  wq16_range_report(k16.range_word, rmax);
#undef WQ_MARK
#undef WQ_MK
}

}  // namespace mb
