// WaveRNN sample loop for FEW fold columns (batched=False and short batched utterances, N <= 4) as ONE persistent
// launch with the weights resident on chip -- the A/B partner of the 5-launch chain of wavernn_fast.h
// (profiles/r02_wavernn_persistent_ab.json).
//
// With <= 8 columns the chain is pure latency: 5 dependent launches x (boundary + ramp + first HBM round trip) per
// sample, 16.3 MB of weights re-streamed every step for a few MFLOP.  Here 192 workgroups stay resident for the whole
// utterance, every weight tile lives in LDS from the first step on, and the layers hand their output vectors to each
// other through 8-byte {value, step tag} granules in device memory (one relaxed agent-scope store / load each:
// global_store/load_dwordx2 sc1, MI355X_MICROARCH.md "R2 granule" -- no fences, the tag IS the flag).
//
//   on-chain workgroups g = 0..63 (the dependent chain of fatchord_version.py:190-228):
//     keys(s-1) -> x      every workgroup reduces the 32 fc3 tiles' Gumbel-argmax keys itself
//     finish              rnn1 elementwise for ALL 512 units, redundantly (no exchange: h1 is private state)
//     rnn2                row tiles 2g, 2g+1 (8 units; h2 of those units is private state)  -> publishes x2, h2
//     fc1  (g <  32)      tile g on x2                                                      -> publishes y1
//     fc2  (g >= 32)      tile g-32 on y1                                                   -> publishes y2
//     fc3  (g >= 32)      tile g-32 on y2 + Gumbel-argmax                                   -> publishes keys
//   off-chain workgroups g = 64..191: row tile g-64 of W_hh1 and of W_hh2 -- the hidden halves of the NEXT step's GRUs
//     (they have a whole step of slack), fed by h1 (published by workgroup 0) and h2, publishing P1 / P2.
//
// Arithmetic: the GEMM of a row tile is the 8-wave K split of fm_gemm.h with the same MFMA sequence and the same
// wave-order reduction, epilogues are the expressions of wavernn_fast.h, the sampler draws the same Philox words:
// the sample stream is bit-identical to the chain's (tests/test_wavernn_gpu.py).
//
// Every exchange buffer is double-buffered by tag parity.  Why that is enough: a granule of tag t is overwritten by
// tag t+2, and the chain is a cycle -- whoever writes tag t+2 of any buffer has (transitively) consumed a value that
// needed every reader of tag t of that buffer to have finished (derivation in DESIGN.md section 4d).
// Every spin has a bail-out: after 0.2 s of wall clock a waiting workgroup raises the abort word and the launch drains; the host
// then runs the launch chain instead (same samples) -- co-residency of the 192 workgroups is not something a launch can
// demand when other work shares the GPU.
#pragma once
#include "wavernn_fast.h"
#include "granule.h"

namespace mb {

constexpr int WP_NCOL = 4;        // fold columns supported (LDS budget of the busiest workgroup: 152 KB of 160)
constexpr int WP_ON = 64;         // on-chain workgroups
constexpr int WP_OFF = 128;       // off-chain workgroups (one GRU row tile of each hidden half)
// exchange area, in granules, per parity
enum { WPX_X2 = 0, WPX_H2 = 8192, WPX_H1 = 16384, WPX_Y1 = 24576, WPX_Y2 = 32768, WPX_P1 = 40960, WPX_P2 = 65536,
       WPX_KEY = 90112, WPX_PER_PARITY = 91136 };
inline size_t wp_exchange_bytes() { return (size_t)2 * WPX_PER_PARITY * 8 + 256 + 8192; }  // + abort word + diagnostics marks

struct WpK {
  const float* w_rnn2; const float* w_fc1; const float* w_fc2; const float* w_fc3; const float* w_hh1; const float* w_hh2;
  const float4* bhh1q; const float4* bhh2q; const float* b_fc3; const float* g1; const float* wI0;
  WfCond cond; const float* G2; const float* F1; const float* F2;
  WfGeom g;
  unsigned long long* ex; int* abort_word;
  float* samples; volatile int* progress;
  unsigned long long seed; int R, FC, C, S, N;
  unsigned long long* trace;  // diagnostics (MBHIP_DIAG=wp_trace=<file>): wall-clock marks of workgroups 0 and 32, steps 1000..1003
};

// spin until the NQ granules p[q * stride] all carry `tag`; false = aborted
template <int NQ>
__device__ __forceinline__ bool wp_wait(const unsigned long long* p, const size_t stride, const unsigned tag, unsigned (&out)[NQ], int* abort_word) {
  unsigned long long v[NQ];
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
    bool ok = true;
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = wp_get(p + q * stride);
    {  // all tags in one xor / or chain (a chain of && compiled to nested exec-mask branches, wavernn_pipe16.h)
      unsigned stale_ = 0u;
#pragma unroll
      for (int q = 0; q < NQ; ++q) stale_ |= (unsigned)(v[q] >> 32) ^ tag;
      ok = ok && stale_ == 0u;
    }
    if (ok) break;
    if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) out[q] = (unsigned)v[q];
  return true;
}

// the same in two halves: issue the loads early (the producer had a whole step), check -- and only then spin -- at the use
template <int NQ>
__device__ __forceinline__ void wp_issue(const unsigned long long* p, const size_t stride, unsigned long long (&v)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) v[q] = wp_get(p + q * stride);
}
template <int NQ>
__device__ __forceinline__ bool wp_take(const unsigned long long* p, const size_t stride, const unsigned tag, const unsigned long long (&v)[NQ],
                                        unsigned (&out)[NQ], int* abort_word) {
  bool ok = true;
  {  // all tags in one xor / or chain (a chain of && compiled to nested exec-mask branches, wavernn_pipe16.h)
    unsigned stale_ = 0u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) stale_ |= (unsigned)(v[q] >> 32) ^ tag;
    ok = ok && stale_ == 0u;
  }
  if (!ok) return wp_wait<NQ>(p, stride, tag, out, abort_word);
#pragma unroll
  for (int q = 0; q < NQ; ++q) out[q] = (unsigned)v[q];
  return true;
}

// One lane of the workgroup spins on one granule until the step's tag shows up; nobody else touches memory meanwhile
// (the latency of a hand-off is set by the CONSUMER compute unit's own memory queue: 512 pollers per unit made every
// exchange 2.5-3 us).  Ends in a barrier.
template <int SLEEP>
__device__ __forceinline__ bool wp_watch(const unsigned long long* p, const unsigned tag, int* abort_word) {
  // (one lane per PRODUCER -- 32 to 64 watching lanes, so that the sweep never starts before the slowest producer --
  //  measured slower, 13.6 vs 11.9 us per step: what counts is how few requests sit in this unit's memory queue)
  if (threadIdx.x == 0) {
    unsigned long long t0 = 0;
    for (int tries = 0; (unsigned)(wp_get(p) >> 32) != tag; ++tries) {
      if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) break;
      __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
  __syncthreads();
  return true;
}
// B fragments of this lane for the 4 k-blocks of its wave from an exchange vector, dense [k][N columns]; dead columns = 0.
// One watching lane first (wp_watch), then every live lane fetches its 16 granules.
template <int SLEEP>
__device__ __forceinline__ bool wp_gather(const unsigned long long* vec, const unsigned tag, const int N, float4 (&b)[4], int* abort_word) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
  wp_watch<SLEEP>(vec + (size_t)511 * N + (N - 1), tag, abort_word);
#pragma unroll
  for (int p = 0; p < 4; ++p) b[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i >= N) return true;
  // granule of (feature k, column n) at k * N + n; this lane: k = (wave + 8 p) * 16 + kq * 4 + c
  const unsigned long long* base = vec + ((size_t)(wave * 16 + kq * 4) * N + i);
  unsigned long long v[16];
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) v[p * 4 + c] = wp_get(base + ((size_t)p * 128 + c) * N);
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

// One 16-row tile (weights in LDS at lw: [32 k-blocks][BLK]) x one column tile, K = 512: the MFMA sequence and the
// reduction order of fm_gemm<1, 4, 4, RL, 1>.  Returns true for wave 0 with the sums in sx.
template <int RL>
__device__ __forceinline__ bool wp_gemm(const float* lw, const float4 (&b)[4], float* red, float (&sx)[4]) {
  constexpr int BLK = 4 * RL * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;
  const float* wl = lw + ((u * RL + tau) * 4 + kq) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float4 a = *reinterpret_cast<const float4*>(wl + (size_t)(wave + 8 * p) * BLK);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[p].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[p].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[p].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[p].w, acc, 0, 0, 0);
  }
  float4* red4 = reinterpret_cast<float4*>(red);
  red4[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
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

// Two row tiles against the same B fragments in one pass (one barrier); wave tt (0 / 1) gets tile tt's sums.
template <int RL>
__device__ __forceinline__ bool wp_gemm2(const float* lw, const int tile_floats, const float4 (&b)[4], float* red, float (&sx)[4]) {
  constexpr int BLK = 4 * RL * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;
  const float* wl = lw + ((u * RL + tau) * 4 + kq) * 4;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float4 a0 = *reinterpret_cast<const float4*>(wl + (size_t)(wave + 8 * p) * BLK);
    const float4 a1 = *reinterpret_cast<const float4*>(wl + tile_floats + (size_t)(wave + 8 * p) * BLK);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b[p].x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b[p].x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b[p].y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b[p].y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b[p].z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b[p].z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b[p].w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b[p].w, acc1, 0, 0, 0);
  }
  float4* red4 = reinterpret_cast<float4*>(red);  // [2 tiles][8 waves][64]
  red4[wave * 64 + lane] = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
  red4[512 + wave * 64 + lane] = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
  __syncthreads();
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

__device__ __forceinline__ void wp_copy_tile(float* dst, const float* __restrict__ src, const int floats) {
  for (int i = threadIdx.x * 4; i < floats; i += blockDim.x * 4)
    *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
}

// (The round-2 kernel for 2..4 columns -- MFMA tiles, every on-chain workgroup recomputing the whole rnn1 finish -- lived here, with the
//  MFMA form of the one-column kernel: superseded by wavernn_pipe.h / wavernn_pipe16.h and removed from the library in round 4; source
//  under tools/rejected/wf_persist_kernel.h.  The helpers above serve the pipelined kernels.)
#define WP_MARK(k)                                                                                          \
  do {                                                                                                      \
    if (a.trace && tid == 0 && (g == 0 || g == 32) && s >= 1000 && s < 1004)                                \
      a.trace[((g ? 1 : 0) * 4 + (s - 1000)) * 16 + (k)] = (unsigned long long)wall_clock64();              \
  } while (0)

// ================================================================================================================
// One fold column (batched=False): the same protocol with the products on the vector ALU.  A 16x16x4 fp32 MFMA on a
// single live column spends 15/16 of the matrix pipe on dead columns (64 of them per SIMD and step = 0.85 us for
// rnn2 alone); a lane-per-(row, k-slice) fmaf chain in the MFMA's own order -- k-block kb of slice w = kb mod 8, inside
// a block MFMA c takes k = 4 kq + c for kq = 0..3 -- gives the same bits (the fp32 MFMA is a k-ordered fmaf chain,
// MI355X_MICROARCH.md section 3; checked sample for sample against the chain) in a fifth of the time.
// Chain-major LDS image of a packed row tile: thread tt = w * 16 + r of the tile owns the 64 weights of (row r, slice w)
// in chain order q = p * 16 + c * 4 + kq  (k = (w + 8 p) * 16 + 4 kq + c), stored [q / 4][tt][q % 4]: one conflict-free
// ds_read_b128 per four chain links.  The x vector is kept in the same order: xperm[w * 64 + q].
__device__ __forceinline__ int wp_xperm(const int k) {  // feature k -> position in the chain-ordered vector
  const int kb = k >> 4, kq = (k >> 2) & 3, c = k & 3;
  return (kb & 7) * 64 + (kb >> 3) * 16 + c * 4 + kq;
}
template <int RL>
__device__ __forceinline__ void wp_copy_tile_chain(float* dst, const float* __restrict__ src) {
  constexpr int BLK = 4 * RL * 16;
  for (int d = threadIdx.x; d < 8192; d += blockDim.x) {
    const int e = d & 3, tt = (d >> 2) & 127, q = (d >> 9) * 4 + e;
    const int w = tt >> 4, r = tt & 15, p = q >> 4, c = (q >> 2) & 3, kq = q & 3;
    const int u = r >> 2, tau = (r & 3) < RL ? (r & 3) : RL - 1;
    dst[d] = src[(w + 8 * p) * BLK + ((u * RL + tau) * 4 + kq) * 4 + c];
  }
}
template <int NTILE>
__device__ __forceinline__ void wp_dot(const float* lw, const float* xp, float* red) {
  const int t = threadIdx.x;
  if (t < NTILE * 128) {
    const int tile = t >> 7, tt = t & 127, w = tt >> 4, r = tt & 15;
    const float4* a4 = reinterpret_cast<const float4*>(lw + tile * 8192) + tt;
    const float4* x4 = reinterpret_cast<const float4*>(xp + w * 64);
    float acc = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < 16; ++q4) {
      const float4 av = a4[q4 * 128], xv = x4[q4];
      acc = fmaf(av.x, xv.x, acc); acc = fmaf(av.y, xv.y, acc); acc = fmaf(av.z, xv.z, acc); acc = fmaf(av.w, xv.w, acc);
    }
    red[(tile * 16 + r) * 8 + w] = acc;
  }
  __syncthreads();
}
// the 8 slice sums of a row in wave order, as fm_gemm reduces them
__device__ __forceinline__ float wp_rowsum(const float* red, const int row) {
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) s += red[row * 8 + w];
  return s;
}
// thread t fetches feature t of an exchange vector (column 0) into LDS (chain order); SLEEP as above for the watch
template <int SLEEP>
__device__ __forceinline__ bool wp_fetch1(const unsigned long long* vec, const unsigned tag, float* xs, int* abort_word) {
  // single-column layout: granule of feature k at vec + k (a wave's 64 granules share four 128-byte lines)
  // one watching lane first, also on the chain (all 512 lanes polling their own granule: 11.8 vs 10.3 us per step)
  wp_watch<SLEEP>(vec + 511, tag, abort_word);
  const unsigned long long* p = vec + threadIdx.x;
  unsigned long long v;
  unsigned long long t0 = 0;
  for (int tries = 0;; ++tries) {
    v = wp_get(p);
    if ((unsigned)(v >> 32) == tag) break;
    if ((tries & 1023) == 1023 && wp_lost(tries, t0, abort_word)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  xs[wp_xperm(threadIdx.x)] = __uint_as_float((unsigned)v);
  return true;
}

constexpr int WP1_LDS_W = 4 * 8192;  // chain-major tiles are 8192 floats each (GRU tiles carry their dead fourth rows)
constexpr size_t WP1_LDS_BYTES = (size_t)(WP1_LDS_W + 256 + 512 + 512) * 4 + 64;

__global__ __launch_bounds__(512) void wf_persist1_kernel(WpK a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lw = lds;
  float* red = lds + WP1_LDS_W;  // [32 rows][8 slices]
  float* xs1 = red + 256;        // x1 of this step
  float* xg = xs1 + 512;         // the vector fetched last
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(xg + 512);
  if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (tests: the fallback path)
  const int g = blockIdx.x, tid = threadIdx.x;
  const int H = a.R, S = a.S;
  const int n_t3 = a.C / 16;
  auto EX = [&](int what, unsigned tag) { return a.ex + (size_t)(tag & 1) * WPX_PER_PARITY + what; };

  if (g >= WP_ON) {
    // ------------------------------------------------------------------ off-chain: hidden halves of the next step
    const int mt = g - WP_ON;
    wp_copy_tile_chain<3>(lw, a.w_hh1 + (size_t)mt * 6144);
    wp_copy_tile_chain<3>(lw + 8192, a.w_hh2 + (size_t)mt * 6144);
    const float4 bq1q = a.bhh1q[mt * 4 + ((tid >> 2) & 3)], bq2q = a.bhh2q[mt * 4 + ((tid >> 2) & 3)];  // thread t < 16: unit t >> 2, gate t & 3
    xg[tid] = 0.f;
    __syncthreads();
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        if (s > 0) {
          if (!wp_fetch1<6>(EX(which ? WPX_H2 : WPX_H1, (unsigned)s), (unsigned)s, xg, a.abort_word)) return;
          __syncthreads();
        }
        wp_dot<1>(lw + which * 8192, xg, red);
        if (tid < 16) {  // dense [unit][r, z, n, -]: the tile's 16 granules as ONE 128-byte store (8-byte pieces of a line, stored
                         // one by one, are merged one after the other at the memory side and reach the readers ~1 us later)
          const float4 bq = which ? bq2q : bq1q;
          const int comp = tid & 3;
          const float bv = comp == 0 ? bq.x : comp == 1 ? bq.y : bq.z;
          wp_put(EX(which ? WPX_P2 : WPX_P1, (unsigned)s + 1) + (size_t)mt * 16 + tid, comp < 3 ? wp_rowsum(red, tid) + bv : 0.f, (unsigned)s + 1);
        }
        // (red / xg are rewritten only behind the next fetch's barrier, which threads 0..3 reach after these reads)
        if (s == 0) __syncthreads();
      }
    }
    return;
  }

  // -------------------------------------------------------------------- on-chain
  const bool lo = g < 32;
  const int ft = lo ? g : g - 32;
  wp_copy_tile_chain<3>(lw, a.w_rnn2 + (size_t)(2 * g) * 6144);
  wp_copy_tile_chain<3>(lw + 8192, a.w_rnn2 + (size_t)(2 * g + 1) * 6144);
  wp_copy_tile_chain<4>(lw + 16384, (lo ? a.w_fc1 : a.w_fc2) + (size_t)ft * 8192);
  if (!lo && ft < n_t3) wp_copy_tile_chain<4>(lw + 24576, a.w_fc3 + (size_t)ft * 8192);
  if (tid == 0) s_key[0] = 0ull;
  const int j = tid;                    // finish: unit j
  const int et = tid >> 2, du = tid & 3;  // epilogue threads: tid < 8 (rnn2: tile et, unit quad du) / tid < 4 (fc)
  const float gr = a.g1[j], gz = a.g1[H + j], gn = a.g1[2 * H + j], w0 = a.wI0[j];
  float h1 = 0.f, h2 = 0.f, tq[4];
  {
    const float4 t4 = wf_cond_row4(a.cond, wf_pos(a.g, 0, 0), (unsigned)a.g.total_len, j, H, a.g.frames);
    tq[0] = t4.x; tq[1] = t4.y; tq[2] = t4.z; tq[3] = t4.w;
  }
  float g2r = 0.f, g2z = 0.f, g2n = 0.f; int g2_row = -1;
  float fpre1 = 0.f; int f_row = -1;
  const float4 b3q = (!lo && ft < n_t3 && tid < 4) ? *reinterpret_cast<const float4*>(a.b_fc3 + ft * 16 + du * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  float x = 0.f;
  __syncthreads();

  for (int s = 0; s <= S; ++s) {
    const unsigned tag_prev = (unsigned)s, tag = (unsigned)s + 1;
    unsigned long long p1v[3], p2v[3];
    if (s < S) wp_issue<3>(EX(WPX_P1, tag) + (size_t)j * 4, 1, p1v);
    const int frow = s < S ? wf_frame_row(a.g, 0, s) : 0;
    WP_MARK(0);
    // ---- A: keys of step s-1 -> sample x ----
    if (s > 0) {
      // (the 32 key lanes polling directly, without the watching lane: 11.0 vs 10.3 us per step)
      wp_watch<1>(EX(WPX_KEY, tag_prev) + (size_t)(n_t3 - 1) * 4 + 1, tag_prev, a.abort_word);
      if (tid < n_t3) {
        unsigned kv[2];
        if (!wp_wait<2>(EX(WPX_KEY, tag_prev) + (size_t)tid * 4, 1, tag_prev, kv, a.abort_word)) return;
        atomicMax(&s_key[0], ((unsigned long long)kv[0] << 32) | (unsigned long long)kv[1]);
      }
      __syncthreads();
      WP_MARK(1);
      const unsigned long long slot = s_key[0];
      x = slot ? 2.f * (float)argmax_class(slot) / ((float)a.C - 1.f) - 1.f : 0.f;
      if (g == 0 && tid == 0) {
        a.samples[s - 1] = x;
        if (a.progress && (s - 1) % 100 == 0) *a.progress = s;
      }
    }
    if (s == S) break;
    if (tid < 8) wp_issue<3>(EX(WPX_P2, tag) + (size_t)((2 * g + et) * 4 + du) * 4, 1, p2v);
    WP_MARK(3);
    // ---- B: rnn1 finish, unit j ----
    {
      unsigned pu[3];
      if (!wp_take<3>(EX(WPX_P1, tag) + (size_t)j * 4, 1, tag, p1v, pu, a.abort_word)) return;
      const float rg = sigmoidf_((tq[0] + x * gr) + __uint_as_float(pu[0]));
      const float zg = sigmoidf_((tq[1] + x * gz) + __uint_as_float(pu[1]));
      const float ng = tanhf((tq[2] + x * gn) + rg * __uint_as_float(pu[2]));
      const float hy = ng + zg * (h1 - ng);
      h1 = hy;
      xs1[wp_xperm(j)] = (tq[3] + x * w0) + hy;
      if ((j >> 3) == g) wp_put(EX(WPX_H1, tag) + j, hy, tag);  // every workgroup has all of h1: each publishes 8 units
    }
    __syncthreads();
    WP_MARK(5);
    if (tid == 0) s_key[0] = 0ull;  // everybody decoded x before the barrier; the next atomicMax is a step away
    // ---- C: next step's table rows ----
    if (s + 1 < S) {
      const float4 t4 = wf_cond_row4(a.cond, wf_pos(a.g, 0, s + 1), (unsigned)a.g.total_len, j, H, a.g.frames);
      tq[0] = t4.x; tq[1] = t4.y; tq[2] = t4.z; tq[3] = t4.w;
    }
    // ---- D: rnn2, row tiles 2g and 2g+1 ----
    wp_dot<2>(lw, xs1, red);
    WP_MARK(11);
    if (tid < 8) {
      const int ju = (2 * g + et) * 4 + du;
      if (frow != g2_row) {
        const float* gp = a.G2 + (size_t)frow * 3 * H + ju;
        g2r = gp[0]; g2z = gp[H]; g2n = gp[2 * H];
        g2_row = frow;
      }
      const int row = et * 16 + du * 4;
      const float s0 = wp_rowsum(red, row), s1 = wp_rowsum(red, row + 1), s2 = wp_rowsum(red, row + 2);
      unsigned pu[3];
      if (!wp_take<3>(EX(WPX_P2, tag) + (size_t)ju * 4, 1, tag, p2v, pu, a.abort_word)) return;
      WP_MARK(12);
      const float xr = xs1[wp_xperm(ju)];
      const float rg = sigmoidf_((s0 + g2r) + __uint_as_float(pu[0]));
      const float zg = sigmoidf_((s1 + g2z) + __uint_as_float(pu[1]));
      const float ng = tanhf((s2 + g2n) + rg * __uint_as_float(pu[2]));
      const float hy = ng + zg * (h2 - ng);
      h2 = hy;
      wp_put(EX(WPX_X2, tag) + ju, xr + hy, tag);
      wp_put(EX(WPX_H2, tag) + ju, hy, tag);
    }
    WP_MARK(6);
    // ---- E: fc1 (lo) | fc2 then fc3 (hi) ----
    if (!wp_fetch1<1>(EX(lo ? WPX_X2 : WPX_Y1, tag), tag, xg, a.abort_word)) return;
    __syncthreads();
    WP_MARK(7);
    wp_dot<1>(lw + 16384, xg, red);
    WP_MARK(8);
    if (tid < 16) {  // the tile's 16 outputs as ONE 128-byte store
      if (frow != f_row) {
        fpre1 = (lo ? a.F1 : a.F2)[(size_t)frow * a.FC + ft * 16 + tid];
        f_row = frow;
      }
      wp_put(EX(lo ? WPX_Y1 : WPX_Y2, tag) + (ft * 16 + tid), fmaxf(wp_rowsum(red, tid) + fpre1, 0.f), tag);
    }
    if (!lo && ft < n_t3) {
      // the Gumbel noise of this step does not depend on the data: drawn before the wait for y2, off the key edge
      float lgn[4] = {0.f, 0.f, 0.f, 0.f};
      if (tid < 4) {
        uint32_t grn[4];
        philox4x32((uint32_t)s, 0u, (uint32_t)((ft * 16 + du * 4) >> 2), 0x57415645u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), grn);
#pragma unroll
        for (int r = 0; r < 4; ++r) lgn[r] = logf(-logf(u32_to_unit(grn[r])));
      }
      if (!wp_fetch1<1>(EX(WPX_Y2, tag), tag, xg, a.abort_word)) return;
      __syncthreads();
      WP_MARK(9);
      wp_dot<1>(lw + 24576, xg, red);
      WP_MARK(10);
      if (tid < 4) {  // wf_fc3_kernel's sampler for column 0
        const float bv[4] = {b3q.x, b3q.y, b3q.z, b3q.w};
        float best = -INFINITY;
        int bcls = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = ft * 16 + du * 4 + r;
          const float v = wp_rowsum(red, du * 4 + r) + bv[r];
          const float gmb = v - lgn[r];
          if (gmb > best) { best = gmb; bcls = row; }
        }
        unsigned long long pk = pack_argmax(best, bcls);
        const unsigned long long o1 = __shfl_xor(pk, 1, 64);
        pk = o1 > pk ? o1 : pk;
        const unsigned long long o2 = __shfl_xor(pk, 2, 64);
        pk = o2 > pk ? o2 : pk;
        // both halves in one 16-byte store, every tile in its own 32-byte word (all four lanes hold the maximum)
        if (tid < 2) wp_put_u(EX(WPX_KEY, tag) + (size_t)ft * 4 + tid, tid == 0 ? (unsigned)(pk >> 32) : (unsigned)pk, tag);
      }
    }
  }
}

}  // namespace mb
