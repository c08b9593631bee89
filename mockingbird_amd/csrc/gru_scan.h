// The bidirectional GRU scan of a CBHG (models/synthesizer/models/sublayer/cbhg.py:76-77: nn.GRU(channels, channels // 2,
// bidirectional); encoder: 111 text positions x 128 units, postnet: every mel frame x 256 units) as ONE resident launch.
//
// The launch-per-step form (rnn_launch_dual_gru: both directions of a step in one launch) costs a kernel boundary per step:
// 5.56 us x (T_text + frames) = 2.8 of the 6.3 ms the encoder + postnet take at B = 32 (profiles/r03_bench_kernel_stats.csv).
// The recurrence itself is tiny -- W_hh is 3 Hg x Hg, 196 KB (encoder) / 786 KB (postnet) in fp32 -- so here it never leaves
// the chip: a direction is split over G = Hg / 64 workgroups, each owning 64 units (all three gates); a wave keeps the
// A fragments of its 16 units in REGISTERS for the whole scan, as fp16 hi / lo halves of W_hh 2^s (the error-compensated product
// of conv1d.hip's split path: x = xh + xl, w 2^s = wh + wl, three v_mfma_f32_16x16x32_f16 per block, fp32 accumulate; h is scaled
// by 2^10 so that its low half stays an fp16 normal down to |h| = 1.2e-4).  Per step: B fragments of h(t-1) from LDS,
// 3 gates x NT column tiles x Hg / 32 k-steps x 3 products per wave, the torch GRUCell update in the accumulator layout
// (a lane owns 4 units x 1 column: r, z, n of a unit meet in the same lane), h(t) written to the [B][2 Hg][F] sequence, its
// hi / lo halves to the owner's LDS and -- for the other G - 1 workgroups of the direction -- as 8-byte tagged granules
// {hi | lo << 16, step tag} (granule.h: one relaxed agent-scope store / load each, the tag is the flag, two parities).
// W_ih x + b_ih for every t is one GEMM per direction beforehand (the time-major table `ih`, as before).
//
// Why two parities suffice: a workgroup writes tag t + 2 only after it has consumed every tag t + 1, which their owners wrote
// after consuming every tag t.  Every spin has the wall-clock bail-out of granule.h: a lost hand-off raises the abort word, the
// launch drains, and the host runs the launch-per-step scan instead (MBHIP_GRU_SCAN=0 selects that one outright).
#pragma once
#include <atomic>
#include <type_traits>
#include "granule.h"

namespace mb {

typedef _Float16 gs_h16;
typedef _Float16 gs_h16x8 __attribute__((ext_vector_type(8)));

struct GruScanK {
  const float* whh[2];      // raw torch weight_hh [3 Hg][Hg] (r, z, n) per direction
  const float* bhh[2];      // [3 Hg]
  const float* ih[2];       // [B][F][3 Hg]: W_ih x + b_ih
  float* seq_tm;            // TIME-major scratch [F][B][2 Hg] (forward units then backward units): 16-byte stores, 64-byte segments;
                            // gru_scan_transpose_kernel turns it into the [B][2 Hg][F] sequence the next conv reads
  unsigned long long* ex;   // [2 directions][2 parities][32 columns][Hg] granules, zeroed before the launch
  int* abort_word;
  int B, F, Hg;
  float unscale[2];         // 2^-(s + 10) per direction
  float wscale[2];          // 2^s
  int dbg;                  // diagnostics (MBHIP_DIAG=gs_dbg=<bits>, wrong results): 1 = no hand-off, 2 = no sequence stores, 4 = no products, 8 = no x loads
};

constexpr float GS_HSCALE = 1024.f;
// the diagnostics bits exist only under -DMB_GS_DBG_BUILD (tools/build_variant.sh): run-time tests around loads / stores leave
// the compiler's vmcnt bookkeeping imprecise (it then waits for the sequence stores' acknowledgements in every step)
#ifdef MB_GS_DBG_BUILD
#define GS_DBG(a_, bit) ((a_).dbg & (bit))
#else
#define GS_DBG(a_, bit) 0
#endif

__device__ __forceinline__ void gs_split8(const float (&v)[8], const float sc, gs_h16x8& hi, gs_h16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float x = v[e] * sc;
    const gs_h16 h = (gs_h16)x;
    hi[e] = h;
    lo[e] = (gs_h16)(x - (float)h);
  }
}

// sigmoid / tanh on the hardware exp2 and reciprocal (1 ulp each; the epilogue of a lone wave per SIMD was 2 of the step's
// 8.8 us with expf / tanhf / a division per value)
__device__ __forceinline__ float gs_sigmoid(const float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float gs_tanh(const float x) {  // 1 - 2 / (1 + e^(2x)); exp2 saturates to 0 / inf at the ends: -1 / +1
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// KS = Hg / 32 k-steps (4: encoder, 8: postnet); NT = column tiles of 16 utterances; NW = waves per workgroup (16 units each):
// the encoder's 128 units are ONE workgroup of 8 waves per direction (no hand-off at all), the postnet's 256 are 4 x 4 waves
// (a wave holds 192 VGPRs of weights there: one wave per SIMD).
template <int KS, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void gru_scan_kernel(GruScanK a) {
  constexpr int Hg = KS * 32, HP = Hg + 8;  // LDS row stride in halves (odd multiple of 16 bytes)
  constexpr int NC = NT * 16, UW = NW * 16, NTH = NW * 64;
  constexpr int G = Hg / UW;                // workgroups per direction
  extern __shared__ __attribute__((aligned(16))) unsigned char gs_lds[];
  typedef gs_h16 (*SB)[2][NC][HP];
  SB sB = reinterpret_cast<SB>(gs_lds);  // [parity][hi | lo][column][unit]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dir = blockIdx.x / G, g = blockIdx.x - dir * G;
  const int c0 = blockIdx.y * NC;  // column group (round 5: a batch of 32 as two groups of one tile each, side by side -- see gru_scan_launch)
  const int u0 = g * UW + wave * 16;
  const int row = lane & 15, kq = lane >> 4;
  if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;

  // ---- the wave's slice of W_hh 2^s as fp16 hi / lo A fragments (lane: unit u0 + row, k = 32 ks + 8 kq + e), for the whole scan ----
  gs_h16x8 ah[3][KS], al[3][KS];
  {
    const float* W = a.whh[dir];
    const float sc = a.wscale[dir];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float* p = W + (size_t)(gate * Hg + u0 + row) * Hg + ks * 32 + kq * 8;
        const float4 q0 = *reinterpret_cast<const float4*>(p), q1 = *reinterpret_cast<const float4*>(p + 4);
        const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        gs_split8(v, sc, ah[gate][ks], al[gate][ks]);
      }
  }
  for (int i = tid; i < 2 * 2 * NC * HP / 8; i += NTH) reinterpret_cast<gs_h16x8*>(&sB[0][0][0][0])[i] = (gs_h16x8)(gs_h16)0.f;  // h(-1) = 0

  // epilogue lane: column col (of tile nt), units ue .. ue + 3
  const int col = lane & 15, ue = u0 + 4 * (lane >> 4);
  float4 bh[3];
#pragma unroll
  for (int gate = 0; gate < 3; ++gate) bh[gate] = *reinterpret_cast<const float4*>(a.bhh[dir] + gate * Hg + ue);
  float hprev[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) hprev[nt][r] = 0.f;
  const float unscale = a.unscale[dir];
  const float* ihd = a.ih[dir];
  unsigned long long* exd = a.ex + (size_t)dir * 2 * 32 * Hg;
  __syncthreads();

  // Dead columns (>= B) duplicate column B - 1: same inputs, same zero start, so they carry the same values and their
  // (unconditional) stores rewrite what column B - 1 writes.
  int cl[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) cl[nt] = min(c0 + nt * 16 + col, a.B - 1);
  // x half of a (step, column tile), with b_ih: three gate quads
  auto load_x = [&](float4 (&x)[3], const int nt, const int s) {
    const int sc = min(s, a.F - 1);
    const int tt = dir ? a.F - 1 - sc : sc;
    const float* p = ihd + ((size_t)cl[nt] * a.F + tt) * (3 * Hg) + ue;
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) x[gate] = *reinterpret_cast<const float4*>(p + gate * Hg);
  };
  // Three x buffers per tile, two steps of lead: the rows of step s + 2 are requested during step s, behind the poll (an HBM round
  // trip: every step reads fresh rows; in front of the poll's loads -- in-order returns -- they cost 0.8 us per step).  (Two buffers
  // and one step of lead: 1719 against 1606 us for the postnet scan -- the register allocation of that variant shuffles freshly
  // loaded rows between accumulation registers, which waits for the loads.)
  float4 xa[NT][3], xb[NT][3], xc[NT][3];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { load_x(xa[nt], nt, 0); load_x(xb[nt], nt, 1); }

  // One PHASE = one step of one column tile.  The two tiles of a 32-utterance batch are independent sequences and alternate:
  // a tile's h(t) is published at the end of its phase and polled at the start of its NEXT phase, so the cross-workgroup
  // hand-off (2 of a step's 4.9 us when polled right behind the stores) rides under the other tile's products and epilogue.
  constexpr int CPI = NTH / UW;                      // columns per poll instruction (NTH lanes = CPI columns of one remote workgroup)
  constexpr int RQ = G > 1 ? (G - 1) * 16 / CPI : 1;  // granules per lane and phase
  unsigned long long vq[RQ];                          // the poll's loads: issued a phase ahead (below), consumed at the phase start
  auto poll_issue = [&](const int nt, const int s) {  // remote units of h(s-1) of tile nt: parity (s-1) & 1
    const unsigned long long* ep = exd + ((size_t)((s - 1) & 1) * 32 + c0 + nt * 16 + tid / UW) * Hg + (tid % UW);
#pragma unroll
    for (int rw = 0; rw < G - 1; ++rw)
#pragma unroll
      for (int cq = 0; cq < 16 / CPI; ++cq)
        vq[rw * (16 / CPI) + cq] = wp_get(ep + (size_t)(cq * CPI) * Hg + (rw < g ? rw : rw + 1) * UW);
  };
  auto phase = [&](auto NTI, const int s, float4 (&xg)[3], float4 (&xn)[3]) -> bool {
    constexpr int nt = decltype(NTI)::value;
    const int par = s & 1;
    if (G > 1 && s > 0 && !GS_DBG(a, 1)) {
      // ---- the other workgroups' units of h(t-1) of this tile: requested during the PREVIOUS phase (in front of that phase's
      //      stores: vector-memory operations retire in order, behind them every poll waited for their acknowledgements), so
      //      normally they are simply there; re-polled until all carry the tag otherwise ----
      const unsigned tag = (unsigned)s;
      // the first check stands OUTSIDE the retry loop: straight-line code lets the compiler wait for exactly these loads
      // (vmcnt = the stores issued behind them); inside a loop it waits for everything, i.e. for those stores' acknowledgements
      auto to_lds = [&]() {
#pragma unroll
        for (int rw = 0; rw < G - 1; ++rw)
#pragma unroll
          for (int cq = 0; cq < 16 / CPI; ++cq) {
            const unsigned w = (unsigned)vq[rw * (16 / CPI) + cq];
            const int c = nt * 16 + cq * CPI + tid / UW, u = (rw < g ? rw : rw + 1) * UW + tid % UW;
            sB[par ^ 1][0][c][u] = __builtin_bit_cast(gs_h16, (unsigned short)(w & 0xffffu));
            sB[par ^ 1][1][c][u] = __builtin_bit_cast(gs_h16, (unsigned short)(w >> 16));
          }
      };
      bool ok = true;
      {  // all tags in one xor / or chain (a chain of && compiled to nested exec-mask branches, wavernn_pipe16.h)
        unsigned stale_ = 0u;
#pragma unroll
        for (int q = 0; q < RQ; ++q) stale_ |= (unsigned)(vq[q] >> 32) ^ tag;
        ok = ok && stale_ == 0u;
      }
      if (ok) to_lds();  // (the use of the loaded values stays on the straight-line path too: behind the join the compiler waits for everything again)
      else {
        unsigned long long t0 = 0;
        for (int tries = 0;; ++tries) {
          __builtin_amdgcn_s_sleep(1);
          poll_issue(nt, s);
          ok = true;
          {  // all tags in one xor / or chain (a chain of && compiled to nested exec-mask branches, wavernn_pipe16.h)
            unsigned stale_ = 0u;
#pragma unroll
            for (int q = 0; q < RQ; ++q) stale_ |= (unsigned)(vq[q] >> 32) ^ tag;
            ok = ok && stale_ == 0u;
          }
          if (ok) break;
          if ((tries & 1023) == 1023 && wp_lost(tries, t0, a.abort_word)) return false;
        }
        to_lds();
      }
    }
    // behind the poll (vector-memory returns are in order: requested in front of it, these fresh HBM rows held its loads back)
    if (!GS_DBG(a, 8)) load_x(xn, nt, s + 2);
    __syncthreads();  // h(t-1) of this tile is complete in LDS (own units: the epilogue of its previous phase)
    // ---- W_hh . h(t-1): three products per (gate, k-step) ----
    f32x4 acc[3];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) acc[gate] = {0.f, 0.f, 0.f, 0.f};
    if (!GS_DBG(a, 4))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const gs_h16x8 bhi = *reinterpret_cast<const gs_h16x8*>(&sB[par ^ 1][0][nt * 16 + row][ks * 32 + kq * 8]);
      const gs_h16x8 blo = *reinterpret_cast<const gs_h16x8*>(&sB[par ^ 1][1][nt * 16 + row][ks * 32 + kq * 8]);
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[gate][ks], bhi, acc[gate], 0, 0, 0);
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[gate][ks], blo, acc[gate], 0, 0, 0);
#pragma unroll
      for (int gate = 0; gate < 3; ++gate) acc[gate] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[gate][ks], bhi, acc[gate], 0, 0, 0);
    }
    if (G > 1 && !GS_DBG(a, 1)) {  // the next phase's poll: the other tile's h of its last phase (NT = 2), published a phase ago
      if (NT == 2 && nt == 0) { if (s > 0) poll_issue(1, s); }
      else if (NT == 2) poll_issue(0, s + 1);
    }
    // ---- torch GRUCell (gate order r, z, n); the fp16 hi / lo halves of h(t) 2^10 -> LDS (own units) ----
    const int c = nt * 16 + col;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float xr = r == 0 ? xg[0].x : r == 1 ? xg[0].y : r == 2 ? xg[0].z : xg[0].w;
      const float xz = r == 0 ? xg[1].x : r == 1 ? xg[1].y : r == 2 ? xg[1].z : xg[1].w;
      const float xq = r == 0 ? xg[2].x : r == 1 ? xg[2].y : r == 2 ? xg[2].z : xg[2].w;
      const float br = r == 0 ? bh[0].x : r == 1 ? bh[0].y : r == 2 ? bh[0].z : bh[0].w;
      const float bz = r == 0 ? bh[1].x : r == 1 ? bh[1].y : r == 2 ? bh[1].z : bh[1].w;
      const float bn = r == 0 ? bh[2].x : r == 1 ? bh[2].y : r == 2 ? bh[2].z : bh[2].w;
      const float rg = gs_sigmoid(xr + (acc[0][r] * unscale + br));
      const float zg = gs_sigmoid(xz + (acc[1][r] * unscale + bz));
      const float ng = gs_tanh(xq + rg * (acc[2][r] * unscale + bn));
      const float hv = ng + zg * (hprev[nt][r] - ng);
      hprev[nt][r] = hv;
      const float x = hv * GS_HSCALE;
      const gs_h16 h = (gs_h16)x;
      sB[par][0][c][ue + r] = h;
      sB[par][1][c][ue + r] = (gs_h16)(x - (float)h);
    }
    if (G > 1 && !GS_DBG(a, 1)) {
      // ---- publish: the wave reads its 16 units x 16 columns back from LDS (its own writes: program order) so that ONE store
      //      instruction covers whole 128-byte lines (16 consecutive lanes = the wave's 16 units of a column); pieces of a line
      //      written by several instructions become visible one after the other (ppg_resident.h) ----
      unsigned long long* ew = exd + ((size_t)par * 32 + c0 + nt * 16) * Hg + u0 + (lane & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cc = q * 4 + (lane >> 4);
        const unsigned h = __builtin_bit_cast(unsigned short, sB[par][0][nt * 16 + cc][u0 + (lane & 15)]);
        const unsigned l = __builtin_bit_cast(unsigned short, sB[par][1][nt * 16 + cc][u0 + (lane & 15)]);
        wp_put_u(ew + (size_t)cc * Hg, h | (l << 16), (unsigned)(s + 1));
      }
    }
    // h(t) leaves time-major: a lane's 4 units are one 16-byte store, the wave's 16 units of a column one 64-byte segment.  (Written
    // straight into [B][2 Hg][F], every value was its own cache line: 4096 line transactions per step through ONE compute unit's
    // address path -- 1.5 of the encoder scan's 3.8 us per step.)
    if (!GS_DBG(a, 2)) {
      const int tt = dir ? a.F - 1 - s : s;
      *reinterpret_cast<float4*>(a.seq_tm + ((size_t)tt * a.B + cl[nt]) * (2 * Hg) + dir * Hg + ue) =
          make_float4(hprev[nt][0], hprev[nt][1], hprev[nt][2], hprev[nt][3]);
    }
    if (G > 1 && NT == 1 && !GS_DBG(a, 1)) poll_issue(0, s + 1);  // one tile: nothing to hide behind, first try right behind the stores
    return true;
  };
  auto step = [&](const int s, float4 (&xg)[NT][3], float4 (&xn)[NT][3]) -> bool {
    if (!phase(std::integral_constant<int, 0>{}, s, xg[0], xn[0])) return false;
    if (NT > 1 && !phase(std::integral_constant<int, NT - 1>{}, s, xg[NT - 1], xn[NT - 1])) return false;
    return true;
  };
  for (int s = 0; s < a.F; s += 3) {
    if (!step(s, xa, xc)) return;
    if (s + 1 < a.F && !step(s + 1, xb, xa)) return;
    if (s + 2 < a.F && !step(s + 2, xc, xb)) return;
  }
}

// [F][B][C] -> [B][C][F], 32 x 32 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void gru_scan_transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int F, int B, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (t < F && c < C) ? x[((size_t)t * B + b) * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    if (c < C && t < F) y[((size_t)b * C + c) * F + t] = tile[tx][ty + 8 * i];
  }
}

// 2^s with max |w| 2^s in [2^13, 2^14): the low halves of all but negligible weights are fp16 normals (conv1d.hip's rule)
static inline int gru_scan_scale_exp(const float* w, size_t n) {
  float wmax = 0.f;
  for (size_t i = 0; i < n; ++i) wmax = std::max(wmax, std::fabs(w[i]));
  if (!(wmax > 0.f) || !std::isfinite(wmax)) return 0;
  int e2;
  (void)std::frexp(wmax, &e2);  // wmax = m 2^e2, m in [0.5, 1)
  return std::max(-24, std::min(40, 14 - e2));  // clamped like mb_conv1d_pack's: a degenerate (denormal) W_hh must not scale to inf
}

static inline bool gru_scan_shape_ok(int B, int Hg) { return B >= 1 && B <= 32 && (Hg == 128 || Hg == 256); }

// Devices on which a resident scan lost a hand-off (or failed to launch): later calls go straight to the launch-per-step scan
// instead of spinning 0.2 s again (the memo wavernn.hip / ppg2mel.hip keep for their resident loops).  Bit d = device d.
static std::atomic<unsigned long long> g_gru_scan_failed{0};
static inline bool gru_scan_device_failed() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  return (g_gru_scan_failed.load(std::memory_order_relaxed) >> dev) & 1ull;
}
static inline void gru_scan_mark_failed() {
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) g_gru_scan_failed.fetch_or(1ull << dev, std::memory_order_relaxed);
}

template <int KS, int NT, int NW>
static int gru_scan_launch_inst(const GruScanK& k, hipStream_t s, int groups = 1) {
  const size_t lds = (size_t)2 * 2 * (NT * 16) * (KS * 32 + 8) * sizeof(gs_h16);
  // the attribute belongs to the function ON THE CURRENT DEVICE: tracked per device (a process-wide flag left a second
  // device's first launch without it), atomically (two host threads may encode at once; setting it twice is harmless)
  static std::atomic<unsigned long long> attr_done{0};  // bit d = set on device d
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if (!bit || !(attr_done.load(std::memory_order_acquire) & bit)) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_scan_kernel<KS, NT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((gru_scan_kernel<KS, NT, NW>), dim3(2 * (KS * 32) / (NW * 16), groups), dim3(NW * 64), lds, s, k);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

static int gru_scan_launch(const GruScanK& k, hipStream_t s) {
  const int nt = (k.B + 15) / 16;
  // 17..32 utterances: two column tiles.  As two PHASES of one workgroup set (NT = 2: a tile's hand-off rides under the other tile's
  // products) or as two workgroup sets side by side (NT = 1, grid.y = 2: the scan uses 4 / 16 of 256 compute units instead of 2 / 8;
  // MBHIP_DIAG=gs_split=0 keeps the phases) -- the sequences of different utterances never meet
  const bool split = nt == 2 && diag_int("gs_split", 1) != 0;
  if (k.Hg == 128) return nt == 1 ? gru_scan_launch_inst<4, 1, 8>(k, s) : split ? gru_scan_launch_inst<4, 1, 8>(k, s, 2) : gru_scan_launch_inst<4, 2, 8>(k, s);
  return nt == 1 ? gru_scan_launch_inst<8, 1, 4>(k, s) : split ? gru_scan_launch_inst<8, 1, 4>(k, s, 2) : gru_scan_launch_inst<8, 2, 4>(k, s);
}

}  // namespace mb
