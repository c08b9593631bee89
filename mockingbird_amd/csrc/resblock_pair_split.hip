// Round 6: the fused ResBlock unit of the GAN vocoders at the REFERENCE's precision on time-major activations:
//
//   y = (acc ? y : 0) + scale * ( x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2 )     conv1: k taps, dilation d; conv2: k taps, dilation 1
//
// = one (convs1[i], convs2[i]) iteration of ResBlock1.forward
//   models/vocoder/hifigan/models.py:39-46, models/vocoder/fregan/generator.py:43-50
// (and with out_scale / accumulate the "xs += resblock(x); x = xs / num_kernels" of Generator.forward, models.py:139-145).
//
// the fp32-result path ran every wide ResBlock conv as its own conv1d_split_kernel launch on channel-major
// [B][C][T] fp32 tensors -- the waves that issue the MFMAs also staged the window with 32 scalar loads per thread per chunk,
// applied the activation, split and transposed it (5 302 vector instructions beside 672 MFMAs per wave, matrix pipe 31 % busy).
// This kernel is resblock_f16.hip's launch structure (4 MMA waves that touch only LDS and their weight stream, 4 support waves
// that own all HBM traffic, the intermediate h in LDS, persistent workgroups) on error-compensated operands:
//
//   * HBM layout: fp32 TIME-major [B][T][C] in and out -- a (window rows x 32 channels) chunk is 128-byte row segments, every
//     load 16 bytes, no transposition anywhere; the tensor keeps all 24 bits (the residual chain x <- x + conv2(..) is fp32).
//   * The SUPPORT waves lay lrelu(x) down as fp16 hi / scaled-lo planes (common.h split_pair: 4 vector instructions per value, on
//     waves that issue no MFMA), and add the residual / write y out from an fp32 LDS tile while the MMA waves are in the next tile.
//   * Arithmetic = conv1d.hip's split scheme: w 2^s = wh + wl (s per conv), a = ah + 2^-11 al,
//     acc += wl.ah + (wh 2^-11).al + wh.ah, fp32 accumulate, v = acc 2^-s + bias.  The third weight image wh 2^-11 is made in
//     registers (v_pk_mul_f16, exact or the same round-to-nearest subnormal the host image held): the weight stream is 2 fragments
//     per (tap, k-step, tile) instead of 3.
//
// Per k-step a wave issues 3 MT NTW MFMAs for 2 MT weight fragments (L2) and 2 NTW B fragments (LDS): three times the matrix
// work of the fp16 kernel per staged byte.
#include <atomic>
#include <cmath>
#include <type_traits>
#include "common.h"
#include "split_tm.h"

namespace mb {

struct ResPairSK {
  const float* x; float* y; const h16* w; const float* b1; const float* b2;
  long long bstride;  // floats per batch item (T * C)
  int T, ntaps, dil;
  int NB, tiles_per_item, n_tiles, x_rows;
  int t_begin, t_end;  // output rows [t_begin, t_end) of every item belong to this launch (the window still reads the whole item)
  int nbuf;           // LDS buffers of the x-window chunk ring (1 = single buffer: one chunk per tile, refilled while phase 2 runs)
  float slope, out_scale;
  float us1, us2;     // 2^-s of conv1 / conv2 (mb_resblock_pair_split_pack)
  int accumulate;
  const int* valid; int valid_mul;  // ragged batches: item b has valid[b] * valid_mul positions (null: T)
  unsigned* range_events;           // MBHIP_CONV_RANGE_CHECK=1: staged values beyond fp16's range are counted here (conv1d.hip)
  unsigned long long* trace;        // diagnostics builds only (-DSPAIR_TRACE_BUILD, MBHIP_DIAG=spair_trace=<file>): shader-clock marks of workgroup 0
};

// NL = support waves per workgroup (beside the 4 MMA waves): 4 (one per SIMD, 256 registers each) or, where the MMA waves need <= 168
// registers (<= 64 channels), 8 (two per SIMD) -- at those widths the support waves' instruction stream, not the MFMAs, sets the pace
#ifndef SPAIR_DUAL
#define SPAIR_DUAL 1  // (A/B builds: 0 = the third weight image made in registers for every instance)
#endif
#ifndef SPAIR_NL_NARROW
#define SPAIR_NL_NARROW 8  // (A/B builds: 4)
#endif
constexpr int SPAIR_MAX_HALO = 80;  // (ksize - 1) * dilation the support waves' window registers are sized for (k = 11, d = 7: 70)
// diagnostics builds only (tools/build_variant.sh ... -DSPAIR_DBG=<bits>; results are wrong, timings isolate one cost each):
// 1 = weight ring never refilled, 2 = B fragments read once per chunk, 4 = no epilogues, 8 = no window fill, 16 = no write-out, 32 = no MFMAs
// mark k of tile `it`, role 0 = MMA wave 0, 1 = support wave 4 (workgroup 0, first 4 tiles, 64 marks each)
#ifdef SPAIR_TRACE_BUILD
#define SP_MARK(role, it, k)                                                                    \
  do {                                                                                          \
    if (a.trace && blockIdx.x == 0 && (it) < 4 && (tid & 63) == 0 && wave == ((role) ? 4 : 0))  \
      a.trace[((role) * 4 + (it)) * 64 + (k)] = (unsigned long long)clock64();                  \
  } while (0)
#else  // (a conditional global store in front of an MFMA loop makes the compiler's vmcnt bookkeeping give up: never in the product)
#define SP_MARK(role, it, k) do { } while (0)
#endif

// valid length of batch item b: a SCALAR load with its own wait (resblock_f16.hip pair_valid_len: a vector-memory load at the head
// of every tile would sit in front of the weight ring)
__device__ __forceinline__ int spair_valid_len(const ResPairSK& a, int b) {
  if (!a.valid) return a.T;
  int v;
  const int* p = a.valid + b;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return min(a.T, v * a.valid_mul);
}

template <int C, int MT_, int WN_, int NTW_> struct SPairGeom {
  static constexpr int CH = C;
  static constexpr int CT = C < 32 ? 32 : C;       // channel rows of the MFMA tiles (C = 16: the upper half has zero weights)
  static constexpr int MT = MT_, WN = WN_, NTW = NTW_, WM = 4 / WN_;
  static_assert(WM * MT * 32 == CT, "the four MMA waves cover all output channels");
  static constexpr int CK = C >= 32 ? 32 : 16;     // input channels per x chunk
  static constexpr int KB = CK / 16;               // k-steps per tap per chunk
  static constexpr int NCH = C / CK;
  static_assert(NCH == 1 || NCH % 2 == 0, "chunks run in pairs (the ring slot of a chunk's first tap alternates)");
  static constexpr int CKP = CK + 8, CP = C + 8;   // LDS row strides in halves (odd multiples of 16 B: conflict-free ds_read_b128)
  static constexpr int CPF = C + 4;                // fp32 y tile row stride in floats
  static constexpr int N1 = WN * NTW * 32;
};

// YS = y staged in its own fp32 LDS tile (the support waves then have the whole tile time for the x prefetch and the write-out);
// without it y is staged over the h planes and must leave before the next h is written.
template <int C, int MT_, int WN_, int NTW_, bool YS, int NL>
__global__ __launch_bounds__(64 * (4 + NL)) __attribute__((amdgpu_waves_per_eu(1 + NL / 4, 1 + NL / 4)))
void resblock_pair_split_kernel(ResPairSK a) {
  using G = SPairGeom<C, MT_, WN_, NTW_>;
  constexpr int CK = G::CK, KB = G::KB, NCH = G::NCH, MT = G::MT, WN = G::WN, NTW = G::NTW;
  constexpr int CKP = G::CKP, CP = G::CP, CPF = G::CPF, N1 = G::N1;
  constexpr int TD = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int XPL = a.x_rows * CKP;             // halves per plane of an x chunk buffer
  const int HR = N1 + a.ntaps - 1;            // rows of an h plane
  const int HPL = HR * CP;                    // halves per h plane
  h16* xs = reinterpret_cast<h16*>(lds_raw);  // [nbuf][hi | lo][x_rows][CKP]
  h16* hs = xs + a.nbuf * 2 * XPL;            // [hi | lo][HR][CP]
  float* ys = reinterpret_cast<float*>(YS ? hs + 2 * HPL : hs);  // [N1][CPF] staged conv2 2^-s + b2 (fp32)
  float* bs = reinterpret_cast<float*>(hs + 2 * HPL) + (YS ? N1 * CPF : 0);  // [2][C]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = a.ntaps;
  const int p2 = (ntaps - 1) >> 1, p1 = p2 * a.dil;
  const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int njobs = my_tiles * NCH;

  for (int i = tid; i < 2 * C; i += 64 * (4 + NL)) bs[i] = i < C ? a.b1[i] : a.b2[i - C];
  __syncthreads();  // Z (resblock_f16.hip: the epilogues read bs behind their own barriers, the fill must be fenced once)

  if (wave >= 4) {
    // ------------------------------ support waves ------------------------------
    // A software pipeline across the tile's barriers: every HBM load is ISSUED at the head of one barrier interval and its result used
    // in the NEXT interval (two register sets alternate), so an interval costs the support waves their vector / LDS work only,
    // never a memory round trip.  Measured on the way (256 channels, k = 3, shader-clock marks of workgroup 0): load -> wait -> store
    // inside one interval made the support waves the pole of phase 1 (7.1 k cycles per chunk interval against 4.3 k of MFMAs: the MMA
    // waves waited 25 k of a tile's 102 k cycles at the chunk barriers); issuing the loads at the END of the previous interval
    // changed nothing (the support waves arrive last, so "one interval ahead" was a few cycles ahead).
#ifndef SPAIR_SUPPORT_PRIO
#define SPAIR_SUPPORT_PRIO 2
#endif
    // the support wave of a SIMD is the YOUNGER of its two waves and loses every vector-issue arbitration to the MMA wave
    // (MI355X_MICROARCH.md "Two waves per SIMD": priority, then age); its ~250 instructions per interval then took 6 k cycles.  An MFMA needs
    // one issue slot per 32 cycles, so the static priority costs the MMA wave next to nothing.
    __builtin_amdgcn_s_setprio(SPAIR_SUPPORT_PRIO);
    constexpr int PPR = CK / 4;                            // 16-byte pieces (4 floats) per row of a chunk
    constexpr int NSL = 64 * NL;                           // support lanes
    constexpr int LBX = ((N1 + SPAIR_MAX_HALO) * PPR + NSL - 1) / NSL;  // window pieces per lane (one chunk)
    constexpr int YPR = C / 4;                             // 16-byte pieces per output row
    constexpr int NP = NCH == 1 ? 2 : NCH;                 // write-out parts per tile
    constexpr int WB = (N1 * YPR / NP + NSL - 1) / NSL;    // output pieces per lane per part
    const float slope = a.slope;
    const int total = a.x_rows * PPR;
    const int ltid = tid - 256;
    // per-lane constants of the window pieces: piece i of a lane is (row, 4-channel group) = fixed for every job
    int xrow[LBX];
    unsigned xcol[LBX], xlds[LBX];
#pragma unroll
    for (int i = 0; i < LBX; ++i) {
      const int idx = min(i * NSL + ltid, total - 1);
      xrow[i] = idx / PPR;
      xcol[i] = (unsigned)(idx - xrow[i] * PPR) * 4u;
      xlds[i] = (unsigned)xrow[i] * CKP + xcol[i];
    }
    // window of job q (tile q / NCH, chunk q % NCH): clamped addresses (the load is always legal), the value selected
    auto issue_x = [&](int q, f32x4 (&vx)[LBX]) __attribute__((always_inline)) {
      if (SPAIR_DBG & 8) return;
      const int tile = (int)blockIdx.x + (q / NCH) * (int)gridDim.x, c = q % NCH;
      const int b = tile / a.tiles_per_item, t0 = a.t_begin + (tile - b * a.tiles_per_item) * a.NB;
      const int Tb = spair_valid_len(a, b);  // beyond: this item's zero padding
      const int tx0 = t0 - p2 - p1;
      const float* xb = a.x + (long long)b * a.bstride + c * CK;
      const __amdgpu_buffer_rsrc_t rs = tm_rsrc(xb, (long long)Tb * C * 4 - c * CK * 4);  // rows outside [0, Tb): zeros from the range check
#pragma unroll
      for (int i = 0; i < LBX; ++i) vx[i] = tm_load16(rs, (unsigned)((tx0 + xrow[i]) * C + (int)xcol[i]) * 4u);
    };
    auto commit_x = [&](int q, const f32x4 (&vx)[LBX]) __attribute__((always_inline)) {  // lrelu -> hi / scaled lo planes of buffer q % nbuf
      if (SPAIR_DBG & 8) return;
      h16* buf = xs + (q % a.nbuf) * 2 * XPL;
#pragma unroll
      for (int i = 0; i < LBX; ++i) {
        float l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = fmaxf(vx[i][e], vx[i][e] * slope);  // leaky_relu, 0 < slope < 1 (bit for bit x > 0 ? x : slope x)
        mb_h2 h0, l0, h1, l1;
        split_pair(l[0], l[1], h0, l0);
        split_pair(l[2], l[3], h1, l1);
        const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        // (pieces past the window repeat its last piece: the same halves to the same place, no predicate)
        *reinterpret_cast<h16x4*>(buf + xlds[i]) = hi;
        *reinterpret_cast<h16x4*>(buf + XPL + xlds[i]) = lo;
      }
      if (a.range_events) {  // diagnostics only (MBHIP_CONV_RANGE_CHECK=1, uniform branch): values the split saturates (NaN / Inf too)
        int n_out = 0;
#pragma unroll
        for (int i = 0; i < LBX; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) n_out += (i * NSL + ltid < total && !(fabsf(vx[i][e]) <= 65504.f)) ? 1 : 0;  // (|lrelu(x)| <= |x|: x beyond the range <=> flagged; slope x is not)
        if (n_out) atomicAdd(a.range_events, (unsigned)n_out);
      }
    };
    // y write-out of a finished tile in NP parts: the MMA waves leave conv2 2^-s + b2 in ys (fp32); the residual x (and the running sum
    // when accumulating) is requested a barrier interval ahead (issue_w), added here and the rows leave as coalesced 16-byte stores
    struct WTile { const float* xb; float* yb; int ytotal; };
    auto wtile = [&](int it) __attribute__((always_inline)) {
      const int tile = (int)blockIdx.x + it * (int)gridDim.x;
      const int b = tile / a.tiles_per_item, t0 = a.t_begin + (tile - b * a.tiles_per_item) * a.NB;
      const int Tb = spair_valid_len(a, b);
      const int rows = max(0, min(a.NB, min(Tb, a.t_end) - t0));
      return WTile{a.x + (long long)b * a.bstride + (long long)t0 * C, a.y + (long long)b * a.bstride + (long long)t0 * C, rows * YPR};
    };
    auto issue_w = [&](int it, int part, f32x4 (&rx)[WB], f32x4 (&ry)[WB]) __attribute__((always_inline)) {
      if (SPAIR_DBG & 16) return;
      const WTile w = wtile(it);
      if (w.ytotal <= 0) return;  // (uniform: a tile beyond its item's valid length)
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const unsigned idx = (unsigned)min(part * (WB * NSL) + i * NSL + ltid, w.ytotal - 1);  // pieces past the tile repeat its last piece
        rx[i] = *reinterpret_cast<const f32x4*>(w.xb + idx * 4u);
        if (a.accumulate) ry[i] = *reinterpret_cast<const f32x4*>(w.yb + idx * 4u);
      }
    };
    auto commit_w = [&](int it, int part, const f32x4 (&rx)[WB], const f32x4 (&ry)[WB]) __attribute__((always_inline)) {
      if (SPAIR_DBG & 16) return;
      const WTile w = wtile(it);
      if (w.ytotal <= 0) return;
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const int idx = part * (WB * NSL) + i * NSL + ltid;
        const unsigned idc = (unsigned)min(idx, w.ytotal - 1);
        const unsigned row = idc / YPR, pc = idc - row * YPR;
        const f32x4 hv = *reinterpret_cast<const f32x4*>(ys + row * CPF + pc * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float f = (rx[i][e] + hv[e]) * a.out_scale;
          if (a.accumulate) f += ry[i][e];
          o[e] = f;
        }
        if (idx < w.ytotal) *reinterpret_cast<f32x4*>(w.yb + idc * 4u) = o;  // (a repeated piece would accumulate twice)
      }
    };
    f32x4 vxA[LBX], rxA[WB], ryA[WB], rxB[WB], ryB[WB];
    // Behind a barrier everything a support wave has outstanding is an interval old.  Draining it THERE, explicitly, is free and tells
    // the compiler that the register set it is about to lay down has landed: the set requested next stays in flight through the whole
    // interval's processing (left to its own counting, the compiler waited for the newest load at the first use of the oldest).
#define SP_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)
    // Barrier schedule per tile (must mirror the MMA waves'): B per chunk, [W], E1, P|YF, Y.
    if (NCH == 1) {
      // one chunk per tile, one buffer: the next tile's window is requested at the head of phase 1 and laid down while the MMA waves
      // run phase 2, which reads only h.  The previous tile's y leaves in two parts (register sets A, B).  With its own y tile: part
      // 0 in phase 1, part 1 in phase 2, each requested one interval earlier; over the h planes both parts leave before W and are
      // requested at the head of the previous tile's phase 2.
      if (my_tiles > 0) { issue_x(0, vxA); commit_x(0, vxA); }
      for (int it = 0; it < my_tiles; ++it) {
        SP_MARK(1, it, 0);
        __syncthreads();  // B
        SP_MARK(1, it, 1);
        SP_DRAIN();
        if (it + 1 < my_tiles) issue_x(it + 1, vxA);
        if (it > 0) {
          if (YS) issue_w(it - 1, 1, rxB, ryB);
          commit_w(it - 1, 0, rxA, ryA);
          if (!YS) commit_w(it - 1, 1, rxB, ryB);
        }
        SP_MARK(1, it, 32);
        if (!YS) __syncthreads();  // W: the h planes are free for h of this tile
        __syncthreads();  // E1: phase 1 has finished reading xs
        SP_MARK(1, it, 33);
        SP_DRAIN();
        issue_w(it, 0, rxA, ryA);
        if (!YS) issue_w(it, 1, rxB, ryB);
        if (it + 1 < my_tiles) commit_x(it + 1, vxA);
        if (YS && it > 0) commit_w(it - 1, 1, rxB, ryB);
        SP_MARK(1, it, 34);
        __syncthreads();  // P (all MMA waves done with h) | YF (ys is free for y of this tile)
        SP_MARK(1, it, 35);
        __syncthreads();  // Y: y of this tile is staged
        SP_MARK(1, it, 36);
      }
      if (my_tiles > 0) {
        if (YS) issue_w(my_tiles - 1, 1, rxB, ryB);
        commit_w(my_tiles - 1, 0, rxA, ryA);
        commit_w(my_tiles - 1, 1, rxB, ryB);
      }
    } else {
      // chunk ring: at the head of interval q (behind B_q) job q + nbuf is requested into the register set of q's parity; the job
      // requested an interval earlier (q + nbuf - 1, the other set) is laid down into the buffer job q - 1 just left.  The previous
      // tile's y leaves in NCH parts the same way (part c behind B_c, requested behind B_{c-1}; part 0 behind E1 of its own tile).
      f32x4 vxB[LBX];
      for (int q = 0; q < a.nbuf - 1 && q < njobs; ++q) { issue_x(q, vxA); commit_x(q, vxA); }
      if (a.nbuf - 1 < njobs) issue_x(a.nbuf - 1, vxB);
      for (int it = 0; it < my_tiles; ++it) {
        for (int c = 0; c < NCH; c += 2) {
          const int q = it * NCH + c;
          SP_MARK(1, it, 2 * c);
          __syncthreads();  // B_q: job q is staged, the buffer of job q-1 is free
          SP_MARK(1, it, 2 * c + 1);
          SP_DRAIN();
          if (q + a.nbuf < njobs) issue_x(q + a.nbuf, vxA);
          if (it > 0) issue_w(it - 1, c + 1, rxA, ryA);
          if (q + a.nbuf - 1 < njobs) commit_x(q + a.nbuf - 1, vxB);
          if (it > 0) commit_w(it - 1, c, rxB, ryB);
          SP_MARK(1, it, 2 * c + 2);
          __syncthreads();  // B_{q+1}
          SP_MARK(1, it, 2 * c + 3);
          SP_DRAIN();
          if (q + 1 + a.nbuf < njobs) issue_x(q + 1 + a.nbuf, vxB);
          if (it > 0 && c + 2 < NCH) issue_w(it - 1, c + 2, rxB, ryB);
          if (q + a.nbuf < njobs) commit_x(q + a.nbuf, vxA);
          if (it > 0) commit_w(it - 1, c + 1, rxA, ryA);
        }
        SP_MARK(1, it, 32);
        if (!YS) __syncthreads();  // W
        SP_MARK(1, it, 33);
        __syncthreads();  // E1
        SP_MARK(1, it, 34);
        issue_w(it, 0, rxB, ryB);
        __syncthreads();  // P | YF
        SP_MARK(1, it, 35);
        __syncthreads();  // Y
        SP_MARK(1, it, 36);
      }
      if (my_tiles > 0) {
        commit_w(my_tiles - 1, 0, rxB, ryB);
        for (int c = 1; c < NCH; ++c) { issue_w(my_tiles - 1, c, rxA, ryA); commit_w(my_tiles - 1, c, rxA, ryA); }
      }
    }
    return;
  }


  // ------------------------------ MMA waves ------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int mt0 = wm * MT;
  const int NFT = 2 * NCH * ntaps;  // flat taps of the circular weight stream
  const h16x8* wp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
    wp[i] = reinterpret_cast<const h16x8*>(a.w) + (size_t)(mt0 + i) * NFT * KB * 2 * 64 + lane;

  h16x8 ring[TD][KB][MT][2];  // [slot][k-step][tile][hi | lo]
#pragma unroll
  for (int s = 0; s < TD; ++s)
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring[s][u][i][p] = wp[i][((size_t)(s * KB + u) * 2 + p) * 64];
  int ftn = TD;  // next flat tap to prefetch
  auto sp_wrap = []() {};

  constexpr bool DUAL = MT == 1 && SPAIR_DUAL != 0;  // split_tm.h: a second accumulator set instead of the third weight image
  f32x16 acc[MT][NTW], acl[MT][NTW];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[i][n][q] = 0.f; if (DUAL) acl[i][n][q] = 0.f; }
  };
  auto acc_at = [&](int i, int n, int q) __attribute__((always_inline)) { return DUAL ? fmaf(acl[i][n][q], 1.f / 2048.f, acc[i][n][q]) : acc[i][n][q]; };
  const h16 k2m11 = (h16)(1.f / 2048.f);


  const int lrow = wn * (NTW * 32) + (lane & 31);  // this lane's row inside an N tile group
  const int lcol = (lane >> 5) * 8;
  const int x_tapstep = a.dil * CKP;
  const float slope = a.slope, us1 = a.us1, us2 = a.us2;
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = (int)blockIdx.x + it * (int)gridDim.x;
    const int t0 = a.t_begin + (tile % a.tiles_per_item) * a.NB;
    const int Tb = spair_valid_len(a, tile / a.tiles_per_item);
    // ---------------- phase 1: h = lrelu(conv1(lrelu(x)) 2^-s + b1) ----------------
#ifdef SPAIR_TRACE_BUILD
    if (a.trace && blockIdx.x == 0 && it < 4 && tid == 0) a.trace[(0 * 4 + it) * 64 + 62] = (unsigned long long)wall_clock64();
#endif
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): a known scoreboard at the head of the tile keeps the compiler's counted waits exact
    zero_acc();
    if (NCH == 1) {
      SP_MARK(0, it, 0);
      __syncthreads();  // B
      SP_MARK(0, it, 1);
      SP_CHUNK(0, xs + ((it * NCH) % a.nbuf) * 2 * XPL + lrow * CKP + lcol, CKP, x_tapstep, XPL);
    } else {
      for (int c = 0; c < NCH; c += 2) {
        SP_MARK(0, it, 2 * c);
        __syncthreads();  // B
        SP_MARK(0, it, 2 * c + 1);
        SP_CHUNK(0, xs + ((it * NCH + c) % a.nbuf) * 2 * XPL + lrow * CKP + lcol, CKP, x_tapstep, XPL);
        SP_MARK(0, it, 2 * c + 2);
        __syncthreads();  // B
        SP_MARK(0, it, 2 * c + 3);
        SP_CHUNK(1, xs + ((it * NCH + c + 1) % a.nbuf) * 2 * XPL + lrow * CKP + lcol, CKP, x_tapstep, XPL);
      }
    }
    SP_MARK(0, it, 32);
    if (!YS) __syncthreads();  // W: the support waves have written out the previous tile's y from the h planes
    SP_MARK(0, it, 33);
    if (!(SPAIR_DBG & 4)) {  // epilogue 1 -> h planes (hi / scaled lo); rows outside [0, Tb) are conv2's zero padding
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int row = lrow + n * 32;
          const int th = t0 - p2 + row;
          const bool inside = th >= 0 && th < Tb;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co0 = (mt0 + i) * 32 + 8 * g + 4 * (lane >> 5);
            if (co0 >= C) continue;  // only C = 16: rows 16..31 of the tile are padding
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + co0);
            float hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = fmaf(acc_at(i, n, 4 * g + e), us1, bq[e]);
              v = fmaxf(v, v * slope);
              hv[e] = inside ? v : 0.f;
            }
            mb_h2 h0, l0, h1, l1;
            split_pair(hv[0], hv[1], h0, l0);
            split_pair(hv[2], hv[3], h1, l1);
            const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
            *reinterpret_cast<h16x4*>(hs + row * CP + co0) = hi;
            *reinterpret_cast<h16x4*>(hs + HPL + row * CP + co0) = lo;
          }
        }
    }
    SP_MARK(0, it, 34);
    __syncthreads();  // E1
    SP_MARK(0, it, 35);
    // ---------------- phase 2: conv2(h) 2^-s + b2 ----------------
    zero_acc();
    if (NCH == 1) {
      SP_CHUNK(1, hs + lrow * CP + lcol, CP, CP, HPL);
    } else {
      for (int c = 0; c < NCH; c += 2) {
        SP_CHUNK(0, hs + lrow * CP + c * CK + lcol, CP, CP, HPL);
        SP_CHUNK(1, hs + lrow * CP + (c + 1) * CK + lcol, CP, CP, HPL);
      }
    }
    SP_MARK(0, it, 36);
    __syncthreads();  // P: every MMA wave has finished reading h | YF: the previous tile's y has left ys
    SP_MARK(0, it, 37);
    if (!(SPAIR_DBG & 4)) {  // epilogue 2 -> ys (fp32); the support waves add the residual and write y out
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int row = lrow + n * 32;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co0 = (mt0 + i) * 32 + 8 * g + 4 * (lane >> 5);
            if (co0 >= C) continue;
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + C + co0);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(acc_at(i, n, 4 * g + e), us2, bq[e]);
            *reinterpret_cast<f32x4*>(ys + row * CPF + co0) = v;
          }
        }
    }
    SP_MARK(0, it, 38);
    __syncthreads();  // Y
    SP_MARK(0, it, 39);
#ifdef SPAIR_TRACE_BUILD
    if (a.trace && blockIdx.x == 0 && it < 4 && tid == 0) a.trace[(0 * 4 + it) * 64 + 63] = (unsigned long long)wall_clock64();
#endif
  }
}

// LDS bytes of an instance for a given conv1 geometry and x-chunk buffer count
template <class G>
static size_t spair_lds_bytes(int ntaps, int dil, int nbuf, bool ys) {
  const int x_rows = G::N1 + (ntaps - 1) * dil;
  const int hr = G::N1 + ntaps - 1;
  return ((size_t)nbuf * 2 * x_rows * G::CKP + (size_t)2 * hr * G::CP) * sizeof(h16) + (ys ? (size_t)G::N1 * G::CPF * sizeof(float) : 0) +
         2 * G::CH * sizeof(float);
}
constexpr size_t SPAIR_LDS_CAP = 160 * 1024;
template <class G>
static int spair_min_nbuf() { return G::NCH == 1 ? 1 : 2; }
template <class G>
static bool spair_fits(int ntaps, int dil, bool ys) { return spair_lds_bytes<G>(ntaps, dil, spair_min_nbuf<G>(), ys) <= SPAIR_LDS_CAP; }

static int spair_cus() {
  static std::atomic<int> cached{0};
  int n = cached.load(std::memory_order_relaxed);
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cached.store(n, std::memory_order_relaxed);
  }
  return n;
}

template <int C, int MT, int WN, int NTW, bool YS, int NL>
static int launch_spair(ResPairSK k, int batch, hipStream_t s) {
  using G = SPairGeom<C, MT, WN, NTW>;
  k.NB = G::N1 - (k.ntaps - 1);
  k.x_rows = G::N1 + (k.ntaps - 1) * k.dil;
  if (k.t_end <= 0) k.t_end = k.T;
  k.tiles_per_item = cdiv(k.t_end - k.t_begin, k.NB);
  k.n_tiles = k.tiles_per_item * batch;
  int nbuf = spair_min_nbuf<G>();
  if (nbuf > 1)  // as many x-chunk buffers as fit (<= 4): the support waves run nbuf - 1 chunks ahead of the MMA waves
    while (nbuf < 4 && spair_lds_bytes<G>(k.ntaps, k.dil, nbuf + 1, YS) <= SPAIR_LDS_CAP) ++nbuf;
  k.nbuf = nbuf;
  const size_t lds = spair_lds_bytes<G>(k.ntaps, k.dil, nbuf, YS);
  static std::atomic<unsigned long long> attr_done{0};  // bit d = attribute set on device d (per kernel instance)
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if (!bit || !(attr_done.load(std::memory_order_acquire) & bit)) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_split_kernel<C, MT, WN, NTW, YS, NL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int grid = std::min(k.n_tiles, spair_cus());
#ifdef SPAIR_TRACE_BUILD
  static unsigned long long* d_trace = nullptr;
  std::string trace_file;
  const char* trace_path = diag_str("spair_trace", &trace_file) ? trace_file.c_str() : nullptr;
  if (trace_path) {
    if (!d_trace) MB_HIP(hipMalloc((void**)&d_trace, 512 * sizeof(unsigned long long)));
    MB_HIP(hipMemsetAsync(d_trace, 0, 512 * sizeof(unsigned long long), s));
    k.trace = d_trace;
  }
#endif
  hipLaunchKernelGGL((resblock_pair_split_kernel<C, MT, WN, NTW, YS, NL>), dim3(grid), dim3(64 * (4 + NL)), lds, s, k);
  MB_HIP(hipGetLastError());
#ifdef SPAIR_TRACE_BUILD
  if (trace_path) {  // append "C MT WN NTW YS ntaps dil nbuf tiles : marks..." per launch
    unsigned long long h[512];
    MB_HIP(hipStreamSynchronize(s));
    MB_HIP(hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "a")) {
      fprintf(f, "%d %d %d %d %d %d %d %d %d :", C, MT, WN, NTW, (int)YS, k.ntaps, k.dil, k.nbuf, k.n_tiles);
      for (int i = 0; i < 512; ++i) fprintf(f, " %llu", h[i]);
      fprintf(f, "\n");
      fclose(f);
    }
  }
#endif
  return MB_OK;
}

// the instance table: per channel count the candidates (MT, WN, NTW, YS), preferred first
using SG256a = SPairGeom<256, 2, 1, 3>;
using SG256b = SPairGeom<256, 2, 1, 2>;
using SG128a = SPairGeom<128, 1, 1, 4>;
using SG128b = SPairGeom<128, 1, 1, 3>;
using SG64a = SPairGeom<64, 1, 2, 2>;  // (192-row tiles at 64 channels / 384-row tiles at 32 measured no faster and need > 256 registers
using SG32a = SPairGeom<32, 1, 4, 2>;  //  in the support waves: two window sets + two residual sets)
using SG32b = SPairGeom<32, 1, 4, 1>;
using SG16a = SPairGeom<16, 1, 4, 2>;
using SG16b = SPairGeom<16, 1, 4, 1>;

}  // namespace mb

using namespace mb;

extern "C" int mb_resblock_pair_split_supported(int channels, int ksize, int dilation) {
  if (!(channels == 16 || channels == 32 || channels == 64 || channels == 128 || channels == 256)) return 0;
  if (ksize < 3 || (ksize & 1) == 0 || dilation < 1 || (ksize - 1) * dilation > SPAIR_MAX_HALO) return 0;
  switch (channels) {
    case 256: return spair_fits<SG256a>(ksize, dilation, false) || spair_fits<SG256b>(ksize, dilation, false);
    case 128: return spair_fits<SG128a>(ksize, dilation, false) || spair_fits<SG128b>(ksize, dilation, false);
    case 64: return spair_fits<SG64a>(ksize, dilation, true) || spair_fits<SG64a>(ksize, dilation, false);
    case 32: return spair_fits<SG32a>(ksize, dilation, true) || spair_fits<SG32b>(ksize, dilation, true);
    default: return spair_fits<SG16a>(ksize, dilation, true) || spair_fits<SG16b>(ksize, dilation, true);
  }
}

extern "C" size_t mb_resblock_pair_split_packed_halves(int channels, int ksize) {
  // per 32-row output tile: 2 convs x (channels / 16) k-blocks x ksize taps x {hi, lo}, 512 halves per fragment
  const int mtt = (channels + 31) / 32;
  return (size_t)mtt * 2 * (channels / 16) * ksize * 2 * 512;
}

// h_w1 / h_w2: fp32 torch Conv1d weights [C][C][k] (weight norm already folded).  h_unscale[2] receives 2^-s of conv1 / conv2.
extern "C" int mb_resblock_pair_split_pack(const float* h_w1, const float* h_w2, int channels, int ksize, uint16_t* h_packed,
                                           float* h_unscale) {
  MB_REQUIRE(h_w1 && h_w2 && h_packed && h_unscale, "resblock_pair_split_pack: null pointer");
  MB_REQUIRE(mb_resblock_pair_split_supported(channels, ksize, 1), "resblock_pair_split_pack: C=%d k=%d unsupported", channels, ksize);
  const int C = channels, CK = C >= 32 ? 32 : 16, KB = CK / 16, NCH = C / CK, MTT = (C + 31) / 32;
  float scale[2];
  for (int ph = 0; ph < 2; ++ph) {  // per conv: w 2^s with the largest weight in [2^13, 2^14) (conv1d.hip / resblock_stage_f32.hip)
    const float* w = ph ? h_w2 : h_w1;
    float wmax = 0.f;
    for (size_t q = 0; q < (size_t)C * C * ksize; ++q) wmax = std::max(wmax, std::fabs(w[q]));
    int sexp = 0;
    if (wmax > 0.f && std::isfinite(wmax)) {
      int e2;
      std::frexp(wmax, &e2);
      sexp = std::max(-24, std::min(40, 14 - e2));
    }
    scale[ph] = std::ldexp(1.f, sexp);
    h_unscale[ph] = std::ldexp(1.f, -sexp);
  }
  h16* out = reinterpret_cast<h16*>(h_packed);
  size_t o = 0;
  for (int mt = 0; mt < MTT; ++mt)
    for (int ph = 0; ph < 2; ++ph) {
      const float* w = ph ? h_w2 : h_w1;
      for (int c = 0; c < NCH; ++c)
        for (int j = 0; j < ksize; ++j)
          for (int u = 0; u < KB; ++u)
            for (int part = 0; part < 2; ++part)
              for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                  // A fragment of v_mfma_f32_32x32x16_f16: lane l holds A[m = l & 31][k = 8 (l >> 5) + e]
                  const int co = mt * 32 + (lane & 31);
                  const int ci = c * CK + u * 16 + (lane >> 5) * 8 + e;
                  const float v = co < C ? w[((size_t)co * C + ci) * ksize + j] * scale[ph] : 0.f;
                  const h16 hi = (h16)v;
                  out[o++] = part == 0 ? hi : (h16)(v - (float)hi);
                }
    }
  return MB_OK;
}

extern "C" int mb_resblock_pair_split(const mb_resblock_pair_split_args* a, mb_stream_t stream) {
  MB_REQUIRE(a && a->d_x && a->d_y && a->d_wpacked && a->d_b1 && a->d_b2, "resblock_pair_split: null pointer");
  MB_REQUIRE(a->d_x != a->d_y, "resblock_pair_split: in-place is not supported (tiles read their neighbours' halo)");
  MB_REQUIRE(mb_resblock_pair_split_supported(a->channels, a->ksize, a->dilation),
             "resblock_pair_split: C=%d k=%d d=%d unsupported", a->channels, a->ksize, a->dilation);
  MB_REQUIRE(a->slope > 0.f && a->slope < 1.f, "resblock_pair_split: leaky_relu slope must be in (0,1)");
  MB_REQUIRE(a->unscale1 > 0.f && a->unscale2 > 0.f, "resblock_pair_split: unscale factors of mb_resblock_pair_split_pack missing");
  if (a->batch <= 0 || a->t <= 0) return MB_OK;
  ResPairSK k;
  memset(&k, 0, sizeof(k));
  k.x = a->d_x; k.y = a->d_y;
  k.w = reinterpret_cast<const h16*>(a->d_wpacked); k.b1 = a->d_b1; k.b2 = a->d_b2;
  k.bstride = (long long)a->t * a->channels;
  k.T = a->t; k.ntaps = a->ksize; k.dil = a->dilation;
  k.valid = a->d_valid; k.valid_mul = a->valid_mul > 0 ? a->valid_mul : 1;
  k.slope = a->slope; k.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale; k.accumulate = a->accumulate;
  k.us1 = a->unscale1; k.us2 = a->unscale2;
  k.range_events = conv_range_word();
  hipStream_t s = (hipStream_t)stream;
  const int ks = a->ksize, dl = a->dilation;
  const int force = diag_int("spair_tile", 0);  // A/B: 1 = first candidate, 2 = second
  // tile choice: the candidate with the smaller makespan (rounds of persistent workgroups x rows per tile) that fits LDS; ties -> the
  // larger tile (less halo recompute, more reuse of every weight fragment)
  const int cus = spair_cus();
  auto cost = [&](int n1) {
    const int nb = n1 - (ks - 1);
    const long long tiles = (long long)cdiv(a->t, nb) * a->batch;
    return ((tiles + cus - 1) / cus) * (long long)n1;
  };
  // ... or both: when the larger tile leaves a last round mostly empty (32 x 1000 rows at 256 channels: 352 tiles = 1.4 rounds), the
  // whole rounds run on the larger tile and the remaining rows of every item on the smaller one (a second launch on rows
  // [j NB_a, T)): 96 + 64 rows of makespan instead of 2 x 96.  OFF by default: measured on the HiFi-GAN forward it loses (8.29 against
  // 8.16 ms) -- the generator's parallel ResBlock chains already fill each other's last rounds from their branch streams (gan.hip) and
  // the cut adds six launches of small tiles; a caller that runs one chain at a time can ask for it (MBHIP_DIAG=spair_split=1).
  const int split_mode = force == 0 ? diag_int("spair_split", 0) : 0;  // 0 = never, 1 = where the makespan says so, 2 = wherever the shapes allow it (tests)
  const bool may_split = split_mode != 0;
#define SP_PICK2(GA, YA, GB, YB)                                                                          \
  do {                                                                                                    \
    const bool fa = spair_fits<GA>(ks, dl, YA), fb = spair_fits<GB>(ks, dl, YB);                          \
    MB_REQUIRE(fa || fb, "resblock_pair_split: no instance fits LDS");                                     \
    constexpr int nl_ = GA::CH <= 64 ? SPAIR_NL_NARROW : 4;                                               \
    if (fa && fb && may_split && GB::N1 < GA::N1) {                                                        \
      const int nb_a = GA::N1 - (ks - 1), nb_b = GB::N1 - (ks - 1);                                        \
      const long long tiles_a = (long long)cdiv(a->t, nb_a) * a->batch, full = tiles_a / cus;              \
      if (full >= 1 && tiles_a % cus != 0 && (full * cus) % a->batch == 0) {                               \
        const int j = (int)(full * cus / a->batch), rest = a->t - j * nb_a;                                \
        const long long tiles_b = rest > 0 ? (long long)cdiv(rest, nb_b) * a->batch : 0;                   \
        const long long c_split = full * GA::N1 + ((tiles_b + cus - 1) / cus) * (GB::N1 + GB::N1 / 8);     \
        if (rest > 0 && (split_mode == 2 || c_split < std::min(cost(GA::N1), cost(GB::N1)))) {                                  \
          k.t_begin = 0; k.t_end = j * nb_a;                                                               \
          const int r = launch_spair<GA::CH, GA::MT, GA::WN, GA::NTW, YA, nl_>(k, a->batch, s);            \
          if (r) return r;                                                                                 \
          k.t_begin = j * nb_a; k.t_end = a->t;                                                            \
          return launch_spair<GB::CH, GB::MT, GB::WN, GB::NTW, YB, nl_>(k, a->batch, s);                   \
        }                                                                                                  \
      }                                                                                                    \
    }                                                                                                      \
    const bool pick_b = fb && (!fa || force == 2 || (force != 1 && cost(GB::N1) < cost(GA::N1)));          \
    if (pick_b) return launch_spair<GB::CH, GB::MT, GB::WN, GB::NTW, YB, nl_>(k, a->batch, s);        \
    return launch_spair<GA::CH, GA::MT, GA::WN, GA::NTW, YA, nl_>(k, a->batch, s);                    \
  } while (0)
  switch (a->channels) {
    case 256: SP_PICK2(SG256a, false, SG256b, false);
    case 128: SP_PICK2(SG128a, false, SG128b, false);
    case 64: SP_PICK2(SG64a, true, SG64a, false);
    case 32: SP_PICK2(SG32a, true, SG32b, true);
    default: SP_PICK2(SG16a, true, SG16b, true);
  }
#undef SP_PICK2
}

// ---- fp32 layout changes between the reference's channel-major [B][C][T] and this path's time-major [B][T][C] ----
namespace mb {
template <bool TO_TM>
__global__ __launch_bounds__(256) void f32_transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T,
                                                            const int* __restrict__ valid, int valid_mul) {
  // TO_TM: x [C][T] -> y [T][C];  else x [T][C] -> y [C][T].  64 x 64 tiles through LDS, both sides in 256-byte row segments.
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int R = TO_TM ? C : T, S = TO_TM ? T : C;  // x is [R][S], y is [S][R]
  const int r0 = blockIdx.y * 64, s0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* xb = x + (long long)b * C * T;
  float* yb = y + (long long)b * C * T;
  const int Tb = valid ? min(T, valid[b] * valid_mul) : T;  // positions beyond: zeros (the item's padding)
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, ss = s0 + tx;
    const int t = TO_TM ? ss : rr;
    tile[r][tx] = (rr < R && ss < S && t < Tb) ? xb[(long long)rr * S + ss] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int ss = s0 + r, rr = r0 + tx;
    if (ss < S && rr < R) yb[(long long)ss * R + rr] = tile[tx][r];
  }
}
}  // namespace mb

extern "C" int mb_f32_cm_to_tm(const float* d_x, float* d_y, int batch, int channels, int t, mb_stream_t stream) {
  MB_REQUIRE(d_x && d_y && d_x != d_y, "f32_cm_to_tm: null pointer / in place");
  if (batch <= 0 || channels <= 0 || t <= 0) return MB_OK;
  dim3 grid(cdiv(t, 64), cdiv(channels, 64), batch);
  hipLaunchKernelGGL((f32_transpose_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, d_x, d_y, channels, t, (const int*)nullptr, 1);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

extern "C" int mb_f32_tm_to_cm(const float* d_x, float* d_y, int batch, int channels, int t, mb_stream_t stream) {
  MB_REQUIRE(d_x && d_y && d_x != d_y, "f32_tm_to_cm: null pointer / in place");
  if (batch <= 0 || channels <= 0 || t <= 0) return MB_OK;
  dim3 grid(cdiv(channels, 64), cdiv(t, 64), batch);
  hipLaunchKernelGGL((f32_transpose_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, d_x, d_y, channels, t, (const int*)nullptr, 1);
  MB_HIP(hipGetLastError());
  return MB_OK;
}
