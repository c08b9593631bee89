// Fused ResBlock unit of the GAN vocoders on the gfx950 fp16 matrix cores:
//
//   y = x + conv2( lrelu( conv1( lrelu(x) ) + b1 ) ) + b2        conv1: k taps, dilation d; conv2: k taps, dilation 1
//
// = one (convs1[i], convs2[i]) iteration of ResBlock1.forward
//   models/vocoder/hifigan/models.py:39-46, models/vocoder/fregan/generator.py:43-50
// with the intermediate activation kept in LDS: the unfused path (conv1d_f16.hip, one launch per
// conv) moves 3 x C*T fp16 tensors through HBM per conv, this moves x in and y out once per PAIR
// (3x less traffic; the C <= 64 stages were HBM-bound).  Optionally y is scaled and accumulated
// into the stage output (the mean over the parallel ResBlocks, models.py:141-145).
//
// Layout: time-major [B][T][C] fp16 (conv1d_f16.hip).  C in {16, 32, 64, 128, 256}, k odd.
//
// One PERSISTENT workgroup per CU walks over output tiles of NB = N1 - (k-1) positions:
//   phase 1  h[N1 x C]  = lrelu(W1 * lrelu(x window) + b1)   -> LDS (fp16), zero outside [0, T)
//   phase 2  conv2(h) + b2 (fp16)                            -> LDS (the h tile, or its own y tile for C <= 64)
//   write-out y = (acc ? y : 0) + scale * (x + conv2(h) + b2) -> HBM, coalesced 16-byte rows
// 8 waves, two roles.  Waves 0-3 ("MMA") run the MFMA loops (v_mfma_f32_32x32x16_f16, each wave MT x NTW
// 32x32 tiles) and touch only LDS and their weight prefetch.  Waves 4-7 ("support") own all HBM traffic:
// they stage the next 64-channel chunk of the x window (leaky-relu applied) into a ring of LDS buffers
// while the MMA waves consume the current one, and write the previous tile's y out (adding the residual)
// while the MMA waves are already in the next tile.  The roles are separate waves because vector-memory
// returns are in order per wave: loads issued by an MMA wave would sit in front of its own weight
// prefetches and stall the MFMA chain.  The waves meet at s_barriers only (B per chunk, [W], E1, P|YF, Y).
//
// Weights never touch LDS: they are packed on the host as ONE circular stream per 32-channel output
// tile in exactly the order the kernel consumes them ([conv1: chunk, tap, k-block][conv2: ...]), and
// each MMA wave keeps a register ring of the next TWO taps' A fragments (8 k-steps ~ 0.9 us of MFMA
// work ahead), which runs seamlessly across chunk, phase and tile boundaries.  k is odd, so the ring
// slot of a chunk's first tap alternates 0,1,0,1,...: the chunk body exists in two statically
// scheduled variants (the compiler's s_waitcnt placement stays exact, no dynamic ring indexing).
#include <type_traits>
#include "common.h"

namespace mb {

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

struct ResPairK {
  const h16* x; h16* y; const h16* w; const float* b1; const float* b2;
  long long bstride;  // elements per batch item (T*C)
  int T, ntaps, dil;
  int NB, tiles_per_item, n_tiles, x_rows;
  unsigned long long* trace;  // diagnostics only (MBHIP_DIAG=pair_trace=<file>): shader-clock marks of workgroup 0
  int dbg;   // diagnostics only (MBHIP_DIAG=pair_dbg=<bits>): 1 = weight stream folded onto its first taps (L1-resident),
             // 2 = no residual read, 4 = no output store -- results are wrong, timings isolate one cost each
  int nbuf;  // LDS buffers of the x window (1 = single buffer, refilled while phase 2 runs; C <= 64 only)
  float slope, out_scale;
  int accumulate;
  const int* valid; int valid_mul;  // ragged batches: item b has valid[b] * valid_mul positions (null: T)
};

#ifndef MB_PAIR_NL
#define MB_PAIR_NL 4
#endif
constexpr int PAIR_NL = MB_PAIR_NL;  // support waves per workgroup (beside the 4 MMA waves)
constexpr int PAIR_LB = 80 / PAIR_NL;  // x-window loads in flight per support lane per batch
// Activations stream through once per launch; the weights are re-read by every tile.  Non-temporal
// activation loads/stores keep the 4 MiB per-XCD L2 for the weight stream.
#ifndef MB_PAIR_NT
#define MB_PAIR_NT 0
#endif
// The MBHIP_DIAG=pair_dbg=<bits> bits exist only in diagnostics builds (-DMB_PAIR_DBG_BUILD): a run-time test around the weight
// prefetch made every ring refill a conditional load, and the compiler then drained vmcnt to 0 at every tap pair.
#ifdef MB_PAIR_DBG_BUILD
#define MB_PDBG(a_, bit) ((a_).dbg & (bit))
#else
#define MB_PDBG(a_, bit) 0
#endif
constexpr bool PAIR_NT = MB_PAIR_NT != 0;

// diagnostics: mark k of tile `it`, role 0 = MMA wave 0, 1 = support wave 4 (workgroup 0, first 8 tiles)
#ifdef MB_PAIR_TRACE_BUILD
#define MB_PMARK(role, it, k)                                                                  \
  do {                                                                                         \
    if (a.trace && blockIdx.x == 0 && (it) < 8 && (tid & 63) == 0 && wave == ((role) ? 4 : 0)) \
      a.trace[((role) * 8 + (it)) * 16 + (k)] = (unsigned long long)clock64();                 \
  } while (0)
#else  // a conditional global store in front of an MFMA loop makes the compiler's vmcnt bookkeeping give up (vmcnt(0) per iteration)
#define MB_PMARK(role, it, k) do { } while (0)
#endif

// Valid length of batch item b (ragged batches).  A SCALAR load with its own wait: as a vector-memory load (what the compiler
// emits for a pointer it cannot prove read-only) it sat, conditionally, at the head of every tile -- draining the MMA waves'
// weight ring and leaving the compiler's vmcnt bookkeeping imprecise (vmcnt(0) in every iteration of the first tap loops).
__device__ __forceinline__ int pair_valid_len(const ResPairK& a, int b) {
  if (!a.valid) return a.T;
  int v;
  const int* p = a.valid + b;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return min(a.T, v * a.valid_mul);
}

template <int C> struct PairGeom {
  static constexpr int CK = C >= 64 ? 64 : (C >= 32 ? 32 : 16);  // channels per x chunk
  static constexpr int KB = CK / 16;            // k-steps per tap per chunk
  static constexpr int NCH = C / CK;
  static constexpr int MTT = (C + 31) / 32;  // C = 16: one 32-row tile whose upper half has zero weights
  static constexpr int WM = MTT >= 8 ? 4 : (MTT >= 4 ? 2 : 1);
  static constexpr int MT = MTT / WM;
  static constexpr int WN = 4 / WM;
  static constexpr int CKP = CK + 8, CP = C + 8;  // LDS row strides (odd multiples of 16 B)
};

// TD = taps of weight prefetch kept in flight per wave (2: 64 VGPRs at MT = 2; 1 for the instances whose
// accumulators leave no room -- the kernel must stay within 256 VGPRs because the loader wave shares
// a SIMD with an MMA wave)
// YS = y staged in its own LDS tile (C <= 64, where LDS has room): the support waves then have the whole
// tile time -- not only phase 1 -- for the x prefetch and the y write-out, which is what bounds those
// HBM-limited stages.  Without it y is staged in the h tile and must be written out before the next h.
template <int C, int NTW, int TD, bool YS>
__global__ __launch_bounds__(64 * (4 + PAIR_NL)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock_pair_f16_kernel(ResPairK a) {
  using G = PairGeom<C>;
  constexpr int CK = G::CK, KB = G::KB, NCH = G::NCH, MT = G::MT, WN = G::WN;
  constexpr int CKP = G::CKP, CP = G::CP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  h16* xs = reinterpret_cast<h16*>(lds_raw);           // [nbuf][x_rows][CKP]
  h16* hs = xs + a.nbuf * a.x_rows * CKP;               // [N1 + ntaps - 1][CP]
  // biases live in LDS: a vector-memory read in the epilogues would queue behind the weight prefetches
  h16* ys = YS ? hs + (WN * NTW * 32 + a.ntaps - 1) * CP : hs;  // [N1][CP] staged conv2 + b2
  float* bs = reinterpret_cast<float*>(ys + (YS ? WN * NTW * 32 * CP : (WN * NTW * 32 + a.ntaps - 1) * CP));  // [2][C]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = a.ntaps;
  const int p2 = (ntaps - 1) >> 1, p1 = p2 * a.dil;
  const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int njobs = my_tiles * NCH;

  for (int i = tid; i < 2 * C; i += 64 * (4 + PAIR_NL)) bs[i] = i < C ? a.b1[i] : a.b2[i - C];
  // Z: the MMA waves start their accumulators from bs BEFORE the first barrier of the tile loop.  Without this one the first
  // tile of a workgroup raced with the fill and took whatever the LDS held (mostly the previous launch's bias block: one tile of
  // a launch now and then off by a bias difference -- found by test_resblock_stage_equals_unit_launches failing once in five runs)
  __syncthreads();

  if (wave >= 4) {
    // ------------------------------ loader waves ------------------------------
    // PAIR_NL waves split the 16-byte pieces of the window; every piece of a wave is in flight at once
    // (one HBM round trip per chunk), then leaky-relu'd and written to LDS.
    constexpr int PPR = CK / 8;  // 16-byte pieces per row
    constexpr int LB = PAIR_LB;  // loads in flight per lane per batch
    const h16 slope = (h16)a.slope;
    const int total = a.x_rows * PPR;
    const int ltid = tid - 256;
    h16x8 v[LB];
    auto load_batch = [&](int q, int base) {
      const int tile = (int)blockIdx.x + (q / NCH) * (int)gridDim.x, c = q % NCH;
      const int b = tile / a.tiles_per_item, t0 = (tile - b * a.tiles_per_item) * a.NB;
      const int Tb = pair_valid_len(a, b);  // beyond: this item's zero padding
      const int tx0 = t0 - p2 - p1;
      const h16* xb = a.x + (long long)b * a.bstride + c * CK;
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int idx = base + i * (64 * PAIR_NL) + ltid;
        const int row = idx / PPR, pc = idx - row * PPR;
        const int tx = tx0 + row;
        v[i] = (h16x8)(h16)0.f;
        if (idx < total && tx >= 0 && tx < Tb)
          v[i] = PAIR_NT ? __builtin_nontemporal_load(reinterpret_cast<const h16x8*>(xb + (long long)tx * C + pc * 8))
                         : *reinterpret_cast<const h16x8*>(xb + (long long)tx * C + pc * 8);
      }
    };
    auto store_batch = [&](int q, int base) {
      h16* buf = xs + (q % a.nbuf) * a.x_rows * CKP;
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int idx = base + i * (64 * PAIR_NL) + ltid;
        const int row = idx / PPR, pc = idx - row * PPR;
        const h16x8 s = __builtin_elementwise_max(v[i], v[i] * slope);  // leaky_relu, 0 < slope < 1
        if (idx < total) *reinterpret_cast<h16x8*>(buf + row * CKP + pc * 8) = s;
      }
    };
    auto fill = [&](int q) {
      for (int base = 0; base < total; base += 64 * PAIR_NL * LB) { load_batch(q, base); store_batch(q, base); }
    };
    // y write-out of a finished tile: the MMA waves leave conv2(h) + b2 in hs (fp16); here the residual
    // x (and the running sum when accumulating) is added and the rows leave as coalesced 16-byte stores.
    // Runs while the MMA waves are already in phase 1 of the next tile -- they never wait on HBM.
    constexpr int WB = 8;         // pieces per lane per batch
    constexpr int YPR = C / 8;    // 16-byte pieces per output row
    auto write_out = [&](int it, int lo, int hi, int den) {  // batches [nbt*lo/den, nbt*hi/den) of tile `it`
      const int tile = (int)blockIdx.x + it * (int)gridDim.x;
      const int b = tile / a.tiles_per_item, t0 = (tile - b * a.tiles_per_item) * a.NB;
      const int Tb = pair_valid_len(a, b);
      const int rows = max(0, min(a.NB, Tb - t0));
      const int ytotal = rows * YPR;
      const h16* xb = a.x + (long long)b * a.bstride + (long long)t0 * C;
      h16* yb = a.y + (long long)b * a.bstride + (long long)t0 * C;
      constexpr int BSZ = 64 * PAIR_NL * WB;
      const int nbt = (ytotal + BSZ - 1) / BSZ;
      const int b_lo = nbt * lo / den, b_hi = nbt * hi / den;
      for (int base = b_lo * BSZ; base < b_hi * BSZ && base < ytotal; base += BSZ) {
        h16x8 rx[WB], ry[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          int idx = base + i * (64 * PAIR_NL) + ltid;
          idx = idx < ytotal ? idx : ytotal - 1;  // clamped: loads legal, store predicated
          rx[i] = MB_PDBG(a, 2) ? (h16x8)(h16)0.f : *reinterpret_cast<const h16x8*>(xb + (long long)idx * 8);
          if (a.accumulate) ry[i] = *reinterpret_cast<const h16x8*>(yb + (long long)idx * 8);
        }
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int idx = base + i * (64 * PAIR_NL) + ltid;
          const int idc = idx < ytotal ? idx : ytotal - 1;
          const int row = idc / YPR, pc = idc - row * YPR;
          const h16x8 hv = *reinterpret_cast<const h16x8*>(ys + row * CP + pc * 8);
          h16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = ((float)hv[e] + (float)rx[i][e]) * a.out_scale;
            if (a.accumulate) f += (float)ry[i][e];
            o[e] = (h16)f;
          }
          if (idx < ytotal && !MB_PDBG(a, 4)) *reinterpret_cast<h16x8*>(yb + (long long)idx * 8) = o;
        }
      }
    };
    // Barrier schedule per tile (must mirror the MMA waves'): B per chunk, [W], E1, P|YF, Y.
    if (a.nbuf == 1) {
      // single buffer (NCH == 1, window = one batch): the next tile's loads fly during phase 1 and are
      // written to LDS while the MMA waves run phase 2, which reads only h
      if (my_tiles > 0) fill(0);
      for (int it = 0; it < my_tiles; ++it) {
        MB_PMARK(1, it, 0);
        __syncthreads();  // B
        MB_PMARK(1, it, 1);
        if (it + 1 < my_tiles) load_batch(it + 1, 0);
        MB_PMARK(1, it, 2);
        if (it > 0) write_out(it - 1, 0, YS ? 3 : 8, 8);  // with its own y tile: 3/8 now, the rest during phase 2
        MB_PMARK(1, it, 3);
        if (!YS) __syncthreads();  // W: hs is free for h of this tile
        __syncthreads();  // E1: phase 1 has finished reading xs
        MB_PMARK(1, it, 4);
        if (it + 1 < my_tiles) store_batch(it + 1, 0);
        if (YS && it > 0) write_out(it - 1, 3, 8, 8);
        MB_PMARK(1, it, 5);
        __syncthreads();  // P (all MMA waves done with h) | YF (ys is free for y of this tile)
        __syncthreads();  // Y: y of this tile is staged
        MB_PMARK(1, it, 6);
      }
    } else {
      for (int q = 0; q < a.nbuf - 1 && q < njobs; ++q) fill(q);
      for (int it = 0; it < my_tiles; ++it) {
        for (int c = 0; c < NCH; ++c) {
          const int q = it * NCH + c;
          if (c == 0) MB_PMARK(1, it, 0);
          __syncthreads();  // B_q: job q is staged, the buffer of job q-1 is free
          if (c == 0) MB_PMARK(1, it, 1);
          if (q + a.nbuf - 1 < njobs) fill(q + a.nbuf - 1);
          if (c == 0) MB_PMARK(1, it, 2);
          if (!YS && it > 0) write_out(it - 1, c, c + 1, NCH);  // spread over the chunks: no B barrier waits long
          if (c == 0) MB_PMARK(1, it, 3);
        }
        if (!YS) __syncthreads();  // W
        __syncthreads();  // E1
        MB_PMARK(1, it, 4);
        if (YS && it > 0) write_out(it - 1, 0, 1, 1);
        __syncthreads();  // P | YF
        __syncthreads();  // Y
        MB_PMARK(1, it, 6);
      }
    }
    if (my_tiles > 0) write_out(my_tiles - 1, 0, 1, 1);
    return;
  }

  // ------------------------------ MMA waves ------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int mt0 = wm * MT;
  const int NFT = 2 * NCH * ntaps;  // flat taps of the circular weight stream
  const h16x8* wp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
    wp[i] = reinterpret_cast<const h16x8*>(a.w) + (size_t)(mt0 + i) * NFT * KB * 64 + lane;

  h16x8 ring[TD][KB][MT];
#pragma unroll
  for (int s = 0; s < TD; ++s)
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i) ring[s][u][i] = wp[i][(size_t)(s * KB + u) * 64];
  int ftn = TD;  // next flat tap to prefetch

  f32x16 acc[MT][NTW];
  // the accumulators start from the bias (accumulator layout: channel 32 mt + 8 g + 4 (lane >> 5) + e at element 4 g + e): what
  // used to be the zeroing costs the same moves, and the epilogues lose their LDS bias reads and adds (132 -> ~60 cycles per group)
  auto init_acc = [&](const float* bias) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co0 = (mt0 + i) * 32 + 8 * g + 4 * (lane >> 5);
        const f32x4 q = co0 < C ? *reinterpret_cast<const f32x4*>(bias + co0) : (f32x4)0.f;  // (C = 16: rows 16..31 are padding)
        acc[i][0][4 * g] = q[0]; acc[i][0][4 * g + 1] = q[1]; acc[i][0][4 * g + 2] = q[2]; acc[i][0][4 * g + 3] = q[3];
      }
#pragma unroll
      for (int n = 1; n < NTW; ++n) acc[i][n] = acc[i][0];
    }
  };

  // one tap: KB k-steps from ring slot S; refills the slot with flat tap ftn.  The B fragments (LDS) are
  // software-pipelined one k-step ahead (bfc_ = current, loaded during the previous step's MFMAs): with
  // one MMA wave per SIMD nothing else would hide the ds_read latency.  NEXT = first row of the next tap
  // (or any valid row after the chunk's last tap: that read is discarded).
#define MB_TAP(S, BPTR, NEXT, RS)                                                                  \
  do {                                                                                             \
    const h16* bp_ = (BPTR);                                                                       \
    const h16* np_ = (NEXT);                                                                       \
    const size_t nf_ = (size_t)(MB_PDBG(a, 1) ? (ftn & 1) : ftn) * KB;                             \
    _Pragma("unroll") for (int u = 0; u < KB; ++u) {                                               \
      h16x8 bfn_[NTW];                                                                             \
      const h16* rp_ = u + 1 < KB ? bp_ + (u + 1) * 16 : np_;                                      \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                              \
        bfn_[n] = *reinterpret_cast<const h16x8*>(rp_ + n * 32 * (RS));                            \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                            \
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i], bfc_[n], acc[i][n], 0, 0, 0); \
      /* the refill of a ring slot is issued AFTER the MFMAs that read it: old and new value never live together, */ \
      /* so the slot keeps its registers around the loop (no copies, no vmcnt drain at the back edge) */ \
      if (!MB_PDBG(a, 8)) { _Pragma("unroll") for (int i = 0; i < MT; ++i) ring[S][u][i] = wp[i][(nf_ + u) * 64]; } \
      /* issue order: ONE LDS read behind each of the first NTW MFMAs (a burst of NTW reads from four waves at once backs */ \
      /* the LDS queue up into the issuing wave: 46 -> 41 cycles per MFMA measured in resblock_stage_f16.hip), then the refills */ \
      _Pragma("unroll") for (int m = 0; m < MT * NTW; ++m) {                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        if (m < NTW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
      }                                                                                            \
      __builtin_amdgcn_sched_group_barrier(0x020, MT, 0);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n) bfc_[n] = bfn_[n];                           \
    }                                                                                              \
    ftn = ftn + 1 == NFT ? 0 : ftn + 1;                                                            \
  } while (0)

  // a chunk = ntaps taps (ntaps odd); S0 = ring slot of its first tap
#define MB_TAPJ(S, J)                                                                              \
  MB_TAP(S, cb_ + (size_t)(J) * ts_, cb_ + (size_t)((J) + 1 < ntaps ? (J) + 1 : 0) * ts_, rs_)
#define MB_CHUNK(S0, BASE, RS, TAPSTEP)                                                            \
  do {                                                                                             \
    const h16* cb_ = (BASE);                                                                       \
    const int rs_ = (RS);                                                                          \
    const size_t ts_ = (size_t)(TAPSTEP);                                                          \
    h16x8 bfc_[NTW];                                                                               \
    _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                                \
      bfc_[n] = *reinterpret_cast<const h16x8*>(cb_ + n * 32 * rs_);                               \
    int j_ = 0;                                                                                    \
    if (TD == 1) {                                                                                 \
      for (; j_ < ntaps; ++j_) MB_TAPJ(0, j_);                                                     \
      break;                                                                                       \
    }                                                                                              \
    if (S0 == 1) { MB_TAPJ(TD - 1, 0); j_ = 1; }                                                   \
    for (; j_ + 1 < ntaps; j_ += 2) {                                                              \
      MB_TAPJ(0, j_);                                                                              \
      MB_TAPJ(TD - 1, j_ + 1);                                                                     \
    }                                                                                              \
    if (S0 == 0) MB_TAPJ(0, ntaps - 1);                                                            \
  } while (0)

  const int lrow = wn * (NTW * 32) + (lane & 31);  // this lane's row inside an N tile group
  const int lcol = (lane >> 5) * 8;
  const int x_tapstep = a.dil * CKP;
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = (int)blockIdx.x + it * (int)gridDim.x;
    const int t0 = (tile % a.tiles_per_item) * a.NB;
    const int Tb = pair_valid_len(a, tile / a.tiles_per_item);
    // ---------------- phase 1: h = lrelu(conv1(lrelu(x)) + b1) ----------------
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    init_acc(bs);
    MB_PMARK(0, it, 0);
    if (NCH == 1) {
      __syncthreads();
      MB_PMARK(0, it, 1);
      MB_CHUNK(0, xs + ((it * NCH) % a.nbuf) * a.x_rows * CKP + lrow * CKP + lcol, CKP, x_tapstep);
    } else {
      for (int c = 0; c < NCH; c += 2) {
        __syncthreads();
        MB_CHUNK(0, xs + ((it * NCH + c) % a.nbuf) * a.x_rows * CKP + lrow * CKP + lcol, CKP, x_tapstep);
        __syncthreads();
        MB_CHUNK(1, xs + ((it * NCH + c + 1) % a.nbuf) * a.x_rows * CKP + lrow * CKP + lcol, CKP, x_tapstep);
      }
    }
    MB_PMARK(0, it, 2);
    if (!YS) __syncthreads();  // W: the support waves have written out the previous tile's y from hs
    MB_PMARK(0, it, 3);
    {  // epilogue 1 -> hs (fp16); rows outside [0, T) are conv2's zero padding (only the first / last tiles of an item have
       // any: every other wave skips the selects).  Packed math: this runs on the MMA waves' critical path (one wave per SIMD)
      const h16 hslope = (h16)a.slope;
      const int tw0 = t0 - p2 + wn * (NTW * 32);
      const bool interior = tw0 >= 0 && tw0 + NTW * 32 <= Tb;  // wave-uniform
      auto epi1 = [&](auto INTERIOR) {  // two instances: a merged one keeps the selects (the compiler folds the predicate in)
        constexpr bool interior_ = decltype(INTERIOR)::value;
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
              const f32x4 v = {acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]};
              h16x4 hv = __builtin_convertvector(v, h16x4);
              hv = __builtin_elementwise_max(hv, hv * hslope);  // leaky_relu in fp16, as the unfused path applies it
              if (!interior_ && !inside) hv = (h16x4)(h16)0.f;
              *reinterpret_cast<h16x4*>(hs + row * CP + co0) = hv;
            }
          }
      };
      if (interior) epi1(std::true_type{});
      else epi1(std::false_type{});
    }
    MB_PMARK(0, it, 4);
    __syncthreads();  // E1
    MB_PMARK(0, it, 5);
    // ---------------- phase 2: y = conv2(h) + b2 + x ----------------
    init_acc(bs + C);
    if (NCH == 1) {
      MB_CHUNK(1, hs + lrow * CP + lcol, CP, CP);
    } else {
      for (int c = 0; c < NCH; c += 2) {
        MB_CHUNK(0, hs + lrow * CP + c * CK + lcol, CP, CP);
        MB_CHUNK(1, hs + lrow * CP + (c + 1) * CK + lcol, CP, CP);
      }
    }
    MB_PMARK(0, it, 6);
    __syncthreads();  // P: every MMA wave has finished reading h | YF: the previous tile's y has left ys
    MB_PMARK(0, it, 7);
    {  // epilogue 2 -> ys: conv2 + b2 (fp16); the support waves add the residual and write y out
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int row = lrow + n * 32;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int co0 = (mt0 + i) * 32 + 8 * g + 4 * (lane >> 5);
            if (co0 >= C) continue;
            const f32x4 v = {acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]};
            *reinterpret_cast<h16x4*>(ys + row * CP + co0) = __builtin_convertvector(v, h16x4);
          }
        }
    }
    MB_PMARK(0, it, 8);
    __syncthreads();  // Y
    MB_PMARK(0, it, 9);
  }
#undef MB_CHUNK
#undef MB_TAPJ
#undef MB_TAP
}

// LDS bytes of a (C, NTW) instance for a given conv1 geometry and x-window buffer count
template <int C>
static size_t pair_lds_bytes(int ntw, int ntaps, int dil, int nbuf, bool ys = false) {
  using G = PairGeom<C>;
  const int n1 = G::WN * ntw * 32;
  const int x_rows = n1 + (ntaps - 1) * dil;
  return ((size_t)nbuf * x_rows * G::CKP + (size_t)(n1 + ntaps - 1 + (ys ? n1 : 0)) * G::CP) * sizeof(h16) + 2 * C * sizeof(float);
}
constexpr size_t PAIR_LDS_CAP = 160 * 1024;
// fewest buffers an instance can run with: 1 (single-buffer mode) when the window is one chunk and one
// loader batch, else 2
template <int C>
static int pair_min_nbuf(int ntw, int ntaps, int dil) {
  using G = PairGeom<C>;
  const int x_rows = G::WN * ntw * 32 + (ntaps - 1) * dil;
  const bool single_ok = G::NCH == 1 && x_rows * (G::CK / 8) <= 64 * PAIR_NL * PAIR_LB;
  return single_ok ? 1 : 2;
}
template <int C>
static bool pair_fits(int ntw, int ntaps, int dil, bool ys = false) {
  return pair_lds_bytes<C>(ntw, ntaps, dil, pair_min_nbuf<C>(ntw, ntaps, dil), ys) <= PAIR_LDS_CAP;
}

template <int C, int NTW, int TD, bool YS>
static int launch_pair(ResPairK k, int batch, hipStream_t s) {
  using G = PairGeom<C>;
  const int n1 = G::WN * NTW * 32;
  k.NB = n1 - (k.ntaps - 1);
  k.x_rows = n1 + (k.ntaps - 1) * k.dil;
  k.tiles_per_item = cdiv(k.T, k.NB);
  k.n_tiles = k.tiles_per_item * batch;
  // as many x-window buffers as fit (<= 4): the loaders run nbuf-1 chunks ahead of the MMA waves
  int nbuf = pair_min_nbuf<C>(NTW, k.ntaps, k.dil);
  while (nbuf < 4 && pair_lds_bytes<C>(NTW, k.ntaps, k.dil, nbuf + 1, YS) <= PAIR_LDS_CAP) ++nbuf;
  k.nbuf = nbuf;
  k.dbg = diag_int("pair_dbg", k.dbg);
  static unsigned long long* d_trace = nullptr;
  std::string trace_file;
  const char* trace_path = diag_str("pair_trace", &trace_file) ? trace_file.c_str() : nullptr;
  if (trace_path) {
    if (!d_trace) MB_HIP(hipMalloc((void**)&d_trace, 256 * sizeof(unsigned long long)));
    MB_HIP(hipMemsetAsync(d_trace, 0, 256 * sizeof(unsigned long long), s));
    k.trace = d_trace;
  }
  const size_t lds = pair_lds_bytes<C>(NTW, k.ntaps, k.dil, nbuf, YS);
  static bool attr_done = false;
  if (!attr_done) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_f16_kernel<C, NTW, TD, YS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  int n_cu = 256;
  {
    static int cached = 0;
    if (!cached) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cached = prop.multiProcessorCount;
      else
        cached = 256;
    }
    n_cu = cached;
  }
  const int grid = std::min(k.n_tiles, n_cu);
  hipLaunchKernelGGL((resblock_pair_f16_kernel<C, NTW, TD, YS>), dim3(grid), dim3(64 * (4 + PAIR_NL)), lds, s, k);
  MB_HIP(hipGetLastError());
  if (trace_path) {  // diagnostics: append "C NTW TD ntaps dil nbuf tiles : marks..." per launch
    unsigned long long h[256];
    MB_HIP(hipStreamSynchronize(s));
    MB_HIP(hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "a")) {
      fprintf(f, "%d %d %d %d %d %d %d :", C, NTW, TD, k.ntaps, k.dil, k.nbuf, k.n_tiles);
      for (int i = 0; i < 256; ++i) fprintf(f, " %llu", h[i]);
      fprintf(f, "\n");
      fclose(f);
    }
  }
  return MB_OK;
}

}  // namespace mb

using namespace mb;

extern "C" int mb_resblock_pair_f16_supported(int channels, int ksize, int dilation) {
  if (!(channels == 16 || channels == 32 || channels == 64 || channels == 128 || channels == 256)) return 0;
  if (ksize < 3 || (ksize & 1) == 0 || dilation < 1) return 0;
  switch (channels) {  // the candidates of mb_resblock_pair_f16's instance choice
    case 256: return pair_fits<256>(3, ksize, dilation) || pair_fits<256>(4, ksize, dilation);
    case 128: return pair_fits<128>(2, ksize, dilation) || pair_fits<128>(3, ksize, dilation);
    case 64: return pair_fits<64>(4, ksize, dilation) || pair_fits<64>(2, ksize, dilation, true);
    case 16: return pair_fits<16>(4, ksize, dilation, true) || pair_fits<16>(8, ksize, dilation, true);
    default: return pair_fits<32>(2, ksize, dilation, true) || pair_fits<32>(4, ksize, dilation, true);
  }
}

extern "C" size_t mb_resblock_pair_f16_packed_halves(int channels, int ksize) {
  // per 32-row output tile: 2 convs x (channels / 16) k-blocks x ksize taps, 512 halves per fragment
  const int mtt = (channels + 31) / 32;
  return (size_t)mtt * 2 * (channels / 16) * ksize * 512;
}

// h_w1 / h_w2: fp32 torch Conv1d weights [C][C][k] (weight norm already folded).
extern "C" int mb_resblock_pair_f16_pack(const float* h_w1, const float* h_w2, int channels, int ksize,
                                         uint16_t* h_packed) {
  MB_REQUIRE(h_w1 && h_w2 && h_packed, "resblock_pair_f16_pack: null pointer");
  MB_REQUIRE(mb_resblock_pair_f16_supported(channels, ksize, 1), "resblock_pair_f16_pack: C=%d k=%d unsupported",
             channels, ksize);
  const int C = channels, CK = C >= 64 ? 64 : (C >= 32 ? 32 : 16), KB = CK / 16, NCH = C / CK, MTT = (C + 31) / 32;
  h16* out = reinterpret_cast<h16*>(h_packed);
  size_t o = 0;
  for (int mt = 0; mt < MTT; ++mt)
    for (int ph = 0; ph < 2; ++ph) {
      const float* w = ph ? h_w2 : h_w1;
      for (int c = 0; c < NCH; ++c)
        for (int j = 0; j < ksize; ++j)
          for (int u = 0; u < KB; ++u)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                // A fragment of v_mfma_f32_32x32x16_f16: lane l holds A[m = l&31][k = 8*(l>>5) + e]
                const int co = mt * 32 + (lane & 31);
                const int ci = c * CK + u * 16 + (lane >> 5) * 8 + e;
                out[o++] = co < C ? (h16)w[((size_t)co * C + ci) * ksize + j] : (h16)0.f;
              }
    }
  return MB_OK;
}

extern "C" int mb_resblock_pair_f16(const mb_resblock_pair_f16_args* a, mb_stream_t stream) {
  MB_REQUIRE(a && a->d_x && a->d_y && a->d_wpacked && a->d_b1 && a->d_b2, "resblock_pair_f16: null pointer");
  MB_REQUIRE(a->d_x != a->d_y, "resblock_pair_f16: in-place is not supported (tiles read their neighbours' halo)");
  MB_REQUIRE(mb_resblock_pair_f16_supported(a->channels, a->ksize, a->dilation),
             "resblock_pair_f16: C=%d k=%d d=%d unsupported", a->channels, a->ksize, a->dilation);
  MB_REQUIRE(a->slope > 0.f && a->slope < 1.f, "resblock_pair_f16: leaky_relu slope must be in (0,1)");
  if (a->batch <= 0 || a->t <= 0) return MB_OK;
  ResPairK k;
  memset(&k, 0, sizeof(k));
  k.x = reinterpret_cast<const h16*>(a->d_x); k.y = reinterpret_cast<h16*>(a->d_y);
  k.w = reinterpret_cast<const h16*>(a->d_wpacked); k.b1 = a->d_b1; k.b2 = a->d_b2;
  k.bstride = (long long)a->t * a->channels;
  k.T = a->t; k.ntaps = a->ksize; k.dil = a->dilation;
  k.valid = a->d_valid; k.valid_mul = a->valid_mul > 0 ? a->valid_mul : 1;
  k.slope = a->slope; k.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale; k.accumulate = a->accumulate;
  hipStream_t s = (hipStream_t)stream;
  // N-tile count per wave: the candidate with the smallest makespan (rounds of 256 persistent
  // workgroups x positions per tile) that fits LDS; ties -> the larger tile (less halo recompute,
  // more reuse of every weight fragment).
  auto cost = [&](int n1) {
    const int nb = n1 - (a->ksize - 1);
    const long long tiles = (long long)cdiv(a->t, nb) * a->batch;
    return ((tiles + 255) / 256) * (long long)n1;
  };
#define MB_PICK2(C_, WN_, NA, TDA, YA, NB_, TDB, YB)                                                \
  do {                                                                                              \
    const bool fa = pair_fits<C_>(NA, a->ksize, a->dilation, YA);                                   \
    const bool fb = pair_fits<C_>(NB_, a->ksize, a->dilation, YB);                                  \
    MB_REQUIRE(fa || fb, "resblock_pair_f16: no instance fits LDS");                                 \
    if (fb && (!fa || prefer_b || cost(WN_ * NB_ * 32) <= cost(WN_ * NA * 32)))                     \
      return launch_pair<C_, NB_, TDB, YB>(k, a->batch, s);                                         \
    return launch_pair<C_, NA, TDA, YA>(k, a->batch, s);                                            \
  } while (0)
  bool prefer_b = false;
  switch (a->channels) {
    case 256:
      // a third candidate for short stages: 160-row tiles (7 per 1000 positions: one round of 224 workgroups where 96-row tiles
      // take two rounds of 384 -- the makespan estimate decides)
      if (pair_fits<256>(5, a->ksize, a->dilation) && cost(160) < cost(96) && cost(160) < cost(128))
        return launch_pair<256, 5, 1, false>(k, a->batch, s);
      MB_PICK2(256, 1, 3, 2, false, 4, 1, false);
    case 128: MB_PICK2(128, 2, 2, 2, false, 3, 2, false);
    case 64:
      // N1 = 256 with its own y tile (support waves overlap the whole tile) measured faster than N1 = 512 sharing the h tile
      prefer_b = true;
      MB_PICK2(64, 4, 4, 1, false, 2, 2, true);
    case 16: prefer_b = true; MB_PICK2(16, 4, 4, 2, true, 8, 2, true);  // 1024-position tiles: per-tile overheads amortise
    default: prefer_b = true; MB_PICK2(32, 4, 2, 2, true, 4, 2, true);
  }
#undef MB_PICK2
}
