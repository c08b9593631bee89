// One upsampling stage's whole ResBlock group of the GAN vocoders on the gfx950 fp16 matrix cores, ONE launch:
//
//   y = (1 / num_kernels) * sum_j ResBlock_j(x),   ResBlock_j:  for u in 0..nd-1:  x <- x + conv2_u(lrelu(conv1_u(lrelu(x)) + b1) ) + b2
//
// = the inner loop of Generator.forward over self.resblocks (models/vocoder/hifigan/models.py:139-145,
// models/vocoder/fregan/generator.py:150-157) with ResBlock1.forward (hifigan/models.py:39-46, fregan/generator.py:43-50)
// for the narrow stages (C <= 32), where the per-unit kernel of resblock_f16.hip is bound by the three tensor passes every
// unit makes through HBM (9 launches per stage: 18 reads + 9 writes of the C x T tensor, measured 4.8 - 6.2 B/clk/CU with the
// matrix cores 48 - 65 % busy inside their phases and idle a third of the tile).  Here a tile's x window is read ONCE, the
// nd units of each of the num_kernels chains run back to back on LDS-resident activations, the running sum of the chains'
// results stays in LDS, and one tensor is written: 2 passes per stage instead of 27.
//
// Layout: time-major [B][T][C] fp16 (conv1d_f16.hip).  C in {16, 32}, k odd.
//
// A tile = N1 = 128 * NTW window rows; row i <-> position t = t0 - Hh + i, Hh = the widest chain's total reach per side
// (sum over its units of (k-1)/2 * (d_u + 1): 60 for k = 11, d = 1,3,5).  Every conv of every unit computes all N1 rows;
// rows closer than the reach consumed so far to the window edge are garbage that never reaches the NB = N1 - 2 Hh rows in
// the middle (a conv output column depends only on input columns inside its own taps), which are the tile's result.
//
// LDS: A [PAD + N1 + PAD][C + 8]  lrelu(x) of the running unit (the conv1 operand), zero outside [0, T)
//      H [PAD + N1 + PAD][C + 8]  the chain's input x as stored (read once into registers), then h = lrelu(conv1 + b1) per unit
//      O [NB][C + 8]              running sum of the chains' results, written out by the support waves during the next tile
// The residual x of a chain lives in the registers of the MMA wave that owns the rows (the accumulator layout, packed fp16):
// a wave produces the same (row, channel) cells in every conv, so the residual never touches LDS or HBM again.
//
// 8 waves, two roles, as in resblock_f16.hip.  Waves 0-3 (MMA): each NTW 32x32 tiles of rows, all channels; B operands from
// LDS, A operands (weights) through a register ring over ONE circular stream in consumption order
// ([chain][unit][conv][tap][k-step], packed on the host), which runs seamlessly across conv, unit, chain and tile boundaries.
// Waves 4-7 (support) own all HBM traffic: they hold the tile's window in registers and lay it down at every chain start
// (raw -> H, lrelu -> A), fetch the next tile's window during the first chain, and write the previous tile's O out.
#include <type_traits>
#include "common.h"

namespace mb {

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

constexpr int STAGE_MAX_CHAINS = 4, STAGE_MAX_UNITS = 4;

struct ResStageK {
  const h16* x; h16* y; const h16* w; const float* bias;  // bias: [chain][unit][2][C]
  long long bstride;
  int T, nchains, nunits;
  int ntaps[STAGE_MAX_CHAINS];
  int dil[STAGE_MAX_CHAINS][STAGE_MAX_UNITS];
  int Hh, PAD, NB, tiles_per_item, n_tiles, NFT;
  float slope, out_scale;
  int accumulate;  // y += (the wider stages run one ResBlock per launch: the mean over the ResBlocks accumulates in y)
  const int* valid; int valid_mul;
  unsigned long long* trace;  // diagnostics only (-DMB_STAGE_TRACE_BUILD + MBHIP_DIAG=stage_trace=<file>): shader-clock marks of workgroup 0, tile 1
};

#ifdef MB_STAGE_TRACE_BUILD
#define MB_SMARK(k)                                                                        \
  do {                                                                                     \
    if (a.trace && blockIdx.x == 0 && it == 1 && tid == 0)                                 \
      a.trace[(c * STAGE_MAX_UNITS + u_) * 8 + (k)] = (unsigned long long)clock64();       \
  } while (0)
#else
#define MB_SMARK(k) do { } while (0)
#endif

__device__ __forceinline__ int stage_valid_len(const ResStageK& a, int b) {  // scalar load + its own wait (resblock_f16.hip)
  if (!a.valid) return a.T;
  int v;
  const int* p = a.valid + b;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return min(a.T, v * a.valid_mul);
}

// C channels = WM x MT 32-row output tiles (WM waves along the channels, MT tiles per wave), WN = 4 / WM waves along the rows with
// NTW 32-row tiles each: <16 | 32, 1, 4, NTW> for the narrow stages (one launch = the whole ResBlock group), <64, 2, 4, 2> and
// <128, 2, 2, 2> for single ResBlocks of the wider stages whose halo is small against what LDS holds (k = 3, 7 at 64 channels,
// k = 3 at 128).  TD = taps of weight prefetch in flight, BD = k-steps the B fragments (LDS) run ahead of the MFMAs.
template <int C, int MT, int WN, int NTW, int TD, int BD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void resblock_stage_f16_kernel(ResStageK a) {
  constexpr int WM = 4 / WN;
  static_assert(C == 16 || WM * MT * 32 == C, "the four MMA waves cover all channels");
  constexpr int KB = C / 16;         // k-steps per tap (the whole window lives in LDS: no channel chunks)
  constexpr int CP = C + 8;          // LDS row stride in halves (odd multiples of 16 B)
  constexpr int N1 = WN * NTW * 32;
  constexpr int CG = C >= 32 ? 4 : 2;  // 8-channel groups of a 32-row output tile that exist (C = 16: half a tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int R = N1 + 2 * a.PAD;
  h16* As = reinterpret_cast<h16*>(lds_raw);
  h16* Hs = As + R * CP;
  h16* Os = Hs + R * CP;
  float* bs = reinterpret_cast<float*>(Os + a.NB * CP);  // [chain][unit][2][C]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const h16 slope = (h16)a.slope;

  {  // zero A and H once (the PAD rows are never written again), stage the biases
    h16x8* z = reinterpret_cast<h16x8*>(lds_raw);
    const int n16 = 2 * R * CP / 8;
    for (int i = tid; i < n16; i += 512) z[i] = (h16x8)(h16)0.f;
    const int nb = a.nchains * a.nunits * 2 * C;
    for (int i = tid; i < nb; i += 512) bs[i] = a.bias[i];
  }
  __syncthreads();  // Z

  if (wave >= 4) {
    // ------------------------------ support waves ------------------------------
    constexpr int PPR = C / 8;              // 16-byte pieces per row
    constexpr int LB = N1 * PPR / 256;      // pieces per lane of one window (10 at C = 32, NTW = 5)
    static_assert(N1 * PPR % 256 == 0, "window pieces divide over the support lanes");
    const int ltid = tid - 256;
    h16x8 cur[LB], nxt[LB];
    auto load_window = [&](int it, h16x8 (&v)[LB]) {
      const int tile = (int)blockIdx.x + it * (int)gridDim.x;
      const int b = tile / a.tiles_per_item, t0 = (tile - b * a.tiles_per_item) * a.NB;
      const int Tb = stage_valid_len(a, b);
      const h16* xb = a.x + (long long)b * a.bstride;
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int idx = i * 256 + ltid;
        const int row = idx / PPR, pc = idx - row * PPR;
        const int t = t0 - a.Hh + row;
        const int tc = min(max(t, 0), a.T - 1);  // clamped: the load is always legal, the value is selected
        const h16x8 ld = *reinterpret_cast<const h16x8*>(xb + (long long)tc * C + pc * 8);
        v[i] = (t >= 0 && t < Tb) ? ld : (h16x8)(h16)0.f;
      }
    };
    auto store_window = [&](const h16x8 (&v)[LB]) {
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int idx = i * 256 + ltid;
        const int row = idx / PPR, pc = idx - row * PPR;
        const int o = (a.PAD + row) * CP + pc * 8;
        *reinterpret_cast<h16x8*>(Hs + o) = v[i];
        *reinterpret_cast<h16x8*>(As + o) = __builtin_elementwise_max(v[i], v[i] * slope);  // leaky_relu, 0 < slope < 1
      }
    };
    auto write_out = [&](int it) {  // O of tile `it` -> y (rows beyond the item's length are not stored)
      const int tile = (int)blockIdx.x + it * (int)gridDim.x;
      const int b = tile / a.tiles_per_item, t0 = (tile - b * a.tiles_per_item) * a.NB;
      const int Tb = stage_valid_len(a, b);
      const int rows = max(0, min(a.NB, Tb - t0));
      const int ytotal = rows * PPR;
      h16* yb = a.y + (long long)b * a.bstride + (long long)t0 * C;
      for (int base = 0; base < ytotal; base += 256 * 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int idx = base + i * 256 + ltid;
          const int idc = idx < ytotal ? idx : ytotal - 1;
          const int row = idc / PPR, pc = idc - row * PPR;
          h16x8 o = *reinterpret_cast<const h16x8*>(Os + row * CP + pc * 8);
          if (a.accumulate) o += *reinterpret_cast<const h16x8*>(yb + (long long)idc * 8);  // the fp16 running sum of the per-unit path
          if (idx < ytotal) *reinterpret_cast<h16x8*>(yb + (long long)idx * 8) = o;
        }
      }
    };
    // Barrier schedule per tile (mirrors the MMA waves'): per chain X, then per unit E1 and (E2 | P); Y after the last chain.
    if (my_tiles > 0) { load_window(0, cur); store_window(cur); }
    for (int it = 0; it < my_tiles; ++it) {
      for (int c = 0; c < a.nchains; ++c) {
        if (c > 0) store_window(cur);  // chain c starts from x again (A and H are free behind the previous chain's P)
        __syncthreads();  // X
        if (c == 0) {
          if (it + 1 < my_tiles) load_window(it + 1, nxt);
          if (it > 0) write_out(it - 1);
        }
        for (int u = 0; u < a.nunits; ++u) {
          __syncthreads();  // E1
          __syncthreads();  // E2 (u < last) | P (last)
        }
      }
      if (it + 1 < my_tiles) {  // the next tile's window goes down while the MMA waves finish this tile's O
#pragma unroll
        for (int i = 0; i < LB; ++i) cur[i] = nxt[i];
        store_window(cur);
      }
      __syncthreads();  // Y
    }
    if (my_tiles > 0) write_out(my_tiles - 1);
    return;
  }

  // ------------------------------ MMA waves ------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int mt0 = wm * MT;  // first 32-row output tile of this wave
  const int NFT = a.NFT;
  const h16x8* wp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) wp[i] = reinterpret_cast<const h16x8*>(a.w) + (size_t)(mt0 + i) * NFT * KB * 64 + lane;

  h16x8 ring[TD][KB][MT];
#pragma unroll
  for (int s = 0; s < TD; ++s)
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i) ring[s][u][i] = wp[i][(size_t)(s * KB + u) * 64];
  int ftn = TD;  // next flat tap to prefetch

  f32x16 acc[MT][NTW];
  // one tap: KB k-steps from ring slot S, refilled with flat tap ftn AFTER the MFMAs that read it (resblock_f16.hip).
  // The B fragments (LDS) run BD k-steps ahead of the MFMAs (bf_[0] = this step, bf_[BD] = loaded here; reads past the conv's
  // last tap wrap to its first taps and are discarded).
  // FIRST: the conv's first tap -- its first k-step takes the bias vector as the C operand (no accumulator initialisation).
#define MB_TAPROW(J) (cb_ + (size_t)((J) < ntaps ? (J) : (J) - ntaps) * ts_)
#define MB_STEPPTR(J, STEP) (MB_TAPROW((J) + (STEP) / KB) + ((STEP) % KB) * 16)
#define MB_TAPJ(S, J, FIRST)                                                                       \
  do {                                                                                             \
    const size_t nf_ = (size_t)ftn * KB;                                                           \
    _Pragma("unroll") for (int u = 0; u < KB; ++u) {                                               \
      const h16* rp_ = MB_STEPPTR(J, u + BD);                                                      \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                              \
        bf_[BD][n] = *reinterpret_cast<const h16x8*>(rp_ + n * 32 * CP);                           \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                               \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                            \
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i], bf_[0][n], (FIRST) && u == 0 ? bv_[i] : acc[i][n], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < MT; ++i) ring[S][u][i] = wp[i][(nf_ + u) * 64];        \
      /* issue order: one LDS read behind each of the first NTW MFMAs (a burst of reads from four waves at once backs the LDS */ \
      /* queue up into the issuing wave), the ring refills behind the last MFMA that reads their registers */ \
      _Pragma("unroll") for (int m = 0; m < MT * NTW; ++m) {                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                         \
        if (m < NTW) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
      }                                                                                            \
      __builtin_amdgcn_sched_group_barrier(0x020, MT, 0);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                           \
      _Pragma("unroll") for (int q = 0; q < BD; ++q)                                               \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n) bf_[q][n] = bf_[q + 1][n];                 \
    }                                                                                              \
    ftn = ftn + 1 == NFT ? 0 : ftn + 1;                                                            \
  } while (0)
  // a conv = ntaps taps (odd): the first one from ring slot S0 (0 for conv1, 1 for conv2: every conv has an odd tap count, the
  // slots alternate; TD = 1: one slot), then (ntaps - 1) / 2 pairs.  BIAS: fp32 [C] in LDS.
#define MB_CONV(S0, BASE, TAPSTEP, BIAS)                                                           \
  do {                                                                                             \
    const h16* cb_ = (BASE);                                                                       \
    const size_t ts_ = (size_t)(TAPSTEP);                                                          \
    f32x16 bv_[MT];  /* accumulator layout: channel 32 mt + 8 g + ch4 + e at element 4 g + e */    \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                 \
      _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                              \
        const f32x4 q_ = g < CG ? *reinterpret_cast<const f32x4*>((BIAS) + (mt0 + i) * 32 + 8 * g + ch4) : (f32x4)0.f; \
        bv_[i][4 * g] = q_[0]; bv_[i][4 * g + 1] = q_[1]; bv_[i][4 * g + 2] = q_[2]; bv_[i][4 * g + 3] = q_[3]; \
      }                                                                                            \
    h16x8 bf_[BD + 1][NTW];                                                                        \
    _Pragma("unroll") for (int q = 0; q < BD; ++q)                                                 \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n)                                              \
        bf_[q][n] = *reinterpret_cast<const h16x8*>(MB_STEPPTR(0, q) + n * 32 * CP);               \
    if (TD == 1) {                                                                                 \
      MB_TAPJ(0, 0, true);                                                                         \
      for (int j_ = 1; j_ < ntaps; ++j_) MB_TAPJ(0, j_, false);                                    \
    } else {                                                                                       \
      constexpr int sa_ = (S0) * (TD - 1), sb_ = (1 - (S0)) * (TD - 1);  /* (index 1 only exists for TD = 2) */ \
      MB_TAPJ(sa_, 0, true);                                                                       \
      for (int j_ = 1; j_ + 1 < ntaps; j_ += 2) {                                                  \
        MB_TAPJ(sb_, j_, false);                                                                   \
        MB_TAPJ(sa_, j_ + 1, false);                                                               \
      }                                                                                            \
    }                                                                                              \
  } while (0)

  const int lrow = wn * (NTW * 32) + (lane & 31);  // this lane's row inside a 32-row group of its wave
  const int lcol = (lane >> 5) * 8;
  const int ch4 = 4 * (lane >> 5);                 // accumulator layout: channels 32 mt + 8 g + ch4 .. + 3, position = lane & 31
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = (int)blockIdx.x + it * (int)gridDim.x;
    const int t0 = (tile % a.tiles_per_item) * a.NB;
    const int Tb = stage_valid_len(a, tile / a.tiles_per_item);
    for (int c = 0; c < a.nchains; ++c) {
      const int ntaps = a.ntaps[c];
      const int p2 = (ntaps - 1) >> 1;
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): one ring drain per chain keeps the compiler's counts in the tap loops exact
      __syncthreads();  // X: the window is down (raw in H, lrelu in A)
      h16x4 xres[MT][NTW][CG];  // the residual x of this wave's cells (its rows x its channels), for the whole chain
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
          for (int g = 0; g < CG; ++g)
            xres[i][n][g] = *reinterpret_cast<const h16x4*>(Hs + (a.PAD + lrow + n * 32) * CP + (mt0 + i) * 32 + 8 * g + ch4);
      // (the cells of the four waves -- rows x channels -- are disjoint: reading H here and overwriting it in epilogue 1 needs no barrier)
      // rows of this wave that are positions outside [0, Tb) exist only in the first and last tiles of an item: everywhere else
      // the zero-padding selects of the epilogues are skipped (wave-uniform branch)
      const int tw0 = t0 - a.Hh + wn * (NTW * 32);
      const bool interior = tw0 >= 0 && tw0 + NTW * 32 <= Tb;
      for (int u = 0; u < a.nunits; ++u) {
        const int u_ = u;
        (void)u_;
        const int d = a.dil[c][u];
        const float* b1 = bs + ((c * a.nunits + u) * 2) * C;
        const float* b2 = b1 + C;
        const bool last = u == a.nunits - 1;
        MB_SMARK(0);
        // ---------------- conv1 (dilation d) on A -> h ----------------
        MB_CONV(0, As + (a.PAD + lrow - p2 * d) * CP + lcol, d * CP, b1);
        MB_SMARK(1);
        auto epi1 = [&](auto INTERIOR) {  // two instances: interior waves carry no zero-padding selects at all
          constexpr bool interior_ = decltype(INTERIOR)::value;
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
              const int row = lrow + n * 32;
              const int t = t0 - a.Hh + row;
              const bool inside = t >= 0 && t < Tb;
#pragma unroll
              for (int g = 0; g < CG; ++g) {
                const f32x4 v = {acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]};
                h16x4 hv = __builtin_convertvector(v, h16x4);
                hv = __builtin_elementwise_max(hv, hv * slope);
                if (!interior_ && !inside) hv = (h16x4)(h16)0.f;  // conv2's zero padding
                *reinterpret_cast<h16x4*>(Hs + (a.PAD + row) * CP + (mt0 + i) * 32 + 8 * g + ch4) = hv;
              }
            }
        };
        if (interior) epi1(std::true_type{});
        else epi1(std::false_type{});
        MB_SMARK(2);
        __syncthreads();  // E1: h is complete, nobody reads A any more
        MB_SMARK(3);
        // ---------------- conv2 (dilation 1) on h ----------------
        MB_CONV(1, Hs + (a.PAD + lrow - p2) * CP + lcol, CP, b2);
        MB_SMARK(4);
        // x <- x + (conv2 + b2): the conv result rounded to fp16, then a packed fp16 add (the two roundings of the per-unit path)
        if (!last) {
          auto epi2 = [&](auto INTERIOR) {
            constexpr bool interior_ = decltype(INTERIOR)::value;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int n = 0; n < NTW; ++n) {
                const int row = lrow + n * 32;
                const int t = t0 - a.Hh + row;
                const bool inside = t >= 0 && t < Tb;
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                  const f32x4 v = {acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]};
                  const h16x4 xn = __builtin_convertvector(v, h16x4) + xres[i][n][g];
                  xres[i][n][g] = xn;
                  h16x4 av = __builtin_elementwise_max(xn, xn * slope);
                  if (!interior_ && !inside) av = (h16x4)(h16)0.f;  // conv1's zero padding
                  *reinterpret_cast<h16x4*>(As + (a.PAD + row) * CP + (mt0 + i) * 32 + 8 * g + ch4) = av;
                }
              }
          };
          if (interior) epi2(std::true_type{});
          else epi2(std::false_type{});
          MB_SMARK(5);
          __syncthreads();  // E2: lrelu(x) of the next unit is complete
          MB_SMARK(6);
        } else {
          __syncthreads();  // P: nobody reads H any more -- the support waves may lay the next window down
          MB_SMARK(5);
          const h16 osc = (h16)a.out_scale;
          // running sum of the chains' results: every read of O in flight before the first add (one LDS round trip, not twenty)
          h16x4 prev[MT][NTW][CG];
          bool mine[NTW];
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            const int orow = lrow + n * 32 - a.Hh;
            mine[n] = orow >= 0 && orow < a.NB;
            const int orc = min(max(orow, 0), a.NB - 1);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int g = 0; g < CG; ++g)
                prev[i][n][g] = c > 0 ? *reinterpret_cast<const h16x4*>(Os + orc * CP + (mt0 + i) * 32 + 8 * g + ch4) : (h16x4)(h16)0.f;
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
              const int orc = min(max(lrow + n * 32 - a.Hh, 0), a.NB - 1);
#pragma unroll
              for (int g = 0; g < CG; ++g) {
                const f32x4 v = {acc[i][n][4 * g], acc[i][n][4 * g + 1], acc[i][n][4 * g + 2], acc[i][n][4 * g + 3]};
                const h16x4 xn = __builtin_convertvector(v, h16x4) + xres[i][n][g];
                // fp16(x / num_kernels), then the fp16 running sum: the roundings of the per-unit path
                if (mine[n]) *reinterpret_cast<h16x4*>(Os + orc * CP + (mt0 + i) * 32 + 8 * g + ch4) = xn * osc + prev[i][n][g];
              }
            }
          MB_SMARK(6);
        }
      }
    }
    __syncthreads();  // Y: O holds the tile's result
  }
#undef MB_CONV
#undef MB_TAPJ
#undef MB_STEPPTR
#undef MB_TAPROW
}

struct StageGeom { int Hh, PAD, NB, NFT; size_t lds; };

template <int C, int N1>
static bool stage_geom(int nk, const int* ksizes, int nd, const int* dil /*[nk][nd]*/, StageGeom* g) {
  const int CP = C + 8;
  int Hh = 0, PAD = 0, NFT = 0;
  for (int c = 0; c < nk; ++c) {
    const int p2 = (ksizes[c] - 1) / 2;
    int h = 0;
    for (int u = 0; u < nd; ++u) { h += p2 * (dil[c * nd + u] + 1); PAD = std::max(PAD, p2 * dil[c * nd + u]); }
    Hh = std::max(Hh, h);
    NFT += 2 * nd * ksizes[c];
  }
  g->Hh = Hh; g->PAD = PAD; g->NB = N1 - 2 * Hh; g->NFT = NFT;
  if (g->NB < 64) return false;
  g->lds = ((size_t)2 * (N1 + 2 * PAD) * CP + (size_t)g->NB * CP) * sizeof(h16) + (size_t)nk * nd * 2 * C * sizeof(float);
  return g->lds <= 160 * 1024;
}

template <int C, int MT, int WN, int NTW, int TD, int BD>
static int launch_stage(ResStageK k, const StageGeom& g, int batch, hipStream_t s) {
  k.Hh = g.Hh; k.PAD = g.PAD; k.NB = g.NB; k.NFT = g.NFT;
  k.tiles_per_item = cdiv(k.T, k.NB);
  k.n_tiles = k.tiles_per_item * batch;
  static bool attr_done = false;
  if (!attr_done) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_stage_f16_kernel<C, MT, WN, NTW, TD, BD>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
               ? prop.multiProcessorCount : 256;
  }
#ifdef MB_STAGE_TRACE_BUILD
  if (diag_int("stage_fold")) k.NFT = 4;  // diagnostics: the weight stream folded onto its first four taps (L1-resident; wrong results)
  static unsigned long long* d_trace = nullptr;
  std::string trace_file;
  const char* trace_path = diag_str("stage_trace", &trace_file) ? trace_file.c_str() : nullptr;
  if (trace_path) {
    if (!d_trace) MB_HIP(hipMalloc((void**)&d_trace, 128 * sizeof(unsigned long long)));
    MB_HIP(hipMemsetAsync(d_trace, 0, 128 * sizeof(unsigned long long), s));
    k.trace = d_trace;
  }
#endif
  hipLaunchKernelGGL((resblock_stage_f16_kernel<C, MT, WN, NTW, TD, BD>), dim3(std::min(k.n_tiles, n_cu)), dim3(512), g.lds, s, k);
  MB_HIP(hipGetLastError());
#ifdef MB_STAGE_TRACE_BUILD
  if (trace_path) {  // diagnostics: "C NTW chains units tiles : marks[chain][unit][8]" per launch
    unsigned long long h[128];
    MB_HIP(hipStreamSynchronize(s));
    MB_HIP(hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "a")) {
      fprintf(f, "%d %d %d %d %d :", C, NTW, k.nchains, k.nunits, k.n_tiles);
      for (int i = 0; i < 128; ++i) fprintf(f, " %llu", h[i]);
      fprintf(f, "\n");
      fclose(f);
    }
  }
#endif
  return MB_OK;
}

static bool stage_shape_ok(int channels, int nk, const int* ksizes, int nd, const int* dil) {
  if (!(channels == 16 || channels == 32 || channels == 64 || channels == 128) || nk < 1 || nk > STAGE_MAX_CHAINS || nd < 1 ||
      nd > STAGE_MAX_UNITS || !ksizes || !dil)
    return false;
  for (int c = 0; c < nk; ++c) {
    if (ksizes[c] < 3 || (ksizes[c] & 1) == 0) return false;
    for (int u = 0; u < nd; ++u)
      if (dil[c * nd + u] < 1) return false;
  }
  return true;
}

// the instance of a channel count: window rows N1 (LDS holds two windows + the result rows)
static bool stage_geom_c(int channels, int nk, const int* ksizes, int nd, const int* dil, StageGeom* g) {
  switch (channels) {
    case 16: return stage_geom<16, 768>(nk, ksizes, nd, dil, g);
    case 32: return stage_geom<32, 640>(nk, ksizes, nd, dil, g);
    case 64: return stage_geom<64, 256>(nk, ksizes, nd, dil, g);
    default: return stage_geom<128, 128>(nk, ksizes, nd, dil, g);
  }
}

}  // namespace mb

using namespace mb;

extern "C" int mb_resblock_stage_f16_supported(int channels, int num_kernels, const int* ksizes, int num_dilations,
                                               const int* dilations) {
  if (!stage_shape_ok(channels, num_kernels, ksizes, num_dilations, dilations)) return 0;
  StageGeom g;
  return stage_geom_c(channels, num_kernels, ksizes, num_dilations, dilations, &g) ? 1 : 0;
}

// useful rows of a tile / window rows of the instance (0 = unsupported): what a caller weighs against the per-unit launches
extern "C" float mb_resblock_stage_f16_efficiency(int channels, int num_kernels, const int* ksizes, int num_dilations,
                                                  const int* dilations) {
  if (!stage_shape_ok(channels, num_kernels, ksizes, num_dilations, dilations)) return 0.f;
  StageGeom g;
  if (!stage_geom_c(channels, num_kernels, ksizes, num_dilations, dilations, &g)) return 0.f;
  return (float)g.NB / (float)(g.NB + 2 * g.Hh);
}

extern "C" size_t mb_resblock_stage_f16_packed_halves(int channels, int num_kernels, const int* ksizes, int num_dilations) {
  if (!(channels == 16 || channels == 32 || channels == 64 || channels == 128) || !ksizes || num_kernels < 1 || num_dilations < 1) return 0;
  size_t taps = 0;
  for (int c = 0; c < num_kernels; ++c) taps += (size_t)2 * num_dilations * ksizes[c];
  return (size_t)((channels + 31) / 32) * taps * (channels / 16) * 512;  // per 32-row output tile: 512 halves per (tap, k-step) fragment
}

// h_w1[c * nd + u], h_w2[c * nd + u]: fp32 torch Conv1d weights [C][C][k_c] (weight norm folded) of chain c, unit u.
extern "C" int mb_resblock_stage_f16_pack(const float* const* h_w1, const float* const* h_w2, int channels, int num_kernels,
                                          const int* ksizes, int num_dilations, uint16_t* h_packed) {
  MB_REQUIRE(h_w1 && h_w2 && h_packed && ksizes, "resblock_stage_f16_pack: null pointer");
  MB_REQUIRE(channels == 16 || channels == 32 || channels == 64 || channels == 128, "resblock_stage_f16_pack: C=%d unsupported", channels);
  const int C = channels, KB = C / 16, MTT = (C + 31) / 32;
  h16* out = reinterpret_cast<h16*>(h_packed);
  size_t o = 0;
  for (int mt = 0; mt < MTT; ++mt)  // one stream per 32-row output tile, each in consumption order
    for (int c = 0; c < num_kernels; ++c)
      for (int u = 0; u < num_dilations; ++u)
        for (int ph = 0; ph < 2; ++ph) {
          const float* w = ph ? h_w2[c * num_dilations + u] : h_w1[c * num_dilations + u];
          MB_REQUIRE(w, "resblock_stage_f16_pack: null weight (chain %d unit %d)", c, u);
          const int k = ksizes[c];
          for (int j = 0; j < k; ++j)
            for (int kb = 0; kb < KB; ++kb)
              for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                  // A fragment of v_mfma_f32_32x32x16_f16: lane l holds A[m = l&31][k = 8*(l>>5) + e]
                  const int co = mt * 32 + (lane & 31);
                  const int ci = kb * 16 + (lane >> 5) * 8 + e;
                  out[o++] = co < C ? (h16)w[((size_t)co * C + ci) * k + j] : (h16)0.f;
                }
        }
  return MB_OK;
}

extern "C" int mb_resblock_stage_f16(const mb_resblock_stage_f16_args* a, mb_stream_t stream) {
  MB_REQUIRE(a && a->d_x && a->d_y && a->d_wpacked && a->d_bias, "resblock_stage_f16: null pointer");
  MB_REQUIRE(a->d_x != a->d_y, "resblock_stage_f16: in-place is not supported (tiles read their neighbours' halo)");
  MB_REQUIRE(a->num_kernels >= 1 && a->num_kernels <= STAGE_MAX_CHAINS && a->num_dilations >= 1 && a->num_dilations <= STAGE_MAX_UNITS,
             "resblock_stage_f16: kernels=%d dilations=%d unsupported", a->num_kernels, a->num_dilations);
  int dil[STAGE_MAX_CHAINS * STAGE_MAX_UNITS];
  for (int c = 0; c < a->num_kernels; ++c)
    for (int u = 0; u < a->num_dilations; ++u) dil[c * a->num_dilations + u] = a->dilation[c][u];
  MB_REQUIRE(stage_shape_ok(a->channels, a->num_kernels, a->ksize, a->num_dilations, dil),
             "resblock_stage_f16: C=%d kernels=%d dilations=%d unsupported", a->channels, a->num_kernels, a->num_dilations);
  MB_REQUIRE(a->slope > 0.f && a->slope < 1.f, "resblock_stage_f16: leaky_relu slope must be in (0,1)");
  if (a->batch <= 0 || a->t <= 0) return MB_OK;
  ResStageK k;
  memset(&k, 0, sizeof(k));
  k.x = reinterpret_cast<const h16*>(a->d_x); k.y = reinterpret_cast<h16*>(a->d_y);
  k.w = reinterpret_cast<const h16*>(a->d_wpacked); k.bias = a->d_bias;
  k.bstride = (long long)a->t * a->channels;
  k.T = a->t; k.nchains = a->num_kernels; k.nunits = a->num_dilations;
  for (int c = 0; c < a->num_kernels; ++c) {
    k.ntaps[c] = a->ksize[c];
    for (int u = 0; u < a->num_dilations; ++u) k.dil[c][u] = a->dilation[c][u];
  }
  k.slope = a->slope; k.out_scale = a->out_scale == 0.f ? 1.f / (float)a->num_kernels : a->out_scale;
  k.valid = a->d_valid; k.valid_mul = a->valid_mul > 0 ? a->valid_mul : 1;
  k.accumulate = a->accumulate;
  StageGeom g;
  hipStream_t s = (hipStream_t)stream;
  MB_REQUIRE(stage_geom_c(a->channels, a->num_kernels, a->ksize, a->num_dilations, dil, &g), "resblock_stage_f16: the tile does not fit LDS");
  switch (a->channels) {
    case 16: return launch_stage<16, 1, 4, 6, 2, 1>(k, g, a->batch, s);
    case 32: return launch_stage<32, 1, 4, 5, 2, 2>(k, g, a->batch, s);
    case 64: return launch_stage<64, 2, 4, 2, 2, 2>(k, g, a->batch, s);
    default: return launch_stage<128, 2, 2, 2, 1, 1>(k, g, a->batch, s);
  }
}
