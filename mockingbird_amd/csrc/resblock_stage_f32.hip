// Round 4: the ResBlock group of an upsampling stage of the GAN vocoders at the REFERENCE's precision (fp32 tensors in HBM, fp32-grade
// results) in ONE launch -- resblock_stage_f16.hip's launch structure on error-compensated operands.
//
//   y = (1 / num_kernels) * sum_j ResBlock_j(x),   ResBlock_j:  for u in 0..nd-1:  x <- x + conv2_u(lrelu(conv1_u(lrelu(x)) + b1)) + b2
//
// = the inner loop of Generator.forward over self.resblocks (models/vocoder/hifigan/models.py:139-145, models/vocoder/fregan/generator.py:150-157)
// with ResBlock1.forward (hifigan/models.py:39-46, fregan/generator.py:43-50).  On the fp32-result path every conv used to be its own
// conv1d_split_kernel launch: 18 launches per stage, each staging (fp32 -> fp16 hi / lo) the same window again and moving the fp32
// tensor through HBM 2-3 times -- 3.6 ms for the 32-channel stage and 4.2 ms for the 64-channel stage of HiFi-GAN 32 x 200 frames
// (91 / 156 TFLOP/s algorithmic of the 833 this scheme allows; profiles/r04_hifigan_f32_before.txt).  Here a tile's window is read
// ONCE (fp32, channel-major, as the reference lays tensors out), split ONCE per chain start, and the units of a chain run back to back
// on LDS-resident operands; the residual x, the chains' running sum and the result live in the registers of the MMA wave that owns the
// cells, which also stores the result (the accumulator layout has 32 consecutive positions of a channel in 32 lanes: coalesced rows).
//
// Arithmetic = conv1d.hip's split scheme: w 2^s = wh + wl (s per conv), a = ah + 2^-11 al (the residual stored scaled: an fp16 normal
// whenever ah is), acc += wl.ah + ws.al + wh.ah with ws = wh 2^-11, fp32 accumulate, v = acc 2^-s + bias in fp32; every intermediate
// that a conv reads (lrelu(x), h) is re-split from its fp32 value, the residual chain x <- x + conv2(..) stays fp32 in registers.
//
// Layout: fp32 [B][C][T] in and out.  LDS (halves): Ah | Al [PAD + N1 + PAD][C + 8]  lrelu(x) of the running unit, zero outside [0, T);
//                                                   Hh | Hl                          h = lrelu(conv1 + b1)
// A tile = N1 = 32 * WN * NTW window rows; row i <-> position t0 - Hh + i (Hh = the widest chain's reach per side); the NB = N1 - 2 Hh rows
// in the middle are the tile's result (resblock_stage_f16.hip).  C = 32 takes N1 = 256 (the whole group of HiFi-GAN's last stage: NB = 136),
// C = 64 takes N1 = 128 (single ResBlocks / single units, accumulating into y): two 32-row tiles per wave -- with three (N1 = 384 / 192, which
// LDS would hold) the kernel needs more than the 512 registers a wave can have and reloads its residual from scratch one word per round trip.
// FOUR waves, one per SIMD, each with the whole 512-register budget of its SIMD: the fp32 residual of a wave's cells (48 VGPRs), its
// accumulators (48), two taps of weight fragments (48-96), a k-step of B fragments ahead (48) and the fp32 window of the current and the
// next tile (2 x 48) do not fit the 256 registers an 8-wave workgroup leaves a wave -- the first version (resblock_stage_f16.hip's two
// roles of four waves) spilled 200-350 VGPRs around every chain start, its 48 residual loads and 48 result read-modify-writes serialised
// behind the spill traffic (vmcnt(0) after every load), and ran SLOWER than the per-conv launches (15.8 against 13.6 ms).  Every wave
// runs the MFMA loops on its rows (B operands from LDS, weights through a register ring over ONE circular stream in consumption order) and
// lays its quarter of the window down (lrelu, hi / lo) at every chain start; the next tile's fp32 window is requested at the start of the
// tile and arrives under the first chain.
#include <atomic>
#include <cmath>
#include <type_traits>
#include "common.h"

namespace mb {

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

constexpr int S32_MAX_CHAINS = 4, S32_MAX_UNITS = 4;

struct ResStage32K {
  const float* x; float* y; const h16* w;
  const float* bias;  // [chain][unit][2][C] biases, then [chain][unit][2] unscale factors 2^-s
  long long bstride;  // floats per batch item (C * T)
  int T, nchains, nunits;
  int ntaps[S32_MAX_CHAINS];
  int dil[S32_MAX_CHAINS][S32_MAX_UNITS];
  int Hh, PAD, NB, tiles_per_item, n_tiles, NFT;
  float slope, out_scale;
  int accumulate;
  const int* valid; int valid_mul;
};

__device__ __forceinline__ int s32_valid_len(const ResStage32K& a, int b) {
  if (!a.valid) return a.T;
  return min(a.T, a.valid[b] * a.valid_mul);
}

// uniform base + 32-bit BYTE offset of the lane: the address form global_load / global_store take as (SGPR pair, one VGPR) -- element
// offsets made the compiler build a 64-bit address per access (zext(off) << 2), two VGPRs each, 48 of them live at once
__device__ __forceinline__ float s32_ld(const float* base, const unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void s32_st(float* base, const unsigned byte_off, const float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// fp32 x 8 -> fp16 hi and scaled fp16 residual (conv1d.hip split_store2)
__device__ __forceinline__ void s32_split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {  // (packed, one clamp per value: common.h split_pair; bit for bit the scalar form's halves)
    mb_h2 h, l;
    split_pair(v[e], v[e + 1], h, l);
    hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
  }
}
__device__ __forceinline__ void s32_split4(const float (&v)[4], h16x4& hi, h16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    mb_h2 h, l;
    split_pair(v[e], v[e + 1], h, l);
    hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
  }
}

// C channels = WM x MT 32-row output tiles (WM = 4 / WN waves along the channels), WN waves along the rows with NTW 32-row tiles each.
// TD = taps of weight prefetch in flight (2: slots alternate between the odd-length convs; 1: one slot).  Several chains per launch
// accumulate into y one after the other (the owner lane re-reads what it wrote: a running sum in registers next to the accumulators and
// the fp32 residual spilled 350 VGPRs -- the extra tensor passes are 0.15 ms of the 32-channel stage).
template <int C, int MT, int WN, int NTW, int TD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void resblock_stage_f32_kernel(ResStage32K a) {
  constexpr int WM = 4 / WN;
  static_assert(WM * MT * 32 == C, "the four MMA waves cover all channels");
  constexpr int KB = C / 16;  // k-steps per tap
  constexpr int CP = C + 8;   // LDS row stride in halves (odd multiple of 16 B)
  constexpr int N1 = WN * NTW * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int R = N1 + 2 * a.PAD;
  const int ARR = R * CP;  // halves per array
  h16* Ah = reinterpret_cast<h16*>(lds_raw);
  h16* Al = Ah + ARR;
  h16* Hhi = Al + ARR;
  h16* Hlo = Hhi + ARR;
  float* bs = reinterpret_cast<float*>(Hlo + ARR);          // [chain][unit][2][C]
  float* uss = bs + a.nchains * a.nunits * 2 * C;           // [chain][unit][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int my_tiles = (a.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const float slope = a.slope;

  {  // zero the four arrays once (the PAD rows are never written again), stage biases and unscale factors
    h16x8* z = reinterpret_cast<h16x8*>(lds_raw);
    const int n16 = 4 * ARR / 8;
    for (int i = tid; i < n16; i += 256) z[i] = (h16x8)(h16)0.f;
    const int nb = a.nchains * a.nunits * 2 * (C + 1);
    for (int i = tid; i < nb; i += 256) bs[i] = a.bias[i];
  }
  __syncthreads();  // Z

  // ------------------------------ the window (all four waves) ------------------------------
  constexpr int PPR = C / 8;              // 8-channel pieces per row
  constexpr int LB = N1 * PPR / 256;      // pieces per lane of one window
  static_assert(N1 * PPR % 256 == 0 && N1 % 64 == 0, "window pieces divide over the support lanes");
  const int ltid = tid;  // all four waves lay the window down
  // (addresses = a uniform base per channel-in-piece (SGPR pair) + ONE 32-bit lane offset per piece: 48 loads in flight with their own
  //  64-bit addresses spilled 112 VGPRs)
  auto load_window = [&](int it, float (&v)[LB][8]) __attribute__((always_inline)) {
    const int tile = (int)blockIdx.x + it * (int)gridDim.x;
    const int b = tile / a.tiles_per_item, t0 = (tile - b * a.tiles_per_item) * a.NB;
    const int Tb = s32_valid_len(a, b);
    const float* xb = a.x + (long long)b * a.bstride;
    unsigned off[LB];
    bool inside[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int idx = i * 256 + ltid;
      const int pc = idx / N1, row = idx - pc * N1;  // consecutive lanes = consecutive positions of one channel: coalesced rows
      const int t = t0 - a.Hh + row;
      const int tc = min(max(t, 0), a.T - 1);  // clamped: the load is always legal, the value is selected
      inside[i] = t >= 0 && t < Tb;
      off[i] = ((unsigned)(pc * 8) * (unsigned)a.T + (unsigned)tc) * 4u;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float* xe = xb + (size_t)e * a.T;  // uniform
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const float ld = s32_ld(xe, off[i]);
        v[i][e] = inside[i] ? ld : 0.f;
      }
    }
  };
  auto store_window = [&](const float (&v)[LB][8]) __attribute__((always_inline)) {  // lrelu(x) -> Ah / Al
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int idx = i * 256 + ltid;
      const int pc = idx / N1, row = idx - pc * N1;
      float l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) l[e] = v[i][e] > 0.f ? v[i][e] : v[i][e] * slope;
      h16x8 hi, lo;
      s32_split8(l, hi, lo);
      const int o = (a.PAD + row) * CP + pc * 8;
      *reinterpret_cast<h16x8*>(Ah + o) = hi;
      *reinterpret_cast<h16x8*>(Al + o) = lo;
    }
  };
  // ------------------------------ the MFMA loops ------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int mt0 = wm * MT;  // first 32-row output tile of this wave
  const int NFT = a.NFT;
  const h16x8* wp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) wp[i] = reinterpret_cast<const h16x8*>(a.w) + (size_t)(mt0 + i) * NFT * KB * 3 * 64 + lane;

  h16x8 ring[TD][KB][MT][3];  // [slot][k-step][tile][hi | lo | hi 2^-11]
#pragma unroll
  for (int s = 0; s < TD; ++s)
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) ring[s][u][i][p] = wp[i][((size_t)(s * KB + u) * 3 + p) * 64];
  int ftn = TD;  // next flat tap to prefetch

  f32x16 acc[MT][NTW];
  // one tap: KB k-steps from ring slot S, refilled with flat tap ftn AFTER the MFMAs that read it.  The B fragments (LDS) run one
  // k-step ahead of the MFMAs (reads past the conv's last tap wrap to its first taps and are discarded).
#define S32_TAPROW(J) ((size_t)((J) < ntaps ? (J) : (J) - ntaps) * ts_)
#define S32_STEPOFF(J, STEP) (S32_TAPROW((J) + (STEP) / KB) + ((STEP) % KB) * 16)
#define S32_TAPJ(S, J, FIRST)                                                                                        \
  do {                                                                                                               \
    const size_t nf_ = (size_t)ftn * KB;                                                                             \
    _Pragma("unroll") for (int u = 0; u < KB; ++u) {                                                                 \
      const size_t ro_ = S32_STEPOFF(J, u + 1);                                                                      \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                                              \
        bfh_[1][n] = *reinterpret_cast<const h16x8*>(cbh_ + ro_ + n * 32 * CP);                                      \
        bfl_[1][n] = *reinterpret_cast<const h16x8*>(cbh_ + ARR + ro_ + n * 32 * CP);                                \
      }                                                                                                              \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                 \
        _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                                            \
          f32x16 c_ = acc[i][n];                                                                                     \
          if ((FIRST) && u == 0) { _Pragma("unroll") for (int q = 0; q < 16; ++q) c_[q] = 0.f; }                     \
          c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i][1], bfh_[0][n], c_, 0, 0, 0);                    \
          c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i][2], bfl_[0][n], c_, 0, 0, 0);                    \
          acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[S][u][i][0], bfh_[0][n], c_, 0, 0, 0);             \
        }                                                                                                            \
      _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                                 \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) ring[S][u][i][p] = wp[i][((nf_ + u) * 3 + p) * 64];            \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
      _Pragma("unroll") for (int n = 0; n < NTW; ++n) { bfh_[0][n] = bfh_[1][n]; bfl_[0][n] = bfl_[1][n]; }          \
    }                                                                                                                \
    ftn = ftn + 1 == NFT ? 0 : ftn + 1;                                                                              \
  } while (0)
  // a conv = ntaps taps (odd): the first one from ring slot S0 (0 for conv1, 1 for conv2: every conv has an odd tap count, the slots
  // alternate; TD = 1: one slot), then (ntaps - 1) / 2 pairs.  BASE: the hi array at (this lane's row of tap 0, its k-quarter).
#define S32_CONV(S0, BASE, TAPSTEP)                                                                                  \
  do {                                                                                                               \
    const h16* cbh_ = (BASE);                                                                                        \
    const size_t ts_ = (size_t)(TAPSTEP);                                                                            \
    h16x8 bfh_[2][NTW], bfl_[2][NTW];                                                                                \
    _Pragma("unroll") for (int n = 0; n < NTW; ++n) {                                                                \
      bfh_[0][n] = *reinterpret_cast<const h16x8*>(cbh_ + n * 32 * CP);                                              \
      bfl_[0][n] = *reinterpret_cast<const h16x8*>(cbh_ + ARR + n * 32 * CP);                                        \
    }                                                                                                                \
    if (TD == 1) {                                                                                                   \
      S32_TAPJ(0, 0, true);                                                                                          \
      for (int j_ = 1; j_ < ntaps; ++j_) S32_TAPJ(0, j_, false);                                                     \
    } else {                                                                                                         \
      constexpr int sa_ = (S0) * (TD - 1), sb_ = (1 - (S0)) * (TD - 1);                                              \
      S32_TAPJ(sa_, 0, true);                                                                                        \
      for (int j_ = 1; j_ + 1 < ntaps; j_ += 2) {                                                                    \
        S32_TAPJ(sb_, j_, false);                                                                                    \
        S32_TAPJ(sa_, j_ + 1, false);                                                                                \
      }                                                                                                              \
    }                                                                                                                \
  } while (0)

  const int lrow = wn * (NTW * 32) + (lane & 31);  // this lane's row inside a 32-row group of its wave
  const int lcol = (lane >> 5) * 8;
  const int ch4 = 4 * (lane >> 5);                 // accumulator layout: channels 32 mt + 8 g + ch4 .. + 3 at elements 4 g .. 4 g + 3, position = lane & 31
  for (int it = 0; it < my_tiles; ++it) {
    const int tile = (int)blockIdx.x + it * (int)gridDim.x;
    const int bi = tile / a.tiles_per_item;
    const int t0 = (tile - bi * a.tiles_per_item) * a.NB;
    const int Tb = s32_valid_len(a, bi);
    const float* xb = a.x + (long long)bi * a.bstride;
    float* yb = a.y + (long long)bi * a.bstride;
    for (int c = 0; c < a.nchains; ++c) {
      const int ntaps = a.ntaps[c];
      const int p2 = (ntaps - 1) >> 1;
      {  // every chain starts from x: the tile's fp32 window (L2-hot after the first chain) -> lrelu -> hi / lo rows.  A is free behind the
         // previous chain's (tile's) last barrier.  (Keeping the window -- and the next tile's -- in registers across the chains cost 96
         // VGPRs that the allocator spilled to scratch and reloaded one word per round trip: 50-100 us per chain start.)
        float win[LB][8];
        load_window(it, win);
        store_window(win);
      }
      __syncthreads();  // X: lrelu(x) is down (Ah / Al)
      // the residual x of this wave's cells (its rows x its channels), fp32, for the whole chain: 32 consecutive positions of a channel
      // per half wave and load
      float xres[MT][NTW][16];  // (scalars, not f32x16: element-wise updates of vector values kept several copies alive)
      unsigned xoff[MT][NTW];  // lane offset of (first channel of the lane's group, its position); element q adds a uniform (8 (q >> 2) + (q & 3)) T
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int t = t0 - a.Hh + lrow + n * 32;
          const int tc = min(max(t, 0), a.T - 1);
          xoff[i][n] = ((unsigned)((mt0 + i) * 32 + ch4) * (unsigned)a.T + (unsigned)tc) * 4u;
        }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float* xq = xb + (size_t)(8 * (q >> 2) + (q & 3)) * a.T;  // uniform
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            const int t = t0 - a.Hh + lrow + n * 32;
            const float ld = s32_ld(xq, xoff[i][n]);
            xres[i][n][q] = (t >= 0 && t < Tb) ? ld : 0.f;
          }
      }
      for (int u = 0; u < a.nunits; ++u) {
        const int d = a.dil[c][u];
        const float* b1 = bs + ((c * a.nunits + u) * 2) * C;
        const float* b2 = b1 + C;
        const float us1 = uss[(c * a.nunits + u) * 2], us2 = uss[(c * a.nunits + u) * 2 + 1];
        const bool last = u == a.nunits - 1;
        // ---------------- conv1 (dilation d) on A -> h = lrelu(conv1 2^-s + b1), zero outside [0, Tb) (conv2's padding) ----------------
        S32_CONV(0, Ah + (a.PAD + lrow - p2 * d) * CP + lcol, d * CP);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            const int row = lrow + n * 32;
            const int t = t0 - a.Hh + row;
            const bool inside = t >= 0 && t < Tb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 bq = *reinterpret_cast<const f32x4*>(b1 + (mt0 + i) * 32 + 8 * g + ch4);
              float hv[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v = fmaf(acc[i][n][4 * g + e], us1, bq[e]);
                v = v > 0.f ? v : v * slope;
                hv[e] = inside ? v : 0.f;
              }
              h16x4 hi, lo;
              s32_split4(hv, hi, lo);
              const int o = (a.PAD + row) * CP + (mt0 + i) * 32 + 8 * g + ch4;
              *reinterpret_cast<h16x4*>(Hhi + o) = hi;
              *reinterpret_cast<h16x4*>(Hlo + o) = lo;
            }
            __builtin_amdgcn_sched_barrier(0);  // one 32-row tile at a time: interleaved, the epilogues of three tiles held > 512 values
          }
        __syncthreads();  // E1: h is complete, nobody reads A any more
        // ---------------- conv2 (dilation 1) on h; x <- x + conv2 2^-s + b2 ----------------
        S32_CONV(1, Hhi + (a.PAD + lrow - p2) * CP + lcol, CP);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            const int row = lrow + n * 32;
            const int t = t0 - a.Hh + row;
            const bool inside = t >= 0 && t < Tb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 bq = *reinterpret_cast<const f32x4*>(b2 + (mt0 + i) * 32 + 8 * g + ch4);
              float av[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float xn = xres[i][n][4 * g + e] + fmaf(acc[i][n][4 * g + e], us2, bq[e]);
                xres[i][n][4 * g + e] = xn;
                const float l = xn > 0.f ? xn : xn * slope;
                av[e] = inside ? l : 0.f;  // conv1's zero padding
              }
              if (!last) {
                h16x4 hi, lo;
                s32_split4(av, hi, lo);
                const int o = (a.PAD + row) * CP + (mt0 + i) * 32 + 8 * g + ch4;
                *reinterpret_cast<h16x4*>(Ah + o) = hi;
                *reinterpret_cast<h16x4*>(Al + o) = lo;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        __syncthreads();  // E2: lrelu(x) of the next unit is complete | P (last): nobody reads H or A any more
      }
      {  // this chain's share of the tile's result: rows [Hh, Hh + NB) of the window that are positions of the item.  Chains after the
         // first (and launches that accumulate) add to what y holds: every read is in flight before the first add.
        const bool acc_y = a.accumulate || c > 0;
        unsigned yoff[MT][NTW];
        bool mine[MT][NTW];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            const int orow = lrow + n * 32 - a.Hh;
            const int t = t0 + orow;
            mine[i][n] = orow >= 0 && orow < a.NB && t < Tb;
            yoff[i][n] = ((unsigned)((mt0 + i) * 32 + ch4) * (unsigned)a.T + (unsigned)min(max(t, 0), a.T - 1)) * 4u;
          }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          float* yq = yb + (size_t)(8 * (q >> 2) + (q & 3)) * a.T;  // uniform
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
              const float v = xres[i][n][q] * a.out_scale;
              float* yp = reinterpret_cast<float*>(reinterpret_cast<char*>(yq) + yoff[i][n]);
              if (mine[i][n]) {
                // later chains (and accumulating launches) ADD: a returnless fp32 atomic -- one lane per element, so the sum is the
                // sequential one, and nothing has to come back (a read-modify-write held 48 more registers and a round trip per chain)
                if (acc_y) unsafeAtomicAdd(yp, v);
                else *yp = v;
              }
            }
        }
      }
    }
    // (no barrier here: the next tile's first window store sits behind this tile's last E2 / P barrier)
  }
#undef S32_CONV
#undef S32_TAPJ
#undef S32_STEPOFF
#undef S32_TAPROW
}

struct Stage32Geom { int Hh, PAD, NB, NFT; size_t lds; };

template <int C, int N1>
static bool stage32_geom(int nk, const int* ksizes, int nd, const int* dil /*[nk][nd]*/, Stage32Geom* g) {
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
  if (g->NB < 32) return false;
  g->lds = (size_t)4 * (N1 + 2 * PAD) * CP * sizeof(h16) + (size_t)nk * nd * 2 * (C + 1) * sizeof(float);
  return g->lds <= 160 * 1024;
}

template <int C, int MT, int WN, int NTW, int TD>
static int launch_stage32(ResStage32K k, const Stage32Geom& g, int batch, hipStream_t s) {
  k.Hh = g.Hh; k.PAD = g.PAD; k.NB = g.NB; k.NFT = g.NFT;
  k.tiles_per_item = cdiv(k.T, k.NB);
  k.n_tiles = k.tiles_per_item * batch;
  static std::atomic<unsigned long long> attr_done{0};  // bit d = attribute set on device d (per kernel instance)
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  static int n_cu[64] = {};
  if (!bit || !(attr_done.load(std::memory_order_acquire) & bit)) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_stage_f32_kernel<C, MT, WN, NTW, TD>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipDeviceProp_t prop;
    MB_HIP(hipGetDeviceProperties(&prop, dev));
    if (bit) n_cu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int cus = bit && n_cu[dev] ? n_cu[dev] : 256;
  hipLaunchKernelGGL((resblock_stage_f32_kernel<C, MT, WN, NTW, TD>), dim3(std::min(k.n_tiles, cus)), dim3(256), g.lds, s, k);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

static bool stage32_shape_ok(int channels, int nk, const int* ksizes, int nd, const int* dil) {
  if (!(channels == 32 || channels == 64) || nk < 1 || nk > S32_MAX_CHAINS || nd < 1 || nd > S32_MAX_UNITS || !ksizes || !dil) return false;
  for (int c = 0; c < nk; ++c) {
    if (ksizes[c] < 3 || (ksizes[c] & 1) == 0) return false;
    for (int u = 0; u < nd; ++u)
      if (dil[c * nd + u] < 1) return false;
  }
  return true;
}
static bool stage32_geom_c(int channels, int nk, const int* ksizes, int nd, const int* dil, Stage32Geom* g) {
  return channels == 32 ? stage32_geom<32, 256>(nk, ksizes, nd, dil, g) : stage32_geom<64, 128>(nk, ksizes, nd, dil, g);
}

}  // namespace mb

using namespace mb;

extern "C" int mb_resblock_stage_f32_supported(int channels, int num_kernels, const int* ksizes, int num_dilations, const int* dilations) {
  if (!stage32_shape_ok(channels, num_kernels, ksizes, num_dilations, dilations)) return 0;
  Stage32Geom g;
  return stage32_geom_c(channels, num_kernels, ksizes, num_dilations, dilations, &g) ? 1 : 0;
}

extern "C" float mb_resblock_stage_f32_efficiency(int channels, int num_kernels, const int* ksizes, int num_dilations, const int* dilations) {
  if (!stage32_shape_ok(channels, num_kernels, ksizes, num_dilations, dilations)) return 0.f;
  Stage32Geom g;
  if (!stage32_geom_c(channels, num_kernels, ksizes, num_dilations, dilations, &g)) return 0.f;
  return (float)g.NB / (float)(g.NB + 2 * g.Hh);
}

extern "C" size_t mb_resblock_stage_f32_packed_halves(int channels, int num_kernels, const int* ksizes, int num_dilations) {
  if (!(channels == 32 || channels == 64) || !ksizes || num_kernels < 1 || num_dilations < 1) return 0;
  size_t taps = 0;
  for (int c = 0; c < num_kernels; ++c) taps += (size_t)2 * num_dilations * ksizes[c];
  return (size_t)(channels / 32) * taps * (channels / 16) * 3 * 512;  // per 32-row output tile: 3 x 512 halves per (tap, k-step)
}

// h_w1[c * nd + u], h_w2[c * nd + u]: fp32 torch Conv1d weights [C][C][k_c] (weight norm folded) of chain c, unit u.
// h_unscale [num_kernels][num_dilations][2]: 2^-s of every conv (conv1, conv2 of each unit), appended to the bias block by the caller.
extern "C" int mb_resblock_stage_f32_pack(const float* const* h_w1, const float* const* h_w2, int channels, int num_kernels, const int* ksizes,
                                          int num_dilations, uint16_t* h_packed, float* h_unscale) {
  MB_REQUIRE(h_w1 && h_w2 && h_packed && ksizes && h_unscale, "resblock_stage_f32_pack: null pointer");
  MB_REQUIRE(channels == 32 || channels == 64, "resblock_stage_f32_pack: C=%d unsupported", channels);
  const int C = channels, KB = C / 16, MTT = C / 32;
  h16* out = reinterpret_cast<h16*>(h_packed);
  size_t o = 0;
  for (int mt = 0; mt < MTT; ++mt)  // one stream per 32-row output tile, each in consumption order
    for (int c = 0; c < num_kernels; ++c)
      for (int u = 0; u < num_dilations; ++u)
        for (int ph = 0; ph < 2; ++ph) {
          const float* w = ph ? h_w2[c * num_dilations + u] : h_w1[c * num_dilations + u];
          MB_REQUIRE(w, "resblock_stage_f32_pack: null weight (chain %d unit %d)", c, u);
          const int k = ksizes[c];
          float wmax = 0.f;
          for (size_t q = 0; q < (size_t)C * C * k; ++q) wmax = std::max(wmax, std::fabs(w[q]));
          int sexp = 0;
          if (wmax > 0.f && std::isfinite(wmax)) {
            int e2;
            std::frexp(wmax, &e2);
            sexp = std::max(-24, std::min(40, 14 - e2));  // wmax * 2^sexp in [2^13, 2^14)
          }
          const float scale = std::ldexp(1.f, sexp);
          h_unscale[(c * num_dilations + u) * 2 + ph] = std::ldexp(1.f, -sexp);
          for (int j = 0; j < k; ++j)
            for (int kb = 0; kb < KB; ++kb)
              for (int part = 0; part < 3; ++part)
                for (int lane = 0; lane < 64; ++lane)
                  for (int e = 0; e < 8; ++e) {
                    // A fragment of v_mfma_f32_32x32x16_f16: lane l holds A[m = l & 31][k = 8 * (l >> 5) + e]
                    const int co = mt * 32 + (lane & 31);
                    const int ci = kb * 16 + (lane >> 5) * 8 + e;
                    const float v = w[((size_t)co * C + ci) * k + j] * scale;
                    const h16 hi = (h16)v;
                    out[o++] = part == 0 ? hi : part == 1 ? (h16)(v - (float)hi) : (h16)((float)hi * (1.f / 2048.f));
                  }
        }
  return MB_OK;
}

// args: the struct of mb_resblock_stage_f16 with d_x / d_y = fp32 [B][channels][t] (the reference's layout) and d_bias =
// [num_kernels][num_dilations][2][channels] biases followed by the [num_kernels][num_dilations][2] unscale factors of the pack.
extern "C" int mb_resblock_stage_f32(const mb_resblock_stage_f16_args* a, mb_stream_t stream) {
  MB_REQUIRE(a && a->d_x && a->d_y && a->d_wpacked && a->d_bias, "resblock_stage_f32: null pointer");
  MB_REQUIRE(a->d_x != a->d_y, "resblock_stage_f32: in-place is not supported (tiles read their neighbours' halo)");
  MB_REQUIRE(a->num_kernels >= 1 && a->num_kernels <= S32_MAX_CHAINS && a->num_dilations >= 1 && a->num_dilations <= S32_MAX_UNITS,
             "resblock_stage_f32: kernels=%d dilations=%d unsupported", a->num_kernels, a->num_dilations);
  int dil[S32_MAX_CHAINS * S32_MAX_UNITS];
  for (int c = 0; c < a->num_kernels; ++c)
    for (int u = 0; u < a->num_dilations; ++u) dil[c * a->num_dilations + u] = a->dilation[c][u];
  MB_REQUIRE(stage32_shape_ok(a->channels, a->num_kernels, a->ksize, a->num_dilations, dil),
             "resblock_stage_f32: C=%d kernels=%d dilations=%d unsupported", a->channels, a->num_kernels, a->num_dilations);
  MB_REQUIRE(a->slope > 0.f && a->slope < 1.f, "resblock_stage_f32: leaky_relu slope must be in (0,1)");
  if (a->batch <= 0 || a->t <= 0) return MB_OK;
  MB_REQUIRE((long long)a->channels * a->t * 4 < (1ll << 31), "resblock_stage_f32: an item of %d x %d floats is beyond the 32-bit byte offsets", a->channels, a->t);
  ResStage32K k;
  memset(&k, 0, sizeof(k));
  k.x = reinterpret_cast<const float*>(a->d_x); k.y = reinterpret_cast<float*>(a->d_y);
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
  Stage32Geom g;
  hipStream_t s = (hipStream_t)stream;
  MB_REQUIRE(stage32_geom_c(a->channels, a->num_kernels, a->ksize, a->num_dilations, dil, &g), "resblock_stage_f32: the tile does not fit LDS");
  if (a->channels == 32) return launch_stage32<32, 1, 4, 2, 2>(k, g, a->batch, s);
  return launch_stage32<64, 1, 2, 2, 2>(k, g, a->batch, s);
}
