// Round 6: a Conv1d of the GAN vocoders at the REFERENCE's precision on time-major fp32 tensors -- everything of the fp32 generators
// that is not a ResBlock unit (resblock_pair_split.hip): conv_pre, the ConvTranspose1d upsamplers, Fre-GAN's cond_up / res_output,
// conv_post.
//
//   y[b][t][m] = act( out_scale * ( sum_{ci, j} W[m][ci][j] * lrelu(x[b][t - pad + j * dil][ci]) + bias[m] + res[b][t][m] ) ) (+ y)
//
// with x fp32 [B][T][c_in], y fp32 [B][T][M].  A ConvTranspose1d(C -> C', kernel 2 u, stride u) IS such a conv: output sample u q + r
// depends on the input rows q - 1, q, q + 1 only, so with M = u C' (m = r C' + co), three taps and the polyphase weights
// W[r C' + co][ci][j] = w[ci][co][u (1 - j) + r + pad] (zero outside [0, 2 u)), the [T][u C'] result is, byte for byte, the time-major
// [u T][C'] tensor of the upsampled signal (gan.hip builds these images: models/vocoder/hifigan/models.py:120-123,
// models/vocoder/fregan/generator.py:95-118).  Nearest-repeat x u followed by a 1x1 conv (Fre-GAN res_output, generator.py:103-110)
// is the same with one tap and the weights repeated u times.
//
// Structure = resblock_pair_split.hip's phase 1: 4 MMA waves (error-compensated fp16 MFMA products from split_tm.h on LDS-resident
// hi / scaled-lo planes and a register ring of {hi, lo} weight fragments) and 4 support waves that own all HBM traffic (window loads
// one barrier interval ahead, lrelu + split, the fp32 result tile's bias / residual / activation and its coalesced write-out).  A
// workgroup owns one group of MG output channels (its weight stream is circular over the chunk x tap sequence) and walks over
// position tiles; no halo is recomputed (every output row of a tile is useful).
#include <atomic>
#include <cmath>
#include "common.h"
#include "split_tm.h"

namespace mb {

struct ConvTmK {
  const float* x; float* y; const h16* w; const float* bias; const float* res;
  const float* gate;               // highway epilogue (out_act 4): y = g relu(v) + (1 - g) res
  const float* post_scale; const float* post_shift;  // per-channel affine behind the activation (BatchNorm after ReLU), or null
  long long x_bstride, y_bstride;  // floats per batch item
  int x_row_stride;                // floats between consecutive rows of x (c_in when dense)
  int x_split;                     // x is a SPLIT tensor: fp16 [B][T][hi | lo][c_in] (what a producer's y_split / mb_maxpool2_tm wrote): staged by copy
  h16* ysplit;                     // also write the result as such a split tensor [B][T][hi | lo][c_out] (or null)
  int T;                           // rows per item (input rows = output rows)
  int c_in, c_out;                 // row strides of x and y in floats (c_out = M)
  int ntaps, dil, pad;
  int NCH;                         // chunks of CK input channels (even; channels beyond c_in read as zero and have zero weights)
  int n_mg;                        // groups of MG output channels
  int tiles_per_item, n_ntiles, x_rows, nbuf;
  float in_slope, us, out_scale;
  int out_act, accumulate;
  const int* valid; int valid_mul;
  unsigned* range_events;
  unsigned long long* trace;       // diagnostics builds only (-DCTM_TRACE_BUILD, MBHIP_DIAG=ctm_trace=<file>): shader-clock marks of workgroup 0
};

constexpr int CTM_NL = 4;  // support waves
// diagnostics builds only (tools/build_variant.sh ... -DCTM_DBG=<bits>; results are wrong, timings isolate one cost each):
// 1 = no global stores in the write-out, 2 = no LDS read of the result tile, 4 = window pieces not laid down, 8 = no window loads
#ifndef CTM_DUAL
#define CTM_DUAL 1
#endif
#ifndef CTM_DBG
#define CTM_DBG 0
#endif
#ifdef CTM_TRACE_BUILD
#define CT_MARK(role, it, k)                                                                    \
  do {                                                                                          \
    if (a.trace && blockIdx.x == 0 && (it) < 4 && (tid & 63) == 0 && wave == ((role) ? 4 : 0))  \
      a.trace[((role) * 4 + (it)) * 64 + (k)] = (unsigned long long)clock64();                  \
  } while (0)
#else
#define CT_MARK(role, it, k) do { } while (0)
#endif

__device__ __forceinline__ int ctm_valid_len(const ConvTmK& a, int b) {
  if (!a.valid) return a.T;
  int v;
  const int* p = a.valid + b;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return min(a.T, v * a.valid_mul);
}

template <int CK_, int MT_, int WN_, int NTW_> struct CtmGeom {
  static constexpr int CK = CK_, KB = CK_ / 16, MT = MT_, WN = WN_, NTW = NTW_, WM = 4 / WN_;
  static constexpr int MG = WM * MT * 32;   // output channels per workgroup
  static constexpr int N1 = WN * NTW * 32;  // positions per tile
  static constexpr int CKP = CK + 8;        // LDS row stride of the x planes in halves
  static constexpr int MGF = MG + 4;        // fp32 result tile row stride in floats
};
constexpr int CTM_MAX_HALO = 80;  // (ntaps - 1) * dil the support waves' window registers are sized for (ResBlock2 units: k = 7, d = 12 -> 72)

// JC = chunks per JOB (1 or 2): a job is what one chunk barrier publishes.  Two chunks per barrier double the MFMAs between barriers for
// the convs whose chunk is short against the round trip of the window loads (pointwise convs, 3-tap convs on split inputs).
template <int CK_, int MT_, int WN_, int NTW_, int JC_>
__global__ __launch_bounds__(64 * (4 + CTM_NL)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_split_tm_kernel(ConvTmK a) {
  using G = CtmGeom<CK_, MT_, WN_, NTW_>;
  constexpr int JC = JC_;
  constexpr int CK = G::CK, KB = G::KB, MT = G::MT, WN = G::WN, NTW = G::NTW, MG = G::MG, N1 = G::N1, CKP = G::CKP, MGF = G::MGF;
  constexpr int TD = CK == 64 ? 1 : 2;        // taps of weight fragments in flight (64-channel chunks: one tap = 4 k-steps = the same cover)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int XPL = a.x_rows * CKP;             // halves per plane of an x chunk buffer
  h16* xs = reinterpret_cast<h16*>(lds_raw);  // [nbuf][JC chunks][hi | lo][x_rows][CKP]
  float* ys = reinterpret_cast<float*>(xs + a.nbuf * JC * 2 * XPL);  // [N1][MGF]
  float* bs = ys + N1 * MGF;                  // [MG]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = a.ntaps, NCH = a.NCH;
  // workgroup -> (channel group, its share of the position tiles)
  const int mg = (int)blockIdx.x % a.n_mg, wslot = (int)blockIdx.x / a.n_mg, nslots = (int)gridDim.x / a.n_mg;
  const int my_tiles = (a.n_ntiles - wslot + nslots - 1) / nslots;
  const int NJT = NCH / JC;          // jobs per tile (even)
  const int njobs = my_tiles * NJT;
  const int m0 = mg * MG;

  for (int i = tid; i < MG; i += 64 * (4 + CTM_NL)) bs[i] = (a.bias && m0 + i < a.c_out) ? a.bias[m0 + i] : 0.f;
  __syncthreads();  // Z

  if (wave >= 4) {
    // ------------------------------ support waves (resblock_pair_split.hip's ring schedule) ------------------------------
    constexpr int PPR = CK / 4;        // pieces per row of a chunk
    constexpr int PPRJ = JC * PPR;     // ... of a job
    constexpr int NSL = 64 * CTM_NL;
    constexpr int LBX = ((N1 + (CK == 64 ? 0 : (JC == 2 ? 8 : CTM_MAX_HALO))) * PPRJ + NSL - 1) / NSL;  // (64-channel chunks: pointwise convs only; two-chunk jobs: k <= 9)
    constexpr int YPR = MG / 4;                            // 16-byte pieces per result row of this channel group
    constexpr int WB = 4;                                  // result pieces per lane per batch
    const float slope = a.in_slope;
    const int total = a.x_rows * PPRJ;
    const int ltid = tid - 256;
    const bool w_loads = a.res != nullptr || a.accumulate || a.gate != nullptr;
    // per-lane constants of the window pieces.  fp32 x: piece = 4 floats of a row; split x: piece = 8 halves of a row's hi plane
    // (pieces 0 .. PPR/2-1) or lo plane -- the same PPR pieces per row either way
    int xrow[LBX];
    unsigned xcol[LBX], xlds[LBX];
#pragma unroll
    for (int i = 0; i < LBX; ++i) {
      const int idx = min(i * NSL + ltid, total - 1);
      xrow[i] = idx / PPRJ;
      const unsigned pcj = (unsigned)(idx - xrow[i] * PPRJ), cj = pcj / PPR, pc = pcj - cj * PPR;  // chunk of the job, piece of that chunk's row
      if (a.x_split) {
        const unsigned plane = pc / (PPR / 2), c8 = (pc % (PPR / 2)) * 8u;
        xcol[i] = plane * 0x10000u + cj * CK + c8;               // (plane, first channel of the piece inside the job)
        xlds[i] = (cj * 2u + plane) * (unsigned)XPL + (unsigned)xrow[i] * CKP + c8;
      } else {
        xcol[i] = cj * CK + pc * 4u;
        xlds[i] = cj * 2u * (unsigned)XPL + (unsigned)xrow[i] * CKP + pc * 4u;
      }
    }
    auto tile_of = [&](int it, int& b, int& t0) __attribute__((always_inline)) {
      const int nt = wslot + it * nslots;
      b = nt / a.tiles_per_item;
      t0 = (nt - b * a.tiles_per_item) * N1;
    };
    auto issue_x = [&](int q, f32x4 (&vx)[LBX]) __attribute__((always_inline)) {
      if (CTM_DBG & 8) return;
      int b, t0;
      tile_of(q / NJT, b, t0);
      const int c = (q % NJT) * JC;  // first chunk of the job
      const int Tb = ctm_valid_len(a, b);
      const int tx0 = t0 - a.pad;
      // rows outside [0, Tb) come back as zeros from the descriptor's range check; channels beyond c_in (a padded last chunk) read
      // finite neighbours (or zeros past the end) against zero weights
      if (a.x_split) {  // (uniform) 16-byte pieces of the hi / lo planes, copied as they are
        const h16* xh = reinterpret_cast<const h16*>(a.x) + (long long)b * a.T * 2 * a.c_in;
        const __amdgpu_buffer_rsrc_t rs = tm_rsrc(xh, (long long)Tb * 2 * a.c_in * 2);
#pragma unroll
        for (int i = 0; i < LBX; ++i) {
          const int plane = (int)(xcol[i] >> 16), ch = c * CK + (int)(xcol[i] & 0xffffu);
          vx[i] = tm_load16(rs, (unsigned)(((tx0 + xrow[i]) * 2 + plane) * a.c_in + ch) * 2u);
        }
        return;
      }
      const __amdgpu_buffer_rsrc_t rs = tm_rsrc(a.x + (long long)b * a.x_bstride, Tb > 0 ? ((long long)(Tb - 1) * a.x_row_stride + a.c_in) * 4 : 0);
#pragma unroll
      for (int i = 0; i < LBX; ++i) vx[i] = tm_load16(rs, (unsigned)((tx0 + xrow[i]) * a.x_row_stride + c * CK + (int)xcol[i]) * 4u);
    };
    auto commit_x = [&](int q, const f32x4 (&vx)[LBX]) __attribute__((always_inline)) {
      if (CTM_DBG & 4) return;
      h16* buf = xs + (q % a.nbuf) * JC * 2 * XPL;
      if (a.x_split) {
#pragma unroll
        for (int i = 0; i < LBX; ++i) *reinterpret_cast<f32x4*>(buf + xlds[i]) = vx[i];
        return;
      }
#pragma unroll
      for (int i = 0; i < LBX; ++i) {
        float l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = slope == 1.f ? vx[i][e] : fmaxf(vx[i][e], vx[i][e] * slope);  // leaky_relu for slopes in (0, 1); 1 = no activation
        mb_h2 h0, l0, h1, l1;
        split_pair(l[0], l[1], h0, l0);
        split_pair(l[2], l[3], h1, l1);
        const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        *reinterpret_cast<h16x4*>(buf + xlds[i]) = hi;  // (pieces past the window repeat its last piece)
        *reinterpret_cast<h16x4*>(buf + XPL + xlds[i]) = lo;
      }
      if (a.range_events) {  // diagnostics only (MBHIP_CONV_RANGE_CHECK=1)
        int n_out = 0;
#pragma unroll
        for (int i = 0; i < LBX; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) n_out += (i * NSL + ltid < total && !(fabsf(vx[i][e]) <= 65504.f)) ? 1 : 0;
        if (n_out) atomicAdd(a.range_events, (unsigned)n_out);
      }
    };
    // result rows of a finished tile: ys (conv 2^-s + bias, fp32) [+ res] -> scale -> [tanh] [+ y] -> HBM, in parts over the next tile's
    // chunk intervals.  Only residual / accumulating launches load anything here (requested one part ahead).
    struct WTile { const float* rb; const float* gb; float* yb; h16* sb; int ytotal, mcols; };
    auto wtile = [&](int it) __attribute__((always_inline)) {
      int b, t0;
      tile_of(it, b, t0);
      const int Tb = ctm_valid_len(a, b);
      const int rows = max(0, min(N1, Tb - t0));
      const long long o = (long long)b * a.y_bstride + (long long)t0 * a.c_out + m0;
      return WTile{a.res ? a.res + o : nullptr, a.gate ? a.gate + o : nullptr, a.y + o,
                   a.ysplit ? a.ysplit + ((long long)b * a.T + t0) * 2 * a.c_out + m0 : nullptr, rows * YPR, min(MG, a.c_out - m0)};
    };
    auto write_part = [&](int it, int lo, int hi, int den) __attribute__((always_inline)) {  // batches [nbt lo / den, nbt hi / den)
      const WTile w = wtile(it);
      if (w.ytotal <= 0) return;
      constexpr int BSZ = NSL * WB;
      const int nbt = (w.ytotal + BSZ - 1) / BSZ;
      for (int base = (nbt * lo / den) * BSZ; base < (nbt * hi / den) * BSZ && base < w.ytotal; base += BSZ) {
        f32x4 rx[WB], ry[WB], rg[WB];
        unsigned goff[WB];
        bool ok[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int idx = base + i * NSL + ltid;
          const unsigned idc = (unsigned)min(idx, w.ytotal - 1);
          const unsigned row = idc / YPR, pc = idc - row * YPR;
          ok[i] = idx < w.ytotal && (int)(pc * 4) < w.mcols;  // (a last channel group may be narrower than MG)
          goff[i] = row * (unsigned)a.c_out + min(pc * 4u, (unsigned)max(w.mcols - 4, 0));
          if (w_loads) {
            rx[i] = w.rb ? *reinterpret_cast<const f32x4*>(w.rb + goff[i]) : (f32x4)0.f;
            ry[i] = a.accumulate ? *reinterpret_cast<const f32x4*>(w.yb + goff[i]) : (f32x4)0.f;
            rg[i] = w.gb ? *reinterpret_cast<const f32x4*>(w.gb + goff[i]) : (f32x4)0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int idx = base + i * NSL + ltid;
          const unsigned idc = (unsigned)min(idx, w.ytotal - 1);
          const unsigned row = idc / YPR, pc = idc - row * YPR;
          const f32x4 hv = (CTM_DBG & 2) ? (f32x4)1.f : *reinterpret_cast<const f32x4*>(ys + row * MGF + pc * 4);
          const unsigned mc = (unsigned)m0 + min(pc * 4u, (unsigned)max(w.mcols - 4, 0));
          f32x4 psc = (f32x4)1.f, psh = (f32x4)0.f;
          if (a.post_scale) { psc = *reinterpret_cast<const f32x4*>(a.post_scale + mc); psh = *reinterpret_cast<const f32x4*>(a.post_shift + mc); }
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float f = hv[e];
            if (a.out_act == 4) {  // highway (common/highway_network.py:12-17): g relu(W1 x) + (1 - g) x, g = sigmoid(W2 x) from the gate launch
              const float g = rg[i][e];
              f = g * fmaxf(f, 0.f) + (1.f - g) * rx[i][e];
            } else {
              if (w_loads) f += rx[i][e];
              f *= a.out_scale;
              if (a.out_act == 1) f = fmaxf(f, 0.f);
              else if (a.out_act == 2) f = tanhf(f);
              else if (a.out_act == 3) f = sigmoidf_(f);
              if (a.post_scale) f = fmaf(f, psc[e], psh[e]);  // BatchNorm behind the ReLU (common/batch_norm_conv.py:11-14)
              if (w_loads) f += ry[i][e];
            }
            o[e] = f;
          }
          if (ok[i] && a.ysplit) {  // the same values as fp16 hi / scaled-lo rows for a consumer that stages by copy (mcols >= 4)
            mb_h2 h0, l0, h1, l1;
            split_pair(o[0], o[1], h0, l0);
            split_pair(o[2], o[3], h1, l1);
            const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
            h16* sp = w.sb + (size_t)row * 2 * a.c_out + min(pc * 4u, (unsigned)max(w.mcols - 4, 0));
            *reinterpret_cast<h16x4*>(sp) = hi;
            *reinterpret_cast<h16x4*>(sp + a.c_out) = lo;
          }
          if (ok[i] && !((CTM_DBG & 1) && o[0] != 12345.f)) {
            if (w.mcols >= 4) *reinterpret_cast<f32x4*>(w.yb + goff[i]) = o;
            else  // fewer than 4 output channels (conv_post: one): element stores
              for (int e = 0; e < w.mcols; ++e) w.yb[row * (unsigned)a.c_out + e] = o[e];
          }
        }
      }
    };
    f32x4 vxA[LBX], vxB[LBX];
    // (Measured and dropped: with three or more buffers, laying a register set down and requesting it again behind the same barrier lets
    // the loads fly for two intervals on the same registers -- no faster here, 5-10 % slower in resblock_pair_split.hip.)
    for (int q = 0; q < a.nbuf - 1 && q < njobs; ++q) { issue_x(q, vxA); commit_x(q, vxA); }
    if (a.nbuf - 1 < njobs) issue_x(a.nbuf - 1, vxB);
    for (int it = 0; it < my_tiles; ++it) {
      for (int c = 0; c < NJT; c += 2) {  // two jobs per turn: the register sets alternate
        const int q = it * NJT + c;
        if (c < 8) CT_MARK(1, it, 4 * c);
        __syncthreads();  // B_q: job q is staged, the buffer of job q-1 is free
        if (c < 8) CT_MARK(1, it, 4 * c + 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // everything outstanding is an interval old: drained here, the next request flies through the interval's processing
        if (q + a.nbuf < njobs) issue_x(q + a.nbuf, vxA);
        if (q + a.nbuf - 1 < njobs) commit_x(q + a.nbuf - 1, vxB);
        if (c < 8) CT_MARK(1, it, 4 * c + 2);
        if (it > 0) write_part(it - 1, c, c + 1, NJT);
        if (c < 8) CT_MARK(1, it, 4 * c + 3);
        __syncthreads();  // B_{q+1}
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (q + 1 + a.nbuf < njobs) issue_x(q + 1 + a.nbuf, vxB);
        if (q + a.nbuf < njobs) commit_x(q + a.nbuf, vxA);
        if (it > 0) write_part(it - 1, c + 1, c + 2, NJT);
      }
      CT_MARK(1, it, 60);
      __syncthreads();  // YF: the previous tile's result has left ys
      __syncthreads();  // Y: this tile's result is staged
    }
    if (my_tiles > 0) write_part(my_tiles - 1, 0, 1, 1);
    return;
  }

  // ------------------------------ MMA waves ------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int mt0 = wm * MT;
  const int NFT = NCH * ntaps;  // flat taps of this channel group's circular weight stream
  const h16x8* wp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
    wp[i] = reinterpret_cast<const h16x8*>(a.w) + (size_t)(mg * (MG / 32) + mt0 + i) * NFT * KB * 2 * 64 + lane;
  h16x8 ring[TD][KB][MT][2];  // [slot][k-step][tile][hi | lo]
#pragma unroll
  for (int s = 0; s < TD; ++s)
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring[s][u][i][p] = wp[i][((size_t)((s % NFT) * KB + u) * 2 + p) * 64];
  int ftn = TD % NFT;  // next flat tap to prefetch
  auto sp_wrap = []() {};
  constexpr bool DUAL = MT == 1 && CTM_DUAL != 0;  // split_tm.h: a second accumulator set instead of the third weight image
  f32x16 acc[MT][NTW], acl[MT][NTW];
  const h16 k2m11 = (h16)(1.f / 2048.f);
  const int lrow = wn * (NTW * 32) + (lane & 31);
  const int lcol = (lane >> 5) * 8;
  const int x_tapstep = a.dil * CKP;
  const float us = a.us;
  for (int it = 0; it < my_tiles; ++it) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): a known scoreboard at the head of the tile keeps the compiler's counted waits exact
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[i][n][q] = 0.f; if (DUAL) acl[i][n][q] = 0.f; }
    // NCH is even and ntaps odd: the ring slot of a chunk's first tap alternates 0, 1 and a tile always starts on slot 0
#define CT_XB(JOB, CJ) (xs + (((JOB)) % a.nbuf) * JC * 2 * XPL + (CJ) * 2 * XPL + lrow * CKP + lcol)
    for (int c = 0; c < NJT; c += 2) {
      const int q = it * NJT + c;
      if (c < 8) CT_MARK(0, it, 4 * c);
      __syncthreads();  // B_q
      if (c < 8) CT_MARK(0, it, 4 * c + 1);
      if (JC == 1) {
        SP_CHUNK(0, CT_XB(q, 0), CKP, x_tapstep, XPL);
      } else {
        SP_CHUNK(0, CT_XB(q, 0), CKP, x_tapstep, XPL);
        SP_CHUNK(1, CT_XB(q, JC - 1), CKP, x_tapstep, XPL);
      }
      if (c < 8) CT_MARK(0, it, 4 * c + 2);
      __syncthreads();  // B_{q+1}
      if (c < 8) CT_MARK(0, it, 4 * c + 3);
      if (JC == 1) {
        SP_CHUNK(1, CT_XB(q + 1, 0), CKP, x_tapstep, XPL);
      } else {
        SP_CHUNK(0, CT_XB(q + 1, 0), CKP, x_tapstep, XPL);
        SP_CHUNK(1, CT_XB(q + 1, JC - 1), CKP, x_tapstep, XPL);
      }
    }
#undef CT_XB
    CT_MARK(0, it, 60);
    __syncthreads();  // YF
    CT_MARK(0, it, 61);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        const int row = lrow + n * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co0 = (mt0 + i) * 32 + 8 * g + 4 * (lane >> 5);
          const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + co0);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(DUAL ? fmaf(acl[i][n][4 * g + e], 1.f / 2048.f, acc[i][n][4 * g + e]) : acc[i][n][4 * g + e], us, bq[e]);
          *reinterpret_cast<f32x4*>(ys + row * MGF + co0) = v;
        }
      }
    CT_MARK(0, it, 62);
    __syncthreads();  // Y
    CT_MARK(0, it, 63);
  }
}


// ---- The same conv with the WINDOW RESIDENT in LDS: one workgroup per position tile computes EVERY group of MG output channels ----
// conv_split_tm_kernel gives a workgroup one channel group, so a conv to n_mg groups stages (loads, activates, splits) every window
// n_mg times, and its MMA waves wait at a chunk barrier per 1-2 k MFMA cycles for support waves that need 2.4-3.4 k per chunk (a 256 ->
// 640 upsampler: 60 k cycles per (tile, group) for 11 k of chunk bodies).  Here every chunk of the tile's window is laid down ONCE
// ([chunk][hi | lo][rows][CKP]); the first channel group runs behind the chunk barriers as before, the others find the window in place:
// no barriers, no support-wave work beside the write-out of the previous group's result tile.  The weight ring rolls from one group's
// circular stream into the next one's (split_tm.h sp_wrap).  Per tile: NCH chunk barriers + 2 per group (YF, Y).
template <int CK_, int MT_, int WN_, int NTW_>
__global__ __launch_bounds__(64 * (4 + CTM_NL)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_split_tm_res_kernel(ConvTmK a) {
  using G = CtmGeom<CK_, MT_, WN_, NTW_>;
  constexpr int CK = G::CK, KB = G::KB, MT = G::MT, WN = G::WN, NTW = G::NTW, MG = G::MG, N1 = G::N1, CKP = G::CKP, MGF = G::MGF;
  constexpr int TD = CK == 64 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int XPL = a.x_rows * CKP;             // halves per plane of a chunk
  const int NCH = a.NCH, NG = a.n_mg, ntaps = a.ntaps;
  h16* xs = reinterpret_cast<h16*>(lds_raw);  // [NCH][hi | lo][x_rows][CKP]
  float* ys = reinterpret_cast<float*>(xs + (size_t)NCH * 2 * XPL);  // [N1][MGF]
  float* bs = ys + N1 * MGF;                  // [NG * MG]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wslot = (int)blockIdx.x, nslots = (int)gridDim.x;
  const int my_tiles = (a.n_ntiles - wslot + nslots - 1) / nslots;

  for (int i = tid; i < NG * MG; i += 64 * (4 + CTM_NL)) bs[i] = (a.bias && i < a.c_out) ? a.bias[i] : 0.f;
  __syncthreads();  // Z

  auto tile_of = [&](int it, int& b, int& t0) __attribute__((always_inline)) {
    const int nt = wslot + it * nslots;
    b = nt / a.tiles_per_item;
    t0 = (nt - b * a.tiles_per_item) * N1;
  };

  if (wave >= 4) {
    // ------------------------------ support waves ------------------------------
    constexpr int PPR = CK / 4;
    constexpr int NSL = 64 * CTM_NL;
    constexpr int LBX = ((N1 + (CK == 64 ? 0 : CTM_MAX_HALO)) * PPR + NSL - 1) / NSL;
    constexpr int YPR = MG / 4;
    constexpr int WB = 4;
    const float slope = a.in_slope;
    const int total = a.x_rows * PPR;
    const int ltid = tid - 256;
    const bool w_loads = a.res != nullptr || a.accumulate || a.gate != nullptr;
    int xrow[LBX];
    unsigned xcol[LBX], xlds[LBX];
#pragma unroll
    for (int i = 0; i < LBX; ++i) {
      const int idx = min(i * NSL + ltid, total - 1);
      xrow[i] = idx / PPR;
      const unsigned pc = (unsigned)(idx - xrow[i] * PPR);
      if (a.x_split) {
        const unsigned plane = pc / (PPR / 2), c8 = (pc % (PPR / 2)) * 8u;
        xcol[i] = plane * 0x10000u + c8;
        xlds[i] = plane * (unsigned)XPL + (unsigned)xrow[i] * CKP + c8;
      } else {
        xcol[i] = pc * 4u;
        xlds[i] = (unsigned)xrow[i] * CKP + pc * 4u;
      }
    }
    auto issue_x = [&](int it, int c, f32x4 (&vx)[LBX]) __attribute__((always_inline)) {
      int b, t0;
      tile_of(it, b, t0);
      const int Tb = ctm_valid_len(a, b);
      const int tx0 = t0 - a.pad;
      if (a.x_split) {
        const h16* xh = reinterpret_cast<const h16*>(a.x) + (long long)b * a.T * 2 * a.c_in;
        const __amdgpu_buffer_rsrc_t rs = tm_rsrc(xh, (long long)Tb * 2 * a.c_in * 2);
#pragma unroll
        for (int i = 0; i < LBX; ++i) {
          const int plane = (int)(xcol[i] >> 16), ch = c * CK + (int)(xcol[i] & 0xffffu);
          vx[i] = tm_load16(rs, (unsigned)(((tx0 + xrow[i]) * 2 + plane) * a.c_in + ch) * 2u);
        }
        return;
      }
      const __amdgpu_buffer_rsrc_t rs = tm_rsrc(a.x + (long long)b * a.x_bstride, Tb > 0 ? ((long long)(Tb - 1) * a.x_row_stride + a.c_in) * 4 : 0);
#pragma unroll
      for (int i = 0; i < LBX; ++i) vx[i] = tm_load16(rs, (unsigned)((tx0 + xrow[i]) * a.x_row_stride + c * CK + (int)xcol[i]) * 4u);
    };
    auto commit_x = [&](int c, const f32x4 (&vx)[LBX]) __attribute__((always_inline)) {
      h16* buf = xs + (size_t)c * 2 * XPL;
      if (a.x_split) {
#pragma unroll
        for (int i = 0; i < LBX; ++i) *reinterpret_cast<f32x4*>(buf + xlds[i]) = vx[i];
        return;
      }
#pragma unroll
      for (int i = 0; i < LBX; ++i) {
        float l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = slope == 1.f ? vx[i][e] : fmaxf(vx[i][e], vx[i][e] * slope);
        mb_h2 h0, l0, h1, l1;
        split_pair(l[0], l[1], h0, l0);
        split_pair(l[2], l[3], h1, l1);
        const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
        *reinterpret_cast<h16x4*>(buf + xlds[i]) = hi;
        *reinterpret_cast<h16x4*>(buf + XPL + xlds[i]) = lo;
      }
      if (a.range_events) {
        int n_out = 0;
#pragma unroll
        for (int i = 0; i < LBX; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) n_out += (i * NSL + ltid < total && !(fabsf(vx[i][e]) <= 65504.f)) ? 1 : 0;
        if (n_out) atomicAdd(a.range_events, (unsigned)n_out);
      }
    };
    // the result tile of channel group g of tile `it`: ys -> epilogue -> HBM (conv_split_tm_kernel's write_part, whole tile)
    auto write_group = [&](int it, int g) __attribute__((always_inline)) {
      int b, t0;
      tile_of(it, b, t0);
      const int Tb = ctm_valid_len(a, b);
      const int rows = max(0, min(N1, Tb - t0));
      const int m0 = g * MG;
      const long long o = (long long)b * a.y_bstride + (long long)t0 * a.c_out + m0;
      const float* rb = a.res ? a.res + o : nullptr;
      const float* gb = a.gate ? a.gate + o : nullptr;
      float* yb = a.y + o;
      h16* sb = a.ysplit ? a.ysplit + ((long long)b * a.T + t0) * 2 * a.c_out + m0 : nullptr;
      const int ytotal = rows * YPR, mcols = min(MG, a.c_out - m0);
      if (ytotal <= 0) return;
      constexpr int BSZ = NSL * WB;
      for (int base = 0; base < ytotal; base += BSZ) {
        f32x4 rx[WB], ry[WB], rg[WB];
        unsigned goff[WB];
        bool ok[WB];
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int idx = base + i * NSL + ltid;
          const unsigned idc = (unsigned)min(idx, ytotal - 1);
          const unsigned row = idc / YPR, pc = idc - row * YPR;
          ok[i] = idx < ytotal && (int)(pc * 4) < mcols;
          goff[i] = row * (unsigned)a.c_out + min(pc * 4u, (unsigned)max(mcols - 4, 0));
          if (w_loads) {
            rx[i] = rb ? *reinterpret_cast<const f32x4*>(rb + goff[i]) : (f32x4)0.f;
            ry[i] = a.accumulate ? *reinterpret_cast<const f32x4*>(yb + goff[i]) : (f32x4)0.f;
            rg[i] = gb ? *reinterpret_cast<const f32x4*>(gb + goff[i]) : (f32x4)0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int idx = base + i * NSL + ltid;
          const unsigned idc = (unsigned)min(idx, ytotal - 1);
          const unsigned row = idc / YPR, pc = idc - row * YPR;
          const f32x4 hv = *reinterpret_cast<const f32x4*>(ys + row * MGF + pc * 4);
          const unsigned mc = (unsigned)m0 + min(pc * 4u, (unsigned)max(mcols - 4, 0));
          f32x4 psc = (f32x4)1.f, psh = (f32x4)0.f;
          if (a.post_scale) { psc = *reinterpret_cast<const f32x4*>(a.post_scale + mc); psh = *reinterpret_cast<const f32x4*>(a.post_shift + mc); }
          f32x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float f = hv[e];
            if (a.out_act == 4) {
              const float gt = rg[i][e];
              f = gt * fmaxf(f, 0.f) + (1.f - gt) * rx[i][e];
            } else {
              if (w_loads) f += rx[i][e];
              f *= a.out_scale;
              if (a.out_act == 1) f = fmaxf(f, 0.f);
              else if (a.out_act == 2) f = tanhf(f);
              else if (a.out_act == 3) f = sigmoidf_(f);
              if (a.post_scale) f = fmaf(f, psc[e], psh[e]);
              if (w_loads) f += ry[i][e];
            }
            o4[e] = f;
          }
          if (ok[i] && a.ysplit) {
            mb_h2 h0, l0, h1, l1;
            split_pair(o4[0], o4[1], h0, l0);
            split_pair(o4[2], o4[3], h1, l1);
            const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
            h16* sp = sb + (size_t)row * 2 * a.c_out + min(pc * 4u, (unsigned)max(mcols - 4, 0));
            *reinterpret_cast<h16x4*>(sp) = hi;
            *reinterpret_cast<h16x4*>(sp + a.c_out) = lo;
          }
          if (ok[i]) {
            if (mcols >= 4) *reinterpret_cast<f32x4*>(yb + goff[i]) = o4;
            else
              for (int e = 0; e < mcols; ++e) yb[row * (unsigned)a.c_out + e] = o4[e];
          }
        }
      }
    };
    f32x4 vxA[LBX], vxB[LBX];
    if (my_tiles > 0) issue_x(0, 0, vxA);
    for (int it = 0; it < my_tiles; ++it) {
      // (the previous tile's last result tile has left ys and every MMA wave is behind its last chunk: the window may be overwritten)
      for (int c = 0; c < NCH; c += 2) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        issue_x(it, c + 1, vxB);
        commit_x(c, vxA);
        __syncthreads();  // B_c: chunk c is in place
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (c + 2 < NCH) issue_x(it, c + 2, vxA);
        else if (it + 1 < my_tiles) issue_x(it + 1, 0, vxA);  // (flies through the other groups; laid down behind the tile's last barrier)
        commit_x(c + 1, vxB);
        __syncthreads();  // B_{c+1}
      }
      for (int g = 0; g < NG; ++g) {
        __syncthreads();  // YF_g: ys is free (the write-out of the group before is done: this wave arrives here behind it)
        __syncthreads();  // Y_g: group g's result tile is staged
        write_group(it, g);
      }
    }
    return;
  }

  // ------------------------------ MMA waves ------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int mt0 = wm * MT;
  const int NFT = NCH * ntaps;  // flat taps of ONE channel group's stream
  const size_t tile_stride = (size_t)NFT * KB * 2 * 64;  // h16x8 units per 32-row tile
  const h16x8* wbase = reinterpret_cast<const h16x8*>(a.w) + lane;
  const h16x8* wp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) wp[i] = wbase + (size_t)(mt0 + i) * tile_stride;
  int grp_ring = 0;  // the channel group whose stream the refills read
  auto sp_wrap = [&]() __attribute__((always_inline)) {
    grp_ring = grp_ring + 1 == NG ? 0 : grp_ring + 1;
#pragma unroll
    for (int i = 0; i < MT; ++i) wp[i] = wbase + (size_t)(grp_ring * (MG / 32) + mt0 + i) * tile_stride;
  };
  h16x8 ring[TD][KB][MT][2];
#pragma unroll
  for (int s = 0; s < TD; ++s)
#pragma unroll
    for (int u = 0; u < KB; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) ring[s][u][i][p] = wp[i][((size_t)(s * KB + u) * 2 + p) * 64];  // (NFT >= 2: launch_ctm_res)
  int ftn = TD;
  if (ftn == NFT) { ftn = 0; sp_wrap(); }
  constexpr bool DUAL = MT == 1 && CTM_DUAL != 0;
  f32x16 acc[MT][NTW], acl[MT][NTW];
  const h16 k2m11 = (h16)(1.f / 2048.f);
  const int lrow = wn * (NTW * 32) + (lane & 31);
  const int lcol = (lane >> 5) * 8;
  const int x_tapstep = a.dil * CKP;
  const float us = a.us;
  for (int it = 0; it < my_tiles; ++it) {
    for (int g = 0; g < NG; ++g) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // a known scoreboard at the head of a group keeps the compiler's counted waits exact
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
          for (int q = 0; q < 16; ++q) { acc[i][n][q] = 0.f; if (DUAL) acl[i][n][q] = 0.f; }
#define CR_XB(C_) (xs + (size_t)(C_) * 2 * XPL + lrow * CKP + lcol)
      for (int c = 0; c < NCH; c += 2) {
        if (g == 0) __syncthreads();  // B_c
        SP_CHUNK(0, CR_XB(c), CKP, x_tapstep, XPL);
        if (g == 0) __syncthreads();  // B_{c+1}
        SP_CHUNK(1, CR_XB(c + 1), CKP, x_tapstep, XPL);
      }
#undef CR_XB
      __syncthreads();  // YF_g
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
          const int row = lrow + n * 32;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int co0 = (mt0 + i) * 32 + 8 * q4 + 4 * (lane >> 5);
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + g * MG + co0);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(DUAL ? fmaf(acl[i][n][4 * q4 + e], 1.f / 2048.f, acc[i][n][4 * q4 + e]) : acc[i][n][4 * q4 + e], us, bq[e]);
            *reinterpret_cast<f32x4*>(ys + row * MGF + co0) = v;
          }
        }
      __syncthreads();  // Y_g
    }
  }
}

template <class G, int JC>
static size_t ctm_lds_bytes(int ntaps, int dil, int nbuf) {
  const int x_rows = G::N1 + (ntaps - 1) * dil;
  return (size_t)nbuf * JC * 2 * x_rows * G::CKP * sizeof(h16) + (size_t)G::N1 * G::MGF * sizeof(float) + G::MG * sizeof(float);
}

static int ctm_cus() {
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

template <int CK, int MT, int WN, int NTW, int JC = 1>
static int launch_ctm(ConvTmK k, int batch, hipStream_t s) {
  using G = CtmGeom<CK, MT, WN, NTW>;
  k.x_rows = G::N1 + (k.ntaps - 1) * k.dil;
  k.tiles_per_item = cdiv(k.T, G::N1);
  k.n_ntiles = k.tiles_per_item * batch;
  k.n_mg = cdiv(k.c_out, G::MG);
  int nbuf = 2;
  while (nbuf < 4 && ctm_lds_bytes<G, JC>(k.ntaps, k.dil, nbuf + 1) <= (size_t)160 * 1024) ++nbuf;
  const size_t lds_bytes = ctm_lds_bytes<G, JC>(k.ntaps, k.dil, nbuf);
  MB_REQUIRE(lds_bytes <= (size_t)160 * 1024, "conv_split_tm: the window does not fit LDS");
  k.nbuf = nbuf;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if (!bit || !(attr_done.load(std::memory_order_acquire) & bit)) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_tm_kernel<CK, MT, WN, NTW, JC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int slots = std::max(1, std::min(k.n_ntiles, ctm_cus() / k.n_mg));
#ifdef CTM_TRACE_BUILD
  static unsigned long long* d_trace = nullptr;
  std::string trace_file;
  const char* trace_path = diag_str("ctm_trace", &trace_file) ? trace_file.c_str() : nullptr;
  if (trace_path) {
    if (!d_trace) MB_HIP(hipMalloc((void**)&d_trace, 512 * sizeof(unsigned long long)));
    MB_HIP(hipMemsetAsync(d_trace, 0, 512 * sizeof(unsigned long long), s));
    k.trace = d_trace;
  }
#endif
  hipLaunchKernelGGL((conv_split_tm_kernel<CK, MT, WN, NTW, JC>), dim3(slots * k.n_mg), dim3(64 * (4 + CTM_NL)), lds_bytes, s, k);
  MB_HIP(hipGetLastError());
#ifdef CTM_TRACE_BUILD
  if (trace_path) {
    unsigned long long h[512];
    MB_HIP(hipStreamSynchronize(s));
    MB_HIP(hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(trace_path, "a")) {
      fprintf(f, "%d %d %d %d %d %d %d %d %d %d :", CK, MT, WN, NTW, k.c_in, k.c_out, k.ntaps, k.NCH, k.nbuf, k.n_ntiles);
      for (int i = 0; i < 512; ++i) fprintf(f, " %llu", h[i]);
      fprintf(f, "\n");
      fclose(f);
    }
  }
#endif
  return MB_OK;
}


template <class G>
static size_t ctm_res_lds_bytes(int ntaps, int dil, int nch, int n_mg) {
  const int x_rows = G::N1 + (ntaps - 1) * dil;
  return (size_t)nch * 2 * x_rows * G::CKP * sizeof(h16) + (size_t)G::N1 * G::MGF * sizeof(float) + (size_t)n_mg * G::MG * sizeof(float);
}

// -> MB_OK after the launch, or +1 when this instance does not fit (the caller falls back to the ring kernel)
template <int CK, int MT, int WN, int NTW>
static int launch_ctm_res(ConvTmK k, int batch, hipStream_t s) {
  using G = CtmGeom<CK, MT, WN, NTW>;
  k.x_rows = G::N1 + (k.ntaps - 1) * k.dil;
  k.tiles_per_item = cdiv(k.T, G::N1);
  k.n_ntiles = k.tiles_per_item * batch;
  k.n_mg = cdiv(k.c_out, G::MG);
  k.nbuf = k.NCH;
  const size_t lds_bytes = ctm_res_lds_bytes<G>(k.ntaps, k.dil, k.NCH, k.n_mg);
  if (lds_bytes > (size_t)160 * 1024 || k.NCH * k.ntaps < 2 || (CK != 64 && (k.ntaps - 1) * k.dil > CTM_MAX_HALO) || (CK == 64 && k.ntaps != 1)) return 1;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if (!bit || !(attr_done.load(std::memory_order_acquire) & bit)) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_tm_res_kernel<CK, MT, WN, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int grid = std::max(1, std::min(k.n_ntiles, ctm_cus()));
  hipLaunchKernelGGL((conv_split_tm_res_kernel<CK, MT, WN, NTW>), dim3(grid), dim3(64 * (4 + CTM_NL)), lds_bytes, s, k);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

static int ctm_ck(int c_in) { return c_in <= 32 ? 16 : 32; }
static int ctm_nch(int c_in) { const int ck = ctm_ck(c_in); return cdiv(c_in, 2 * ck) * 2; }  // chunks (even)
static int ctm_mg(int m) { return (m % 256 == 0) ? 256 : (m % 128 == 0 ? 128 : (m % 64 == 0 ? 64 : 32)); }  // channels per workgroup

}  // namespace mb

using namespace mb;

extern "C" int mb_conv_split_tm_supported(int c_out, int c_in, int ksize, int dilation) {
  if (c_out < 1 || c_in < 4 || (c_in & 3) || ksize < 1 || (ksize & 1) == 0 || dilation < 1) return 0;
  return (ksize - 1) * dilation <= CTM_MAX_HALO;
}

extern "C" size_t mb_conv_split_tm_packed_halves(int c_out, int c_in, int ksize) {
  const int ck = ctm_ck(c_in), nch = ctm_nch(c_in), mg = ctm_mg(c_out);
  const int mtt = cdiv(c_out, mg) * (mg / 32);
  return (size_t)mtt * nch * ksize * (ck / 16) * 2 * 512;
}

// h_w: the conv's fp32 weights [c_out][c_in][ksize] (torch Conv1d layout; for a ConvTranspose1d the polyphase image described at the
// top of this file).  h_unscale receives 2^-s.
extern "C" int mb_conv_split_tm_pack(const float* h_w, int c_out, int c_in, int ksize, uint16_t* h_packed, float* h_unscale) {
  MB_REQUIRE(h_w && h_packed && h_unscale, "conv_split_tm_pack: null pointer");
  MB_REQUIRE(mb_conv_split_tm_supported(c_out, c_in, ksize, 1), "conv_split_tm_pack: %d -> %d channels, k = %d unsupported", c_in, c_out, ksize);
  const int CK = ctm_ck(c_in), KB = CK / 16, NCH = ctm_nch(c_in), MG = ctm_mg(c_out), MTT = cdiv(c_out, MG) * (MG / 32);
  float wmax = 0.f;
  for (size_t q = 0; q < (size_t)c_out * c_in * ksize; ++q) wmax = std::max(wmax, std::fabs(h_w[q]));
  int sexp = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    int e2;
    std::frexp(wmax, &e2);
    sexp = std::max(-24, std::min(40, 14 - e2));
  }
  const float scale = std::ldexp(1.f, sexp);
  *h_unscale = std::ldexp(1.f, -sexp);
  h16* out = reinterpret_cast<h16*>(h_packed);
  size_t o = 0;
  for (int mt = 0; mt < MTT; ++mt)
    for (int c = 0; c < NCH; ++c)
      for (int j = 0; j < ksize; ++j)
        for (int u = 0; u < KB; ++u)
          for (int part = 0; part < 2; ++part)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                const int co = mt * 32 + (lane & 31);
                const int ci = c * CK + u * 16 + (lane >> 5) * 8 + e;
                const float v = (co < c_out && ci < c_in) ? h_w[((size_t)co * c_in + ci) * ksize + j] * scale : 0.f;
                const h16 hi = (h16)v;
                out[o++] = part == 0 ? hi : (h16)(v - (float)hi);
              }
  return MB_OK;
}

extern "C" int mb_conv_split_tm(const mb_conv_split_tm_args* a, mb_stream_t stream) {
  MB_REQUIRE(a && a->d_x && a->d_y && a->d_wpacked, "conv_split_tm: null pointer");
  MB_REQUIRE((const void*)a->d_x != (const void*)a->d_y, "conv_split_tm: in-place is not supported");
  MB_REQUIRE(mb_conv_split_tm_supported(a->c_out, a->c_in, a->ksize, a->dilation), "conv_split_tm: %d -> %d channels, k = %d, d = %d unsupported",
             a->c_in, a->c_out, a->ksize, a->dilation);
  MB_REQUIRE(a->in_slope > 0.f && a->in_slope <= 1.f, "conv_split_tm: in_slope must be in (0, 1] (1 = no activation)");
  MB_REQUIRE(a->unscale > 0.f, "conv_split_tm: the unscale factor of mb_conv_split_tm_pack is missing");
  MB_REQUIRE(a->out_act >= 0 && a->out_act <= 4, "conv_split_tm: out_act %d (0 none, 1 relu, 2 tanh, 3 sigmoid, 4 highway)", a->out_act);
  MB_REQUIRE(a->out_act != 4 || (a->d_gate && a->d_res && !a->accumulate), "conv_split_tm: the highway epilogue needs d_gate and d_res");
  MB_REQUIRE((!a->d_post_scale) == (!a->d_post_shift), "conv_split_tm: d_post_scale and d_post_shift come together");
  MB_REQUIRE((!a->d_gate && !a->d_post_scale) || a->c_out % 4 == 0, "conv_split_tm: gate / post affine need c_out %% 4 == 0");
  MB_REQUIRE((!a->d_res && !a->accumulate) || a->c_out % 4 == 0, "conv_split_tm: residual / accumulating launches need c_out %% 4 == 0");
  if (a->batch <= 0 || a->t <= 0) return MB_OK;
  MB_REQUIRE((long long)a->t * std::max(a->x_row_stride > 0 ? a->x_row_stride : a->c_in, a->c_out) < (1ll << 31), "conv_split_tm: an item of %d rows is beyond the 32-bit offsets", a->t);
  ConvTmK k;
  memset(&k, 0, sizeof(k));
  k.x = a->d_x; k.y = a->d_y; k.w = reinterpret_cast<const h16*>(a->d_wpacked); k.bias = a->d_bias; k.res = a->d_res;
  k.gate = a->d_gate; k.post_scale = a->d_post_scale; k.post_shift = a->d_post_shift;
  k.x_split = a->x_split ? 1 : 0; k.ysplit = reinterpret_cast<h16*>(a->d_ysplit);
  MB_REQUIRE(!a->x_split || (a->c_in % 8 == 0 && a->in_slope == 1.f && a->x_row_stride <= 0 && a->x_bstride <= 0),
             "conv_split_tm: a split x needs c_in %% 8 == 0, no input activation and the dense layout");
  MB_REQUIRE(!a->d_ysplit || a->c_out % 4 == 0, "conv_split_tm: d_ysplit needs c_out %% 4 == 0");
  k.x_row_stride = a->x_row_stride > 0 ? a->x_row_stride : a->c_in;
  k.x_bstride = a->x_bstride > 0 ? a->x_bstride : (long long)a->t * a->c_in; k.y_bstride = (long long)a->t * a->c_out;
  k.T = a->t; k.c_in = a->c_in; k.c_out = a->c_out; k.ntaps = a->ksize; k.dil = a->dilation; k.pad = a->pad;
  k.NCH = ctm_nch(a->c_in);
  k.in_slope = a->in_slope; k.us = a->unscale; k.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
  k.out_act = a->out_act; k.accumulate = a->accumulate;
  k.valid = a->d_valid; k.valid_mul = a->valid_mul > 0 ? a->valid_mul : 1;
  k.range_events = conv_range_word();
  hipStream_t s = (hipStream_t)stream;
  const int mg = ctm_mg(a->c_out);
  MB_REQUIRE(!a->x_split || ctm_ck(a->c_in) == 32, "conv_split_tm: a split x needs more than 32 input channels");
  if (ctm_ck(a->c_in) == 16) {
    switch (mg) {
      case 256: return launch_ctm<16, 2, 1, 3>(k, a->batch, s);
      case 128: return launch_ctm<16, 1, 1, 4>(k, a->batch, s);
      case 64: return launch_ctm<16, 1, 2, 2>(k, a->batch, s);
      default: return launch_ctm<16, 1, 4, 2>(k, a->batch, s);
    }
  }
  // tile shape for wide outputs: the candidate with the smallest makespan = rounds of workgroups x (channels x rows) per tile.  (A
  // 2560 -> 512 projection over 32 x 400 rows: 256 x 96 tiles are 320 tiles = two rounds on 256 compute units, a fifth of the
  // second-round tiles 16 rows tall; 256 x 64 tiles are 448 = two rounds of two thirds the work each.)
  const int cus = ctm_cus();
  auto cost = [&](int mgc, int n1) {
    const long long tiles = (long long)cdiv(a->c_out, mgc) * cdiv(a->t, n1) * a->batch;
    return ((tiles + cus - 1) / cus) * (long long)mgc * n1;
  };
  // pointwise convs over >= 128 input channels take 64-channel chunks: 36 MFMAs per wave between two chunk barriers (1.1 k cycles) are
  // less than the round trip of the window loads the support waves wait for (the image is the same: one tap = consecutive k-steps)
  const bool wide_chunk = a->ksize == 1 && a->c_in % 128 == 0;
  if (wide_chunk && mg >= 128) k.NCH = a->c_in / 64;
  const int tile = diag_int("ctm_tile", 0);  // A/B: 1 = 256 x 96, 2 = 256 x 64, 3 = 128 x 128
  // Several channel groups over enough position tiles to fill the chip: the resident-window kernel (the window is staged once per tile
  // instead of once per (tile, group)); its instances trade rows per tile for the LDS the whole window needs.  (A/B: MBHIP_DIAG=ctm_nores)
  if (tile == 0 && !diag_int("ctm_nores") && mg >= 128 && cdiv(a->c_out, mg) >= 2) {
    const long long rows = (long long)a->t * a->batch;
    int r = 1;
    if (wide_chunk) {
      if (mg == 256 && rows >= 32ll * cus / 2) r = launch_ctm_res<64, 2, 1, 1>(k, a->batch, s);
      if (r == 1 && mg == 128 && rows >= 64ll * cus / 2) r = launch_ctm_res<64, 1, 1, 2>(k, a->batch, s);
    } else {
      if (mg == 256 && rows >= 64ll * cus / 2) r = launch_ctm_res<32, 2, 1, 2>(k, a->batch, s);
      if (r == 1 && mg == 256 && rows >= 32ll * cus / 2) r = launch_ctm_res<32, 2, 1, 1>(k, a->batch, s);
      if (r == 1 && mg == 128 && rows >= 128ll * cus / 2) r = launch_ctm_res<32, 1, 1, 4>(k, a->batch, s);
      if (r == 1 && mg == 128 && rows >= 64ll * cus / 2) r = launch_ctm_res<32, 1, 1, 2>(k, a->batch, s);
    }
    if (r <= 0) return r;
  }
  if (mg == 256) {
    const long long c96 = cost(256, 96), c64 = cost(256, 64), c128 = cost(128, 128);
    int pick = (c64 < c96 && c64 <= c128) ? 2 : (c128 < c96 && c128 < c64 ? 3 : 1);
    if (tile >= 1 && tile <= 3) pick = tile;
    // two chunks per barrier where a chunk's MFMAs are few against the window loads' round trip (measured with shader-clock marks: a
    // chunk interval never ran below ~3 k cycles, a pointwise chunk has 1.0-1.5 k of MFMAs): pointwise convs, and <= 9-tap convs on
    // split inputs (copied, so a two-chunk window costs the support waves nothing)
    const bool jc2 = !diag_int("ctm_jc1") && k.NCH % 4 == 0 && (wide_chunk || (a->x_split && (a->ksize - 1) * a->dilation <= 8));
    if (wide_chunk) {  // (256 x 96 tiles on 64-channel chunks need more than 256 registers: 256 x 64 or 128 x 128)
      if (pick == 3 || (pick == 1 && c128 < c64)) return launch_ctm<64, 1, 1, 4>(k, a->batch, s);  // (128-row tiles: one chunk per job, the window registers of two do not fit)
      return jc2 ? launch_ctm<64, 2, 1, 2, 2>(k, a->batch, s) : launch_ctm<64, 2, 1, 2>(k, a->batch, s);
    }
    if (jc2 && pick != 3) return launch_ctm<32, 2, 1, 2, 2>(k, a->batch, s);
    if (pick == 2) return launch_ctm<32, 2, 1, 2>(k, a->batch, s);
    if (pick == 3) return launch_ctm<32, 1, 1, 4>(k, a->batch, s);
    return launch_ctm<32, 2, 1, 3>(k, a->batch, s);
  }
  switch (mg) {
    case 128: return wide_chunk ? launch_ctm<64, 1, 1, 4>(k, a->batch, s) : launch_ctm<32, 1, 1, 4>(k, a->batch, s);
    case 64: return launch_ctm<32, 1, 2, 2>(k, a->batch, s);
    default: return launch_ctm<32, 1, 4, 2>(k, a->batch, s);
  }
}

// MaxPool1d(2, stride 1, padding 1)[:t] over time on a time-major tensor: y[t] = max(x[t - 1], x[t]), y[0] = x[0]  (sublayer/cbhg.py:20,61-62),
// written as fp32 and / or as the split tensor [B][t][hi | lo][C] the next conv stages by copy
namespace mb {
__global__ __launch_bounds__(256) void maxpool2_tm_kernel(const float* __restrict__ x, float* __restrict__ y, h16* __restrict__ ysplit, int T, int C4, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / C4;
    const f32x4 cur = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 prev = (row % T) ? reinterpret_cast<const f32x4*>(x)[i - C4] : cur;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaxf(cur[e], prev[e]);
    if (y) reinterpret_cast<f32x4*>(y)[i] = o;
    if (ysplit) {
      mb_h2 h0, l0, h1, l1;
      split_pair(o[0], o[1], h0, l0);
      split_pair(o[2], o[3], h1, l1);
      const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
      h16* sp = ysplit + row * 2 * (size_t)(4 * C4) + (i - row * C4) * 4;
      *reinterpret_cast<h16x4*>(sp) = hi;
      *reinterpret_cast<h16x4*>(sp + 4 * C4) = lo;
    }
  }
}

// Highway combine (common/highway_network.py:12-17) on time-major tensors: hg [rows][2 C] = (W1 x + b1 | W2 x + b2) from ONE conv launch,
// x [rows][C]  ->  y = g relu(h) + (1 - g) x, g = sigmoid(W2 x + b2), as fp32 [rows][C] and as a split tensor [rows][hi | lo][C]
__global__ __launch_bounds__(256) void highway_tm_kernel(const float* __restrict__ hg, const float* __restrict__ x, float* __restrict__ y,
                                                         h16* __restrict__ ysplit, int C4, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / C4, c4 = i - row * C4;
    const f32x4 h = reinterpret_cast<const f32x4*>(hg)[row * 2 * C4 + c4];
    const f32x4 gp = reinterpret_cast<const f32x4*>(hg)[row * 2 * C4 + C4 + c4];
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = sigmoidf_(gp[e]);
      o[e] = g * fmaxf(h[e], 0.f) + (1.f - g) * xv[e];
    }
    reinterpret_cast<f32x4*>(y)[i] = o;
    if (ysplit) {
      mb_h2 h0, l0, h1, l1;
      split_pair(o[0], o[1], h0, l0);
      split_pair(o[2], o[3], h1, l1);
      const h16x4 hi = {h0[0], h0[1], h1[0], h1[1]}, lo = {l0[0], l0[1], l1[0], l1[1]};
      h16* sp = ysplit + row * 2 * (size_t)(4 * C4) + c4 * 4;
      *reinterpret_cast<h16x4*>(sp) = hi;
      *reinterpret_cast<h16x4*>(sp + 4 * C4) = lo;
    }
  }
}

// A Conv1d to ONE output channel on a time-major fp32 tensor (the GANs' conv_post: models/vocoder/hifigan/models.py:146-148,
// fregan/generator.py:161-163, vits.py:296-297): y[b][t] = act(bias + sum_{j, c} w[j][c] lrelu(x[b][t - pad + j dil][c])).
// 2 k C flops per 4 C bytes read: an HBM stream.  On the MFMA kernel it was 31 of 32 tile rows of zero weights and a write-out of one
// float per row (266 us for 164 MB at 32 x 200 frames); here a workgroup lays 256 + halo rows down in LDS (activation applied once)
// and every thread owns one output row, exact fp32 FMAs (j-major, then c), weights as LDS broadcasts.
constexpr int C1_ROWS = 256;
__global__ __launch_bounds__(256) void conv_c1_tm_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias, float* __restrict__ y,
                                                         int T, int C, int ntaps, int dil, int pad, float slope, int out_act,
                                                         const int* __restrict__ valid, int valid_mul, int tiles_per_item) {
  extern __shared__ __attribute__((aligned(16))) float c1_lds[];
  const int CS = C + 4;                      // row stride in floats (16-byte aligned rows, 4 banks of skew)
  const int halo = (ntaps - 1) * dil;
  float* sw = c1_lds;                        // [ntaps][C]
  float* sx = c1_lds + ntaps * C;            // [C1_ROWS + halo][CS]
  const int b = blockIdx.x / tiles_per_item, t0 = (blockIdx.x - b * tiles_per_item) * C1_ROWS;
  const int Tb = valid ? min(T, valid[b] * valid_mul) : T;
  if (t0 >= Tb) return;  // (uniform)
  const int tid = threadIdx.x;
  for (int i = tid; i < ntaps * C; i += 256) sw[i] = w[i];
  const int c4 = C / 4, rows = C1_ROWS + halo;
  const __amdgpu_buffer_rsrc_t rs = tm_rsrc(x + (long long)b * T * C, (long long)Tb * C * 4);
  for (int i = tid; i < rows * c4; i += 256) {
    const int r = i / c4, q = i - r * c4;
    f32x4 v = tm_load16(rs, (unsigned)((t0 - pad + r) * C + q * 4) * 4u);  // rows outside [0, Tb): zeros
    if (slope != 1.f) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
    }
    *reinterpret_cast<f32x4*>(sx + r * CS + q * 4) = v;
  }
  __syncthreads();
  float acc = bias;
  for (int j = 0; j < ntaps; ++j) {
    const float* xr = sx + (tid + j * dil) * CS;
    const float* wr = sw + j * C;
    for (int q = 0; q < c4; ++q) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + q * 4), wv = *reinterpret_cast<const f32x4*>(wr + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(wv[e], xv[e], acc);
    }
  }
  if (out_act == 2) acc = tanhf(acc);
  if (t0 + tid < Tb) y[(long long)b * T + t0 + tid] = acc;
}
}  // namespace mb

extern "C" int mb_conv_c1_tm(const float* d_x, const float* d_w, float bias, float* d_y, int batch, int t, int c_in, int ksize, int dilation,
                             int pad, float in_slope, int out_act, const int32_t* d_valid, int valid_mul, mb_stream_t stream) {
  MB_REQUIRE(d_x && d_w && d_y, "conv_c1_tm: null pointer");
  MB_REQUIRE(c_in > 0 && c_in % 4 == 0 && ksize >= 1 && dilation >= 1 && 2 * pad == dilation * (ksize - 1),
             "conv_c1_tm: c_in=%d ksize=%d dilation=%d pad=%d (c_in %% 4 == 0, 'same' padding)", c_in, ksize, dilation, pad);
  MB_REQUIRE(out_act == 0 || out_act == 2, "conv_c1_tm: out_act %d (0 none, 2 tanh)", out_act);
  MB_REQUIRE(in_slope > 0.f && in_slope <= 1.f, "conv_c1_tm: in_slope %g outside (0, 1]", in_slope);
  if (batch <= 0 || t <= 0) return MB_OK;
  const size_t lds = sizeof(float) * ((size_t)ksize * c_in + (size_t)(mb::C1_ROWS + (ksize - 1) * dilation) * (c_in + 4));
  MB_REQUIRE(lds <= 160 * 1024, "conv_c1_tm: window of %zu B does not fit in LDS (c_in=%d ksize=%d dilation=%d)", lds, c_in, ksize, dilation);
  MB_REQUIRE((long long)t * c_in * 4 < 0xffffffffll, "conv_c1_tm: an item of %d x %d floats is beyond a buffer descriptor's 4 GB", t, c_in);
  if (lds > 64 * 1024) {  // (per device: the attribute belongs to the function on the current device)
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    MB_HIP(hipGetDevice(&dev));
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (!bit || !(attr_done.load(std::memory_order_acquire) & bit)) {
      MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mb::conv_c1_tm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done.fetch_or(bit, std::memory_order_release);
    }
  }
  const int tiles = mb::cdiv(t, mb::C1_ROWS);
  hipLaunchKernelGGL(mb::conv_c1_tm_kernel, dim3((unsigned)(batch * tiles)), dim3(256), lds, (hipStream_t)stream, d_x, d_w, bias, d_y, t, c_in, ksize,
                     dilation, pad, in_slope, out_act, d_valid, valid_mul > 0 ? valid_mul : 1, tiles);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

extern "C" int mb_maxpool2_tm(const float* d_x, float* d_y, void* d_ysplit, int batch, int t, int channels, mb_stream_t stream) {
  MB_REQUIRE(d_x && (d_y || d_ysplit) && d_x != d_y, "maxpool2_tm: null pointer / in place");
  MB_REQUIRE(channels % 4 == 0, "maxpool2_tm: channels %% 4 != 0");
  if (batch <= 0 || t <= 0 || channels <= 0) return MB_OK;
  const size_t n4 = (size_t)batch * t * (channels / 4);
  hipLaunchKernelGGL(maxpool2_tm_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, d_x, d_y,
                     reinterpret_cast<h16*>(d_ysplit), t, channels / 4, n4);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

extern "C" int mb_highway_tm(const float* d_hg, const float* d_x, float* d_y, void* d_ysplit, long long rows, int channels, mb_stream_t stream) {
  MB_REQUIRE(d_hg && d_x && d_y && d_x != d_y, "highway_tm: null pointer / in place");
  MB_REQUIRE(channels % 4 == 0, "highway_tm: channels %% 4 != 0");
  if (rows <= 0 || channels <= 0) return MB_OK;
  const size_t n4 = (size_t)rows * (channels / 4);
  hipLaunchKernelGGL(highway_tm_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, d_hg, d_x, d_y,
                     reinterpret_cast<h16*>(d_ysplit), channels / 4, n4);
  MB_HIP(hipGetLastError());
  return MB_OK;
}
