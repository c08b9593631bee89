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
  long long x_bstride, y_bstride;  // floats per batch item
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
};

constexpr int CTM_NL = 4;  // support waves

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

template <int CK_, int MT_, int WN_, int NTW_>
__global__ __launch_bounds__(64 * (4 + CTM_NL)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_split_tm_kernel(ConvTmK a) {
  using G = CtmGeom<CK_, MT_, WN_, NTW_>;
  constexpr int CK = G::CK, KB = G::KB, MT = G::MT, WN = G::WN, NTW = G::NTW, MG = G::MG, N1 = G::N1, CKP = G::CKP, MGF = G::MGF;
  constexpr int TD = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int XPL = a.x_rows * CKP;             // halves per plane of an x chunk buffer
  h16* xs = reinterpret_cast<h16*>(lds_raw);  // [nbuf][hi | lo][x_rows][CKP]
  float* ys = reinterpret_cast<float*>(xs + a.nbuf * 2 * XPL);  // [N1][MGF]
  float* bs = ys + N1 * MGF;                  // [MG]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntaps = a.ntaps, NCH = a.NCH;
  // workgroup -> (channel group, its share of the position tiles)
  const int mg = (int)blockIdx.x % a.n_mg, wslot = (int)blockIdx.x / a.n_mg, nslots = (int)gridDim.x / a.n_mg;
  const int my_tiles = (a.n_ntiles - wslot + nslots - 1) / nslots;
  const int njobs = my_tiles * NCH;
  const int m0 = mg * MG;

  for (int i = tid; i < MG; i += 64 * (4 + CTM_NL)) bs[i] = (a.bias && m0 + i < a.c_out) ? a.bias[m0 + i] : 0.f;
  __syncthreads();  // Z

  if (wave >= 4) {
    // ------------------------------ support waves (resblock_pair_split.hip's ring schedule) ------------------------------
    constexpr int PPR = CK / 4;
    constexpr int NSL = 64 * CTM_NL;
    constexpr int LBX = ((N1 + CTM_MAX_HALO) * PPR + NSL - 1) / NSL;
    constexpr int YPR = MG / 4;                            // 16-byte pieces per result row of this channel group
    constexpr int WB = 4;                                  // result pieces per lane per batch
    const float slope = a.in_slope;
    const int total = a.x_rows * PPR;
    const int ltid = tid - 256;
    const bool w_loads = a.res != nullptr || a.accumulate;
    int xrow[LBX];
    unsigned xcol[LBX], xlds[LBX];
#pragma unroll
    for (int i = 0; i < LBX; ++i) {
      const int idx = min(i * NSL + ltid, total - 1);
      xrow[i] = idx / PPR;
      xcol[i] = (unsigned)(idx - xrow[i] * PPR) * 4u;
      xlds[i] = (unsigned)xrow[i] * CKP + xcol[i];
    }
    auto tile_of = [&](int it, int& b, int& t0) __attribute__((always_inline)) {
      const int nt = wslot + it * nslots;
      b = nt / a.tiles_per_item;
      t0 = (nt - b * a.tiles_per_item) * N1;
    };
    auto issue_x = [&](int q, f32x4 (&vx)[LBX]) __attribute__((always_inline)) {
      int b, t0;
      tile_of(q / NCH, b, t0);
      const int c = q % NCH;
      const int Tb = ctm_valid_len(a, b);
      const int tx0 = t0 - a.pad;
      const float* xb = a.x + (long long)b * a.x_bstride;
#pragma unroll
      for (int i = 0; i < LBX; ++i) {
        const int tx = tx0 + xrow[i];
        const unsigned ch = (unsigned)c * CK + xcol[i];
        const bool in = tx >= 0 && tx < Tb && ch < (unsigned)a.c_in;  // rows beyond the item / channels beyond c_in: zeros
        const unsigned off = (unsigned)min(max(tx, 0), a.T - 1) * (unsigned)a.c_in + min(ch, (unsigned)a.c_in - 4u);
        const f32x4 ld = *reinterpret_cast<const f32x4*>(xb + off);
        vx[i] = in ? ld : (f32x4)0.f;
      }
    };
    auto commit_x = [&](int q, const f32x4 (&vx)[LBX]) __attribute__((always_inline)) {
      h16* buf = xs + (q % a.nbuf) * 2 * XPL;
#pragma unroll
      for (int i = 0; i < LBX; ++i) {
        float l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = fmaxf(vx[i][e], vx[i][e] * slope);  // leaky_relu for slopes in (0, 1]; slope 1 = no activation
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
    struct WTile { const float* rb; float* yb; int ytotal, mcols; };
    auto wtile = [&](int it) __attribute__((always_inline)) {
      int b, t0;
      tile_of(it, b, t0);
      const int Tb = ctm_valid_len(a, b);
      const int rows = max(0, min(N1, Tb - t0));
      const long long o = (long long)b * a.y_bstride + (long long)t0 * a.c_out + m0;
      return WTile{a.res ? a.res + o : nullptr, a.y + o, rows * YPR, min(MG, a.c_out - m0)};
    };
    auto write_part = [&](int it, int lo, int hi, int den) __attribute__((always_inline)) {  // batches [nbt lo / den, nbt hi / den)
      const WTile w = wtile(it);
      if (w.ytotal <= 0) return;
      constexpr int BSZ = NSL * WB;
      const int nbt = (w.ytotal + BSZ - 1) / BSZ;
      for (int base = (nbt * lo / den) * BSZ; base < (nbt * hi / den) * BSZ && base < w.ytotal; base += BSZ) {
        f32x4 rx[WB], ry[WB];
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
          }
        }
#pragma unroll
        for (int i = 0; i < WB; ++i) {
          const int idx = base + i * NSL + ltid;
          const unsigned idc = (unsigned)min(idx, w.ytotal - 1);
          const unsigned row = idc / YPR, pc = idc - row * YPR;
          const f32x4 hv = *reinterpret_cast<const f32x4*>(ys + row * MGF + pc * 4);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float f = hv[e];
            if (w_loads) f += rx[i][e];
            f *= a.out_scale;
            if (a.out_act == 2) f = tanhf(f);
            if (w_loads) f += ry[i][e];
            o[e] = f;
          }
          if (ok[i]) {
            if (w.mcols >= 4) *reinterpret_cast<f32x4*>(w.yb + goff[i]) = o;
            else  // fewer than 4 output channels (conv_post: one): element stores
              for (int e = 0; e < w.mcols; ++e) w.yb[row * (unsigned)a.c_out + e] = o[e];
          }
        }
      }
    };
    f32x4 vxA[LBX], vxB[LBX];
    for (int q = 0; q < a.nbuf - 1 && q < njobs; ++q) { issue_x(q, vxA); commit_x(q, vxA); }
    if (a.nbuf - 1 < njobs) issue_x(a.nbuf - 1, vxB);
    for (int it = 0; it < my_tiles; ++it) {
      for (int c = 0; c < NCH; c += 2) {
        const int q = it * NCH + c;
        __syncthreads();  // B_q: job q is staged, the buffer of job q-1 is free
        if (q + a.nbuf < njobs) issue_x(q + a.nbuf, vxA);
        if (q + a.nbuf - 1 < njobs) commit_x(q + a.nbuf - 1, vxB);
        if (it > 0) write_part(it - 1, c, c + 1, NCH);
        __syncthreads();  // B_{q+1}
        if (q + 1 + a.nbuf < njobs) issue_x(q + 1 + a.nbuf, vxB);
        if (q + a.nbuf < njobs) commit_x(q + a.nbuf, vxA);
        if (it > 0) write_part(it - 1, c + 1, c + 2, NCH);
      }
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
  f32x16 acc[MT][NTW];
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
        for (int q = 0; q < 16; ++q) acc[i][n][q] = 0.f;
    // NCH is even and ntaps odd: the ring slot of a chunk's first tap alternates 0, 1 and a tile always starts on slot 0
    for (int c = 0; c < NCH; c += 2) {
      __syncthreads();  // B
      SP_CHUNK(0, xs + ((it * NCH + c) % a.nbuf) * 2 * XPL + lrow * CKP + lcol, CKP, x_tapstep, XPL);
      __syncthreads();  // B
      SP_CHUNK(1, xs + ((it * NCH + c + 1) % a.nbuf) * 2 * XPL + lrow * CKP + lcol, CKP, x_tapstep, XPL);
    }
    __syncthreads();  // YF
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
          for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[i][n][4 * g + e], us, bq[e]);
          *reinterpret_cast<f32x4*>(ys + row * MGF + co0) = v;
        }
      }
    __syncthreads();  // Y
  }
}

template <class G>
static size_t ctm_lds_bytes(int ntaps, int dil, int nbuf) {
  const int x_rows = G::N1 + (ntaps - 1) * dil;
  return (size_t)nbuf * 2 * x_rows * G::CKP * sizeof(h16) + (size_t)G::N1 * G::MGF * sizeof(float) + G::MG * sizeof(float);
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

template <int CK, int MT, int WN, int NTW>
static int launch_ctm(ConvTmK k, int batch, hipStream_t s) {
  using G = CtmGeom<CK, MT, WN, NTW>;
  k.x_rows = G::N1 + (k.ntaps - 1) * k.dil;
  k.tiles_per_item = cdiv(k.T, G::N1);
  k.n_ntiles = k.tiles_per_item * batch;
  k.n_mg = cdiv(k.c_out, G::MG);
  int nbuf = 2;
  while (nbuf < 4 && ctm_lds_bytes<G>(k.ntaps, k.dil, nbuf + 1) <= (size_t)160 * 1024) ++nbuf;
  MB_REQUIRE(ctm_lds_bytes<G>(k.ntaps, k.dil, nbuf) <= (size_t)160 * 1024, "conv_split_tm: the window does not fit LDS");
  k.nbuf = nbuf;
  static std::atomic<unsigned long long> attr_done{0};
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
  if (!bit || !(attr_done.load(std::memory_order_acquire) & bit)) {
    MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_split_tm_kernel<CK, MT, WN, NTW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int slots = std::max(1, std::min(k.n_ntiles, ctm_cus() / k.n_mg));
  hipLaunchKernelGGL((conv_split_tm_kernel<CK, MT, WN, NTW>), dim3(slots * k.n_mg), dim3(64 * (4 + CTM_NL)), ctm_lds_bytes<G>(k.ntaps, k.dil, nbuf), s, k);
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
  MB_REQUIRE(a->out_act == 0 || a->out_act == 2, "conv_split_tm: out_act %d (0 = none, 2 = tanh)", a->out_act);
  MB_REQUIRE((!a->d_res && !a->accumulate) || a->c_out % 4 == 0, "conv_split_tm: residual / accumulating launches need c_out %% 4 == 0");
  if (a->batch <= 0 || a->t <= 0) return MB_OK;
  MB_REQUIRE((long long)a->t * std::max(a->c_in, a->c_out) < (1ll << 31), "conv_split_tm: an item of %d rows is beyond the 32-bit offsets", a->t);
  ConvTmK k;
  memset(&k, 0, sizeof(k));
  k.x = a->d_x; k.y = a->d_y; k.w = reinterpret_cast<const h16*>(a->d_wpacked); k.bias = a->d_bias; k.res = a->d_res;
  k.x_bstride = (long long)a->t * a->c_in; k.y_bstride = (long long)a->t * a->c_out;
  k.T = a->t; k.c_in = a->c_in; k.c_out = a->c_out; k.ntaps = a->ksize; k.dil = a->dilation; k.pad = a->pad;
  k.NCH = ctm_nch(a->c_in);
  k.in_slope = a->in_slope; k.us = a->unscale; k.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
  k.out_act = a->out_act; k.accumulate = a->accumulate;
  k.valid = a->d_valid; k.valid_mul = a->valid_mul > 0 ? a->valid_mul : 1;
  k.range_events = conv_range_word();
  hipStream_t s = (hipStream_t)stream;
  const int mg = ctm_mg(a->c_out);
  if (ctm_ck(a->c_in) == 16) {
    switch (mg) {
      case 256: return launch_ctm<16, 2, 1, 3>(k, a->batch, s);
      case 128: return launch_ctm<16, 1, 1, 4>(k, a->batch, s);
      case 64: return launch_ctm<16, 1, 2, 2>(k, a->batch, s);
      default: return launch_ctm<16, 1, 4, 2>(k, a->batch, s);
    }
  }
  switch (mg) {
    case 256: return launch_ctm<32, 2, 1, 3>(k, a->batch, s);
    case 128: return launch_ctm<32, 1, 1, 4>(k, a->batch, s);
    case 64: return launch_ctm<32, 1, 2, 2>(k, a->batch, s);
    default: return launch_ctm<32, 1, 4, 2>(k, a->batch, s);
  }
}
