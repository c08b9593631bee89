// Conv1d / ConvTranspose1d as an fp32 MFMA implicit GEMM for gfx950.
//
//   y[b][co][t] = epilogue( sum_{ci,j} W[co][ci][j] * pre(x[b][ci][t_in(t,j)]) )
//
// GEMM view: M = c_out (32-row MFMA tiles), N = output time positions (32-col
// tiles, 2 per wave), K = c_in * taps walked 2 channels per
// v_mfma_f32_32x32x2_f32.  fp32 inputs / fp32 accumulate keeps the result an
// exact k-ordered fmaf chain (MI355X guide section 3), which is what lets the GAN
// vocoders meet the 1e-4 RMS audio parity bar against the fp32 reference.
//
// Data movement per workgroup (256 threads = 4 waves):
//   * x tile: CK=16 input channels x (NT + halo) positions staged through LDS,
//     coalesced along time, input activation (leaky-relu / max-pool) fused
//     into the staging pass -> every tap / output-channel tile re-reads LDS.
//   * weights: pre-packed on the host in A-fragment order so each wave fetches
//     one coalesced 1 KiB float4 row per (8 channels x 1 tap); they stay
//     L2-resident across the grid.  Next fragment is prefetched across the
//     staging barrier.
//   * transposed conv (stride `up`) is run as `up` polyphase sub-convolutions
//     (grid.z), each a plain conv with ksize/up taps.
//   * epilogue: bias, activation, BatchNorm scale/shift, residual, accumulate
//     fused; optional time-major store by swapping the MFMA operands (the A and
//     B fragment lane maps are identical, so D comes out transposed for free).
#include "common.h"
#include <cmath>

namespace mb {

static constexpr int CK = 16;  // input channels staged per LDS chunk

struct ConvK {
  const float* x; const float* w; const float* bias; const float* res;
  const float* post_scale; const float* post_shift; const float* gate; float* y;
  long long x_bstride, y_bstride, res_bstride;
  int c_in, cin_pad, c_out, t_in, t_out;
  int ntaps, up, down, step, min_off, span;
  int off0[8];
  int in_act; float in_slope, in_scale;
  int out_act, accumulate, in_repeat;
  float out_scale, out_slope;
  const int* valid; int valid_mul;  // ragged batches (mb_conv1d_args.d_valid)
  unsigned* range_events;           // diagnostics (MBHIP_CONV_RANGE_CHECK=1): staged values beyond the split path's range are counted here
};

template <int WM, int WN, bool TR>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(ConvK a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = 64 * WN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int p = blockIdx.z % a.up, b = blockIdx.z / a.up;
  const int q0 = blockIdx.x * NT;
  const int Tq = (a.t_out - p + a.up - 1) / a.up;  // outputs of this phase
  if (q0 >= Tq) return;
  // ragged batch: this item's valid input / output extent (positions beyond are its zero padding / never consumed)
  const int t_lim = a.valid ? min(a.t_in, a.valid[b] * a.valid_mul * a.in_repeat) : a.t_in;
  if (a.valid) {
    const int t_out_b = a.up > 1 ? t_lim * a.up : t_lim + (a.t_out - a.t_in);
    if (q0 * a.up + p >= t_out_b) return;
  }
  const int n_mt = (a.c_out + 31) >> 5;
  const int mt = blockIdx.y * WM + wm;
  const bool active = mt < n_mt;
  const int rowlen = NT * a.down + a.span;  // down > 1: strided conv, LDS holds every input position
  const int n_cb = a.cin_pad >> 3;
  const int n_w = n_cb * a.ntaps;  // A fragments this wave walks
  const float* xb = a.x + (long long)b * a.x_bstride;
  const float4* wp = reinterpret_cast<const float4*>(a.w) +
                     ((size_t)(p * n_mt + (active ? mt : 0)) * n_w) * 64 + lane;
  const int off_base = a.off0[p] - a.min_off;  // >= 0 for every tap

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  int wi = 0;
  float4 av = active ? wp[0] : make_float4(0.f, 0.f, 0.f, 0.f);

  for (int c0 = 0; c0 < a.cin_pad; c0 += CK) {
    __syncthreads();  // previous chunk fully consumed
    for (int c = wave; c < CK; c += 4) {
      const int ci = c0 + c;
      const bool cok = ci < a.c_in;
      const int t_src = a.in_repeat > 1 ? a.t_in / a.in_repeat : a.t_in;  // stored row length
      const float* xr = xb + (long long)ci * t_src;
      float* lrow = lds + c * rowlen;
      for (int tt = lane; tt < rowlen; tt += 64) {
        const int ti = q0 * a.down + a.min_off + tt;
        float v = 0.f;
        if (cok && ti >= 0 && ti < t_lim) {
          if (a.in_repeat > 1) {
            v = xr[ti / a.in_repeat] * a.in_scale;
            if (a.in_act == 1) v = v > 0.f ? v : v * a.in_slope;
          } else if (a.in_act == 2) {  // MaxPool1d(2, stride 1, pad 1)[:T] (cbhg.py:20,61)
            v = xr[ti];
            if (ti > 0) v = fmaxf(v, xr[ti - 1]);
          } else {
            v = xr[ti];
            v *= a.in_scale;
            if (a.in_act == 1) v = v > 0.f ? v : v * a.in_slope;
          }
        }
        lrow[tt] = v;
      }
    }
    __syncthreads();
    if (active) {
      const int ncb2 = min(CK / 8, n_cb - (c0 >> 3));
      for (int cb2 = 0; cb2 < ncb2; ++cb2) {
        const float* lbase = lds + (cb2 * 8 + (lane >> 5)) * rowlen + (wn * 64 + (lane & 31)) * a.down + off_base;
        const int h32 = 32 * a.down;
        for (int j = 0; j < a.ntaps; ++j) {
          ++wi;
          const float4 an = (wi < n_w) ? wp[(size_t)wi * 64] : av;
          const float* lp = lbase + j * a.step;
          const float b00 = lp[0], b01 = lp[h32];
          const float b10 = lp[2 * rowlen], b11 = lp[2 * rowlen + h32];
          const float b20 = lp[4 * rowlen], b21 = lp[4 * rowlen + h32];
          const float b30 = lp[6 * rowlen], b31 = lp[6 * rowlen + h32];
          if (!TR) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b00, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b01, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b10, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b11, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b20, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b21, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b30, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b31, acc1, 0, 0, 0);
          } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b00, av.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b01, av.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b10, av.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b11, av.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b20, av.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b21, av.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b30, av.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b31, av.w, acc1, 0, 0, 0);
          }
          av = an;
        }
      }
    }
  }
  if (!active) return;

  // ---- epilogue ----
  // D fragment map (32x32): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* yb = a.y + (long long)b * a.y_bstride;
  const float* rb = a.res ? a.res + (long long)b * a.res_bstride : nullptr;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int drow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int dcol = lane & 31;
      const int co = mt * 32 + (TR ? dcol : drow);
      const int q = q0 + wn * 64 + half * 32 + (TR ? drow : dcol);
      const int t = q * a.up + p;
      if (co < a.c_out && q < Tq) {
        float v = half ? acc1[r] : acc0[r];
        if (a.bias) v += a.bias[co];
        if (a.out_act == 1) v = fmaxf(v, 0.f);
        else if (a.out_act == 2) v = tanhf(v);
        else if (a.out_act == 3) v = 1.0f / (1.0f + expf(-v));
        else if (a.out_act == 5) v = v > 0.f ? v : v * a.out_slope;
        if (a.post_scale) v = v * a.post_scale[co] + a.post_shift[co];
        const long long o = TR ? ((long long)t * a.c_out + co) : ((long long)co * a.t_out + t);
        if (a.out_act == 4) {
          const float g = a.gate[(long long)b * a.y_bstride + o];
          v = g * fmaxf(v, 0.f) + (1.f - g) * rb[o];
        } else if (rb) v += rb[o];
        v *= a.out_scale;
        if (a.accumulate) v += yb[o];
        yb[o] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same implicit GEMM on the fp16 matrix pipe with error compensation ("split" path, the default for c_in >= 16):
//   w * 2^s = wh + wl          (wh = fp16(w 2^s), wl = fp16(w 2^s - wh); s per conv so that max |w| 2^s is in [2^13, 2^14):
//                               the low halves of all but negligible weights are fp16 normals)
//   x = xh + 2^-11 xl          (xh = fp16(x), xl = fp16((x - xh) 2^11): the residual is stored SCALED, so it is an fp16 normal
//                               whenever xh is -- round 3 stored it unscaled and a stage with |x| ~ 1e-3 kept 14 bits, not 22:
//                               1.1e-4 of the output RMS against 6e-6 for the fp32-input kernel, tests/test_conv1d_gpu.py)
//   acc += wl.xh + ws.xl + wh.xh   with the third weight image ws = fp16(wh 2^-11): three v_mfma_f32_32x32x16_f16 per
//                               (16 channels, tap), fp32 accumulate;  y = acc * 2^-s ...
// The dropped wl.xl term is 2^-22 of a product and each operand keeps 22 bits down to |x| = 2^-14 (an absolute 2^-36 below):
// fp32-grade results at 1/5.3 of the matrix-pipe time of the fp32-input MFMA (3 x 32 cycles per 32x32x16 block instead of 8 x 64).
// Range: |x| <= 65504 (x is clamped to fp16's largest value before it is split; audio / mel activations are O(1..100));
// MBHIP_CONV_RANGE_CHECK=1 counts the values beyond it (mb_conv1d_range_events).
// MBHIP_CONV_SPLIT=0 selects the exact fp32-input kernel above (tests/test_conv1d_gpu.py runs both).
// Layout: x staged per chunk of SCK = 32 channels as [position][32 hi | 32 lo | pad] fp16 rows of 144 bytes (16-byte
// B fragments = 8 consecutive channels of one position, conflict-free for ds_read_b128 / ds_write_b128: 144 / 16 is odd);
// weights pre-split on the host in A-fragment order [phase][mt][16-channel step][tap][hi | lo | hi 2^-11][lane][8].
typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
static constexpr int SCK = 32;
static constexpr int SROW = SCK * 4 + 16;  // bytes per staged position

// One workgroup = 4 waves on 128 output positions (WM x WN waves, NTW 32-position tiles per wave: 4x1x4, 2x2x2 or 1x4x1).
// Pipeline per 32-channel chunk: the NEXT chunk's x values are already in flight to registers (SITEMS x 8 floats per
// thread, issued before this chunk's MFMAs), the chunk itself is computed from LDS, then the registers are split into
// hi / lo halves and stored: global latency sits behind the MFMAs, three workgroups per CU hide the rest (the register budget is
// capped for three waves per SIMD: 158 VGPRs for 4x1x4, accumulators included -- 172 us against 185 at two for the 128-channel k = 7
// conv of HiFi-GAN 32 x 200; a 4x1x8 form with 256 positions per workgroup -- half the weight stream per output -- needs 384
// registers or spills: 342 us).
// The first version of this kernel was VALU-bound, not matrix-bound: ~5500 vector instructions per wave next to 168 MFMAs
// (rocprofv3 SQ_INSTS_VALU; the do-nothing skeleton alone cost half the kernel).  Hence: loads are unconditional from
// clamped 32-bit offsets (one add per element, the select happens on the value), work items (position, 8-channel group) are
// dealt evenly over the 256 threads, every stride is hoisted, and the epilogue has a straight-line path for interior tiles.
static constexpr int SNT = 128;    // output positions per workgroup
static constexpr int SITEMS = 4;   // (position, 8-channel group) items per thread and chunk: rowlen <= 256
static constexpr float SPLIT_RANGE = 65504.f;  // |x| beyond this saturates (the clamp of split_store2)

__device__ __forceinline__ void split_store2(char* row, const int grp, const float (&v)[8]) {
  h16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {  // (packed: common.h split_pair)
    mb_h2 h, l;
    split_pair(v[e], v[e + 1], h, l);
    hi[e] = h[0]; hi[e + 1] = h[1]; lo[e] = l[0]; lo[e + 1] = l[1];
  }
  *reinterpret_cast<h16x8*>(row + grp * 16) = hi;
  *reinterpret_cast<h16x8*>(row + SCK * 2 + grp * 16) = lo;
}

template <int WM, int WN, int NTW, bool TR, bool POOL>
__device__ __forceinline__ void conv1d_split_body(const ConvK& a, const uint4* __restrict__ wsplit, const float* __restrict__ whdr, const int nbuf, char* slds) {
  constexpr int SNT = 32 * NTW * WN;  // output positions of this instance's workgroup: 128, or 32 for short rows (shadows the default)
  static_assert(SNT == 128 || SNT == 32, "tile shape");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int p = blockIdx.z % a.up, b = blockIdx.z / a.up;
  const int q0 = blockIdx.x * SNT;
  const int Tq = (a.t_out - p + a.up - 1) / a.up;
  if (q0 >= Tq) return;
  const int t_lim = a.valid ? min(a.t_in, a.valid[b] * a.valid_mul * a.in_repeat) : a.t_in;
  if (a.valid) {
    const int t_out_b = a.up > 1 ? t_lim * a.up : t_lim + (a.t_out - a.t_in);
    if (q0 * a.up + p >= t_out_b) return;
  }
  const int n_mt = (a.c_out + 31) >> 5;
  const int mt = blockIdx.y * WM + wm;
  const bool active = mt < n_mt;
  const int rowlen = SNT * a.down + a.span;  // <= 256 (host)
  const int n_ks = (a.c_in + 15) >> 4;
  const int n_w = n_ks * a.ntaps;
  const float* xb = a.x + (long long)b * a.x_bstride;
  const uint4* wp = wsplit + ((size_t)(p * n_mt + (active ? mt : 0)) * n_w) * 192 + lane;
  const int off_base = a.off0[p] - a.min_off;
  const int t_src = a.in_repeat > 1 ? a.t_in / a.in_repeat : a.t_in;

  // ---- staging items of this thread: item = tid + 256 i -> row = item >> 2 (position of the window), grp = item & 3 ----
  int lds_off[SITEMS];        // byte offset of the row (hi part of group 0) or -1: no such row
  unsigned x_off[SITEMS];     // element offset of (channel grp * 8 of chunk 0, position) inside this batch item, clamped to a legal one
  bool x_ok[SITEMS], x_prev[SITEMS];
  const int ti0 = q0 * a.down + a.min_off;
#pragma unroll
  for (int it = 0; it < SITEMS; ++it) {
    const int item = tid + 256 * it, row = item >> 2, grp = item & 3;
    const int ti = ti0 + row;
    lds_off[it] = row < rowlen ? row * SROW + grp * 16 : -1;
    x_ok[it] = row < rowlen && ti >= 0 && ti < t_lim;
    const int tsi = x_ok[it] ? (a.in_repeat > 1 ? ti / a.in_repeat : ti) : 0;
    x_prev[it] = POOL && x_ok[it] && ti > 0;
    x_off[it] = (unsigned)(grp * 8) * (unsigned)t_src + (unsigned)tsi;
  }
  const unsigned chunk_step = (unsigned)SCK * (unsigned)t_src;
  const unsigned last_ch = (unsigned)(a.c_in - 1) * (unsigned)t_src;  // offsets beyond the last channel are clamped (value zeroed)

  float pre[SITEMS][8], pre2[POOL ? SITEMS : 1][8];
  auto issue = [&](const int c0) {  // global loads of chunk c0 -> registers (nothing waits here)
#pragma unroll
    for (int it = 0; it < SITEMS; ++it) {
      unsigned o = x_off[it] + (unsigned)(c0 / SCK) * chunk_step;
      const int cbase = c0 + (tid & 3) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = x_ok[it] && cbase + e < a.c_in;
        const unsigned oc = min(o, last_ch + (unsigned)(t_src - 1));
        const float v = xb[oc];
        pre[it][e] = ok ? v : 0.f;
        if (POOL) { const float v2 = xb[oc > 0 ? oc - 1 : 0]; pre2[it][e] = (ok && x_prev[it]) ? v2 : -INFINITY; }
        o += (unsigned)t_src;
      }
    }
  };
  const float slope_eff = a.in_act == 1 ? a.in_slope : 1.f;
  auto store = [&](const int boff) {  // registers -> fused input activation -> fp16 hi / lo rows (buffer at byte offset boff)
#pragma unroll
    for (int it = 0; it < SITEMS; ++it) {
      if (lds_off[it] < 0) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = pre[it][e];
        if (POOL) x = fmaxf(x, pre2[it][e]);  // MaxPool1d(2, stride 1, pad 1)[:T] (cbhg.py:20,61)
        else {
          x *= a.in_scale;                        // (1.0 when unused: exact)
          x = fmaxf(x, x * slope_eff);            // leaky_relu for slopes in (0, 1]: x > 0 ? x : slope x, bit for bit; slope_eff = 1 = no activation
        }
        v[e] = x;
      }
      if (a.range_events) {  // diagnostics only (uniform branch): x saturates beyond 65504
        int n_out = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) n_out += !(fabsf(v[e]) <= SPLIT_RANGE) ? 1 : 0;  // counts NaN / Inf as well
        if (n_out) atomicAdd(a.range_events, (unsigned)n_out);
      }
      split_store2(slds + boff + lds_off[it], 0, v);
    }
  };

  f32x16 acc[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

  int wi = 0;
  uint4 ah = make_uint4(0, 0, 0, 0), al = make_uint4(0, 0, 0, 0), as = make_uint4(0, 0, 0, 0);
  if (active) { ah = wp[0]; al = wp[64]; as = wp[128]; }

  const int cin16 = n_ks * 16;
  const int tap_stride = a.step * SROW, tile_stride = 32 * a.down * SROW;
  const char* lbase0 = slds + ((wn * NTW * 32 + (lane & 31)) * a.down + off_base) * SROW + (lane >> 5) * 16;
  // nbuf = 2 (many chunks, e.g. the 2560-channel CBHG projection): the next chunk is stored into the other buffer while slower
  // waves still compute -- one barrier per chunk; nbuf = 1: half the LDS, more workgroups per CU, two barriers per chunk.
  const int buf_bytes = rowlen * SROW;
  int cur = 0;
  issue(0);
  store(0);
  if (SCK < cin16) issue(SCK);
  for (int c0 = 0; c0 < cin16; c0 += SCK) {
    __syncthreads();  // chunk c0 is in LDS[cur] (and every wave is done with the other buffer)
    if (active) {
      const int nks2 = min(SCK, cin16 - c0) >> 4;
      for (int ks2 = 0; ks2 < nks2; ++ks2) {
        const char* lp = lbase0 + cur * buf_bytes + ks2 * 32;
        for (int j = 0; j < a.ntaps; ++j, lp += tap_stride) {
          ++wi;
          const h16x8 Ah = __builtin_bit_cast(h16x8, ah), Al = __builtin_bit_cast(h16x8, al), As = __builtin_bit_cast(h16x8, as);
          if (wi < n_w) { ah = wp[(size_t)wi * 192]; al = wp[(size_t)wi * 192 + 64]; as = wp[(size_t)wi * 192 + 128]; }
#pragma unroll
          for (int n = 0; n < NTW; ++n) {  // (three products per accumulator in a row: interleaving two tiles' chains measured no faster)
            const h16x8 Bh = *reinterpret_cast<const h16x8*>(lp + n * tile_stride);
            const h16x8 Bl = *reinterpret_cast<const h16x8*>(lp + n * tile_stride + SCK * 2);
            if (!TR) {
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(As, Bl, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc[n], 0, 0, 0);
            } else {
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bh, Al, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bl, As, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bh, Ah, acc[n], 0, 0, 0);
            }
          }
        }
      }
    }
    if (c0 + SCK < cin16) {  // the next chunk has arrived in the registers by now
      if (nbuf == 1) __syncthreads();  // one buffer: every wave is done reading it
      else cur ^= 1;
      store(cur * buf_bytes);
      if (c0 + 2 * SCK < cin16) issue(c0 + 2 * SCK);
    }
  }
  if (!active) return;

  // ---- epilogue.  D fragment (32x32): col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); without TR row = output channel,
  //      col = position; with TR (operands swapped) row = position, col = output channel.  32-bit element offsets inside the batch
  //      item (c_out * t_out < 2^31, checked on the host), built by additions. ----
  const float w_unscale = whdr[0];  // 2^-s of the host-side weight scaling
  float* yb = a.y + (long long)b * a.y_bstride;
  const float* rb = a.res ? a.res + (long long)b * a.res_bstride : nullptr;
  const float* gb = a.out_act == 4 ? a.gate + (long long)b * a.y_bstride : nullptr;
  const int dcol = lane & 31, drow0 = 4 * (lane >> 5);
  const unsigned ch_stride = TR ? 1u : (unsigned)a.t_out;
  const unsigned pos_stride = (TR ? (unsigned)a.c_out : 1u) * (unsigned)a.up;
  const unsigned row_step = TR ? pos_stride : ch_stride;
  const int co_b = mt * 32 + (TR ? dcol : drow0);
  const bool simple = a.out_act == 0 || a.out_act == 1 || a.out_act == 5;
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    const int q_b = q0 + (wn * NTW + n) * 32 + (TR ? drow0 : dcol);
    const unsigned o_b = (unsigned)co_b * ch_stride + ((unsigned)q_b * (unsigned)a.up + (unsigned)p) * (TR ? (unsigned)a.c_out : 1u);
    const bool full = (TR ? (mt * 32 + 31 < a.c_out && q_b + 27 < Tq) : (mt * 32 + 31 < a.c_out && q0 + (wn * NTW + n) * 32 + 31 < Tq));
    // every residual / running-sum value of the tile is requested before the first is used: one round trip per tile (a
    // load -> add -> store chain per element made the second conv of a ResBlock unit twice as slow as the first)
    float rv[16], yv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      const bool ok = full || ((TR ? co_b : co_b + dr) < a.c_out && (TR ? q_b + dr : q_b) < Tq);
      const unsigned o = ok ? o_b + (unsigned)dr * row_step : 0u;
      rv[r] = rb ? rb[o] : 0.f;
      yv[r] = a.accumulate ? yb[o] : 0.f;
    }
    if (simple && !gb) {  // no / relu / leaky-relu activation: no branch per element (absent operands read as 0 / 1 through selects)
      const float* bp = a.bias ? a.bias : whdr;
      const float* psp = a.post_scale ? a.post_scale : whdr;
      const float* ptp = a.post_scale ? a.post_shift : whdr;
      const float bsel = a.bias ? 1.f : 0.f, psel = a.post_scale ? 1.f : 0.f;
      const float oslope = a.out_act == 1 ? 0.f : (a.out_act == 5 ? a.out_slope : 1.f);  // v > 0 ? v : v * oslope
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        const int co = TR ? co_b : co_b + dr, q = TR ? q_b + dr : q_b;
        const int cc = min(co, a.c_out - 1);
        float v = fmaf(acc[n][r], w_unscale, bp[a.bias ? cc : 0] * bsel);
        v = v > 0.f ? v : v * oslope;
        v = fmaf(v, fmaf(psp[a.post_scale ? cc : 0], psel, 1.f - psel), ptp[a.post_scale ? cc : 0] * psel);
        v += rv[r];
        v *= a.out_scale;
        v += yv[r];
        if (full || (co < a.c_out && q < Tq)) yb[o_b + (unsigned)dr * row_step] = v;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        const int co = TR ? co_b : co_b + dr, q = TR ? q_b + dr : q_b;
        const unsigned o = o_b + (unsigned)dr * row_step;
        const int cc = min(co, a.c_out - 1);
        float v = acc[n][r] * w_unscale;
        if (a.bias) v += a.bias[cc];
        if (a.out_act == 1) v = fmaxf(v, 0.f);
        else if (a.out_act == 2) v = tanhf(v);
        else if (a.out_act == 3) v = 1.0f / (1.0f + expf(-v));
        else if (a.out_act == 5) v = v > 0.f ? v : v * a.out_slope;
        if (a.post_scale) v = v * a.post_scale[cc] + a.post_shift[cc];
        if (full || (co < a.c_out && q < Tq)) {
          if (gb) { const float g = gb[o]; v = g * fmaxf(v, 0.f) + (1.f - g) * rv[r]; }  // highway (rare: per-element gate load)
          else v += rv[r];
          v *= a.out_scale;
          v += yv[r];
          yb[o] = v;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // one tile's operands in flight at a time
  }
}

// The register budget is capped for three waves per SIMD (see above); the max-pool instances hold a second value per staged element
// (32 more registers) and would spill under that cap: they keep the allocator's own choice (two waves).
template <int WM, int WN, int NTW, bool TR, bool POOL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv1d_split_kernel(ConvK a, const uint4* __restrict__ wsplit, const float* __restrict__ whdr, const int nbuf) {
  extern __shared__ __attribute__((aligned(16))) char slds[];
  static_assert(!POOL, "the pooled instances are conv1d_split_pool_kernel");
  conv1d_split_body<WM, WN, NTW, TR, false>(a, wsplit, whdr, nbuf, slds);
}
template <int WM, int WN, int NTW, bool TR>
__global__ __launch_bounds__(256) void conv1d_split_pool_kernel(ConvK a, const uint4* __restrict__ wsplit, const float* __restrict__ whdr, const int nbuf) {
  extern __shared__ __attribute__((aligned(16))) char slds[];
  conv1d_split_body<WM, WN, NTW, TR, true>(a, wsplit, whdr, nbuf, slds);
}

static int conv_geometry(const mb_conv1d_args* a, ConvK* k) {
  MB_REQUIRE(a->up >= 1 && a->up <= 8, "conv1d: up=%d out of range", a->up);
  MB_REQUIRE(a->ksize >= 1 && a->c_in >= 1 && a->c_out >= 1, "conv1d: bad shape");
  k->up = a->up;
  k->down = a->down > 1 ? a->down : 1;
  MB_REQUIRE(k->down == 1 || (a->up == 1 && a->in_repeat <= 1 && !a->d_valid),
             "conv1d: down=%d (strided conv) excludes up / in_repeat / d_valid", a->down);
  if (a->up == 1) {
    k->ntaps = a->ksize;
    k->step = a->dilation;
    k->off0[0] = -a->pad;
    k->min_off = -a->pad;
    k->span = (a->ksize - 1) * a->dilation;
  } else {
    MB_REQUIRE(a->ksize % a->up == 0, "conv_transpose1d: ksize %d not a multiple of stride %d",
               a->ksize, a->up);
    MB_REQUIRE(a->dilation == 1, "conv_transpose1d: dilation unsupported");
    k->ntaps = a->ksize / a->up;
    k->step = -1;
    int cmax = 0;
    for (int p = 0; p < a->up; ++p) {
      k->off0[p] = (p + a->pad) / a->up;
      if (k->off0[p] > cmax) cmax = k->off0[p];
    }
    k->min_off = -(k->ntaps - 1);
    k->span = cmax + (k->ntaps - 1);
  }
  return MB_OK;
}

}  // namespace mb

using namespace mb;

// image = [fp32 A fragments: n_mt * n_cb * ksize * 256 floats][split header: 64 floats, [0] = 2^-s][split A fragments:
// n_mt * n_ks * ksize * (hi 512 + lo 512) halves]
static size_t conv_f32_image_floats(int c_out, int c_in, int ksize) {
  const int n_mt = (c_out + 31) / 32, n_cb = (c_in + 7) / 8;
  return (size_t)n_mt * n_cb * ksize * 256;  // up phases x (ksize/up) taps == ksize
}
static size_t conv_split_image_floats(int c_out, int c_in, int ksize) {
  const int n_mt = (c_out + 31) / 32, n_ks = (c_in + 15) / 16;
  return (size_t)n_mt * n_ks * ksize * 768;  // three fp16 A fragments (hi | lo | hi 2^-11) of 64 lanes x 8 per (mt, 16-channel step, tap)
}
extern "C" size_t mb_conv1d_packed_floats(int c_out, int c_in, int ksize, int up) {
  (void)up;
  return conv_f32_image_floats(c_out, c_in, ksize) + 64 + conv_split_image_floats(c_out, c_in, ksize);
}

extern "C" int mb_conv1d_pack(const float* h_w, int c_out, int c_in, int ksize, int up,
                              int transposed, int pad, float* h_packed) {
  MB_REQUIRE(up >= 1 && (transposed || up == 1), "conv1d_pack: up>1 needs transposed=1");
  MB_REQUIRE(ksize % up == 0, "conv1d_pack: ksize %% up != 0");
  const int n_mt = (c_out + 31) / 32, n_cb = (c_in + 7) / 8, ntaps = ksize / up;
  size_t o = 0;
  for (int p = 0; p < up; ++p) {
    const int j0 = transposed ? (p + pad) % up : 0;
    for (int mt = 0; mt < n_mt; ++mt)
      for (int cb = 0; cb < n_cb; ++cb)
        for (int j = 0; j < ntaps; ++j) {
          const int jj = transposed ? j0 + j * up : j;
          for (int lane = 0; lane < 64; ++lane)
            for (int q = 0; q < 4; ++q) {
              const int co = mt * 32 + (lane & 31);
              const int ci = cb * 8 + q * 2 + (lane >> 5);
              float v = 0.f;
              if (co < c_out && ci < c_in)
                v = transposed ? h_w[((size_t)ci * c_out + co) * ksize + jj]
                               : h_w[((size_t)co * c_in + ci) * ksize + jj];
              h_packed[o++] = v;
            }
        }
  }
  // ---- split image (conv1d_split_kernel): weights scaled by a power of two so that their low halves are fp16 normals ----
  const size_t total = (size_t)c_out * c_in * ksize;
  float wmax = 0.f;
  for (size_t i = 0; i < total; ++i) wmax = std::max(wmax, std::fabs(h_w[i]));
  int sexp = 0;
  if (wmax > 0.f && std::isfinite(wmax)) {
    int e2;
    std::frexp(wmax, &e2);            // wmax = f * 2^e2, f in [0.5, 1)
    sexp = std::max(-24, std::min(40, 14 - e2));  // wmax * 2^sexp in [2^13, 2^14)
  }
  const float scale = std::ldexp(1.f, sexp);
  float* hdr = h_packed + o;
  for (int i = 0; i < 64; ++i) hdr[i] = 0.f;
  hdr[0] = std::ldexp(1.f, -sexp);
  h16* hp = reinterpret_cast<h16*>(hdr + 64);
  const int n_ks = (c_in + 15) / 16;
  size_t oh = 0;
  for (int p = 0; p < up; ++p) {
    const int j0 = transposed ? (p + pad) % up : 0;
    for (int mt = 0; mt < n_mt; ++mt)
      for (int ks = 0; ks < n_ks; ++ks)
        for (int j = 0; j < ntaps; ++j) {
          const int jj = transposed ? j0 + j * up : j;
          for (int part = 0; part < 3; ++part)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                // A fragment of v_mfma_f32_32x32x16_f16: lane l holds A[m = l & 31][k = 8 * (l >> 5) + e]
                const int co = mt * 32 + (lane & 31);
                const int ci = ks * 16 + (lane >> 5) * 8 + e;
                float v = 0.f;
                if (co < c_out && ci < c_in)
                  v = (transposed ? h_w[((size_t)ci * c_out + co) * ksize + jj] : h_w[((size_t)co * c_in + ci) * ksize + jj]) * scale;
                const h16 hi = (h16)v;
                hp[oh++] = part == 0 ? hi : part == 1 ? (h16)(v - (float)hi) : (h16)((float)hi * (1.f / 2048.f));
              }
        }
  }
  return MB_OK;
}

// Range diagnostics of the split path: with MBHIP_CONV_RANGE_CHECK=1 every conv1d_split_kernel launch
// counts the staged input values with |x| > 65504 (or NaN / Inf) -- values the fp32 reference would carry and the hi / lo
// clamp saturates silently -- into one device word per device; mb_conv1d_range_events reads (and optionally clears) it.
static unsigned* g_range_word[16] = {};
static unsigned* range_word() {
  const char* e = getenv("MBHIP_CONV_RANGE_CHECK");
  if (!e || atoi(e) == 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_range_word[dev]) {
    unsigned* p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, sizeof(unsigned)) != hipSuccess) { (void)hipFree(p); return nullptr; }
    g_range_word[dev] = p;
  }
  return g_range_word[dev];
}

namespace mb { unsigned* conv_range_word() { return range_word(); } }

extern "C" long long mb_conv1d_range_events(int reset) {
  int dev = 0;
  MB_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || !g_range_word[dev]) return 0;  // the check never ran on this device
  unsigned v = 0;
  MB_HIP(hipDeviceSynchronize());
  MB_HIP(hipMemcpy(&v, g_range_word[dev], sizeof(unsigned), hipMemcpyDeviceToHost));
  if (reset) MB_HIP(hipMemset(g_range_word[dev], 0, sizeof(unsigned)));
  return (long long)v;
}

extern "C" int mb_conv1d(const mb_conv1d_args* a, mb_stream_t stream) {
  MB_REQUIRE(a && a->d_x && a->d_wpacked && a->d_y, "conv1d: null pointer");
  MB_REQUIRE(!a->transpose_out || a->up == 1, "conv1d: transpose_out needs up==1");
  ConvK k;
  int rc = conv_geometry(a, &k);
  if (rc) return rc;
  k.x = a->d_x; k.w = a->d_wpacked; k.bias = a->d_bias; k.res = a->d_res;
  k.post_scale = a->d_post_scale; k.post_shift = a->d_post_shift; k.y = a->d_y; k.gate = a->d_gate;
  MB_REQUIRE(a->out_act != 4 || (a->d_gate && a->d_res), "conv1d: highway epilogue needs d_gate and d_res");
  k.x_bstride = a->x_bstride; k.y_bstride = a->y_bstride; k.res_bstride = a->res_bstride;
  k.c_in = a->c_in; k.cin_pad = (a->c_in + 7) / 8 * 8; k.c_out = a->c_out;
  k.t_in = a->t_in; k.t_out = a->t_out;
  k.in_act = a->in_act; k.in_slope = a->in_slope;
  k.in_scale = a->in_scale == 0.f ? 1.f : a->in_scale;  // 0 (zero-initialised struct) means 1
  k.out_act = a->out_act; k.accumulate = a->accumulate;
  k.in_repeat = a->in_repeat > 1 ? a->in_repeat : 1;
  k.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
  k.out_slope = a->out_slope;
  k.valid = a->d_valid; k.valid_mul = a->valid_mul > 0 ? a->valid_mul : 1;
  k.range_events = range_word();
  MB_REQUIRE(k.in_repeat == 1 || a->t_in % k.in_repeat == 0, "conv1d: t_in %% in_repeat != 0");
  if (a->batch <= 0 || a->t_out <= 0) return MB_OK;

  const int n_mt = (a->c_out + 31) / 32;
  const int tq = cdiv(a->t_out, a->up);
  hipStream_t s = (hipStream_t)stream;
  // ---- split path (error-compensated fp16 MFMA), the default from 16 input channels up ----
  const char* senv = getenv("MBHIP_CONV_SPLIT");
  const bool split_off = senv && atoi(senv) == 0;
  const bool slope_ok = a->in_act != 1 || (a->in_slope >= 0.f && a->in_slope <= 1.f);  // the split kernel's max-form leaky_relu
  if (!split_off && a->c_in >= 16 && slope_ok) {
    const float* hdr = a->d_wpacked + conv_f32_image_floats(a->c_out, a->c_in, a->ksize);
    const uint4* wsplit = reinterpret_cast<const uint4*>(hdr + 64);
    // 128 output positions per workgroup; waves along the output channels when there are enough of them (A fragments are
    // then reused over 4 position tiles), else along time
    const int wm = n_mt >= 4 ? 4 : (n_mt >= 2 ? 2 : 1);
    // Short rows (the Tacotron encoder's 100-odd text positions, 32 utterances: 32 workgroups of 128 positions on 256 compute
    // units): 32 positions per workgroup instead -- the same sums in the same order, four times the workgroups.
    const int wgs128 = cdiv(tq, SNT) * cdiv(n_mt, wm) * a->batch * a->up;
    const bool narrow = wm == 4 && wgs128 < diag_int("conv_narrow_below", 256);
    const int snt = narrow ? 32 : SNT;
    dim3 grid(cdiv(tq, snt), cdiv(n_mt, wm), a->batch * a->up);
    const int rowlen = snt * k.down + k.span;
    const int l2 = diag_int("conv_lds2", -1);  // A/B: force one (0) or two (1) x-tile buffers
    int nbuf = l2 >= 0 ? (l2 == 1 && k.c_in > SCK ? 2 : 1) : (k.c_in >= 512 ? 2 : 1);
    if ((size_t)nbuf * rowlen * SROW > 64 * 1024) nbuf = 1;  // two buffers of a long window (rowlen > 227) would pass the 64 KB a
    const size_t lds = (size_t)nbuf * rowlen * SROW;          // launch gets without hipFuncSetAttribute; one always fits (<= 36 KB)
    const bool fits32 = (long long)a->c_in * a->t_in < (1ll << 31) && (long long)a->c_out * a->t_out < (1ll << 31);  // 32-bit offsets inside an item
    if (rowlen <= 256 && fits32) {  // (strided convs with long halos, giant items: the fp32-input kernel below)
      const bool pool = a->in_act == 2;
#define MB_SLAUNCH2(WM_, WN_, NTW_, TR_)                                                                                          \
  do {                                                                                                                            \
    if (pool) hipLaunchKernelGGL((conv1d_split_pool_kernel<WM_, WN_, NTW_, TR_>), grid, dim3(256), lds, s, k, wsplit, hdr, nbuf);       \
    else hipLaunchKernelGGL((conv1d_split_kernel<WM_, WN_, NTW_, TR_, false>), grid, dim3(256), lds, s, k, wsplit, hdr, nbuf);          \
  } while (0)
#define MB_SLAUNCH(WM_, WN_, NTW_)                                          \
  do {                                                                      \
    if (a->transpose_out) MB_SLAUNCH2(WM_, WN_, NTW_, true);                \
    else MB_SLAUNCH2(WM_, WN_, NTW_, false);                                \
  } while (0)
      if (narrow) MB_SLAUNCH(4, 1, 1);
      else if (wm == 4) MB_SLAUNCH(4, 1, 4);
      else if (wm == 2) MB_SLAUNCH(2, 2, 2);
      else MB_SLAUNCH(1, 4, 1);
#undef MB_SLAUNCH
#undef MB_SLAUNCH2
      MB_HIP(hipGetLastError());
      return MB_OK;
    }
  }
  // wave arrangement: few output channels -> all 4 waves along time.
  const int wm = (n_mt >= 4 && tq <= 64) ? 4 : (n_mt >= 2 ? 2 : 1);
  const int wn = 4 / wm;
  const int NT = 64 * wn;
  dim3 grid(cdiv(tq, NT), cdiv(n_mt, wm), a->batch * a->up);
  const size_t lds = (size_t)CK * (NT * k.down + k.span) * sizeof(float);
  MB_REQUIRE(lds <= 160 * 1024, "conv1d: halo too large for LDS (%zu B)", lds);
#define MB_LAUNCH(WM_, WN_)                                                                     \
  do {                                                                                          \
    if (a->transpose_out)                                                                       \
      hipLaunchKernelGGL((conv1d_mfma_kernel<WM_, WN_, true>), grid, dim3(256), lds, s, k);     \
    else                                                                                        \
      hipLaunchKernelGGL((conv1d_mfma_kernel<WM_, WN_, false>), grid, dim3(256), lds, s, k);    \
  } while (0)
  if (wm == 4) MB_LAUNCH(4, 1);
  else if (wm == 2) MB_LAUNCH(2, 2);
  else MB_LAUNCH(1, 4);
#undef MB_LAUNCH
  MB_HIP(hipGetLastError());
  return MB_OK;
}
