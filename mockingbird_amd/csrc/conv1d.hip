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

extern "C" size_t mb_conv1d_packed_floats(int c_out, int c_in, int ksize, int up) {
  const int n_mt = (c_out + 31) / 32, n_cb = (c_in + 7) / 8;
  return (size_t)n_mt * n_cb * ksize * 256;  // up phases x (ksize/up) taps == ksize
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
  return MB_OK;
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
  MB_REQUIRE(k.in_repeat == 1 || a->t_in % k.in_repeat == 0, "conv1d: t_in %% in_repeat != 0");
  if (a->batch <= 0 || a->t_out <= 0) return MB_OK;

  const int n_mt = (a->c_out + 31) / 32;
  const int tq = cdiv(a->t_out, a->up);
  hipStream_t s = (hipStream_t)stream;
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
