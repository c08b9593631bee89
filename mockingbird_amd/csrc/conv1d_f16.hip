// Conv1d / ConvTranspose1d on the gfx950 fp16 matrix cores (fp16 storage, fp32 accumulate):
// the throughput path of the GAN vocoders (BASELINE configs[4] "Fre-GAN vocoder fp16 with MFMA
// Conv1d"; HiFi-GAN shares it).  The fp32 kernel in conv1d.hip stays the 1e-4 RMS parity path.
//
//   y[b][t][co] = epilogue( sum_{j,ci} W[co][ci][j] * pre(x[b][t_in(t,j)][ci]) )
//
// Layout: activations are TIME-MAJOR [B][T][C] fp16.  The GEMM K dimension is (tap, channel)
// with the channel innermost, so the B fragment of v_mfma_f32_32x32x16_f16 (8 consecutive k per
// lane) is 8 consecutive channels of one time row = ONE 16-byte LDS read, and a (time tile x all
// channels) input window is one contiguous span of HBM.
//
// Work decomposition (256 threads = 4 waves, WM x WN):
//   * block tile  = (WM*MT*32 output channels) x (WN*128 output positions); each wave owns
//     MT x 4 MFMA 32x32 tiles (MT*4*16 fp32 accumulators per lane) so every B fragment read from
//     LDS feeds MT MFMAs and every A fragment feeds 4.
//   * x window (positions + halo) x `ck` channels staged through LDS with the input activation
//     (leaky-relu) applied on packed halves; rows padded by 8 halves -> row stride is an odd
//     multiple of 16 B -> the ds_read_b128 of the 16-lane groups is conflict free.
//   * weights: packed on the host in A-fragment order [phase][mtile][tap][cin/16][lane][8 halves];
//     a wave fetches one coalesced 1 KiB row per fragment straight into registers (they are
//     L2-resident across the grid), one k-step ahead of use.
//   * transposed conv (stride `up`) = `up` polyphase sub-convolutions (grid.z), as in conv1d.hip.
//   * epilogue: bias, activation, residual, scale, accumulate fused; the D fragment hands a lane 4
//     consecutive channels of one time row -> 8-byte packed stores.
#include "common.h"

namespace mb {

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

struct ConvHK {
  const h16* x; const h16* w; const float* bias; const h16* res; void* y;
  long long x_bstride, y_bstride, res_bstride;
  int c_in, n_cb, c_out, t_in, t_out;
  int ntaps, up, step, min_off, span;
  int off0[8];
  int in_act; float in_slope;
  int out_act; float out_scale;
  int accumulate, in_repeat, y_f32, ck;
  const int* valid; int valid_mul;  // ragged batches (mb_conv1d_f16_args.d_valid)
};

__device__ __forceinline__ h16x8 lrelu8(h16x8 v, h16 slope) {
  // slope in (0,1): leaky_relu(x) = max(x, slope*x)
  h16x8 s = v * slope;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = v[i] > s[i] ? v[i] : s[i];
  return v;
}

template <int MT, int WM, bool VEC>
__global__ __launch_bounds__(256, MT == 2 ? 2 : 3) void conv1d_f16_kernel(ConvHK a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  h16* lds = reinterpret_cast<h16*>(lds_raw);
  constexpr int WN = 4 / WM, NTW = 4, NB = WN * NTW * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> SGPR
  const int wm = wave / WN, wn = wave % WN;
  const int p = blockIdx.z % a.up, b = blockIdx.z / a.up;
  const int q0 = blockIdx.x * NB;
  const int Tq = (a.t_out - p + a.up - 1) / a.up;  // outputs of this phase
  if (q0 >= Tq) return;
  // ragged batch: this item's valid input / output extent (positions beyond are its zero padding / never consumed)
  const int t_lim = a.valid ? min(a.t_in, a.valid[b] * a.valid_mul * a.in_repeat) : a.t_in;
  if (a.valid) {
    const int t_out_b = a.up > 1 ? t_lim * a.up : t_lim + (a.t_out - a.t_in);
    if (q0 * a.up + p >= t_out_b) return;
  }
  const int n_mt = (a.c_out + 31) >> 5;
  const int mt0 = (blockIdx.y * WM + wm) * MT;
  const bool active = mt0 < n_mt;
  const int rowlen = NB + a.span;
  const int ckp = a.ck + 8;
  const int c_tot = a.n_cb * 16;
  const h16* xb = a.x + (long long)b * a.x_bstride;
  const int off_base = a.off0[p] - a.min_off;  // >= 0 for every tap
  const size_t frags_per_mt = (size_t)a.ntaps * a.n_cb;

  // A-fragment row pointers of this wave's MT output-channel tiles (clamped: tiles past c_out
  // compute on a valid tile's weights and are dropped in the epilogue)
  const h16x8* wp[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mt = min(mt0 + i, n_mt - 1);
    wp[i] = reinterpret_cast<const h16x8*>(a.w) + ((size_t)(p * n_mt + mt) * frags_per_mt) * 64 + lane;
  }

  f32x16 acc[MT][NTW];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

  h16x8 acur[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acur[i] = wp[i][0];  // (tap 0, channel block 0)

  const h16 slope = (h16)a.in_slope;
  for (int c0 = 0; c0 < c_tot; c0 += a.ck) {
    const int ck_cur = min(a.ck, c_tot - c0);
    const int ppr = ck_cur >> 3;  // 16-byte pieces per row
    __syncthreads();              // previous chunk fully consumed
    for (int idx = tid; idx < rowlen * ppr; idx += 256) {
      const int row = idx / ppr, pc = idx - row * ppr;
      const int ti = q0 + a.min_off + row;
      const int ci = c0 + pc * 8;
      h16x8 v;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (h16)0.f;
      if (ti >= 0 && ti < t_lim && ci < a.c_in) {
        const int ts = a.in_repeat > 1 ? ti / a.in_repeat : ti;
        v = *reinterpret_cast<const h16x8*>(xb + (long long)ts * a.c_in + ci);
        if (a.in_act == 1) v = lrelu8(v, slope);
      }
      *reinterpret_cast<h16x8*>(lds + row * ckp + pc * 8) = v;
    }
    __syncthreads();
    if (active) {
      const int ncb2 = ck_cur >> 4;
      const int cb0 = c0 >> 4;
      const bool more_chunks = c0 + a.ck < c_tot;
      const h16* lb = lds + (wn * (NTW * 32) + (lane & 31) + off_base) * ckp + (lane >> 5) * 8;
      for (int j = 0; j < a.ntaps; ++j) {
        const h16* lj = lb + j * a.step * ckp;
        for (int cb2 = 0; cb2 < ncb2; ++cb2) {
          // next A fragments: next channel block, else next tap, else first step of the next chunk
          size_t nf;
          if (cb2 + 1 < ncb2) nf = (size_t)j * a.n_cb + cb0 + cb2 + 1;
          else if (j + 1 < a.ntaps) nf = (size_t)(j + 1) * a.n_cb + cb0;
          else nf = more_chunks ? (size_t)(cb0 + ncb2) : (size_t)j * a.n_cb + cb0 + cb2;
          h16x8 anext[MT];
#pragma unroll
          for (int i = 0; i < MT; ++i) anext[i] = wp[i][nf * 64];
          // pin the weight prefetch ABOVE this step's LDS reads + MFMAs (hipcc otherwise sinks the
          // loads to their consumer and waits vmcnt(0) on the spot)
          __builtin_amdgcn_sched_barrier(0);
          h16x8 bf[NTW];
#pragma unroll
          for (int n = 0; n < NTW; ++n)
            bf[n] = *reinterpret_cast<const h16x8*>(lj + n * 32 * ckp + cb2 * 16);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int n = 0; n < NTW; ++n)
              acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(acur[i], bf[n], acc[i][n], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < MT; ++i) acur[i] = anext[i];
        }
      }
    }
  }
  if (!active) return;

  // ---- epilogue ----
  // D fragment (32x32): col = lane&31 (time), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel):
  // registers 4g..4g+3 are 4 consecutive channels co0 = 32*mt + 8g + 4*(lane>>5).
  // VEC: c_out % 4 == 0 and fp16 output -> 8-byte packed loads/stores; else per element.
  const h16* rb = a.res ? a.res + (long long)b * a.res_bstride : nullptr;
  h16* yh = reinterpret_cast<h16*>(a.y) + (long long)b * a.y_bstride;
  float* yf = reinterpret_cast<float*>(a.y) + (long long)b * a.y_bstride;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mt = mt0 + i;
    if (mt >= n_mt) break;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int q = q0 + wn * (NTW * 32) + n * 32 + (lane & 31);
      const long long t = (long long)q * a.up + p;
      if (q < Tq) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co0 = mt * 32 + 8 * g + 4 * (lane >> 5);
          if (co0 >= a.c_out) continue;
          const long long o = t * a.c_out + co0;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][n][4 * g + e];
          if (VEC) {
            if (a.bias) {
              const float4 bv = *reinterpret_cast<const float4*>(a.bias + co0);
              v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
            }
            if (a.out_act == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (a.out_act == 2) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
            }
            if (rb) {
              const h16x4 rv = *reinterpret_cast<const h16x4*>(rb + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= a.out_scale;
            if (a.accumulate) {
              const h16x4 ov = *reinterpret_cast<const h16x4*>(yh + o);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)ov[e];
            }
            h16x4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (h16)v[e];
            *reinterpret_cast<h16x4*>(yh + o) = hv;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (co0 + e < a.c_out) {
                float u = v[e];
                if (a.bias) u += a.bias[co0 + e];
                if (a.out_act == 1) u = fmaxf(u, 0.f);
                else if (a.out_act == 2) u = tanhf(u);
                if (rb) u += (float)rb[o + e];
                u *= a.out_scale;
                if (a.y_f32) {
                  if (a.accumulate) u += yf[o + e];
                  yf[o + e] = u;
                } else {
                  if (a.accumulate) u += (float)yh[o + e];
                  yh[o + e] = (h16)u;
                }
              }
            }
          }
        }
      }
      // keep the compiler from hoisting every tile's residual/bias loads above the first store
      asm volatile("" ::: "memory");
    }
  }
}

// ConvTranspose1d upsamplers (ksize = taps x stride) as ONE workgroup per (NQ input positions, batch item): the x window with
// ALL input channels is staged once (the general kernel above runs one workgroup per polyphase and re-stages the window per
// 32- / 64-channel chunk), the `up` polyphase outputs of the tile are computed back to back -- each of the 8 waves owns one
// 32 x 32 (output channels x positions) tile per phase -- and staged in LDS in OUTPUT order, so that the tile leaves as whole
// rows (the general kernel's 8-byte stores land `up` rows apart: partial lines, written by `up` different workgroups).
// A wave's A fragments (its output-channel tile, all phases: one contiguous walk through the packed image per phase) come
// through an EIGHT-deep register ring, refilled behind the MFMA that read the slot: with the one-step prefetch of the general
// kernel every k-step waited an L2 round trip for 32 cycles of MFMA work (a first version of this kernel: 18 us per tile).
__global__ __launch_bounds__(512) void convt_f16_kernel(ConvHK a, const int n_mt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NQ = 32 * (8 / n_mt);
  const int b = blockIdx.y, q0 = blockIdx.x * NQ;
  const int t_lim = a.valid ? min(a.t_in, a.valid[b] * a.valid_mul) : a.t_in;  // ragged batch: this item's input rows
  const int t_out_b = a.valid ? min(a.t_out, t_lim * a.up) : a.t_out;
  if (q0 * a.up >= t_out_b) return;
  const int cinp = a.c_in + 8, coutp = a.c_out + 8;
  const int rowlen = NQ + a.span;
  h16* xs = reinterpret_cast<h16*>(lds_raw);  // [rowlen][c_in + 8]
  h16* os = xs + rowlen * cinp;               // [NQ * up][c_out + 8]
  const int mt = wave % n_mt, nt = wave / n_mt;
  const int nst = a.ntaps * a.n_cb;           // k-steps per phase (a multiple of 8: checked on the host)
  const h16x8* wbase = reinterpret_cast<const h16x8*>(a.w) + lane;
  // the wave's fragments in consumption order: nst contiguous ones per phase, phases n_mt * nst apart; the walk stops at the
  // last fragment (re-reads, discarded)
  const h16x8* wnext = wbase + (size_t)mt * nst * 64;
  int pf_left = nst, pf_phases = a.up - 1;
  auto next_frag = [&]() {
    const h16x8 v = *wnext;
    if (--pf_left > 0) wnext += 64;
    else if (pf_phases > 0) { --pf_phases; pf_left = nst; wnext += (size_t)((n_mt - 1) * nst + 1) * 64; }
    else pf_left = 1;
    return v;
  };
  h16x8 ring[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) ring[r] = next_frag();  // in flight under the window fill
  const h16* xb = a.x + (long long)b * a.x_bstride;
  {  // the window, input activation applied
    const int ppr = a.c_in >> 3;
    const h16 slope = (h16)a.in_slope;
    for (int idx = tid; idx < rowlen * ppr; idx += 512) {
      const int row = idx / ppr, pc = idx - row * ppr;
      const int ti = q0 + a.min_off + row;
      h16x8 v = (h16x8)(h16)0.f;
      if (ti >= 0 && ti < t_lim) {
        v = *reinterpret_cast<const h16x8*>(xb + (long long)ti * a.c_in + pc * 8);
        if (a.in_act == 1) v = lrelu8(v, slope);
      }
      *reinterpret_cast<h16x8*>(xs + row * cinp + pc * 8) = v;
    }
  }
  __syncthreads();
  const int qrow = nt * 32 + (lane & 31);
  for (int p = 0; p < a.up; ++p) {
    const h16* lb = xs + (qrow + a.off0[p] - a.min_off) * cinp + (lane >> 5) * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int j = 0, cb = 0;
    h16x8 bf = *reinterpret_cast<const h16x8*>(lb);  // the B fragment runs one k-step ahead of its MFMA
    for (int st = 0; st < nst; st += 8) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (++cb == a.n_cb) { cb = 0; ++j; }
        const int jn = j < a.ntaps ? j : 0;  // (past the phase's last step: any valid row, discarded)
        const h16x8 bn = *reinterpret_cast<const h16x8*>(lb + jn * a.step * cinp + cb * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[r], bf, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ring[r] = next_frag();
        bf = bn;
      }
    }
    // D fragment: col = lane & 31 (position), rows 8 g + 4 (lane >> 5) + e (channel): + bias -> fp16 -> output order in LDS
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co0 = mt * 32 + 8 * g + 4 * (lane >> 5);
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + co0);
      h16x4 hv;
      hv[0] = (h16)(acc[4 * g] + bv.x); hv[1] = (h16)(acc[4 * g + 1] + bv.y);
      hv[2] = (h16)(acc[4 * g + 2] + bv.z); hv[3] = (h16)(acc[4 * g + 3] + bv.w);
      *reinterpret_cast<h16x4*>(os + (qrow * a.up + p) * coutp + co0) = hv;
    }
  }
  __syncthreads();
  {  // the tile's rows, whole 16-byte pieces, consecutive lanes -> consecutive bytes of y
    const int ppr = a.c_out >> 3;
    const int rows = min(NQ * a.up, t_out_b - q0 * a.up);
    h16* yb = reinterpret_cast<h16*>(a.y) + (long long)b * a.y_bstride + (long long)q0 * a.up * a.c_out;
    for (int idx = tid; idx < rows * ppr; idx += 512) {
      const int row = idx / ppr, pc = idx - row * ppr;
      *reinterpret_cast<h16x8*>(yb + (long long)idx * 8) = *reinterpret_cast<const h16x8*>(os + row * coutp + pc * 8);
    }
  }
}

// [B][C][T] fp32 (the reference's layout) -> [B][T][C] fp16, 32x32 tiles through LDS so both
// sides are coalesced.
__global__ __launch_bounds__(256) void cm_f32_to_tm_f16_kernel(const float* __restrict__ x,
                                                              h16* __restrict__ y, int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* xb = x + (long long)b * C * T;
  h16* yb = y + (long long)b * C * T;
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    tile[r][tx] = (c < C && t < T) ? xb[(long long)c * T + t] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    if (t < T && c < C) yb[(long long)t * C + c] = (h16)tile[tx][r];
  }
}

__global__ __launch_bounds__(256) void tm_f16_to_cm_f32_kernel(const h16* __restrict__ x,
                                                              float* __restrict__ y, int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const h16* xb = x + (long long)b * C * T;
  float* yb = y + (long long)b * C * T;
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    tile[r][tx] = (c < C && t < T) ? (float)xb[(long long)t * C + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    if (t < T && c < C) yb[(long long)c * T + t] = tile[tx][r];
  }
}

int conv_f16_geometry(const mb_conv1d_f16_args* a, ConvHK* k) {
  MB_REQUIRE(a->up >= 1 && a->up <= 8, "conv1d_f16: up=%d out of range", a->up);
  MB_REQUIRE(a->ksize >= 1 && a->c_in >= 1 && a->c_out >= 1, "conv1d_f16: bad shape");
  MB_REQUIRE(a->c_in % 8 == 0, "conv1d_f16: c_in=%d must be a multiple of 8 (16-byte rows)", a->c_in);
  k->up = a->up;
  if (a->up == 1) {
    k->ntaps = a->ksize;
    k->step = a->dilation;
    k->off0[0] = -a->pad;
    k->min_off = -a->pad;
    k->span = (a->ksize - 1) * a->dilation;
  } else {
    MB_REQUIRE(a->ksize % a->up == 0, "conv_transpose1d_f16: ksize %d not a multiple of stride %d",
               a->ksize, a->up);
    MB_REQUIRE(a->dilation == 1, "conv_transpose1d_f16: dilation unsupported");
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

extern "C" size_t mb_conv1d_f16_packed_halves(int c_out, int c_in, int ksize, int up) {
  (void)up;  // up phases x (ksize/up) taps == ksize fragments per (mtile, channel block)
  const int n_mt = (c_out + 31) / 32, n_cb = (c_in + 15) / 16;
  return (size_t)n_mt * n_cb * ksize * 512;
}

extern "C" int mb_conv1d_f16_pack(const float* h_w, int c_out, int c_in, int ksize, int up,
                                  int transposed, int pad, uint16_t* h_packed) {
  MB_REQUIRE(h_w && h_packed, "conv1d_f16_pack: null pointer");
  MB_REQUIRE(up >= 1 && (transposed || up == 1), "conv1d_f16_pack: up>1 needs transposed=1");
  MB_REQUIRE(ksize % up == 0, "conv1d_f16_pack: ksize %% up != 0");
  const int n_mt = (c_out + 31) / 32, n_cb = (c_in + 15) / 16, ntaps = ksize / up;
  h16* out = reinterpret_cast<h16*>(h_packed);
  size_t o = 0;
  for (int p = 0; p < up; ++p) {
    const int j0 = transposed ? (p + pad) % up : 0;
    for (int mt = 0; mt < n_mt; ++mt)
      for (int j = 0; j < ntaps; ++j) {
        const int jj = transposed ? j0 + j * up : j;
        for (int cb = 0; cb < n_cb; ++cb)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              // A fragment of v_mfma_f32_32x32x16_f16: lane l holds A[m = l&31][k = 8*(l>>5) + e]
              const int co = mt * 32 + (lane & 31);
              const int ci = cb * 16 + (lane >> 5) * 8 + e;
              float v = 0.f;
              if (co < c_out && ci < c_in)
                v = transposed ? h_w[((size_t)ci * c_out + co) * ksize + jj]
                               : h_w[((size_t)co * c_in + ci) * ksize + jj];
              out[o++] = (h16)v;
            }
      }
  }
  return MB_OK;
}

extern "C" int mb_conv1d_f16(const mb_conv1d_f16_args* a, mb_stream_t stream) {
  MB_REQUIRE(a && a->d_x && a->d_wpacked && a->d_y, "conv1d_f16: null pointer");
  ConvHK k;
  int rc = conv_f16_geometry(a, &k);
  if (rc) return rc;
  k.x = reinterpret_cast<const h16*>(a->d_x);
  k.w = reinterpret_cast<const h16*>(a->d_wpacked);
  k.bias = a->d_bias;
  k.res = reinterpret_cast<const h16*>(a->d_res);
  k.y = a->d_y;
  k.x_bstride = a->x_bstride; k.y_bstride = a->y_bstride; k.res_bstride = a->res_bstride;
  k.c_in = a->c_in; k.n_cb = (a->c_in + 15) / 16; k.c_out = a->c_out;
  k.t_in = a->t_in; k.t_out = a->t_out;
  k.valid = a->d_valid; k.valid_mul = a->valid_mul > 0 ? a->valid_mul : 1;
  k.in_act = a->in_act; k.in_slope = a->in_slope;
  MB_REQUIRE(a->in_act == 0 || (a->in_act == 1 && a->in_slope > 0.f && a->in_slope < 1.f),
             "conv1d_f16: in_act must be 0 or leaky_relu with 0 < slope < 1");
  MB_REQUIRE(a->out_act >= 0 && a->out_act <= 2, "conv1d_f16: out_act %d unsupported", a->out_act);
  k.out_act = a->out_act; k.accumulate = a->accumulate;
  k.in_repeat = a->in_repeat > 1 ? a->in_repeat : 1;
  k.out_scale = a->out_scale == 0.f ? 1.f : a->out_scale;
  k.y_f32 = a->y_f32;
  MB_REQUIRE(k.in_repeat == 1 || a->t_in % k.in_repeat == 0, "conv1d_f16: t_in %% in_repeat != 0");
  if (a->batch <= 0 || a->t_out <= 0) return MB_OK;

  const int n_mt = (a->c_out + 31) / 32;
  const int tq = cdiv(a->t_out, a->up);
  hipStream_t s = (hipStream_t)stream;
  // ---- upsamplers: one workgroup per tile of input positions, all polyphases, coalesced output (convt_f16_kernel) ----
  if (a->up > 1 && !getenv("MBHIP_CONVT_GENERAL") && (a->c_out == 32 || a->c_out == 64 || a->c_out == 128 || a->c_out == 256) &&
      a->c_in % 16 == 0 && (k.ntaps * k.n_cb) % 8 == 0 && !a->d_res && !a->accumulate && a->out_act == 0 && !a->y_f32 &&
      k.in_repeat == 1 && k.out_scale == 1.f && a->t_out == a->t_in * a->up) {
    const int NQ = 32 * (8 / n_mt);
    const size_t lds = ((size_t)(NQ + k.span) * (a->c_in + 8) + (size_t)NQ * a->up * (a->c_out + 8)) * sizeof(h16);
    if (lds <= 160 * 1024) {
      static bool attr_done = false;
      if (!attr_done) {
        MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_done = true;
      }
      hipLaunchKernelGGL(convt_f16_kernel, dim3(cdiv(a->t_in, NQ), a->batch), dim3(512), lds, s, k, n_mt);
      MB_HIP(hipGetLastError());
      return MB_OK;
    }
  }
  // wave arrangement: >= 4 channel tiles -> 2x2 waves of 2 tiles; else all waves along time.
  // Packed 8-byte epilogue needs c_out % 4 == 0 and fp16 output; anything else (conv_post with
  // c_out = 1, fp32 output) takes the per-element epilogue with one channel tile per wave.
  const bool vec = (a->c_out % 4 == 0) && !a->y_f32;
  const int wm = (vec && n_mt >= 4) ? 2 : 1;
  const int mtw = (vec && n_mt >= 2) ? 2 : 1;
  const int NB = (4 / wm) * 128;
  k.ck = std::min(k.n_cb * 16, wm == 2 ? 64 : 32);
  dim3 grid(cdiv(tq, NB), cdiv(n_mt, wm * mtw), a->batch * a->up);
  const size_t lds = (size_t)(NB + k.span) * (k.ck + 8) * sizeof(h16);
  MB_REQUIRE(lds <= 160 * 1024, "conv1d_f16: halo too large for LDS (%zu B)", lds);
  if (wm == 2) hipLaunchKernelGGL((conv1d_f16_kernel<2, 2, true>), grid, dim3(256), lds, s, k);
  else if (mtw == 2) hipLaunchKernelGGL((conv1d_f16_kernel<2, 1, true>), grid, dim3(256), lds, s, k);
  else if (vec) hipLaunchKernelGGL((conv1d_f16_kernel<1, 1, true>), grid, dim3(256), lds, s, k);
  else hipLaunchKernelGGL((conv1d_f16_kernel<1, 1, false>), grid, dim3(256), lds, s, k);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

extern "C" int mb_f32_to_f16_tm(const float* d_x, void* d_y, int batch, int channels, int t,
                                mb_stream_t stream) {
  MB_REQUIRE(d_x && d_y, "f32_to_f16_tm: null pointer");
  if (batch <= 0 || channels <= 0 || t <= 0) return MB_OK;
  dim3 grid(cdiv(t, 32), cdiv(channels, 32), batch);
  hipLaunchKernelGGL(cm_f32_to_tm_f16_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_x,
                     reinterpret_cast<h16*>(d_y), channels, t);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

extern "C" int mb_f16_tm_to_f32(const void* d_x, float* d_y, int batch, int channels, int t,
                                mb_stream_t stream) {
  MB_REQUIRE(d_x && d_y, "f16_tm_to_f32: null pointer");
  if (batch <= 0 || channels <= 0 || t <= 0) return MB_OK;
  dim3 grid(cdiv(t, 32), cdiv(channels, 32), batch);
  hipLaunchKernelGGL(tm_f16_to_cm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const h16*>(d_x), d_y, channels, t);
  MB_HIP(hipGetLastError());
  return MB_OK;
}
