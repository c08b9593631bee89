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
#ifndef MB_CONVT_FB
#define MB_CONVT_FB 6  // window pieces per thread in flight (convt_f16_kernel); 1 = diagnostics: one round trip per piece
#endif

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
// Also the form for SHORT plain convs (up = 1: conv_pre, 80 -> 512 channels over a few thousand frames: 35 dependent k-steps of
// latency in the general kernel): more than 8 channel tiles are dealt over blockIdx.z in groups of 8.
// The k-steps of a phase run in GROUPS of RD (= the ring depth, a template parameter): a group is RD channel blocks of one tap
// (TPG = 1, n_cb a multiple of RD) or TPG whole taps (n_cb * TPG = RD), so every address inside a group is the group's base plus
// a compile-time offset and the bookkeeping (next group's bases, with selects) happens once per group.  A version that walked
// (tap, channel block) per step spent ~40 scalar instructions per MFMA -- the scalar unit is shared by the CU's four SIMDs, the
// 8 waves queued on it and the upsamplers got slower.
// The B fragment of a step is read from LDS TWO steps ahead (b_cur / b_nxt / the new one): with one step the wait in front of
// an MFMA was for the read issued just before the previous one -- LDS latency on every step of the dependent MFMA chain.
// Every MFMA takes a fresh 1 KB A fragment from L2, so this shape is L2-bound at a quarter of the matrix rate: right for the
// upsamplers and conv_pre (a few GFLOP), wrong for anything large (the host keeps those on the general kernel).
template <int RD, int TPG>
__global__ __launch_bounds__(512) void convt_f16_kernel(ConvHK a, const int n_mt_tot, const int n_mt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int CBG = RD / TPG;               // channel blocks of one tap inside a group
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NQ = 32 * (8 / n_mt);
  const int b = blockIdx.y, q0 = blockIdx.x * NQ;
  const int t_lim = a.valid ? min(a.t_in, a.valid[b] * a.valid_mul) : a.t_in;  // ragged batch: this item's input rows
  const int t_out_b = a.valid ? min(a.t_out, t_lim * a.up) : a.t_out;
  if (q0 * a.up >= t_out_b) return;
  const int c_wg = min(n_mt * 32, a.c_out);   // output channels of this workgroup (16: half a tile, the other rows are padding)
  const int cinp = a.c_in + 8, coutp = c_wg + 8;
  const int rowlen = NQ + a.span;
  h16* xs = reinterpret_cast<h16*>(lds_raw);  // [rowlen][c_in + 8]
  h16* os = xs + rowlen * cinp;               // [NQ * up][c_wg + 8]
  const int mtl = wave % n_mt, nt = wave / n_mt;
  const int mt = blockIdx.z * 8 + mtl;
  const int nst = a.ntaps * a.n_cb;           // k-steps per phase, a multiple of RD
  const int ngroups = nst / RD;
  // the wave's A fragments: nst contiguous ones per phase (consumption order), phases n_mt_tot * nst apart
  const h16x8* wphase = reinterpret_cast<const h16x8*>(a.w) + (size_t)mt * nst * 64;
  const size_t wpstride = (size_t)n_mt_tot * nst * 64;
  h16x8 ring[RD];
#pragma unroll
  for (int r = 0; r < RD; ++r) {  // phase 0, group 0: in flight under the window fill
    ring[r] = wphase[r * 64 + lane];
    // in slot order: issued out of order, the youngest-but-one slot being the first one consumed, the compiler's wait in
    // front of that MFMA is vmcnt(1) in EVERY iteration (loop-entry and back-edge states are merged)
    __builtin_amdgcn_sched_barrier(0);
  }
  const h16* xb = a.x + (long long)b * a.x_bstride;
  {  // the window, input activation applied.  Six pieces per thread are requested before the first is used (clamped rows, the
     // zero padding is a select on the value): a load -> store loop made every workgroup wait five HBM round trips in a row
    const int ppr = a.c_in >> 3;
    const h16 slope = (h16)a.in_slope;
    const int total = rowlen * ppr;
    constexpr int FB = MB_CONVT_FB;
    for (int base = 0; base < total; base += 512 * FB) {
      h16x8 v[FB];
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        const int idx = min(base + i * 512 + tid, total - 1);
        const int row = idx / ppr, pc = idx - row * ppr;
        const int tc = min(max(q0 + a.min_off + row, 0), a.t_in - 1);
        v[i] = *reinterpret_cast<const h16x8*>(xb + (long long)tc * a.c_in + pc * 8);
      }
#pragma unroll
      for (int i = 0; i < FB; ++i) {
        const int idx = base + i * 512 + tid;
        const int row = idx / ppr, pc = idx - row * ppr;
        const int ti = q0 + a.min_off + row;
        h16x8 w = (ti >= 0 && ti < t_lim) ? v[i] : (h16x8)(h16)0.f;
        if (a.in_act == 1) w = lrelu8(w, slope);
        if (idx < total) *reinterpret_cast<h16x8*>(xs + row * cinp + pc * 8) = w;
      }
    }
  }
  __syncthreads();
  const int qrow = nt * 32 + (lane & 31);
  const int tapstride = a.step * cinp;        // halves between the rows of consecutive taps
  for (int p = 0; p < a.up; ++p) {
    const h16* lb = xs + (qrow + a.off0[p] - a.min_off) * cinp + (lane >> 5) * 8;
    // where the ring refills of this phase's LAST group come from: the next phase's first group (after the last phase: this
    // phase's own first group again, discarded)
    const h16x8* wnextphase = p + 1 < a.up ? wphase + wpstride : wphase;
    const h16x8* wref = wphase;               // + RD * 64 per group
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // B fragment of step r of the group at (tap jg, channel block cg): lb + jg * tapstride + cg * 16 + a compile-time part
    auto bfrag = [&](const h16* gb, const int r) {
      return *reinterpret_cast<const h16x8*>(gb + (r / CBG) * tapstride + (r % CBG) * 16);
    };
    int jg = 0, cg = 0;
    const h16* gb = lb;
    h16x8 b_cur = bfrag(gb, 0), b_nxt = bfrag(gb, 1 % RD);
    for (int g = 0; g < ngroups; ++g) {
      const bool lastg = g == ngroups - 1;
      wref = lastg ? wnextphase : wref + RD * 64;
      // next group's base (the last group re-reads its own first two fragments, discarded)
      if (TPG > 1) jg += lastg ? 0 : TPG;
      else {
        const bool wrap = cg + RD == a.n_cb;
        cg = lastg ? cg : (wrap ? 0 : cg + RD);
        jg += (wrap && !lastg) ? 1 : 0;
      }
      const h16* gbn = lb + jg * tapstride + cg * 16;
#pragma unroll
      for (int r = 0; r < RD; ++r) {
        const h16x8 b_new = r + 2 < RD ? bfrag(gb, r + 2) : bfrag(gbn, r + 2 - RD);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[r], b_cur, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        ring[r] = wref[r * 64 + lane];
        b_cur = b_nxt;
        b_nxt = b_new;
      }
      gb = gbn;
    }
    wphase = wnextphase;
    // D fragment: col = lane & 31 (position), rows 8 g + 4 (lane >> 5) + e (channel): + bias -> fp16 -> output order in LDS
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = mtl * 32 + 8 * g + 4 * (lane >> 5);
      if (cl >= c_wg) continue;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.bias) bv = *reinterpret_cast<const float4*>(a.bias + blockIdx.z * 256 + cl);
      h16x4 hv;
      hv[0] = (h16)(acc[4 * g] + bv.x); hv[1] = (h16)(acc[4 * g + 1] + bv.y);
      hv[2] = (h16)(acc[4 * g + 2] + bv.z); hv[3] = (h16)(acc[4 * g + 3] + bv.w);
      *reinterpret_cast<h16x4*>(os + (qrow * a.up + p) * coutp + cl) = hv;
    }
  }
  __syncthreads();
  {  // the tile's rows, whole 16-byte pieces, consecutive lanes -> consecutive bytes of y (c_wg channels of each row)
    const int ppr = c_wg >> 3;
    const int rows = min(NQ * a.up, t_out_b - q0 * a.up);
    h16* yb = reinterpret_cast<h16*>(a.y) + (long long)b * a.y_bstride + (long long)q0 * a.up * a.c_out + blockIdx.z * 256;
    for (int idx = tid; idx < rows * ppr; idx += 512) {
      const int row = idx / ppr, pc = idx - row * ppr;
      *reinterpret_cast<h16x8*>(yb + (long long)row * a.c_out + pc * 8) = *reinterpret_cast<const h16x8*>(os + row * coutp + pc * 8);
    }
  }
}

// One output channel (conv_post: 32 -> 1 channels, k = 7, tanh): a dot product per output position on the packed-fp16 dot unit
// (v_dot2_f32_f16, fp32 accumulate) instead of a 32-row MFMA tile with one live row.  256 positions per workgroup, the window
// staged once with the input activation, the 224 weights of the channel in registers (read out of the A-fragment image: row
// co = 0 lives in lanes 0 / 32 of every fragment).  Bound by the one read of x.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
static constexpr int C1_TPW = 4;  // 256-position tiles per workgroup
template <int CIN, int KS>
__global__ __launch_bounds__(256, 3) void conv_c1_f16_kernel(ConvHK a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int NB = 256, CP = CIN + 8, PPR = CIN / 8, TPW = C1_TPW;
  h16* xs = reinterpret_cast<h16*>(lds_raw);  // [NB + span][CIN + 8]
  const int tid = threadIdx.x, b = blockIdx.y, qbase = blockIdx.x * NB * TPW;
  const int t_lim = a.valid ? min(a.t_in, a.valid[b] * a.valid_mul) : a.t_in;
  const int q_end = a.valid ? min(a.t_out, t_lim + (a.t_out - a.t_in)) : a.t_out;  // tiles from here on are never consumed
  if (qbase >= q_end) return;
  h16x8 w[KS][PPR];
  // vector loads through an opaque zero lane offset: as provably uniform loads the 112 weight words went to SGPRs, 47 of them
  // spilled to VGPR lanes (a v_readlane in front of every other dot product)
  int vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  const h16x8* wp = reinterpret_cast<const h16x8*>(a.w) + vz;
#pragma unroll
  for (int j = 0; j < KS; ++j)
#pragma unroll
    for (int pc = 0; pc < PPR; ++pc) w[j][pc] = wp[(j * (CIN / 16) + (pc >> 1)) * 64 + (pc & 1) * 32];
  const h16* xb = a.x + (long long)b * a.x_bstride;
  const int rowlen = NB + a.span;
  const h16 slope = (h16)a.in_slope;
  const float bias = a.bias ? a.bias[0] : 0.f;
  // a tile's window: every 16-byte piece of a thread requested before the first is used, and the NEXT tile's pieces while this
  // one is computed (the first version -- one tile per workgroup, load -> store per piece -- was latency-bound at 2 TB/s);
  // loads from clamped rows, the zero padding is a select on the value
  constexpr int NIT = ((NB + 64) * PPR + 255) / 256;  // span <= 64 (host)
  h16x8 xv[NIT];
  auto issue = [&](const int q0) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it, row = idx / PPR, pc = idx % PPR;
      const int tc = min(max(q0 + a.min_off + row, 0), a.t_in - 1);
      xv[it] = *reinterpret_cast<const h16x8*>(xb + (long long)tc * CIN + pc * 8);
    }
  };
  issue(qbase);
  for (int tt = 0; tt < TPW; ++tt) {
    const int q0 = qbase + tt * NB;
    if (q0 >= q_end) break;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 256 * it, row = idx / PPR, pc = idx % PPR;
      const int ti = q0 + a.min_off + row;
      h16x8 v = (ti >= 0 && ti < t_lim) ? xv[it] : (h16x8)(h16)0.f;
      if (a.in_act == 1) v = lrelu8(v, slope);
      if (row < rowlen) *reinterpret_cast<h16x8*>(xs + row * CP + pc * 8) = v;
    }
    __syncthreads();
    if (tt + 1 < TPW && q0 + NB < q_end) issue(q0 + NB);
    const int q = q0 + tid;
    float acc = bias;
    const h16* lb = xs + tid * CP;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
#pragma unroll
      for (int pc = 0; pc < PPR; ++pc) {
        const h16x8 xr = *reinterpret_cast<const h16x8*>(lb + j * a.step * CP + pc * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const h16x2 x2 = {xr[2 * e], xr[2 * e + 1]}, w2 = {w[j][pc][2 * e], w[j][pc][2 * e + 1]};
          acc = __builtin_amdgcn_fdot2(x2, w2, acc, false);
        }
      }
    }
    if (a.out_act == 1) acc = fmaxf(acc, 0.f);
    else if (a.out_act == 2) acc = tanhf(acc);
    acc *= a.out_scale;
    if (q < a.t_out) {
      if (a.y_f32) reinterpret_cast<float*>(a.y)[(long long)b * a.y_bstride + q] = acc;
      else reinterpret_cast<h16*>(a.y)[(long long)b * a.y_bstride + q] = (h16)acc;
    }
    __syncthreads();  // every wave is done with the window
  }
}

// 1x1 conv over a nearest-repeated input (+ residual): Fre-GAN's res_output (generator.py:104-110, 145-159).  Bandwidth work
// with a few MFMAs in it: the general kernel staged chunks through LDS behind barriers and fetched its (4 KB of) weights one
// k-step ahead per tile -- 3 to 4 x the time of the bytes.  Here a wave keeps ALL A fragments in registers, takes its B
// fragments straight from HBM (lane = source position, 16 bytes = 8 channels; the next tile's are in flight during this
// one's products), computes each SOURCE position once, and writes the `in_repeat` output rows that share it -- residual added
// per row -- as whole rows from a wave-private LDS tile.  No barrier anywhere.
template <int NMT, int NCB>
__global__ __launch_bounds__(256) void conv_pw_f16_kernel(ConvHK a, const int tiles_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int CP = 32 * NMT + 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  h16* os = reinterpret_cast<h16*>(lds_raw) + wave * 32 * CP;  // [32 source positions][32 NMT + 8]
  const int b = blockIdx.y, u = a.in_repeat;
  const int t_lim = a.valid ? min(a.t_in, a.valid[b] * a.valid_mul * u) : a.t_in;  // output rows of this item
  const int src_lim = t_lim / u, src_all = a.t_in / u;
  const int tile0 = (blockIdx.x * 4 + wave) * tiles_per_wave;
  if (tile0 * 32 >= src_all) return;
  h16x8 af[NMT][NCB];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) af[mt][cb] = reinterpret_cast<const h16x8*>(a.w)[(mt * NCB + cb) * 64 + lane];
  float4 bv[NMT][4];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co0 = mt * 32 + 8 * g + 4 * (lane >> 5);
      bv[mt][g] = (a.bias && co0 < a.c_out) ? *reinterpret_cast<const float4*>(a.bias + co0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  const h16* xb = a.x + (long long)b * a.x_bstride + (lane >> 5) * 8;
  const h16* rb = a.res ? a.res + (long long)b * a.res_bstride : nullptr;
  h16* yb = reinterpret_cast<h16*>(a.y) + (long long)b * a.y_bstride;
  const h16 slope = (h16)a.in_slope;
  h16x8 bnext[NCB];
  auto issue = [&](const int tile) {
    const int q = min(tile * 32 + (lane & 31), src_all - 1);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) bnext[cb] = *reinterpret_cast<const h16x8*>(xb + (long long)q * a.c_in + cb * 16);
  };
  issue(tile0);
  const int ppr = a.c_out >> 3;            // 16-byte pieces per output row
  for (int tt = 0; tt < tiles_per_wave; ++tt) {
    const int tile = tile0 + tt, q0 = tile * 32;
    if (q0 >= src_all) break;
    h16x8 bf[NCB];
    const bool live = q0 + (lane & 31) < src_lim;  // beyond the item's length: its zero padding
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      bf[cb] = live ? bnext[cb] : (h16x8)(h16)0.f;
      if (a.in_act == 1) bf[cb] = lrelu8(bf[cb], slope);
    }
    if (tt + 1 < tiles_per_wave && q0 + 32 < src_all) issue(tile + 1);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][cb], bf[cb], acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h16x4 hv;
        hv[0] = (h16)(acc[4 * g] + bv[mt][g].x); hv[1] = (h16)(acc[4 * g + 1] + bv[mt][g].y);
        hv[2] = (h16)(acc[4 * g + 2] + bv[mt][g].z); hv[3] = (h16)(acc[4 * g + 3] + bv[mt][g].w);
        *reinterpret_cast<h16x4*>(os + (lane & 31) * CP + mt * 32 + 8 * g + 4 * (lane >> 5)) = hv;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the tile is this wave's own: LDS operations of a wave complete in order
    // the tile's 32 u output rows, contiguous in y: consecutive lanes -> consecutive 16-byte pieces
    const int rows = min(32 * u, a.t_out - q0 * u);
    const long long o0 = (long long)q0 * u * a.c_out;
    for (int idx = lane; idx < rows * ppr; idx += 64) {
      const int orow = idx / ppr, pc = idx - orow * ppr;
      h16x8 v = *reinterpret_cast<const h16x8*>(os + (orow / u) * CP + pc * 8);
      if (rb) v += *reinterpret_cast<const h16x8*>(rb + o0 + (long long)idx * 8);
      *reinterpret_cast<h16x8*>(yb + o0 + (long long)idx * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
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
  // ---- upsamplers and short plain convs: one workgroup per tile of input positions, all polyphases, eight-deep weight ring,
  //      coalesced output (convt_f16_kernel) ----
  const bool plain = !a->d_res && !a->accumulate && k.in_repeat == 1;
  const bool tiles8 = (a->c_out % 32 == 0 || a->c_out == 16) && (n_mt == 1 || n_mt == 2 || n_mt == 4 || n_mt % 8 == 0);
  // group shape: RD k-steps = RD channel blocks of one tap (n_cb % RD == 0) or TPG whole taps (n_cb * TPG == RD)
  const int nst = k.ntaps * k.n_cb;
  int rd = 0, tpg = 1;
  for (int d = 8; d >= 5 && !rd; --d)
    if (k.n_cb % d == 0) rd = d;
  if (!rd)
    for (int d = 8; d >= 4 && !rd; d -= 2)
      if (d % k.n_cb == 0 && k.ntaps % (d / k.n_cb) == 0) { rd = d; tpg = d / k.n_cb; }
  if (rd == 6 && tpg != 1 && tpg != 2) rd = 0;  // instances below
  if (rd == 4 && tpg != 1 && tpg != 2) rd = 0;
  // (the choice must not depend on the batch: an item of a ragged batch and its single run take the same kernel -- same sums)
  const bool small = a->up > 1 || (a->t_out <= 8192 && nst >= 8 && n_mt >= 8);
  if (!getenv("MBHIP_CONVT_GENERAL") && plain && tiles8 && small && rd && a->c_in % 16 == 0 && a->out_act == 0 && !a->y_f32 &&
      k.out_scale == 1.f && a->t_out == a->t_in * a->up) {
    const int n_wg = std::min(n_mt, 8);
    const int NQ = 32 * (8 / n_wg);
    const size_t lds = ((size_t)(NQ + k.span) * (a->c_in + 8) + (size_t)NQ * a->up * (std::min(n_wg * 32, a->c_out) + 8)) * sizeof(h16);
    if (lds <= 160 * 1024) {
      const dim3 grid(cdiv(a->t_in, NQ), a->batch, cdiv(n_mt, 8));
#define MB_CONVT(RD_, TPG_)                                                                                                  \
  if (rd == RD_ && tpg == TPG_) {                                                                                            \
    static bool attr_done = false;                                                                                           \
    if (!attr_done) {                                                                                                        \
      MB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_f16_kernel<RD_, TPG_>),                                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                  \
      attr_done = true;                                                                                                      \
    }                                                                                                                        \
    hipLaunchKernelGGL((convt_f16_kernel<RD_, TPG_>), grid, dim3(512), lds, s, k, n_mt, n_wg);                               \
  }
      MB_CONVT(8, 1) MB_CONVT(7, 1) MB_CONVT(6, 1) MB_CONVT(5, 1)
      MB_CONVT(8, 2) MB_CONVT(8, 4) MB_CONVT(8, 8) MB_CONVT(6, 2) MB_CONVT(4, 2) MB_CONVT(4, 1)
#undef MB_CONVT
      MB_HIP(hipGetLastError());
      return MB_OK;
    }
  }
  // ---- 1x1 conv over a nearest-repeated input, optional residual (Fre-GAN res_output): streaming kernel, no LDS staging of x ----
  if (!getenv("MBHIP_CONVT_GENERAL") && a->up == 1 && a->ksize == 1 && !a->accumulate && a->out_act == 0 && !a->y_f32 &&
      k.out_scale == 1.f && a->c_in % 16 == 0 && (a->c_out % 32 == 0 || a->c_out == 16) && a->t_out == a->t_in) {
    const int tpw = 4;  // 32-position tiles per wave
    const dim3 grid(cdiv(cdiv(a->t_in / k.in_repeat, 32), 4 * tpw), a->batch);
#define MB_PW(NMT_, NCB_)                                                                                                   \
  if (n_mt == NMT_ && k.n_cb == NCB_) {                                                                                     \
    hipLaunchKernelGGL((conv_pw_f16_kernel<NMT_, NCB_>), grid, dim3(256), 4 * 32 * (32 * NMT_ + 8) * sizeof(h16), s, k, tpw); \
    MB_HIP(hipGetLastError());                                                                                              \
    return MB_OK;                                                                                                           \
  }
    MB_PW(2, 8) MB_PW(1, 4) MB_PW(1, 2)
#undef MB_PW
  }
  // ---- one output channel (conv_post): dot products, no matrix tile with one live row ----
  if (!getenv("MBHIP_CONVT_GENERAL") && plain && a->up == 1 && a->c_out == 1 && (a->c_in == 32 || a->c_in == 16) && a->ksize == 7 &&
      a->t_out == a->t_in && k.span <= 64) {
    const size_t lds = (size_t)(256 + k.span) * (a->c_in + 8) * sizeof(h16);
    const dim3 grid(cdiv(a->t_out, 256 * C1_TPW), a->batch);
    if (a->c_in == 32) hipLaunchKernelGGL((conv_c1_f16_kernel<32, 7>), grid, dim3(256), lds, s, k);
    else hipLaunchKernelGGL((conv_c1_f16_kernel<16, 7>), grid, dim3(256), lds, s, k);
    MB_HIP(hipGetLastError());
    return MB_OK;
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
