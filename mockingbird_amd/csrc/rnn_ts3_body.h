// Round 4: the wide-batch recurrent GEMM of rnn_ts2_body.h (mb_wavernn_generate_batch: hundreds of fold columns, north_star's
// "batch-32 synthetic input") off the fp32 matrix pipe.
//
// rnn_ts2_body runs v_mfma_f32_16x16x4_f32 -- 1/16 of the fp16 rate -- and stood at 33 % of that pipe's 157 TFLOP/s for three
// rounds (22.4-24.9 us per 128 x 46-tile launch, 12.1 us of it the MFMA train itself).  Here the same wave tiles (MT row tiles x NT
// column tiles per wave, 4 waves = 8 consecutive row tiles per workgroup, the same D fragments, hence the SAME epilogue code) are
// computed as error-compensated fp16 products (conv1d.hip's scheme, as in wavernn_pipe16.h):
//   w 2^s = wh + wl  pre-split on the host into the A fragments of v_mfma_f32_16x16x32_f16 ([tile][k-step of 32][hi | lo][lane][8]: the
//                    images wavernn_pipe16.h uses), one coalesced 1 KB row per fragment, requested a stage ahead;
//   x = xh + xl      split ONCE per workgroup: the 4 waves share their column tiles, so the fp32 activations of a 64-wide K stage
//                    (NT x 16 columns x 64 k) are loaded cooperatively, split into fp16 hi / lo rows in LDS (144-byte rows: conflict-free
//                    16-byte fragment reads), double-buffered, one barrier per stage.  Fetched per wave straight from memory the B
//                    fragments would be 142 B/clk per compute unit -- the fp16 pipe is 5.3x faster, the L1 is not;
//   acc += wl.xh + wh.xh, acl += wh.xl   in fp32, y = (acc + 2^-11 acl) 2^-s: 18 MFMAs of 16 cycles per k-step and wave (2 x 3 tiles)
//                    against 48 of 32.  Round 5: the residual is stored SCALED, xl = fp16((x - xh) 2^11) (conv1d.hip's fix: unscaled it
//                    is an fp16 subnormal below |x| = 0.125 -- 14 bits at |x| = 1e-3, which relu(fc1 ..) of a real checkpoint may be), at
//                    the price of a second accumulator set; the split itself is two packed conversions per pair.
// Range: |x| <= 65504.  A staged value beyond it (or a NaN) raises RnnK::range_word; mb_wavernn_generate_batch then discards the loop's
// samples and reruns it on rnn_ts2_body (fp32 MFMA) -- no silent clamp.
// Results are fp32-grade (21-22 bits per operand), NOT bit-identical to the fp32 forms any more: a column's sums still do not
// depend on the batch it is in (same order for every column), and the batch loop is held to the oracle with the exported noise
// (tests/test_wavernn_gpu.py::test_production_batch*).  MBHIP_RNN_WIDE=ts2 selects rnn_ts2_body.
#pragma once
#include "rnn_ts2_body.h"

namespace mb {

typedef _Float16 t3h;
typedef _Float16 t3h8 __attribute__((ext_vector_type(8)));
typedef _Float16 t3h4 __attribute__((ext_vector_type(4)));
typedef _Float16 t3h2 __attribute__((ext_vector_type(2)));
typedef float t3f2 __attribute__((ext_vector_type(2)));
constexpr int TS3_KS = 64;          // K per stage (two k-steps of 32)
constexpr int TS3_ROW = TS3_KS + 8;  // halves per staged column row (144 bytes)
template <int NT> constexpr size_t ts3_lds_bytes() { return (size_t)2 * 2 * NT * 16 * TS3_ROW * sizeof(t3h); }  // [buffer][hi | lo][column][k]

template <int EPI, unsigned F, int MT, int NT>
__device__ __forceinline__ void rnn_ts3_body(const RnnDev& d, const int bx, const int by, t3h* lds) {
  static_assert(!(F & RF_GENERIC), "ts3: specialised instances only");
  static_assert(!(F & (RF_AFFINE | RF_MASK | RF_DROP | RF_SEQ | RF_SKIP | RF_PREIDX | RF_MULTISEG | RF_BIASH)),
                "ts3: feature not wired");
  static_assert(EPI != EPI_GRU || (F & RF_HPRE), "ts3: GRU instances take the hidden half precomputed");
  static_assert(EPI != EPI_LSTM, "ts3: no LSTM instance");
  constexpr int RL = (EPI == EPI_GRU) ? 3 : 4;
  constexpr bool f_biasx = (F & RF_BIASX) != 0, f_pre = (F & RF_PRE) != 0, f_frame = (F & RF_FRAME) != 0;
  constexpr bool f_xres = (F & RF_XRES) != 0, f_xout = (F & RF_XOUT) != 0, f_gum = (F & RF_GUMBEL) != 0;
  constexpr bool f_zero = (F & RF_ZERO) != 0, f_hpre = (F & RF_HPRE) != 0, f_ftab = (F & RF_FOLDTAB) != 0;
  constexpr int act = (int)((F >> RF_ACT_SHIFT) & 3);
  const RnnK& a = d.k;

  trace_begin(a.trace);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_mt_all = (EPI == EPI_LINEAR) ? (a.units + 15) / 16 : (a.units + 3) / 4;
  const int i = lane & 15, kb = lane >> 4;
  const int edu = lane >> 4;  // epilogue unit (or row quad) within the tile
  const int H = a.units;
  const int nks = a.nkb_total >> 1;   // k-steps of 32
  const int NST = (a.dbg & 1) ? 2 : a.nkb_total >> 2;   // stages of 64 (launches require nkb_total % 8 == 0); diagnostics: two stages only

  int mt_raw[MT], mt[MT], en_raw[NT], en[NT];
  const uint4* wA[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    mt_raw[m] = (bx * TS2_WAVES + wave) * MT + m;
    mt[m] = mt_raw[m] < n_mt_all ? mt_raw[m] : n_mt_all - 1;  // loads stay legal, nothing is stored
    wA[m] = reinterpret_cast<const uint4*>(a.w16) + ((size_t)mt[m] * nks) * 2 * 64 + lane;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    en_raw[n] = (by * NT + n) * 16 + i;
    en[n] = en_raw[n] < a.N ? en_raw[n] : a.N - 1;  // duplicate a live column; its result is never stored
  }
  // staging item of this thread: column c16 of every column tile, float4 k4 of the 64-wide stage
  const int c16 = tid >> 4, k4 = tid & 15;
  const float* pS[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = (by * NT + n) * 16 + c16;
    pS[n] = d.segp[0] + (size_t)(col < a.N ? col : a.N - 1) * d.segld[0] + k4 * 4;
  }

  int fr_s = 0;
  if (f_frame) fr_s = *a.fr_base + a.fr_off;
  // per-fold descriptors (stable data, a few hundred cycles): requested first, used by the late operand loads
  int4 dsc0[NT], dsc1[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    dsc0[n] = make_int4(0, 0, 0, 0); dsc1[n] = make_int4(0, 0, 0, 0);
    if (f_frame && f_ftab) {
      dsc0[n] = *reinterpret_cast<const int4*>(a.fr_desc + (size_t)(a.fr_n_off + en[n]) * 8);
      dsc1[n] = *reinterpret_cast<const int4*>(a.fr_desc + (size_t)(a.fr_n_off + en[n]) * 8 + 4);
    }
  }

  struct FragA { uint4 v[2][MT][2]; };  // [k-step of the stage][row tile][hi | lo]
  auto issueA = [&](FragA& f, const int st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int part = 0; part < 2; ++part) f.v[ks][m][part] = wA[m][((size_t)(st * 2 + ks) * 2 + part) * 64];
  };
  float4 xs[NT];
  float rmax = 0.f;  // largest |a| + |b| this thread staged (NaN sticks)
  auto issueB = [&](const int st) {
#pragma unroll
    for (int n = 0; n < NT; ++n) xs[n] = *reinterpret_cast<const float4*>(pS[n] + st * TS3_KS);
  };
  auto storeB = [&](const int buf) {  // fp32 -> fp16 hi / lo rows of LDS buffer `buf`
    t3h* hb = lds + (size_t)buf * 2 * NT * 16 * TS3_ROW;
    t3h* lb = hb + (size_t)NT * 16 * TS3_ROW;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const t3f2 va = {xs[n].x, xs[n].y}, vb = {xs[n].z, xs[n].w};
      const t3h2 ha = __builtin_convertvector(va, t3h2), hb2 = __builtin_convertvector(vb, t3h2);
      const t3h2 la = __builtin_convertvector((va - __builtin_convertvector(ha, t3f2)) * 2048.f, t3h2);   // exact differences, scaled residual
      const t3h2 lb2 = __builtin_convertvector((vb - __builtin_convertvector(hb2, t3f2)) * 2048.f, t3h2);
      // range: a running NaN-propagating maximum of |a| + |b| (v_maximum3_f32: three instructions per four values, no branch -- a
      // compare-and-branch per float4 cost the 736-column step 4 us of 63), looked at once behind the k loop
      rmax = __builtin_elementwise_maximum(__builtin_elementwise_maximum(rmax, __builtin_fabsf(va[0]) + __builtin_fabsf(va[1])),
                                           __builtin_fabsf(vb[0]) + __builtin_fabsf(vb[1]));
      const t3h4 h = {ha[0], ha[1], hb2[0], hb2[1]}, l = {la[0], la[1], lb2[0], lb2[1]};
      const int o = (n * 16 + c16) * TS3_ROW + k4 * 4;
      *reinterpret_cast<t3h4*>(hb + o) = h;
      *reinterpret_cast<t3h4*>(lb + o) = l;
    }
  };
  f32x4 sum[MT][NT], sul[MT][NT];  // wl.xh + wh.xh | wh.xl (scaled residual)
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) { sum[m][n] = {0.f, 0.f, 0.f, 0.f}; sul[m][n] = {0.f, 0.f, 0.f, 0.f}; }
  auto compute = [&](const FragA& f, const int buf) {
    const t3h* hb = lds + (size_t)buf * 2 * NT * 16 * TS3_ROW;
    const t3h* lb = hb + (size_t)NT * 16 * TS3_ROW;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      t3h8 bh[NT], bl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int o = (n * 16 + i) * TS3_ROW + ks * 32 + kb * 8;
        bh[n] = *reinterpret_cast<const t3h8*>(hb + o);
        bl[n] = *reinterpret_cast<const t3h8*>(lb + o);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const t3h8 ah = __builtin_bit_cast(t3h8, f.v[ks][m][0]), al = __builtin_bit_cast(t3h8, f.v[ks][m][1]);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          sum[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[n], sum[m][n], 0, 0, 0);
          sul[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[n], sul[m][n], 0, 0, 0);
          sum[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[n], sum[m][n], 0, 0, 0);
        }
      }
    }
  };

  // stage pipeline: the NEXT stage's weight fragments and activation rows are in flight (registers) while this one is computed
  // from LDS; then the rows are split and stored into the other buffer; one barrier per stage
  FragA fa, fb;
  issueB(0);
  issueA(fa, 0);
  storeB(0);
  __syncthreads();
  for (int st = 0; st + 1 < NST; st += 2) {  // NST is even (nkb_total % 8 == 0)
    issueB(st + 1);
    issueA(fb, st + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(fa, 0);
    storeB(1);
    __syncthreads();
    if (st + 2 < NST) { issueB(st + 2); issueA(fa, st + 2); }
    __builtin_amdgcn_sched_barrier(0);
    if (st + 2 >= NST) break;  // the last stage (buffer 1) follows the operand loads below
    compute(fb, 1);
    storeB(0);
    __syncthreads();
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue operands of the MT x NT tiles: requested behind the LAST fragment loads (vector memory
  //      returns in order: anything queued earlier would stall the k-loop for an HBM round trip), two
  //      k-steps of MFMA work before they are needed ----
  unsigned posE[NT];
  int prow[NT], dsc_fold[NT];
  unsigned dsc_slo[NT], dsc_shi[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    posE[n] = 0u; dsc_fold[n] = 0; dsc_slo[n] = 0u; dsc_shi[n] = 0u;
    prow[n] = a.pre_base_row + en[n] * a.pre_n_stride;
    if (f_frame && f_ftab) {  // several utterances: per-fold descriptor (rnn.h RnnK::fr_desc)
      const int4 d0 = dsc0[n], d1 = dsc1[n];
      posE[n] = (unsigned)(d0.x + fr_s);
      prow[n] = d0.w + (posE[n] < (unsigned)d0.y ? (int)(posE[n] / (unsigned)a.fr_hop) : d1.x);
      dsc_fold[n] = d1.y; dsc_slo[n] = (unsigned)d1.z; dsc_shi[n] = (unsigned)d1.w;
    } else if (f_frame) {
      posE[n] = (unsigned)(a.fr_n_off + en[n]) * (unsigned)a.fr_fold_stride + (unsigned)fr_s;
      prow[n] = posE[n] < (unsigned)a.fr_total_len ? (int)(posE[n] / (unsigned)a.fr_hop) : a.fr_frames;
    }
  }
  int ej[MT];
  float l_bx[MT][4], l_pre[MT][NT][4], l_hs[MT][NT][4], l_hp[MT][NT], l_xr[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    ej[m] = mt[m] * 4 + edu;
    if (ej[m] >= H) ej[m] = H - 1;
    const int erow = mt[m] * 16 + edu * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      l_bx[m][r] = 0.f;
      if (EPI == EPI_LINEAR && f_biasx) l_bx[m][r] = a.biasX[erow + r < H ? erow + r : H - 1];  // (stable data: the compiler merges the four)
      if (EPI != EPI_LINEAR && f_biasx && r < RL) l_bx[m][r] = a.biasX[r * H + ej[m]];
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const float* prp = a.pre_table + (size_t)prow[n] * a.pre_stride;
      const size_t so = (size_t)en[n] * H + ej[m];
      l_hp[m][n] = 0.f; l_xr[m][n] = 0.f;
      if (EPI == EPI_LINEAR && f_pre && (H & 15) == 0) {  // the lane's four rows are one aligned 16-byte piece of the table row
        const float4 p4 = *reinterpret_cast<const float4*>(prp + erow);
        l_pre[m][n][0] = p4.x; l_pre[m][n][1] = p4.y; l_pre[m][n][2] = p4.z; l_pre[m][n][3] = p4.w;
        l_hs[m][n][0] = l_hs[m][n][1] = l_hs[m][n][2] = l_hs[m][n][3] = 0.f;
      } else
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        l_pre[m][n][r] = 0.f; l_hs[m][n][r] = 0.f;
        if (EPI == EPI_LINEAR) {
          if (f_pre) l_pre[m][n][r] = prp[erow + r < H ? erow + r : H - 1];
        } else if (r < RL) {
          if (f_pre) l_pre[m][n][r] = prp[r * H + ej[m]];
          if (f_hpre) l_hs[m][n][r] = a.h_pre[(size_t)en[n] * (RL * H) + r * H + ej[m]];
        }
      }
      if (EPI == EPI_GRU) l_hp[m][n] = a.h_prev[so];
      if (EPI != EPI_LINEAR && f_xres) l_xr[m][n] = a.x_res[so];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  compute(fb, 1);  // the last stage, in the shadow of the operand loads
  if (a.range_word && !(rmax <= 65504.f)) __hip_atomic_store(a.range_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    const float us = a.w16_unscale;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[m][n][e] = (sum[m][n][e] + sul[m][n][e] * 4.8828125e-4f) * us;
  }

  if (a.dbg & 2) {  // diagnostics: no epilogue (the sums still have to be computed)
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) t += sum[m][n][0] + sum[m][n][1] + sum[m][n][2] + sum[m][n][3];
    if (t == 12345.678f) a.h_out[0] = t;
    return;
  }
  // ---- the MT x NT epilogues: rnn_rowtile_body's, per tile ----
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (mt_raw[m] >= n_mt_all) continue;
#pragma unroll
    for (int nn = 0; nn < NT; ++nn) {
      const int n = en_raw[nn], du = edu, mtt = mt[m];
      float sx[4], sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) sx[e] = sum[m][nn][e];
      if (EPI == EPI_GRU && f_hpre) {
#pragma unroll
        for (int g = 0; g < RL; ++g) sh[g] += l_hs[m][nn][g];
      }
      if (f_zero && mtt == 0 && du == 0 && n < a.N) a.zero_slot[n] = 0ull;
      if (n >= a.N) continue;
      if (EPI == EPI_LINEAR) {
        float best = -INFINITY;
        int bcls = 0;
        uint32_t gr[4] = {0u, 0u, 0u, 0u};
        if (f_gum && f_ftab) philox4x32((uint32_t)fr_s, (uint32_t)dsc_fold[nn], (uint32_t)((mtt * 16 + du * 4) >> 2), 0x57415645u,
                                        dsc_slo[nn], dsc_shi[nn], gr);
        else if (f_gum) philox4x32((uint32_t)fr_s, (uint32_t)(a.fr_n_off + n), (uint32_t)((mtt * 16 + du * 4) >> 2), 0x57415645u,
                                   (uint32_t)a.gum_seed, (uint32_t)(a.gum_seed >> 32), gr);
        float yv[4] = {0.f, 0.f, 0.f, 0.f};
        const bool yvec = a.y && (a.units & 15) == 0 && (a.ldy & 3) == 0;  // four consecutive rows of a column: one 16-byte store
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mtt * 16 + du * 4 + r;
          if (row < a.units) {
            float v = sx[r] + (l_bx[m][r] + l_pre[m][nn][r]);
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = sigmoidf_(v);
            else if (act == 3) v = tanhf(v);
            yv[r] = v;
            if (a.y && !yvec) a.y[(size_t)n * a.ldy + row] = v;
            if (f_gum) {
              const float g = v - logf(-logf(u32_to_unit(gr[r])));
              if (g > best) { best = g; bcls = row; }  // ascending rows: first maximum kept
            }
          }
        }
        if (yvec) *reinterpret_cast<float4*>(a.y + (size_t)n * a.ldy + mtt * 16 + du * 4) = make_float4(yv[0], yv[1], yv[2], yv[3]);
        if (f_gum) {  // the 4 row quads of this column sit in lanes l, l+16, l+32, l+48
          unsigned long long pk = pack_argmax(best, bcls);
          const unsigned long long o1 = __shfl_xor(pk, 16, 64);
          pk = o1 > pk ? o1 : pk;
          const unsigned long long o2 = __shfl_xor(pk, 32, 64);
          pk = o2 > pk ? o2 : pk;
          if (du == 0) atomicMax(a.gum_slot + n, pk);
        }
        continue;
      }
      const int j = mtt * 4 + du;  // hidden unit
      if (j >= a.units) continue;
      const size_t so = (size_t)n * H + j;
      const float e_xr = l_xr[m][nn];
      if (EPI == EPI_GRU) {  // torch GRUCell, as rnn_rowtile_body (absent biases are the same literal zeros there)
        const float zero = 0.f;
        // gates on the hardware exp2 / reciprocal (1 ulp): this form has no bit-identical partner to keep
        const float rg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * ((sx[0] + (l_bx[m][0] + l_pre[m][nn][0])) + (sh[0] + zero))));
        const float zg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * ((sx[1] + (l_bx[m][1] + l_pre[m][nn][1])) + (sh[1] + zero))));
        const float ng = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * ((sx[2] + (l_bx[m][2] + l_pre[m][nn][2])) + rg * (sh[2] + zero))));
        const float hy = ng + zg * (l_hp[m][nn] - ng);
        a.h_out[so] = hy;
        if (f_xout) a.x_out[so] = e_xr + hy;
      }
    }
  }
  trace_end(a.trace);
}

}  // namespace mb
