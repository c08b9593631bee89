// Register-tiled form of the row-tile recurrent GEMM for WIDE batches (hundreds of columns: several
// utterances in one WaveRNN loop, mb_wavernn_generate_batch).  Same packed weights, same epilogues and --
// bit for bit -- the same sums as rnn_rowtile_body (rnn_body.h); only the walk over the work differs.
//
// Why (measured at 736 columns, tools/variant_bench.sh variants, profiles/r01_wavernn_batch_ts2.md): the first wide
// form (rnn_body.h TS: one 16x16 MFMA tile per wave, whole K) ran at 15-23 % of the fp32 MFMA peak.  Taking it
// apart: (1) 8 fragment loads feed only 32 MFMAs and nothing is reused from registers; (2) 768 workgroups on
// 256 CUs packed 2-3 per CU thrash each other's L1 (one workgroup per CU alone: -18 %); (3) the work came in
// 1.44 equal pieces per SIMD, so half the SIMDs ran two pieces and the rest idled: 70 % at best; (4) ~40
// scattered epilogue-operand loads per lane sat in the in-order vector-memory queue IN FRONT of the second
// k-step's fragments, stalling every wave for an HBM round trip before the loop got going.
// Here a wave owns MT x NT MFMA tiles (2 x 3 for the big launches: 128 row tiles x 46 column tiles = exactly
// one 6-tile piece for each of the 1024 SIMDs), every fragment feeds MT or NT tiles, a workgroup is 4 waves on
// 8 consecutive row tiles (its epilogue touches whole 128-byte lines of the gate-major tables), and the
// epilogue operands are requested behind the LAST fragment loads, two k-steps before they are needed.
//
// Bit-exactness with the K-split form (a column's result must not depend on how many columns share the
// launch): there, wave c accumulates the k-blocks kb = c (mod 8) in ascending order and the eight partial
// sums are added in wave order.  Here the eight chains are walked two at a time (chains 2p, 2p+1 use the
// adjacent k-blocks 8r+2p, 8r+2p+1: both halves of every 128-byte activation line are consumed together);
// when a chain pair is complete it is added to the running sum, pairs in ascending order -- the same
// additions in the same order, with 2 x 4 live accumulators per tile instead of 8 x 4.
#pragma once
#include "rnn_body.h"
// diagnostics only (tools/build_variant.sh): shifted operand strides -- wrong data, same amount of work
#ifndef MB_TS2_FAKEPAD_A
#define MB_TS2_FAKEPAD_A 0
#endif
#ifndef MB_TS2_FAKEPAD_B
#define MB_TS2_FAKEPAD_B 0
#endif

namespace mb {

constexpr int TS2_WAVES = 4;  // waves per workgroup, stacked along the rows
#ifndef MB_TS2_TAIL
#define MB_TS2_TAIL 2  // 4 (all four tail fragments in flight under the operand loads) measured 3 % slower
#endif

template <int EPI, unsigned F, int MT, int NT>
__device__ __forceinline__ void rnn_ts2_body(const RnnDev& d, const int bx, const int by) {
  static_assert(!(F & RF_GENERIC), "ts2: specialised instances only");
  static_assert(!(F & (RF_AFFINE | RF_MASK | RF_DROP | RF_SEQ | RF_SKIP | RF_PREIDX | RF_MULTISEG | RF_BIASH)),
                "ts2: feature not wired");
  static_assert(EPI != EPI_GRU || (F & RF_HPRE), "ts2: GRU instances take the hidden half precomputed");
  static_assert(EPI != EPI_LSTM, "ts2: no LSTM instance");
  constexpr int RL = (EPI == EPI_GRU) ? 3 : 4;
  constexpr int BLK = 4 * RL * 16;
  constexpr bool f_biasx = (F & RF_BIASX) != 0, f_pre = (F & RF_PRE) != 0, f_frame = (F & RF_FRAME) != 0;
  constexpr bool f_xres = (F & RF_XRES) != 0, f_xout = (F & RF_XOUT) != 0, f_gum = (F & RF_GUMBEL) != 0;
  constexpr bool f_zero = (F & RF_ZERO) != 0, f_hpre = (F & RF_HPRE) != 0, f_ftab = (F & RF_FOLDTAB) != 0;
  constexpr int act = (int)((F >> RF_ACT_SHIFT) & 3);
  constexpr int TAIL = MB_TS2_TAIL;  // k-steps (2 or 4) left to run when the epilogue operands are requested
  const RnnK& a = d.k;

  trace_begin(a.trace);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_mt_all = (EPI == EPI_LINEAR) ? (a.units + 15) / 16 : (a.units + 3) / 4;
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;  // dead 4th GRU row re-reads row 2
  const int edu = lane >> 4;  // epilogue unit (or row quad) within the tile
  const int H = a.units;

  int mt_raw[MT], mt[MT], en_raw[NT], en[NT];
  const float* wA[MT];
  const float* pB[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    mt_raw[m] = (bx * TS2_WAVES + wave) * MT + m;
    mt[m] = mt_raw[m] < n_mt_all ? mt_raw[m] : n_mt_all - 1;  // loads stay legal, nothing is stored
    wA[m] = a.w + (size_t)mt[m] * (a.nkb_total * BLK + MB_TS2_FAKEPAD_A) + ((u * RL + tau) * 4 + kq) * 4;
  }
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    en_raw[n] = (by * NT + n) * 16 + i;
    en[n] = en_raw[n] < a.N ? en_raw[n] : a.N - 1;  // duplicate a live column; its result is never stored
    pB[n] = d.segp[0] + (size_t)en[n] * (d.segld[0] + MB_TS2_FAKEPAD_B) + kq * 4;
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

  // ---- fragments of one step: chain pair (2p, 2p+1), round r -> k-blocks 8r+2p, 8r+2p+1 ----
  struct Frag { float4 a[MT][2]; float4 b[NT][2]; };  // [tile][chain of the pair]
  const int R = a.nkb_total >> 3;                      // rounds per chain (launches require nkb_total % 8 == 0)
#ifdef MB_TS2_DIAG_NOLOOP
  const int NIT = 0;
#else
  const int NIT = 4 * R;
#endif
  auto issue = [&](Frag& f, int it) {
    const int p = it / R, r = it - p * R;
    const int kb0 = 8 * r + 2 * p;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#ifdef MB_TS2_DIAG_NOLOAD_A
      if (it < 2)
#endif
#pragma unroll
      for (int m = 0; m < MT; ++m) f.a[m][c] = *reinterpret_cast<const float4*>(wA[m] + (size_t)(kb0 + c) * BLK);
#ifdef MB_TS2_DIAG_NOLOAD_B
      if (it < 2)
#endif
#pragma unroll
      for (int n = 0; n < NT; ++n) f.b[n][c] = *reinterpret_cast<const float4*>(pB[n] + (kb0 + c) * 16);
    }
  };
  f32x4 acc[2][MT][NT];  // [chain of the pair][row tile][column tile]
  f32x4 sum[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      sum[m][n] = {0.f, 0.f, 0.f, 0.f};
      acc[0][m][n] = {0.f, 0.f, 0.f, 0.f};
      acc[1][m][n] = {0.f, 0.f, 0.f, 0.f};
    }
  auto consume = [&](const Frag& f, int it) {
    // per (chain, tile): k-blocks ascending, components x,y,z,w -- the K-split wave's order; the 2*MT*NT
    // accumulators are independent, so consecutive MFMAs never wait on each other
#pragma unroll
    for (int cmp = 0; cmp < 4; ++cmp)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const float av = cmp == 0 ? f.a[m][c].x : cmp == 1 ? f.a[m][c].y : cmp == 2 ? f.a[m][c].z : f.a[m][c].w;
            const float bv = cmp == 0 ? f.b[n][c].x : cmp == 1 ? f.b[n][c].y : cmp == 2 ? f.b[n][c].z : f.b[n][c].w;
            acc[c][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c][m][n], 0, 0, 0);
          }
    const int p = it / R, r = it - p * R;
    if (r == R - 1) {  // chains 2p, 2p+1 complete: join the running sum in chain order (wave-uniform branch)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[m][n][e] += acc[c][m][n][e];
            acc[c][m][n] = {0.f, 0.f, 0.f, 0.f};
          }
    }
  };
  // Two fragment buffers, loads one step ahead of the MFMAs (three steps ahead measured no faster: the loop is
  // not latency-bound).  The scheduling barriers keep each step's loads in front of the MFMAs they overlap; left
  // alone, the scheduler sinks them to the end of the MFMA block.  No branches around loads: they make the
  // waitcnt pass assume the worst and drain the new loads too.
  Frag f0, f1, f2, f3;
  if (NIT) issue(f0, 0);
  for (int it = 0; it + TAIL < NIT; it += 2) {  // NIT = 4R; the last TAIL steps follow the operand loads
    issue(f1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
    consume(f0, it);
    __builtin_amdgcn_sched_barrier(0);
    issue(f0, it + 2);
    __builtin_amdgcn_sched_barrier(0);
    consume(f1, it + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // the fragments of the last TAIL steps are all requested here, so that the operand loads can queue behind them
  if (NIT) {
    issue(f1, NIT - TAIL + 1);
    if (TAIL == 4) { issue(f2, NIT - 2); issue(f3, NIT - 1); }
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
#ifndef MB_TS2_DIAG_NOEPI
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    ej[m] = mt[m] * 4 + edu;
    if (ej[m] >= H) ej[m] = H - 1;
    const int erow = mt[m] * 16 + edu * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      l_bx[m][r] = 0.f;
      if (EPI == EPI_LINEAR && f_biasx) l_bx[m][r] = a.biasX[erow + r < H ? erow + r : H - 1];
      if (EPI != EPI_LINEAR && f_biasx && r < RL) l_bx[m][r] = a.biasX[r * H + ej[m]];
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const float* prp = a.pre_table + (size_t)prow[n] * a.pre_stride;
      const size_t so = (size_t)en[n] * H + ej[m];
      l_hp[m][n] = 0.f; l_xr[m][n] = 0.f;
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
#endif
  __builtin_amdgcn_sched_barrier(0);
  if (NIT) {
    consume(f0, NIT - TAIL);
    consume(f1, NIT - TAIL + 1);
    if (TAIL == 4) { consume(f2, NIT - 2); consume(f3, NIT - 1); }
  }

#ifdef MB_TS2_DIAG_NOEPI
  {
    float t = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) t += sum[m][n][0] + sum[m][n][1] + sum[m][n][2] + sum[m][n][3];
    if (t == 12345.678f) a.h_out[0] = t;
    return;
  }
#endif
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mtt * 16 + du * 4 + r;
          if (row < a.units) {
            float v = sx[r] + (l_bx[m][r] + l_pre[m][nn][r]);
            if (act == 1) v = fmaxf(v, 0.f);
            else if (act == 2) v = sigmoidf_(v);
            else if (act == 3) v = tanhf(v);
            if (a.y) a.y[(size_t)n * a.ldy + row] = v;
            if (f_gum) {
              const float g = v - logf(-logf(u32_to_unit(gr[r])));
              if (g > best) { best = g; bcls = row; }  // ascending rows: first maximum kept
            }
          }
        }
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
        const float rg = sigmoidf_((sx[0] + (l_bx[m][0] + l_pre[m][nn][0])) + (sh[0] + zero));
        const float zg = sigmoidf_((sx[1] + (l_bx[m][1] + l_pre[m][nn][1])) + (sh[1] + zero));
        const float ng = tanhf((sx[2] + (l_bx[m][2] + l_pre[m][nn][2])) + rg * (sh[2] + zero));
        const float hy = ng + zg * (l_hp[m][nn] - ng);
        a.h_out[so] = hy;
        if (f_xout) a.x_out[so] = e_xr + hy;
      }
    }
  }
  trace_end(a.trace);
}

}  // namespace mb
