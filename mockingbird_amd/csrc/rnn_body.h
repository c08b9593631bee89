// Device body of the row-tile recurrent GEMM (rnn.h) as a header, so that other translation units can
// run it as one JOB of a multi-job launch (tacotron.hip: LSTM hidden halves beside the attention kernel).
#pragma once
#include "rnn.h"

namespace mb {

// Feature bits of a specialised instance.  RF_GENERIC = decide everything at run time (fallback).
enum : unsigned {
  RF_BIASX = 1u << 0, RF_BIASH = 1u << 1, RF_PRE = 1u << 2, RF_PREIDX = 1u << 3, RF_FRAME = 1u << 4,
  RF_XRES = 1u << 5, RF_SKIP = 1u << 6, RF_MASK = 1u << 7, RF_DROP = 1u << 8, RF_SEQ = 1u << 9,
  RF_XOUT = 1u << 10, RF_AFFINE = 1u << 11, RF_GUMBEL = 1u << 12, RF_ZERO = 1u << 13, RF_MULTISEG = 1u << 14,
  RF_HPRE = 1u << 15, RF_FOLDTAB = 1u << 21,
  RF_ACT_SHIFT = 16,  // 2 bits
  RF_GENERIC = 1u << 31
};

static inline unsigned rnn_features(int epi, const RnnK& k) {
  unsigned f = 0;
  if (k.biasX) f |= RF_BIASX;
  if (k.biasH) f |= RF_BIASH;
  if (k.pre_table) f |= RF_PRE;
  if (k.pre_idx) f |= RF_PREIDX;
  if (k.fr_base) f |= RF_FRAME;
  if (k.x_res) f |= RF_XRES;
  if (k.skip_flag) f |= RF_SKIP;
  if (k.mask) f |= RF_MASK;
  if (k.drop_on && !k.mask) f |= RF_DROP;
  if (k.seq_out) f |= RF_SEQ;
  if (k.x_out) f |= RF_XOUT;
  if (k.aff_slot) f |= RF_AFFINE;
  if (k.gum_slot) f |= RF_GUMBEL;
  if (k.zero_slot) f |= RF_ZERO;
  if (k.nseg > 1) f |= RF_MULTISEG;
  if (k.h_pre) f |= RF_HPRE;
  if (k.fr_desc) f |= RF_FOLDTAB;
  if (epi == EPI_LINEAR) f |= (unsigned)(k.act & 3) << RF_ACT_SHIFT;
  return f;
}

// Device-side argument block: RnnK with the K segments flattened so that every access uses a
// compile-time index (a runtime-indexed kernarg array makes hipcc fetch the descriptor through
// vector memory and wait on it before every k-block).
struct RnnDev {
  RnnK k;
  const float* segp[4];
  int segld[4], segstart[4], segpart[4];  // segstart[t] = first k-block of segment t (INT_MAX if absent)
};

#define RHAS(bit, cond) ((F & RF_GENERIC) ? (cond) : ((F & (bit)) != 0))

// TS ("tile split", wide batches: hundreds of columns): the 8 waves of a workgroup own 2 row tiles x 4
// column tiles and each runs the WHOLE K, instead of splitting K for one row tile -- the operands of a
// workgroup tile of 32 rows x 64 columns are fetched from L2 once and re-used through L1, which is what
// bounds wide launches.  The arithmetic is kept bit-identical to the K-split form: k-block kb still
// accumulates in chain (kb mod 8) and the eight chains are added in the same order as the LDS reduction,
// so a column's result does not depend on how many other columns share the launch.
template <int EPI, int NT, int UB, unsigned F, bool TS = false>
__device__ __forceinline__ void rnn_rowtile_body(const RnnDev& d, const int bx, const int by) {
  constexpr int NW = 8;
  constexpr int KS = TS ? 1 : NW;  // k-block stride between the UB blocks of a batch
  constexpr int RL = (EPI == EPI_GRU) ? 3 : 4;
  constexpr int BLK = 4 * RL * 16;  // floats per (tile, k-block)
  // GRU keeps the hidden-part sums apart (n = tanh(i_n + r*h_n)); an instance whose hidden part comes
  // precomputed (RF_HPRE) has only input-part k-blocks and reduces one partial per wave
  constexpr int NPART = (EPI == EPI_GRU && !(F & RF_HPRE)) ? 2 : 1;
  static_assert(!TS || (UB == 8 && NT == 1 && NPART == 1), "tile-split instances: one column tile per wave, 8 accumulation chains");
  __shared__ __attribute__((aligned(16))) float red[TS ? 4 : NW * NPART * NT * 256];
  const RnnK& a = d.k;
  const bool f_biasx = RHAS(RF_BIASX, a.biasX != nullptr), f_biash = RHAS(RF_BIASH, a.biasH != nullptr);
  const bool f_pre = RHAS(RF_PRE, a.pre_table != nullptr), f_preidx = RHAS(RF_PREIDX, a.pre_idx != nullptr);
  const bool f_frame = RHAS(RF_FRAME, a.fr_base != nullptr), f_xres = RHAS(RF_XRES, a.x_res != nullptr);
  const bool f_skip = RHAS(RF_SKIP, a.skip_flag != nullptr), f_mask = RHAS(RF_MASK, a.mask != nullptr);
  const bool f_drop = RHAS(RF_DROP, a.drop_on != 0 && a.mask == nullptr), f_seq = RHAS(RF_SEQ, a.seq_out != nullptr);
  const bool f_xout = RHAS(RF_XOUT, a.x_out != nullptr), f_aff = RHAS(RF_AFFINE, a.aff_slot != nullptr);
  const bool f_gum = RHAS(RF_GUMBEL, a.gum_slot != nullptr), f_zero = RHAS(RF_ZERO, a.zero_slot != nullptr);
  const bool f_mseg = RHAS(RF_MULTISEG, a.nseg > 1);
  const bool f_hpre = RHAS(RF_HPRE, a.h_pre != nullptr);
  const bool f_ftab = RHAS(RF_FOLDTAB, a.fr_desc != nullptr);
  const int act = (F & RF_GENERIC) ? a.act : (int)((F >> RF_ACT_SHIFT) & 3);

  MB_MARK(a.trace, 0, 0);
  trace_begin(a.trace);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_mt_all = (EPI == EPI_LINEAR) ? (a.units + 15) / 16 : (a.units + 3) / 4;
  const int mt_raw = TS ? bx * 2 + (wave >> 2) : bx;
  const int mt = mt_raw < n_mt_all ? mt_raw : n_mt_all - 1;  // TS: an odd tile count leaves one wave row idle (loads legal, no stores)
  const int ntile0 = TS ? by * 4 + (wave & 3) : by * NT;  // NT column tiles share one weight fetch
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;  // dead 4th GRU row re-reads row 2
  const float* wl = a.w + (size_t)mt * a.nkb_total * BLK + ((u * RL + tau) * 4 + kq) * 4;
  int ncol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ncol[nt] = (ntile0 + nt) * 16 + i;
    if (ncol[nt] >= a.N) ncol[nt] = a.N - 1;  // duplicate a live column; its result is never stored
  }
  const bool epi_wave = TS ? true : wave < NT;
  const int en_raw = TS ? ntile0 * 16 + (lane & 15) : (ntile0 + (epi_wave ? wave : 0)) * 16 + (lane & 15);
  const int en = en_raw < a.N ? en_raw : a.N - 1;  // clamped: loads always legal
  const int edu = lane >> 4;                      // epilogue unit (or row quad) within the tile

  // ---- scalars: skip flag (stop rule), table-row index, step index.  Unconditional loads. ----
  int skip = 0, idx_raw = 0, fr_s = 0;
  if (f_skip) skip = *a.skip_flag;
  if (f_preidx) idx_raw = a.pre_idx[en];
  if (f_frame) fr_s = *a.fr_base + a.fr_off;
  // fused-sampling word of the previous step (fresh data: requested first)
  unsigned long long slotE = 0;
  if (f_aff) slotE = a.aff_slot[en];

  // ---- fragments of one batch (UB k-blocks of this wave): UB*(1+NT) float4 loads, no waits ----
  struct Frag { float4 a[UB]; float4 b[UB][NT]; int part[UB]; };
  // position of column n in the conditioning sequence (fold geometry), clamped to the zero row
  auto cond_pos = [&](int n) -> unsigned {
    const unsigned pos = (unsigned)(a.fr_n_off + n) * (unsigned)a.fr_fold_stride + (unsigned)fr_s;
    return pos;
  };
  unsigned brow[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    brow[nt] = 0;
    if (f_aff) { brow[nt] = cond_pos(ncol[nt]); if (brow[nt] > (unsigned)a.fr_total_len) brow[nt] = (unsigned)a.fr_total_len; }
  }
  auto issue = [&](Frag& f, int kb_base) {
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      int kb = kb_base + ub * KS;
      const bool valid = kb < a.nkb_total;
      if (!valid) kb = a.nkb_total - 1;
      const float* sp = d.segp[0];
      int ld = d.segld[0], local = kb, pt = d.segpart[0];
      bool seg0 = true;
      if (f_mseg) {
#pragma unroll
        for (int t = 1; t < 4; ++t) {
          const bool in = kb >= d.segstart[t];
          sp = in ? d.segp[t] : sp;
          ld = in ? d.segld[t] : ld;
          local = in ? kb - d.segstart[t] : local;
          pt = in ? d.segpart[t] : pt;
          seg0 = seg0 && !in;
        }
      }
      f.part[ub] = valid ? pt : 2;  // 2 = padding block: contributes nothing
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        // rebuilt segment 0 (f_aff): column n reads row pos_n of the conditioning table instead of row n
        const float* bp = (f_aff && seg0) ? a.aff_table + (size_t)brow[nt] * a.aff_ld : sp + (size_t)ncol[nt] * ld;
        f.b[ub][nt] = *reinterpret_cast<const float4*>(bp + local * 16 + kq * 4);
      }
      f.a[ub] = *reinterpret_cast<const float4*>(wl + (size_t)kb * BLK);
    }
  };
  f32x4 accX[NT], accH[NT];
  f32x4 accT[TS ? 8 : 1];  // TS: chain c accumulates the k-blocks kb = c (mod 8), as wave c does in the K-split form
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { accX[nt] = {0.f, 0.f, 0.f, 0.f}; accH[nt] = {0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int c = 0; c < (TS ? 8 : 1); ++c) accT[c] = {0.f, 0.f, 0.f, 0.f};
  auto consume = [&](const Frag& f) {
    if (TS) {
      // batches start at multiples of 8: block ub of a batch belongs to chain ub.  The four MFMAs of a
      // k-block depend on each other; walking the eight chains inside each component keeps consecutive
      // MFMAs independent (the counters showed 54 % issue stalls with the chains walked one after the other)
#pragma unroll
      for (int cmp = 0; cmp < 4; ++cmp)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {  // no padding blocks: tile-split launches require nkb_total % 8 == 0
          const float av = cmp == 0 ? f.a[ub].x : cmp == 1 ? f.a[ub].y : cmp == 2 ? f.a[ub].z : f.a[ub].w;
          const float bv = cmp == 0 ? f.b[ub][0].x : cmp == 1 ? f.b[ub][0].y : cmp == 2 ? f.b[ub][0].z : f.b[ub][0].w;
          accT[ub & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, accT[ub & 7], 0, 0, 0);
        }
      return;
    }
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      if (f.part[ub] == 2) continue;  // wave-uniform
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 b = f.b[ub][nt];
        if (NPART == 2 && f.part[ub] == 1) {
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].x, b.x, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].y, b.y, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].z, b.z, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].w, b.w, accH[nt], 0, 0, 0);
        } else {
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].x, b.x, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].y, b.y, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].z, b.z, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].w, b.w, accX[nt], 0, 0, 0);
        }
      }
    }
  };
  Frag f0, f1;
  issue(f0, TS ? 0 : wave);

  // ---- epilogue operands: issued by every wave right behind the first batch and consumed only
  //      after the reduction barrier, so the MFMA chain never waits on them ----
  const int H = a.units;
  int ej = mt * 4 + edu;  // GRU/LSTM hidden unit of this lane
  if (ej >= H) ej = H - 1;
  const int erow = mt * 16 + edu * 4;  // LINEAR first row of this lane's quad
  int prow = a.pre_base_row + en * a.pre_n_stride;
  if (f_preidx) prow = idx_raw;
  unsigned posE = 0;
  int dsc_fold = 0;
  unsigned dsc_slo = 0, dsc_shi = 0;
  if (f_frame && f_ftab) {  // several utterances: per-fold descriptor (stable data) instead of the fold arithmetic
    const int4 d0 = *reinterpret_cast<const int4*>(a.fr_desc + (size_t)(a.fr_n_off + en) * 8);
    const int4 d1 = *reinterpret_cast<const int4*>(a.fr_desc + (size_t)(a.fr_n_off + en) * 8 + 4);
    posE = (unsigned)(d0.x + fr_s);
    prow = d0.w + (posE < (unsigned)d0.y ? (int)(posE / (unsigned)a.fr_hop) : d1.x);
    dsc_fold = d1.y; dsc_slo = (unsigned)d1.z; dsc_shi = (unsigned)d1.w;
  } else if (f_frame) {
    posE = cond_pos(en);
    prow = posE < (unsigned)a.fr_total_len ? (int)(posE / (unsigned)a.fr_hop) : a.fr_frames;
  }
  float l_bx[4] = {0.f, 0.f, 0.f, 0.f}, l_pre[4] = {0.f, 0.f, 0.f, 0.f}, l_bh[4] = {0.f, 0.f, 0.f, 0.f};
  float l_mask[4] = {1.f, 1.f, 1.f, 1.f};
  float l_hp = 0.f, l_cp = 0.f, l_xr = 0.f, l_xw = 0.f, l_ag[4] = {0.f, 0.f, 0.f, 0.f};
  float l_hs[4] = {0.f, 0.f, 0.f, 0.f};  // precomputed hidden-part pre-activations (W_hh.h + b_hh)
  {
    const float* prp = a.pre_table + (size_t)prow * a.pre_stride;
    const size_t so = (size_t)en * H + ej;
    if (EPI == EPI_LINEAR) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = erow + r < H ? erow + r : H - 1;
        if (f_biasx) l_bx[r] = a.biasX[row];
        if (f_pre) l_pre[r] = prp[row];
        if (f_mask) l_mask[r] = a.mask[(size_t)en * a.ldy + row];
      }
    } else {
#pragma unroll
      for (int g = 0; g < RL; ++g) {
        if (f_biasx) l_bx[g] = a.biasX[g * H + ej];
        if (f_pre) l_pre[g] = prp[g * H + ej];
        if (f_biash) l_bh[g] = a.biasH[g * H + ej];
        if (EPI != EPI_LINEAR && f_hpre) l_hs[g] = a.h_pre[(size_t)en * (RL * H) + g * H + ej];
      }
      if (EPI == EPI_GRU) l_hp = a.h_prev[so];
      if (EPI == EPI_LSTM) l_cp = a.c_prev[so];
      if (f_aff) {
        unsigned pos = posE > (unsigned)a.fr_total_len ? (unsigned)a.fr_total_len : posE;
        l_xr = a.aff_table[(size_t)pos * a.aff_ld + ej];
        l_xw = a.aff_vec[ej];
#pragma unroll
        for (int g = 0; g < RL; ++g) l_ag[g] = a.aff_gate[g * H + ej];
      } else if (f_xres) {
        l_xr = a.x_res[so];
      }
    }
  }
  MB_MARK(a.trace, 1, 0);
  MB_MARK(a.trace, 2, 1);

  // This wave owns k-blocks wave, wave+NW, ... of the concatenated K, UB per batch; the next
  // batch's loads are in flight while the current batch feeds the MFMA chain.
  for (int kb_base = TS ? 0 : wave; kb_base < a.nkb_total; kb_base += 2 * KS * UB) {
    const int kb1 = kb_base + KS * UB, kb2 = kb_base + 2 * KS * UB;
    if (kb1 < a.nkb_total) issue(f1, kb1);
    consume(f0);
    if (kb1 < a.nkb_total) {
      if (kb2 < a.nkb_total) issue(f0, kb2);
      consume(f1);
    }
  }
  MB_MARK(a.trace, 3, 0);
  // cross-wave reduction through LDS: D fragment lane = (unit = lane>>4, col = lane&15), reg = gate
  float4* red4 = reinterpret_cast<float4*>(red);
  if (!TS) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      red4[((wave * NT + nt) * NPART + 0) * 64 + lane] = make_float4(accX[nt][0], accX[nt][1], accX[nt][2], accX[nt][3]);
      if (NPART == 2)
        red4[((wave * NT + nt) * NPART + 1) * 64 + lane] = make_float4(accH[nt][0], accH[nt][1], accH[nt][2], accH[nt][3]);
    }
    __syncthreads();
  }
  MB_MARK(a.trace, 4, 0);
  if (!epi_wave || skip) return;
  if (TS && mt_raw >= n_mt_all) return;
  float sx[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (TS) {  // the eight chains, added in the order the K-split form adds its eight waves
#pragma unroll
    for (int c = 0; c < 8; ++c) { sx[0] += accT[c & (TS ? 7 : 0)][0]; sx[1] += accT[c & (TS ? 7 : 0)][1]; sx[2] += accT[c & (TS ? 7 : 0)][2]; sx[3] += accT[c & (TS ? 7 : 0)][3]; }
  }
#pragma unroll
  for (int w = 0; w < (TS ? 0 : NW); ++w) {
    const float4 v = red4[((w * NT + wave) * NPART + 0) * 64 + lane];
    sx[0] += v.x; sx[1] += v.y; sx[2] += v.z; sx[3] += v.w;
    if (NPART == 2) {
      const float4 h = red4[((w * NT + wave) * NPART + 1) * 64 + lane];
      sh[0] += h.x; sh[1] += h.y; sh[2] += h.z; sh[3] += h.w;
    }
  }
  MB_MARK(a.trace, 5, 1);
  if (EPI == EPI_GRU && f_hpre) {
#pragma unroll
    for (int g = 0; g < RL; ++g) sh[g] += l_hs[g];
  }
  if (EPI == EPI_LSTM && f_hpre) {  // LSTM gates are plain sums: the precomputed hidden half joins the input half
#pragma unroll
    for (int g = 0; g < RL; ++g) sx[g] += l_hs[g];
  }
  const int n = en_raw, du = edu;
  const float xsE = (f_aff && slotE) ? 2.f * (float)argmax_class(slotE) / ((float)a.aff_C - 1.f) - 1.f : 0.f;
  if (f_aff && mt == 0 && du == 0 && n < a.N && fr_s > 0) {  // previous step's sample -> output tensor
    a.aff_samples[(size_t)(a.fr_n_off + n) * a.aff_S + (fr_s - 1)] = xsE;
    if (a.aff_progress && a.fr_n_off + n == 0 && (fr_s - 1) % 100 == 0) *a.aff_progress = fr_s;
  }
  if (f_zero && mt == 0 && du == 0 && n < a.N) a.zero_slot[n] = 0ull;
  if (n >= a.N) return;

  if (EPI == EPI_LINEAR) {
    float best = -INFINITY;
    int bcls = 0;
    uint32_t gr[4] = {0u, 0u, 0u, 0u};
    if (f_gum && f_ftab) philox4x32((uint32_t)fr_s, (uint32_t)dsc_fold, (uint32_t)((mt * 16 + du * 4) >> 2), 0x57415645u,
                                    dsc_slo, dsc_shi, gr);
    else if (f_gum) philox4x32((uint32_t)fr_s, (uint32_t)(a.fr_n_off + n), (uint32_t)((mt * 16 + du * 4) >> 2), 0x57415645u,
                               (uint32_t)a.gum_seed, (uint32_t)(a.gum_seed >> 32), gr);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mt * 16 + du * 4 + r;
      if (row < a.units) {
        float v = sx[r] + (l_bx[r] + l_pre[r]);
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = sigmoidf_(v);
        else if (act == 3) v = tanhf(v);
        if (f_mask) v = v * (l_mask[r] * a.mask_scale);
        else if (f_drop) {
          uint32_t rr[4];
          philox4x32((uint32_t)a.drop_iter, (uint32_t)a.drop_layer, (uint32_t)n, (uint32_t)(row >> 2),
                     (uint32_t)a.drop_seed, (uint32_t)(a.drop_seed >> 32), rr);
          v = v * ((rr[row & 3] >= a.drop_thresh) ? a.mask_scale : 0.f);
        }
        if (a.y) a.y[(size_t)n * a.ldy + row] = v;
        if (f_gum) {
          const float g = v - logf(-logf(u32_to_unit(gr[r])));
          if (g > best) { best = g; bcls = row; }  // ascending rows: first maximum kept
        }
      }
    }
    if (f_gum) {
      // the 4 row quads of this column sit in lanes l, l+16, l+32, l+48
      unsigned long long pk = pack_argmax(best, bcls);
      const unsigned long long o1 = __shfl_xor(pk, 16, 64);
      pk = o1 > pk ? o1 : pk;
      const unsigned long long o2 = __shfl_xor(pk, 32, 64);
      pk = o2 > pk ? o2 : pk;
      if (du == 0) atomicMax(a.gum_slot + n, pk);
    }
    MB_MARK(a.trace, 6, 0);
    trace_end(a.trace);
    return;
  }
  const int j = mt * 4 + du;  // hidden unit
  if (j >= a.units) return;
  const size_t so = (size_t)n * H + j;
  const float e_xr = f_aff ? l_xr + xsE * l_xw : l_xr;
  if (f_aff) {  // W_ih.(row + x*w) = W_ih.row + x*(W_ih.w): the second term is a per-gate constant vector
#pragma unroll
    for (int g = 0; g < RL; ++g) sx[g] += xsE * l_ag[g];
  }
  if (EPI == EPI_GRU) {
    // torch GRUCell (gate order r,z,n): r = s(i_r+h_r), z = s(i_z+h_z), n = tanh(i_n + r*h_n),
    // h' = n + z*(h - n).   models/vocoder/wavernn/models/fatchord_version.py:196-200,265-271;
    // models/synthesizer/models/tacotron.py:60,98
    const float rg = sigmoidf_((sx[0] + (l_bx[0] + l_pre[0])) + (sh[0] + l_bh[0]));
    const float zg = sigmoidf_((sx[1] + (l_bx[1] + l_pre[1])) + (sh[1] + l_bh[1]));
    const float ng = tanhf((sx[2] + (l_bx[2] + l_pre[2])) + rg * (sh[2] + l_bh[2]));
    const float hy = ng + zg * (l_hp - ng);
    a.h_out[so] = hy;
    if (f_xout) a.x_out[so] = e_xr + hy;
    if (f_seq) a.seq_out[(long long)n * a.seq_n_stride + (long long)j * a.seq_j_stride + a.seq_off] = hy;
  } else {
    // torch LSTMCell (gate order i,f,g,o).  tacotron.py:62-63,112-125
    const float gi = sigmoidf_(sx[0] + (l_bx[0] + l_pre[0]) + l_bh[0]);
    const float gf = sigmoidf_(sx[1] + (l_bx[1] + l_pre[1]) + l_bh[1]);
    const float gg = tanhf(sx[2] + (l_bx[2] + l_pre[2]) + l_bh[2]);
    const float go = sigmoidf_(sx[3] + (l_bx[3] + l_pre[3]) + l_bh[3]);
    const float cy = gf * l_cp + gi * gg;
    const float hy = go * tanhf(cy);
    a.c_out[so] = cy;
    a.h_out[so] = hy;
    if (f_xout) a.x_out[so] = e_xr + hy;
  }
  MB_MARK(a.trace, 6, 0);
  trace_end(a.trace);
}


// fill the flat-segment argument block; returns MB_EINVAL if the segments do not cover nkb_total
static inline int make_rnn_dev(const RnnK& k, RnnDev* d) {
  d->k = k;
  int start = 0;
  for (int t = 0; t < 4; ++t) {
    if (t < k.nseg) {
      d->segp[t] = k.seg[t].p; d->segld[t] = k.seg[t].ld; d->segpart[t] = k.seg[t].part; d->segstart[t] = start;
      start += k.seg[t].nkb;
    } else {
      d->segp[t] = k.seg[0].p; d->segld[t] = 0; d->segpart[t] = 0; d->segstart[t] = 0x7fffffff;
    }
  }
  MB_REQUIRE(start == k.nkb_total, "rnn_launch: segments cover %d k-blocks, nkb_total=%d", start, k.nkb_total);
  return MB_OK;
}

}  // namespace mb
