// Row-tile MFMA recurrent GEMM kernels (see rnn.h for the design).
//
// What bounds these launches on MI355X (profiles/r01_wavernn_step_timeline.md): each one is a few us of
// dependent latencies, not bandwidth --
//   * ~1.5 us launch-to-launch gap, ~0.5 us for the first instruction fetch of a launch (the
//     instruction cache is cold at every dispatch; sequential fetch then keeps up, but every taken
//     branch into a line that was not prefetched pays the miss again),
//   * one ~0.9 us round trip for data the PREVIOUS launch wrote (it comes from another XCD's L2 via
//     the fabric), ~0.2 us for data that is stable across launches (weights, tables, biases),
//   * the fp32 MFMA chain itself (256 FLOP/clk/CU).
// Hence: (1) the kernel is specialised at compile time on the feature set the caller uses (F), so a
// launch executes straight-line code with no dead branches; (2) every load is issued before the
// first wait; (3) nothing the previous launch wrote is read except the activations themselves.
#include "rnn.h"

namespace mb {

// Feature bits of a specialised instance.  RF_GENERIC = decide everything at run time (fallback).
enum : unsigned {
  RF_BIASX = 1u << 0, RF_BIASH = 1u << 1, RF_PRE = 1u << 2, RF_PREIDX = 1u << 3, RF_FRAME = 1u << 4,
  RF_XRES = 1u << 5, RF_SKIP = 1u << 6, RF_MASK = 1u << 7, RF_DROP = 1u << 8, RF_SEQ = 1u << 9,
  RF_XOUT = 1u << 10, RF_AFFINE = 1u << 11, RF_GUMBEL = 1u << 12, RF_ZERO = 1u << 13, RF_MULTISEG = 1u << 14,
  RF_HPRE = 1u << 15, RF_ARRIVE = 1u << 20,
  RF_ACT_SHIFT = 16,  // 2 bits
  RF_GENERIC = 1u << 31
};

static unsigned rnn_features(int epi, const RnnK& k) {
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
  if (k.arrive) f |= RF_ARRIVE;
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

template <int EPI, int NT, int UB, unsigned F>
__device__ __forceinline__ void rnn_rowtile_body(const RnnDev& d, const int bx, const int by) {
  constexpr int NW = 8;
  constexpr int RL = (EPI == EPI_GRU) ? 3 : 4;
  constexpr int BLK = 4 * RL * 16;  // floats per (tile, k-block)
  // GRU keeps the hidden-part sums apart (n = tanh(i_n + r*h_n)); an instance whose hidden part comes
  // precomputed (RF_HPRE) has only input-part k-blocks and reduces one partial per wave
  constexpr int NPART = (EPI == EPI_GRU && !(F & RF_HPRE)) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float red[NW * NPART * NT * 256];
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
  const bool f_arrive = RHAS(RF_ARRIVE, a.arrive != nullptr);
  const int act = (F & RF_GENERIC) ? a.act : (int)((F >> RF_ACT_SHIFT) & 3);

  MB_MARK(a.trace, 0, 0);
  trace_begin(a.trace);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = bx, ntile0 = by * NT;  // NT column tiles share one weight fetch
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;  // dead 4th GRU row re-reads row 2
  const float* wl = a.w + (size_t)mt * a.nkb_total * BLK + ((u * RL + tau) * 4 + kq) * 4;
  int ncol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ncol[nt] = (ntile0 + nt) * 16 + i;
    if (ncol[nt] >= a.N) ncol[nt] = a.N - 1;  // duplicate a live column; its result is never stored
  }
  const bool epi_wave = wave < NT;
  const int en_raw = (ntile0 + (epi_wave ? wave : 0)) * 16 + (lane & 15);
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
      int kb = kb_base + ub * NW;
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
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { accX[nt] = {0.f, 0.f, 0.f, 0.f}; accH[nt] = {0.f, 0.f, 0.f, 0.f}; }
  auto consume = [&](const Frag& f) {
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
  issue(f0, wave);

  // ---- epilogue operands: issued by every wave right behind the first batch and consumed only
  //      after the reduction barrier, so the MFMA chain never waits on them ----
  const int H = a.units;
  int ej = mt * 4 + edu;  // GRU/LSTM hidden unit of this lane
  if (ej >= H) ej = H - 1;
  const int erow = mt * 16 + edu * 4;  // LINEAR first row of this lane's quad
  int prow = a.pre_base_row + en * a.pre_n_stride;
  if (f_preidx) prow = idx_raw;
  unsigned posE = 0;
  if (f_frame) {
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
        if (EPI == EPI_GRU && f_hpre) l_hs[g] = a.h_pre[(size_t)en * (RL * H) + g * H + ej];
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
  for (int kb_base = wave; kb_base < a.nkb_total; kb_base += 2 * NW * UB) {
    const int kb1 = kb_base + NW * UB, kb2 = kb_base + 2 * NW * UB;
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
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    red4[((wave * NT + nt) * NPART + 0) * 64 + lane] = make_float4(accX[nt][0], accX[nt][1], accX[nt][2], accX[nt][3]);
    if (NPART == 2)
      red4[((wave * NT + nt) * NPART + 1) * 64 + lane] = make_float4(accH[nt][0], accH[nt][1], accH[nt][2], accH[nt][3]);
  }
  __syncthreads();
  MB_MARK(a.trace, 4, 0);
  if (!epi_wave || skip) return;
  float sx[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w) {
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
    if (f_gum) philox4x32((uint32_t)fr_s, (uint32_t)(a.fr_n_off + n), (uint32_t)((mt * 16 + du * 4) >> 2), 0x57415645u,
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
          v = v * ((rr[row & 3] & 0x80000000u) ? a.mask_scale : 0.f);
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
      if (f_arrive) {
        // the returned value proves the atomic was performed at the device coherence point; only then
        // does this workgroup count as arrived (relaxed agent-scope add, nothing else to publish)
        unsigned int lo = 0;
        if (du == 0) lo = (unsigned int)atomicMax(a.gum_slot + n, pk);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(lo) : : "memory");
        if (lane == 0) __hip_atomic_fetch_add(a.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (du == 0) atomicMax(a.gum_slot + n, pk);
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

template <int EPI, int NT, int UB, unsigned F>
__global__ __launch_bounds__(512) void rnn_rowtile_kernel(RnnDev d) {
  rnn_rowtile_body<EPI, NT, UB, F>(d, blockIdx.x, blockIdx.y);
}

// ---- gru1-finish job (Fin1K, rnn.h): one thread per (fold, unit), all loads issued before first use ----
template <bool WAIT>
__device__ __forceinline__ void gru1_finish_body(const Fin1K& a, const int block, const int nthreads) {
  const int idx = block * nthreads + threadIdx.x;
  const int H = a.R;
  int n = idx / H;
  const int j = idx - n * H;
  const bool live = n < a.nl;
  if (!live) n = a.nl - 1;  // clamped: loads legal, nothing stored (keeps every thread at the barrier below)
  const int s = *a.step_base + a.step_off;
  const float* p1 = a.P1 + (size_t)n * 3 * H + j;
  const float hr = p1[0], hz = p1[H], hn = p1[2 * H];
  const float hp = a.h_prev[(size_t)n * H + j];
  unsigned pos = (unsigned)(a.n_off + n) * (unsigned)a.fold_stride + (unsigned)s;
  if (pos > (unsigned)a.total_len) pos = (unsigned)a.total_len;  // zero-conditioning row
  const float* t1 = a.T1 + (size_t)pos * 3 * H + j;
  const float tr = t1[0], tz = t1[H], tn = t1[2 * H];
  const float ip = a.Ipre[(size_t)pos * H + j];
  const float gr = a.g1[j], gz = a.g1[H + j], gn = a.g1[2 * H + j], w0 = a.wI0[j];
  unsigned long long slot;
  if (WAIT) {
    // every fc3 workgroup of this launch has added 1 to *arrive after its argmax atomics were performed
    if (threadIdx.x == 0) {
      const unsigned int target = (unsigned int)s * a.arrive_per_step;  // s = index of the step being prepared
      int spins = 0;
      while (__hip_atomic_load(a.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 17))
        __builtin_amdgcn_s_sleep(1);  // bounded: a lost arrival costs wrong samples, never a hung GPU
    }
    __syncthreads();
    slot = __hip_atomic_load(a.slot + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    slot = a.slot[n];
  }
  if (!live) return;
  const float x = slot ? 2.f * (float)argmax_class(slot) / ((float)a.C - 1.f) - 1.f : 0.f;
  // torch GRUCell, gate order (r, z, n)
  const float rg = sigmoidf_((tr + x * gr) + hr);
  const float zg = sigmoidf_((tz + x * gz) + hz);
  const float ng = tanhf((tn + x * gn) + rg * hn);
  const float hy = ng + zg * (hp - ng);
  a.h_out[(size_t)n * H + j] = hy;
  a.x_out[(size_t)n * H + j] = (ip + x * w0) + hy;
  if (j == 0 && s > 0) {  // previous step's sample -> output tensor
    a.samples[(size_t)(a.n_off + n) * a.S + (s - 1)] = x;
    if (a.progress && a.n_off + n == 0 && (s - 1) % 100 == 0) *a.progress = s;
  }
}

__global__ __launch_bounds__(256) void wavernn_gru1_finish_kernel(Fin1K a) {
  trace_begin(a.trace);
  gru1_finish_body<false>(a, blockIdx.x, 256);
  trace_end(a.trace);
}

// fc3 + sampler (job 0, first workgroups) and the NEXT step's gru1-finish (job 1) in one launch.
// Job 1 needs every job-0 workgroup's argmax: job 0 arrives on a device-scope counter (RF_ARRIVE),
// job 1 prefetches everything that does not depend on the sample and then polls the counter.  All
// nx0*gridDim.y + finish workgroups are co-resident (far fewer than CUs), job 0 is dispatched first.
template <unsigned F0>
__global__ __launch_bounds__(512) void rnn_fc3_finish_kernel(RnnDev d0, Fin1K f, int nx0) {
  if ((int)blockIdx.x < nx0) { rnn_rowtile_body<EPI_LINEAR, 1, 4, F0>(d0, blockIdx.x, blockIdx.y); return; }
  if (blockIdx.y != 0) return;
  gru1_finish_body<true>(f, blockIdx.x - nx0, 512);
}

// Two independent LINEAR jobs in ONE launch: workgroups with blockIdx.x < nx0 run job 0 (the one on
// the critical path: they are dispatched first), the rest run job 1 on the CUs job 0 leaves idle.
// The WaveRNN loop uses it to compute the hidden halves W_hh.h + b_hh of the NEXT step's GRUs beside
// fc1 / fc2 (wavernn.hip), which takes them off the dependent chain.
template <int UB0, unsigned F0, int UB1, unsigned F1>
__global__ __launch_bounds__(512) void rnn_dual_linear_kernel(RnnDev d0, RnnDev d1, int nx0) {
  if ((int)blockIdx.x < nx0) rnn_rowtile_body<EPI_LINEAR, 1, UB0, F0>(d0, blockIdx.x, blockIdx.y);
  else rnn_rowtile_body<EPI_LINEAR, 1, UB1, F1>(d1, blockIdx.x - nx0, blockIdx.y);
}

void pack_rowtile(const float* rows, int n_live_rows, int K, int RL, std::vector<float>* out) {
  const int per_tile = 4 * RL;
  const int n_mt = (n_live_rows + per_tile - 1) / per_tile;
  const int nkb = K / 16;
  out->assign((size_t)n_mt * nkb * per_tile * 16, 0.f);
  for (int mt = 0; mt < n_mt; ++mt)
    for (int kb = 0; kb < nkb; ++kb)
      for (int r = 0; r < per_tile; ++r) {
        const int row = mt * per_tile + r;
        if (row >= n_live_rows) continue;
        float* dst = out->data() + (((size_t)mt * nkb + kb) * per_tile + r) * 16;
        const float* src = rows + (size_t)row * K + kb * 16;
        for (int q = 0; q < 16; ++q) dst[q] = src[q];
      }
}

void cell_rows(const float* w_ih, int kx, int ldx, const float* w_hh, int kh, int H, int G,
               std::vector<float>* rows) {
  const int K = kx + kh;
  rows->assign((size_t)H * G * K, 0.f);
  for (int j = 0; j < H; ++j)
    for (int g = 0; g < G; ++g) {
      float* dst = rows->data() + ((size_t)j * G + g) * K;
      memcpy(dst, w_ih + (size_t)(g * H + j) * ldx, sizeof(float) * kx);
      memcpy(dst + kx, w_hh + (size_t)(g * H + j) * kh, sizeof(float) * kh);
    }
}

// Specialised instances: the exact (epilogue, column tiles, batch depth, feature set) tuples the two
// autoregressive loops and the CBHG scan launch.  Anything else runs the RF_GENERIC instance.
#define ACT(n) ((unsigned)(n) << RF_ACT_SHIFT)
#define MB_RNN_INSTANCES(X)                                                                               \
  /* WaveRNN: rnn1 (exact / fused), rnn2 (+ slot clear), fc1|fc2, fc3 (exact / fused) */                  \
  X(EPI_GRU, 1, 8, RF_BIASX | RF_BIASH | RF_XRES | RF_XOUT | RF_MULTISEG)                                 \
  X(EPI_GRU, 1, 8, RF_BIASX | RF_BIASH | RF_FRAME | RF_AFFINE | RF_XOUT | RF_MULTISEG)                    \
  X(EPI_GRU, 1, 8, RF_PRE | RF_FRAME | RF_BIASH | RF_XRES | RF_XOUT | RF_MULTISEG)                        \
  X(EPI_GRU, 1, 8, RF_PRE | RF_FRAME | RF_BIASH | RF_XRES | RF_XOUT | RF_MULTISEG | RF_ZERO)              \
  X(EPI_LINEAR, 1, 4, RF_PRE | RF_FRAME | ACT(1))                                                         \
  X(EPI_LINEAR, 1, 4, RF_BIASX)                                                                           \
  X(EPI_LINEAR, 1, 4, RF_BIASX | RF_FRAME | RF_GUMBEL)                                                    \
  /* WaveRNN split-hidden chain: rnn2 on its input half only (hidden half precomputed) */                 \
  X(EPI_GRU, 1, 4, RF_PRE | RF_FRAME | RF_HPRE | RF_XRES | RF_XOUT | RF_ZERO)                             \
  /* Tacotron decoder: prenet fc1/fc2 (mask / on-device dropout), attention GRU, rnn_input, LSTMs, mel */ \
  X(EPI_LINEAR, 1, 2, RF_BIASX | RF_SKIP | RF_MASK | ACT(1))                                              \
  X(EPI_LINEAR, 1, 2, RF_BIASX | RF_SKIP | RF_DROP | ACT(1))                                              \
  X(EPI_GRU, 1, 8, RF_BIASX | RF_BIASH | RF_SKIP | RF_MULTISEG)                                           \
  X(EPI_LINEAR, 1, 8, RF_BIASX | RF_SKIP | RF_MULTISEG)                                                   \
  X(EPI_LSTM, 2, 4, RF_BIASX | RF_BIASH | RF_XRES | RF_XOUT | RF_SKIP | RF_MULTISEG)                      \
  X(EPI_LSTM, 1, 8, RF_BIASX | RF_BIASH | RF_XRES | RF_XOUT | RF_SKIP | RF_MULTISEG)                      \
  X(EPI_LINEAR, 1, 8, RF_SKIP)                                                                            \
  /* CBHG bidirectional GRU scan */                                                                       \
  X(EPI_GRU, 1, 2, RF_PRE | RF_BIASH | RF_SEQ)

static int make_rnn_dev(const RnnK& k, RnnDev* d) {
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

int rnn_launch_dual_linear(const RnnK& k0, const RnnK& k1, hipStream_t s) {
  constexpr int NW = 8;
  MB_REQUIRE(k0.N >= 1 && k0.N == k1.N && k0.nseg == 1 && k1.nseg == 1, "rnn_launch_dual: bad shape");
  RnnDev d0, d1;
  int rc = make_rnn_dev(k0, &d0);
  if (!rc) rc = make_rnn_dev(k1, &d1);
  if (rc) return rc;
  const int nx0 = cdiv(k0.units, 16), nx1 = cdiv(k1.units, 16);
  const unsigned f0 = rnn_features(EPI_LINEAR, k0), f1 = rnn_features(EPI_LINEAR, k1);
  constexpr unsigned F0 = RF_PRE | RF_FRAME | (1u << RF_ACT_SHIFT), F1 = RF_BIASX;
  const int pw0 = cdiv(k0.nkb_total, NW), pw1 = cdiv(k1.nkb_total, NW);
  MB_REQUIRE(f0 == F0 && f1 == F1 && pw0 == 4 && pw1 == 4,
             "rnn_launch_dual: only the (relu table linear, biased linear) K=512 pair is instantiated (features %x/%x)", f0, f1);
  dim3 grid(nx0 + nx1, cdiv(k0.N, 16));
  hipLaunchKernelGGL((rnn_dual_linear_kernel<4, F0, 4, F1>), grid, dim3(NW * 64), 0, s, d0, d1, nx0);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

int rnn_launch_finish(const Fin1K& f, hipStream_t s) {
  MB_REQUIRE(f.nl >= 1 && f.R >= 1, "rnn_launch_finish: bad shape");
  hipLaunchKernelGGL(wavernn_gru1_finish_kernel, dim3(cdiv(f.nl * f.R, 256)), dim3(256), 0, s, f);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

int rnn_launch_fc3_finish(const RnnK& k, const Fin1K& f, hipStream_t s) {
  constexpr int NW = 8;
  constexpr unsigned F0 = RF_BIASX | RF_FRAME | RF_GUMBEL | RF_ARRIVE;
  RnnDev d0;
  int rc = make_rnn_dev(k, &d0);
  if (rc) return rc;
  MB_REQUIRE(rnn_features(EPI_LINEAR, k) == F0 && cdiv(k.nkb_total, NW) == 4 && k.nseg == 1 && f.arrive == k.arrive,
             "rnn_launch_fc3_finish: needs the fused-sampler fc3 instance with K = 512 (features %x)", rnn_features(EPI_LINEAR, k));
  const int nx0 = cdiv(k.units, 16), ny = cdiv(k.N, 16);
  MB_REQUIRE(f.arrive_per_step == (unsigned)(nx0 * ny), "rnn_launch_fc3_finish: arrive_per_step must be %d", nx0 * ny);
  dim3 grid(nx0 + cdiv(f.nl * f.R, NW * 64), ny);
  hipLaunchKernelGGL((rnn_fc3_finish_kernel<F0>), grid, dim3(NW * 64), 0, s, d0, f, nx0);
  MB_HIP(hipGetLastError());
  return MB_OK;
}

int rnn_launch(int epi, const RnnK& k, hipStream_t s) {
  MB_REQUIRE(k.N >= 1 && k.units >= 1 && k.nseg >= 1 && k.nseg <= 4, "rnn_launch: bad shape");
  MB_REQUIRE(!k.aff_slot || (k.fr_base && k.nseg >= 1 && epi == EPI_GRU), "rnn_launch: rebuilt segment needs the fold geometry");
  MB_REQUIRE(!k.gum_slot || (k.fr_base && epi == EPI_LINEAR && k.units % 4 == 0), "rnn_launch: fused sampler needs the step index");
  MB_REQUIRE(!k.h_pre || (epi == EPI_GRU && !k.biasH), "rnn_launch: h_pre is a GRU feature and already holds b_hh");
  constexpr int NW = 8;
  const int n_mt = (epi == EPI_LINEAR) ? cdiv(k.units, 16) : cdiv(k.units, 4);
  // More than 16 columns AND a weight matrix big enough to be bandwidth-bound (the batch-32
  // Tacotron LSTMs, 33.5 MB): one workgroup covers two 16-column tiles so the weights are
  // streamed once per 32 columns.  Small matrices (WaveRNN, <= 6.6 MB) are latency-bound and run
  // faster with twice the workgroups.
  const int rl = (epi == EPI_GRU) ? 3 : 4;
  const size_t wbytes = (size_t)n_mt * k.nkb_total * 4 * rl * 16 * sizeof(float);
  const int nt = (k.N > 16 && wbytes >= ((size_t)12 << 20)) ? 2 : 1;
  RnnDev d;
  int rcd = make_rnn_dev(k, &d);
  if (rcd) return rcd;
  const int per_wave = cdiv(k.nkb_total, NW);
  const int ub = nt == 2 ? (per_wave >= 4 ? 4 : 2) : (per_wave >= 8 ? 8 : (per_wave >= 4 ? 4 : 2));
  const unsigned feat = rnn_features(epi, k);
  dim3 grid(n_mt, cdiv(k.N, 16 * nt));
  bool done = false;
#define MB_TRY(EPI_, NT_, UB_, F_)                                                                         \
  if (!done && epi == (EPI_) && nt == (NT_) && ub == (UB_) && feat == (unsigned)(F_)) {                    \
    hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_, NT_, UB_, (unsigned)(F_)>), grid, dim3(NW * 64), 0, s, d); \
    done = true;                                                                                           \
  }
  if (!getenv("MBHIP_RNN_GENERIC")) { MB_RNN_INSTANCES(MB_TRY) }
#undef MB_TRY
#define MB_RNN(EPI_, NT_, UB_) hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_, NT_, UB_, RF_GENERIC>), grid, dim3(NW * 64), 0, s, d)
#define MB_RNN_E(EPI_)                                                                    \
  do {                                                                                    \
    if (nt == 2) { if (ub == 4) MB_RNN(EPI_, 2, 4); else MB_RNN(EPI_, 2, 2); }            \
    else if (ub == 8) MB_RNN(EPI_, 1, 8);                                                 \
    else if (ub == 4) MB_RNN(EPI_, 1, 4);                                                 \
    else MB_RNN(EPI_, 1, 2);                                                              \
  } while (0)
  if (!done) {
    if (epi == EPI_LINEAR) MB_RNN_E(EPI_LINEAR);
    else if (epi == EPI_GRU) MB_RNN_E(EPI_GRU);
    else MB_RNN_E(EPI_LSTM);
  }
#undef MB_RNN_E
#undef MB_RNN
  MB_HIP(hipGetLastError());
  return MB_OK;
}

}  // namespace mb
