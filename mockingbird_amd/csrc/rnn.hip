// Row-tile MFMA recurrent GEMM kernels (see rnn.h for the design).
#include "rnn.h"

namespace mb {

// Device-side argument block: RnnK with the K segments flattened so that every access uses a
// compile-time index (a runtime-indexed kernarg array makes hipcc fetch the descriptor through
// vector memory and wait on it before every k-block -- measured 10 us per launch, see
// profiles/r01_wavernn_step_timeline.md).
struct RnnDev {
  RnnK k;
  const float* segp[4];
  int segld[4], segstart[4], segpart[4];  // segstart[t] = first k-block of segment t (INT_MAX if absent)
};

// Latency-first structure (the launch is ~2 us of work, so ONE exposed memory round trip matters):
//   1. every load the workgroup will ever need is issued before the first wait, in one
//      straight-line block: epilogue indices, A/B fragments of the wave's first UB k-blocks,
//      epilogue operands (bias / table row / previous state).  No load sits behind a branch whose
//      condition depends on memory, and nothing is consumed until the MFMA chain starts.
//   2. the wave's k-block -> segment mapping is wave-uniform scalar selects (no indexed kernarg).
//   3. GRU tiles have 3 live gate rows per unit; lanes of the dead 4th row re-read gate row 2
//      instead of being predicated off (their D rows are never read).
template <int EPI, int NT, int UB>
__global__ __launch_bounds__(512) void rnn_rowtile_kernel(RnnDev d) {
  constexpr int NW = 8;
  constexpr int RL = (EPI == EPI_GRU) ? 3 : 4;
  constexpr int BLK = 4 * RL * 16;  // floats per (tile, k-block)
  constexpr int NPART = (EPI == EPI_GRU) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float red[NW * NPART * NT * 256];
  const RnnK& a = d.k;

  MB_MARK(a.trace, 0, 0);
  trace_begin(a.trace);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = blockIdx.x, ntile0 = blockIdx.y * NT;  // NT column tiles share one weight fetch
  const int i = lane & 15, kq = lane >> 4;
  const int u = i >> 2, tau = (i & 3) < RL ? (i & 3) : RL - 1;
  const float* wl = a.w + (size_t)mt * a.nkb_total * BLK + ((u * RL + tau) * 4 + kq) * 4;
  int ncol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ncol[nt] = (ntile0 + nt) * 16 + i;
    if (ncol[nt] >= a.N) ncol[nt] = a.N - 1;  // duplicate a live column; its result is never stored
  }
  // skip flag (decoder stop rule) and table-row index: fetched unconditionally (a null pointer is
  // replaced by a harmless valid address) so no load hides behind a branch; tested / used later.
  const int en_raw = (ntile0 + (wave < NT ? wave : 0)) * 16 + (lane & 15);
  const int en = en_raw < a.N ? en_raw : a.N - 1;  // clamped: loads always legal
  const int* skp = a.skip_flag ? a.skip_flag : reinterpret_cast<const int*>(a.w);
  const int* idp = a.pre_idx ? a.pre_idx + en : reinterpret_cast<const int*>(a.w);
  const int skip_raw = *skp;
  const int idx_raw = *idp;
  const int fr_s = (a.fr_base ? *a.fr_base : 0) + a.fr_off;
  const bool epi_wave = wave < NT;
  const int edu = lane >> 4;  // epilogue unit (or row quad) within the tile

  // ---- fragments of one batch (UB k-blocks of this wave): UB*(1+NT) float4 loads, no waits ----
  struct Frag { float4 a[UB]; float4 b[UB][NT]; int part[UB]; };
  auto issue = [&](Frag& f, int kb_base) {
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      int kb = kb_base + ub * NW;
      const bool valid = kb < a.nkb_total;
      if (!valid) kb = a.nkb_total - 1;
      const float* sp = d.segp[0];
      int ld = d.segld[0], local = kb, pt = d.segpart[0];
#pragma unroll
      for (int t = 1; t < 4; ++t) {
        const bool in = kb >= d.segstart[t];
        sp = in ? d.segp[t] : sp;
        ld = in ? d.segld[t] : ld;
        local = in ? kb - d.segstart[t] : local;
        pt = in ? d.segpart[t] : pt;
      }
      f.part[ub] = valid ? pt : 2;  // 2 = padding block: contributes nothing
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        f.b[ub][nt] = *reinterpret_cast<const float4*>(sp + (size_t)ncol[nt] * ld + local * 16 + kq * 4);
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
        if (NPART == 2 && f.part[ub] == 1) {
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].x, f.b[ub][nt].x, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].y, f.b[ub][nt].y, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].z, f.b[ub][nt].z, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].w, f.b[ub][nt].w, accH[nt], 0, 0, 0);
        } else {
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].x, f.b[ub][nt].x, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].y, f.b[ub][nt].y, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].z, f.b[ub][nt].z, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[ub].w, f.b[ub][nt].w, accX[nt], 0, 0, 0);
        }
      }
    }
  };
  Frag f0, f1;
  issue(f0, wave);
  const int skip = a.skip_flag ? skip_raw : 0;
  int prow = a.pre_idx ? idx_raw : a.pre_base_row + en * a.pre_n_stride;
  if (a.fr_base) {
    const long long pos = (long long)(a.fr_n_off + en) * a.fr_fold_stride + fr_s;
    prow = pos < a.fr_total_len ? (int)(pos / a.fr_hop) : a.fr_frames;
  }

  // ---- epilogue operands: unconditional loads (absent tensors read a.w[0] and are masked out by a
  //      select), issued by every wave right behind the first batch and consumed only after the
  //      reduction barrier, so the MFMA chain never waits on them ----
  const int H = a.units;
  int ej = mt * 4 + edu;             // GRU/LSTM hidden unit of this lane
  if (ej >= H) ej = H - 1;
  const int erow = mt * 16 + edu * 4;  // LINEAR first row of this lane's quad
  float l_bx[4], l_pre[4], l_bh[4], l_hp = 0.f, l_cp = 0.f, l_xr = 0.f;
  {
    const float* bxp = a.biasX ? a.biasX : a.w;
    const float* bhp = a.biasH ? a.biasH : a.w;
    const float* prp = a.pre_table ? a.pre_table + (size_t)prow * a.pre_stride : a.w;
    const float* mkp = a.mask ? a.mask + (size_t)en * a.ldy : a.w;
    const size_t so = (size_t)en * H + ej;
    if (EPI == EPI_LINEAR) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = erow + r < H ? erow + r : H - 1;
        l_bx[r] = bxp[a.biasX ? row : 0];
        l_pre[r] = prp[a.pre_table ? row : 0];
        l_bh[r] = mkp[a.mask ? row : 0];  // LINEAR: the dropout mask rides in l_bh
      }
    } else {
#pragma unroll
      for (int g = 0; g < RL; ++g) {
        l_bx[g] = bxp[a.biasX ? g * H + ej : 0];
        l_pre[g] = prp[a.pre_table ? g * H + ej : 0];
        l_bh[g] = bhp[a.biasH ? g * H + ej : 0];
      }
      if (RL == 3) { l_bx[3] = 0.f; l_pre[3] = 0.f; l_bh[3] = 0.f; }
      if (EPI == EPI_GRU) l_hp = a.h_prev[so];
      if (EPI == EPI_LSTM) l_cp = a.c_prev[so];
      l_xr = (a.x_res ? a.x_res : a.w)[a.x_res ? so : 0];
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
  float e_bx[4], e_bh[4], e_pre[4], e_mask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    e_bx[r] = a.biasX ? l_bx[r] : 0.f;
    e_pre[r] = a.pre_table ? l_pre[r] : 0.f;
    e_bh[r] = (EPI != EPI_LINEAR && a.biasH) ? l_bh[r] : 0.f;
    e_mask[r] = (EPI == EPI_LINEAR && a.mask) ? l_bh[r] : 1.f;
  }
  const float e_hp = l_hp, e_cp = l_cp, e_xr = a.x_res ? l_xr : 0.f;
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
  const int n = en_raw, du = edu;
  if (n >= a.N) return;

  if (EPI == EPI_LINEAR) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mt * 16 + du * 4 + r;
      if (row < a.units) {
        float v = sx[r] + (e_bx[r] + e_pre[r]);
        if (a.act == 1) v = fmaxf(v, 0.f);
        else if (a.act == 2) v = sigmoidf_(v);
        else if (a.act == 3) v = tanhf(v);
        if (a.mask) v = v * (e_mask[r] * a.mask_scale);
        else if (a.drop_on) {
          uint32_t rr[4];
          philox4x32((uint32_t)a.drop_iter, (uint32_t)a.drop_layer, (uint32_t)n, (uint32_t)(row >> 2),
                     (uint32_t)a.drop_seed, (uint32_t)(a.drop_seed >> 32), rr);
          v = v * ((rr[row & 3] & 0x80000000u) ? a.mask_scale : 0.f);
        }
        a.y[(size_t)n * a.ldy + row] = v;
      }
    }
    MB_MARK(a.trace, 6, 0);
    trace_end(a.trace);
    return;
  }
  const int j = mt * 4 + du;  // hidden unit
  if (j >= a.units) return;
  const size_t so = (size_t)n * H + j;
  if (EPI == EPI_GRU) {
    // torch GRUCell (gate order r,z,n): r = s(i_r+h_r), z = s(i_z+h_z), n = tanh(i_n + r*h_n),
    // h' = n + z*(h - n).   models/vocoder/wavernn/models/fatchord_version.py:196-200,265-271;
    // models/synthesizer/models/tacotron.py:60,98
    const float rg = sigmoidf_((sx[0] + (e_bx[0] + e_pre[0])) + (sh[0] + e_bh[0]));
    const float zg = sigmoidf_((sx[1] + (e_bx[1] + e_pre[1])) + (sh[1] + e_bh[1]));
    const float ng = tanhf((sx[2] + (e_bx[2] + e_pre[2])) + rg * (sh[2] + e_bh[2]));
    const float hy = ng + zg * (e_hp - ng);
    a.h_out[so] = hy;
    if (a.x_out) a.x_out[so] = e_xr + hy;
    if (a.seq_out) a.seq_out[(long long)n * a.seq_n_stride + (long long)j * a.seq_j_stride + a.seq_off] = hy;
  } else {
    // torch LSTMCell (gate order i,f,g,o).  tacotron.py:62-63,112-125
    const float gi = sigmoidf_(sx[0] + (e_bx[0] + e_pre[0]) + e_bh[0]);
    const float gf = sigmoidf_(sx[1] + (e_bx[1] + e_pre[1]) + e_bh[1]);
    const float gg = tanhf(sx[2] + (e_bx[2] + e_pre[2]) + e_bh[2]);
    const float go = sigmoidf_(sx[3] + (e_bx[3] + e_pre[3]) + e_bh[3]);
    const float cy = gf * e_cp + gi * gg;
    const float hy = go * tanhf(cy);
    a.c_out[so] = cy;
    a.h_out[so] = hy;
    if (a.x_out) a.x_out[so] = e_xr + hy;
  }
  MB_MARK(a.trace, 6, 0);
  trace_end(a.trace);
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

int rnn_launch(int epi, const RnnK& k, hipStream_t s) {
  MB_REQUIRE(k.N >= 1 && k.units >= 1 && k.nseg >= 1 && k.nseg <= 4, "rnn_launch: bad shape");
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
  d.k = k;
  int start = 0;
  for (int t = 0; t < 4; ++t) {
    if (t < k.nseg) {
      d.segp[t] = k.seg[t].p; d.segld[t] = k.seg[t].ld; d.segpart[t] = k.seg[t].part; d.segstart[t] = start;
      start += k.seg[t].nkb;
    } else {
      d.segp[t] = k.seg[0].p; d.segld[t] = 0; d.segpart[t] = 0; d.segstart[t] = 0x7fffffff;
    }
  }
  MB_REQUIRE(start == k.nkb_total, "rnn_launch: segments cover %d k-blocks, nkb_total=%d", start, k.nkb_total);
  const int per_wave = cdiv(k.nkb_total, NW);
  const int ub = nt == 2 ? (per_wave >= 4 ? 4 : 2) : (per_wave >= 8 ? 8 : (per_wave >= 4 ? 4 : 2));
  dim3 grid(n_mt, cdiv(k.N, 16 * nt));
#define MB_RNN(EPI_, NT_, UB_) hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_, NT_, UB_>), grid, dim3(NW * 64), 0, s, d)
#define MB_RNN_E(EPI_)                                                                    \
  do {                                                                                    \
    if (nt == 2) { if (ub == 4) MB_RNN(EPI_, 2, 4); else MB_RNN(EPI_, 2, 2); }            \
    else if (ub == 8) MB_RNN(EPI_, 1, 8);                                                 \
    else if (ub == 4) MB_RNN(EPI_, 1, 4);                                                 \
    else MB_RNN(EPI_, 1, 2);                                                              \
  } while (0)
  if (epi == EPI_LINEAR) MB_RNN_E(EPI_LINEAR);
  else if (epi == EPI_GRU) MB_RNN_E(EPI_GRU);
  else MB_RNN_E(EPI_LSTM);
#undef MB_RNN_E
#undef MB_RNN
  MB_HIP(hipGetLastError());
  return MB_OK;
}

}  // namespace mb
