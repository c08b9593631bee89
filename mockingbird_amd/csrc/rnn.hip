// Row-tile MFMA recurrent GEMM kernels (see rnn.h for the design).
#include "rnn.h"

namespace mb {

template <int EPI, int NW, int NT>
__global__ __launch_bounds__(NW * 64) void rnn_rowtile_kernel(RnnK a) {
  constexpr int RL = (EPI == EPI_GRU) ? 3 : 4;
  constexpr int BLK = 4 * RL * 16;  // floats per (tile, k-block)
  constexpr int NPART = (EPI == EPI_GRU) ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float red[NW * NPART * NT * 256];

  if (a.skip_flag && *a.skip_flag) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mt = blockIdx.x, ntile0 = blockIdx.y * NT;  // this workgroup covers NT column tiles: the
  const int i = lane & 15, kq = lane >> 4;              // weight fragment is fetched once for all of them
  const int u = i >> 2, tau = i & 3;
  const bool live = tau < RL;
  const float* wl = a.w + (size_t)mt * a.nkb_total * BLK + ((u * RL + tau) * 4 + kq) * 4;
  int ncol[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    ncol[nt] = (ntile0 + nt) * 16 + i;
    if (ncol[nt] >= a.N) ncol[nt] = a.N - 1;  // duplicate a live column; its result is never stored
  }
  if (a.step_counter && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.step_counter += 1;

  // Epilogue operands (biases, table row, previous state) are fetched by the epilogue waves
  // (wave w < NT owns column tile w) BEFORE the GEMM so their latency hides under it.
  const int en = (ntile0 + (wave < NT ? wave : 0)) * 16 + (lane & 15);  // epilogue column of this lane
  const int edu = lane >> 4;                                            // epilogue unit within the tile
  float e_bx[4] = {0.f, 0.f, 0.f, 0.f}, e_bh[4] = {0.f, 0.f, 0.f, 0.f};
  float e_hp = 0.f, e_cp = 0.f, e_xr = 0.f, e_mask[4] = {1.f, 1.f, 1.f, 1.f};
  if (wave < NT && en < a.N) {
    const int prow = a.pre_idx ? a.pre_idx[en] : a.pre_base_row + en * a.pre_n_stride;
    const float* pre = a.pre_table ? a.pre_table + (size_t)prow * a.pre_stride : nullptr;
    if (EPI == EPI_LINEAR) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + edu * 4 + r;
        if (row < a.units) {
          if (a.biasX) e_bx[r] += a.biasX[row];
          if (pre) e_bx[r] += pre[row];
          if (a.mask) e_mask[r] = a.mask[(size_t)en * a.ldy + row] * a.mask_scale;
          else if (a.drop_on) {
            uint32_t rr[4];
            philox4x32((uint32_t)a.drop_iter, (uint32_t)a.drop_layer, (uint32_t)en, (uint32_t)(row >> 2),
                       (uint32_t)a.drop_seed, (uint32_t)(a.drop_seed >> 32), rr);
            e_mask[r] = (rr[row & 3] & 0x80000000u) ? a.mask_scale : 0.f;
          }
        }
      }
    } else {
      const int j = mt * 4 + edu;
      if (j < a.units) {
        const int H = a.units;
#pragma unroll
        for (int g = 0; g < RL; ++g) {
          if (a.biasX) e_bx[g] += a.biasX[g * H + j];
          if (pre) e_bx[g] += pre[g * H + j];
          if (a.biasH) e_bh[g] += a.biasH[g * H + j];
        }
        const size_t so = (size_t)en * H + j;
        if (EPI == EPI_GRU) e_hp = a.h_prev[so];
        if (EPI == EPI_LSTM) e_cp = a.c_prev[so];
        if (a.x_res) e_xr = a.x_res[so];
      }
    }
  }

  f32x4 accX[NT], accH[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { accX[nt] = {0.f, 0.f, 0.f, 0.f}; accH[nt] = {0.f, 0.f, 0.f, 0.f}; }
  // This wave owns k-blocks wave, wave+NW, ... of the concatenated K.  They are walked UB at a
  // time with ALL fragment loads of a batch issued before the first MFMA, so a wave pays one
  // memory round trip per batch instead of one per block (the loop is latency-, not FLOP-bound).
  constexpr int UB = (NT == 1) ? 8 : 4;
  for (int kb_base = wave; kb_base < a.nkb_total; kb_base += NW * UB) {
    float4 av[UB], bv[UB][NT];
    int part[UB];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
      const int kb = kb_base + ub * NW;
      av[ub] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bv[ub][nt] = make_float4(0.f, 0.f, 0.f, 0.f);
      part[ub] = 0;
      if (kb < a.nkb_total) {
        int local = kb, sgi = 0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (sgi < a.nseg - 1 && local >= a.seg[sgi].nkb) { local -= a.seg[sgi].nkb; ++sgi; }
        const RnnSeg sg = a.seg[sgi];
        part[ub] = sg.part;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          bv[ub][nt] = *reinterpret_cast<const float4*>(sg.p + (size_t)ncol[nt] * sg.ld + local * 16 + kq * 4);
        if (live) av[ub] = *reinterpret_cast<const float4*>(wl + (size_t)kb * BLK);
      }
    }
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (NPART == 2 && part[ub] == 1) {
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].x, bv[ub][nt].x, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].y, bv[ub][nt].y, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].z, bv[ub][nt].z, accH[nt], 0, 0, 0);
          accH[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].w, bv[ub][nt].w, accH[nt], 0, 0, 0);
        } else {
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].x, bv[ub][nt].x, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].y, bv[ub][nt].y, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].z, bv[ub][nt].z, accX[nt], 0, 0, 0);
          accX[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ub].w, bv[ub][nt].w, accX[nt], 0, 0, 0);
        }
      }
    }
  }
  // cross-wave reduction through LDS: D fragment lane = (unit = lane>>4, col = lane&15), reg = gate
  float4* red4 = reinterpret_cast<float4*>(red);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    red4[((wave * NT + nt) * NPART + 0) * 64 + lane] = make_float4(accX[nt][0], accX[nt][1], accX[nt][2], accX[nt][3]);
    if (NPART == 2)
      red4[((wave * NT + nt) * NPART + 1) * 64 + lane] = make_float4(accH[nt][0], accH[nt][1], accH[nt][2], accH[nt][3]);
  }
  __syncthreads();
  if (wave >= NT) return;
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
  const int n = en, du = edu;
  if (n >= a.N) return;

  if (EPI == EPI_LINEAR) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mt * 16 + du * 4 + r;
      if (row < a.units) {
        float v = sx[r] + e_bx[r];
        if (a.act == 1) v = fmaxf(v, 0.f);
        else if (a.act == 2) v = sigmoidf_(v);
        else if (a.act == 3) v = tanhf(v);
        if (a.mask || a.drop_on) v = v * e_mask[r];
        a.y[(size_t)n * a.ldy + row] = v;
      }
    }
    return;
  }
  const int j = mt * 4 + du;  // hidden unit
  if (j >= a.units) return;
  const int H = a.units;
  const size_t so = (size_t)n * H + j;
  if (EPI == EPI_GRU) {
    // torch GRUCell (gate order r,z,n): r = s(i_r+h_r), z = s(i_z+h_z), n = tanh(i_n + r*h_n),
    // h' = n + z*(h - n).   models/vocoder/wavernn/models/fatchord_version.py:196-200,265-271;
    // models/synthesizer/models/tacotron.py:60,98
    const float rg = sigmoidf_((sx[0] + e_bx[0]) + (sh[0] + e_bh[0]));
    const float zg = sigmoidf_((sx[1] + e_bx[1]) + (sh[1] + e_bh[1]));
    const float ng = tanhf((sx[2] + e_bx[2]) + rg * (sh[2] + e_bh[2]));
    const float hy = ng + zg * (e_hp - ng);
    a.h_out[so] = hy;
    if (a.x_out) a.x_out[so] = e_xr + hy;
    if (a.seq_out) a.seq_out[(long long)n * a.seq_n_stride + (long long)j * a.seq_j_stride + a.seq_off] = hy;
  } else {
    // torch LSTMCell (gate order i,f,g,o).  tacotron.py:62-63,112-125
    const float gi = sigmoidf_(sx[0] + e_bx[0] + e_bh[0]);
    const float gf = sigmoidf_(sx[1] + e_bx[1] + e_bh[1]);
    const float gg = tanhf(sx[2] + e_bx[2] + e_bh[2]);
    const float go = sigmoidf_(sx[3] + e_bx[3] + e_bh[3]);
    const float cy = gf * e_cp + gi * gg;
    const float hy = go * tanhf(cy);
    a.c_out[so] = cy;
    a.h_out[so] = hy;
    if (a.x_out) a.x_out[so] = e_xr + hy;
  }
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
  // faster with twice the workgroups (measured: 45 vs 51 us per WaveRNN step).
  const int rl = (epi == EPI_GRU) ? 3 : 4;
  const size_t wbytes = (size_t)n_mt * k.nkb_total * 4 * rl * 16 * sizeof(float);
  const int nt = (k.N > 16 && wbytes >= ((size_t)12 << 20)) ? 2 : 1;
  dim3 grid(n_mt, cdiv(k.N, 16 * nt));
#define MB_RNN(EPI_, NT_) hipLaunchKernelGGL((rnn_rowtile_kernel<EPI_, NW, NT_>), grid, dim3(NW * 64), 0, s, k)
  if (epi == EPI_LINEAR) { if (nt == 2) MB_RNN(EPI_LINEAR, 2); else MB_RNN(EPI_LINEAR, 1); }
  else if (epi == EPI_GRU) { if (nt == 2) MB_RNN(EPI_GRU, 2); else MB_RNN(EPI_GRU, 1); }
  else { if (nt == 2) MB_RNN(EPI_LSTM, 2); else MB_RNN(EPI_LSTM, 1); }
#undef MB_RNN
  MB_HIP(hipGetLastError());
  return MB_OK;
}

}  // namespace mb
