// Removed from libmbhip.so in round 4: the round-2 persistent WaveRNN kernel for 2..4 fold columns with MFMA tiles
// (every on-chain workgroup recomputing the whole rnn1 finish), and the MFMA form of the one-column kernel (MBHIP_WP_MFMA=1: 13.5 against
// 10.0 us per step for the fmaf form).  Superseded by wavernn_pipe.h / wavernn_pipe16.h for 2..64 columns (round 3: 13.1 against 15.3 us
// per step at 3 columns).  It was a member of csrc/wavernn_persist.h (its helpers -- wp_watch, wp_gather, wp_gemm* -- stay there).  Not built.
// dynamic LDS (floats): [weights: 12288 rnn2 (2 GRU tiles) | 8192 fc a | 8192 fc b] [red 4096 (two-tile pass) + 2 x 2048 (single-tile GEMMs, alternating)] [x1s 128 x NCOL x 4] [keys]
constexpr int WP_LDS_W = 12288 + 8192 + 8192, WP_LDS_RED = 2 * 4096, WP_LDS_X = 128 * WP_NCOL * 4;
constexpr size_t WP_LDS_BYTES = (size_t)(WP_LDS_W + WP_LDS_RED + WP_LDS_X) * 4 + 2 * WP_NCOL * 8 + 64;

__global__ __launch_bounds__(512) void wf_persist_kernel(WpK a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lw = lds;
  float* red = lds + WP_LDS_W;
  float* x1s = red + WP_LDS_RED;  // float4 index (k / 4) * NCOL + n
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(x1s + WP_LDS_X);  // [NCOL] max key of the step
  float* s_x = reinterpret_cast<float*>(s_key + WP_NCOL);                             // [NCOL] decoded sample
  if (__hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (tests: the fallback path)
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int du = lane >> 4, i = lane & 15;
  const int N = a.N, H = a.R, S = a.S;
  const int n_t3 = a.C / 16;  // fc3 row tiles
  int rb = 0;                 // red buffer of the next GEMM
  auto EX = [&](int what, unsigned tag) { return a.ex + (size_t)(tag & 1) * WPX_PER_PARITY + what; };

  if (g >= WP_ON) {
    // ------------------------------------------------------------------ off-chain: hidden halves of the next step
    const int mt = g - WP_ON;
    wp_copy_tile(lw, a.w_hh1 + (size_t)mt * 6144, 6144);
    wp_copy_tile(lw + 6144, a.w_hh2 + (size_t)mt * 6144, 6144);
    const float4 bq1 = a.bhh1q[mt * 4 + du], bq2 = a.bhh2q[mt * 4 + du];
    __syncthreads();
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        float4 b[4];
        if (s == 0) {
#pragma unroll
          for (int p = 0; p < 4; ++p) b[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (!wp_gather<6>(EX(which ? WPX_H2 : WPX_H1, (unsigned)s), (unsigned)s, N, b, a.abort_word)) return;
        float sx[4];
        const bool epi = wp_gemm<3>(lw + which * 6144, b, red + 4096 + rb * 2048, sx);
        rb ^= 1;
        if (epi && i < N) {
          const float4 bq = which ? bq2 : bq1;
          unsigned long long* P = EX(which ? WPX_P2 : WPX_P1, (unsigned)s + 1) + (size_t)(mt * 4 + du) * 4 * N + i;  // dense [unit][gate][column]
          wp_put(P, sx[0] + bq.x, (unsigned)s + 1);
          wp_put(P + N, sx[1] + bq.y, (unsigned)s + 1);
          wp_put(P + 2 * N, sx[2] + bq.z, (unsigned)s + 1);
        }
      }
    }
    return;
  }

  // -------------------------------------------------------------------- on-chain
  const bool lo = g < 32;
  const int ft = lo ? g : g - 32;           // fc1 tile (lo) / fc2 + fc3 tile (hi)
  wp_copy_tile(lw, a.w_rnn2 + (size_t)(2 * g) * 6144, 12288);
  wp_copy_tile(lw + 12288, (lo ? a.w_fc1 : a.w_fc2) + (size_t)ft * 8192, 8192);
  if (!lo && ft < n_t3) wp_copy_tile(lw + 20480, a.w_fc3 + (size_t)ft * 8192, 8192);
  if (tid < WP_NCOL) { s_key[tid] = 0ull; s_x[tid] = 0.f; }
  // finish: thread j = unit j
  const int j = tid;
  const float gr = a.g1[j], gz = a.g1[H + j], gn = a.g1[2 * H + j], w0 = a.wI0[j];
  float h1[WP_NCOL], tq[WP_NCOL][4];
#pragma unroll
  for (int n = 0; n < WP_NCOL; ++n) {
    h1[n] = 0.f;
    if (n < N) {
      const float4 t4 = wf_cond_row4(a.cond, wf_pos(a.g, n, 0), (unsigned)a.g.total_len, j, H, a.g.frames);
      tq[n][0] = t4.x; tq[n][1] = t4.y; tq[n][2] = t4.z; tq[n][3] = t4.w;
    }
  }
  float h2 = 0.f;  // waves 0 / 1: unit (2g + wave) * 4 + du, column i
  float g2r = 0.f, g2z = 0.f, g2n = 0.f; int g2_row = -1;   // cached per-frame rows (they change once per hop)
  float4 fpre = make_float4(0.f, 0.f, 0.f, 0.f); int f_row = -1;
  const float4 b3q = (!lo && ft < n_t3) ? *reinterpret_cast<const float4*>(a.b_fc3 + ft * 16 + du * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

#define WP_MARK(k)                                                                                          \
  do {                                                                                                      \
    if (a.trace && tid == 0 && (g == 0 || g == 32) && s >= 1000 && s < 1004)                                \
      a.trace[((g ? 1 : 0) * 4 + (s - 1000)) * 16 + (k)] = (unsigned long long)wall_clock64();              \
  } while (0)
  constexpr int WP_PRE = 2;  // columns whose hidden-half granules are requested before the key wait
  for (int s = 0; s <= S; ++s) {
    const unsigned tag_prev = (unsigned)s, tag = (unsigned)s + 1;
    // the hidden halves of this step were published a step ago: request them now, in the shadow of the key wait
    unsigned long long p1v[WP_PRE][3], p2v[3];
    if (s < S) {
#pragma unroll
      for (int n = 0; n < WP_PRE; ++n)
        if (n < N) wp_issue<3>(EX(WPX_P1, tag) + (size_t)j * 4 * N + n, N, p1v[n]);
    }
    const int ncl = i < N ? i : N - 1;  // clamped column: loads legal, nothing published for dead columns
    const int frow = s < S ? wf_frame_row(a.g, ncl, s) : 0;
    WP_MARK(0);
    // ---- A: keys of step s-1 -> sample x (every workgroup for itself) ----
    if (s > 0) {
      wp_watch<1>(EX(WPX_KEY, tag_prev) + (size_t)((n_t3 - 1) * 2 + 1) * N + (N - 1), tag_prev, a.abort_word);
      if (tid < 32 * N && (tid & 31) < n_t3) {
        const int tile = tid & 31, n = tid >> 5;
        unsigned kv[2];
        if (!wp_wait<2>(EX(WPX_KEY, tag_prev) + (size_t)tile * 2 * N + n, N, tag_prev, kv, a.abort_word)) return;
        atomicMax(&s_key[n], ((unsigned long long)kv[0] << 32) | (unsigned long long)kv[1]);
      }
      WP_MARK(1);
      __syncthreads();
      WP_MARK(2);
      if (tid < N) {
        const unsigned long long slot = s_key[tid];
        const float x = slot ? 2.f * (float)argmax_class(slot) / ((float)a.C - 1.f) - 1.f : 0.f;
        s_x[tid] = x;
        if (g == 0) {
          a.samples[(size_t)tid * S + (s - 1)] = x;
          if (a.progress && tid == 0 && (s - 1) % 100 == 0) *a.progress = s;
        }
      }
      __syncthreads();
      if (tid < N) s_key[tid] = 0ull;  // next atomicMax is several barriers away
    }
    if (s == S) break;
    // rnn2's hidden half (published ~3 us after rnn2 of the previous step): requested here, used after the GEMM
    if (wave < 2 && i < N) wp_issue<3>(EX(WPX_P2, tag) + (size_t)((2 * g + wave) * 4 + du) * 4 * N + i, N, p2v);
    WP_MARK(3);
    // ---- B: rnn1 finish for unit j, all columns (wf_finish_kernel's expressions) ----
#pragma unroll
    for (int n = 0; n < WP_NCOL; ++n) {
      if (n >= N) continue;
      unsigned pu[3];
      if (n < WP_PRE) { if (!wp_take<3>(EX(WPX_P1, tag) + (size_t)j * 4 * N + n, N, tag, p1v[n], pu, a.abort_word)) return; }
      else if (!wp_wait<3>(EX(WPX_P1, tag) + (size_t)j * 4 * N + n, N, tag, pu, a.abort_word)) return;
      const float hqx = __uint_as_float(pu[0]), hqy = __uint_as_float(pu[1]), hqz = __uint_as_float(pu[2]);
      const float x = s_x[n];
      const float rg = sigmoidf_((tq[n][0] + x * gr) + hqx);
      const float zg = sigmoidf_((tq[n][1] + x * gz) + hqy);
      const float ng = tanhf((tq[n][2] + x * gn) + rg * hqz);
      const float hy = ng + zg * (h1[n] - ng);
      h1[n] = hy;
      x1s[((j >> 2) * WP_NCOL + n) * 4 + (j & 3)] = (tq[n][3] + x * w0) + hy;
      if ((j >> 3) == g) wp_put(EX(WPX_H1, tag) + (size_t)j * N + n, hy, tag);  // every workgroup has all of h1: each publishes 8 units
    }
    WP_MARK(4);
    __syncthreads();
    WP_MARK(5);
    // ---- C: next step's table rows (a whole step of latency to hide behind) ----
    if (s + 1 < S) {
#pragma unroll
      for (int n = 0; n < WP_NCOL; ++n) {
        if (n >= N) continue;
        const float4 t4 = wf_cond_row4(a.cond, wf_pos(a.g, n, s + 1), (unsigned)a.g.total_len, j, H, a.g.frames);
        tq[n][0] = t4.x; tq[n][1] = t4.y; tq[n][2] = t4.z; tq[n][3] = t4.w;
      }
    }
    // ---- D: rnn2, row tiles 2g and 2g+1 in one pass; wave tt finishes tile tt ----
    {
      float4 b[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        b[p] = i < WP_NCOL ? *reinterpret_cast<const float4*>(x1s + (((wave + 8 * p) * 4 + (lane >> 4)) * WP_NCOL + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      float sx[4];
      const bool epi = wp_gemm2<3>(lw, 6144, b, red, sx);
      WP_MARK(11);
      if (epi && i < N) {
        const int ju = (2 * g + wave) * 4 + du;
        if (frow != g2_row) {  // the per-frame rows change once per hop: kept in registers in between
          const float* gp = a.G2 + (size_t)frow * 3 * H + ju;
          g2r = gp[0]; g2z = gp[H]; g2n = gp[2 * H];
          g2_row = frow;
        }
        unsigned pu[3];
        if (!wp_take<3>(EX(WPX_P2, tag) + (size_t)ju * 4 * N + i, N, tag, p2v, pu, a.abort_word)) return;
        WP_MARK(12);
        const float xr = x1s[((ju >> 2) * WP_NCOL + i) * 4 + (ju & 3)];
        const float rg = sigmoidf_((sx[0] + g2r) + __uint_as_float(pu[0]));
        const float zg = sigmoidf_((sx[1] + g2z) + __uint_as_float(pu[1]));
        const float ng = tanhf((sx[2] + g2n) + rg * __uint_as_float(pu[2]));
        const float hy = ng + zg * (h2 - ng);
        h2 = hy;
        wp_put(EX(WPX_X2, tag) + (size_t)ju * N + i, xr + hy, tag);
        wp_put(EX(WPX_H2, tag) + (size_t)ju * N + i, hy, tag);
      }
    }
    WP_MARK(6);
    // ---- E: fc1 (lo) | fc2 then fc3 (hi) ----
    {
      if (wave == 0 && frow != f_row) {
        fpre = *reinterpret_cast<const float4*>((lo ? a.F1 : a.F2) + (size_t)frow * a.FC + ft * 16 + du * 4);
        f_row = frow;
      }
      const float4 pre = fpre;
      float4 b[4];
      if (!wp_gather<1>(EX(lo ? WPX_X2 : WPX_Y1, tag), tag, N, b, a.abort_word)) return;
      WP_MARK(7);
      float sx[4];
      const bool epi = wp_gemm<4>(lw + 12288, b, red + 4096 + rb * 2048, sx);
      rb ^= 1;
      WP_MARK(8);
      if (epi && i < N) {
        unsigned long long* Y = EX(lo ? WPX_Y1 : WPX_Y2, tag) + (size_t)(ft * 16 + du * 4) * N + i;
        wp_put(Y, fmaxf(sx[0] + pre.x, 0.f), tag);
        wp_put(Y + N, fmaxf(sx[1] + pre.y, 0.f), tag);
        wp_put(Y + 2 * N, fmaxf(sx[2] + pre.z, 0.f), tag);
        wp_put(Y + 3 * N, fmaxf(sx[3] + pre.w, 0.f), tag);
      }
    }
    if (!lo && ft < n_t3) {
      float4 b[4];
      if (!wp_gather<1>(EX(WPX_Y2, tag), tag, N, b, a.abort_word)) return;
      WP_MARK(9);
      float sx[4];
      const bool epi = wp_gemm<4>(lw + 20480, b, red + 4096 + rb * 2048, sx);
      rb ^= 1;
      WP_MARK(10);
      if (epi) {  // wf_fc3_kernel's sampler; lanes of dead columns take part in the shuffles only
        uint32_t grn[4];
        philox4x32((uint32_t)s, (uint32_t)ncl, (uint32_t)((ft * 16 + du * 4) >> 2), 0x57415645u, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), grn);
        const float bv[4] = {b3q.x, b3q.y, b3q.z, b3q.w};
        float best = -INFINITY;
        int bcls = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = ft * 16 + du * 4 + r;
          const float v = sx[r] + bv[r];
          const float gmb = v - logf(-logf(u32_to_unit(grn[r])));
          if (gmb > best) { best = gmb; bcls = row; }
        }
        unsigned long long pk = pack_argmax(best, bcls);
        const unsigned long long o1 = __shfl_xor(pk, 16, 64);
        pk = o1 > pk ? o1 : pk;
        const unsigned long long o2 = __shfl_xor(pk, 32, 64);
        pk = o2 > pk ? o2 : pk;
        if (du == 0 && i < N) {
          unsigned long long* K = EX(WPX_KEY, tag) + (size_t)ft * 2 * N + i;
          wp_put_u(K, (unsigned)(pk >> 32), tag);
          wp_put_u(K + N, (unsigned)pk, tag);
        }
      }
    }
  }
}

