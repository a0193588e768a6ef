// DCA feature builder (reference predict.py:41-61): weighted shrunk covariance of the
// one-hot alignment, dense SPD inverse, APC-corrected contact map.
#include "common.h"

namespace dmp {

// ---------------------------------------------------------------------------------------
// covariance
// ---------------------------------------------------------------------------------------
// sc[0] = S = sum(w); sc[1] = num_points = S - sqrt(mean(w))          (predict.py:45)
__global__ __launch_bounds__(256) void wsum_kernel(const float* __restrict__ w, int N,
                                                   float* __restrict__ sc) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) acc += (double)w[n];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float S = (float)red[0];
    const float mean = S / (float)N;
    sc[0] = S;
    sc[1] = S - sqrtf(mean);
  }
}

// mean[l*21+a] = sum_n w_n [code_nl == a] / num_points                 (predict.py:47)
__global__ __launch_bounds__(256) void colmean_kernel(const uint8_t* __restrict__ msa,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ sc, int N, int L,
                                                      float* __restrict__ mean) {
  __shared__ double bins[NS][257];
  const int l = blockIdx.x;
  double acc[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) acc[a] = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) {
    int c = msa[(int64_t)n * L + l];
    c = c > 20 ? 20 : c;
    const double wn = (double)w[n];
#pragma unroll
    for (int a = 0; a < NS; ++a) acc[a] += (c == a) ? wn : 0.0;
  }
#pragma unroll
  for (int a = 0; a < NS; ++a) bins[a][threadIdx.x] = acc[a];
  __syncthreads();
  if (threadIdx.x < NS) {
    double s = 0.0;
    for (int t = 0; t < 256; ++t) s += bins[threadIdx.x][t];
    mean[l * NS + threadIdx.x] = (float)s / sc[1];
  }
}

// xc[n][d] = (onehot - mean[d]) * sqrt(w_n)                             (predict.py:48)
__global__ __launch_bounds__(256) void center_kernel(const uint8_t* __restrict__ msa,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ mean, int N, int L,
                                                     float* __restrict__ xc) {
  const int n = blockIdx.y;
  const int D = L * NS;
  const float sw = sqrtf(w[n]);
  for (int d = blockIdx.x * 256 + threadIdx.x; d < D; d += gridDim.x * 256) {
    const int l = d / NS, a = d - l * NS;
    int c = msa[(int64_t)n * L + l];
    c = c > 20 ? 20 : c;
    const float x = (c == a) ? 1.0f : 0.0f;
    xc[(int64_t)n * D + d] = (x - mean[d]) * sw;
  }
}

// cov = G / num_points + I * (4.5 / sqrt(S))                            (predict.py:50-51)
__global__ __launch_bounds__(256) void cov_finish_kernel(float* __restrict__ cov, int D,
                                                         const float* __restrict__ sc, bool lower_only) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)D * D) return;
  const int i = idx / D, j = idx - (int64_t)i * D;
  if (lower_only && (j >> 7) > (i >> 7)) return;
  float v = cov[idx] / sc[1];
  if (i == j) v += 4.5f / sqrtf(sc[0]);
  cov[idx] = v;
}

// lower_only (the prediction path, where the in-place inverse is the only reader): the product's 128 x 128 tiles above
// the diagonal are neither computed nor scaled - spd_inverse_steps reads diagonal tiles and the lower triangle alone and
// rebuilds the upper one at its end.  2.07 -> 1.1 ms at N = 2000, D = 6300.  (A tile below the diagonal holds the
// bits its mirror image would: the products commute, the k order is the same.)
int cov_build(dmp_ctx* c, const uint8_t* d_msa, const float* d_w, int N, int L, float* d_cov,
              hipStream_t s, bool lower_only) {
  const int D = L * NS;
  float* sc = (float*)c->wsum;
  hipLaunchKernelGGL(wsum_kernel, dim3(1), dim3(256), 0, s, d_w, N, sc);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(colmean_kernel, dim3(L), dim3(256), 0, s, d_msa, d_w, sc, N, L, c->colmean);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(center_kernel, dim3(std::min(cdiv(D, 256), 64), N), dim3(256), 0, s, d_msa, d_w,
                     c->colmean, N, L, c->xc);
  DMP_LAUNCH_CHECK();
  GemmArgs g{};
  g.A = c->xc; g.sam = 1; g.sak = D;
  g.B = c->xc; g.sbk = D; g.sbn = 1;
  g.C = d_cov; g.ldc = D;
  g.M = D; g.N = D; g.K = N;
  g.alpha = 1.f; g.beta = 0.f; g.bias_n = nullptr;
  g.lower_tiles = lower_only;
  int rc = gemm_f32(g, s);
  if (rc) return rc;
  hipLaunchKernelGGL(cov_finish_kernel, dim3((unsigned)cdiv64((int64_t)D * D, 256)), dim3(256), 0,
                     s, d_cov, D, sc, lower_only);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// SPD inverse: in-place blocked Gauss-Jordan (no pivoting; the ridge keeps the matrix SPD)
//   per block k:  P = inv(A_kk);  R = P A_k,:  (columns of block k zeroed);
//                 C = A_:,k       (rows of block k zeroed);
//                 A -= C R;  A_:,k = -C P;  A_k,: = R;  A_kk = P
// ---------------------------------------------------------------------------------------
// NH = row groups of the block (threads = 128 NH): thread (j, ih) keeps rows ih*RH .. ih*RH + RH - 1 of column j in
// RH = 128 / NH registers.  Every element sees the same operations whatever NH (same bits); four groups (512 threads,
// 32 registers, two waves per SIMD) halve a thread's share of a pivot step against the two of rounds 1-3.
template <int NH>
__global__ __launch_bounds__(128 * NH) void gj_diag_kernel(const float* __restrict__ A, int D, int k0,
                                                          int bs, float* __restrict__ P) {
  // Symmetric sweep operator (Goodnight): sweeping every pivot of an SPD block in place gives
  // -inverse and keeps the matrix symmetric, so a step only needs the pivot ROW broadcast.
  // (static register indexing: the pivot loop is unrolled over the row-in-group index); the pivot row goes through LDS.
  constexpr int RH = GJ_NB / NH;
  __shared__ float rowk[2][GJ_NB];
  const int tid = threadIdx.x;
  const int j = tid & 127, ih = tid >> 7;
  float sreg[RH];
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int i = ih * RH + r;
    sreg[r] = (i < bs && j < bs) ? A[(int64_t)(k0 + i) * D + k0 + j] : (i == j ? 1.f : 0.f);
  }
#pragma unroll 1
  for (int kh = 0; kh < NH; ++kh) {
#pragma unroll
    for (int r = 0; r < RH; ++r) {
      const int k = kh * RH + r;
      if (k < bs) {                                   // uniform
        float* rk = rowk[k & 1];
        if (ih == kh) rk[j] = sreg[r];                // publish row k (column j's element)
        __syncthreads();
        const float invd = 1.0f / rk[k];
        const bool jk = (j == k);
        const float bj = rk[j] * invd;                // new a_kj
        const float mul = jk ? -invd : bj;            // column k: a_ik/d ; elsewhere a_ij - a_ik*b_j
#pragma unroll
        for (int rr = 0; rr < RH; ++rr) {
          const float aik = rk[ih * RH + rr];         // a_ik = a_ki by symmetry
          sreg[rr] = fmaf(-aik, mul, jk ? 0.f : sreg[rr]);
        }
        if (ih == kh) sreg[r] = jk ? -invd : bj;      // the pivot row itself
      }
    }
  }
  // P = inverse = -swept matrix; padding rows/columns of a partial block become identity again
#pragma unroll
  for (int r = 0; r < RH; ++r) {
    const int i = ih * RH + r;
    P[i * GJ_NB + j] = (i < bs && j < bs) ? -sreg[r] : (i == j ? 1.f : 0.f);
  }
}

typedef float gj_f32x16 __attribute__((ext_vector_type(16)));
typedef float gj_f32x4 __attribute__((ext_vector_type(4)));

// Round 5: the same sweep, BLOCKED (option "gj_diag_blocked" = 1; the chain above stays the default, see the header).  The kernel above is a chain of 128 pivots,
// each a round trip barrier -> LDS -> division -> update through eight waves (0.41 us per pivot alone, 2.5 us beside
// two convolutions).  Sweeping a SET K of pivots at once is the block form of the same operator:
//     P = inv(M_KK);   M_KK <- -P;   M_RK <- M_RK P  (= W);   M_RR <- M_RR - W M_KR
// so the block is swept in 8 sub-blocks of 16: (1) the 16 x 16 pivot block by 16 scalar sweeps inside ONE wave (four
// elements per lane; the pivot row goes through 16 words of LDS that only this wave touches - no workgroup barrier
// inside the chain), (2) W = T P for the 128 rows on v_mfma_f32_16x16x4_f32 (one 16-row tile per wave), (3) the rank-16
// update of the 128 x 128 block on v_mfma_f32_32x32x2_f32 (two 32 x 32 tiles per wave), (4) the pivot columns / rows
// become W / W^T.  The matrix lives in LDS (pitch 129).  Identity padding of a partial last block is swept like real
// pivots (d = 1: the row stays e_k), so there is no special case.  Different order of operations from the scalar chain:
// the inverse agrees with it to float32 rounding, not bit for bit (tests: against the oracle's inverse at 1e-5).
#ifndef GD_FAST_RCP
#define GD_FAST_RCP 0
#endif
constexpr int GD_MP = GJ_NB + 1;                              // 129: pitch of the block in LDS
constexpr int GD_WP = 17;                                    // pitch of W [128][16] and of the pivot block's inverse [16][16]
constexpr int GJ_DIAGB_LDS = (GJ_NB * GD_MP + GJ_NB * GD_WP + 16 * GD_WP + 16) * 4;       // 77 904 bytes
__global__ __launch_bounds__(512) void gj_diag_blocked_kernel(const float* __restrict__ A, int D, int k0, int bs,
                                                              float* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) float gd_smem[];
  float* M = gd_smem;                                          // [128][129]
  float* Wl = M + GJ_NB * GD_MP;                               // [128][17]
  float* Pb = Wl + GJ_NB * GD_WP;                              // [16][17]
  float* rowk = Pb + 16 * GD_WP;                               // [16] (16-byte aligned: every size above is a multiple of 4 words)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    // all 32 loads of a thread in flight at once (as a loop the compiler issued them one by one, each behind a
    // vmcnt(0): 32 trips to memory = more than the whole sweep)
    float reg[GJ_NB * GJ_NB / 512];
#pragma unroll
    for (int it = 0; it < GJ_NB * GJ_NB / 512; ++it) {
      const int e = tid + 512 * it, i = e >> 7, j = e & 127;
      const bool in = i < bs && j < bs;
      const float v = A[(int64_t)(k0 + (in ? i : 0)) * D + k0 + (in ? j : 0)];
      reg[it] = in ? v : (i == j ? 1.f : 0.f);
    }
#pragma unroll
    for (int it = 0; it < GJ_NB * GJ_NB / 512; ++it) {
      const int e = tid + 512 * it;
      M[(e >> 7) * GD_MP + (e & 127)] = reg[it];
    }
  }
  __syncthreads();
  const int kk = lane >> 5, li = lane & 31;
  for (int b = 0; b < GJ_NB / 16; ++b) {
    const int kb = 16 * b;
    if (wave == 0) {
      // (1) the pivot block by 16 scalar sweeps (the operator of gj_diag_kernel): lane (c, rg) holds rows 4 rg .. 4 rg + 3 of
      // column c; row k is published through `rowk`, which only this wave reads and writes (LDS operations of one wave
      // complete in order; the wavefront-scope fences keep the compiler from moving them across each other)
      const int c = lane & 15, rg = lane >> 4;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = M[(kb + 4 * rg + i) * GD_MP + kb + c];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (rg == (k >> 2)) rowk[c] = v[k & 3];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float rc = rowk[c], d = rowk[k];
        const gj_f32x4 rr = *reinterpret_cast<const gj_f32x4*>(rowk + 4 * rg);      // a_ik = a_ki by symmetry
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#if GD_FAST_RCP
        float invd = __builtin_amdgcn_rcpf(d);                // 1 ulp, + one Newton step (the IEEE division is ten dependent
        invd = fmaf(fmaf(-d, invd, 1.0f), invd, invd);        // instructions on the critical path of every pivot: 2 us per sweep)
#else
        const float invd = 1.0f / d;
#endif
        const bool ck = c == k;
        const float bj = rc * invd;
        const float mul = ck ? -invd : bj;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaf(-rr[i], mul, ck ? 0.f : v[i]);
        if (rg == (k >> 2)) v[k & 3] = ck ? -invd : bj;       // the pivot row itself
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        M[(kb + 4 * rg + i) * GD_MP + kb + c] = v[i];          // M_KK <- -P
        Pb[(4 * rg + i) * GD_WP + c] = -v[i];
      }
    }
    __syncthreads();
    {
      // (2) W = T P, T = the pivot columns (rows inside K give values nobody uses); wave w: rows 16 w .. 16 w + 15
      const int c = lane & 15, kq = lane >> 4;
      gj_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(M[(16 * wave + c) * GD_MP + kb + 4 * s4 + kq], Pb[(4 * s4 + kq) * GD_WP + c],
                                                   acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Wl[(16 * wave + 4 * kq + r) * GD_WP + c] = acc[r];
    }
    __syncthreads();
    // (3) M_RR <- M_RR - W T^T, rows and columns outside K; wave w: the 32 x 32 tiles 2 w, 2 w + 1 of the 4 x 4
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 2 * wave + tt, i0 = 32 * (t >> 2), j0 = 32 * (t & 3);
      gj_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wl[(i0 + li) * GD_WP + 2 * s2 + kk], M[(j0 + li) * GD_MP + kb + 2 * s2 + kk],
                                                   acc, 0, 0, 0);
      // (rows / columns inside K keep their values: the subtraction of 0 rewrites what is there - T, which other waves
      // are still reading, and -P - with the same bits; unconditional stores instead of sixteen branches)
      const int col = j0 + li;
      const bool col_ok = (col >> 4) != b;
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) old[r] = M[(i0 + 8 * (r >> 2) + 4 * kk + (r & 3)) * GD_MP + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + 8 * (r >> 2) + 4 * kk + (r & 3);
        M[row * GD_MP + col] = (col_ok && (row >> 4) != b) ? old[r] - acc[r] : old[r];
      }
    }
    __syncthreads();
    // (4) the pivot columns and rows
#pragma unroll
    for (int it = 0; it < GJ_NB * 16 / 512; ++it) {
      const int e = tid + 512 * it, r = e >> 4, c = e & 15;
      if ((r >> 4) != b) {
        const float w = Wl[r * GD_WP + c];
        M[r * GD_MP + kb + c] = w;
        M[(kb + c) * GD_MP + r] = w;
      }
    }
    __syncthreads();
  }
  // P = inverse = -swept matrix; padding rows / columns of a partial block are the identity again
#pragma unroll 8
  for (int it = 0; it < GJ_NB * GJ_NB / 512; ++it) {
    const int e = tid + 512 * it, i = e >> 7, j = e & 127;
    P[e] = (i < bs && j < bs) ? -M[i * GD_MP + j] : (i == j ? 1.f : 0.f);
  }
}

// Everything of a block step between the diagonal block's inverse P and the trailing update, for one strip of 64
// matrix rows / columns per workgroup (round 4: four launches before - the refresh of the stale upper triangle, the
// row-panel GEMM, the panel re-layout and the write-back - each reading what the one before had just written):
//   * the step's "cross" is read from the LOWER triangle only (the upper one is stale between the first and the last
//     step, section comment of gj_trailing_kernel), with the sign rule M_ij = t_i t_j M_ji:
//       strip below the block:   C[i][c] = A[i][k0+c],               S[c][i] = A[k0+c][i] =  A[i][k0+c]
//       strip left of / above:   S[c][j] = A[k0+c][j] (lower),       C[j][c] = A[j][k0+c] = -A[k0+c][j]
//       strip inside the block:  C = 0,  the row panel there is P itself
//   * R[:, strip] = P S on the f32 matrix cores - the k pairing and order of gemm_kernel (lane half = k parity, k
//     ascending), which computed this product in rounds 1-3: the same bits
//   * CT / RT = the panels as k quads for the trailing kernel's 16-byte loads:
//       CT[k/4][i][4] = C[i][k] (zero for rows inside the block and for k >= bs),  RT[k/4][j][4] = R[k][j]
//   * A_k,: = R left of the block (the lower triangle's part of the row panel), A_kk = P
// grid: Dp / 64 strips   block: 256 (wave w = rows 32w .. 32w+31 of R)   LDS: P^T 128 x 129 + S 128 x 65 floats
constexpr int GJ_PP = GJ_NB + 1, GJ_SP = 64 + 1;       // odd pitches: the transposing stores and the reads are conflict-free
constexpr int GJ_CROSS_LDS = (GJ_NB * GJ_PP + GJ_NB * GJ_SP) * 4;
__global__ __launch_bounds__(256) void gj_cross_kernel(float* __restrict__ A, int D, int Dp, int k0, int bs,
                                                       const float* __restrict__ P, float4* __restrict__ CT,
                                                       float4* __restrict__ RT) {
  extern __shared__ __attribute__((aligned(16))) float gj_smem[];
  float* Ps = gj_smem;                                  // Ps[c][r] = P[r][c]
  float* Ss = gj_smem + GJ_NB * GJ_PP;                  // Ss[c][x] = S[c][o + x]
  const int o = blockIdx.x * 64, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool inblk = (o >= k0 && o < k0 + GJ_NB), above = (o < k0);
  if (inblk) {
    // row panel = P, column panel = 0, A_kk = P (its first 64 or last 64 columns)
    const int x = tid & 63, j = o - k0 + x, gi = o + x;
    for (int q = tid >> 6; q < 32; q += 4) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        v[e] = (gi < D && k < bs) ? P[k * GJ_NB + j] : 0.f;
        if (gi < D && k < bs) A[(int64_t)(k0 + k) * D + gi] = v[e];
      }
      RT[(int64_t)q * Dp + gi] = make_float4(v[0], v[1], v[2], v[3]);
      CT[(int64_t)q * Dp + gi] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  for (int it = 0; it < 64; ++it) {                     // P, coalesced along its rows
    const int r = 2 * it + (tid >> 7), c = tid & 127;
    Ps[c * GJ_PP + r] = P[r * GJ_NB + c];
  }
  if (above) {
    for (int it = 0; it < 32; ++it) {                   // rows of the block, coalesced along the strip
      const int c = 4 * it + (tid >> 6), x = tid & 63;
      Ss[c * GJ_SP + x] = (c < bs) ? A[(int64_t)(k0 + c) * D + o + x] : 0.f;
    }
  } else {
    for (int it = 0; it < 32; ++it) {                   // rows of the strip, coalesced along the block's columns
      const int x = 2 * it + (tid >> 7), c = tid & 127;
      Ss[c * GJ_SP + x] = (o + x < D && c < bs) ? A[(int64_t)(o + x) * D + k0 + c] : 0.f;
    }
  }
  __syncthreads();
  {
    const int x = tid & 63, gi = o + x;
    const float sgn = above ? -1.f : 1.f;
    for (int q = tid >> 6; q < 32; q += 4)
      CT[(int64_t)q * Dp + gi] = make_float4(sgn * Ss[(4 * q) * GJ_SP + x], sgn * Ss[(4 * q + 1) * GJ_SP + x],
                                             sgn * Ss[(4 * q + 2) * GJ_SP + x], sgn * Ss[(4 * q + 3) * GJ_SP + x]);
  }
  const int kk = lane >> 5, li = lane & 31;
  gj_f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll 8
  for (int k = 0; k < GJ_NB; k += 2) {
    const float a = Ps[(k + kk) * GJ_PP + 32 * wave + li];
    const float b0 = Ss[(k + kk) * GJ_SP + li], b1 = Ss[(k + kk) * GJ_SP + 32 + li];
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gi = o + 32 * j + li;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q = 8 * wave + 2 * g + kk;              // rows 4q .. 4q+3 of R
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (gi < D && 4 * q + e < bs) ? acc[j][4 * g + e] : 0.f;
        if (above && 4 * q + e < bs) A[(int64_t)(k0 + 4 * q + e) * D + gi] = v[e];
      }
      RT[(int64_t)q * Dp + gi] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}


// Trailing update of a block step on the f32 matrix cores (exact: an MFMA chain is an fmaf chain):
//   A[i][j] <- A[i][j] - sum_k C[i][k] R[k][j]     outside the block column,
//   A[i][j] <-         - sum_k C[i][k] P[k][j-k0]  inside it (the old values are the panel C itself),
// rows inside the block are left to gj_cross_kernel.  K = 128; a workgroup owns a 128 x 128 tile
// (wave = 64 x 64 = 2 x 2 MFMA blocks), operands come straight from the L2-resident panels with one
// 16-byte load per lane and k quad (MFMA e of an octet pairs k = 8o+e with 8o+4+e: the two lane
// halves load consecutive quads).
// Measured at D = 6300 (2450 tiles): 142 us per update = 72 TFLOP/s (the generic gemm_kernel needed
// 224 us for the two updates this replaces).  Knobs that measured the same or worse: deeper source-level
// prefetch (the compiler schedules the loads itself; pinning them with scheduling barriers: 150-240
// us), register budgets for 3-5 waves per SIMD (165 us), reading the tile before the MFMA chain
// (168 us), the panels through LDS (round 3: 11.1-12.1 against 10.8 ms per inverse).
//
// Gauss-Jordan on a symmetric matrix keeps M_ij = t_i t_j M_ji with t = -1 for processed blocks and +1
// otherwise, so only the tiles on and below the diagonal are computed and (round 3) they are not mirrored into the
// upper triangle at every block step (80 of the 240 MB a step moved at D = 6300): the upper triangle goes stale,
// gj_cross_kernel forms a step's panels from the lower one, and gj_mirror_kernel rebuilds the upper triangle once at
// the end.  The values used are the ones the mirrored matrix held: results are unchanged bit for bit.
//
// PHASE (round 4, look-ahead): 0 = every tile of the step; 1 = only the tiles the NEXT step's diagonal inverse and
// panels read - block column kb+1 from the diagonal down and block row kb+1 left of it; 2 = all the others.  1 then 2
// is 0 tile by tile, and between them the next step's gj_diag_kernel / gj_cross_kernel run on a second stream beside
// phase 2 (spd_inverse_steps).
// grid: PHASE 1: 4 Dp/128 workgroups of ONE wave (64 x 64 each: the Dp/128 tiles are on the critical path, as one-wave
// workgroups they take a third of the time); else (Dp/128)(Dp/128 + 1)/2 lower-triangle tiles, row-block major, block 256
template <int PHASE>
__global__ __launch_bounds__(256, 2) void gj_trailing_kernel(float* __restrict__ A, int D, int Dp, int k0,
                                                             const float4* __restrict__ CT,
                                                             const float4* __restrict__ RT) {
  const int nt = Dp >> 7, kb = k0 >> 7, nb = kb + 1;
  int tm, tn;
  if constexpr (PHASE == 1) {
    const int t = blockIdx.x >> 2;                        // one-wave workgroups: four to a tile
    if (t < nt - nb) { tm = nb + t; tn = nb; }
    else { tm = nb; tn = t - (nt - nb); }
  } else {
    const int t = blockIdx.x;
    tm = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;            // (the float square root may be one off either way)
    while (tm * (tm + 1) / 2 > t) --tm;
    tn = t - tm * (tm + 1) / 2;
    if (PHASE == 2 && (tm == nb || tn == nb)) return;
  }
  if (tm == kb) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = PHASE == 1 ? (int)(blockIdx.x & 3) : tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int m0 = tm * 128 + (wave >> 1) * 64, n0 = tn * 128 + (wave & 1) * 64;
  const bool blockcol = (tn == kb);
  const float4* cp = CT + (int64_t)kk * Dp + m0 + li;
  const float4* rp = RT + (int64_t)kk * Dp + n0 + li;
  // Operand loads AHEAD k octets ahead of the MFMAs that use them, in a ring of AHEAD + 1 register sets.
  // PHASE 0 / 2 (every tile resident at once at D = 6300: 92 registers, five waves per SIMD): one octet ahead, and the
  // compiler is free to merge the two sets and issue an octet's loads behind the MFMAs of the one before - a wave then
  // waits a trip to the L2 / Infinity Cache per octet (31 us per 64 x 64 wave tile against 6.8 us of MFMA time, kernel
  // trace of one-wave workgroups), which the five waves of a SIMD cover for each other.  A real ring three octets ahead
  // (142 registers, three waves) makes a lone wave faster (20 us) and the full update slower (90-96 against 80 us:
  // gpurun r04i), so only PHASE 1 - a few one-wave workgroups on the critical path - uses it.
  constexpr int AHEAD = PHASE == 1 ? 3 : 1, RING = AHEAD + 1;
  float4 av[RING][2], bv[RING][2];
  auto load = [&](int buf, int c) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      av[buf][x] = cp[(int64_t)2 * c * Dp + 32 * x];
      bv[buf][x] = rp[(int64_t)2 * c * Dp + 32 * x];
    }
  };
#pragma unroll
  for (int c = 0; c < AHEAD; ++c) load(c, c);
  gj_f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c + AHEAD < 16) load((c + AHEAD) % RING, c + AHEAD);
    if constexpr (PHASE == 1) __builtin_amdgcn_sched_barrier(0);      // the loads stay where they are issued
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const float4 a = av[c % RING][mi];
      const float a4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const float4 b = bv[c % RING][ni];
        const float b4[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[mi][ni], 0, 0, 0);
      }
    }
    if constexpr (PHASE == 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + 32 * ni + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * mi + 8 * (r >> 2) + 4 * kk + (r & 3);
        if (row < D && col < D) {
          float* p = A + (int64_t)row * D + col;
          *p = (blockcol ? 0.f : *p) - acc[mi][ni][r];
        }
      }
    }
}

// TWO block steps' trailing updates in one pass over the tiles (round 4): a tile is read and written once for the
// steps a = k0 / 128 and b = a + 1, each step's sum in its own MFMA chain, applied one after the other -
//   v = (old - sum_a) - sum_b,  old - sum_a rounded to float32 exactly as step a stored it
// - so the result is the two single-step updates bit for bit, with half the passes over the matrix (a block step
// moves the whole lower triangle for 128 k per element: 32 FLOP per byte of read-modify-write).  Before it runs, the
// tiles step b's diagonal inverse and panels read have had step a applied on their own (gj_trailing_kernel<1>) and
// gj_cross_kernel of step b has written row block b and both panels; per tile:
//   row block b            - nothing (written by step b's cross kernel)
//   row block a            - step b only (the row holds step a's row panel)
//   column block b, below  - step b only, as its block column (step a is already in)
//   column block a, below  - step a as its block column, then step b
//   elsewhere              - both
// 247 registers (both steps' sums and the tile: two waves per SIMD; at three the compiler spills 76 words).
// Measured (tools/time_inverse.py): D = 21 000: 119 against 173 ms (0.49 of the f32 MFMA peak counting D^3), D = 10 500:
// 20.4 against 26.0, D = 6300: 7.67 against 7.64 - there the pass over 1225 tiles is 141 us against 2 x 80 and the extra
// launch for step k+1's cross costs what that saves.
// grid: (Dp/128)(Dp/128 + 1)/2 lower-triangle tiles, row-block major   block: 256
__global__ __launch_bounds__(256, 2) void gj_trailing2_kernel(float* __restrict__ A, int D, int Dp, int k0,
                                                              const float4* __restrict__ CTa,
                                                              const float4* __restrict__ RTa,
                                                              const float4* __restrict__ CTb,
                                                              const float4* __restrict__ RTb) {
  const int ka = k0 >> 7, kb = ka + 1;
  const int t = blockIdx.x;
  int tm = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
  while (tm * (tm + 1) / 2 > t) --tm;
  const int tn = t - tm * (tm + 1) / 2;
  if (tm == kb) return;
  const bool do_a = (tm != ka) && (tn != kb), a_col = (tn == ka), b_col = (tn == kb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int m0 = tm * 128 + (wave >> 1) * 64, n0 = tn * 128 + (wave & 1) * 64;
  auto chain = [&](const float4* __restrict__ CT, const float4* __restrict__ RT, gj_f32x16 (&acc)[2][2]) {
    const float4* cp = CT + (int64_t)kk * Dp + m0 + li;
    const float4* rp = RT + (int64_t)kk * Dp + n0 + li;
    float4 av[2][2], bv[2][2];
    auto load = [&](int buf, int c) {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        av[buf][x] = cp[(int64_t)2 * c * Dp + 32 * x];
        bv[buf][x] = rp[(int64_t)2 * c * Dp + 32 * x];
      }
    };
    load(0, 0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c + 1 < 16) load((c + 1) & 1, c + 1);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float4 a = av[c & 1][mi];
        const float a4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float4 b = bv[c & 1][ni];
          const float b4[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[mi][ni], 0, 0, 0);
        }
      }
    }
  };
  gj_f32x16 v[2][2], acc[2][2];
  if (do_a) chain(CTa, RTa, acc);
  const bool need_old = !b_col && !(do_a && a_col);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + 32 * ni + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * mi + 8 * (r >> 2) + 4 * kk + (r & 3);
        const float old = (need_old && row < D && col < D) ? A[(int64_t)row * D + col] : 0.f;
        v[mi][ni][r] = do_a ? old - acc[mi][ni][r] : old;
      }
    }
  chain(CTb, RTb, acc);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + 32 * ni + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * mi + 8 * (r >> 2) + 4 * kk + (r & 3);
        if (row < D && col < D) A[(int64_t)row * D + col] = (b_col ? 0.f : v[mi][ni][r]) - acc[mi][ni][r];
      }
    }
}

int gj_kernel_attrs(dmp_ctx* c) {
  static bool done[64] = {};
  if (c->device >= 0 && c->device < 64 && done[c->device]) return DMP_OK;
  DMP_HIP(hipFuncSetAttribute((const void*)gj_cross_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GJ_CROSS_LDS));
  DMP_HIP(hipFuncSetAttribute((const void*)gj_diag_blocked_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GJ_DIAGB_LDS));
  if (c->device >= 0 && c->device < 64) done[c->device] = true;
  return DMP_OK;
}

// After the last block step every block is processed (t = -1 everywhere: t_i t_j = +1): upper = lower, transposed.
// grid: (nt, nt) 64 x 64 tiles, only those above the diagonal act   block: 256
__global__ __launch_bounds__(256) void gj_mirror_kernel(float* __restrict__ A, int D) {
  const int ti = blockIdx.y, tj = blockIdx.x;          // destination tile (rows ti, columns tj), tj >= ti
  if (tj < ti) return;
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  for (int it = 0; it < 16; ++it) {                    // source tile (rows tj, columns ti): coalesced along its columns
    const int r = 4 * it + (tid >> 6), c = tid & 63;
    const int gr = tj * 64 + r, gc = ti * 64 + c;
    tile[r][c] = (gr < D && gc < D) ? A[(int64_t)gr * D + gc] : 0.f;
  }
  __syncthreads();
  for (int it = 0; it < 16; ++it) {
    const int r = 4 * it + (tid >> 6), c = tid & 63;
    const int gr = ti * 64 + r, gc = tj * 64 + c;
    if (gr < D && gc < D && gc > gr) A[(int64_t)gr * D + gc] = tile[c][r];
  }
}

int spd_inverse(dmp_ctx* c, float* A, int D, hipStream_t s, hipStream_t la) {
  return spd_inverse_steps(c, A, D, 0, cdiv(D, GJ_NB), s, la);
}

// Block steps [blk_lo, blk_hi).  Per step: the diagonal block's inverse (one workgroup), the cross kernel (panels),
// the trailing update - three launches (six in rounds 1-3); by default (option gj_pairs) two steps share one pass of
// the trailing update over the tiles.  With gj_pairs = 0:
// `la` (a second stream, or null): LOOK-AHEAD inside the range.  The trailing update of step k is issued in two parts,
// first the tiles step k+1's diagonal inverse and panels read (PHASE 1: Dp/128 tiles), then the rest; the moment the
// first part is done, step k+1's diagonal inverse and cross kernel run on `la`, into the other set of panel buffers,
// beside the second part.  The one-workgroup sweep (52 us of a step's 150 at D = 6300) and the panel kernel leave the
// critical path; every tile sees the same operations in the same order: same bits as without.  Used where one
// prediction has the machine to itself (dmp_predict, dmp_spd_inverse); a scheduler's engines keep one stream each.
// Measured (tools/time_inverse.py, profiles/r04_inverse.txt): D = 10500: 23.5 against 25.4 ms, D = 21000: 169.5 against
// 173.4; D = 6300: 8.7 against 7.7 - beside the trailing update the one-workgroup sweep takes 100 us instead of 52 (the
// clocks under a chip full of f32 MFMAs, the shared CU), which at that size is longer than the update it hides behind.
// So the look-ahead is used from 64 tile rows (D > 8064) on.
constexpr int GJ_LOOKAHEAD_MIN_TILES = 64;
int spd_inverse_steps(dmp_ctx* c, float* A, int D, int blk_lo, int blk_hi, hipStream_t s, hipStream_t la) {
  const int Dp = round_up(D, GJ_NB), nt = Dp / GJ_NB;
  const int64_t pan = (int64_t)Dp * GJ_NB;
  int rc;
  auto Pb = [&](int k) { return c->gj_p + (int64_t)(k & 1) * GJ_NB * GJ_NB; };
  auto CTb = [&](int k) { return reinterpret_cast<float4*>(c->gj_c + (k & 1) * pan); };
  auto RTb = [&](int k) { return reinterpret_cast<float4*>(c->gj_rt + (k & 1) * pan); };
  auto prepare = [&](int k, hipStream_t st) -> int {    // P, panels and row write-back of step k
    const int k0 = k * GJ_NB, bs = std::min(GJ_NB, D - k0);
    if (c->gj_diag_blocked) hipLaunchKernelGGL(gj_diag_blocked_kernel, dim3(1), dim3(512), GJ_DIAGB_LDS, st, A, D, k0, bs, Pb(k));
    else if (c->gj_diag_groups == 2) hipLaunchKernelGGL(gj_diag_kernel<2>, dim3(1), dim3(256), 0, st, A, D, k0, bs, Pb(k));
    else if (c->gj_diag_groups == 8) hipLaunchKernelGGL(gj_diag_kernel<8>, dim3(1), dim3(1024), 0, st, A, D, k0, bs, Pb(k));
    else hipLaunchKernelGGL(gj_diag_kernel<4>, dim3(1), dim3(512), 0, st, A, D, k0, bs, Pb(k));
    DMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gj_cross_kernel, dim3(Dp / 64), dim3(256), GJ_CROSS_LDS, st, A, D, Dp, k0, bs, Pb(k), CTb(k), RTb(k));
    DMP_LAUNCH_CHECK();
    return DMP_OK;
  };
  if (nt < (c->gj_lookahead == 2 ? 0 : GJ_LOOKAHEAD_MIN_TILES)) la = nullptr;     // option 2: at every size (tests)
  const int k_end = std::min(blk_hi, cdiv(D, GJ_NB));
  const dim3 lower(nt * (nt + 1) / 2);
  auto mirror_if_last = [&](int k) -> int {
    const int k0 = k * GJ_NB, bs = std::min(GJ_NB, D - k0);
    if (k0 + bs >= D && D > GJ_NB) {                   // the last block step: rebuild the upper triangle
      const int n64 = cdiv(D, 64);
      hipLaunchKernelGGL(gj_mirror_kernel, dim3(n64, n64), dim3(256), 0, s, A, D);
      DMP_LAUNCH_CHECK();
    }
    return DMP_OK;
  };
  if (c->gj_pairs) {
    // block steps in PAIRS: step k's panels, step k applied to the tiles step k+1's sweep and panels read, step k+1's
    // panels, then both steps' updates of everything else in one pass over the tiles (gj_trailing2_kernel)
    int k = blk_lo;
    while (k < k_end) {
      const int k0 = k * GJ_NB;
      if ((rc = prepare(k, s))) return rc;
      if (k + 1 < k_end) {
        hipLaunchKernelGGL(gj_trailing_kernel<1>, dim3(4 * nt), dim3(64), 0, s, A, D, Dp, k0, CTb(k), RTb(k));
        DMP_LAUNCH_CHECK();
        if ((rc = prepare(k + 1, s))) return rc;
        hipLaunchKernelGGL(gj_trailing2_kernel, lower, dim3(256), 0, s, A, D, Dp, k0, CTb(k), RTb(k), CTb(k + 1), RTb(k + 1));
        DMP_LAUNCH_CHECK();
        if ((rc = mirror_if_last(k + 1))) return rc;
        k += 2;
      } else {
        hipLaunchKernelGGL(gj_trailing_kernel<0>, lower, dim3(256), 0, s, A, D, Dp, k0, CTb(k), RTb(k));
        DMP_LAUNCH_CHECK();
        if ((rc = mirror_if_last(k))) return rc;
        k += 1;
      }
    }
    return DMP_OK;
  }
  for (int k = blk_lo; k < k_end; ++k) {
    const int k0 = k * GJ_NB;
    const bool ahead = la && k + 1 < k_end;
    if (k == blk_lo || !la) { if ((rc = prepare(k, s))) return rc; }
    // A -= C R outside the block column, A[:, block] = -C P inside it
    if (ahead) {
      hipLaunchKernelGGL(gj_trailing_kernel<1>, dim3(4 * nt), dim3(64), 0, s, A, D, Dp, k0, CTb(k), RTb(k));
      DMP_LAUNCH_CHECK();
      DMP_HIP(hipEventRecord((hipEvent_t)c->gj_ev[0], s));
      DMP_HIP(hipStreamWaitEvent(la, (hipEvent_t)c->gj_ev[0], 0));
      if ((rc = prepare(k + 1, la))) return rc;
      DMP_HIP(hipEventRecord((hipEvent_t)c->gj_ev[1], la));
      hipLaunchKernelGGL(gj_trailing_kernel<2>, lower, dim3(256), 0, s, A, D, Dp, k0, CTb(k), RTb(k));
      DMP_LAUNCH_CHECK();
      DMP_HIP(hipStreamWaitEvent(s, (hipEvent_t)c->gj_ev[1], 0));
    } else {
      hipLaunchKernelGGL(gj_trailing_kernel<0>, lower, dim3(256), 0, s, A, D, Dp, k0, CTb(k), RTb(k));
      DMP_LAUNCH_CHECK();
    }
    if ((rc = mirror_if_last(k))) return rc;
  }
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// contacts with average-product correction                              (predict.py:58-60)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void contact_norm_kernel(const float* __restrict__ inv, int L,
                                                           float* __restrict__ x3) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  const int64_t D = (int64_t)L * NS;
  float s = 0.f;
  for (int a = 0; a < NS - 1; ++a) {
    const float* row = inv + ((int64_t)i * NS + a) * D + (int64_t)j * NS;
#pragma unroll
    for (int b = 0; b < NS - 1; ++b) s += row[b] * row[b];
  }
  x3[(int64_t)i * L + j] = (i == j) ? 0.f : sqrtf(s);
}

// sums[0..L) column sums, sums[L..2L) row sums, sums[2L] total
__global__ __launch_bounds__(256) void apc_sums_kernel(const float* __restrict__ x3, int L,
                                                       double* __restrict__ sums) {
  __shared__ double red[2][256];
  const int t = blockIdx.x;
  double cs = 0.0, rs = 0.0;
  for (int k = threadIdx.x; k < L; k += 256) {
    cs += (double)x3[(int64_t)k * L + t];
    rs += (double)x3[(int64_t)t * L + k];
  }
  red[0][threadIdx.x] = cs;
  red[1][threadIdx.x] = rs;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[t] = red[0][0];
    sums[L + t] = red[1][0];
  }
}

__global__ __launch_bounds__(256) void apc_total_kernel(int L, double* __restrict__ sums) {
  __shared__ double red[256];
  double a = 0.0;
  for (int k = threadIdx.x; k < L; k += 256) a += sums[L + k];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[2 * L] = red[0];
}

__global__ __launch_bounds__(256) void apc_apply_kernel(const float* __restrict__ x3, int L,
                                                        const double* __restrict__ sums,
                                                        float* __restrict__ contacts) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  const float apc = ((float)sums[j] * (float)sums[L + i]) / (float)sums[2 * L];
  contacts[(int64_t)i * L + j] = (i == j) ? 0.f : (x3[(int64_t)i * L + j] - apc);
}

// fast_dca's return value as the reference lays it out (predict.py:54-61): out[i][j][21a+b] = inv[21i+a][21j+b],
// out[i][j][441] = contacts[i][j].  One thread per output float: the writes are contiguous, the reads are runs
// of 21 floats.  (The prediction path never builds this tensor - the stem reads `inv` in place; this is the
// training-side consumer's layout, train.py:175-190.)
__global__ __launch_bounds__(256) void dca_features_kernel(const float* __restrict__ inv,
                                                           const float* __restrict__ contacts, int L,
                                                           float* __restrict__ out) {
  const int64_t total = (int64_t)L * L * NUM_DCA;
  const int64_t D = (int64_t)L * NS;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t pair = e / NUM_DCA;
    const int ch = (int)(e - pair * NUM_DCA);
    const int i = (int)(pair / L), j = (int)(pair - (int64_t)i * L);
    float v;
    if (ch == NUM_DCA - 1) v = contacts[pair];
    else {
      const int a = ch / NS, b = ch - a * NS;
      v = inv[((int64_t)i * NS + a) * D + (int64_t)j * NS + b];
    }
    out[e] = v;
  }
}

int dca_features(const float* d_inv, const float* d_contacts, int L, float* d_out, hipStream_t s) {
  const int64_t total = (int64_t)L * L * NUM_DCA;
  const int grid = (int)std::min<int64_t>(cdiv64(total, 256), 256 * 32);
  hipLaunchKernelGGL(dca_features_kernel, dim3(grid), dim3(256), 0, s, d_inv, d_contacts, L, d_out);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

int dca_contacts(dmp_ctx* c, const float* d_inv, int L, float* d_contacts, hipStream_t s) {
  dim3 grid(cdiv(L, 256), L);
  hipLaunchKernelGGL(contact_norm_kernel, grid, dim3(256), 0, s, d_inv, L, c->x3);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(apc_sums_kernel, dim3(L), dim3(256), 0, s, c->x3, L, c->apc_sums);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(apc_total_kernel, dim3(1), dim3(256), 0, s, L, c->apc_sums);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(apc_apply_kernel, grid, dim3(256), 0, s, c->x3, L, c->apc_sums, d_contacts);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
