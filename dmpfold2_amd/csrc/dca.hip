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
                                                         const float* __restrict__ sc) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)D * D) return;
  const int i = idx / D, j = idx - (int64_t)i * D;
  float v = cov[idx] / sc[1];
  if (i == j) v += 4.5f / sqrtf(sc[0]);
  cov[idx] = v;
}

int cov_build(dmp_ctx* c, const uint8_t* d_msa, const float* d_w, int N, int L, float* d_cov,
              hipStream_t s) {
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
  int rc = gemm_f32(g, s);
  if (rc) return rc;
  hipLaunchKernelGGL(cov_finish_kernel, dim3((unsigned)cdiv64((int64_t)D * D, 256)), dim3(256), 0,
                     s, d_cov, D, sc);
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

// Panels of a block step in the layout the trailing-update kernel loads with 16 bytes per lane:
//   CT[k/4][i][4] = A[i][k0+k]  (column panel; zero for rows i inside the block and for k >= bs)
//   RT[k/4][j][4] = R[k][j]     (row panel P A_k,: ; inside the block columns: P[k][j-k0], which makes
//                                the trailing update of a zeroed block column produce -C P)
// grid: Dp/64 blocks (64 rows / columns each)   block: 256
__global__ __launch_bounds__(256) void gj_panels_kernel(const float* __restrict__ A, int D, int Dp, int k0,
                                                        int bs, const float* __restrict__ P,
                                                        const float* __restrict__ R,
                                                        float4* __restrict__ CT, float4* __restrict__ RT) {
  __shared__ float tile[64][GJ_NB + 1];
  const int i0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int r = 0; r < 32; ++r) {                      // coalesced rows of the column panel
    const int row = 2 * r + (tid >> 7), col = tid & 127, i = i0 + row;
    const bool inblk = (i >= k0 && i < k0 + bs);
    tile[row][col] = (i < D && col < bs && !inblk) ? A[(int64_t)i * D + k0 + col] : 0.f;
  }
  __syncthreads();
  const int i = tid & 63;
  const int gi = i0 + i;
  const bool inblk = (gi >= k0 && gi < k0 + bs);
  for (int it = 0; it < 8; ++it) {
    const int q = (tid >> 6) + 4 * it;
    CT[(int64_t)q * Dp + gi] = make_float4(tile[i][4 * q], tile[i][4 * q + 1], tile[i][4 * q + 2], tile[i][4 * q + 3]);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * q + e;
      v[e] = (gi < D && k < bs) ? (inblk ? P[k * GJ_NB + (gi - k0)] : R[(int64_t)k * D + gi]) : 0.f;
    }
    RT[(int64_t)q * Dp + gi] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

typedef float gj_f32x16 __attribute__((ext_vector_type(16)));

// Trailing update of a block step on the f32 matrix cores (exact: an MFMA chain is an fmaf chain):
//   A[i][j] <- A[i][j] - sum_k C[i][k] R[k][j]     outside the block column,
//   A[i][j] <-         - sum_k C[i][k] P[k][j-k0]  inside it (the old values are the panel C itself),
// rows inside the block are left to gj_writeback_kernel.  K = 128; a workgroup owns a 128 x 128 tile
// (wave = 64 x 64 = 2 x 2 MFMA blocks), operands come straight from the L2-resident panels with one
// 16-byte load per lane and k quad (MFMA e of an octet pairs k = 8o+e with 8o+4+e: the two lane
// halves load consecutive quads).
// Measured at D = 6300 (2450 tiles): 142 us per update = 72 TFLOP/s (the generic gemm_kernel needed
// 224 us for the two updates this replaces).  Knobs that measured the same or worse: deeper source-level
// prefetch (the compiler schedules the loads itself; pinning them with scheduling barriers: 150-240
// us), register budgets for 3-5 waves per SIMD (165 us), reading the tile before the MFMA chain
// (168 us).  A CU pulls 64 KB of panel per wave and tile from L2; sharing the panels
// through LDS is the next step.
// grid: (Dp/128)^2 blocks, row-block major   block: 256
__global__ __launch_bounds__(256, 2) void gj_trailing_kernel(float* __restrict__ A, int D, int Dp, int k0,
                                                                   const float4* __restrict__ CT,
                                                                   const float4* __restrict__ RT) {
  const int nt = Dp >> 7;
  const int tm = blockIdx.x / nt, tn = blockIdx.x % nt, kb = k0 >> 7;
  if (tm == kb) return;
  // Gauss-Jordan on a symmetric matrix keeps M_ij = t_i t_j M_ji with t = -1 for processed blocks and +1
  // otherwise, so only the tiles on and below the diagonal are computed.  Round 3: they are no longer mirrored
  // into the upper triangle at every block step (80 of the 240 MB a step moved at D = 6300): the upper triangle goes
  // stale, gj_symm_kernel refreshes the two panels a block step reads from it - the block row right of the diagonal
  // block and the block column above it - from the lower triangle before the step, and gj_mirror_kernel
  // rebuilds the whole upper triangle once at the end.  The values read are the ones the mirrored matrix held:
  // results are unchanged bit for bit.
  if (tn > tm) return;                                  // upper triangle (for tn == kb: rows above the block)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int m0 = tm * 128 + (wave >> 1) * 64, n0 = tn * 128 + (wave & 1) * 64;
  const bool blockcol = (tn == kb);
  const float4* cp = CT + (int64_t)kk * Dp + m0 + li;
  const float4* rp = RT + (int64_t)kk * Dp + n0 + li;
  constexpr int PF = 1;      // k octets of operand loads in flight ahead of the MFMAs
  float4 av[2][PF][2], bv[2][PF][2];
  auto load = [&](int buf, int c) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        av[buf][u][x] = cp[(int64_t)2 * (PF * c + u) * Dp + 32 * x];
        bv[buf][u][x] = rp[(int64_t)2 * (PF * c + u) * Dp + 32 * x];
      }
  };
  load(0, 0);
  gj_f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
  for (int c = 0; c < 16 / PF; ++c) {
    if (c + 1 < 16 / PF) load((c + 1) & 1, c + 1);
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float4 a = av[c & 1][u][mi];
        const float a4[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float4 b = bv[c & 1][u][ni];
          const float b4[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b4[e], acc[mi][ni], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + 32 * ni + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * mi + 8 * (r >> 2) + 4 * kk + (r & 3);
        float v = 0.f;
        if (row < D && col < D) {
          float* p = A + (int64_t)row * D + col;
          v = (blockcol ? 0.f : *p) - acc[mi][ni][r];
          *p = v;
        }
      }
    }
}

int gj_kernel_attrs(dmp_ctx* c) {
  static bool done[64] = {};
  if (c->device >= 0 && c->device < 64 && done[c->device]) return DMP_OK;
  if (c->device >= 0 && c->device < 64) done[c->device] = true;
  return DMP_OK;
}

// Before block step k: the entries the step reads from the (stale) upper triangle, rebuilt from the lower one with
// the sign rule M_ij = t_i t_j M_ji (block k itself not yet processed: t_k = +1):
//   block row, right of the diagonal block   A[k0 + r][j] =  A[j][k0 + r]   (j >= k0 + bs, unprocessed: +1)
//   block column, above the diagonal block   A[i][k0 + c] = -A[k0 + c][i]   (i < k0, processed: -1)
// grid: ceil(D / 64) tiles of 64 matrix rows / columns outside the block   block: 256
__global__ __launch_bounds__(256) void gj_symm_kernel(float* __restrict__ A, int D, int k0, int bs) {
  __shared__ float tile[64][GJ_NB + 1];
  const int o0 = blockIdx.x * 64, tid = threadIdx.x;
  if (o0 + 64 <= k0) {
    // columns i = o0 .. o0+63 left of the block: read A[k0 + c][i] (coalesced along i), write A[i][k0 + c] = -that
    for (int it = 0; it < 32; ++it) {
      const int c = 4 * it + (tid >> 6), x = tid & 63;
      tile[x][c] = (c < bs) ? A[(int64_t)(k0 + c) * D + o0 + x] : 0.f;
    }
    __syncthreads();
    for (int it = 0; it < 32; ++it) {
      const int x = 2 * it + (tid >> 7), c = tid & 127;
      if (c < bs) A[(int64_t)(o0 + x) * D + k0 + c] = -tile[x][c];
    }
  } else if (o0 >= k0 + bs) {
    // rows j = o0 .. o0+63 below the block: read A[j][k0 + r] (coalesced along r), write A[k0 + r][j] = that
    for (int it = 0; it < 32; ++it) {
      const int x = 2 * it + (tid >> 7), r = tid & 127;
      tile[x][r] = (o0 + x < D && r < bs) ? A[(int64_t)(o0 + x) * D + k0 + r] : 0.f;
    }
    __syncthreads();
    for (int it = 0; it < 32; ++it) {
      const int r = 4 * it + (tid >> 6), x = tid & 63;
      if (r < bs && o0 + x < D) A[(int64_t)(k0 + r) * D + o0 + x] = tile[x][r];
    }
  } else {
    // the 64-wide tile straddles the block's edges (k0 is a multiple of 128, so only its far edge when bs < 128, or
    // nothing at all): element-wise, both rules
    for (int e = tid; e < 64 * GJ_NB; e += 256) {
      const int x = e / GJ_NB, c = e % GJ_NB, o = o0 + x;
      if (c >= bs || o >= D) continue;
      if (o < k0) A[(int64_t)o * D + k0 + c] = -A[(int64_t)(k0 + c) * D + o];
      else if (o >= k0 + bs) A[(int64_t)(k0 + c) * D + o] = A[(int64_t)o * D + k0 + c];
    }
  }
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

// A_k,: = R (outside the block), A_kk = P
__global__ __launch_bounds__(256) void gj_writeback_kernel(float* __restrict__ A, int D, int k0,
                                                           int bs, const float* __restrict__ R,
                                                           const float* __restrict__ P) {
  const int r = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= D || r >= bs) return;
  const bool inblk = (j >= k0 && j < k0 + bs);
  A[(int64_t)(k0 + r) * D + j] = inblk ? P[r * GJ_NB + (j - k0)] : R[(int64_t)r * D + j];
}

int spd_inverse(dmp_ctx* c, float* A, int D, hipStream_t s) {
  return spd_inverse_steps(c, A, D, 0, cdiv(D, GJ_NB), s);
}

int spd_inverse_steps(dmp_ctx* c, float* A, int D, int blk_lo, int blk_hi, hipStream_t s) {
  float *P = c->gj_p, *R = c->gj_r;
  const int Dp = round_up(D, GJ_NB);
  float4* CT = reinterpret_cast<float4*>(c->gj_c);
  float4* RT = reinterpret_cast<float4*>(c->gj_rt);
  for (int k0 = blk_lo * GJ_NB; k0 < D && k0 < blk_hi * GJ_NB; k0 += GJ_NB) {
    const int bs = std::min(GJ_NB, D - k0);
    if (k0 > 0 || k0 + bs < D) {
      hipLaunchKernelGGL(gj_symm_kernel, dim3(cdiv(D, 64)), dim3(256), 0, s, A, D, k0, bs);
      DMP_LAUNCH_CHECK();
    }
    if (c->gj_diag_groups == 2) hipLaunchKernelGGL(gj_diag_kernel<2>, dim3(1), dim3(256), 0, s, A, D, k0, bs, P);
    else if (c->gj_diag_groups == 8) hipLaunchKernelGGL(gj_diag_kernel<8>, dim3(1), dim3(1024), 0, s, A, D, k0, bs, P);
    else hipLaunchKernelGGL(gj_diag_kernel<4>, dim3(1), dim3(512), 0, s, A, D, k0, bs, P);
    DMP_LAUNCH_CHECK();
    GemmArgs g{};
    // R = P * A[k0:k0+bs, :]
    g.A = P; g.sam = GJ_NB; g.sak = 1;
    g.B = A + (int64_t)k0 * D; g.sbk = D; g.sbn = 1;
    g.C = R; g.ldc = D; g.M = bs; g.N = D; g.K = bs; g.alpha = 1.f; g.beta = 0.f;
    int rc = gemm_f32(g, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gj_panels_kernel, dim3(Dp / 64), dim3(256), 0, s, A, D, Dp, k0, bs, P, R, CT, RT);
    DMP_LAUNCH_CHECK();
    // A -= C R outside the block column, A[:, block] = -C P inside it
    const dim3 tgrid((Dp / 128) * (Dp / 128));
    hipLaunchKernelGGL(gj_trailing_kernel, tgrid, dim3(256), 0, s, A, D, Dp, k0, CT, RT);
    DMP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gj_writeback_kernel, dim3(cdiv(D, 256), bs), dim3(256), 0, s, A, D, k0, bs,
                       R, P);
    DMP_LAUNCH_CHECK();
    if (k0 + bs >= D && D > GJ_NB) {                   // the last block step: rebuild the upper triangle
      const int nt = cdiv(D, 64);
      hipLaunchKernelGGL(gj_mirror_kernel, dim3(nt, nt), dim3(256), 0, s, A, D);
      DMP_LAUNCH_CHECK();
    }
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
