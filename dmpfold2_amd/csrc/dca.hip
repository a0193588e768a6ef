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
__global__ __launch_bounds__(256) void gj_diag_kernel(const float* __restrict__ A, int D, int k0,
                                                      int bs, float* __restrict__ P) {
  // Symmetric sweep operator (Goodnight): sweeping every pivot of an SPD block in place gives
  // -inverse and keeps the matrix symmetric, so a step only needs the pivot ROW broadcast.
  // Thread (j, ih) keeps rows ih*64 .. ih*64+63 of column j in 64 registers (static indexing:
  // the pivot loop is unrolled over the row-in-half index); the pivot row goes through LDS.
  __shared__ float rowk[2][GJ_NB];
  const int tid = threadIdx.x;
  const int j = tid & 127, ih = tid >> 7;
  float sreg[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int i = ih * 64 + r;
    sreg[r] = (i < bs && j < bs) ? A[(int64_t)(k0 + i) * D + k0 + j] : (i == j ? 1.f : 0.f);
  }
#pragma unroll 1
  for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      const int k = kh * 64 + r;
      if (k < bs) {                                   // uniform
        float* rk = rowk[k & 1];
        if (ih == kh) rk[j] = sreg[r];                // publish row k (column j's element)
        __syncthreads();
        const float invd = 1.0f / rk[k];
        const bool jk = (j == k);
        const float bj = rk[j] * invd;                // new a_kj
        const float mul = jk ? -invd : bj;            // column k: a_ik/d ; elsewhere a_ij - a_ik*b_j
#pragma unroll
        for (int rr = 0; rr < 64; ++rr) {
          const float aik = rk[ih * 64 + rr];         // a_ik = a_ki by symmetry
          sreg[rr] = fmaf(-aik, mul, jk ? 0.f : sreg[rr]);
        }
        if (ih == kh) sreg[r] = jk ? -invd : bj;      // the pivot row itself
      }
    }
  }
  // P = inverse = -swept matrix; padding rows/columns of a partial block become identity again
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const int i = ih * 64 + r;
    P[i * GJ_NB + j] = (i < bs && j < bs) ? -sreg[r] : (i == j ? 1.f : 0.f);
  }
}

// C[i][kk] = A[i][k0+kk] (zero for rows inside the block); also zero R's block columns
__global__ __launch_bounds__(256) void gj_panels_kernel(const float* __restrict__ A, int D, int k0,
                                                        int bs, float* __restrict__ Cp,
                                                        float* __restrict__ R) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)D * GJ_NB) return;
  const int i = idx >> 7, kk = idx & 127;
  const bool inblk = (i >= k0 && i < k0 + bs);
  Cp[idx] = (kk < bs && !inblk) ? A[(int64_t)i * D + k0 + kk] : 0.f;
  // R is [bs][D]; reuse the same index space: row kk, column i
  if (kk < bs && inblk) R[(int64_t)kk * D + i] = 0.f;
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
  float *P = c->gj_p, *R = c->gj_r, *Cp = c->gj_c;
  for (int k0 = blk_lo * GJ_NB; k0 < D && k0 < blk_hi * GJ_NB; k0 += GJ_NB) {
    const int bs = std::min(GJ_NB, D - k0);
    hipLaunchKernelGGL(gj_diag_kernel, dim3(1), dim3(256), 0, s, A, D, k0, bs, P);
    DMP_LAUNCH_CHECK();
    GemmArgs g{};
    // R = P * A[k0:k0+bs, :]
    g.A = P; g.sam = GJ_NB; g.sak = 1;
    g.B = A + (int64_t)k0 * D; g.sbk = D; g.sbn = 1;
    g.C = R; g.ldc = D; g.M = bs; g.N = D; g.K = bs; g.alpha = 1.f; g.beta = 0.f;
    int rc = gemm_f32(g, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gj_panels_kernel, dim3((unsigned)cdiv64((int64_t)D * GJ_NB, 256)),
                       dim3(256), 0, s, A, D, k0, bs, Cp, R);
    DMP_LAUNCH_CHECK();
    // A -= C * R
    g.A = Cp; g.sam = GJ_NB; g.sak = 1;
    g.B = R; g.sbk = D; g.sbn = 1;
    g.C = A; g.ldc = D; g.M = D; g.N = D; g.K = bs; g.alpha = -1.f; g.beta = 1.f;
    rc = gemm_f32(g, s);
    if (rc) return rc;
    // A[:, block] = -C * P
    g.A = Cp; g.sam = GJ_NB; g.sak = 1;
    g.B = P; g.sbk = GJ_NB; g.sbn = 1;
    g.C = A + k0; g.ldc = D; g.M = D; g.N = bs; g.K = bs; g.alpha = -1.f; g.beta = 0.f;
    rc = gemm_f32(g, s);
    if (rc) return rc;
    hipLaunchKernelGGL(gj_writeback_kernel, dim3(cdiv(D, 256), bs), dim3(256), 0, s, A, D, k0, bs,
                       R, P);
    DMP_LAUNCH_CHECK();
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
