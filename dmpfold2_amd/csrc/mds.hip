// Top-8 symmetric eigensolver for the MDS step (reference network.py:247-250: torch.symeig,
// clamp, V*sqrt(lambda), last 8 columns), entirely on the device in float64:
//   1. Householder tridiagonalisation  A = Q T Q^T          (one launch per step, up to 256 workgroups;
//      option "tridiag_single" = 1 keeps everything in one workgroup)
//   2. bisection (Sturm counts, 64-way multisection per wave) for the 8 largest eigenvalues of T
//   3. inverse iteration with pivoted tridiagonal LU for their eigenvectors, Gram-Schmidt inside
//      clusters of close eigenvalues
//   4. back-transformation with the stored reflectors, sign rule, scaling by sqrt(max(lambda,1e-8))
// Sign rule (eigenvector signs are implementation-defined in the reference): the component of
// largest magnitude (first index on ties) is made positive.
#include "common.h"

namespace dmp {

constexpr int NEV = 8;

__global__ __launch_bounds__(256) void eig_load_kernel(const float* __restrict__ M, int n,
                                                       double* __restrict__ A) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int a = i < j ? i : j, b = i < j ? j : i;    // upper triangle, as symeig(upper=True)
  A[(int64_t)i * n + j] = (double)M[(int64_t)a * n + b];
}

__device__ __forceinline__ double block_sum_1024(double v, double* red) {
  // red: 16 doubles of LDS
  v = wave_sum_f64(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  return t;
}

// column walks with 8 independent loads in flight per thread (the plain loops are latency bound)
__device__ __forceinline__ double col_dot(const double* __restrict__ col, int n, const double* v,
                                          int i0, int i1) {
  double acc = 0.0;
  int i = i0;
  for (; i + 8 <= i1; i += 8) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = col[(int64_t)(i + u) * n];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += a[u] * v[i + u];
  }
  for (; i < i1; ++i) acc += col[(int64_t)i * n] * v[i];
  return acc;
}

__device__ __forceinline__ void col_rank2(double* __restrict__ col, int n, const double* v,
                                          const double* w, double vj, double wj, int i0, int i1) {
  int i = i0;
  for (; i + 8 <= i1; i += 8) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = col[(int64_t)(i + u) * n];
#pragma unroll
    for (int u = 0; u < 8; ++u) col[(int64_t)(i + u) * n] = a[u] - (v[i + u] * wj + w[i + u] * vj);
  }
  for (; i < i1; ++i) col[(int64_t)i * n] -= v[i] * wj + w[i] * vj;
}

// A (n x n, full symmetric storage) -> d, e, tau and the reflectors V[k][0..n-k-2] (row-wise).
// block: 1024 threads; dynamic LDS: v[n], p[n] doubles.
__global__ __launch_bounds__(1024) void tridiag_kernel(double* __restrict__ A, int n,
                                                       double* __restrict__ d, double* __restrict__ e,
                                                       double* __restrict__ tau, double* __restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* v = sm;
  double* p = sm + n;
  double* part = sm + 2 * n;           // 1024 doubles
  __shared__ double red[16];
  const int tid = threadIdx.x;
  for (int k = 0; k < n - 1; ++k) {
    const int m = n - k - 1;                       // length of x = A[k+1:, k]
    double* A22 = A + (int64_t)(k + 1) * n + (k + 1);
    // x -> v (LDS), norm of x[1:]
    double ss = 0.0;
    for (int i = tid; i < m; i += 1024) {
      const double xi = A[(int64_t)(k + 1 + i) * n + k];
      v[i] = xi;
      if (i > 0) ss += xi * xi;
    }
    const double xnorm2 = block_sum_1024(ss, red);
    const double alpha = v[0];
    double beta, tk;
    if (xnorm2 == 0.0) {
      beta = alpha;
      tk = 0.0;
    } else {
      const double nrm = sqrt(alpha * alpha + xnorm2);
      beta = alpha >= 0.0 ? -nrm : nrm;
      tk = (beta - alpha) / beta;
    }
    __syncthreads();
    if (tk != 0.0) {
      const double sc = 1.0 / (alpha - beta);
      for (int i = tid; i < m; i += 1024) v[i] = (i == 0) ? 1.0 : v[i] * sc;
    } else {
      for (int i = tid; i < m; i += 1024) v[i] = (i == 0) ? 1.0 : 0.0;
    }
    if (tid == 0) {
      d[k] = A[(int64_t)k * n + k];
      e[k] = beta;
      tau[k] = tk;
    }
    __syncthreads();
    for (int i = tid; i < m; i += 1024) V[(int64_t)k * n + i] = v[i];
    if (tk != 0.0) {
      // Thread = (column j, row group g): consecutive threads touch consecutive columns of a row
      // (coalesced), each walks its share of the rows with independent loads in flight.
      // p = tau * A22 * v, using A22 = A22^T.
      const int G = m >= 1024 ? 1 : 1024 / m;
      const int chunk = (m + G - 1) / G;
      if (m <= 1024) {
        if (tid < G * m) {
          const int j = tid % m, g = tid / m;
          const int i0 = g * chunk, i1 = (i0 + chunk < m) ? i0 + chunk : m;
          part[g * m + j] = col_dot(A22 + j, n, v, i0, i1);
        }
        __syncthreads();
        if (tid < m) {
          double sacc = 0.0;
          for (int g = 0; g < G; ++g) sacc += part[g * m + tid];
          p[tid] = tk * sacc;
        }
      } else {
        for (int j = tid; j < m; j += 1024) {
          p[j] = tk * col_dot(A22 + j, n, v, 0, m);
        }
      }
      __syncthreads();
      double pv = 0.0;
      for (int i = tid; i < m; i += 1024) pv += p[i] * v[i];
      const double dot = block_sum_1024(pv, red);
      const double a2 = -0.5 * tk * dot;
      for (int i = tid; i < m; i += 1024) p[i] += a2 * v[i];      // p now holds w
      __syncthreads();
      if (m <= 1024) {
        if (tid < G * m) {
          const int j = tid % m, g = tid / m;
          const int i0 = g * chunk, i1 = (i0 + chunk < m) ? i0 + chunk : m;
          const double vj = v[j], wj = p[j];
          col_rank2(A22 + j, n, v, p, vj, wj, i0, i1);
        }
      } else {
        for (int j = tid; j < m; j += 1024) {
          const double vj = v[j], wj = p[j];
          col_rank2(A22 + j, n, v, p, vj, wj, 0, m);
        }
      }
    }
    __syncthreads();
    __threadfence_block();
  }
  if (tid == 0) {
    d[n - 1] = A[(int64_t)(n - 1) * n + (n - 1)];
    e[n - 1] = 0.0;
  }
}

// ---- multi-workgroup tridiagonalisation: one launch per Householder step ------------------------
// Step k applies the rank-2 update of step k-1 and, in the same pass over the trailing matrix, forms
// the product with the NEW Householder vector:
//   every workgroup rebuilds w_{k-1} = p_{k-1} - (tau/2)(p.v) v and the updated pivot row (O(n) work,
//   redundantly), derives v_k, beta, tau from it, then for each of its rows r of the trailing block:
//       a_rj <- a_rj - (v_r w_j + w_r v_j),      p_k[r] = tau_k * sum_j a_rj v_k[j]
//   (a wave per row, lanes along the contiguous row).  A row's product needs no other workgroup, the
//   dot p.v that couples rows is taken by the next launch.  The trailing matrix is read and written
//   once per step by up to 256 workgroups instead of twice by one; the launches are replayed from a
//   hipGraph chain (step = k0 + node index, k0 in a device record), 1.7 us apart.
struct TriRun { int k0, n; };

__global__ void tri_set_run_kernel(TriRun* run, int k0, int n) { run->k0 = k0; run->n = n; }

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  v = wave_sum_f64(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// the same sum with ONE barrier: `red4` must be a slot that no other sum touches until another barrier has passed
__device__ __forceinline__ double block_sum_256_once(double v, double* red4) {
  v = wave_sum_f64(v);
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}

// grid: G workgroups   block: 256   dynamic LDS: 4 n doubles
// All global loads a step needs (previous reflector and product, pivot row, the wave's first row)
// are issued before the first reduction, so a launch pays one memory round trip, not four.
// NI = rows of the pivot column per thread: 256 NI >= n (5 up to order 1280, 8 up to DMP_MAX_L = 2048).
template <int NI>
__global__ __launch_bounds__(256) void tridiag_step_kernel(double* __restrict__ A, const TriRun* __restrict__ run,
                                                           int idx, double* __restrict__ d,
                                                           double* __restrict__ e, double* __restrict__ tau,
                                                           double* __restrict__ V, double* __restrict__ P) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double red[4];
  const int n = run->n, k = run->k0 + idx;
  if (k > n - 1) return;
  double* vp = sm;            // v_{k-1}, indexed from global row k
  double* w = sm + n;         // w_{k-1}
  double* vn = sm + 2 * n;    // v_k, indexed from global row k+1
  double* x = sm + 3 * n;     // updated pivot row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mp = n - k;                                   // order of the block A[k.., k..]
  double* A22p = A + (int64_t)k * n + k;
  constexpr int NJ = 5;                                   // prefetched columns of the first row: 64 * 5
  const int kp = k > 0 ? k - 1 : 0;
  const double* pp = P + (int64_t)(kp & 1) * n;
  const double tkp = k > 0 ? tau[kp] : 0.0;
  double vi[NI], pi[NI], xi[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int i = tid + 256 * u;
    const bool in = i < mp;
    vi[u] = (in && k > 0) ? V[(int64_t)kp * n + i] : 0.0;
    pi[u] = (in && k > 0) ? pp[i] : 0.0;
    xi[u] = in ? A22p[i] : 0.0;
  }
  const int r0 = 1 + blockIdx.x * 4 + wave;
  double a0[NJ];
#pragma unroll
  for (int u = 0; u < NJ; ++u) {
    const int j = 1 + lane + 64 * u;
    a0[u] = (r0 < mp && j < mp) ? A22p[(int64_t)r0 * n + j] : 0.0;
  }
  const bool upd = tkp != 0.0;
  double pv = 0.0;
#pragma unroll
  for (int u = 0; u < NI; ++u) pv += pi[u] * vi[u];
  const double a2 = upd ? -0.5 * tkp * block_sum_256(pv, red) : 0.0;
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int i = tid + 256 * u;
    if (i < mp) {
      vp[i] = upd ? vi[u] : 0.0;
      w[i] = upd ? pi[u] + a2 * vi[u] : 0.0;
    }
  }
  __syncthreads();
  // updated pivot row (local row 0), and from it the reflector of this step
  const double v0 = vp[0], w0 = w[0];
  double ss = 0.0;
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int j = tid + 256 * u;
    if (j < mp) {
      const double xj = xi[u] - (v0 * w[j] + w0 * vp[j]);
      x[j] = xj;
      if (j > 1) ss += xj * xj;
    }
  }
  const double xnorm2 = block_sum_256(ss, red);
  const int m = mp - 1;                                   // length of x[1..]
  if (m == 0) {
    if (blockIdx.x == 0 && tid == 0) { d[k] = x[0]; e[k] = 0.0; }
    return;
  }
  const double alpha = x[1];
  double beta, tk;
  if (xnorm2 == 0.0) {
    beta = alpha;
    tk = 0.0;
  } else {
    const double nrm = sqrt(alpha * alpha + xnorm2);
    beta = alpha >= 0.0 ? -nrm : nrm;
    tk = (beta - alpha) / beta;
  }
  const double sc = tk != 0.0 ? 1.0 / (alpha - beta) : 0.0;
  for (int i = tid; i < m; i += 256) vn[i] = (i == 0) ? 1.0 : x[1 + i] * sc;
  if (blockIdx.x == 0) {
    if (tid == 0) { d[k] = x[0]; e[k] = beta; tau[k] = tk; }
    for (int i = tid; i < m; i += 256) V[(int64_t)k * n + i] = (i == 0) ? 1.0 : x[1 + i] * sc;
  }
  __syncthreads();
  // rows of the trailing block: update with (v_{k-1}, w_{k-1}), product with v_k
  double* pn = P + (int64_t)(k & 1) * n;
  for (int r = r0; r < mp; r += 4 * gridDim.x) {
    double* row = A22p + (int64_t)r * n;
    const double vr = vp[r], wr = w[r];
    double acc = 0.0;
    if (r == r0) {
#pragma unroll
      for (int u = 0; u < NJ; ++u) {
        const int j = 1 + lane + 64 * u;
        if (j < mp) {
          double a = a0[u];
          if (upd) {
            a -= vr * w[j] + wr * vp[j];
            row[j] = a;
          }
          acc += a * vn[j - 1];
        }
      }
    }
    for (int j = 1 + lane + (r == r0 ? 64 * NJ : 0); j < mp; j += 64) {
      double a = row[j];
      if (upd) {
        a -= vr * w[j] + wr * vp[j];
        row[j] = a;
      }
      acc += a * vn[j - 1];
    }
    acc = wave_sum_f64(acc);
    if (lane == 0) pn[r - 1] = tk * acc;
  }
}

// ---- cluster tridiagonalisation: every Householder step in ONE launch ------------------------------------------
// TC_G workgroups of one XCD (block id % 8 == xcd); workgroup g owns the matrix rows i with i % TC_G == g and keeps
// them in LDS for the whole reduction (n / TC_G rows of n doubles); row i belongs to wave (i / TC_G) % 4.  Step k is
// the step of tridiag_step_kernel, operation for operation - the O(n) part (w_{k-1}, the updated pivot row, v_k, beta,
// tau) redundantly in every workgroup, then each wave updates its rows with (v_{k-1}, w_{k-1}) and forms their
// products with v_k, the lanes along the row with the SAME lane -> column map - so d, e, tau and the reflectors are
// the per-step launches' bit for bit.  What a launch boundary carried is handed over per step: the products
// p_k[r] by their rows' owners and the next pivot row by its owner, as 8-byte {epoch, half of a double} granules
// (parity-alternating arrays, epoch = step + 1), published with cluster_publish (plain stores that stop in the
// XCD's L2 when the cluster finds itself on one XCD - common.h) and gathered with sc1 loads.
// Round 2 measured this form at 5.8 us per step against 4.5 for the launches - with agent-scope (written-through)
// stores; the XCD-local publication is what makes it pay.
#ifndef TC_G_N
#define TC_G_N 32
#endif
constexpr int TC_G = TC_G_N;
constexpr int TC_MAX_N = 640;      // (n / 32 rows + 4 vectors) of n doubles: 123 KB of LDS at 640
static size_t tri_cluster_lds_bytes(int n) { return sizeof(double) * ((size_t)cdiv(n, TC_G) + 4) * (size_t)n; }
typedef unsigned long long tc_u64;
struct TriClusterArgs {
  const double* A;     // [n][n] symmetric, read once
  double *d, *e, *tau, *V;
  tc_u64* gx;          // [2 parity][2 vectors: p, pivot row][2 halves][n] granules + [2] placement header; zeroed before the launch
  int* abort_flag;     // DMP_FAULT_EIG_HANDOFF is set if a hand-off times out
  int n, xcd;
  int allow_local;     // 0: never the XCD-local publication
};

__device__ __forceinline__ void tc_publish(tc_u64* lo, tc_u64* hi, int idx, double v, unsigned epoch, bool local) {
  const tc_u64 bits = (tc_u64)__double_as_longlong(v), tag = (tc_u64)epoch << 32;
  cluster_publish(&lo[idx], tag | (bits & 0xffffffffull), local);
  cluster_publish(&hi[idx], tag | (bits >> 32), local);
}

// A workgroup leaves when it owns no row below the pivot any more (nobody waits for a pure consumer, so it could
// fall behind by more than the one step the two granule parities cover); d[k], e[k], tau[k] and reflector k are
// written by the owner of row k + 1, which is always still there; the owner of the last row also runs the last step.
// grid: 8 * TC_G blocks   block: 256   dynamic LDS: (rows_per_wg * n + 4 n) doubles
__global__ __launch_bounds__(256) void tridiag_cluster_kernel(TriClusterArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double red[8];
  __shared__ int sh_local, sh_abort;
  __shared__ double sh_p0;
  if ((int)(blockIdx.x & 7) != a.xcd) return;
  const int g = blockIdx.x >> 3, n = a.n;
  if (g >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrows = (n - g + TC_G - 1) / TC_G;            // rows g, g + TC_G, ...
  const int last_row = g + TC_G * (nrows - 1);
  double* vbuf[2] = {sm, sm + n};                         // v_{k-1} / v_k, alternating (indexed from global row k / k+1)
  double* w = sm + 2 * n;
  double* x = sm + 3 * n;
  double* rows = sm + 4 * n;                              // row slot t = global row g + TC_G t
  for (int t = 0; t < nrows; ++t) {
    const double* src = a.A + (int64_t)(g + TC_G * t) * n;
    for (int j = tid; j < n; j += 256) rows[(int64_t)t * n + j] = src[j];
  }
  if (tid == 0) {
    sh_abort = 0;
    sh_local = cluster_on_one_xcd(a.gx + (int64_t)8 * n, TC_G < n ? TC_G : n, a.allow_local != 0) ? 1 : 0;
  }
  __syncthreads();
  const bool local = sh_local != 0;
  constexpr int NI = 3;                                   // 256 * 3 >= 640 = largest order of this kernel
  double tkprev = 0.0;                                    // tau_{k-1}
#define TC_T(i)
  for (int k = 0; k < n; ++k) {
    if (k >= last_row && !(k == n - 1 && last_row == n - 1)) break;
    const int mp = n - k;
    double* vp = vbuf[k & 1];                             // v_{k-1}: written as v_k by the previous step
    double* vn = vbuf[(k + 1) & 1];
    // ---- p_{k-1} and the pivot row as step k-1 published them (row 0 of the matrix at the start)
    double vi[NI], pi[NI], xi[NI];
    {
      tc_u64* gp = a.gx + (int64_t)((k + 1) & 1) * 4 * n;  // parity of step k-1: [p lo][p hi][row lo][row hi]
      const unsigned epoch = (unsigned)k;                 // step k-1 published with epoch (k-1) + 1
#pragma unroll
      for (int u = 0; u < NI; ++u) {
        const int i = tid + 256 * u;
        const bool in = i < mp;
        vi[u] = (in && k > 0) ? vp[i] : 0.0;
        pi[u] = 0.0;
        xi[u] = (in && k == 0) ? a.A[i] : 0.0;
      }
      if (k > 0) {
        // all of this thread's granules in one sweep (a sweep costs an L2 round trip whatever it loads)
        tc_u64 q[NI][4];
        for (unsigned spins = 0;; ++spins) {
          bool ok = true;
#pragma unroll
          for (int u = 0; u < NI; ++u) {
            const int i = tid + 256 * u;
            if (i < mp) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                q[u][c] = __hip_atomic_load(&gp[(int64_t)c * n + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (unsigned)(q[u][c] >> 32) == epoch;
              }
            }
          }
          if (ok) break;
          if (spins > 2000000u || sh_abort) { sh_abort = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
          if (tid + 256 * u < mp) {
            pi[u] = __longlong_as_double((long long)((q[u][0] & 0xffffffffull) | (q[u][1] << 32)));
            xi[u] = __longlong_as_double((long long)((q[u][2] & 0xffffffffull) | (q[u][3] << 32)));
          }
        }
      }
    }
    TC_T(0)
    __syncthreads();                                      // also: every wave has left the previous step's row pass
    TC_T(1)
    if (sh_abort) break;
    const bool upd = tkprev != 0.0;
    double pv = 0.0;
#pragma unroll
    for (int u = 0; u < NI; ++u) pv += pi[u] * vi[u];
    if (tid == 0) sh_p0 = pi[0];                          // w[0] = p[0] + a2 v[0] is needed by every thread
    const double a2 = upd ? -0.5 * tkprev * block_sum_256_once(pv, red) : 0.0;
    if (!upd) __syncthreads();                            // sh_p0 (the sum's barrier otherwise)
    // w_{k-1} for this thread's indices, the updated pivot row x from it - one pass, no barrier in between
    const double v0 = upd ? vp[0] : 0.0;
    const double w0 = upd ? sh_p0 + a2 * v0 : 0.0;
    TC_T(2)
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int j = tid + 256 * u;
      if (j < mp) {
        const double vj = upd ? vi[u] : 0.0;              // as the launches: a step with tau = 0 contributes nothing
        const double wj = upd ? pi[u] + a2 * vi[u] : 0.0;
        if (!upd) vp[j] = 0.0;
        w[j] = wj;
        const double xj = xi[u] - (v0 * wj + w0 * vj);
        x[j] = xj;
        if (j > 1) ss += xj * xj;
      }
    }
    const double xnorm2 = block_sum_256_once(ss, red + 4);
    const int m = mp - 1;
    if (m == 0) {
      if (tid == 0) { a.d[k] = x[0]; a.e[k] = 0.0; }
      break;
    }
    const bool writer = ((k + 1) % TC_G) == g;            // the owner of row k + 1
    const double alpha = x[1];
    double beta, tk;
    if (xnorm2 == 0.0) {
      beta = alpha;
      tk = 0.0;
    } else {
      const double nrm = sqrt(alpha * alpha + xnorm2);
      beta = alpha >= 0.0 ? -nrm : nrm;
      tk = (beta - alpha) / beta;
    }
    const double sc = tk != 0.0 ? 1.0 / (alpha - beta) : 0.0;
    for (int i = tid; i < m; i += 256) vn[i] = (i == 0) ? 1.0 : x[1 + i] * sc;
    if (writer) {
      if (tid == 0) { a.d[k] = x[0]; a.e[k] = beta; a.tau[k] = tk; }
      for (int i = tid; i < m; i += 256) a.V[(int64_t)k * n + i] = (i == 0) ? 1.0 : x[1 + i] * sc;
    }
    __syncthreads();
    tkprev = tk;
    TC_T(3)
    // ---- this workgroup's rows of the trailing block: update with (v_{k-1}, w_{k-1}), product with v_k, publish
    tc_u64* gq = a.gx + (int64_t)(k & 1) * 4 * n;
    const unsigned epoch = (unsigned)k + 1u;
    const int t_lo = k >= g ? (k - g) / TC_G + 1 : 0;     // first slot whose row lies below the pivot
    const int t_w = t_lo + ((wave - t_lo) & 3);           // this wave's first one (slot t belongs to wave t % 4)
    for (int t = t_w; t < nrows; t += 8) {                // two rows at a time: their LDS latencies overlap
      // (three rows with the operands of five column chunks loaded ahead measured slower: 1.06 against 0.91 ms at n = 300)
      const bool two = t + 4 < nrows;
      const int r0 = g + TC_G * t - k, r1 = two ? r0 + 4 * TC_G : r0;     // local row indices, >= 1
      double* row0 = rows + (int64_t)t * n + k;           // local column j at row[j]
      double* row1 = two ? row0 + (int64_t)4 * n : row0;
      const double vr0 = vp[r0], wr0 = w[r0], vr1 = vp[r1], wr1 = w[r1];
      double acc0 = 0.0, acc1 = 0.0;
      for (int j = 1 + lane; j < mp; j += 64) {
        const double wj = w[j], vj = vp[j], vnj = vn[j - 1];
        double a0 = row0[j], a1 = row1[j];
        if (upd) {
          a0 -= vr0 * wj + wr0 * vj;
          row0[j] = a0;
          if (two) {
            a1 -= vr1 * wj + wr1 * vj;
            row1[j] = a1;
          }
        }
        acc0 += a0 * vnj;
        acc1 += a1 * vnj;
        if (r0 == 1) tc_publish(gq + 2 * n, gq + 3 * n, j - 1, a0, epoch, local);   // the next pivot row
      }
      acc0 = wave_sum_f64(acc0);
      if (two) acc1 = wave_sum_f64(acc1);
      if (lane == 0) {
        tc_publish(gq, gq + n, r0 - 1, tk * acc0, epoch, local);
        if (two) tc_publish(gq, gq + n, r1 - 1, tk * acc1, epoch, local);
      }
    }
    TC_T(4)
  }
  if (sh_abort && tid == 0) atomicOr(a.abort_flag, DMP_FAULT_EIG_HANDOFF);
}

// 1/b to float64 rounding error without the IEEE division sequence (v_div_scale/fmas/fixup is a
// chain of about ten dependent instructions): hardware reciprocal + two Newton steps.  The callers
// keep |b| away from zero and infinity (pivmin / tiny guards), and the Sturm recurrence and the
// triangular solves are latency chains of one division per row.
__device__ __forceinline__ double fast_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ int sturm_count(const double* d, const double* e2, int n, double x,
                                           double pivmin) {
  double q = d[0] - x;
  if (fabs(q) < pivmin) q = -pivmin;
  int cnt = q < 0.0;
  for (int i = 1; i < n; ++i) {
    q = d[i] - x - e2[i - 1] * fast_rcp(q);
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < 0.0;
  }
  return cnt;
}

// Bisection + inverse iteration on the tridiagonal matrix.  block: 512 (8 waves: one eigenvalue each).
// dynamic LDS: dd[n], e2[n], ee[n] doubles, then by MODE (chosen by the order, tri_eig_lds_bytes):
//   0 (n <= 384): x[n][8] and the five LU factor arrays [n][8] (408 n bytes): the two triangular solves of an
//                 iteration read nothing but LDS (from L2 every 8 steps of the recurrence paid a round trip:
//                 0.3 of the kernel's 0.66 ms at n = 300);
//   1 (n <= 1280): x[n][8] in LDS, the factors in the global scratch F (5 arrays [n][8]);
//   2 (larger): x is Z itself (global), only the 3 n doubles the Sturm counts read stay in LDS.
// Outputs: lam[8] (ascending), Z[n][8] (unit eigenvectors of T).
template <int MODE>
__global__ __launch_bounds__(512) void tri_eig_kernel(const double* __restrict__ d,
                                                      const double* __restrict__ e, int n,
                                                      double* __restrict__ F,
                                                      double* __restrict__ lam_out,
                                                      double* __restrict__ Z) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* dd = sm;
  double* e2 = sm + n;
  double* ee = sm + 2 * n;             // off-diagonal (the LU recurrence reads it n times per lane)
  double* x = MODE == 2 ? Z : sm + 3 * n;   // [n][8]
  if (MODE == 0) F = sm + (3 + NEV) * n;
  __shared__ double lam[NEV];
  __shared__ double sh_scal[4];
  __shared__ double red[8][NEV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#define TE_T(i)

  // Gershgorin interval and norms
  double lo = 1e300, hi = -1e300, emax = 0.0, nrm1 = 0.0;
  for (int i = tid; i < n; i += 512) {
    const double di = d[i];
    const double el = i > 0 ? fabs(e[i - 1]) : 0.0, er = i < n - 1 ? fabs(e[i]) : 0.0;
    dd[i] = di;
    e2[i] = (i < n - 1) ? e[i] * e[i] : 0.0;
    ee[i] = (i < n - 1) ? e[i] : 0.0;
    lo = fmin(lo, di - el - er);
    hi = fmax(hi, di + el + er);
    emax = fmax(emax, er);
    nrm1 = fmax(nrm1, fabs(di) + el + er);
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, off, 64));
    hi = fmax(hi, __shfl_xor(hi, off, 64));
    emax = fmax(emax, __shfl_xor(emax, off, 64));
    nrm1 = fmax(nrm1, __shfl_xor(nrm1, off, 64));
  }
  if (lane == 0) { red[wave][0] = lo; red[wave][1] = hi; red[wave][2] = emax; red[wave][3] = nrm1; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 8; ++w) {
      red[0][0] = fmin(red[0][0], red[w][0]);
      red[0][1] = fmax(red[0][1], red[w][1]);
      red[0][2] = fmax(red[0][2], red[w][2]);
      red[0][3] = fmax(red[0][3], red[w][3]);
    }
    const double span = red[0][1] - red[0][0];
    sh_scal[0] = red[0][0] - 1e-12 * fabs(span) - 1e-300;
    sh_scal[1] = red[0][1] + 1e-12 * fabs(span) + 1e-300;
    sh_scal[2] = 1e-290 * fmax(1.0, red[0][2] * red[0][2]);   // pivmin
    sh_scal[3] = red[0][3];                                    // ||T||_1
  }
  __syncthreads();
  const double glo = sh_scal[0], ghi = sh_scal[1], pivmin = sh_scal[2], tnorm = sh_scal[3];

  TE_T(0)
  // ---- bisection: wave w finds eigenvalue w of the top 8 (ascending order); a Sturm count is a
  // chain of n float64 divisions, so the eight counts run side by side
  {
    const int ev = wave;
    const int idx = n - NEV + ev;           // 0-based index in the ascending spectrum
    double a = glo, b = ghi;
    for (int round = 0; round < 14; ++round) {
      const double xt = a + (b - a) * ((double)(lane + 1) / 65.0);
      const int cnt = sturm_count(dd, e2, n, xt, pivmin);
      const unsigned long long above = __ballot(cnt >= idx + 1);
      // counts are monotone in x: the lanes with cnt >= idx+1 form a suffix
      const int first = above ? __ffsll((long long)above) - 1 : 64;
      const double na = first > 0 ? __shfl(xt, first - 1, 64) : a;
      const double nb = first < 64 ? __shfl(xt, first < 64 ? first : 63, 64) : b;
      a = na;
      b = nb;
      if (b - a <= 4.0e-16 * fmax(fabs(a), fabs(b)) + 2.0 * pivmin) break;
    }
    if (lane == 0) lam[ev] = 0.5 * (a + b);
  }
  __syncthreads();

  TE_T(1)
  // ---- inverse iteration: lanes 0..7 of wave 0, one eigenvalue each, in lockstep
  const double eps = 2.220446049250313e-16;
  const double ortol = 1e-3 * tnorm;
  double* fa = F;                      // U diagonal
  double* fb = F + (int64_t)n * NEV;   // U first super-diagonal
  double* fc = F + (int64_t)2 * n * NEV;  // multipliers
  double* fd = F + (int64_t)3 * n * NEV;  // U second super-diagonal
  double* fp = F + (int64_t)4 * n * NEV;  // interchange flags
  if (tid == 0) {
    // separate coincident eigenvalues slightly so the factorizations differ
    for (int j = 1; j < NEV; ++j) {
      const double pert = 10.0 * eps * fmax(fabs(lam[j]), tnorm * 1e-3);
      if (lam[j] - lam[j - 1] < pert) lam[j] = lam[j - 1] + pert;
    }
  }
  __syncthreads();
  if (tid < NEV) {
    const double l = lam[tid];
    const double tiny = eps * fmax(tnorm, 1e-300);
    // LU with partial pivoting of T - l*I (rows k, k+1 at a time)
    // (the stored U diagonal is its guarded reciprocal: the back substitution multiplies)
    auto guarded_rcp = [&](double a) {
      if (fabs(a) < tiny) a = a < 0.0 ? -tiny : tiny;
      return fast_rcp(a);
    };
    double ak = dd[0] - l;                                   // current diagonal of row k
    double bk = n > 1 ? ee[0] : 0.0;                         // current super-diagonal of row k
    for (int k = 0; k < n - 1; ++k) {
      const double ck = ee[k];                               // sub-diagonal entry (row k+1, col k)
      const double a1 = dd[k + 1] - l;                       // row k+1 diagonal
      const double b1 = (k < n - 2) ? ee[k + 1] : 0.0;       // row k+1 super-diagonal
      double ua, ub, ud, mult, na1, nb1;
      double flag;
      if (fabs(ck) <= fabs(ak)) {
        flag = 0.0;
        mult = (ak != 0.0) ? ck * fast_rcp(ak) : 0.0;
        ua = ak; ub = bk; ud = 0.0;
        na1 = a1 - mult * bk;
        nb1 = b1;
      } else {
        flag = 1.0;
        mult = ak * fast_rcp(ck);
        ua = ck; ub = a1; ud = b1;
        na1 = bk - mult * a1;
        nb1 = -mult * b1;
      }
      ua = guarded_rcp(ua);
      fa[(int64_t)k * NEV + tid] = ua;
      fb[(int64_t)k * NEV + tid] = ub;
      fd[(int64_t)k * NEV + tid] = ud;
      fc[(int64_t)k * NEV + tid] = mult;
      fp[(int64_t)k * NEV + tid] = flag;
      ak = na1;
      bk = nb1;
    }
    fa[(int64_t)(n - 1) * NEV + tid] = guarded_rcp(ak);
    fb[(int64_t)(n - 1) * NEV + tid] = 0.0;
    fd[(int64_t)(n - 1) * NEV + tid] = 0.0;
    // start vector: deterministic, no special structure
    unsigned s = 0x9E3779B9u * (unsigned)(tid + 1);
    for (int i = 0; i < n; ++i) {
      s = s * 1664525u + 1013904223u;
      x[i * NEV + tid] = ((double)(s >> 8) / 16777216.0) - 0.5;
    }
    (void)tiny;
  }
  if (MODE == 2) __threadfence();      // x lives in global memory: lanes read what other lanes wrote
  __syncthreads();
  TE_T(2)
  // From here on only wave 0 works (lanes 0..7 solve, then all 64 lanes orthonormalise with wave
  // shuffles): no workgroup barrier inside the iteration loop.
  if (wave == 0) {
    for (int iter = 0; iter < 5; ++iter) {
      if (lane < NEV) {
        // forward substitution with the recorded interchanges.  The factors live in global memory
        // (L2): they are fetched eight steps at a time ahead of the recurrence, which only touches LDS.
        double yk = x[lane];
        for (int k0 = 0; k0 < n - 1; k0 += 8) {
          double mu[8], fl[8], xs[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + u < n - 1 ? k0 + u : n - 2;
            mu[u] = fc[(int64_t)k * NEV + lane];
            fl[u] = fp[(int64_t)k * NEV + lane];
            xs[u] = x[(k + 1) * NEV + lane];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + u;
            if (k < n - 1) {
              const double y1 = xs[u];
              if (fl[u] == 0.0) {
                x[k * NEV + lane] = yk;
                yk = y1 - mu[u] * yk;
              } else {
                x[k * NEV + lane] = y1;
                yk = yk - mu[u] * y1;
              }
            }
          }
        }
        x[(n - 1) * NEV + lane] = yk;
        // back substitution
        double x1 = 0.0, x2 = 0.0;
        for (int k0 = n - 1; k0 >= 0; k0 -= 8) {
          double ua[8], ub[8], ud[8], xs[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 - u >= 0 ? k0 - u : 0;
            ua[u] = fa[(int64_t)k * NEV + lane];        // reciprocal of the (guarded) U diagonal
            ub[u] = fb[(int64_t)k * NEV + lane];
            ud[u] = fd[(int64_t)k * NEV + lane];
            xs[u] = x[k * NEV + lane];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 - u;
            if (k >= 0) {
              const double t = (xs[u] - ub[u] * x1 - ud[u] * x2) * ua[u];
              x[k * NEV + lane] = t;
              x2 = x1;
              x1 = t;
            }
          }
        }
      }
      if (MODE == 2) __threadfence();
      __builtin_amdgcn_wave_barrier();
      TE_T(3)
      // modified Gram-Schmidt inside clusters + normalisation
      for (int j = 0; j < NEV; ++j) {
        for (int i = j - 1; i >= 0; --i) {
          if (lam[i + 1] - lam[i] > ortol) break;          // cluster chain ends
          double dot = 0.0;
          for (int r = lane; r < n; r += 64) dot += x[r * NEV + i] * x[r * NEV + j];
          dot = wave_sum_f64(dot);
          for (int r = lane; r < n; r += 64) x[r * NEV + j] -= dot * x[r * NEV + i];
        }
        double nn = 0.0;
        for (int r = lane; r < n; r += 64) nn += x[r * NEV + j] * x[r * NEV + j];
        nn = wave_sum_f64(nn);
        const double nrm = sqrt(nn);
        const double sc = nrm > 0.0 ? 1.0 / nrm : 0.0;
        for (int r = lane; r < n; r += 64) x[r * NEV + j] *= sc;
        if (MODE == 2) __threadfence();                    // the next column's dot products read this one
      }
      __builtin_amdgcn_wave_barrier();
      TE_T(4)
    }
  }
  __syncthreads();
  if (MODE != 2)
    for (int r = tid; r < n * NEV; r += 512) Z[r] = x[r];
  if (tid < NEV) lam_out[tid] = lam[tid];
}

// Z <- Q Z with Q = H_0 H_1 ... H_{n-2}; then sign rule and scaling.
// One wave per eigenvector column, the column in registers (row i = lane + 64 t): applying a
// reflector is a dot product and an axpy inside the wave - shuffles, no workgroup barrier (the
// previous version spent its 1.0 ms at L = 300 in 600 barriers).  The reflector rows are fetched G
// steps ahead.  block: 512 (8 waves = 8 columns); T = ceil(n / 64) rows per lane.
template <int T, int G>
__global__ __launch_bounds__(512) void backtransform_kernel(const double* __restrict__ V,
                                                            const double* __restrict__ tau,
                                                            const double* __restrict__ lam,
                                                            const double* __restrict__ Z, int n,
                                                            float* __restrict__ mds) {
  const int lane = threadIdx.x & 63;
  const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double z[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = lane + 64 * t;
    z[t] = i < n ? Z[(int64_t)i * NEV + c] : 0.0;
  }
  for (int kg = n - 2; kg >= 0; kg -= G) {
    double vv[G][T];
    double tk[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k = kg - g;
      tk[g] = k >= 0 ? tau[k] : 0.0;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int ip = lane + 64 * t - (k + 1);          // index into reflector k (acts on rows k+1..)
        vv[g][t] = (k >= 0 && ip >= 0 && lane + 64 * t < n) ? V[(int64_t)k * n + ip] : 0.0;
      }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (tk[g] == 0.0) continue;                        // uniform
      double part = 0.0;
#pragma unroll
      for (int t = 0; t < T; ++t) part += vv[g][t] * z[t];
      part = wave_sum_f64(part);
      const double s = tk[g] * part;
#pragma unroll
      for (int t = 0; t < T; ++t) z[t] -= s * vv[g][t];
    }
  }
  // sign rule: largest |component| positive, first index on ties
  double bv = -1.0;
  int bi = 0x7fffffff;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = lane + 64 * t;
    const double a = fabs(z[t]);
    if (i < n && a > bv) { bv = a; bi = i; }
  }
  double bz = 0.0;
#pragma unroll
  for (int t = 0; t < T; ++t)
    if (lane + 64 * t == bi) bz = z[t];
  for (int off = 1; off < 64; off <<= 1) {
    const double ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    const double oz = __shfl_xor(bz, off, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bz = oz; }
  }
  const double sgn = bz < 0.0 ? -1.0 : 1.0;
  const float lf = (float)lam[c];
  const float scale = sqrtf(fmaxf(fmaxf(lf, 0.0f), 1e-8f));
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int i = lane + 64 * t;
    if (i < n) mds[(int64_t)i * NEV + c] = (float)(sgn * z[t]) * scale;
  }
}

constexpr int TRI_CHAIN = 64;     // Householder steps per graph replay

// chain of TRI_CHAIN step nodes for matrices of order n (pointers and LDS size are baked into the nodes)
static int tridiag_graph(dmp_ctx* c, int n, double* A, TriRun* run, double* d, double* e, double* tau,
                         double* V, double* P, hipGraphExec_t* out) {
  auto it = c->tri_graphs.find(n);
  if (it != c->tri_graphs.end()) { *out = (hipGraphExec_t)it->second; return DMP_OK; }
  const int grid = std::min(256, cdiv(n, 4));
  const TriRun* crun = run;
  hipGraph_t g;
  DMP_HIP(hipGraphCreate(&g, 0));
  hipGraphNode_t prev = nullptr;
  for (int idx = 0; idx < TRI_CHAIN; ++idx) {
    int idx_arg = idx;
    void* params[8] = {(void*)&A, (void*)&crun, (void*)&idx_arg, (void*)&d, (void*)&e, (void*)&tau, (void*)&V,
                       (void*)&P};
    hipKernelNodeParams kp{};
    kp.func = n <= 1280 ? (void*)tridiag_step_kernel<5> : (void*)tridiag_step_kernel<8>;
    kp.gridDim = dim3(grid);
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = (unsigned)(sizeof(double) * 4 * n);
    kp.kernelParams = params;
    kp.extra = nullptr;
    hipGraphNode_t node;
    hipError_t err = hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
    if (err != hipSuccess) { (void)hipGraphDestroy(g); return hip_fail(err, "hipGraphAddKernelNode", __FILE__, __LINE__); }
    prev = node;
  }
  hipGraphExec_t ge;
  hipError_t err = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (err != hipSuccess) return hip_fail(err, "hipGraphInstantiate", __FILE__, __LINE__);
  c->tri_graphs[n] = (void*)ge;
  *out = ge;
  return DMP_OK;
}

// Dynamic LDS of tri_eig_kernel by order: see its MODE comment.
static int tri_eig_mode(int n) { return n <= 384 ? 0 : (n <= 1280 ? 1 : 2); }
static size_t tri_eig_lds_bytes(int n) {
  const int mode = tri_eig_mode(n);
  return sizeof(double) * (size_t)n * (mode == 0 ? 3 + 6 * NEV : (mode == 1 ? 3 + NEV : 3));
}

// Above the 64 KB a kernel may use without the opt-in: tri_eig_kernel (157 KB at order 384 in mode 0, 113 KB at 1280
// in mode 1) and the step kernel of the large orders (4 n doubles = 64 KB at 2048).  Set once per device at context
// creation (see trunk_kernel_attrs for why never between launches).
int mds_kernel_attrs(dmp_ctx* c) {
  static bool done[64] = {};
  if (c->device >= 0 && c->device < 64 && done[c->device]) return DMP_OK;
  DMP_HIP(hipFuncSetAttribute((const void*)tri_eig_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)tri_eig_lds_bytes(384)));
  DMP_HIP(hipFuncSetAttribute((const void*)tri_eig_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)tri_eig_lds_bytes(1280)));
  DMP_HIP(hipFuncSetAttribute((const void*)tri_eig_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)tri_eig_lds_bytes(DMP_MAX_L)));
  DMP_HIP(hipFuncSetAttribute((const void*)tridiag_step_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(double) * 4 * DMP_MAX_L)));
  DMP_HIP(hipFuncSetAttribute((const void*)tridiag_cluster_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)tri_cluster_lds_bytes(TC_MAX_N)));
  if (c->device >= 0 && c->device < 64) done[c->device] = true;
  return DMP_OK;
}

int eigh_top8(dmp_ctx* c, const float* d_M, int L, float* d_mds, hipStream_t s) {
  const int n = L;
  double* A = c->eig_a;
  double* ws = c->eig_ws;
  double* d = ws;
  double* e = ws + n;
  double* tau = ws + 2 * n;
  double* lam = ws + 3 * n;                 // 8
  double* Z = ws + 3 * n + 16;              // n*8
  double* F = Z + (int64_t)n * NEV;         // 5*n*8
  double* V = F + (int64_t)5 * n * NEV;     // n*n
  hipLaunchKernelGGL(eig_load_kernel, dim3(cdiv(n, 256), n), dim3(256), 0, s, d_M, n, A);
  DMP_LAUNCH_CHECK();
  double* P = V + (int64_t)n * n;           // 2*n
  TriRun* run = reinterpret_cast<TriRun*>(P + 2 * n);
  if (c->tridiag_single) {
    hipLaunchKernelGGL(tridiag_kernel, dim3(1), dim3(1024), sizeof(double) * (2 * n + 1024), s, A, n, d,
                       e, tau, V);
    DMP_LAUNCH_CHECK();
  } else if (c->tridiag_cluster && n <= TC_MAX_N) {
    TriClusterArgs ta{};
    ta.A = A; ta.d = d; ta.e = e; ta.tau = tau; ta.V = V;
    ta.gx = (tc_u64*)c->tri_gx; ta.abort_flag = c->seq_abort; ta.n = n; ta.xcd = (c->refine_xcd + 4) & 7; ta.allow_local = c->cluster_local;
    CoResident guard(c, s, false);                     // not beside a persistent vertical-GRU launch (common.h)
    if (guard.status()) return guard.status();
    DMP_HIP(hipMemsetAsync(c->tri_gx, 0, sizeof(tc_u64) * (8 * (size_t)n + 2), s));
    hipLaunchKernelGGL(tridiag_cluster_kernel, dim3(8 * TC_G), dim3(256), tri_cluster_lds_bytes(n), s, ta);
    DMP_LAUNCH_CHECK();
    int grc = guard.done();
    if (grc) return grc;
  } else {
    hipGraphExec_t ge;
    int rc = tridiag_graph(c, n, A, run, d, e, tau, V, P, &ge);
    if (rc) return rc;
    for (int k0 = 0; k0 < n; k0 += TRI_CHAIN) {
      hipLaunchKernelGGL(tri_set_run_kernel, dim3(1), dim3(1), 0, s, run, k0, n);
      DMP_HIP(hipGraphLaunch(ge, s));
    }
    DMP_LAUNCH_CHECK();
  }
  static_assert(DMP_MAX_L <= 2048, "tridiag_step_kernel<8> / backtransform_kernel<32, 1> cover orders up to 2048");
  switch (tri_eig_mode(n)) {
    case 0: hipLaunchKernelGGL(tri_eig_kernel<0>, dim3(1), dim3(512), tri_eig_lds_bytes(n), s, d, e, n, F, lam, Z); break;
    case 1: hipLaunchKernelGGL(tri_eig_kernel<1>, dim3(1), dim3(512), tri_eig_lds_bytes(n), s, d, e, n, F, lam, Z); break;
    default: hipLaunchKernelGGL(tri_eig_kernel<2>, dim3(1), dim3(512), tri_eig_lds_bytes(n), s, d, e, n, F, lam, Z);
  }
  DMP_LAUNCH_CHECK();
  if (n <= 320)
    hipLaunchKernelGGL((backtransform_kernel<5, 4>), dim3(1), dim3(512), 0, s, V, tau, lam, Z, n, d_mds);
  else if (n <= 640)
    hipLaunchKernelGGL((backtransform_kernel<10, 2>), dim3(1), dim3(512), 0, s, V, tau, lam, Z, n, d_mds);
  else if (n <= 1280)
    hipLaunchKernelGGL((backtransform_kernel<20, 1>), dim3(1), dim3(512), 0, s, V, tau, lam, Z, n, d_mds);
  else if (n <= 2048)
    hipLaunchKernelGGL((backtransform_kernel<32, 1>), dim3(1), dim3(512), 0, s, V, tau, lam, Z, n, d_mds);
  else { set_error("eigh_top8: order %d exceeds %d", n, DMP_MAX_L); return DMP_ERR_CAPACITY; }
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
