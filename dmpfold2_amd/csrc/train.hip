// Training-side slice (reference train.py:318-344 runs autograd through every ResNet_Block,
// network.py:85-103): the backward of ONE residual block as two entry points that mirror the forward's two kernels,
//   dmp_block_norm_scse_residual_bwd   d(out) -> d(u), d(gamma, beta, cSE fc, sSE conv)      (this file, second half)
//   dmp_block_conv5x5_maxout_bwd       d(u)   -> d(x), d(W), d(b)                            (this file, first half)
// and the residual branch is the identity: d(block input) = d(x) of the convolution + d(out).
// Parity-first, float32 arithmetic with float64 reductions; none of this is on the inference path (the workspace is
// allocated on first use, outside the "no allocation after dmp_ctx_create" rule of the prediction entry points).
#include "common.h"

#pragma clang fp contract(off)

namespace dmp {

static int bwd_workspace(dmp_ctx* c, int64_t need_floats) {
  if (c->bwd_ws_floats >= need_floats) return DMP_OK;
  if (c->bwd_ws) (void)hipFree(c->bwd_ws);
  c->bwd_ws = nullptr;
  c->bwd_ws_floats = 0;
  DMP_HIP(hipMalloc((void**)&c->bwd_ws, sizeof(float) * (size_t)need_floats));
  c->bwd_ws_floats = need_floats;
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// First half (network.py:25-31): backward of a block's convolution + maxout,  u[g] = max_q (conv(x, W)[4g+q] + b[4g+q]).
//   dz[4g+q] = du[g] where q is the FIRST maximal channel of the quadruple (torch.max), 0 elsewhere
//   dW[o][c,tap] = sum_p dz[o][p] x[c][p + tap]        (wgrad: GEMM over the pixels)
//   db[o]        = sum_p dz[o][p]
//   dx[c][p]     = sum_{o,tap} W[o][c,tap] dz[o][p - tap]   (dgrad: GEMM + gather)
// Parity-first form: the 25-tap patches as an explicit matrix (im2col, 3200 x L^2) and three products on the float32
// matrix cores (gemm_f32: exact fmaf chains); the winners are recomputed with the same float32 product, so a
// near-tie can resolve differently from the inference kernels' split-f16 sums (measure-zero for the gradients' checks).
// Workspace (3200 + 1024) x L^2 floats, allocated on first use (this entry point is outside the inference path's
// "no allocation after dmp_ctx_create" rule).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bwd_im2col_kernel(const float* __restrict__ x, int L, float* __restrict__ col) {
  const int r = blockIdx.y;                              // c * 25 + tap
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int LL = L * L;
  if (p >= LL) return;
  const int c = r / 25, tap = r % 25, dy = tap / 5 - 2, dx = tap % 5 - 2;
  const int y = p / L + dy, xx = p % L + dx;
  col[(int64_t)r * LL + p] = (y >= 0 && y < L && xx >= 0 && xx < L) ? x[(int64_t)c * LL + y * L + xx] : 0.f;
}

__global__ __launch_bounds__(256) void bwd_maxout_route_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                               const float* __restrict__ du, int LL,
                                                               float* __restrict__ dz) {
  const int g = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= LL) return;
  int win = 0;
  float best = z[(int64_t)(4 * g) * LL + p] + bias[4 * g];
#pragma unroll
  for (int q = 1; q < 4; ++q) {
    const float v = z[(int64_t)(4 * g + q) * LL + p] + bias[4 * g + q];
    if (v > best) { best = v; win = q; }                 // strict: the first maximal value wins, as torch.max
  }
  const float d = du[(int64_t)g * LL + p];
#pragma unroll
  for (int q = 0; q < 4; ++q) dz[(int64_t)(4 * g + q) * LL + p] = q == win ? d : 0.f;
}

__global__ __launch_bounds__(256) void bwd_bias_kernel(const float* __restrict__ dz, int LL, float* __restrict__ db) {
  __shared__ double red[256];
  const int o = blockIdx.x;
  double acc = 0.0;
  for (int p = threadIdx.x; p < LL; p += 256) acc += (double)dz[(int64_t)o * LL + p];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[o] = (float)red[0];
}

__global__ __launch_bounds__(256) void bwd_col2im_kernel(const float* __restrict__ dcol, int L, float* __restrict__ dx) {
  const int c = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int LL = L * L;
  if (p >= LL) return;
  const int y = p / L, x = p % L;
  float acc = 0.f;
#pragma unroll
  for (int tap = 0; tap < 25; ++tap) {
    const int yy = y - (tap / 5 - 2), xx = x - (tap % 5 - 2);      // the output pixel whose patch holds (y, x) at `tap`
    if (yy >= 0 && yy < L && xx >= 0 && xx < L) acc += dcol[(int64_t)(c * 25 + tap) * LL + yy * L + xx];
  }
  dx[(int64_t)c * LL + p] = acc;
}

// the block's weights as the state_dict holds them, W[o][c*25 + tap], from the exact-f32 kernel's pack
// [split 4][chunk 64][tap 25][cc 2][m 128] (the raw tensors are not kept after dmp_weights_finalize)
__global__ __launch_bounds__(256) void bwd_unpack_weights_kernel(const float* __restrict__ wpack, float* __restrict__ w) {
  const int i = blockIdx.x * 256 + threadIdx.x;           // o * 3200 + c * 25 + tap
  if (i >= 512 * 3200) return;
  const int o = i / 3200, r = i % 3200, cch = r / 25, tap = r % 25;
  w[i] = wpack[((((int64_t)(o >> 7) * (CW / CONV_CC) + cch / CONV_CC) * 25 + tap) * CONV_CC + cch % CONV_CC) * 128 + (o & 127)];
}

int conv5x5_maxout_bwd(dmp_ctx* c, int block, const float* d_x, const float* d_du, int L, float* d_dx, float* d_dw,
                       float* d_db, hipStream_t s) {
  const BlockW& B = c->W.blk[block - 1];
  const int LL = L * L;
  int rc;
  if ((rc = bwd_workspace(c, (int64_t)(3200 + 1024) * LL))) return rc;
  if (!c->bwd_w) {                                       // the block weights as uploaded (512 x 3200), once per context
    DMP_HIP(hipMalloc((void**)&c->bwd_w, sizeof(float) * 512 * 3200));
    c->bwd_w_block = 0;
  }
  if (c->bwd_w_block != block) {
    hipLaunchKernelGGL(bwd_unpack_weights_kernel, dim3(cdiv(512 * 3200, 256)), dim3(256), 0, s, B.wpack, c->bwd_w);
    DMP_LAUNCH_CHECK();
    c->bwd_w_block = block;
  }
  float* col = c->bwd_ws;
  float* z = col + (int64_t)3200 * LL;
  float* dz = z + (int64_t)512 * LL;
  hipLaunchKernelGGL(bwd_im2col_kernel, dim3(cdiv(LL, 256), 3200), dim3(256), 0, s, d_x, L, col);
  DMP_LAUNCH_CHECK();
  GemmArgs g{};
  // z = W col
  g.A = c->bwd_w; g.sam = 3200; g.sak = 1; g.B = col; g.sbk = LL; g.sbn = 1; g.C = z; g.ldc = LL;
  g.M = 512; g.N = LL; g.K = 3200; g.alpha = 1.f; g.beta = 0.f; g.bias_n = nullptr;
  if ((rc = gemm_f32(g, s))) return rc;
  hipLaunchKernelGGL(bwd_maxout_route_kernel, dim3(cdiv(LL, 256), 128), dim3(256), 0, s, z, B.bias, d_du, LL, dz);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(bwd_bias_kernel, dim3(512), dim3(256), 0, s, dz, LL, d_db);
  DMP_LAUNCH_CHECK();
  // dW = dz col^T
  g.A = dz; g.sam = LL; g.sak = 1; g.B = col; g.sbk = 1; g.sbn = LL; g.C = d_dw; g.ldc = 3200;
  g.M = 512; g.N = 3200; g.K = LL;
  if ((rc = gemm_f32(g, s))) return rc;
  // dcol = W^T dz (over the patch matrix, which the weight gradient no longer needs)
  g.A = c->bwd_w; g.sam = 1; g.sak = 3200; g.B = dz; g.sbk = LL; g.sbn = 1; g.C = col; g.ldc = LL;
  g.M = 3200; g.N = LL; g.K = 512;
  if ((rc = gemm_f32(g, s))) return rc;
  hipLaunchKernelGGL(bwd_col2im_kernel, dim3(cdiv(LL, 256), 128), dim3(256), 0, s, col, L, d_dx);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}


// ---------------------------------------------------------------------------------------
// Second half (network.py:32, 36-83, 99-101): backward of InstanceNorm + scSE + residual add,
//   v = gamma uh + beta, uh = (u - mean) rstd;   out = v (y + s) + x,   y_c = sigma(W2 relu(W1 m)), m = mean_p v,
//   s_p = sigma(a_p), a_p = sum_c ws_c v_cp + bs.
// With D = d(out):
//   d(s)_p = sum_c D v,  d(a) = d(s) s (1 - s);   d(y)_c = sum_p D v -> the two small fc layers -> d(m)
//   d(v) = D (y + s) + ws d(a) + d(m) / P
//   d(beta) = sum_p d(v), d(gamma) = sum_p d(v) uh,  d(u) = gamma rstd (d(v) - mean d(v) - uh mean(d(v) uh))
// (the d(m) / P term is constant per channel: it reaches d(beta) and cancels in d(u), as m = beta does not depend on u).
// Every channel sum is linear in seven pixel sums, so the activations are read three times (pixel gates, channel
// sums, d(u)) and d(v) is never stored:
//   T1 = sum D, T2 = sum D uh, T3 = sum D s, T4 = sum D s uh, T5 = sum d(a) uh, T6 = sum d(a), T7 = sum uh
// ---------------------------------------------------------------------------------------
constexpr int TB_CHUNK = 4096;                           // pixels per workgroup of the reduction kernels
constexpr int TB_NSUM = 8;

template <int NS>
__device__ __forceinline__ void tb_block_sums(double (&v)[NS], double* __restrict__ out) {
  __shared__ double red[4][NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) v[k] = wave_sum_f64(v[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NS; ++k) red[wave][k] = v[k];
  __syncthreads();
  if (threadIdx.x < NS) out[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// grid: (chunks, 128)   block: 256
__global__ __launch_bounds__(256) void tb_stats_kernel(const float* __restrict__ u, int LL, double* __restrict__ part) {
  const int c = blockIdx.y, nch = gridDim.x;
  const int p1 = min(LL, (int)(blockIdx.x + 1) * TB_CHUNK);
  double v[2] = {0.0, 0.0};
  for (int p = blockIdx.x * TB_CHUNK + threadIdx.x; p < p1; p += 256) {
    const double x = (double)u[(int64_t)c * LL + p];
    v[0] += x;
    v[1] += x * x;
  }
  tb_block_sums<2>(v, part + ((int64_t)c * nch + blockIdx.x) * 2);
}

// coef[c] = {alpha = gamma rstd, beta' = beta - mean alpha, mean, rstd}: the forward's arithmetic (norm_coeff_kernel)
// grid: 1   block: 128
__global__ __launch_bounds__(128) void tb_coef_kernel(const double* __restrict__ part, int nch, double count,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ coef) {
  const int c = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int t = 0; t < nch; ++t) { s1 += part[((int64_t)c * nch + t) * 2]; s2 += part[((int64_t)c * nch + t) * 2 + 1]; }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float invstd = (float)(1.0 / sqrt(var + 1e-5));
  const float alpha = invstd * gamma[c];
  coef[c * 4 + 0] = alpha;
  coef[c * 4 + 1] = beta[c] - (float)mean * alpha;
  coef[c * 4 + 2] = (float)mean;
  coef[c * 4 + 3] = invstd;
}

// one thread = one pixel: the spatial gate and its gradient.  The channel sum is formed as the forward kernel forms
// it, (q0 + q1) + (q2 + q3) with each quarter an fmaf chain in channel order.
// grid: ceil(LL / 256)   block: 256
__global__ __launch_bounds__(256) void tb_pixel_gates_kernel(const float* __restrict__ u, const float* __restrict__ dout,
                                                             const float* __restrict__ coef,
                                                             const float* __restrict__ sse_w, float sse_b, int LL,
                                                             float* __restrict__ sg, float* __restrict__ da) {
  __shared__ float sh_a[CW], sh_b[CW], sh_w[CW];
  if (threadIdx.x < CW) {
    sh_a[threadIdx.x] = coef[threadIdx.x * 4];
    sh_b[threadIdx.x] = coef[threadIdx.x * 4 + 1];
    sh_w[threadIdx.x] = sse_w[threadIdx.x];
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= LL) return;
  float dot[4], ds[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dot[q] = 0.f;
    ds[q] = 0.f;
    for (int i = 0; i < CW / 4; ++i) {
      const int c = q * (CW / 4) + i;
      const float v = u[(int64_t)c * LL + p] * sh_a[c] + sh_b[c];
      dot[q] = fmaf(sh_w[c], v, dot[q]);
      ds[q] = fmaf(dout[(int64_t)c * LL + p], v, ds[q]);
    }
  }
  const float s = 1.0f / (1.0f + expf(-(((dot[0] + dot[1]) + (dot[2] + dot[3])) + sse_b)));
  sg[p] = s;
  da[p] = ((ds[0] + ds[1]) + (ds[2] + ds[3])) * (s * (1.0f - s));
}

// grid: (chunks, 128)   block: 256
__global__ __launch_bounds__(256) void tb_channel_sums_kernel(const float* __restrict__ u, const float* __restrict__ dout,
                                                              const float* __restrict__ coef,
                                                              const float* __restrict__ sg, const float* __restrict__ da,
                                                              int LL, double* __restrict__ part) {
  const int c = blockIdx.y, nch = gridDim.x;
  const float mean = coef[c * 4 + 2], rstd = coef[c * 4 + 3];
  const int p1 = min(LL, (int)(blockIdx.x + 1) * TB_CHUNK);
  double v[TB_NSUM] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = blockIdx.x * TB_CHUNK + threadIdx.x; p < p1; p += 256) {
    const double uh = (double)((u[(int64_t)c * LL + p] - mean) * rstd);
    const double d = (double)dout[(int64_t)c * LL + p], s = (double)sg[p], a = (double)da[p];
    v[0] += d;
    v[1] += d * uh;
    v[2] += d * s;
    v[3] += d * s * uh;
    v[4] += a * uh;
    v[5] += a;
    v[6] += uh;
  }
  tb_block_sums<TB_NSUM>(v, part + ((int64_t)c * nch + blockIdx.x) * TB_NSUM);
}

// The per-channel end of the reductions, the two fc layers of the channel gate forwards (as api.hip computes the
// constant gate at dmp_weights_finalize) and backwards, the parameter gradients, and the coefficients of the last
// pass: k[c] = {gamma rstd, y, ws, (d(m) - S1) / P, S2 / P, mean, rstd}.
// dparams: [d gamma 128][d beta 128][d fc.0.weight 8 x 128][d fc.2.weight 128 x 8][d sSE weight 128][d sSE bias 1]
// grid: 1   block: 128 (thread = channel)
__global__ __launch_bounds__(128) void tb_finish_kernel(const double* __restrict__ part, int nch, double count,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ sse_w, const float* __restrict__ fc,
                                                        const float* __restrict__ coef, float* __restrict__ kco,
                                                        float* __restrict__ dparams) {
  __shared__ double sh_m[CW], sh_dz2[CW], sh_h[8], sh_z1[8], sh_dz1[8];
  const int c = threadIdx.x;
  const float* w1 = fc;                                  // [8][128]
  const float* w2 = fc + 8 * CW;                         // [128][8]
  double T[TB_NSUM];
  for (int k = 0; k < TB_NSUM; ++k) T[k] = 0.0;
  for (int t = 0; t < nch; ++t)
    for (int k = 0; k < TB_NSUM; ++k) T[k] += part[((int64_t)c * nch + t) * TB_NSUM + k];
  const double g = (double)gamma[c], b = (double)beta[c], ws = (double)sse_w[c];
  sh_m[c] = b + g * (T[6] / count);                      // mean_p v
  __syncthreads();
  if (c < 8) {
    double a = 0.0;
    for (int q = 0; q < CW; ++q) a += (double)w1[c * CW + q] * sh_m[q];
    sh_z1[c] = a;
    sh_h[c] = a > 0.0 ? a : 0.0;
  }
  __syncthreads();
  double z2 = 0.0;
  for (int r = 0; r < 8; ++r) z2 += (double)w2[c * 8 + r] * sh_h[r];
  const double y = 1.0 / (1.0 + exp(-z2));
  const double dy = g * T[1] + b * T[0];                 // sum_p D v
  const double dz2 = dy * y * (1.0 - y);
  sh_dz2[c] = dz2;
  for (int r = 0; r < 8; ++r) dparams[2 * CW + 8 * CW + c * 8 + r] = (float)(dz2 * sh_h[r]);
  __syncthreads();
  if (c < 8) {
    double a = 0.0;
    for (int q = 0; q < CW; ++q) a += (double)w2[q * 8 + c] * sh_dz2[q];
    sh_dz1[c] = sh_z1[c] > 0.0 ? a : 0.0;
  }
  __syncthreads();
  double dm = 0.0;
  for (int r = 0; r < 8; ++r) {
    dm += (double)w1[r * CW + c] * sh_dz1[r];
    dparams[2 * CW + r * CW + c] = (float)(sh_dz1[r] * sh_m[c]);
  }
  const double S1 = y * T[0] + T[2] + ws * T[5] + dm;
  const double S2 = y * T[1] + T[3] + ws * T[4] + (dm / count) * T[6];
  dparams[c] = (float)S2;                                // d gamma
  dparams[CW + c] = (float)S1;                           // d beta
  dparams[2 * CW + 16 * CW + c] = (float)(g * T[4] + b * T[5]);      // d sSE weight = sum_p d(a) v
  if (c == 0) dparams[2 * CW + 16 * CW + CW] = (float)T[5];          // d sSE bias (T6 is the same in every channel)
  kco[c * 8 + 0] = coef[c * 4 + 0];
  kco[c * 8 + 1] = (float)y;
  kco[c * 8 + 2] = sse_w[c];
  kco[c * 8 + 3] = (float)((dm - S1) / count);
  kco[c * 8 + 4] = (float)(S2 / count);
  kco[c * 8 + 5] = coef[c * 4 + 2];
  kco[c * 8 + 6] = coef[c * 4 + 3];
  kco[c * 8 + 7] = 0.f;
}

// grid: (ceil(LL / 256), 128)   block: 256
__global__ __launch_bounds__(256) void tb_norm_input_kernel(const float* __restrict__ u, const float* __restrict__ dout,
                                                            const float* __restrict__ kco, const float* __restrict__ sg,
                                                            const float* __restrict__ da, int LL, float* __restrict__ du) {
  const int c = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= LL) return;
  const float k0 = kco[c * 8], y = kco[c * 8 + 1], ws = kco[c * 8 + 2], cst = kco[c * 8 + 3], k2 = kco[c * 8 + 4];
  const float uh = (u[(int64_t)c * LL + p] - kco[c * 8 + 5]) * kco[c * 8 + 6];
  const float dv = fmaf(dout[(int64_t)c * LL + p], y + sg[p], ws * da[p]);
  du[(int64_t)c * LL + p] = k0 * ((dv + cst) - uh * k2);
}

int norm_scse_residual_bwd(dmp_ctx* c, int block, const float* d_u, const float* d_dout, int L, float* d_du,
                           float* d_dparams, hipStream_t s) {
  const BlockW& B = c->W.blk[block - 1];
  const int LL = L * L, nch = cdiv(LL, TB_CHUNK);
  // workspace: gate planes 2 LL floats | coef 512 | kco 1024 | partial sums 128 nch 8 doubles
  const int64_t fl = 2 * (int64_t)LL + 512 + 1024, need = fl + 2 * (int64_t)CW * nch * TB_NSUM + 16;
  int rc;
  if ((rc = bwd_workspace(c, need))) return rc;
  float* sg = c->bwd_ws;
  float* da = sg + LL;
  float* coef = da + LL;
  float* kco = coef + 512;
  double* part = reinterpret_cast<double*>(c->bwd_ws + ((fl + 3) & ~(int64_t)3));
  hipLaunchKernelGGL(tb_stats_kernel, dim3(nch, CW), dim3(256), 0, s, d_u, LL, part);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_coef_kernel, dim3(1), dim3(CW), 0, s, part, nch, (double)LL, B.gamma, B.beta, coef);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_pixel_gates_kernel, dim3(cdiv(LL, 256)), dim3(256), 0, s, d_u, d_dout, coef, B.sse_w, B.sse_b, LL,
                     sg, da);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_channel_sums_kernel, dim3(nch, CW), dim3(256), 0, s, d_u, d_dout, coef, sg, da, LL, part);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_finish_kernel, dim3(1), dim3(CW), 0, s, part, nch, (double)LL, B.gamma, B.beta, B.sse_w, B.cse_fc,
                     coef, kco, d_dparams);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_norm_input_kernel, dim3(cdiv(LL, 256), CW), dim3(256), 0, s, d_u, d_dout, kco, sg, da, LL, d_du);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
