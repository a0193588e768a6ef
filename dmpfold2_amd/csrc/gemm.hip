// Generic float32 GEMM on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32).
// Used for the covariance outer product, the Gauss-Jordan panel/trailing updates and the
// GRU input projections.  Exact f32: an MFMA chain is bitwise an fmaf chain in k order.
#include "common.h"

namespace dmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

// WB = 32 x 32 MFMA blocks per wave and dimension: 2 = a 128 x 128 tile per workgroup, 1 = 64 x 64 (for products whose
// 128-tiles would be a few dozen workgroups - the GRU input projections, the Gauss-Jordan row panel: 36 / 50 workgroups
// on 256 CUs, each walking the whole K).  An element's accumulation chain is the same in both (k ascending, the same
// flush points), so the tile size does not change a bit of the result.
// TWO_LEVEL (K > 256 only) costs 64 more VGPRs; without it the kernel stays under the 160 registers
// that are free beside two f16x3 convolution workgroups per CU.
template <bool A_MCONTIG, bool B_NCONTIG, bool TWO_LEVEL, int WB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int BM = 64 * WB, BN = 64 * WB, LDS_PITCH = BM + 4, NLD = 4 * WB;   // NLD: elements per thread and operand tile
  __shared__ float As[2][BK][LDS_PITCH];
  __shared__ float Bs[2][BK][LDS_PITCH];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (g.lower_tiles && (n0 >> 7) > (m0 >> 7)) return;     // (uniform: before any barrier)

  float ra[NLD], rb[NLD];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int m, k;
      if (A_MCONTIG) { m = tid & (BM - 1); k = tid / BM + (256 / BM) * i; }
      else           { k = tid & 15;  m = (tid >> 4) + 16 * i; }
      const int gm = m0 + m, gk = k0 + k;
      ra[i] = (gm < g.M && gk < g.K) ? g.A[(int64_t)gm * g.sam + (int64_t)gk * g.sak] : 0.f;
      int n, kb;
      if (B_NCONTIG) { n = tid & (BN - 1); kb = tid / BN + (256 / BN) * i; }
      else           { kb = tid & 15; n = (tid >> 4) + 16 * i; }
      const int gn = n0 + n, gkb = k0 + kb;
      rb[i] = (gn < g.N && gkb < g.K) ? g.B[(int64_t)gkb * g.sbk + (int64_t)gn * g.sbn] : 0.f;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      int m, k;
      if (A_MCONTIG) { m = tid & (BM - 1); k = tid / BM + (256 / BM) * i; }
      else           { k = tid & 15;  m = (tid >> 4) + 16 * i; }
      As[buf][k][m] = ra[i];
      int n, kb;
      if (B_NCONTIG) { n = tid & (BN - 1); kb = tid / BN + (256 / BN) * i; }
      else           { kb = tid & 15; n = (tid >> 4) + 16 * i; }
      Bs[buf][kb][n] = rb[i];
    }
  };

  f32x16 acc[2][2];                                       // WB = 1 uses acc[0][0] only
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Two-level accumulation: the MFMA chain is flushed into a second accumulator every 256 k.  A
  // single float32 chain over thousands of equal-signed terms drifts (covariance diagonal at
  // N = 2000: 1.7e-5 relative, tests/test_gpu_fullsize.py); chunks of 256 keep it at the 1e-7 level
  // of a blocked CPU GEMM.  K <= 256 (Gauss-Jordan updates) is unchanged.
  f32x16 tot[TWO_LEVEL ? 2 : 1][TWO_LEVEL ? 2 : 1];
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  }

  const int nk = (g.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    const int kk = lane >> 5, li = lane & 31;
    // fragments of k + 2 are read while the four MFMAs of k run (the scheduling barrier keeps the compiler from
    // sinking the ds_reads to their first use, where their latency would show before every fourth MFMA)
    constexpr int WT = 32 * WB;                             // rows / columns of a wave's tile
    float an0 = As[buf][kk][wm * WT + li], an1 = WB == 2 ? As[buf][kk][wm * WT + 32 + li] : 0.f;
    float bn0 = Bs[buf][kk][wn * WT + li], bn1 = WB == 2 ? Bs[buf][kk][wn * WT + 32 + li] : 0.f;
#pragma unroll
    for (int k = 0; k < BK; k += 2) {
      const float a0 = an0, a1 = an1, b0 = bn0, b1 = bn1;
      if (k + 2 < BK) {
        an0 = As[buf][k + 2 + kk][wm * WT + li];
        bn0 = Bs[buf][k + 2 + kk][wn * WT + li];
        if (WB == 2) {
          an1 = As[buf][k + 2 + kk][wm * WT + 32 + li];
          bn1 = Bs[buf][k + 2 + kk][wn * WT + 32 + li];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      if (WB == 2) {
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (TWO_LEVEL && (kt & 15) == 15 && kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[TWO_LEVEL ? i : 0][TWO_LEVEL ? j : 0][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  const int hi = lane >> 5, col = lane & 31;
#pragma unroll
  for (int i = 0; i < WB; ++i)
#pragma unroll
    for (int j = 0; j < WB; ++j) {
      const int gn = n0 + wn * 32 * WB + j * 32 + col;
      if (gn >= g.N) continue;
      const float bn = g.bias_n ? g.bias_n[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm * 32 * WB + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (gm >= g.M) continue;
        float* p = g.C + (int64_t)gm * g.ldc + gn;
        float v = g.alpha * (TWO_LEVEL ? tot[TWO_LEVEL ? i : 0][TWO_LEVEL ? j : 0][r] + acc[i][j][r] : acc[i][j][r]) + bn;
        if (g.beta != 0.f) v += g.beta * *p;
        *p = v;
      }
    }
}

int gemm_f32(const GemmArgs& g, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0) return DMP_OK;
  // 64 x 64 tiles when 128 x 128 ones would leave most of the 256 CUs without a workgroup (same bits either way)
  const bool small = (int64_t)cdiv(g.N, 128) * cdiv(g.M, 128) < 128;
  const int T = small ? 64 : 128;
  dim3 grid(cdiv(g.N, T), cdiv(g.M, T));
  const bool am = (g.sam == 1), bn = (g.sbn == 1);
  const bool two = g.K > 256;
#define GEMM_LAUNCH(A_, B_)                                                                            \
  do {                                                                                                 \
    if (two && small) hipLaunchKernelGGL((gemm_kernel<A_, B_, true, 1>), grid, dim3(256), 0, s, g);    \
    else if (two) hipLaunchKernelGGL((gemm_kernel<A_, B_, true, 2>), grid, dim3(256), 0, s, g);        \
    else if (small) hipLaunchKernelGGL((gemm_kernel<A_, B_, false, 1>), grid, dim3(256), 0, s, g);     \
    else hipLaunchKernelGGL((gemm_kernel<A_, B_, false, 2>), grid, dim3(256), 0, s, g);                \
  } while (0)
  if (am && bn) GEMM_LAUNCH(true, true);
  else if (am && !bn) GEMM_LAUNCH(true, false);
  else if (!am && bn) GEMM_LAUNCH(false, true);
  else GEMM_LAUNCH(false, false);
#undef GEMM_LAUNCH
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
