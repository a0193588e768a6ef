// Pair trunk (reference network.py:12-103, 194-207): 1x1 maxout stem, 16 residual blocks of
// 5x5 conv (128->512) + 4-way maxout + InstanceNorm + scSE, 2-channel head.
//
// Activations between blocks live in a zero-bordered layout [128][P][P], P = 16*ceil(L/16)+4,
// interior at [2, 2+L): the 5x5 halo reads need no bounds checks and the conv tiles of
// 16x16 pixels never straddle the border logic.  The pre-norm maxout output `u` is dense.
#include "common.h"
#include <type_traits>
#define CONV_BF16_KERNELS
#include "conv_bf16.h"
#define CONV_F16_KERNELS
#include "conv_f16.h"

// Element-wise stages mirror separately rounded float32 tensor ops of the reference; fused
// multiply-adds are written explicitly (fmaf) where they are wanted.
#pragma clang fp contract(off)

namespace dmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Eight consecutive channels of one pixel -> the 16-byte operands of the split-product
// convolutions.  MODE 0: two f16 pieces (conv_f16.h), MODE 2: three bf16 pieces (conv_bf16.h).
// `slot` is the uint4 index inside one piece plane set, `piece_stride` the uint4 count per piece.
// MODE 0: the pieces are those of xscale * o (a power of two per block: exact, undone in the convolution's epilogue),
// so that the low pieces of the bulk of the activations are normal f16 numbers whatever the trunk's activation
// scale is, and the range check is made on the scaled value.
template <int MODE>
__device__ __forceinline__ void store_pieces(const float (&o)[8], uint16_t* __restrict__ xs, int64_t slot,
                                             int64_t piece_stride, int* __restrict__ fault, float xscale) {
  constexpr int NP = (MODE == 0) ? 2 : 3;
  uint16_t pc[NP][8];
  bool bad = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if constexpr (MODE == 0) {
      uint16_t p2[2];
      const float v = o[e] * xscale;
      split2_f16(v, p2);
      pc[0][e] = p2[0]; pc[1][e] = p2[1];
      bad = bad || !(fabsf(v) < 60000.f);             // f16 range (also catches NaN)
    } else {
      uint16_t p3[3];
      split3_bf16(o[e], p3);
      pc[0][e] = p3[0]; pc[1][e] = p3[1]; pc[2][e] = p3[2];
    }
  }
  if (bad) atomicOr(fault, 2);
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    uint4 v;
    v.x = pc[p][0] | ((uint32_t)pc[p][1] << 16);
    v.y = pc[p][2] | ((uint32_t)pc[p][3] << 16);
    v.z = pc[p][4] | ((uint32_t)pc[p][5] << 16);
    v.w = pc[p][6] | ((uint32_t)pc[p][7] << 16);
    reinterpret_cast<uint4*>(xs)[p * piece_stride + slot] = v;
  }
}

// ---------------------------------------------------------------------------------------
// layout helpers
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_pad_kernel(const float* __restrict__ dense, int L, int P,
                                                      float* __restrict__ xpad) {
  const int c = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= P) return;
  const int yy = y - 2, xx = x - 2;
  float v = 0.f;
  if (yy >= 0 && yy < L && xx >= 0 && xx < L) v = dense[((int64_t)c * L + yy) * L + xx];
  xpad[((int64_t)c * P + y) * P + x] = v;
}

__global__ __launch_bounds__(256) void act_unpad_kernel(const float* __restrict__ xpad, int L, int P,
                                                        float* __restrict__ dense) {
  const int c = blockIdx.z, y = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= L) return;
  dense[((int64_t)c * L + y) * L + x] = xpad[((int64_t)c * P + y + 2) * P + x + 2];
}

int act_pad(const float* d_dense, int L, float* d_xpad, hipStream_t s) {
  const int P = act_pitch(L);
  hipLaunchKernelGGL(act_pad_kernel, dim3(cdiv(P, 256), P, CW), dim3(256), 0, s, d_dense, L, P,
                     d_xpad);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}
int act_unpad(const float* d_xpad, int L, float* d_dense, hipStream_t s) {
  const int P = act_pitch(L);
  hipLaunchKernelGGL(act_unpad_kernel, dim3(cdiv(L, 256), L, CW), dim3(256), 0, s, d_xpad, L, P,
                     d_dense);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}
int act_clear(float* d_xpad, int L, hipStream_t s) {
  const int P = act_pitch(L);
  DMP_HIP(hipMemsetAsync(d_xpad, 0, sizeof(float) * CW * (size_t)P * P, s));
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// stem, input-independent part (once per structure)
//   Z0[o,i,j] = b_o + sum_{c<512} W[o,c] m[c,i] m[c,j]
//             + sum_{a,b} W[o,512+21a+b] inv[21i+a, 21j+b] + W[o,953] contacts[i,j]
// GEMM M = 384 output channels (A = W^T[k][o]), N = 64 consecutive pixels, K = 954 on the f32 matrix cores
// (32x32x2; same k pairing and order as before: results unchanged bit for bit).  Round 3: the coupling channels
// are first re-laid as planes[21a+b][i][j] (stem_planes_kernel: coalesced rows of the inverse through LDS; the
// contact map becomes plane 441), so every B operand is a coalesced load, and the k loop is software-pipelined in
// chunks of 8 k pairs - the loads of chunk c+1 are issued before the MFMAs of chunk c and pinned there (the
// round-2 kernel gathered inv with an 84-byte lane stride and waited for every load right where it was issued:
// 1.8 ms = 0.23 of the f32 MFMA peak at L = 300).
// grid: ceil(L*L/64)   block: 256 (wave w owns channel blocks 3w..3w+2)
// ---------------------------------------------------------------------------------------
constexpr int SP_JT = 256;      // columns j per workgroup of the re-layout kernel
// grid: (ceil(L/SP_JT), L rows i, 21 a)   block: 256
__global__ __launch_bounds__(256) void stem_planes_kernel(const float* __restrict__ inv, int L,
                                                          float* __restrict__ planes) {
  __shared__ float row[SP_JT * NS];
  const int j0 = blockIdx.x * SP_JT, i = blockIdx.y, a = blockIdx.z;
  const int nj = min(SP_JT, L - j0);
  const int64_t D = (int64_t)L * NS, LL = (int64_t)L * L;
  const float* src = inv + ((int64_t)i * NS + a) * D + (int64_t)j0 * NS;
  for (int t = threadIdx.x; t < nj * NS; t += 256) row[t] = src[t];
  __syncthreads();
  const int j = threadIdx.x;
  if (j < nj) {
#pragma unroll
    for (int b = 0; b < NS; ++b) planes[(int64_t)(a * NS + b) * LL + (int64_t)i * L + j0 + j] = row[j * NS + b];
  }
}

constexpr int SS_CH = 8;        // k pairs per pipeline chunk
__global__ __launch_bounds__(256, 2) void stem_static_kernel(const float* __restrict__ wT,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ m,
                                                             const float* __restrict__ planes, int L,
                                                             float* __restrict__ z0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int64_t LL = (int64_t)L * L;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int64_t pq[2] = {p0 + li, p0 + 32 + li};
  const bool ok[2] = {pq[0] < LL, pq[1] < LL};
  const int64_t pc[2] = {ok[0] ? pq[0] : LL - 1, ok[1] ? pq[1] : LL - 1};      // clamped: loads stay in range
  const int iq[2] = {(int)(pc[0] / L), (int)(pc[1] / L)};
  const int jq[2] = {(int)(pc[0] - (int64_t)iq[0] * L), (int)(pc[1] - (int64_t)iq[1] * L)};

  f32x16 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][q][r] = 0.f;

  // buffer loads: 128-bit descriptor + wave-uniform byte offset of the k pair (scalar) + a 32-bit lane offset that
  // is fixed for the whole loop (with flat loads the compiler kept a 64-bit address per load in flight and spilled)
  const unsigned woff = 4u * (unsigned)(kk * STEM_OUT + wave * 96 + li);
  const unsigned moff_i[2] = {4u * (unsigned)(kk * L + iq[0]), 4u * (unsigned)(kk * L + iq[1])};
  const unsigned moff_j[2] = {4u * (unsigned)(kk * L + jq[0]), 4u * (unsigned)(kk * L + jq[1])};
  const unsigned poff[2] = {4u * (unsigned)(kk * LL + pc[0]), 4u * (unsigned)(kk * LL + pc[1])};
  constexpr int NCH1 = WIDTH / 2 / SS_CH;                           // 32 chunks of outer-product channels
  const int npairs = planes ? (STEM_IN - 1) / 2 : WIDTH / 2;        // 477 (k = 0..953) or 256
  const int nchunks = (npairs + SS_CH - 1) / SS_CH;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wT), 0, (STEM_IN - 1) * STEM_OUT * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m), 0, WIDTH * L * 4, 0x00020000);
  auto ldb = [](__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  float av[2][SS_CH][3], r0[2][SS_CH][2], r1[2][SS_CH][2];
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  // chunk c -> registers of buffer BUF.  Phase 1 (k < 512): B = r0 * r1 = m[k][i] m[k][j]; phase 2: B = r0 = plane
  auto load1 = [&](auto BUF, int c) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int u = 0; u < SS_CH; ++u) {
      const unsigned k2 = 2u * (unsigned)(c * SS_CH + u);           // uniform
#pragma unroll
      for (int x = 0; x < 3; ++x) av[buf][u][x] = ldb(wrs, woff + 128u * x, k2 * (STEM_OUT * 4u));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        r0[buf][u][q] = ldb(mrs, moff_i[q], k2 * (4u * (unsigned)L));
        r1[buf][u][q] = ldb(mrs, moff_j[q], k2 * (4u * (unsigned)L));
      }
    }
  };
  auto load2 = [&](auto BUF, int c) {
    constexpr int buf = decltype(BUF)::value;
    // the chunk's 16 planes through their own descriptor (the whole tensor may exceed 4 GB); pairs past the last
    // one (the padded tail of the last chunk) read out of range, which a buffer load answers with 0
    const int k0 = 2 * c * SS_CH;
    const int live_k = min(2 * SS_CH, 2 * npairs - k0);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(planes) + (int64_t)(k0 - WIDTH) * LL, 0, (int)((int64_t)live_k * LL * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(wT) + (int64_t)k0 * STEM_OUT, 0, live_k * STEM_OUT * 4, 0x00020000);
#pragma unroll
    for (int u = 0; u < SS_CH; ++u) {
#pragma unroll
      for (int x = 0; x < 3; ++x) av[buf][u][x] = ldb(wr2, woff + 128u * x, 2u * u * (STEM_OUT * 4u));
#pragma unroll
      for (int q = 0; q < 2; ++q) r0[buf][u][q] = ldb(prs, poff[q], 2u * u * (4u * (unsigned)LL));
    }
  };
  auto compute = [&](auto BUF, auto PHASE) {
    constexpr int buf = decltype(BUF)::value;
#pragma unroll
    for (int u = 0; u < SS_CH; ++u) {
      float b0 = r0[buf][u][0], b1 = r0[buf][u][1];
      if (decltype(PHASE)::value == 1) { b0 *= r1[buf][u][0]; b1 *= r1[buf][u][1]; }
      b0 = ok[0] ? b0 : 0.f;
      b1 = ok[1] ? b1 : 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][a], b0, acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][a], b1, acc[a][1], 0, 0, 0);
      }
    }
  };
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  // the loads of chunk c+1 are issued before the MFMAs of chunk c and pinned there (scheduling barriers)
  load1(B0{}, 0);
#pragma unroll 1
  for (int c = 0; c < NCH1; c += 2) {                               // NCH1 is even: chunk c lives in buffer c & 1
    load1(B1{}, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(B0{}, P1{});
    __builtin_amdgcn_sched_barrier(0);
    if (c + 2 < NCH1) load1(B0{}, c + 2);
    else if (c + 2 < nchunks) load2(B0{}, c + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(B1{}, P1{});
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int c = NCH1; c < nchunks; c += 2) {
    if (c + 1 < nchunks) load2(B1{}, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(B0{}, P2{});
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < nchunks) {
      if (c + 2 < nchunks) load2(B0{}, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(B1{}, P2{});
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!ok[q]) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = wave * 96 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        z0[(int64_t)o * LL + pq[q]] = acc[a][q][r] + bias[o];
      }
    }
}

int stem_static(dmp_ctx* c, const float* d_mat1d, const float* d_inv, const float* d_contacts,
                int L, float* d_z0, hipStream_t s) {
  const Weights& W = c->W;
  const int64_t LL = (int64_t)L * L;
  const float* planes = nullptr;
  if (d_inv != nullptr) {
    hipLaunchKernelGGL(stem_planes_kernel, dim3(cdiv(L, SP_JT), L, NS), dim3(256), 0, s, d_inv, L, c->planes);
    DMP_LAUNCH_CHECK();
    DMP_HIP(hipMemcpyAsync(c->planes + (int64_t)NS * NS * LL, d_contacts, sizeof(float) * LL, hipMemcpyDeviceToDevice, s));
    planes = c->planes;
  }
  hipLaunchKernelGGL(stem_static_kernel, dim3((unsigned)cdiv64(LL, 64)), dim3(256), 0, s, W.stemT, W.stem_b,
                     d_mat1d, planes, L, d_z0);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// stem, per-pass part: + W[o,954]*dmap, max over triples, InstanceNorm
// ---------------------------------------------------------------------------------------
// Also the InstanceNorm statistics of what it writes (round 4: a separate pass over u read its 46 MB again):
// part[blockIdx.x][g][2] = float64 (sum, sum of squares) of the block's 256 pixels - wave sums by the xor butterfly,
// the four waves added in order; stats_reduce_kernel then adds the blocks in a fixed order.
__global__ __launch_bounds__(256) void stem_maxout_kernel(const float* __restrict__ z0,
                                                          const float* __restrict__ wd,
                                                          const float* __restrict__ dmap, int64_t LL,
                                                          float* __restrict__ u, double* __restrict__ part) {
  __shared__ double red[2][4];
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  double s1 = 0.0, s2 = 0.0;
  if (p < LL) {
    const float d = dmap[p];
    const float v0 = z0[(int64_t)(3 * g) * LL + p] + wd[3 * g] * d;
    const float v1 = z0[(int64_t)(3 * g + 1) * LL + p] + wd[3 * g + 1] * d;
    const float v2 = z0[(int64_t)(3 * g + 2) * LL + p] + wd[3 * g + 2] * d;
    const float v = fmaxf(fmaxf(v0, v1), v2);
    u[(int64_t)g * LL + p] = v;
    s1 = (double)v;
    s2 = s1 * s1;
  }
  s1 = wave_sum_f64(s1);
  s2 = wave_sum_f64(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const double t = ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
    part[((int64_t)blockIdx.x * CW + g) * 2 + threadIdx.x] = t;
  }
}

// stats[c] = sum over partials, in a fixed order; with `ab` also the InstanceNorm coefficients of the
// channel (the arithmetic of norm_coeff_kernel below), which saves the trunk a dependent launch per block
// grid: 128 channels, block: 64 (one wave); lane l sums partials l, l+64, ... then a fixed tree
__global__ __launch_bounds__(64) void stats_reduce_kernel(const double* __restrict__ part, int nparts,
                                                          double* __restrict__ stats, double count,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ ab) {
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int t = threadIdx.x; t < nparts; t += 64) {
    s1 += part[((int64_t)t * CW + c) * 2 + 0];
    s2 += part[((int64_t)t * CW + c) * 2 + 1];
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  if (threadIdx.x == 0) {
    stats[c * 2 + 0] = s1;
    stats[c * 2 + 1] = s2;
    if (ab) {
      const double mean = s1 / count;
      double var = s2 / count - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const float invstd = (float)(1.0 / sqrt(var + 1e-5));
      const float alpha = invstd * gamma[c];
      ab[c * 2 + 0] = alpha;
      ab[c * 2 + 1] = beta[c] - (float)mean * alpha;
    }
  }
}

// InstanceNorm2d (eps 1e-5, biased variance) folded to y = u*alpha + beta' per channel
__global__ void norm_coeff_kernel(const double* __restrict__ stats, double count,
                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                  float* __restrict__ ab) {
  const int c = threadIdx.x;
  const double mean = stats[c * 2] / count;
  double var = stats[c * 2 + 1] / count - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float invstd = (float)(1.0 / sqrt(var + 1e-5));
  const float alpha = invstd * gamma[c];
  ab[c * 2 + 0] = alpha;
  ab[c * 2 + 1] = beta[c] - (float)mean * alpha;
}

// InstanceNorm of the stem's maxout output into the padded planes - and, SPLIT = 0 / 2, the f16 / bf16 pieces of
// block 1's input in the same pass (round 4: a separate act_split read the 48 MB it had just written again).  One
// thread = 8 consecutive channels of one position of the PADDED plane: border positions write zero pieces (the
// float32 border is kept zero by act_clear), interior ones the normalised values and their pieces.
template <int SPLIT>
__global__ __launch_bounds__(256) void stem_norm_kernel(const float* __restrict__ u,
                                                        const float* __restrict__ ab, int L, int P,
                                                        float* __restrict__ xpad, uint16_t* __restrict__ xs,
                                                        int* __restrict__ fault, float xscale) {
  const int y = blockIdx.y, cgp = blockIdx.z;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= P) return;
  const int yy = y - 2, xx = x - 2;
  const bool inside = yy >= 0 && yy < L && xx >= 0 && xx < L;
  if (SPLIT < 0 && !inside) return;
  const int64_t PP = (int64_t)P * P, LL = (int64_t)L * L;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = cgp * 8 + e;
    o[e] = 0.f;
    if (inside) {
      o[e] = u[(int64_t)c * LL + (int64_t)yy * L + xx] * ab[c * 2] + ab[c * 2 + 1];
      xpad[(int64_t)c * PP + (int64_t)y * P + x] = o[e];
    }
  }
  if constexpr (SPLIT == 0) store_pieces<0>(o, xs, (int64_t)cgp * PP + (int64_t)y * P + x, 16 * PP, fault, xscale);
  if constexpr (SPLIT == 2) store_pieces<2>(o, xs, (int64_t)cgp * PP + (int64_t)y * P + x, 16 * PP, fault, 1.0f);
}

// `split` = also write the pieces block 1's convolution reads (inside a trunk pass, conv_mode 0 / 2)
int stem_update_padded(dmp_ctx* c, const float* d_z0, const float* d_dmap, int L, float* d_xpad,
                       hipStream_t s, bool split) {
  const Weights& W = c->W;
  const int64_t LL = (int64_t)L * L;
  const int P = act_pitch(L);
  const unsigned nblk = (unsigned)cdiv64(LL, 256);
  hipLaunchKernelGGL(stem_maxout_kernel, dim3(nblk, CW), dim3(256), 0, s, d_z0,
                     W.stem_wd, d_dmap, LL, c->u, c->part);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(stats_reduce_kernel, dim3(CW), dim3(64), 0, s, c->part, (int)nblk, c->stats, (double)LL,
                     W.stem_gamma, W.stem_beta, c->ab);
  DMP_LAUNCH_CHECK();
  const dim3 grid(cdiv(P, 256), P, 16);
  const int mode = (split && c->conv_mode != 1) ? c->conv_mode : -1;
  if (mode == 0)
    hipLaunchKernelGGL(stem_norm_kernel<0>, grid, dim3(256), 0, s, c->u, c->ab, L, P, d_xpad, c->xsplit, c->seq_abort,
                       c->act_scaling ? W.blk[0].x_scale : 1.0f);
  else if (mode == 2)
    hipLaunchKernelGGL(stem_norm_kernel<2>, grid, dim3(256), 0, s, c->u, c->ab, L, P, d_xpad, c->xsplit, c->seq_abort, 1.0f);
  else
    hipLaunchKernelGGL(stem_norm_kernel<-1>, grid, dim3(256), 0, s, c->u, c->ab, L, P, d_xpad, (uint16_t*)nullptr,
                       c->seq_abort, 1.0f);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// conv 5x5 (128 -> 512) + bias + max over channel quadruples, with per-channel sums
//
// Implicit GEMM on the f32 matrix cores: D[conv channel, pixel] += W[ch, (c,tap)] * X[(c,tap), pixel]
//   M = 128 conv channels per workgroup (4 x 32-row MFMA blocks), 4 workgroups ("splits") cover 512
//   N = 256 pixels = one 16x16 tile, as eight 4x8 patches of 32 pixels (MFMA N blocks)
//   K = 3200 = 64 stages of (2 input channels x 25 taps); one 32x32x2 MFMA consumes a channel pair
// Each wave owns 2 pixel patches x 4 channel blocks = 8 accumulators (128 registers).
// Per stage the workgroup stages a 2 x 20 x 20 input halo tile and a 25 x 2 x 128 weight slab in
// LDS (double buffered, one barrier per stage).  MFMA rows (reg&3) are 4 consecutive conv
// channels = one maxout group, so the 4-way max is register-local.
// ---------------------------------------------------------------------------------------
constexpr int HALO = CONV_TILE + 4;       // 20
constexpr int IN_PITCH = 24;              // row pitch in LDS: 4 rows x 8 columns hit 32 distinct banks
constexpr int NTAP = 25;
constexpr int MCH = 128;                  // conv channels per workgroup
constexpr int NCHUNK = CW / CONV_CC;      // 64
// 2.26 ms per launch at L = 300 (floors: no staging 2.18, MFMA only 2.09).  The scheduling variants that
// were measured against this one (branchy / register-staged weight prefetch 2.50 / 2.36, LDS-DMA input
// image 2.25, tap-ahead fragment reads, sched_group_barrier pinning, b128 weight fragments: neutral or
// slower) live in the history of this file and in tools/ubench_conv.hip.
constexpr int W_STAGE = NTAP * CONV_CC * MCH;          // 6400 floats
constexpr int IN_STAGE = CONV_CC * HALO * IN_PITCH;    // 960 floats

// IDX (the training-side slice, train.hip): also write which channel of each quadruple won the maximum - the FIRST
// maximal one, as torch.max (network.py:31) - one byte per maxout channel and pixel.  The inference instantiation is
// <false>: its code is what it was.
template <bool IDX>
__global__ __launch_bounds__(256, 2) void conv5x5_maxout_kernel(const float* __restrict__ xpad,
                                                                const float* __restrict__ wpack,
                                                                const float* __restrict__ bias, int L,
                                                                int P, int tiles, int nwork,
                                                                float* __restrict__ u,
                                                                double* __restrict__ part,
                                                                uint8_t* __restrict__ idx) {
  // one LDS object (a second one makes hipcc drain vmcnt in front of LDS reads of a DMA pipeline)
  __shared__ __attribute__((aligned(16))) float smem[2 * W_STAGE + 2 * IN_STAGE];
  float (*w_lds)[W_STAGE] = reinterpret_cast<float (*)[W_STAGE]>(smem);
  float (*in_lds)[IN_STAGE] = reinterpret_cast<float (*)[IN_STAGE]>(smem + 2 * W_STAGE);
  constexpr int IP = IN_PITCH;       // LDS row pitch of the input tile (bank-conflict free)
  // XCD-aware remap: block b runs on XCD b % 8; give each XCD a contiguous range of work items
  // so the 4 channel splits of a tile and neighbouring tiles share one L2.
  const int id = blockIdx.x;
  const int per = gridDim.x >> 3;
  const int work = (id & 7) * per + (id >> 3);
  if (work >= nwork) return;
  const int tile = work >> 2, split = work & 3;
  const int ty0 = (tile / tiles) * CONV_TILE, tx0 = (tile % tiles) * CONV_TILE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int64_t PP = (int64_t)P * P;

  // staging assignments
  const float4* wsrc = reinterpret_cast<const float4*>(wpack + (int64_t)split * NCHUNK * W_STAGE);
  // Every prefetch load is unconditional (a load under a branch makes hipcc wait vmcnt(0) right
  // behind it): out-of-range slots re-read a valid element and are dropped at the LDS store.
  int in_off[4], in_dst[4];
  bool in_ok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = tid + e * 256;   // < 800 valid
    in_ok[e] = idx < CONV_CC * HALO * HALO;
    const int idc = in_ok[e] ? idx : 0;
    const int cc = idc / (HALO * HALO), rem = idc % (HALO * HALO);
    const int yy = rem / HALO, xx = rem % HALO;
    in_off[e] = (int)(cc * PP + (int64_t)(ty0 + yy) * P + tx0 + xx);
    in_dst[e] = cc * HALO * IP + yy * IP + xx;
  }
  float ireg[4];
  // weight slab of the next stage by LDS-DMA (25 wave-instructions of 1 KB, lane-linear destination); the
  // input halo tile through registers
  auto prefetch = [&](int chunk) {
    const float4* ws = wsrc + (int64_t)chunk * (W_STAGE / 4);
    const float* xs = xpad + (int64_t)chunk * CONV_CC * PP;
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    float* dst = w_lds[(chunk & 1)] + (size_t)wave * 256;
#pragma unroll
    for (int e = 0; e < 6; ++e)
      __builtin_amdgcn_global_load_lds((gptr_t)(ws + tid + e * 256), (lptr_t)(dst + e * 1024), 16, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_global_load_lds((gptr_t)(ws + tid + 6 * 256), (lptr_t)(dst + 6 * 1024), 16, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) ireg[e] = xs[in_off[e]];
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (in_ok[e]) in_lds[buf][in_dst[e]] = ireg[e];
  };

  // fragment addresses
  int b_off[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int nb = 2 * wave + q;
    const int y = (nb >> 1) * 4 + (li >> 3), x = (nb & 1) * 8 + (li & 7);
    b_off[q] = kk * HALO * IP + y * IP + x;
  }
  const int a_off = kk * MCH + li;

  f32x16 acc[4][2];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;

  prefetch(0);
  commit(0);
  __syncthreads();
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < NCHUNK) prefetch(chunk + 1);
    const float* wl = w_lds[buf] + a_off;
    const float* il = in_lds[buf];
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int dy = tap / 5, dx = tap % 5;
      float a[4], b[2];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) a[mb] = wl[tap * CONV_CC * MCH + mb * 32];
#pragma unroll
      for (int q = 0; q < 2; ++q) b[q] = il[b_off[q] + dy * IP + dx];
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b[q], acc[mb][q], 0, 0, 0);
    }
    if (chunk + 1 < NCHUNK) commit(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias, 4-way max, store, per-channel partial sums
  float* sred = w_lds[0];   // [wave][32 channels][2]
  const float* bsp = bias + split * MCH;
  const int64_t LL = (int64_t)L * L;
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int cl = mb * 32 + 8 * g4 + 4 * kk;      // first conv channel of the group (local)
      const int gch = split * 32 + (cl >> 2);         // maxout channel
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v = acc[mb][q][4 * g4] + bsp[cl];
        int win = 0;
        if constexpr (IDX) {
#pragma unroll
          for (int e = 1; e < 4; ++e) {
            const float t = acc[mb][q][4 * g4 + e] + bsp[cl + e];
            if (t > v) { v = t; win = e; }               // strict: the first maximal channel wins
          }
        } else {
          v = fmaxf(v, acc[mb][q][4 * g4 + 1] + bsp[cl + 1]);
          v = fmaxf(v, acc[mb][q][4 * g4 + 2] + bsp[cl + 2]);
          v = fmaxf(v, acc[mb][q][4 * g4 + 3] + bsp[cl + 3]);
        }
        const int nb = 2 * wave + q;
        const int y = ty0 + (nb >> 1) * 4 + (li >> 3), x = tx0 + (nb & 1) * 8 + (li & 7);
        if (y < L && x < L) {
          if constexpr (IDX) idx[(int64_t)gch * LL + (int64_t)y * L + x] = (uint8_t)win;
          u[(int64_t)gch * LL + (int64_t)y * L + x] = v;
          s1 += v;
          s2 += v * v;
        }
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off, 32);
        s2 += __shfl_xor(s2, off, 32);
      }
      if (li == 0) {
        const int chl = (cl >> 2);                    // 0..31 within the split
        sred[(wave * 32 + chl) * 2 + 0] = s1;
        sred[(wave * 32 + chl) * 2 + 1] = s2;
      }
    }
  __syncthreads();
  if (tid < 64) {
    const int chl = tid >> 1, which = tid & 1;
    double t = 0.0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) t += (double)sred[(wv * 32 + chl) * 2 + which];
    part[((int64_t)tile * CW + split * 32 + chl) * 2 + which] = t;
  }
}

// block = 0: statistics only (stage-level API); block = k: also block k's InstanceNorm coefficients into
// c->ab, which norm_scse_residual_padded then finds ready (c->ab_current)
int conv5x5_reduce_stats(dmp_ctx* c, int L, double* d_stats, hipStream_t s, int block) {
  const int nparts = c->part_count;                    // what the convolution launched last wrote (tiles or half tiles)
  if (block > 0) {
    const BlockW& B = c->W.blk[block - 1];
    hipLaunchKernelGGL(stats_reduce_kernel, dim3(CW), dim3(64), 0, s, c->part, nparts, d_stats,
                       (double)L * (double)L, B.gamma, B.beta, c->ab);
    c->ab_current = true;
  } else {
    hipLaunchKernelGGL(stats_reduce_kernel, dim3(CW), dim3(64), 0, s, c->part, nparts, d_stats, 1.0,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
  }
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// float32 planes [128][P][P] -> the piece planes [pieces][16][P][P][8] the split-product
// convolutions read (MODE 0: two f16 pieces, MODE 2: three bf16 pieces); borders stay 0.
template <int MODE>
__global__ __launch_bounds__(256) void act_split_kernel(const float* __restrict__ xpad, int P,
                                                        uint16_t* __restrict__ xs, int* __restrict__ fault,
                                                        float xscale) {
  const int y = blockIdx.y, cgp = blockIdx.z;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= P) return;
  const int64_t PP = (int64_t)P * P;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = xpad[(int64_t)(cgp * 8 + e) * PP + (int64_t)y * P + x];
  store_pieces<MODE>(o, xs, (int64_t)cgp * PP + (int64_t)y * P + x, 16 * PP, fault, xscale);
}

// pieces of the input of residual block `block` (1..16)
int act_split(dmp_ctx* c, const float* d_xpad, int L, int block, hipStream_t s) {
  const int P = act_pitch(L);
  dim3 grid(cdiv(P, 256), P, 16);
  if (c->conv_mode == 2)
    hipLaunchKernelGGL(act_split_kernel<2>, grid, dim3(256), 0, s, d_xpad, P, c->xsplit, c->seq_abort, 1.0f);
  else
    hipLaunchKernelGGL(act_split_kernel<0>, grid, dim3(256), 0, s, d_xpad, P, c->xsplit, c->seq_abort,
                       c->act_scaling ? c->W.blk[block - 1].x_scale : 1.0f);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// Function attributes (dynamic LDS above 64 KB needs an opt-in) are per process and device.  They are
// set when a context is created, never between launches: changing them while another stream has
// launches of the same kernel in flight corrupted those launches (first target of a fresh scheduler,
// about one run in three).
int trunk_kernel_attrs(dmp_ctx* c) {
  static bool done[64] = {};
  if (c->device >= 0 && c->device < 64 && done[c->device]) return DMP_OK;
  DMP_HIP(hipFuncSetAttribute((const void*)conv5x5_bf16x6_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              convq_lds_bytes<8>()));
  DMP_HIP(hipFuncSetAttribute((const void*)conv5x5_bf16x6_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              convq_lds_bytes<4>()));
  DMP_HIP(hipFuncSetAttribute((const void*)conv5x5_f16x3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              convh_lds_bytes<8>()));
  DMP_HIP(hipFuncSetAttribute((const void*)conv5x5_f16x3_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              convh_lds_bytes<4>()));
  if (c->device >= 0 && c->device < 64) done[c->device] = true;
  return DMP_OK;
}

int conv5x5_maxout_padded(dmp_ctx* c, int block, const float* d_xpad, int L, float* d_u,
                          double* d_stats, hipStream_t s, bool reduce) {
  const BlockW& B = c->W.blk[block - 1];
  const int tiles = act_tiles(L), P = act_pitch(L);
  const int nwork = tiles * tiles * CONV_SPLIT;
  const int grid = round_up(nwork, 8);
  if (c->conv_mode != 1) {
    // float32-grade products from f16 / bf16 pieces on the 16-bit matrix cores
    if (!c->xsplit_current) {
      int rc = act_split(c, d_xpad, L, block, s);
      if (rc) return rc;
    }
    // tile shape: 16 x 16 pixels, or 8 x 16 where the 16 x 16 shape would leave half the CUs without a workgroup
    // (L <= 80; conv_f16.h ChShape: the same bits either way).  Both write their InstanceNorm partial sums per half tile.
    const int bands = c->conv_tile_bands > 0 ? c->conv_tile_bands : conv_split_rows(tiles);
    const float scale = c->act_scaling ? B.wh_inv_scale / B.x_scale : B.wh_inv_scale;
    if (c->conv_mode == 2) {
      if (bands == 2)
        hipLaunchKernelGGL(conv5x5_bf16x6_kernel<4>, dim3(conv_bf16_grid(tiles, 2)), dim3(256), convq_lds_bytes<4>(), s,
                           c->xsplit, B.wq, B.bias, L, P, tiles, nwork, d_u, c->part);
      else
        hipLaunchKernelGGL(conv5x5_bf16x6_kernel<8>, dim3(conv_bf16_grid(tiles, 1)), dim3(256), convq_lds_bytes<8>(), s,
                           c->xsplit, B.wq, B.bias, L, P, tiles, nwork, d_u, c->part);
    } else {
      if (bands == 2)
        hipLaunchKernelGGL(conv5x5_f16x3_kernel<4>, dim3(conv_f16_grid(tiles, 2)), dim3(256), convh_lds_bytes<4>(), s,
                           c->xsplit, B.wh, B.bias, scale, L, P, tiles, nwork, d_u, c->part);
      else
        hipLaunchKernelGGL(conv5x5_f16x3_kernel<8>, dim3(conv_f16_grid(tiles, 1)), dim3(256), convh_lds_bytes<8>(), s,
                           c->xsplit, B.wh, B.bias, scale, L, P, tiles, nwork, d_u, c->part);
    }
    DMP_LAUNCH_CHECK();
    c->part_count = 2 * tiles * tiles;                 // half tiles
    return reduce ? conv5x5_reduce_stats(c, L, d_stats, s) : DMP_OK;
  }
  hipLaunchKernelGGL(conv5x5_maxout_kernel<false>, dim3(grid), dim3(256), 0, s, d_xpad, B.wpack, B.bias, L,
                     P, tiles, nwork, d_u, c->part, (uint8_t*)nullptr);
  DMP_LAUNCH_CHECK();
  c->part_count = tiles * tiles;
  return reduce ? conv5x5_reduce_stats(c, L, d_stats, s) : DMP_OK;
}

// the block's convolution + maxout in float32 with the winner of every quadruple (train.hip: the backward routes the
// gradient of a maxout channel to that one convolution channel)
int conv5x5_maxout_winners(dmp_ctx* c, int block, const float* d_xpad, int L, float* d_u, uint8_t* d_idx, hipStream_t s) {
  const BlockW& B = c->W.blk[block - 1];
  const int tiles = act_tiles(L), P = act_pitch(L);
  const int nwork = tiles * tiles * CONV_SPLIT;
  hipLaunchKernelGGL(conv5x5_maxout_kernel<true>, dim3(round_up(nwork, 8)), dim3(256), 0, s, d_xpad, B.wpack, B.bias, L,
                     P, tiles, nwork, d_u, c->part, d_idx);
  DMP_LAUNCH_CHECK();
  c->part_count = tiles * tiles;
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// InstanceNorm + scSE + residual (network.py:32, 37-82, 99-101):
//   y_c = u_c*alpha_c + beta'_c ;  s = sigmoid(sum_c ws_c y_c + bs)
//   out_c = (y_c*cse_c + y_c*s) + x_c
// ---------------------------------------------------------------------------------------
// SPLIT: -1 = float32 output only, 0 / 2 = also write f16 / bf16 pieces (store_pieces) for the next block's convolution.
// HEAD = 1 (the last block inside a trunk pass): also the two channels of the 1x1 head convolution (network.py:207) of
// the block's OUTPUT, as planes - the head used to read the 128 output channels again.
// FOUR lanes per pixel (round 4; a lane = 32 channels in registers: about 60 VGPRs): beside two f16x3 convolution
// workgroups a CU has 176 registers per SIMD lane left, which was ONE wave per SIMD of the two-lanes-per-pixel form
// (108 registers; the kernel took 135 us there against 38 alone) and is three waves of this one.  A wave = 16 pixels
// x 4 channel quarters; sums over the channels are formed as (q0 + q1) + (q2 + q3), each quarter an fmaf chain in
// channel order - the same in every lane of a pixel.
#ifndef NORM_PRIO
#define NORM_PRIO 0
#endif
template <int SPLIT, int HEAD>
__global__ __launch_bounds__(256) void norm_scse_residual_kernel(
    const float* __restrict__ u, const float* __restrict__ ab, const float* __restrict__ cse,
    const float* __restrict__ sse_w, float sse_b, const float* __restrict__ xin, int L, int P,
    float* __restrict__ xout, uint16_t* __restrict__ xs, int* __restrict__ fault, float xscale,
    const float* __restrict__ hw, float hb0, float hb1, float* __restrict__ head0, float* __restrict__ head1) {
  __shared__ float sh_a[CW], sh_b[CW], sh_g[CW], sh_w[CW], sh_h[HEAD ? 2 * CW : 1];
#if NORM_PRIO
  __builtin_amdgcn_s_setprio(NORM_PRIO);                 // ahead of the convolution waves it shares its CUs with (see NORM_PRIO)
#endif
  if (threadIdx.x < CW) {
    sh_a[threadIdx.x] = ab[threadIdx.x * 2];
    sh_b[threadIdx.x] = ab[threadIdx.x * 2 + 1];
    sh_g[threadIdx.x] = cse[threadIdx.x];
    sh_w[threadIdx.x] = sse_w[threadIdx.x];
  }
  if constexpr (HEAD) sh_h[threadIdx.x] = hw[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4;                          // channels [32 q, 32 q + 32)
  const int y = blockIdx.y;
  const int x = blockIdx.x * 64 + wave * 16 + (lane & 15);
  const bool live = x < L;
  const int xc = live ? x : L - 1;                  // clamped lanes compute, do not store
  const int64_t LL = (int64_t)L * L, PP = (int64_t)P * P;
  const int64_t p = (int64_t)y * L + xc, pp = (int64_t)(y + 2) * P + xc + 2;
  constexpr int QC = CW / 4;
  float yv[QC];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < QC; ++i) {
    const int c = q * QC + i;
    yv[i] = u[c * LL + p] * sh_a[c] + sh_b[c];
    dot = fmaf(sh_w[c], yv[i], dot);
  }
  dot += __shfl_xor(dot, 16, 64);                   // q0 + q1 | q2 + q3
  dot += __shfl_xor(dot, 32, 64);                   // (q0 + q1) + (q2 + q3)
  const float sg = sigmoid_f(dot + sse_b);
  float h0 = 0.f, h1 = 0.f;
#pragma unroll
  for (int g8 = 0; g8 < QC / 8; ++g8) {
    const int cgp = q * (QC / 8) + g8;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cgp * 8 + e;
      const float t = yv[g8 * 8 + e] * sh_g[c] + yv[g8 * 8 + e] * sg;      // contraction is off for this file
      o[e] = t + xin[c * PP + pp];
      if (live) xout[c * PP + pp] = o[e];
      if constexpr (HEAD) {
        h0 = fmaf(sh_h[c], o[e], h0);
        h1 = fmaf(sh_h[CW + c], o[e], h1);
      }
    }
    // inside a trunk pass: also emit the pieces the next block's split-product convolution reads
    if (live) {
      if constexpr (SPLIT == 0) store_pieces<0>(o, xs, (int64_t)cgp * PP + pp, 16 * PP, fault, xscale);
      if constexpr (SPLIT == 2) store_pieces<2>(o, xs, (int64_t)cgp * PP + pp, 16 * PP, fault, 1.0f);
    }
  }
  if constexpr (HEAD) {
    h0 += __shfl_xor(h0, 16, 64);
    h0 += __shfl_xor(h0, 32, 64);
    h1 += __shfl_xor(h1, 16, 64);
    h1 += __shfl_xor(h1, 32, 64);
    if (live && q == 0) {
      head0[(int64_t)y * L + x] = h0 + hb0;
      head1[(int64_t)y * L + x] = h1 + hb1;
    }
  }
}

// `head` = the last block of a trunk pass: the kernel also leaves the head's two planes in c->head0 / c->head1
// (head_gram_padded then skips its pass over the 128 channels)
int norm_scse_residual_padded(dmp_ctx* c, int block, const float* d_u, const double* d_stats,
                              const float* d_xpad_in, int L, float* d_xpad_out, hipStream_t s, bool head) {
  const BlockW& B = c->W.blk[block - 1];
  const Weights& W = c->W;
  const int P = act_pitch(L);
  if (!c->ab_current) {
    hipLaunchKernelGGL(norm_coeff_kernel, dim3(1), dim3(CW), 0, s, d_stats, (double)L * (double)L,
                       B.gamma, B.beta, c->ab);
    DMP_LAUNCH_CHECK();
  }
  c->ab_current = false;
  // inside a trunk pass the kernel also emits the pieces the next convolution reads (scaled for THAT block; the last
  // block's output is read by the head in float32 only: no pieces)
  const int split = (c->conv_mode != 1 && c->xsplit_current && block < NBLOCK) ? c->conv_mode : -1;
  const float xscale = (c->act_scaling && block < NBLOCK) ? c->W.blk[block].x_scale : 1.0f;
  dim3 grid(cdiv(L, 64), L);
#define NORM_LAUNCH(S, H)                                                                              \
  hipLaunchKernelGGL((norm_scse_residual_kernel<S, H>), grid, dim3(256), 0, s, d_u, c->ab, B.cse, B.sse_w, \
                     B.sse_b, d_xpad_in, L, P, d_xpad_out, c->xsplit, c->seq_abort, xscale, W.head_w,   \
                     W.head_b[0], W.head_b[1], c->head0, c->head1)
  if (head) NORM_LAUNCH(-1, 1);
  else if (split == 0) NORM_LAUNCH(0, 0);
  else if (split == 2) NORM_LAUNCH(2, 0);
  else NORM_LAUNCH(-1, 0);
#undef NORM_LAUNCH
  DMP_LAUNCH_CHECK();
  c->head_current = head;
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// head 1x1 conv (128 -> 2), row means of channel 1, Gram matrix of |sym(channel 0)|
// ---------------------------------------------------------------------------------------
// the two head planes of padded activations (stage-level API; inside a trunk pass the last block's norm kernel
// writes them): per pixel the channel sum as four quarter chains, (q0 + q1) + (q2 + q3), like that kernel
__global__ __launch_bounds__(256) void head_planes_kernel(const float* __restrict__ xpad,
                                                          const float* __restrict__ hw, float b0, float b1,
                                                          int L, int P, float* __restrict__ head0,
                                                          float* __restrict__ head1) {
  __shared__ float sw[2 * CW];
  sw[threadIdx.x] = hw[threadIdx.x];
  __syncthreads();
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  const int64_t PP = (int64_t)P * P;
  const float* px = xpad + (int64_t)(i + 2) * P + j + 2;
  float p0[4], p1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    p0[q] = 0.f; p1[q] = 0.f;
#pragma unroll 8
    for (int k = 0; k < CW / 4; ++k) {
      const int c = q * (CW / 4) + k;
      const float v = px[c * PP];
      p0[q] = fmaf(sw[c], v, p0[q]);
      p1[q] = fmaf(sw[CW + c], v, p1[q]);
    }
  }
  head0[(int64_t)i * L + j] = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + b0;
  head1[(int64_t)i * L + j] = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + b1;
}

// conf[i] = mean over j of head channel 1 (float64 sum: lane-strided, then a fixed tree)
__global__ __launch_bounds__(256) void head_conf_kernel(const float* __restrict__ head1, int L,
                                                        float* __restrict__ conf) {
  __shared__ double red[256];
  const int i = blockIdx.x;
  double rsum = 0.0;
  for (int j = threadIdx.x; j < L; j += 256) rsum += (double)head1[(int64_t)i * L + j];
  red[threadIdx.x] = rsum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) conf[i] = (float)(red[0] / (double)L);
}

// dm = |(h + h^T)/2| ; M_ij = 0.5*((dm_0j^2 + dm_i0^2) - dm_ij^2), every step rounded to f32
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ h0, int L,
                                                   float* __restrict__ M) {
  const int i = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= L) return;
  auto dm = [&](int a, int b) {
    return fabsf((h0[(int64_t)a * L + b] + h0[(int64_t)b * L + a]) * 0.5f);
  };
  const float d0j = dm(0, j), di0 = dm(i, 0), dij = dm(i, j);
  const float t = d0j * d0j + di0 * di0;
  M[(int64_t)i * L + j] = 0.5f * (t - dij * dij);
}

int head_gram_padded(dmp_ctx* c, const float* d_xpad, int L, float* d_conf, float* d_M,
                     hipStream_t s) {
  const Weights& W = c->W;
  const int P = act_pitch(L);
  if (!c->head_current) {
    hipLaunchKernelGGL(head_planes_kernel, dim3(cdiv(L, 256), L), dim3(256), 0, s, d_xpad, W.head_w, W.head_b[0],
                       W.head_b[1], L, P, c->head0, c->head1);
    DMP_LAUNCH_CHECK();
  }
  c->head_current = false;
  hipLaunchKernelGGL(head_conf_kernel, dim3(L), dim3(256), 0, s, c->head1, L, d_conf);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_kernel, dim3(cdiv(L, 256), L), dim3(256), 0, s, c->head0, L, d_M);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
