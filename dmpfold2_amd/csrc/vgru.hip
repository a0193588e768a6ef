// gru_vertical: the 2-layer GRU that runs DOWN the alignment (reference network.py:189, 223-224:
// time axis = N sequences, batch = L columns, hidden 512).
//
// One launch per time step computes layer 0 at step t and layer 1 at step t-1 (both read the same
// h0 state) as f32-MFMA GEMMs  gates^T[j, b] = sum_k W[j, k] * x[k, b]  with the hidden index j
// on the MFMA M axis and the alignment column b on the N axis.
//
// Layouts are chosen so one 16-byte load feeds several MFMAs (4-byte operand loads made the
// first version VMEM-issue bound: 10 us of load issue next to 12 us of MFMA per step, measured
// with tools/ubench_vgru.hip):
//   weights  Wp[k][j][4]   = {W_r[j,k], W_z[j,k], W_n[j,k], 0}  - one load = the A operands of the
//                             three gate MFMAs of a k;
//   state    hP[k/4][b][4] = h[4(k/4) .. +3, b]                  - one load = the B operands of four
//                             k steps, and the epilogue stores four consecutive j per lane.
// An MFMA 32x32x2 consumes a k pair (lane half kk = 0/1); pairs are formed as (8o+e, 8o+4+e),
// e = 0..3, inside a "k octet" o, which is what the two layouts deliver to the two lane halves.
//
// Workgroup = 8 waves on one (32 hidden x 32 column) tile: waves 0-3 split K of the recurrent
// product W_hh h, waves 4-7 split K of the input product (layer 1: W_ih h0; layer 0: the one-hot
// input generated in registers, K = 22 -> 24, wave 4 only).  The K slices are summed through LDS
// and waves 0-3 apply the gate maths (ATen gru_cell: h' = (h - n) z + n).
// Block b runs on XCD b % 8; XCD x owns hidden tiles 2x, 2x+1 of both layers, so the weights it
// streams every step (1.6 MB) stay in its 4 MB L2.
#include "common.h"

namespace dmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float vsigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

struct VStepArgs {
  const uint8_t* codes;     // row t of the alignment (L bytes) or nullptr
  const float* wxP[2];      // [layer]: input weights  [Kx][512][4]
  const float* whP[2];      // [layer]: hidden weights [512][512][4]
  const float* bias[2];     // [layer]: [4][512]: r (b_ir+b_hr), z (b_iz+b_hz), b_in, b_hn
  const float* h0_prev;     // packed state [128][Lb][4]
  float* h0_next;
  const float* h1_prev;
  float* h1_next;
  int L, Lb;
  int do_l0, do_l1;
};

// One wave's quarter of a K = 512 contraction: octets o = w, w+4, ..., w+60, three accumulators.
// Hand-staged: the 10 loads of the next two octets are issued before the 24 MFMAs of the current.
__device__ __forceinline__ void k512_octets(const float* __restrict__ wp, const float* __restrict__ xp,
                                            int Lb, int w, int kk, f32x16& a0, f32x16& a1, f32x16& a2) {
  float4 wv[2][2][4];
  float4 xv[2][2];
  auto load_chunk = [&](int buf, int c) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int o = w + 4 * (2 * c + u);
      xv[buf][u] = *reinterpret_cast<const float4*>(xp + (int64_t)(2 * o + kk) * Lb * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        wv[buf][u][e] = *reinterpret_cast<const float4*>(wp + (int64_t)(8 * o + 4 * kk + e) * 2048);
    }
  };
  load_chunk(0, 0);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c + 1 < 8) load_chunk((c + 1) & 1, c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float xs[4] = {xv[c & 1][u].x, xv[c & 1][u].y, xv[c & 1][u].z, xv[c & 1][u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[c & 1][u][e].x, xs[e], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[c & 1][u][e].y, xs[e], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[c & 1][u][e].z, xs[e], a2, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// grid: 8 * 2 * (Lb/32) * 2 blocks, 1-D (heavier layer-1 tiles first)   block: 512
__global__ __launch_bounds__(512) void vgru_step_kernel(VStepArgs a) {
  __shared__ float red[4][4][16][64];
  const int nbt = a.Lb >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rest = slot >> 1;
  const int layer = 1 - rest / nbt;
  if (layer == 0 && !a.do_l0) return;
  if (layer == 1 && !a.do_l1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int part = wave >> 2, w = wave & 3;
  const int kk = lane >> 5, li = lane & 31;
  const int b0 = (rest % nbt) * 32, j0 = (2 * xcd + (slot & 1)) * 32;
  const int Lb = a.Lb;
  const float* hprev = (layer == 0) ? a.h0_prev : a.h1_prev;

  f32x16 acc_r, acc_z, acc_t;       // third = W_hn h (part 0) or W_in x (part 1)
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_t[r] = 0.f; }

  if (part == 0) {
    k512_octets(a.whP[layer] + (int64_t)(j0 + li) * 4, hprev + (int64_t)(b0 + li) * 4, Lb, w, kk,
                acc_r, acc_z, acc_t);
  } else if (layer == 1) {
    k512_octets(a.wxP[1] + (int64_t)(j0 + li) * 4, a.h0_prev + (int64_t)(b0 + li) * 4, Lb, w, kk,
                acc_r, acc_z, acc_t);
  } else if (w == 0) {
    // layer 0 input: one-hot of the residue code, K = 24 (rows 22, 23 of the packed weights are 0)
    const int b = b0 + li;
    const int code = (b < a.L) ? (int)a.codes[b] : 0;
    const float* wp = a.wxP[0] + (int64_t)(j0 + li) * 4;
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = 8 * o + 4 * kk + e;
        const float4 w4 = *reinterpret_cast<const float4*>(wp + (int64_t)k * 2048);
        const float x = (code == k) ? 1.0f : 0.0f;
        acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, x, acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, x, acc_z, 0, 0, 0);
        acc_t = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, x, acc_t, 0, 0, 0);
      }
  }

  // ---- stage 1: the input-product waves hand their partial sums to the recurrent-product waves
  if (part == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[w][0][r][lane] = acc_r[r];
      red[w][1][r][lane] = acc_z[r];
      red[w][2][r][lane] = acc_t[r];
    }
  }
  __syncthreads();
  f32x16 acc_in;
  if (part == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc_r[r] += red[w][0][r][lane];
      acc_z[r] += red[w][1][r][lane];
      acc_in[r] = red[w][2][r][lane];
    }
  }
  __syncthreads();
  // ---- stage 2: sum the four K slices, each wave finishing a quarter of the tile
  if (part == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[w][0][r][lane] = acc_r[r];
      red[w][1][r][lane] = acc_z[r];
      red[w][2][r][lane] = acc_in[r];
      red[w][3][r][lane] = acc_t[r];
    }
  }
  __syncthreads();
  if (part != 0) return;
  const float* bias = a.bias[layer];
  float* hnext = (layer == 0) ? a.h0_next : a.h1_next;
  const int j4 = j0 + 8 * w + 4 * kk;              // rows (4w+q): j = j4 + q, q = 0..3
  const int b = b0 + li;
  const float4 bR = *reinterpret_cast<const float4*>(bias + j4);
  const float4 bZ = *reinterpret_cast<const float4*>(bias + 512 + j4);
  const float4 bI = *reinterpret_cast<const float4*>(bias + 1024 + j4);
  const float4 bH = *reinterpret_cast<const float4*>(bias + 1536 + j4);
  const float br[4] = {bR.x, bR.y, bR.z, bR.w}, bz[4] = {bZ.x, bZ.y, bZ.z, bZ.w};
  const float bi[4] = {bI.x, bI.y, bI.z, bI.w}, bh[4] = {bH.x, bH.y, bH.z, bH.w};
  const int64_t hoff = ((int64_t)(j4 >> 2) * Lb + b) * 4;
  const float4 hp4 = *reinterpret_cast<const float4*>(hprev + hoff);
  const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
  float hn[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = w * 4 + q;
    float s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      s[g] = (red[0][g][r][lane] + red[1][g][r][lane]) + (red[2][g][r][lane] + red[3][g][r][lane]);
    const float rg = vsigmoid(s[0] + br[q]);
    const float zg = vsigmoid(s[1] + bz[q]);
    const float ng = tanhf((s[2] + bi[q]) + rg * (s[3] + bh[q]));
    hn[q] = (hp[q] - ng) * zg + ng;
  }
  *reinterpret_cast<float4*>(hnext + hoff) = make_float4(hn[0], hn[1], hn[2], hn[3]);
}

// out[l][j] = hP[j/4][l][j%4]
__global__ __launch_bounds__(128) void vgru_out_kernel(const float* __restrict__ hP, int Lb,
                                                       float* __restrict__ out) {
  const int l = blockIdx.x, j4 = threadIdx.x;
  const float4 v = *reinterpret_cast<const float4*>(hP + ((int64_t)j4 * Lb + l) * 4);
  *reinterpret_cast<float4*>(out + (int64_t)l * WIDTH + 4 * j4) = v;
}

int gru_vertical(dmp_ctx* c, const uint8_t* d_msa, int N, int L, float* d_out, hipStream_t s) {
  const int Lb = round_up(L, 32);
  const size_t hbytes = sizeof(float) * WIDTH * Lb;
  DMP_HIP(hipMemsetAsync(c->hT[0][0], 0, hbytes, s));
  DMP_HIP(hipMemsetAsync(c->hT[1][0], 0, hbytes, s));
  const Weights& W = c->W;
  VStepArgs a{};
  a.wxP[0] = W.v_wx[0]; a.whP[0] = W.v_wh[0]; a.bias[0] = W.v_b0;
  a.wxP[1] = W.v_wx[1]; a.whP[1] = W.v_wh[1]; a.bias[1] = W.v_b1;
  a.L = L; a.Lb = Lb;
  dim3 grid(8 * 2 * (Lb / 32) * 2);
  for (int t = 0; t <= N; ++t) {
    a.codes = (t < N) ? d_msa + (int64_t)t * L : nullptr;
    a.do_l0 = (t < N);
    a.do_l1 = (t >= 1);
    a.h0_prev = c->hT[0][t & 1];
    a.h0_next = c->hT[0][(t + 1) & 1];
    a.h1_prev = c->hT[1][(t + 1) & 1];   // layer 1 runs step t-1: parity (t-1)&1
    a.h1_next = c->hT[1][t & 1];
    hipLaunchKernelGGL(vgru_step_kernel, grid, dim3(512), 0, s, a);
  }
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(vgru_out_kernel, dim3(L), dim3(128), 0, s, c->hT[1][N & 1], Lb, d_out);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
