// GRU kernels.
//  * gru_vertical: the 2-layer GRU that runs DOWN the alignment (reference network.py:189,
//    223-224: time axis = N sequences, batch = L columns).  One launch per time step computes
//    layer 0 at step t and layer 1 at step t-1 (both read the same h0 state), as f32-MFMA
//    GEMMs  gates^T[j, b] = sum_k W^T[k, j] * X^T[k, b]  with the hidden index on the MFMA M
//    axis and the batch (alignment column) on the N axis, so weights, state reads and state
//    writes are all 128-byte coalesced.  The one-hot layer-0 input is generated in registers.
//  * gru_bidir: bidirectional multi-layer GRU along the sequence with batch 1 (hgru,
//    network.py:190/225; coord_gru, network.py:211/253).
// Gate maths follow ATen's gru_cell: r,z = sigmoid(gi+gh), n = tanh(gi_n + r*gh_n),
// h' = (h - n)*z + n.
#include "common.h"

namespace dmp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct VStepArgs {
  const uint8_t* codes;     // row t of the alignment (L bytes) or nullptr
  const float* wxT[2];      // [layer]: x-part weights, [Kx][1536]
  const float* whT[2];      // [layer]: h-part weights, [512][1536]
  const float* bias[2];     // [layer]: [4][512]
  const float* h0_prev;     // [512][Lb]
  float* h0_next;
  const float* h1_prev;
  float* h1_next;
  int L, Lb;
  int do_l0, do_l1;
};

// grid: (Lb/32, 16, 2)   block: 256 (4 waves split K)
__global__ __launch_bounds__(256) void vgru_step_kernel(VStepArgs a) {
  __shared__ float red[4][4][16][64];
  const int layer = blockIdx.z;
  if (layer == 0 && !a.do_l0) return;
  if (layer == 1 && !a.do_l1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int b0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  const int Lb = a.Lb;

  f32x16 acc_r, acc_z, acc_in, acc_hn;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_in[r] = 0.f; acc_hn[r] = 0.f; }

  const float* wx = a.wxT[layer] + j0 + li;
  const float* wh = a.whT[layer] + j0 + li;
  const float* hprev = (layer == 0) ? a.h0_prev : a.h1_prev;

  // ---- input part
  if (layer == 0) {
    const int b = b0 + li;
    const int code = (b < a.L) ? (int)a.codes[b] : 0;
    // K = 24 (22 real + 2 zero rows): 12 k-pairs, 3 per wave
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int k = 2 * (wave + 4 * p) + kk;
      const float x = (code == k) ? 1.0f : 0.0f;
      const float* wk = wx + (int64_t)k * 1536;
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[0], x, acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[512], x, acc_z, 0, 0, 0);
      acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[1024], x, acc_in, 0, 0, 0);
    }
  } else {
    const float* xin = a.h0_prev + b0 + li;
#pragma unroll 8
    for (int p = 0; p < 64; ++p) {
      const int k = 2 * (wave + 4 * p) + kk;
      const float x = xin[(int64_t)k * Lb];
      const float* wk = wx + (int64_t)k * 1536;
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[0], x, acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[512], x, acc_z, 0, 0, 0);
      acc_in = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[1024], x, acc_in, 0, 0, 0);
    }
  }
  // ---- recurrent part
  {
    const float* hin = hprev + b0 + li;
#pragma unroll 8
    for (int p = 0; p < 64; ++p) {
      const int k = 2 * (wave + 4 * p) + kk;
      const float x = hin[(int64_t)k * Lb];
      const float* wk = wh + (int64_t)k * 1536;
      acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[0], x, acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[512], x, acc_z, 0, 0, 0);
      acc_hn = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[1024], x, acc_hn, 0, 0, 0);
    }
  }
  // ---- reduce the 4 K-slices through LDS, then gate maths on a quarter of the tile per wave
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    red[wave][0][r][lane] = acc_r[r];
    red[wave][1][r][lane] = acc_z[r];
    red[wave][2][r][lane] = acc_in[r];
    red[wave][3][r][lane] = acc_hn[r];
  }
  __syncthreads();
  const float* bias = a.bias[layer];
  float* hnext = (layer == 0) ? a.h0_next : a.h1_next;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = wave * 4 + q;
    float s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      s[g] = (red[0][g][r][lane] + red[1][g][r][lane]) + (red[2][g][r][lane] + red[3][g][r][lane]);
    const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
    const int b = b0 + li;
    const float rg = sigmoidf_(s[0] + bias[j]);
    const float zg = sigmoidf_(s[1] + bias[512 + j]);
    const float ng = tanhf((s[2] + bias[1024 + j]) + rg * (s[3] + bias[1536 + j]));
    const float hp = hprev[(int64_t)j * Lb + b];
    hnext[(int64_t)j * Lb + b] = (hp - ng) * zg + ng;
  }
}

// out[l][j] = hT[j][l]
__global__ __launch_bounds__(256) void vgru_out_kernel(const float* __restrict__ hT, int L, int Lb,
                                                       float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int l0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) tile[r][tx] = hT[(int64_t)(j0 + r) * Lb + l0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (l0 + r < L) out[(int64_t)(l0 + r) * WIDTH + j0 + tx] = tile[tx][r];
}

int gru_vertical(dmp_ctx* c, const uint8_t* d_msa, int N, int L, float* d_out, hipStream_t s) {
  const int Lb = round_up(L, 32);
  const size_t hbytes = sizeof(float) * WIDTH * Lb;
  DMP_HIP(hipMemsetAsync(c->hT[0][0], 0, hbytes, s));
  DMP_HIP(hipMemsetAsync(c->hT[1][0], 0, hbytes, s));
  const Weights& W = c->W;
  VStepArgs a{};
  a.wxT[0] = W.v_wih0T; a.whT[0] = W.v_whh0T; a.bias[0] = W.v_b0;
  a.wxT[1] = W.v_wih1T; a.whT[1] = W.v_whh1T; a.bias[1] = W.v_b1;
  a.L = L; a.Lb = Lb;
  dim3 grid(Lb / 32, WIDTH / 32, 2);
  for (int t = 0; t <= N; ++t) {
    a.codes = (t < N) ? d_msa + (int64_t)t * L : nullptr;
    a.do_l0 = (t < N);
    a.do_l1 = (t >= 1);
    a.h0_prev = c->hT[0][t & 1];
    a.h0_next = c->hT[0][(t + 1) & 1];
    a.h1_prev = c->hT[1][(t + 1) & 1];   // layer 1 runs step t-1: parity (t-1)&1
    a.h1_next = c->hT[1][t & 1];
    hipLaunchKernelGGL(vgru_step_kernel, grid, dim3(256), 0, s, a);
  }
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(vgru_out_kernel, dim3(Lb / 32, WIDTH / 32), dim3(256), 0, s,
                     c->hT[1][N & 1], L, Lb, d_out);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// bidirectional sequence GRU, batch 1
// ---------------------------------------------------------------------------------------
struct SeqArgs {
  const float* G;        // [T][1536] input projections incl. b_ih (fwd | rev)
  const float* whh[2];   // [768][256]
  const float* bhh[2];   // [768]
  float* out;            // [T][512]
  int T;
};

// grid: 2 (direction)   block: 1024
__global__ __launch_bounds__(1024) void seq_gru_kernel(SeqArgs a) {
  __shared__ __attribute__((aligned(16))) float h[HID2];
  __shared__ float gh[3 * HID2];
  const int dir = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rs = lane >> 4, kc = lane & 15;
  const float* whh = a.whh[dir];
  const float* bhh = a.bhh[dir];
  if (tid < HID2) h[tid] = 0.f;
  __syncthreads();
  for (int step = 0; step < a.T; ++step) {
    const int t = dir ? (a.T - 1 - step) : step;
    float hv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(&h[kc * 16 + q * 4]);
      hv[q * 4 + 0] = v.x; hv[q * 4 + 1] = v.y; hv[q * 4 + 2] = v.z; hv[q * 4 + 3] = v.w;
    }
#pragma unroll 4
    for (int it = 0; it < 12; ++it) {
      const int row = wave * 48 + it * 4 + rs;
      const float4* wr = reinterpret_cast<const float4*>(whh + (int64_t)row * HID2 + kc * 16);
      float sacc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w4 = wr[q];
        sacc = fmaf(w4.x, hv[q * 4 + 0], sacc);
        sacc = fmaf(w4.y, hv[q * 4 + 1], sacc);
        sacc = fmaf(w4.z, hv[q * 4 + 2], sacc);
        sacc = fmaf(w4.w, hv[q * 4 + 3], sacc);
      }
      sacc += __shfl_xor(sacc, 8, 16);
      sacc += __shfl_xor(sacc, 4, 16);
      sacc += __shfl_xor(sacc, 2, 16);
      sacc += __shfl_xor(sacc, 1, 16);
      if (kc == 0) gh[row] = sacc + bhh[row];
    }
    __syncthreads();
    if (tid < HID2) {
      const float* g = a.G + (int64_t)t * 1536 + dir * 768;
      const float rg = sigmoidf_(g[tid] + gh[tid]);
      const float zg = sigmoidf_(g[HID2 + tid] + gh[HID2 + tid]);
      const float ng = tanhf(g[2 * HID2 + tid] + rg * gh[2 * HID2 + tid]);
      const float hn = (h[tid] - ng) * zg + ng;
      h[tid] = hn;
      a.out[(int64_t)t * 512 + dir * HID2 + tid] = hn;
    }
    __syncthreads();
  }
}

int gru_bidir(dmp_ctx* c, int which, const float* d_in, int T, float* d_out, hipStream_t s) {
  const int layers = which == 0 ? 2 : 3;
  const float* in = d_in;
  for (int l = 0; l < layers; ++l) {
    const GruDirW* w = which == 0 ? c->W.hgru[l] : c->W.cgru[l];
    for (int dir = 0; dir < 2; ++dir) {
      GemmArgs g{};
      g.A = in; g.sam = w[dir].nin; g.sak = 1;
      g.B = w[dir].wihT; g.sbk = 768; g.sbn = 1;
      g.C = c->seq_g + dir * 768; g.ldc = 1536;
      g.M = T; g.N = 768; g.K = w[dir].nin;
      g.alpha = 1.f; g.beta = 0.f; g.bias_n = w[dir].bih;
      int rc = gemm_f32(g, s);
      if (rc) return rc;
    }
    float* out = (l == layers - 1) ? d_out : ((l & 1) ? c->seq_b : c->seq_a);
    SeqArgs a{};
    a.G = c->seq_g;
    a.whh[0] = w[0].whh; a.whh[1] = w[1].whh;
    a.bhh[0] = w[0].bhh; a.bhh[1] = w[1].bhh;
    a.out = out; a.T = T;
    hipLaunchKernelGGL(seq_gru_kernel, dim3(2), dim3(1024), 0, s, a);
    DMP_LAUNCH_CHECK();
    in = out;
  }
  return DMP_OK;
}

}  // namespace dmp
