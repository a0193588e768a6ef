// gru_vertical: the 2-layer GRU that runs DOWN the alignment (reference network.py:189, 223-224:
// time axis = N sequences, batch = L columns, hidden 512).
//
// One launch per time step computes layer 0 at step t and layer 1 at step t-1 (both read the same
// h0 state) as GEMMs  gates^T[j, b] = sum_k W[j, k] * x[k, b]  with the hidden index j on the MFMA M
// axis and the alignment column b on the N axis.
//
// Arithmetic: float32-grade products on the f16 matrix cores, the same scheme as conv_f16.h.  Every
// operand is the sum of two f16 pieces (weights: S*w = w0 + w1 with a power-of-two S per layer;
// state: 1024*h = h0 + h1, |h| < 1); v_mfma_f32_32x32x16_f16 accumulates w0 h1 + w1 h0 + w0 h0 in
// float32 and the sums are scaled back by 1/(1024 S) (exact).  The dropped w1 h1 term is 2^-22 of a
// product.  The one-hot layer-0 input is the exact f16 value 1024 (low piece 0: two products).
// The previous version of this kernel ran the exact-f32 MFMA (32x32x2): 17.6 us of matrix-core time
// per step against 3.3 us here.
//
// Layouts (one 16-byte load = one MFMA operand):
//   weights  Wq[piece 2][gate 3][k/8][512 j][8]  f16     A operand of (gate, piece) for 8 k
//   state    hH[piece 2][k/8 = 64][Lb][8]        f16     B operand;  written by the epilogue
//            hP[j/4 = 128][Lb][4]                f32     the state itself, for h' = (h - n) z + n
//
// Workgroup = 8 waves = two groups of four, each group one K = 512 product on a (32 hidden x 32
// column) tile split four ways (layer 1: recurrent and input product of one tile; layer 0: the
// recurrent products of two tiles, the K = 22 -> 32 one-hot input product generated in registers by
// wave 0 of the group).  The K slices are summed through LDS (two rounds: r and z, then the third
// gate) and four waves per tile apply the gate maths (ATen gru_cell: h' = (h - n) z + n).
// Block b runs on XCD b % 8; XCD x owns hidden tiles 2x, 2x+1 of both layers, so the weights it
// streams every step (1.6 MB) stay in its 4 MB L2.
// The N + 1 dependent launches are replayed from hipGraph chains of 128 (or 16) step nodes; what
// changes between chunks lives in a small device record (VRun) written by a 1-thread kernel.
#include "common.h"

namespace dmp {

#ifndef VG_CH
#define VG_CH 1      // MFMA steps per load chunk (double-buffered)
#endif
#ifndef VG_OCC
#define VG_OCC 4     // waves per SIMD the register budget is compiled for (4 = 128 VGPRs: two step
                     // workgroups per CU, or one beside a convolution workgroup)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 vg_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float vsigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ f32x16 vg_mfma(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vg_f16x8, a), __builtin_bit_cast(vg_f16x8, b),
                                                c, 0, 0, 0);
}

// Per-context constants of the step kernel (baked into the hipGraph nodes) ...
struct VStatic {
  const uint4* wx[2];       // [layer]: input weight pieces   [2][3][KQ][512] x 16 bytes (KQ = 4 / 64)
  const uint4* wh[2];       // [layer]: hidden weight pieces  [2][3][64][512] x 16 bytes
  const float* bias[2];     // [layer]: [4][512]: r (b_ir+b_hr), z (b_iz+b_hz), b_in, b_hn
  float inv_scale[2];
  float* hT[2][2];          // [layer][parity] float32 state [128][Lb][4]
  uint16_t* hH[2][2];       // [layer][parity] f16 pieces of 1024*state [2][64][Lb][8]
};
// ... and what changes from chunk to chunk, read from device memory (vgru_set_run_kernel)
struct VRun {
  const uint8_t* msa;       // N x L residue codes
  int N, L, Lb;
  int t0, t_end;            // node idx of the chunk runs time step t = t0 + idx if t < t_end
};


#ifndef VG_NSUB_N
#define VG_NSUB_N 1
#endif
constexpr int VG_NSUB = VG_NSUB_N;     // 32-column subtiles per wave: the weight operands are reused VG_NSUB times
constexpr int VG_TB = 32 * VG_NSUB;    // columns per tile

struct VAcc { f32x16 r[VG_NSUB], z[VG_NSUB], t[VG_NSUB]; };

// One wave's quarter of a K = 512 contraction: MFMA steps s = w, w+4, ..., w+28 (16 k each), three
// gate accumulators per column subtile.  Hand-staged: the loads of the next chunk are issued before
// the MFMAs of the current one.
// Measured at L = 300 (240 workgroups, 12.3 us per step): the same kernel without MFMAs takes as
// long; with all loads redirected to one cache line 7.6 us.  A CU sustains about 50 GB/s of L1
// misses here whatever the request pattern (rotating the k order per workgroup changes nothing), so
// the step time follows the bytes one CU has to pull: 64-column tiles (VG_NSUB_N = 2: weights
// reused from registers, 0.6x the total L2 traffic, but half as many workgroups pulling 1.25x the
// bytes each) run 18 us per step.
__device__ __forceinline__ void k512_steps(const uint4* __restrict__ wp, const uint4* __restrict__ xp, int Lb,
                                           int w, int kk, VAcc& A) {
  uint4 wv[2][VG_CH][3][2];          // [buffer][step][gate][piece]
  uint4 xv[2][VG_CH][VG_NSUB][2];    // [buffer][step][subtile][piece]
  auto load_chunk = [&](int buf, int c) {
#pragma unroll
    for (int u = 0; u < VG_CH; ++u) {
      const int kq = 2 * (w + 4 * (VG_CH * c + u)) + kk;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int n = 0; n < VG_NSUB; ++n) xv[buf][u][n][p] = xp[(int64_t)(p * 64 + kq) * Lb + 32 * n];
#pragma unroll
        for (int g = 0; g < 3; ++g) wv[buf][u][g][p] = wp[(int64_t)((p * 3 + g) * 64 + kq) * 512];
      }
    }
  };
  load_chunk(0, 0);
#pragma unroll
  for (int c = 0; c < 8 / VG_CH; ++c) {
    if (c + 1 < 8 / VG_CH) load_chunk((c + 1) & 1, c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < VG_CH; ++u) {
#pragma unroll
      for (int n = 0; n < VG_NSUB; ++n) {
        const uint4 x0 = xv[c & 1][u][n][0], x1 = xv[c & 1][u][n][1];
        A.r[n] = vg_mfma(wv[c & 1][u][0][0], x1, A.r[n]);
        A.z[n] = vg_mfma(wv[c & 1][u][1][0], x1, A.z[n]);
        A.t[n] = vg_mfma(wv[c & 1][u][2][0], x1, A.t[n]);
        A.r[n] = vg_mfma(wv[c & 1][u][0][1], x0, A.r[n]);
        A.z[n] = vg_mfma(wv[c & 1][u][1][1], x0, A.z[n]);
        A.t[n] = vg_mfma(wv[c & 1][u][2][1], x0, A.t[n]);
        A.r[n] = vg_mfma(wv[c & 1][u][0][0], x0, A.r[n]);
        A.z[n] = vg_mfma(wv[c & 1][u][1][0], x0, A.z[n]);
        A.t[n] = vg_mfma(wv[c & 1][u][2][0], x0, A.t[n]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

#ifndef VG_LDS2
#define VG_LDS2 1    // 1: two-round K reduction through 72 KB of LDS instead of one round through 104 KB.
                     // Standalone both take 24.5 ms at L=300, N=2000; in throughput mode the small
                     // footprint (72 KB, 128 VGPRs) fits beside one convolution workgroup of another
                     // target: 5.95 -> 6.13 structures/s
#endif
constexpr int VG_LDS_BYTES = (8 * (VG_LDS2 ? 2 : 3) + 2) * 16 * 64 * 4;   // per-wave partial sums + layer-0 input part

// column pitch of the state buffers / number of workgroups of a step: per XCD 2 hidden tiles x
// (nbt layer-1 tiles + ceil(nbt/2) layer-0 pairs)
__host__ __device__ inline int vgru_pitch(int L) { return (L + VG_TB - 1) / VG_TB * VG_TB; }
__host__ __device__ inline int vgru_grid(int Lb) {
  const int nbt = Lb / VG_TB;
  return 8 * 2 * (nbt + ((nbt + 1) >> 1));
}

// Every workgroup is two groups of four waves, each group one K = 512 product split four ways on a
// (32 hidden x 64 column) tile:
//   layer 1:  group 0 = W_hh h1, group 1 = W_ih h0 of the SAME tile;
//   layer 0:  group g = W_hh h0 of column tile 2p+g (two tiles per workgroup); the K = 32 one-hot
//             input product of a tile is done by wave 0 of its group.
// All workgroups therefore carry the same load (8 waves x 8 MFMA steps x 2 subtiles); at L = 300
// the 128 of them run as a single round.  Per column subtile: every wave stores its partial sums,
// barrier, four waves per tile add them up and apply the gate maths.
// grid: vgru_grid(Lb)   block: 512   dynamic LDS: VG_LDS_BYTES
__global__ __launch_bounds__(512, VG_OCC) void vgru_step_kernel(VStatic st, const VRun* __restrict__ run,
                                                                int idx) {
  extern __shared__ __attribute__((aligned(16))) float vg_red[];   // [8 waves][3][16][64] | [2 groups][16][64]
  const int t = run->t0 + idx;
  if (t >= run->t_end) return;
  const int N = run->N, L = run->L, Lb = run->Lb;
  const int nbt = Lb / VG_TB;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rest = slot >> 1;                       // [0, nbt): layer 1 tiles, [nbt, nbt + ceil(nbt/2)): layer 0 pairs
  const int layer = rest < nbt ? 1 : 0;
  if (layer == 0 && !(t < N)) return;
  if (layer == 1 && !(t >= 1)) return;
  const int par = t & 1;      // layer 0 reads parity t, writes t+1; layer 1 (step t-1) reads t+1, writes t
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w = wave & 3;
  const int kk = lane >> 5, li = lane & 31;
  const int j0 = (2 * xcd + (slot & 1)) * 32;
  const int bt = layer ? rest : 2 * (rest - nbt) + grp;      // this group's column tile
  const bool valid = bt < nbt;                               // odd tile count: the last pair is half empty
  const int b0 = bt * VG_TB;

  VAcc A;                                 // A.t = W_hn h (recurrent groups) or W_in x (layer-1 input group)
  f32x16 acc_in[VG_NSUB];                 // layer 0, wave 0 of a group: W_in x of the one-hot input
#pragma unroll
  for (int n = 0; n < VG_NSUB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) { A.r[n][r] = 0.f; A.z[n][r] = 0.f; A.t[n][r] = 0.f; acc_in[n][r] = 0.f; }

  if (layer == 1) {
    const uint4* wp = (grp == 0 ? st.wh[1] : st.wx[1]) + (j0 + li);
    const uint4* xp = reinterpret_cast<const uint4*>(grp == 0 ? st.hH[1][par ^ 1] : st.hH[0][par]) + (b0 + li);
    k512_steps(wp, xp, Lb, w, kk, A);
  } else if (valid) {
    k512_steps(st.wh[0] + (j0 + li), reinterpret_cast<const uint4*>(st.hH[0][par]) + (b0 + li), Lb, w, kk, A);
    if (w == 0) {
      // layer 0 input: one-hot of the residue code (value 1024 = the state scale), K = 32 (rows
      // 22..31 of the packed weights are 0)
      const uint4* wp = st.wx[0] + (j0 + li);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kq = 2 * s + kk;
        uint4 wq[3][2];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int p = 0; p < 2; ++p) wq[g][p] = wp[(int64_t)((p * 3 + g) * 4 + kq) * 512];
#pragma unroll
        for (int n = 0; n < VG_NSUB; ++n) {
          const int b = b0 + 32 * n + li;
          const int code = (b < L) ? (int)run->msa[(int64_t)t * L + b] : 0;
          const int d = code - 8 * kq;                   // position of the hot element among this lane's 8 k
          const unsigned hot = (d >= 0 && d < 8) ? (0x6400u << (16 * (d & 1))) : 0u;
          uint4 x;
          x.x = (d >> 1) == 0 ? hot : 0u;
          x.y = (d >> 1) == 1 ? hot : 0u;
          x.z = (d >> 1) == 2 ? hot : 0u;
          x.w = (d >> 1) == 3 ? hot : 0u;
          A.r[n] = vg_mfma(wq[0][1], x, A.r[n]);
          A.z[n] = vg_mfma(wq[1][1], x, A.z[n]);
          acc_in[n] = vg_mfma(wq[2][1], x, acc_in[n]);
          A.r[n] = vg_mfma(wq[0][0], x, A.r[n]);
          A.z[n] = vg_mfma(wq[1][0], x, A.z[n]);
          acc_in[n] = vg_mfma(wq[2][0], x, acc_in[n]);
        }
      }
    }
  }

  const bool finisher = layer == 1 ? grp == 0 : valid;      // layer 1: waves 0-3; layer 0: every valid group
  const float* bias = st.bias[layer];
  const float inv = st.inv_scale[layer];
  const float* hprev = layer ? st.hT[1][par ^ 1] : st.hT[0][par];
  float* hnext = layer ? st.hT[1][par] : st.hT[0][par ^ 1];
  uint16_t* gnext = layer ? st.hH[1][par] : st.hH[0][par ^ 1];
  const int j4 = j0 + 8 * w + 4 * kk;              // rows (4w+q): j = j4 + q, q = 0..3
  const float4 bR = *reinterpret_cast<const float4*>(bias + j4);
  const float4 bZ = *reinterpret_cast<const float4*>(bias + 512 + j4);
  const float4 bI = *reinterpret_cast<const float4*>(bias + 1024 + j4);
  const float4 bH = *reinterpret_cast<const float4*>(bias + 1536 + j4);
  const float br[4] = {bR.x, bR.y, bR.z, bR.w}, bz[4] = {bZ.x, bZ.y, bZ.z, bZ.w};
  const float bi[4] = {bI.x, bI.y, bI.z, bI.w}, bh[4] = {bH.x, bH.y, bH.z, bH.w};
  // Partial sums per wave: all three gates at once (VG_LDS2 = 0: 104 KB of LDS, one barrier) or r, z
  // first and the third gate in a second round through the same buffer (VG_LDS2 = 1: 72 KB, three
  // barriers) - the smaller footprint lets a step workgroup share a CU with a convolution workgroup.
  constexpr int WS = (VG_LDS2 ? 2 : 3) * 16 * 64;           // floats per wave
  float* mine = vg_red + (int64_t)wave * WS;
  float* red_in = vg_red + 8 * WS + grp * (16 * 64);
  const float* g0 = vg_red + (int64_t)(layer ? 0 : 4 * grp) * WS;   // the recurrent group's 4 waves
  const float* g1 = vg_red + (int64_t)4 * WS;                        // layer 1: the input group's
  auto sum4 = [&](const float* p) { return (p[0] + p[WS]) + (p[2 * WS] + p[3 * WS]); };

#pragma unroll
  for (int n = 0; n < VG_NSUB; ++n) {
    // ---- every wave stores its partial sums of subtile n; barrier; four waves per tile finish it
    if (n > 0) __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mine[(0 * 16 + r) * 64 + lane] = A.r[n][r];
      mine[(1 * 16 + r) * 64 + lane] = A.z[n][r];
      if (!VG_LDS2) mine[(2 * 16 + r) * 64 + lane] = A.t[n][r];
    }
    if (layer == 0 && w == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red_in[r * 64 + lane] = acc_in[n][r];
    }
    __syncthreads();
    float S[4][4];      // [q][r, z, in, hn]
    if (finisher) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = w * 4 + q;
#pragma unroll
        for (int g = 0; g < (VG_LDS2 ? 2 : 3); ++g) {
          float v = sum4(g0 + (g * 16 + r) * 64 + lane);
          if (layer && g < 2) v += sum4(g1 + (g * 16 + r) * 64 + lane);
          S[q][g == 2 ? 3 : g] = v;
        }
        if (!VG_LDS2 && layer) S[q][2] = sum4(g1 + (2 * 16 + r) * 64 + lane);
        if (!layer) S[q][2] = red_in[r * 64 + lane];
      }
    }
    if (VG_LDS2) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = A.t[n][r];
      __syncthreads();
      if (finisher) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = w * 4 + q;
          S[q][3] = sum4(g0 + r * 64 + lane);
          if (layer) S[q][2] = sum4(g1 + r * 64 + lane);
        }
      }
    }
    if (!finisher) continue;
    const int b = b0 + 32 * n + li;
    const int64_t hoff = ((int64_t)(j4 >> 2) * Lb + b) * 4;
    const float4 hp4 = *reinterpret_cast<const float4*>(hprev + hoff);
    const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
    float hn[4];
    unsigned short q0[4], q1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* s = S[q];
      const float rg = vsigmoid(s[0] * inv + br[q]);
      const float zg = vsigmoid(s[1] * inv + bz[q]);
      const float ng = tanhf((s[2] * inv + bi[q]) + rg * (s[3] * inv + bh[q]));
      hn[q] = (hp[q] - ng) * zg + ng;
      const float hs = hn[q] * VGRU_STATE_SCALE;
      const _Float16 p0 = (_Float16)hs;
      const _Float16 p1 = (_Float16)(hs - (float)p0);
      q0[q] = __builtin_bit_cast(unsigned short, p0);
      q1[q] = __builtin_bit_cast(unsigned short, p1);
    }
    *reinterpret_cast<float4*>(hnext + hoff) = make_float4(hn[0], hn[1], hn[2], hn[3]);
    const int64_t goff = ((int64_t)(j4 >> 3) * Lb + b) * 8 + (j4 & 7);
    *reinterpret_cast<uint2*>(gnext + goff) =
        make_uint2((unsigned)q0[0] | ((unsigned)q0[1] << 16), (unsigned)q0[2] | ((unsigned)q0[3] << 16));
    *reinterpret_cast<uint2*>(gnext + (int64_t)64 * Lb * 8 + goff) =
        make_uint2((unsigned)q1[0] | ((unsigned)q1[1] << 16), (unsigned)q1[2] | ((unsigned)q1[3] << 16));
  }
}

// out[l][j] = hP[j/4][l][j%4]
__global__ __launch_bounds__(128) void vgru_out_kernel(const float* __restrict__ hP, int Lb,
                                                       float* __restrict__ out) {
  const int l = blockIdx.x, j4 = threadIdx.x;
  const float4 v = *reinterpret_cast<const float4*>(hP + ((int64_t)j4 * Lb + l) * 4);
  *reinterpret_cast<float4*>(out + (int64_t)l * WIDTH + 4 * j4) = v;
}

int gru_vertical(dmp_ctx* c, const uint8_t* d_msa, int N, int L, float* d_out, hipStream_t s) {
  return gru_vertical_steps(c, d_msa, N, L, 0, N + 1, d_out, s);
}

__global__ void vgru_set_run_kernel(VRun* run, const uint8_t* msa, int N, int L, int Lb, int t0, int t_end) {
  run->msa = msa; run->N = N; run->L = L; run->Lb = Lb; run->t0 = t0; run->t_end = t_end;
}

// A chain of `len` step-kernel nodes (idx = 0..len-1) for a given grid; built once per context and
// grid size.  Replaying it costs one host call instead of `len` launches, and the gap between
// dependent kernels is 1.8 us instead of 2.8 us (tools/ubench_launch.hip).
static int vgru_graph(dmp_ctx* c, int grid, int len, hipGraphExec_t* out) {
  const int64_t key = ((int64_t)grid << 16) | len;
  auto it = c->vgru_graphs.find(key);
  if (it != c->vgru_graphs.end()) { *out = (hipGraphExec_t)it->second; return DMP_OK; }
  const Weights& W = c->W;
  VStatic st{};
  for (int l = 0; l < 2; ++l) {
    st.wx[l] = reinterpret_cast<const uint4*>(W.v_wx[l]);
    st.wh[l] = reinterpret_cast<const uint4*>(W.v_wh[l]);
    st.inv_scale[l] = W.v_inv_scale[l];
    for (int p = 0; p < 2; ++p) { st.hT[l][p] = c->hT[l][p]; st.hH[l][p] = c->hH[l][p]; }
  }
  st.bias[0] = W.v_b0; st.bias[1] = W.v_b1;
  const VRun* run = reinterpret_cast<const VRun*>(c->vgru_run);
  DMP_HIP(hipFuncSetAttribute((const void*)vgru_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              VG_LDS_BYTES));
  hipGraph_t g;
  DMP_HIP(hipGraphCreate(&g, 0));
  hipGraphNode_t prev = nullptr;
  for (int idx = 0; idx < len; ++idx) {
    int idx_arg = idx;
    void* params[3] = {(void*)&st, (void*)&run, (void*)&idx_arg};
    hipKernelNodeParams kp{};
    kp.func = (void*)vgru_step_kernel;
    kp.gridDim = dim3(grid);
    kp.blockDim = dim3(512);
    kp.sharedMemBytes = VG_LDS_BYTES;
    kp.kernelParams = params;
    kp.extra = nullptr;
    hipGraphNode_t node;
    hipError_t e = hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); return hip_fail(e, "hipGraphAddKernelNode", __FILE__, __LINE__); }
    prev = node;
  }
  hipGraphExec_t ge;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
  c->vgru_graphs[key] = (void*)ge;
  *out = ge;
  return DMP_OK;
}

int gru_vertical_steps(dmp_ctx* c, const uint8_t* d_msa, int N, int L, int t_lo, int t_hi, float* d_out,
                       hipStream_t s) {
  const int Lb = vgru_pitch(L);
  const size_t hbytes = sizeof(float) * WIDTH * Lb;
  if (t_lo <= 0) {
    t_lo = 0;
    for (int l = 0; l < 2; ++l) {
      DMP_HIP(hipMemsetAsync(c->hT[l][0], 0, hbytes, s));
      DMP_HIP(hipMemsetAsync(c->hH[l][0], 0, hbytes, s));     // 2 pieces x 512 x Lb x 2 bytes
    }
  }
  if (t_hi > N + 1) t_hi = N + 1;
  const int grid = vgru_grid(Lb);
  VRun* run = reinterpret_cast<VRun*>(c->vgru_run);
  for (int t = t_lo; t < t_hi;) {
    const int len = (t_hi - t > VGRU_CHUNK / 2) ? VGRU_CHUNK : VGRU_CHUNK_SMALL;
    hipGraphExec_t ge;
    int rc = vgru_graph(c, grid, len, &ge);
    if (rc) return rc;
    hipLaunchKernelGGL(vgru_set_run_kernel, dim3(1), dim3(1), 0, s, run, d_msa, N, L, Lb, t, t_hi);
    DMP_HIP(hipGraphLaunch(ge, s));
    t += len;
  }
  DMP_LAUNCH_CHECK();
  if (t_hi == N + 1) {
    hipLaunchKernelGGL(vgru_out_kernel, dim3(L), dim3(128), 0, s, c->hT[1][N & 1], Lb, d_out);
    DMP_LAUNCH_CHECK();
  }
  return DMP_OK;
}

}  // namespace dmp
