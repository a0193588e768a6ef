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


constexpr int VG_TB = 32;              // columns per tile


// One wave's quarter of a K = 512 contraction: MFMA steps s = w, w+4, ..., w+28 (16 k each), three
// gate accumulators per column subtile.  Hand-staged: the loads of the next chunk are issued before
// the MFMAs of the current one.
// Measured at L = 300 (240 workgroups, 12.3 us per step): the same kernel without MFMAs takes as
// long; with all loads redirected to one cache line 7.6 us.  A CU sustains about 50 GB/s of L1
// misses here whatever the request pattern (rotating the k order per workgroup changes nothing), so
// the step time follows the bytes one CU has to pull.  64-column tiles (weight operands used for two
// column subtiles from registers: 0.6x the total L2 traffic, half as many workgroups pulling 1.25x the
// bytes each, 246 VGPRs) were built and measured in round 2, bit-identical and slower everywhere: alone
// 31.9 against 26.3 ms at L = 300, N = 2000, two targets side by side 43.5 against 34.5 ms, four 119
// against 115 ms, and 3 % less throughput in the scheduler.
__device__ __forceinline__ void k512_steps(const uint4* __restrict__ wp, const uint4* __restrict__ xp, int Lb,
                                           int w, int kk, f32x16& ar, f32x16& az, f32x16& at) {
  // Two operand sets (6 weight + 2 state loads of 16 bytes each) alternate: the loads of step c+1
  // are in flight during the nine MFMAs of step c.  The loop is kept rolled (two steps per trip) so
  // the kernel stays inside the register budget that lets it share a CU with two convolution
  // workgroups; fully unrolled the compiler hoists loads and needs 200 registers.
  const uint4* wq = wp + (int64_t)(2 * w + kk) * 512;
  const uint4* xq = xp + (int64_t)(2 * w + kk) * Lb;
  uint4 a0[6], b0[2], a1[6], b1[2];
  auto load = [&](uint4* a, uint4* b, int c) {           // step c of this wave: k octet pair 8c + 2w + kk
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      b[p] = xq[(int64_t)(p * 64 + 8 * c) * Lb];
#pragma unroll
      for (int g = 0; g < 3; ++g) a[p * 3 + g] = wq[(int64_t)((p * 3 + g) * 64 + 8 * c) * 512];
    }
  };
  auto mfma9 = [&](const uint4* a, const uint4* b) {     // a[piece*3 + gate], b[piece]
    ar = vg_mfma(a[0], b[1], ar);
    az = vg_mfma(a[1], b[1], az);
    at = vg_mfma(a[2], b[1], at);
    ar = vg_mfma(a[3], b[0], ar);
    az = vg_mfma(a[4], b[0], az);
    at = vg_mfma(a[5], b[0], at);
    ar = vg_mfma(a[0], b[0], ar);
    az = vg_mfma(a[1], b[0], az);
    at = vg_mfma(a[2], b[0], at);
  };
  load(a0, b0, 0);
#pragma unroll 1
  for (int c = 0; c < 8; c += 2) {
    load(a1, b1, c + 1);
    mfma9(a0, b0);
    if (c + 2 < 8) load(a0, b0, c + 2);
    mfma9(a1, b1);
  }
}

constexpr int VG_LDS_BYTES = 4 * 2 * 16 * 64 * 4;   // 32 KB: partial sums of two gates from four waves

// column pitch of the state buffers / number of workgroups of a step: per XCD 2 hidden tiles x
// (nbt layer-1 tiles + nbt layer-0 tiles)
__host__ __device__ inline int vgru_pitch(int L) { return (L + VG_TB - 1) / VG_TB * VG_TB; }
__host__ __device__ inline int vgru_grid(int Lb) { return 8 * 2 * 2 * (Lb / VG_TB); }

// Workgroup = 4 waves (one per SIMD) on one (32 hidden x 32 column) tile of one layer; wave w takes
// the K quarter {w, w+4, ...} of the recurrent product and, for layer 1, of the input product as
// well (the r and z accumulators are shared, W_hn h and W_in x are kept apart); for layer 0 wave 0
// adds the K = 32 one-hot input product.  The partial sums go through LDS two gates at a time
// (32 KB); every wave then finishes a quarter of the tile's rows.
// Footprint: 4 waves of at most 160 VGPRs and 32 KB of LDS - exactly what two f16x3 convolution
// workgroups leave free on a CU (512 - 2*176 registers per SIMD lane, 160 - 2*62 KB), so the steps of
// one target run beside the convolutions of another (7.3 structures/s if this kernel cost nothing,
// 6.4 with the previous 8-wave / 72 KB version that needed one of the two convolution slots).
// grid: vgru_grid(Lb)   block: 256   dynamic LDS: VG_LDS_BYTES
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void vgru_step_kernel(VStatic st, const VRun* __restrict__ run, int idx) {
  extern __shared__ __attribute__((aligned(16))) float vg_red[];   // [4 waves][2 gates][16][64]
  const int t = run->t0 + idx;
  if (t >= run->t_end) return;
  const int N = run->N, L = run->L, Lb = run->Lb;
  const int nbt = Lb / VG_TB;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rest = slot >> 1;                       // [0, nbt): layer 1 tiles, [nbt, 2 nbt): layer 0 tiles
  const int layer = rest < nbt ? 1 : 0;
  if (layer == 0 && !(t < N)) return;
  if (layer == 1 && !(t >= 1)) return;
  const int par = t & 1;      // layer 0 reads parity t, writes t+1; layer 1 (step t-1) reads t+1, writes t
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int j0 = (2 * xcd + (slot & 1)) * 32;
  const int b0 = (layer ? rest : rest - nbt) * VG_TB;

  f32x16 acc_r, acc_z, acc_hn, acc_in;    // r and z: both products; W_hn h; W_in x
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_hn[r] = 0.f; acc_in[r] = 0.f; }

  if (layer == 1) {
    k512_steps(st.wh[1] + (j0 + li), reinterpret_cast<const uint4*>(st.hH[1][par ^ 1]) + (b0 + li), Lb, w, kk,
               acc_r, acc_z, acc_hn);
    k512_steps(st.wx[1] + (j0 + li), reinterpret_cast<const uint4*>(st.hH[0][par]) + (b0 + li), Lb, w, kk,
               acc_r, acc_z, acc_in);
  } else {
    k512_steps(st.wh[0] + (j0 + li), reinterpret_cast<const uint4*>(st.hH[0][par]) + (b0 + li), Lb, w, kk,
               acc_r, acc_z, acc_hn);
    if (w == 0) {
      // layer 0 input: one-hot of the residue code (value 1024 = the state scale), K = 32 (rows
      // 22..31 of the packed weights are 0)
      const uint4* wp = st.wx[0] + (j0 + li);
      const int b = b0 + li;
      const int code = (b < L) ? (int)run->msa[(int64_t)t * L + b] : 0;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kq = 2 * s + kk;
        const int d = code - 8 * kq;                     // position of the hot element among this lane's 8 k
        const unsigned hot = (d >= 0 && d < 8) ? (0x6400u << (16 * (d & 1))) : 0u;
        uint4 x;
        x.x = (d >> 1) == 0 ? hot : 0u;
        x.y = (d >> 1) == 1 ? hot : 0u;
        x.z = (d >> 1) == 2 ? hot : 0u;
        x.w = (d >> 1) == 3 ? hot : 0u;
        acc_r = vg_mfma(wp[(int64_t)((1 * 3 + 0) * 4 + kq) * 512], x, acc_r);
        acc_z = vg_mfma(wp[(int64_t)((1 * 3 + 1) * 4 + kq) * 512], x, acc_z);
        acc_in = vg_mfma(wp[(int64_t)((1 * 3 + 2) * 4 + kq) * 512], x, acc_in);
        acc_r = vg_mfma(wp[(int64_t)((0 * 3 + 0) * 4 + kq) * 512], x, acc_r);
        acc_z = vg_mfma(wp[(int64_t)((0 * 3 + 1) * 4 + kq) * 512], x, acc_z);
        acc_in = vg_mfma(wp[(int64_t)((0 * 3 + 2) * 4 + kq) * 512], x, acc_in);
      }
    }
  }

  const float* bias = st.bias[layer];
  const float inv = st.inv_scale[layer];
  const float* hprev = layer ? st.hT[1][par ^ 1] : st.hT[0][par];
  float* hnext = layer ? st.hT[1][par] : st.hT[0][par ^ 1];
  uint16_t* gnext = layer ? st.hH[1][par] : st.hH[0][par ^ 1];
  const int j4 = j0 + 8 * w + 4 * kk;              // rows (4w+q): j = j4 + q, q = 0..3
  const int b = b0 + li;
  const int64_t hoff = ((int64_t)(j4 >> 2) * Lb + b) * 4;
  constexpr int WS = 2 * 16 * 64;                  // floats per wave and round
  float* mine = vg_red + (int64_t)w * WS;
  auto sum4 = [&](const float* p) { return (p[0] + p[WS]) + (p[2 * WS] + p[3 * WS]); };
  float rg[4], zg[4];
  // round 1: r, z
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    mine[(0 * 16 + r) * 64 + lane] = acc_r[r];
    mine[(1 * 16 + r) * 64 + lane] = acc_z[r];
  }
  __syncthreads();
  {
    const float4 bR = *reinterpret_cast<const float4*>(bias + j4);
    const float4 bZ = *reinterpret_cast<const float4*>(bias + 512 + j4);
    const float br[4] = {bR.x, bR.y, bR.z, bR.w}, bz[4] = {bZ.x, bZ.y, bZ.z, bZ.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = w * 4 + q;
      rg[q] = vsigmoid(sum4(vg_red + (0 * 16 + r) * 64 + lane) * inv + br[q]);
      zg[q] = vsigmoid(sum4(vg_red + (1 * 16 + r) * 64 + lane) * inv + bz[q]);
    }
  }
  __syncthreads();
  // round 2: W_in x, W_hn h
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    mine[(0 * 16 + r) * 64 + lane] = acc_in[r];
    mine[(1 * 16 + r) * 64 + lane] = acc_hn[r];
  }
  const float4 hp4 = *reinterpret_cast<const float4*>(hprev + hoff);
  __syncthreads();
  const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
  const float4 bI = *reinterpret_cast<const float4*>(bias + 1024 + j4);
  const float4 bH = *reinterpret_cast<const float4*>(bias + 1536 + j4);
  const float bi[4] = {bI.x, bI.y, bI.z, bI.w}, bh[4] = {bH.x, bH.y, bH.z, bH.w};
  float hn[4];
  unsigned short q0[4], q1[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = w * 4 + q;
    const float s_in = sum4(vg_red + (0 * 16 + r) * 64 + lane);
    const float s_hn = sum4(vg_red + (1 * 16 + r) * 64 + lane);
    const float ng = tanhf((s_in * inv + bi[q]) + rg[q] * (s_hn * inv + bh[q]));
    hn[q] = (hp[q] - ng) * zg[q] + ng;
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
  hipGraph_t g;
  DMP_HIP(hipGraphCreate(&g, 0));
  hipGraphNode_t prev = nullptr;
  for (int idx = 0; idx < len; ++idx) {
    int idx_arg = idx;
    void* params[3] = {(void*)&st, (void*)&run, (void*)&idx_arg};
    hipKernelNodeParams kp{};
    kp.func = (void*)vgru_step_kernel;
    kp.gridDim = dim3(grid);
    kp.blockDim = dim3(256);
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
