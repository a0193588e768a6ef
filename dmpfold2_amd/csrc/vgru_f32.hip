// gru_vertical in the REFERENCE'S ARITHMETIC (network.py:189, 223-224: nn.GRU(22, 512, num_layers=2) in float32): float32
// operands, float32 products and accumulation on v_mfma_f32_16x16x4_f32, gates on the device library's expf / tanhf
// (1-ulp functions, as ATen's vectorised sigmoid / tanh are) - no f16 pieces, no hardware v_exp / v_rcp approximations.
// Selected with option "precision" = 1 (together with the exact-f32 convolution, conv_mode 1) or "vgru_f32" = 1; the
// default path stays the split-f16 form of vgru.hip (three f16 MFMAs per float32 product, 5.3 x the matrix-core rate).
//
// Same decomposition as vgru_persist_kernel (vgru.hip), because a float32 weight is as many bytes as its two f16 pieces:
//   * weight-stationary and persistent: the group's column tiles are partitioned over the 8 XCDs, the hidden units over the
//     32 CUs of an XCD (CU u: units 16u .. 16u+15 of both layers, all gates, all three products), K over the 4 waves of a
//     CU; wave w keeps its K quarter of the two layer-1 products in 192 AGPRs (A operands of the MFMA straight from the
//     accumulation registers) and of the layer-0 recurrent product in 24 KB of LDS;
//   * the state of a column never leaves its XCD: row boundaries are XCD-local barriers (plain stores + sc1 loads);
//   * per row and 32-column tile a wave multiplies its K quarter: 8 super-steps of 16 k, 72 MFMAs each (3 products x 3 gates
//     x 2 column halves x 4 k quads) = 576 MFMAs of 32 cycles per tile = 7.7 us at the spec clock (the f16 form: 1.44);
//   * the four waves' partial sums meet in LDS, 256 threads finish 16 units x 32 columns x 2 layers.
// k order inside a super-step: the state is stored as float4 rows [k/4][column][k%4]; lane (column c, quad q) loads the
// float4 of k = 16 s + 4 q .. + 3 and MFMA i of the super-step takes component i from every lane, i.e. the k set
// {16 s + 4 q + i : q = 0..3} - the weights are laid out to match when they are loaded into the registers (once per launch).
// Summation order (K quarters in wave order, super-steps ascending, i ascending) does not depend on the grouping: a member's
// result is the same bits alone, in any group and as a rider.
//
// Fallback (the device is shared with another process and the 256 workgroups are not all resident -> row barrier time-out
// -> DMP_FAULT_VGRU_HANDOFF -> option "vgru_persistent" = 0): the same kernel, one launch per alignment row, no barrier,
// workgroup -> (XCD, unit slice) from the block id; every launch re-reads its weights, so this is slow (and correct).
#include "vgru.h"

namespace dmp {

struct VStaticF32 {
  const float4* wh0;        // layer-0 recurrent weights  [gate 3][k/4 = 128][512 j] x float4 (k % 4)
  const float4* wx1;        // layer-1 input weights      (same layout)
  const float4* wh1;        // layer-1 recurrent weights
  const float* wx0;         // layer-0 input weights with the embedding folded in: [gate 3][code 22][512 j]
  const float* bias[2];     // [layer]: [4][512]: r (b_ir+b_hr), z (b_iz+b_hz), b_in, b_hn
  float* hT[2][2];          // [layer][parity] float32 state [128][Lb][4]
};

// The inline-assembly MFMAs are invisible to the compiler's hazard recogniser in BOTH directions.  Found on the GPU (first
// build of this kernel, tools/debug_vgru_f32.py): the compiler sinks the zero-initialisation of an accumulator
// (v_mov_b64 v[98:99], 0) to just in front of its first MFMA, and that MFMA then accumulates onto the STALE lower
// half - columns 16..31, rows 4g and 4g+1 wrong by 2e-2, nothing else.  So (1) the accumulators are pinned behind a
// statement that takes all of them and pads wait states before the first MFMA can issue (vf_accumulators_ready), and
// (2) every assembly MFMA carries two wait states of its own in front, for whatever VALU write of an operand the
// compiler may schedule there (free: the matrix pipe is busy 32 cycles per instruction).
__device__ __forceinline__ vp_f32x4 vf_mfma_a(float a, float b, vp_f32x4 c) {                // A from an AGPR
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
  return c;
}
__device__ __forceinline__ void vf_accumulators_ready(vp_f32x4 (&a)[4][2]) {
  asm volatile("s_nop 3" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]),
                           "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]));
}
__device__ __forceinline__ vp_f32x4 vf_mfma_v(float a, float b, vp_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// the compiler does not see the inline-assembly MFMAs: 16 wait states before anything but an MFMA reads their results
// (an 8-pass MFMA needs 11)
__device__ __forceinline__ void vf_mfma_results_ready(vp_f32x4 (&a)[4][2]) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]),
                                       "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]));
}

__device__ __forceinline__ float vf_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int VF_SS = 8;                                  // super-steps (16 k) of a wave's K quarter
static_assert(4 * VF_SS * 3 * 64 == VP_WL0_SLOTS, "layer 0's float32 weights fill the LDS area of the f16 pieces");

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void vgru_persist_f32_kernel(VStaticF32 st, const VGroupRec* __restrict__ rec, VPSync* __restrict__ sync,
                             int* __restrict__ fault, int t_lo, int t_hi, int ntiles, int barrier) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vf_smem[];
  float4* wl0 = reinterpret_cast<float4*>(vf_smem);                                          // [wave][super-step][gate][lane]
  vp_f32x4* red = reinterpret_cast<vp_f32x4*>(vf_smem + VP_WL0_SLOTS * 16);
  float* tab = reinterpret_cast<float*>(vf_smem + (VP_WL0_SLOTS + VP_RED_SLOTS) * 16);
  __shared__ int sh_u, sh_abort;
  __shared__ int sh_tile[VP_MAX_XCD_TILES][4];
  __shared__ unsigned long long sh_msa[VG_MAX_MEMBERS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned xcc;
  if (barrier) {
    // the XCD this workgroup really runs on (its L2 is where the row barrier's plain stores stop)
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (tid == 0) {
      sh_u = (int)__hip_atomic_fetch_add(&sync->count[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh_abort = 0;
    }
  } else {
    xcc = blockIdx.x & 7u;                               // one row per launch: nothing depends on the placement
    if (tid == 0) { sh_u = (int)(blockIdx.x >> 3); sh_abort = 0; }
  }
  __syncthreads();
  const int u = sh_u;                                    // this workgroup's hidden-unit slice on its XCD
  if (u >= 32) {                                         // more than 32 workgroups landed on this XCD: another one is short
    if (tid == 0) atomicOr(fault, DMP_FAULT_VGRU_HANDOFF);
    return;
  }
  const int j0 = 16 * u, lr = lane & 15, lq = lane >> 4;
  const int Lb = ntiles * VG_TB;
  const int c_lo = (int)(((long long)ntiles * xcc) / 8), c_hi = (int)(((long long)ntiles * (xcc + 1)) / 8);

  // ---- this wave's weights: K quarter w, rows j0 .. j0+15.  Register (s, g, i) of lane (row lr, quad lq) holds
  // W[g 512 + j0 + lr][128 w + 16 s + 4 lq + i] = component i of the float4 at k/4 = 32 w + 4 s + lq.
  float WA[VF_SS * 12], WB[VF_SS * 12];                  // layer-1 input (from h0) and recurrent (from h1) products
#pragma unroll
  for (int s = 0; s < VF_SS; ++s)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int64_t off = (int64_t)(g * 128 + 32 * w + 4 * s + lq) * 512 + j0 + lr;
      const float4 a = st.wx1[off], b = st.wh1[off];
      WA[(s * 3 + g) * 4 + 0] = a.x; WA[(s * 3 + g) * 4 + 1] = a.y; WA[(s * 3 + g) * 4 + 2] = a.z; WA[(s * 3 + g) * 4 + 3] = a.w;
      WB[(s * 3 + g) * 4 + 0] = b.x; WB[(s * 3 + g) * 4 + 1] = b.y; WB[(s * 3 + g) * 4 + 2] = b.z; WB[(s * 3 + g) * 4 + 3] = b.w;
      wl0[((w * VF_SS + s) * 3 + g) * 64 + lane] = st.wh0[off];
    }
  // one-hot input of layer 0: the term a residue code adds to a gate's sum is one weight
  for (int i = tid; i < 3 * 22 * 16; i += 256) {
    const int g = i / (22 * 16), code = (i / 16) % 22, row = i & 15;
    tab[(g * 24 + code) * 16 + row] = st.wx0[(g * 22 + code) * 512 + j0 + row];
  }
  // finishing thread: layer fl, rows j0 + 4 fg .. +3, column fc of the tile
  const int fl = __builtin_amdgcn_readfirstlane(tid >> 7), fg = (tid >> 5) & 3, fc = tid & 31;   // fl: uniform in a wave
  const int flane = 16 * fg + (fc & 15), fnt = fc >> 4;
  const int j4 = j0 + 4 * fg;
  const float4 bR = *reinterpret_cast<const float4*>(st.bias[fl] + j4);
  const float4 bZ = *reinterpret_cast<const float4*>(st.bias[fl] + 512 + j4);
  const float4 bI = *reinterpret_cast<const float4*>(st.bias[fl] + 1024 + j4);
  const float4 bH = *reinterpret_cast<const float4*>(st.bias[fl] + 1536 + j4);
  const int nmem = rec->nmem;
  if (c_hi - c_lo > VP_MAX_XCD_TILES) {                  // (the host never builds such a group: 8 x 2048 columns = 64 tiles per XCD)
    if (tid == 0) atomicOr(fault, DMP_FAULT_VGRU_HANDOFF);
    return;
  }
  for (int i = tid; i < c_hi - c_lo; i += 256) {
    int mi = 0;
    for (int m = 1; m < VG_MAX_MEMBERS; ++m)
      if (m < nmem && c_lo + i >= rec->mem[m].tile0) mi = m;
    sh_tile[i][0] = rec->mem[mi].N;
    sh_tile[i][1] = rec->mem[mi].L;
    sh_tile[i][2] = (c_lo + i - rec->mem[mi].tile0) * VG_TB;
    sh_tile[i][3] = mi;
  }
  if (tid < VG_MAX_MEMBERS) sh_msa[tid] = tid < nmem ? (unsigned long long)rec->mem[tid].msa : 0ull;
  __syncthreads();
  auto next_active = [&](int ct, int t) {                // first tile >= ct of this XCD that is computed at row t, or c_hi
    for (; ct < c_hi; ++ct) {
      const int N = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][0]);
      if (t < N || (t >= 1 && t <= N)) break;            // layer 0: t < N; layer 1, one row behind: 1 <= t <= N
    }
    return ct;
  };
  typedef unsigned vf_u32x4v __attribute__((vector_size(16)));
  constexpr int VF_SC1 = 16;                             // cache-policy bit of the buffer loads: past the L1, served by the L2
  // super-step ss of a tile's state (both layers, both column halves) -> register slot
  auto load_ss = [&](__amdgpu_buffer_rsrc_t r0, __amdgpu_buffer_rsrc_t r1, int tcol, int ss, vp_f32x4 (&d0)[2], vp_f32x4 (&d1)[2]) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const unsigned off = (unsigned)(((32 * w + 4 * ss + lq) * Lb + tcol + nt * 16 + lr) * 16);
      const vf_u32x4v x0 = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, VF_SC1);
      const vf_u32x4v x1 = __builtin_amdgcn_raw_buffer_load_b128(r1, off, 0, VF_SC1);
      d0[nt] = __builtin_bit_cast(vp_f32x4, x0);
      d1[nt] = __builtin_bit_cast(vp_f32x4, x1);
    }
  };
  const unsigned state_bytes = (unsigned)(128 * Lb * 16);          // [j/4 128][Lb] x float4

  float4 f[3];                                           // layer 0's weight fragments of the super-step ahead (from LDS)
#pragma unroll
  for (int g = 0; g < 3; ++g) f[g] = wl0[((w * VF_SS + 0) * 3 + g) * 64 + lane];
  for (int t = t_lo; t < t_hi; ++t) {
    const int par = t & 1;
    // layer 0 state at row t (both layers read it), layer 1 state at row t-1, this thread's layer's previous state
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(st.hT[0][par], 0, state_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(st.hT[1][par ^ 1], 0, state_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = fl ? r1 : r0;
    float* hnext = fl ? st.hT[1][par] : st.hT[0][par ^ 1];
    int ct = next_active(c_lo, t);
    // the state streams through four register slots three super-steps ahead of the MFMAs that read it (slot s & 3 holds
    // super-step s of whichever tile needs it next); a row's first tile cannot be requested before the row barrier
    vp_f32x4 b0[4][2], b1[4][2];
    if (ct < c_hi) {
#pragma unroll
      for (int s = 0; s < 3; ++s) load_ss(r0, r1, ct * VG_TB, s, b0[s], b1[s]);
    }
    while (ct < c_hi) {
      const int nx = next_active(ct + 1, t);
      const int ncol = (nx < c_hi ? nx : ct) * VG_TB;
      const int N = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][0]);
      const int L = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][1]);
      const bool act0 = t < N, act1 = t >= 1 && t <= N;
      const int tcol = ct * VG_TB;
      // what the finishing threads need of this tile - their previous state, layer 0's residue code - is requested now
      const int b = tcol + fc;
      const int64_t hoff = ((int64_t)(j4 >> 2) * Lb + b) * 4;
      // (four one-word loads: with ONE 16-byte load the compiler's packed-f32 gate arithmetic took element 0 for every
      // row in the f16 form of this kernel - v_pk_add_f32 ... op_sel_hi:[0,1]; vgru.hip)
      float hp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        hp[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, (unsigned)(hoff * 4 + 4 * i), 0, VF_SC1));
      const int bm = sh_tile[ct - c_lo][2] + fc;                               // the column in ITS alignment
      const uint8_t* msa = reinterpret_cast<const uint8_t*>(sh_msa[sh_tile[ct - c_lo][3]]);
      const int code = (fl == 0 && act0 && bm < L) ? (int)msa[(int64_t)t * L + bm] : 0;
      vp_f32x4 a0[3][2], a1[4][2];                       // layer 0: r z hn; layer 1: r z hn in; x column half
#pragma unroll
      for (int g = 0; g < 3; ++g) { a0[g][0] = vp_f32x4{0, 0, 0, 0}; a0[g][1] = vp_f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int g = 0; g < 4; ++g) { a1[g][0] = vp_f32x4{0, 0, 0, 0}; a1[g][1] = vp_f32x4{0, 0, 0, 0}; }
      vf_accumulators_ready(a1);
#pragma unroll
      for (int s = 0; s < VF_SS; ++s) {
        // (unconditional - the row's last tile requests its own state again: behind a branch the compiler cannot count
        // the loads in flight and waits for all of them)
        load_ss(r0, r1, s + 3 < VF_SS ? tcol : ncol, (s + 3) & (VF_SS - 1), b0[(s + 3) & 3], b1[(s + 3) & 3]);
        __builtin_amdgcn_sched_barrier(0);               // the loads stay where they are issued
        if (act0) {
          // per k quad the six accumulators (gate x column half) take their MFMA one after the other: a dependent MFMA
          // is six instructions behind the one it waits for
          const float fr[3][4] = {{f[0].x, f[0].y, f[0].z, f[0].w}, {f[1].x, f[1].y, f[1].z, f[1].w}, {f[2].x, f[2].y, f[2].z, f[2].w}};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int g = 0; g < 3; ++g) a0[g][nt] = vf_mfma_v(fr[g][i], b0[s & 3][nt][i], a0[g][nt]);
        }
        // layer 0's weight fragments of the NEXT super-step (of the next tile after the last one: they do not depend on
        // the tile) leave the LDS under layer 1's MFMAs
#pragma unroll
        for (int g = 0; g < 3; ++g) f[g] = wl0[((w * VF_SS + ((s + 1) & (VF_SS - 1))) * 3 + g) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        if (act1) {
          // recurrent product (into r, z, hn), then the input product (into r, z, in)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int g = 0; g < 3; ++g) a1[g][nt] = vf_mfma_a(WB[(s * 3 + g) * 4 + i], b1[s & 3][nt][i], a1[g][nt]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int g = 0; g < 3; ++g) {
                const int qa = g == 2 ? 3 : g;           // the third gate's input and recurrent sums stay apart
                a1[qa][nt] = vf_mfma_a(WA[(s * 3 + g) * 4 + i], b0[s & 3][nt][i], a1[qa][nt]);
              }
        }
      }
      vf_mfma_results_ready(a1);
      // ---- partial sums of the four K quarters meet in LDS
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) red[(w * 14 + g * 2 + nt) * 64 + lane] = a0[g][nt];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) red[(w * 14 + 6 + g * 2 + nt) * 64 + lane] = a1[g][nt];
      __syncthreads();
      if (fl ? act1 : act0) {
        const int nq = fl ? 4 : 3, base = fl ? 6 : 0;
        vp_f32x4 sum[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q < nq) {
            const int a = base + q * 2 + fnt;
            sum[q] = ((red[(0 * 14 + a) * 64 + flane] + red[(1 * 14 + a) * 64 + flane]) + red[(2 * 14 + a) * 64 + flane]) +
                     red[(3 * 14 + a) * 64 + flane];
          }
        }
        if (fl == 0) {
          const float* tr = tab + (0 * 24 + code) * 16 + 4 * fg;
          const float* tz = tab + (1 * 24 + code) * 16 + 4 * fg;
          const float* tn = tab + (2 * 24 + code) * 16 + 4 * fg;
#pragma unroll
          for (int i = 0; i < 4; ++i) { sum[0][i] += tr[i]; sum[1][i] += tz[i]; sum[3][i] = tn[i]; }
        }
        const float br[4] = {bR.x, bR.y, bR.z, bR.w}, bz[4] = {bZ.x, bZ.y, bZ.z, bZ.w};
        const float bi[4] = {bI.x, bI.y, bI.z, bI.w}, bh[4] = {bH.x, bH.y, bH.z, bH.w};
        float hn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // network.py:224 -> ATen's GRU cell: r = sigmoid(i_r + h_r), z = sigmoid(i_z + h_z),
          // n = tanh(i_n + r * h_n), h' = (h - n) * z + n
          const float rg = vf_sigmoid(sum[0][i] + br[i]);
          const float zg = vf_sigmoid(sum[1][i] + bz[i]);
          const float ng = tanhf((sum[3][i] + bi[i]) + rg * (sum[2][i] + bh[i]));
          hn[i] = (hp[i] - ng) * zg + ng;
        }
        *reinterpret_cast<float4*>(hnext + hoff) = make_float4(hn[0], hn[1], hn[2], hn[3]);
      }
      __syncthreads();                                   // `red` is free for the next tile
      ct = nx;
    }
    if (!barrier) break;                                 // one row per launch: the kernel boundary is the barrier
    // ---- row boundary: every workgroup of this XCD has written its rows of the new state
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores have reached the L2
    __syncthreads();
    const unsigned epoch = (unsigned)(t - t_lo + 1);
    if (tid == 0) asm volatile("global_store_dword %0, %1, off" :: "v"(&sync->flag[xcc][u]), "v"(epoch) : "memory");
    if (w == 0) {
      const unsigned* fp = &sync->flag[xcc][lane & 31];
      bool ok = false;
      const unsigned bound = t == t_lo ? VP_BARRIER_SPINS_FIRST : VP_BARRIER_SPINS;      // vgru.h: the first barrier is the residency wait
      for (unsigned spins = 0; spins < bound && !ok; ++spins) {
        unsigned v;
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(fp) : "memory");
        ok = __builtin_amdgcn_ballot_w64(v < epoch) == 0ull;
      }
      if (!ok && lane == 0) { atomicOr(fault, DMP_FAULT_VGRU_HANDOFF); sh_abort = 1; }
    }
    __syncthreads();
    if (sh_abort) break;                                 // a workgroup is missing for good: leave together (vgru.hip)
  }
}

int vgru_f32_kernel_attrs(dmp_ctx* c) {
  (void)c;
  DMP_HIP(hipFuncSetAttribute((const void*)vgru_persist_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VP_LDS_BYTES));
  return DMP_OK;
}

// rows [t_lo, t_hi) of the group set up on `lead` (vgru_group_setup), float32
int vgru_f32_group_steps(dmp_ctx* lead, int t_lo, int t_hi, hipStream_t s) {
  const int nt = lead->vg_ntiles;
  if (t_hi > lead->vg_maxN + 1) t_hi = lead->vg_maxN + 1;
  if (t_lo < 0) t_lo = 0;
  if (t_lo >= t_hi) return DMP_OK;
  const Weights& W = lead->W;
  VStaticF32 st{};
  st.wh0 = reinterpret_cast<const float4*>(W.v_f32[0]);
  st.wx1 = reinterpret_cast<const float4*>(W.v_f32[1]);
  st.wh1 = reinterpret_cast<const float4*>(W.v_f32[2]);
  st.wx0 = W.v_wx0f;
  st.bias[0] = W.v_b0; st.bias[1] = W.v_b1;
  for (int l = 0; l < 2; ++l)
    for (int p = 0; p < 2; ++p) st.hT[l][p] = lead->hT[l][p];
  const VGroupRec* rec = reinterpret_cast<const VGroupRec*>(lead->vgru_run);
  VPSync* sync = reinterpret_cast<VPSync*>(lead->vgru_sync);
  // A device without 256 CUs in 8 XCDs (partitioned / CPX modes: vgru_persist_ok is false) gets the launch-per-row form
  // below: with the barrier off the kernel takes its (XCD, slice) from the block id and depends on no placement.
  if (lead->vgru_persist && lead->vgru_persist_ok) {
    // needs every one of its 256 workgroups resident (row barriers): ordered against the process's other persistent
    // launches and cluster kernels on this device (CoResident, common.h)
    CoResident guard(lead, s, true);
    if (guard.status()) return guard.status();
    DMP_HIP(hipMemsetAsync(sync, 0, sizeof(VPSync), s));
    hipLaunchKernelGGL(vgru_persist_f32_kernel, dim3(VP_GRID - (lead->vgru_debug_drop_wg ? 1 : 0)), dim3(256), VP_LDS_BYTES, s, st, rec,
                       sync, lead->seq_abort, t_lo, t_hi, nt, 1);
    DMP_LAUNCH_CHECK();
    return guard.done();
  }
  for (int t = t_lo; t < t_hi; ++t)
    hipLaunchKernelGGL(vgru_persist_f32_kernel, dim3(VP_GRID), dim3(256), VP_LDS_BYTES, s, st, rec, sync, lead->seq_abort,
                       t, t + 1, nt, 0);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
