// gru_vertical with FULL-WIDTH OPERANDS on the bf16 matrix cores (round 6; option "vgru_f32" = 2, part of "precision" 2):
// every float32 operand of the three recurrent / input products of network.py:189, 223-224 (nn.GRU(22, 512, 2)) - weights
// AND state - is split exactly into three bf16 pieces (3 x 8 = 24 significand bits), the six piece products above 2^-24
// are accumulated in float32 by v_mfma_f32_16x16x32_bf16, the gates run on the device library's expf / tanhf: the
// arithmetic of conv_bf16.h for the vertical GRU.  432 MFMAs of 16 cycles per tile and wave instead of the 576 of 32 cycles
// of the float32 form (vgru_f32.hip): 2.7 x less matrix-core time.
//
// Decomposition: that of vgru_persist_f32_kernel, unchanged (columns over the 8 XCDs, hidden units over the 32 CUs of an XCD,
// K over the 4 waves, XCD-local row barriers, partial sums of the K quarters meeting in LDS, 256 finishing threads, the
// state as float32 [k/4][column][4] - the same buffers, so the chain's output and the launch-per-row fallback are the
// float32 form's).  What changes is where the operands' pieces come from:
//   * WEIGHTS are split once per launch.  A CU's weights are 442 KB as three pieces (295 KB as float32) and do not fit:
//     per lane the three pieces of layer 1's input product and the two big pieces of its recurrent product fill 240
//     registers, layer 0's recurrent product keeps its two big pieces in the 96 KB of LDS the float32 form uses for the
//     same weights, and the smallest pieces of those two products - each read by two MFMAs per K step - go to a table in
//     device memory (24 KB per wave, L2-resident) and stream through two three-quad windows a K step ahead.
//   * The STATE is split in registers: a lane's B operand of a K step is one octet of float32 (two 16-byte loads), turned
//     into three pieces by 44 VALU instructions in the shadow of the previous unit's MFMAs (an MFMA leaves three issue
//     slots; pinned with sched_group_barrier), after which the octet's registers are requested again for the K step after.
//     4 bytes per state element from the L2 instead of the 6 of stored pieces - and the L2 is what bounds this kernel: all
//     32 CUs of an XCD read the whole state of a tile (131 KB per tile and CU).
// Registers are the scarce resource (512 per lane, one wave per SIMD).  The builds that held all layer-1 pieces (288
// registers) or double-buffered the state a K step ahead spilled 54 - 101 registers and waited for the reloads in the
// loop; LABNOTES "Round 6, vertical GRU on the bf16 cores" has the variants and their times.
// The slices of an XCD start a row at different tiles and walk their K quarter from different K steps, so that they do not
// ask the L2 for the same lines at the same time (- 9 % chain time).
// Summation order (K quarters in wave order, the slice's K-step order, piece products (w0 x2) (w1 x1) (w0 x1) (w2 x0)
// (w1 x0) (w0 x0)) does not depend on the grouping: a member's result is the same bits alone, in any group and as a rider;
// the launch-per-row fallback is the same kernel with the barrier off.
#include "vgru.h"

namespace dmp {

typedef __bf16 vx_bf16x8 __attribute__((ext_vector_type(8)));

struct VStaticX3 {
  const float4* wh0;        // layer-0 recurrent weights  [gate 3][k/4 = 128][512 j] x float4 (k % 4)  (the float32 pack)
  const float4* wx1;        // layer-1 input weights      (same layout)
  const float4* wh1;        // layer-1 recurrent weights
  const float* wx0;         // layer-0 input weights with the embedding folded in: [gate 3][code 22][512 j]
  const float* bias[2];     // [layer]: [4][512]: r (b_ir+b_hr), z (b_iz+b_hz), b_in, b_hn
  float* hT[2][2];          // [layer][parity] float32 state [128][Lb][4]
  uint4* wq;                // the weight pieces a wave streams instead of holding: [workgroup 256][wave 4][K step 4][6][lane 64]
};


// x == p[0] + p[1] + p[2] exactly (round-to-nearest pieces, conv_bf16.h split3_bf16), eight values at a time
__device__ __forceinline__ void vx_split8(const float (&w)[8], vx_bf16x8 (&p)[3]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { const __bf16 h = (__bf16)w[e]; p[0][e] = h; r[e] = w[e] - (float)h; }
#pragma unroll
  for (int e = 0; e < 8; ++e) { const __bf16 h = (__bf16)r[e]; p[1][e] = h; r[e] = r[e] - (float)h; }
#pragma unroll
  for (int e = 0; e < 8; ++e) p[2][e] = (__bf16)r[e];
}

__device__ __forceinline__ vp_f32x4 vx_mfma(vx_bf16x8 a, vx_bf16x8 b, vp_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float vx_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int VX_KS = 4;                                  // K steps (32 k) of a wave's K quarter
constexpr int VX_WQ = 6;                                  // streamed weight quads per K step: the smallest pieces of layer 1's
                                                          // recurrent product (3 gates) and of layer 0's (3 gates)
constexpr size_t VX_WQ_BYTES = (size_t)VP_GRID * 4 * VX_KS * VX_WQ * 64 * 16;
static_assert(4 * VX_KS * 3 * 2 * 64 == VP_WL0_SLOTS, "two pieces of layer 0's weights fill the LDS area of the f16 form's pieces");

constexpr int VX_SC1 = 16;                                // cache-policy bit of the buffer loads: past the L1, served by the L2

// One tile (32 columns) x this wave's K quarter = 16 units of work: K step S = 0..3 x (h1's column half 0, 1: layer 1's
// recurrent product, 18 MFMAs each; h0's column half 0, 1: layer 0's recurrent and layer 1's input product, 36 MFMAs each).
// A unit's B operand is one octet of float32 state per lane (k = 8 lq .. + 7 of the K step at column lr: two float4 loads),
// split into three bf16 pieces with VALU instructions in the shadow of the PREVIOUS unit's MFMAs (44 per octet; an MFMA
// leaves three issue slots), after which the octet's registers are requested again for the K step after (or the next tile).
// On entry Q[0] holds the pieces of unit 0 and the four octet slots the state of their next unit; the same on exit for the
// tile at column `ncol`.  ACT0 / ACT1: the layer is computed at this row (all but a member's first / last row).
struct VxAddr {
  __amdgpu_buffer_rsrc_t r0, r1, rq;       // h0(t), h1(t-1) as float32 [k/4 128][Lb] x float4; this wave's streamed weight pieces
  unsigned vlane, vq;                      // the lane's part of a state / weight-table address
  unsigned k4w, krot, Lb;                  // k / 4 of the wave's K quarter, the K step the slice starts a tile with, row pitch
};

// octet slot `u4` (0, 1: h1's column halves; 2, 3: h0's) <- K step S of the tile at column `col`
__device__ __forceinline__ void vx_request(vp_f32x4 (&F)[4][2], int u4, unsigned S, unsigned col, const VxAddr& A) {
  const unsigned base = ((A.k4w + 8 * ((S + A.krot) & (VX_KS - 1))) * A.Lb + col + (u4 & 1) * 16) * 16;
#pragma unroll
  for (int h = 0; h < 2; ++h)
    F[u4][h] = __builtin_bit_cast(vp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(u4 < 2 ? A.r1 : A.r0, A.vlane, base + h * A.Lb * 16, VX_SC1));
}

__device__ __forceinline__ void vx_split_octet(const vp_f32x4 (&f)[2], vx_bf16x8 (&q)[3]) {
  const float wv[8] = {f[0][0], f[0][1], f[0][2], f[0][3], f[1][0], f[1][1], f[1][2], f[1][3]};
  vx_split8(wv, q);
}

template <bool ACT0, bool ACT1>
__device__ __forceinline__ void vx_tile_products(vp_f32x4 (&a0)[3][2], vp_f32x4 (&a1)[4][2], const vx_bf16x8 (&PA)[VX_KS][3][3],
                                                 const vx_bf16x8 (&PB)[VX_KS][3][2], vx_bf16x8 (&WB)[3], vx_bf16x8 (&WC)[3],
                                                 const vx_bf16x8* __restrict__ wl0w, vp_f32x4 (&F)[4][2], vx_bf16x8 (&Q)[2][3],
                                                 const VxAddr& A, unsigned tcol, unsigned ncol) {
#pragma unroll
  for (int S = 0; S < VX_KS; ++S) {
    // layer 0's two big weight pieces of this K step leave the LDS under the recurrent product
    vx_bf16x8 T[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      T[g][0] = wl0w[((S * 3 + g) * 2 + 0) * 64];
      T[g][1] = wl0w[((S * 3 + g) * 2 + 1) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u4 = 0; u4 < 4; ++u4) {
      const int unit = 4 * S + u4, nt = u4 & 1, cur = unit & 1, nxt = cur ^ 1, n4 = (u4 + 1) & 3;
      vx_bf16x8 (&P)[3] = Q[cur];
      const int nm = u4 < 2 ? (ACT1 ? 18 : 0) : (ACT0 ? 18 : 0) + (ACT1 ? 18 : 0);
      // ---- this unit's MFMAs, piece products in the order (w0 x2) (w1 x1) (w0 x1) (w2 x0) (w1 x0) (w0 x0)
#pragma unroll
      for (int pb = 2; pb >= 0; --pb)
#pragma unroll
        for (int pa = 2 - pb; pa >= 0; --pa)
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            if (u4 < 2) {                                // h1: layer 1's recurrent product (into r, z, hn)
              if (ACT1) a1[g][nt] = vx_mfma(pa == 2 ? WB[g] : PB[S][g][pa == 2 ? 0 : pa], P[pb], a1[g][nt]);
            } else {                                     // h0: layer 0's recurrent product (into r, z, hn) and layer 1's input
                                                         // product (into r, z, in: the third gate's two sums stay apart)
              if (ACT0) a0[g][nt] = vx_mfma(pa == 2 ? WC[g] : T[g][pa == 2 ? 0 : pa], P[pb], a0[g][nt]);
              if (ACT1) a1[g == 2 ? 3 : g][nt] = vx_mfma(PA[S][g][pa], P[pb], a1[g == 2 ? 3 : g][nt]);
            }
          }
      // ---- under them: the pieces of the next unit's octet, whose registers are then requested for the unit four on.
      // (Unconditional - the row's last tile requests its own state again: behind a branch the compiler cannot count the
      // loads in flight and waits for all of them.)
      vx_split_octet(F[n4], Q[nxt]);
      vx_request(F, n4, (unit + 5) >> 2 & (VX_KS - 1), unit + 5 < 4 * VX_KS ? tcol : ncol, A);
      if (u4 == 1) {                                     // the streamed weight pieces' last readers were in this unit
#pragma unroll
        for (int g = 0; g < 3; ++g)
          WB[g] = __builtin_bit_cast(vx_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(A.rq, A.vq, (((S + 1) & (VX_KS - 1)) * VX_WQ + g) * 1024, 0));
      }
      if (u4 == 3) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
          WC[g] = __builtin_bit_cast(vx_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(A.rq, A.vq, (((S + 1) & (VX_KS - 1)) * VX_WQ + 3 + g) * 1024, 0));
      }
      // (three VALU slots per MFMA, pinned: left to itself the scheduler clusters the conversions, + 5 % chain time)
#pragma unroll
      for (int i = 0; i < 36; ++i)
        if (i < nm) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      __builtin_amdgcn_sched_barrier(0);                 // the order of the units is the prefetch schedule
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void vgru_persist_x3_kernel(VStaticX3 st, const VGroupRec* __restrict__ rec, VPSync* __restrict__ sync,
                            int* __restrict__ fault, int t_lo, int t_hi, int ntiles, int barrier) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vx_smem[];
  vx_bf16x8* wl0 = reinterpret_cast<vx_bf16x8*>(vx_smem);                                    // [wave][K step][gate][piece 0 1][lane]
  vp_f32x4* red = reinterpret_cast<vp_f32x4*>(vx_smem + VP_WL0_SLOTS * 16);
  float* tab = reinterpret_cast<float*>(vx_smem + (VP_WL0_SLOTS + VP_RED_SLOTS) * 16);
  __shared__ int sh_u, sh_abort;
  __shared__ __attribute__((aligned(16))) float sh_bias[2][4][16];          // [layer][r z in hn][row]
  __shared__ int sh_tile[VP_MAX_XCD_TILES][5];          // N, L, column in its alignment, member, tile - in THIS workgroup's order
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

  // ---- this wave's weights: K quarter w, rows j0 .. j0+15.  Element e of operand (S, g) of lane (row lr, k group lq) is
  // W[g 512 + j0 + lr][128 w + 32 S' + 8 lq + e], e = 0..7: the eight k of the lane's A operand in K step S, which covers
  // the k of S' = (S + u) mod 4: the slices walk their K quarter from different starting points (see the tile order below;
  // the order of a hidden unit's sum is its slice's, whatever the group).
  // Layer 1's input product (from h0) keeps its three pieces in registers, its recurrent product (from h1) the two big
  // ones; layer 0's recurrent product has the two big ones in LDS.  The smallest pieces of the latter two - read by two
  // MFMAs a K step each - go to this wave's table in device memory (L2) and stream through two three-quad windows.
  vx_bf16x8 PA[VX_KS][3][3], PB[VX_KS][3][2];           // [K step][gate][piece]
  uint4* wqw = st.wq + (size_t)(blockIdx.x * 4 + w) * VX_KS * VX_WQ * 64 + lane;
#pragma unroll
  for (int S = 0; S < VX_KS; ++S)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      float wa[8], wb[8], wc[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t off = (int64_t)(g * 128 + 32 * w + 8 * ((S + u) & (VX_KS - 1)) + 2 * lq + h) * 512 + j0 + lr;
        const float4 a = st.wx1[off], b = st.wh1[off], c = st.wh0[off];
        wa[4 * h + 0] = a.x; wa[4 * h + 1] = a.y; wa[4 * h + 2] = a.z; wa[4 * h + 3] = a.w;
        wb[4 * h + 0] = b.x; wb[4 * h + 1] = b.y; wb[4 * h + 2] = b.z; wb[4 * h + 3] = b.w;
        wc[4 * h + 0] = c.x; wc[4 * h + 1] = c.y; wc[4 * h + 2] = c.z; wc[4 * h + 3] = c.w;
      }
      vx_split8(wa, PA[S][g]);
      vx_bf16x8 pb[3], pc[3];
      vx_split8(wb, pb);
      vx_split8(wc, pc);
      PB[S][g][0] = pb[0];
      PB[S][g][1] = pb[1];
      wl0[(((w * VX_KS + S) * 3 + g) * 2 + 0) * 64 + lane] = pc[0];
      wl0[(((w * VX_KS + S) * 3 + g) * 2 + 1) * 64 + lane] = pc[1];
      wqw[(S * VX_WQ + g) * 64] = __builtin_bit_cast(uint4, pb[2]);
      wqw[(S * VX_WQ + 3 + g) * 64] = __builtin_bit_cast(uint4, pc[2]);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the table is in the L2 before this wave reads it back
  const vx_bf16x8* wl0w = wl0 + w * VX_KS * 3 * 2 * 64 + lane;
  if (tid < 128) sh_bias[tid >> 6][(tid >> 4) & 3][tid & 15] = st.bias[tid >> 6][((tid >> 4) & 3) * 512 + j0 + (tid & 15)];
  // one-hot input of layer 0: the term a residue code adds to a gate's sum is one weight
  for (int i = tid; i < 3 * 22 * 16; i += 256) {
    const int g = i / (22 * 16), code = (i / 16) % 22, row = i & 15;
    tab[(g * 24 + code) * 16 + row] = st.wx0[(g * 22 + code) * 512 + j0 + row];
  }
  // finishing thread: layer fl, rows j0 + 4 fg .. +3, column fc of the tile
  const int fl = __builtin_amdgcn_readfirstlane(tid >> 7), fg = (tid >> 5) & 3, fc = tid & 31;   // fl: uniform in a wave
  const int flane = 16 * fg + (fc & 15), fnt = fc >> 4;
  const int j4 = j0 + 4 * fg;
  const int nmem = rec->nmem;
  if (c_hi - c_lo > VP_MAX_XCD_TILES) {                  // (the host never builds such a group)
    if (tid == 0) atomicOr(fault, DMP_FAULT_VGRU_HANDOFF);
    return;
  }
  // The 32 workgroups of an XCD read the same state: each starts the row at another tile (its slice number on), so that
  // they do not ask the L2 for the same lines at the same time.  The order of a row's tiles changes nothing in the sums.
  for (int i = tid; i < c_hi - c_lo; i += 256) {
    const int tile = c_lo + (i + u) % (c_hi - c_lo);
    int mi = 0;
    for (int m = 1; m < VG_MAX_MEMBERS; ++m)
      if (m < nmem && tile >= rec->mem[m].tile0) mi = m;
    sh_tile[i][0] = rec->mem[mi].N;
    sh_tile[i][1] = rec->mem[mi].L;
    sh_tile[i][2] = (tile - rec->mem[mi].tile0) * VG_TB;
    sh_tile[i][3] = mi;
    sh_tile[i][4] = tile;
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
  const unsigned state_bytes = (unsigned)(128 * Lb * 16);          // [j/4 128][Lb] x float4

  VxAddr A;
  A.k4w = (unsigned)(32 * w);                           // k / 4 of this wave's K quarter
  A.krot = (unsigned)(u & (VX_KS - 1));                 // ... and the K step this slice starts a tile with
  A.Lb = (unsigned)Lb;
  A.vlane = (unsigned)((2 * lq * Lb + lr) * 16);        // the lane's part of an octet's address: k group lq, column lr
  A.vq = (unsigned)(lane * 16);
  A.rq = __builtin_amdgcn_make_buffer_rsrc(st.wq + (size_t)(blockIdx.x * 4 + w) * VX_KS * VX_WQ * 64, 0, VX_KS * VX_WQ * 1024, 0x00020000);
  vx_bf16x8 WB[3], WC[3];                               // the streamed pieces of the K step ahead (they do not depend on the tile)
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    WB[g] = __builtin_bit_cast(vx_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(A.rq, A.vq, g * 1024, 0));
    WC[g] = __builtin_bit_cast(vx_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(A.rq, A.vq, (3 + g) * 1024, 0));
  }
  for (int t = t_lo; t < t_hi; ++t) {
    const int par = t & 1;
    // layer 0 state at row t (both layers read it), layer 1 state at row t-1, this thread's layer's previous state
    A.r0 = __builtin_amdgcn_make_buffer_rsrc(st.hT[0][par], 0, state_bytes, 0x00020000);
    A.r1 = __builtin_amdgcn_make_buffer_rsrc(st.hT[1][par ^ 1], 0, state_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh = fl ? A.r1 : A.r0;
    float* hnext = fl ? st.hT[1][par] : st.hT[0][par ^ 1];
    int ct = next_active(c_lo, t);
    // the state streams through four octet slots (vx_tile_products); a row's first tile cannot be requested before the
    // row barrier
    vp_f32x4 F[4][2];
    vx_bf16x8 Q[2][3];
    if (ct < c_hi) {
      const unsigned col0 = (unsigned)(__builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][4]) * VG_TB);
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) vx_request(F, u4, 0, col0, A);
      vx_split_octet(F[0], Q[0]);
      vx_request(F, 0, 1, col0, A);
    }
    while (ct < c_hi) {
      const int nx = next_active(ct + 1, t);
      const int ncol = __builtin_amdgcn_readfirstlane(sh_tile[(nx < c_hi ? nx : ct) - c_lo][4]) * VG_TB;
      const int N = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][0]);
      const int L = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][1]);
      const bool act0 = t < N, act1 = t >= 1 && t <= N;
      const int tcol = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][4]) * VG_TB;
      vp_f32x4 a0[3][2], a1[4][2];                       // layer 0: r z hn; layer 1: r z hn in; x column half
#pragma unroll
      for (int g = 0; g < 3; ++g) { a0[g][0] = vp_f32x4{0, 0, 0, 0}; a0[g][1] = vp_f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int g = 0; g < 4; ++g) { a1[g][0] = vp_f32x4{0, 0, 0, 0}; a1[g][1] = vp_f32x4{0, 0, 0, 0}; }
      if (act0 && act1)
        vx_tile_products<true, true>(a0, a1, PA, PB, WB, WC, wl0w, F, Q, A, (unsigned)tcol, (unsigned)ncol);
      else if (act0)                                     // a member's first row
        vx_tile_products<true, false>(a0, a1, PA, PB, WB, WC, wl0w, F, Q, A, (unsigned)tcol, (unsigned)ncol);
      else                                               // ... and the row after its last
        vx_tile_products<false, true>(a0, a1, PA, PB, WB, WC, wl0w, F, Q, A, (unsigned)tcol, (unsigned)ncol);
      // what the finishing threads need of this tile - their previous state, layer 0's residue code - is requested now:
      // it arrives under the reduction (requested before the products it would hold five registers through them)
      const int b = tcol + fc;
      const int64_t hoff = ((int64_t)(j4 >> 2) * Lb + b) * 4;
      // (four one-word loads: see vgru_f32.hip)
      float hp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        hp[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, (unsigned)(hoff * 4 + 4 * i), 0, VX_SC1));
      const int bm = sh_tile[ct - c_lo][2] + fc;                               // the column in ITS alignment
      const uint8_t* msa = reinterpret_cast<const uint8_t*>(sh_msa[sh_tile[ct - c_lo][3]]);
      const int code = (fl == 0 && act0 && bm < L) ? (int)msa[(int64_t)t * L + bm] : 0;
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
        const float* br = &sh_bias[fl][0][4 * fg];
        const float* bz = &sh_bias[fl][1][4 * fg];
        const float* bi = &sh_bias[fl][2][4 * fg];
        const float* bh = &sh_bias[fl][3][4 * fg];
        float hn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // network.py:224 -> ATen's GRU cell: r = sigmoid(i_r + h_r), z = sigmoid(i_z + h_z),
          // n = tanh(i_n + r * h_n), h' = (h - n) * z + n
          const float rg = vx_sigmoid(sum[0][i] + br[i]);
          const float zg = vx_sigmoid(sum[1][i] + bz[i]);
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

size_t vgru_x3_stream_bytes() { return VX_WQ_BYTES; }

int vgru_x3_kernel_attrs(dmp_ctx* c) {
  (void)c;
  DMP_HIP(hipFuncSetAttribute((const void*)vgru_persist_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VP_LDS_BYTES));
  return DMP_OK;
}

// rows [t_lo, t_hi) of the group set up on `lead` (vgru_group_setup), exact three-piece bf16 products
int vgru_x3_group_steps(dmp_ctx* lead, int t_lo, int t_hi, hipStream_t s) {
  const int nt = lead->vg_ntiles;
  if (t_hi > lead->vg_maxN + 1) t_hi = lead->vg_maxN + 1;
  if (t_lo < 0) t_lo = 0;
  if (t_lo >= t_hi) return DMP_OK;
  const Weights& W = lead->W;
  VStaticX3 st{};
  st.wh0 = reinterpret_cast<const float4*>(W.v_f32[0]);
  st.wx1 = reinterpret_cast<const float4*>(W.v_f32[1]);
  st.wh1 = reinterpret_cast<const float4*>(W.v_f32[2]);
  st.wx0 = W.v_wx0f;
  st.bias[0] = W.v_b0; st.bias[1] = W.v_b1;
  for (int l = 0; l < 2; ++l)
    for (int p = 0; p < 2; ++p) st.hT[l][p] = lead->hT[l][p];
  st.wq = reinterpret_cast<uint4*>(lead->vgru_wq);
  const VGroupRec* rec = reinterpret_cast<const VGroupRec*>(lead->vgru_run);
  VPSync* sync = reinterpret_cast<VPSync*>(lead->vgru_sync);
  // (a device without 256 CUs in 8 XCDs gets the launch-per-row form: with the barrier off the kernel takes its (XCD,
  // slice) from the block id and depends on no placement)
  if (lead->vgru_persist && lead->vgru_persist_ok) {
    CoResident guard(lead, s, true);
    if (guard.status()) return guard.status();
    DMP_HIP(hipMemsetAsync(sync, 0, sizeof(VPSync), s));
    hipLaunchKernelGGL(vgru_persist_x3_kernel, dim3(VP_GRID - (lead->vgru_debug_drop_wg ? 1 : 0)), dim3(256), VP_LDS_BYTES, s, st, rec,
                       sync, lead->seq_abort, t_lo, t_hi, nt, 1);
    DMP_LAUNCH_CHECK();
    return guard.done();
  }
  for (int t = t_lo; t < t_hi; ++t)
    hipLaunchKernelGGL(vgru_persist_x3_kernel, dim3(VP_GRID), dim3(256), VP_LDS_BYTES, s, st, rec, sync, lead->seq_abort,
                       t, t + 1, nt, 0);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
