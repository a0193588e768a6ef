// gru_vertical: the 2-layer GRU that runs DOWN the alignment (reference network.py:189, 223-224:
// time axis = N sequences, batch = L columns, hidden 512), for up to 8 alignments at once (a "group":
// the targets a throughput scheduler starts together, and the riders of its next round).
//
// Row t computes layer 0 at step t and layer 1 at step t-1 (both read the same h0 state) as GEMMs
// gates^T[j, b] = sum_k W[j, k] * x[k, b]  with the hidden index j on the MFMA M axis and the alignment
// column b on the N axis.
//
// Arithmetic: float32-grade products on the f16 matrix cores, the same scheme as conv_f16.h.  Every
// operand is the sum of two f16 pieces (weights: S*w = w0 + w1 with a power-of-two S per layer;
// state: 1024*h = h0 + h1, |h| < 1); the MFMA accumulates w0 h1 + w1 h0 + w0 h0 in float32 and the sums
// are scaled back by 1/(1024 S) (exact).  The dropped w1 h1 term is 2^-22 of a product.  The one-hot
// layer-0 input contributes 1024 (w0 + w1)[j, code].
//
// Layouts (one 16-byte load = one MFMA operand):
//   weights  Wq[piece 2][gate 3][k/8][512 j][8]  f16     A operand of (gate, piece) for 8 k
//   state    hH[piece 2][k/8 = 64][Lb][8]        f16     B operand;  written by the epilogue
//            hP[j/4 = 128][Lb][4]                f32     the state itself, for h' = (h - n) z + n
//
// Two forms, same interface (vgru_group_setup / vgru_group_steps / vgru_group_output):
//   * the PERSISTENT weight-stationary launch (round 4, the default on a 256-CU device): the whole chain in
//     one launch, columns partitioned over the XCDs, hidden units over the CUs of an XCD, weights resident
//     in registers and LDS, XCD-local row barriers - see vgru_persist_kernel;
//   * one launch PER ROW (round 3, option "vgru_persistent" = 0 and the fallback on other devices): the
//     group step kernel below, replayed from hipGraph chains of 128 (or 16) nodes.
#include "common.h"
#include "vgru.h"
#include <type_traits>
#include <map>
#include <mutex>

namespace dmp {

// =====================================================================================================
// Round 3: the GROUP step kernel.  One launch per alignment row serves the columns of several targets
// (contexts) at once, and inside a workgroup the weight fragments are fetched ONCE and shared by the
// waves through LDS.
//
//   unit of work   one wave = one (32 hidden x 32 column) tile of ONE product (layer 1: recurrent W_hh h1 or
//                  input W_ih h0; layer 0: recurrent W_hh h0 + the one-hot input), the whole K = 512 in one
//                  accumulator chain per gate: 32 k-steps x 9 MFMAs = 288 MFMAs, the same for every wave
//   workgroup      2 NW waves on one hidden tile: layer 1 = NW column tiles x {recurrent, input} (the two
//                  products of a column tile meet through LDS in the epilogue), layer 0 = 2 NW column tiles.
//                  The weight fragments of a product (6 KB per k-step: 2 pieces x 3 gates) are streamed into an
//                  LDS ring by LDS-DMA (global_load_lds_dwordx4), two k-steps per slot, three slots, and read by
//                  all waves of the product; each wave streams the state fragments of its own column tile
//                  into registers one slot ahead.  All VMEM traffic of the main loop is inline asm with
//                  counted waits (the compiler would otherwise drain the DMA queue at every use).
//   traffic        per workgroup and step (layer 1, NW = 4): weights 2 x 196 KB once + state 8 x 64 KB, for the
//                  work the round-2 kernel pulled 4 x (393 + 128) KB for; the column tiles of a launch may belong
//                  to different contexts (VTile), so four targets' GRUs cost 215 MB of L2 reads per step
//                  instead of 4 x 123 MB
//   arithmetic     identical whatever NW and whatever the grouping: a tile's result depends only on its own
//                  operands (accumulation order k = 0..511 for the recurrent product, then the exchange adds
//                  recurrent + input), so a target predicted alone and in a group gives the same bits.
// Block b runs on XCD b % 8; XCD x owns hidden tiles 2x, 2x+1 (1.2 MB of weight pieces stay in its L2).
// =====================================================================================================
#ifndef VG_CK_N
#define VG_CK_N 2       // measured: 4 k-steps per slot with 2 or 3 slots (half the barriers, 96-144 KB of LDS) gives the
#endif                  // same step time as 2 x 3 (72 KB): the loop is bound by what a CU's vector memory path delivers
#ifndef VG_R_N
#define VG_R_N 3
#endif
constexpr int VG_CK = VG_CK_N;                            // k-steps (16 k each) per ring slot (2 or 4)
constexpr int VG_R = VG_R_N;                              // ring slots per product: the weight DMA runs VG_R - 1 slots ahead
static_assert((VG_CK == 2 || VG_CK == 4) && VG_R >= 2 && VG_R <= 3, "unsupported ring shape");
constexpr int VG_FRAGS = 6;                               // weight fragments per k-step: piece x gate
constexpr int VG_NC = 32 / VG_CK;                         // slots per K = 512 product
constexpr int VG_RING_SLOTS = VG_R * VG_CK * VG_FRAGS * 64;   // 16-byte slots of one product's ring (36 KB)
constexpr int VG2_LDS_BYTES = 2 * VG_RING_SLOTS * 16;     // 73 728: two products

__global__ void vgru_set_group_kernel(VGroupRec* rec, VGroupRec v) { *rec = v; }
__global__ void vgru_set_run2_kernel(VGroupRec* rec, int t0, int t_end) { rec->t0 = t0; rec->t_end = t_end; }


// Gate functions on the hardware exponential and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): sigma(x) =
// 1 / (1 + 2^(-x log2 e)), tanh(x) = 1 - 2 sigma(-2x).  Absolute error below 2e-7 over the whole range (saturating
// correctly at +-1 / 0 / 1), a tenth of the float32 rounding of the 1024-term sums they are fed with; the device
// library's expf / tanhf cost 4.8 us per step (8 evaluations per lane, dependent chains) - measured, more than
// the matrix products.
#ifndef VG_ACC_GATES
#define VG_ACC_GATES 0
#endif
#if VG_ACC_GATES
__device__ __forceinline__ float vg_sigmoid(float x) { return gate_sigmoid(x); }
__device__ __forceinline__ float vg_tanh(float x) { return gate_tanh(x); }
#else
__device__ __forceinline__ float vg_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.442695040888963f));
}
__device__ __forceinline__ float vg_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.885390081777927f));
}
#endif

__device__ __forceinline__ void vg_dma16(const void* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ vg_u32x4 vg_gload16(const void* gsrc) {
  vg_u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(gsrc) : "memory");
  return v;
}
// wait until at most N VMEM operations of this wave are outstanding; the state fragments about to be used are
// threaded through the statement so that no MFMA that reads them can be scheduled above it
template <int N> __device__ __forceinline__ void vg_wait(vg_u32x4 (&b)[VG_CK][2]) {
#if VG_CK_N == 2
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]) : "n"(N) : "memory");
#else
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]),
                                       "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]) : "n"(N) : "memory");
#endif
}
__device__ __forceinline__ f32x16 vg_mfma2(uint4 a, vg_u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vg_f16x8, a), __builtin_bit_cast(vg_f16x8, b),
                                                c, 0, 0, 0);
}

// cgs = column groups of the XCD map (1, 2 or 4): XCD x works on hidden group x / cgs (16 / (8 / cgs) = 2 cgs hidden
// tiles) and on the supertiles st with st % cgs == x % cgs.
__host__ __device__ inline int vgru2_grid(int ntiles, int nw, int cgs) {
  const int nst = (ntiles + nw - 1) / nw;
  return 8 * 3 * cgs * ((nst + cgs - 1) / cgs);
}

// grid: vgru2_grid(ntiles, NW)   block: 128 NW   dynamic LDS: VG2_LDS_BYTES
// Per hidden-tile pair and supertile (NW column tiles): two layer-1 workgroups (one per hidden tile; halves = the
// recurrent and the input product) and one layer-0 workgroup (halves = the two hidden tiles).  Every wave belongs
// to a half of NW waves that feeds and reads its own weight ring, so all waves run the same VMEM schedule.
// XCD map.  The L2 of an XCD keeps nothing across a kernel boundary (PMC: 24 MB of fabric reads per step for one
// target = every weight piece + every XCD's copy of the state), so a step costs what the XCDs pull over the
// fabric: weights x (column groups) + state x (hidden groups).  One target: cgs = 1 (9.4 + 10.5 MB); four
// targets: cgs = 2 (18.8 + 21 MB instead of 9.4 + 42).
template <int NW>
__global__ __launch_bounds__(128 * NW) void vgru2_step_kernel(VStatic st, const VGroupRec* __restrict__ rec,
                                                             int idx, int cgs, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vg2_smem[];
  constexpr int D = VG_CK * VG_FRAGS / NW;            // LDS-DMA instructions per slot and wave
  // Every address of the main loop comes from the kernel arguments: with a cold L2 (see above) a dependent global
  // read at the head of a step costs 1-2 us, so the weight and state streams start at once and the group record
  // (which row is this, is the tile still active, where are its residue codes) is consumed after the loop.  The
  // buffer parity of a row is the parity of the node index: chains start at even rows.
  const int nst = (ntiles + NW - 1) / NW, nsth = (nst + cgs - 1) / cgs;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nj = 2 * cgs, hg = xcd / cgs, cg = xcd % cgs;     // hidden tiles hg nj .. hg nj + nj - 1
  if (slot >= 3 * cgs * nsth) return;
  const int layer = slot < nj * nsth ? 1 : 0;
  const int s2 = slot - nj * nsth;
  const int st_i = (layer ? slot / nj : s2 / cgs) * cgs + cg;
  if (st_i >= nst) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int half = w / NW, wl = w % NW;               // the half's ring; index inside the half = column tile
  const int prod = layer ? half : 0;                  // layer 1: 0 = recurrent (h1), 1 = input (h0)
  const int j0 = (hg * nj + (layer ? slot % nj : 2 * (s2 % cgs) + half)) * 32;
  const int tix = st_i * NW + wl;                     // a wave past the last tile shadows tile 0 and stores nothing
  const int tcol = (tix < ntiles ? tix : 0) * VG_TB;  // first column of the tile in the group's state
  const int Lb = ntiles * VG_TB;                      // column pitch of the group's state
  const int par = idx & 1;    // layer 0 reads parity t, writes t+1; layer 1 (step t-1) reads t+1, writes t
  const uint4* wbase = layer ? (prod ? st.wx[1] : st.wh[1]) : st.wh[0];
  const uint4* wsrc = wbase + (int64_t)kk * 512 + j0 + li;                        // + (f*64 + 2s) * 512
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)vg2_smem;
  const unsigned ring_addr = lds_base + half * (VG_RING_SLOTS * 16);
  const uint4* ring_l = reinterpret_cast<const uint4*>(vg2_smem) + half * VG_RING_SLOTS + lane;

  auto dma_slot = [&](int c) {                        // this wave's share of slot c (weights of k-steps 2c, 2c+1)
    const unsigned dst = ring_addr + (unsigned)((c % VG_R) * VG_CK * VG_FRAGS) * 1024u;
#pragma unroll
    for (int n = 0; n < D; ++n) {
      const int i = wl + n * NW, ks = i / VG_FRAGS, f = i - ks * VG_FRAGS;
      vg_dma16(wsrc + (int64_t)(f * 64 + 2 * (c * VG_CK + ks)) * 512, dst + (unsigned)i * 1024u);
    }
  };
  constexpr int DIST = VG_R - 1;                      // the weight DMA runs DIST slots ahead of the MFMAs
#pragma unroll
  for (int i = 0; i < DIST; ++i) dma_slot(i);
  const uint16_t* hsrc = layer ? (prod ? st.hH[0][par] : st.hH[1][par ^ 1]) : st.hH[0][par];
  const uint4* xsrc = reinterpret_cast<const uint4*>(hsrc) + (int64_t)kk * Lb + tcol + li;   // + (p*64 + 2s) * Lb
  const float* hprev = layer ? st.hT[1][par ^ 1] : st.hT[0][par];
  // the group record: requested now (scalar loads), first used after the main loop
  const int t = rec->t0 + idx;
  const int t_end = rec->t_end, nmem = rec->nmem;
  int mi = 0;
  for (int m = 1; m < VG_MAX_MEMBERS; ++m)
    if (m < nmem && tix >= rec->mem[m].tile0) mi = m;
  const VMember M = rec->mem[mi];
  const int N = M.N, L = M.L, b0 = (tix - M.tile0) * VG_TB;          // b0: the tile's first column in ITS alignment
  // What the epilogue will read - biases, the tile's previous state - is touched now (one word
  // per 128-byte line, results discarded) so that it is in the L2 by then.  The touches land in ONE register that
  // stays allocated until the first counted wait has retired them (a dead asm output would be handed to another
  // value and overwritten when the data arrives).
  unsigned pf = 0;
  {
    const unsigned char* q;
    if (lane < 4) q = reinterpret_cast<const unsigned char*>(st.bias[layer] + lane * 512 + j0);
    else q = reinterpret_cast<const unsigned char*>(hprev + ((int64_t)((j0 >> 2) + (((lane - 4) >> 2) & 7)) * Lb + tcol) * 4) + ((lane - 4) & 3) * 128;
    asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(q) : "memory");
    if (layer == 0) {
      // layer 0's one-hot input weights (24 rows of 512 bytes: piece, gate, k octet): 96 lines, read after the loop
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int line = (n * 64 + lane) % 96;
        q = reinterpret_cast<const unsigned char*>(st.wx[0] + (int64_t)(line >> 2) * 512 + j0) + (line & 3) * 128;
        asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(q) : "memory");
      }
    }
  }
  // The state fragments are asm loads the compiler does not count: every one of them is unconditional and is
  // waited for by exactly one statement that names its registers, so that no copy of a register can be placed
  // between a load and its wait (a conditional load or a wait chosen by a branch makes the compiler merge
  // registers with v_mov BEFORE the data has landed - seen, as NaNs in every second column tile).
  auto load_b = [&](vg_u32x4 (&b)[VG_CK][2], int c) {
#pragma unroll
    for (int ks = 0; ks < VG_CK; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p) b[ks][p] = vg_gload16(xsrc + (int64_t)(p * 64 + 2 * (c * VG_CK + ks)) * Lb);
  };

  f32x16 acc_r, acc_z, acc_t, acc_in;     // r, z, third gate (recurrent: W_hn h; input: W_in x); layer 0: + W_in x
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_z[r] = 0.f; acc_t[r] = 0.f; acc_in[r] = 0.f; }

  // The weight fragments of k-step ks+1 are read from the ring while the nine MFMAs of k-step ks run: left to itself
  // the compiler sinks every ds_read to its first use (read, wait, one MFMA, read, wait, ... - the LDS latency shows
  // four times per k-step); the scheduling barrier behind the reads keeps them a k-step ahead.  
  auto compute = [&](const vg_u32x4 (&b)[VG_CK][2], int c) {
    const uint4* a_l = ring_l + (c % VG_R) * (VG_CK * VG_FRAGS * 64);
    uint4 an[VG_FRAGS];
#pragma unroll
    for (int f = 0; f < VG_FRAGS; ++f) an[f] = a_l[f * 64];
#pragma unroll
    for (int ks = 0; ks < VG_CK; ++ks) {
      uint4 a[VG_FRAGS];                               // a[piece * 3 + gate]
#pragma unroll
      for (int f = 0; f < VG_FRAGS; ++f) a[f] = an[f];
      if (ks + 1 < VG_CK) {
#pragma unroll
        for (int f = 0; f < VG_FRAGS; ++f) an[f] = a_l[((ks + 1) * VG_FRAGS + f) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc_r = vg_mfma2(a[0], b[ks][1], acc_r);         // small products first: w0 h1, w1 h0, then w0 h0
      acc_z = vg_mfma2(a[1], b[ks][1], acc_z);
      acc_t = vg_mfma2(a[2], b[ks][1], acc_t);
      acc_r = vg_mfma2(a[3], b[ks][0], acc_r);
      acc_z = vg_mfma2(a[4], b[ks][0], acc_z);
      acc_t = vg_mfma2(a[5], b[ks][0], acc_t);
      acc_r = vg_mfma2(a[0], b[ks][0], acc_r);
      acc_z = vg_mfma2(a[1], b[ks][0], acc_z);
      acc_t = vg_mfma2(a[2], b[ks][0], acc_t);
    }
  };

  // VMEM order of a wave: DMA(0 .. DIST-1) touch B(0) | slot c: B(c+1) DMA(c+DIST).  At the top of slot c+1
  // everything up to B(c+1) must have landed; behind it only DMA(c+DIST) is outstanding, which may stay in flight
  // if it fills a later slot (DIST >= 2: D operations).  The first wait and the last two slots stand outside the
  // loop so that every wait is a straight-line statement.
  constexpr int WAITN = DIST >= 2 ? D : 0;
  vg_u32x4 bA[VG_CK][2], bB[VG_CK][2];
  load_b(bA, 0);
  vg_wait<0>(bA);
  asm volatile("" : "+v"(pf));                         // the touches have retired
  // the group record has arrived by now (requested before the first loads): layer 0 touches the line of residue
  // codes it reads after the loop; the touch retires at the next counted wait, `pf2` lives until after the loop
  unsigned pf2 = 0;
  if (layer == 0) {
    const int col = b0 + (li & 16) < L ? b0 + (li & 16) : L - 1;
    const unsigned char* q = M.msa + (int64_t)(t < N ? t : 0) * L + col;
    asm volatile("global_load_ubyte %0, %1, off" : "+v"(pf2) : "v"(q) : "memory");
  }
  constexpr int C_END = VG_NC - 2;
#pragma unroll 1
  for (int c = 0; c < C_END; c += 2) {
    __syncthreads();                                   // slot c is complete; every wave is done with slot c-1
    load_b(bB, c + 1);
    dma_slot(c + DIST);
    compute(bA, c);
    vg_wait<WAITN>(bB);
    __syncthreads();
    load_b(bA, c + 2);
    dma_slot(c + 1 + DIST);                            // c <= VG_NC - 4 and DIST <= 2: the slot exists
    compute(bB, c + 1);
    vg_wait<WAITN>(bA);
  }
  {
    constexpr int c = VG_NC - 2;
    __syncthreads();
    load_b(bB, c + 1);
    if (c + DIST < VG_NC) dma_slot(c + DIST);
    compute(bA, c);
    vg_wait<0>(bB);
    __syncthreads();
    compute(bB, c + 1);
  }

  asm volatile("" : "+v"(pf2));
  const bool active = tix < ntiles && t < t_end && (layer ? (t >= 1 && t <= N) : (t < N));
  if (layer == 0 && active) {
    // layer 0 input: one-hot of the residue code (value 1024 = the state scale), K = 32 (rows 22..31 of the
    // packed weights are 0); fragments straight from L2
    const uint4* wp = st.wx[0] + (j0 + li);
    const int b = b0 + li;
    const int code = (b < L) ? (int)M.msa[(int64_t)t * L + b] : 0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kq = 2 * s + kk;
      const int d = code - 8 * kq;                     // position of the hot element among this lane's 8 k
      const unsigned hot = (d >= 0 && d < 8) ? (0x6400u << (16 * (d & 1))) : 0u;
      vg_u32x4 x;
      x.x = (d >> 1) == 0 ? hot : 0u;
      x.y = (d >> 1) == 1 ? hot : 0u;
      x.z = (d >> 1) == 2 ? hot : 0u;
      x.w = (d >> 1) == 3 ? hot : 0u;
      acc_r = vg_mfma2(wp[(int64_t)((1 * 3 + 0) * 4 + kq) * 512], x, acc_r);
      acc_z = vg_mfma2(wp[(int64_t)((1 * 3 + 1) * 4 + kq) * 512], x, acc_z);
      acc_in = vg_mfma2(wp[(int64_t)((1 * 3 + 2) * 4 + kq) * 512], x, acc_in);
      acc_r = vg_mfma2(wp[(int64_t)((0 * 3 + 0) * 4 + kq) * 512], x, acc_r);
      acc_z = vg_mfma2(wp[(int64_t)((0 * 3 + 1) * 4 + kq) * 512], x, acc_z);
      acc_in = vg_mfma2(wp[(int64_t)((0 * 3 + 2) * 4 + kq) * 512], x, acc_in);
    }
  }

  // ---- epilogue.  Accumulator register r of lane (kk, li): hidden row j0 + 8 (r / 4) + 4 kk + (r % 4), column
  // b0 + li.  Layer 1: the two waves of a column tile swap halves through LDS - the recurrent wave finishes the
  // register groups g4 = 0, 1, the input wave g4 = 2, 3; sums are formed as recurrent + input.
  float* xb = reinterpret_cast<float*>(vg2_smem);      // [wl][dir 2][gate 3][reg 8][64]  (12 KB per column tile)
  int g_lo = 0, g_hi = 4;
  if (layer == 1) {
    __syncthreads();                                   // every wave is done with the rings
    float* out = xb + ((wl * 2 + prod) * 3) * 8 * 64 + lane;
    const int ro = prod ? 0 : 8;                       // the half the partner finishes
    if (active) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        out[(0 * 8 + r) * 64] = acc_r[ro + r];
        out[(1 * 8 + r) * 64] = acc_z[ro + r];
        out[(2 * 8 + r) * 64] = acc_t[ro + r];
      }
    }
    __syncthreads();
    g_lo = prod ? 2 : 0;
    g_hi = g_lo + 2;
  }
  if (!active) return;
  const float* in_l = xb + ((wl * 2 + (prod ^ 1)) * 3) * 8 * 64 + lane;
  const float* bias = st.bias[layer];
  const float inv = st.inv_scale[layer];
  float* hnext = layer ? st.hT[1][par] : st.hT[0][par ^ 1];
  uint16_t* gnext = layer ? st.hH[1][par] : st.hH[0][par ^ 1];
  const int b = tcol + li;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    if (g4 < g_lo || g4 >= g_hi) continue;
    const int j4 = j0 + 8 * g4 + 4 * kk;
    const int64_t hoff = ((int64_t)(j4 >> 2) * Lb + b) * 4;
    const float4 hp4 = *reinterpret_cast<const float4*>(hprev + hoff);
    const float4 bR = *reinterpret_cast<const float4*>(bias + j4);
    const float4 bZ = *reinterpret_cast<const float4*>(bias + 512 + j4);
    const float4 bI = *reinterpret_cast<const float4*>(bias + 1024 + j4);
    const float4 bH = *reinterpret_cast<const float4*>(bias + 1536 + j4);
    const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
    const float br[4] = {bR.x, bR.y, bR.z, bR.w}, bz[4] = {bZ.x, bZ.y, bZ.z, bZ.w};
    const float bi[4] = {bI.x, bI.y, bI.z, bI.w}, bh[4] = {bH.x, bH.y, bH.z, bH.w};
    float hn[4];
    unsigned short q0[4], q1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * g4 + q;
      float s_r, s_z, s_in, s_hn;
      if (layer == 0) {
        s_r = acc_r[r]; s_z = acc_z[r]; s_hn = acc_t[r]; s_in = acc_in[r];
      } else {
        const int rl = r & 7;                          // register inside the exchanged half
        const float o_r = in_l[(0 * 8 + rl) * 64], o_z = in_l[(1 * 8 + rl) * 64], o_t = in_l[(2 * 8 + rl) * 64];
        if (prod == 0) { s_r = acc_r[r] + o_r; s_z = acc_z[r] + o_z; s_hn = acc_t[r]; s_in = o_t; }
        else           { s_r = o_r + acc_r[r]; s_z = o_z + acc_z[r]; s_hn = o_t; s_in = acc_t[r]; }
      }
      const float rg = vg_sigmoid(s_r * inv + br[q]);
      const float zg = vg_sigmoid(s_z * inv + bz[q]);
      const float ng = vg_tanh((s_in * inv + bi[q]) + rg * (s_hn * inv + bh[q]));
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

// ---------------------------------------------------------------------------------------------------------------
// Persistent, weight-stationary form (round 4): the whole chain of a group in ONE launch.
//
// The per-row launches above restart from cold L2s 2001 times (24 MB of fabric reads per row for one target against
// 9.4 MB of weight pieces) and a row costs 16.8 (one target) .. 47 us (eight) against 2.2 .. 17.6 us of matrix-core time.
// Here the chain is partitioned by COLUMNS over the XCDs and by HIDDEN UNITS over the 32 CUs of an XCD:
//   * XCD x owns a contiguous range of the group's column tiles - the state of a column never leaves its XCD, so a row
//     boundary is an XCD-local barrier (plain stores that stop in the XCD's L2, sc1 loads that find them there:
//     common.h, cluster hand-off), and the eight XCDs never synchronise with each other at all;
//   * CU u of an XCD owns hidden units 16u .. 16u+15 of BOTH layers, all three gates, all three products (layer-0
//     recurrent, layer-1 input, layer-1 recurrent): its gate arithmetic is local, and its share of the weight pieces -
//     295 KB - never moves again: wave w keeps the K quarter [128w, 128w+128) of the two layer-1 products in 192
//     accumulation registers (the A operands of v_mfma_f32_16x16x32_f16 are read straight from AGPRs) and of the
//     layer-0 product in 24 KB of LDS.  9.4 MB of weights per XCD live in 32 x (registers + LDS); nothing is streamed;
//   * per row and column tile a wave multiplies its K quarter (216 MFMAs 16x16x32, float32-grade: w0 h1 + w1 h0 + w0 h0
//     as above) against the tile's state pieces read from the L2 (32 KB per wave and tile), the four waves' partial
//     sums meet in LDS, 256 threads finish 16 hidden units x 32 columns x 2 layers.
// One workgroup per CU (160 KB of LDS, 1 wave per SIMD with up to 512 registers): the launch owns the machine while it
// runs - which is what the scheduler wants of a chain (a front-end phase has no convolutions to run beside it).
// Summation order: K quarters in wave order, k ascending inside a quarter - independent of what else is in the group,
// so a member's result is the same bits whatever the grouping; against the per-row kernel (K in 32 ascending k-steps
// of one accumulator) the results agree to float32 rounding (tests: 1e-5 against the oracle's nn.GRU either way).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ vp_f32x4 vp_mfma_a(vg_u32x4 a, vg_u32x4 b, vp_f32x4 c) {        // A from an AGPR quad
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "a"(a), "v"(b));
  return c;
}
__device__ __forceinline__ vp_f32x4 vp_mfma_v(uint4 a, vg_u32x4 b, vp_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vg_f16x8, a), __builtin_bit_cast(vg_f16x8, b), c, 0, 0, 0);
}
// The MFMAs of vp_mfma_a are inline assembly: the compiler's hazard recogniser does not see them, and the hardware
// does not interlock a matrix-core result against a non-MFMA reader (the compiler puts `s_nop 7` between a
// v_mfma_f32_16x16x32_f16 and a ds_write of its result).  Dependent MFMAs on the same accumulator need no wait states;
// before anything else reads the eight layer-1 accumulators, this pads 16.
__device__ __forceinline__ void vp_mfma_results_ready(vp_f32x4 (&a)[4][2]) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]),
                                       "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]));
}

// ... and the other direction (round 5, found in the float32 form of this kernel, vgru_f32.hip): the compiler sinks the
// zero-initialisation of an accumulator (v_mov_b64) to just in front of its first assembly MFMA, which it does not know
// to be a reader with a hazard - there, with nothing in between, the MFMA accumulated onto stale registers.  This
// kernel's schedule happened to leave one instruction in between (and every test passed); the accumulators are now
// pinned behind a statement that takes all of them and pads four wait states.  Same arithmetic, same bits.
__device__ __forceinline__ void vp_accumulators_ready(vp_f32x4 (&a)[4][2]) {
  asm volatile("s_nop 3" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]),
                           "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]));
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void vgru_persist_kernel(VStatic st, const VGroupRec* __restrict__ rec, VPSync* __restrict__ sync, int* __restrict__ fault,
                         int t_lo, int t_hi, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char vp_smem[];
  uint4* wl0 = reinterpret_cast<uint4*>(vp_smem);
  vp_f32x4* red = reinterpret_cast<vp_f32x4*>(vp_smem + VP_WL0_SLOTS * 16);
  float* tab = reinterpret_cast<float*>(vp_smem + (VP_WL0_SLOTS + VP_RED_SLOTS) * 16);
  __shared__ int sh_u, sh_abort;
  // this XCD's column tiles: {rows N, columns L, first column of the tile in ITS alignment, member} and the members'
  // alignments - from the group record in global memory once per launch (a tile's bookkeeping was a chain of four to
  // six dependent scalar loads at the head of every tile before)
  __shared__ int sh_tile[VP_MAX_XCD_TILES][4];
  __shared__ unsigned long long sh_msa[VG_MAX_MEMBERS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  if (tid == 0) {
    sh_u = (int)__hip_atomic_fetch_add(&sync->count[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh_abort = 0;
  }
  __syncthreads();
  const int u = sh_u;                                  // this workgroup's hidden-unit slice on its XCD
  if (u >= 32) {                                       // more than 32 workgroups landed on this XCD: another one is short
    if (tid == 0) atomicOr(fault, DMP_FAULT_VGRU_HANDOFF);
    return;
  }
  const int j0 = 16 * u, lr = lane & 15, lq = lane >> 4;
  const int Lb = ntiles * VG_TB;
  const int c_lo = (int)(((long long)ntiles * xcc) / 8), c_hi = (int)(((long long)ntiles * (xcc + 1)) / 8);

  // ---- this wave's weights: K quarter w, rows j0 .. j0+15; fragment f = (k-step, piece, gate)
  vg_u32x4 WA[24], WB[24];                             // layer-1 input (from h0) and recurrent (from h1) products
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int pg = 0; pg < 6; ++pg) {
      const int64_t off = (int64_t)(pg * 64 + 16 * w + 4 * ks + lq) * 512 + j0 + lr;
      const uint4 a = st.wx[1][off], b = st.wh[1][off], c0 = st.wh[0][off];
      WA[ks * 6 + pg] = vg_u32x4{a.x, a.y, a.z, a.w};
      WB[ks * 6 + pg] = vg_u32x4{b.x, b.y, b.z, b.w};
      wl0[((w * 4 + ks) * 6 + pg) * 64 + lane] = c0;
    }
  // one-hot input of layer 0: the term a residue code adds to a gate's sum = 1024 (w0 + w1)[row, code], as the
  // product of the f16 pieces with the state scale
  for (int i = tid; i < 3 * 22 * 16; i += 256) {
    const int g = i / (22 * 16), code = (i / 16) % 22, row = i & 15;
    const uint16_t* wp = reinterpret_cast<const uint16_t*>(st.wx[0]);
    const int64_t e0 = ((int64_t)((0 * 3 + g) * 4 + code / 8) * 512 + j0 + row) * 8 + code % 8;
    const int64_t e1 = ((int64_t)((1 * 3 + g) * 4 + code / 8) * 512 + j0 + row) * 8 + code % 8;
    tab[(g * 24 + code) * 16 + row] = VGRU_STATE_SCALE * ((float)__builtin_bit_cast(_Float16, wp[e0]) + (float)__builtin_bit_cast(_Float16, wp[e1]));
  }
  // finishing thread: layer fl, rows j0 + 4 fg .. +3, column fc of the tile
  const int fl = __builtin_amdgcn_readfirstlane(tid >> 7), fg = (tid >> 5) & 3, fc = tid & 31;   // fl: uniform in a wave
  const int flane = 16 * fg + (fc & 15), fnt = fc >> 4;
  const int j4 = j0 + 4 * fg;
  const float4 bR = *reinterpret_cast<const float4*>(st.bias[fl] + j4);
  const float4 bZ = *reinterpret_cast<const float4*>(st.bias[fl] + 512 + j4);
  const float4 bI = *reinterpret_cast<const float4*>(st.bias[fl] + 1024 + j4);
  const float4 bH = *reinterpret_cast<const float4*>(st.bias[fl] + 1536 + j4);
  const float inv = st.inv_scale[fl];
  const int nmem = rec->nmem;
  __syncthreads();

  // member of a column tile / is the tile computed at row t (layer 0: t < N; layer 1, one row behind: 1 <= t <= N)
  auto member_of = [&](int ct) {
    int mi = 0;
    for (int m = 1; m < VG_MAX_MEMBERS; ++m)
      if (m < nmem && ct >= rec->mem[m].tile0) mi = m;
    return mi;
  };
  if (c_hi - c_lo > VP_MAX_XCD_TILES) {                 // (the host never builds such a group: 8 x 2048 columns = 64 tiles per XCD)
    if (tid == 0) atomicOr(fault, DMP_FAULT_VGRU_HANDOFF);
    return;
  }
  for (int i = tid; i < c_hi - c_lo; i += 256) {
    const int mi = member_of(c_lo + i);
    sh_tile[i][0] = rec->mem[mi].N;
    sh_tile[i][1] = rec->mem[mi].L;
    sh_tile[i][2] = (c_lo + i - rec->mem[mi].tile0) * VG_TB;
    sh_tile[i][3] = mi;
  }
  if (tid < VG_MAX_MEMBERS) sh_msa[tid] = tid < nmem ? (unsigned long long)rec->mem[tid].msa : 0ull;
  __syncthreads();
  auto next_active = [&](int ct, int t) {               // first tile >= ct of this XCD that is computed at row t, or c_hi
    for (; ct < c_hi; ++ct) {
      const int N = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][0]);
      if (t < N || (t >= 1 && t <= N)) break;
    }
    return ct;
  };
  // State pieces of this wave's K quarter for one column tile: [k-step][piece][column half], as buffer loads the
  // compiler tracks itself (raw_buffer_load with the sc1 bit: past the L1, served by the XCD's L2).  A first version
  // used inline-assembly loads with hand-counted waits, as the launch-per-row kernel does; with registers carried
  // around this loop's back edge that produced wrong values for every tile after an XCD's first (run-to-run
  // different, invisible to a static in-flight-register check), so the waits are the compiler's here.
  typedef unsigned vp_u32x4v __attribute__((vector_size(16)));
  constexpr int VP_SC1 = 16;                            // cache-policy bit of the buffer load: agent scope
  // one k-step (a quarter of this wave's K quarter) of a tile's pieces -> register slot ks
  auto load_ks = [&](__amdgpu_buffer_rsrc_t r0, __amdgpu_buffer_rsrc_t r1, int tcol, int ks, vg_u32x4 (&d0)[4][2][2],
                     vg_u32x4 (&d1)[4][2][2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const unsigned off = (unsigned)(((p * 64 + 16 * w + 4 * ks + lq) * Lb + tcol + nt * 16 + lr) * 16);
        const vp_u32x4v x0 = __builtin_amdgcn_raw_buffer_load_b128(r0, off, 0, VP_SC1);
        const vp_u32x4v x1 = __builtin_amdgcn_raw_buffer_load_b128(r1, off, 0, VP_SC1);
        d0[ks][p][nt] = vg_u32x4{x0[0], x0[1], x0[2], x0[3]};
        d1[ks][p][nt] = vg_u32x4{x1[0], x1[1], x1[2], x1[3]};
      }
  };
  const unsigned piece_bytes = (unsigned)(2 * 64 * Lb * 16);     // one layer / parity: [piece 2][k/8 64][Lb] x 16 bytes
  const unsigned state_bytes = (unsigned)(128 * Lb * 16);        // [j/4 128][Lb] x float4

  uint4 f[6];                                          // layer 0's weight fragments of the k-step ahead (from LDS)
#pragma unroll
  for (int pg = 0; pg < 6; ++pg) f[pg] = wl0[((w * 4 + 0) * 6 + pg) * 64 + lane];
  for (int t = t_lo; t < t_hi; ++t) {
    const int par = t & 1;
    // layer 0 state at row t (both layers read it), layer 1 state at row t-1, this thread's layer's previous state
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(st.hH[0][par], 0, piece_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(st.hH[1][par ^ 1], 0, piece_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rh =
        __builtin_amdgcn_make_buffer_rsrc(fl ? st.hT[1][par ^ 1] : st.hT[0][par], 0, state_bytes, 0x00020000);
    float* hnext = fl ? st.hT[1][par] : st.hT[0][par ^ 1];
    uint16_t* gnext = fl ? st.hH[1][par] : st.hH[0][par ^ 1];
    int ct = next_active(c_lo, t);
    vg_u32x4 b0[4][2][2], b1[4][2][2];
    // The pieces stream through the four register slots THREE k-steps ahead of the MFMAs that read them - slot ks holds
    // k-step ks of whichever tile is next to need it: k-step 0 issues the loads of this tile's k-step 3, k-steps 1..3 the
    // loads of the NEXT tile's k-steps 0..2, each into the slot the k-step before has just read.  (Until round 4's last
    // session a tile's 32 loads were issued together behind its last MFMA and waited for under the reduction: every CU
    // of an XCD reads the tile's whole state, 4 MB per tile out of one L2 - 0.8 of a tile's 4.2 us.)  A row's first
    // tile cannot be requested before the row barrier.
    if (ct < c_hi) {
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) load_ks(r0, r1, ct * VG_TB, ks, b0, b1);
    }
    while (ct < c_hi) {
      const int nx = next_active(ct + 1, t);
      const int ncol = (nx < c_hi ? nx : ct) * VG_TB;
      const int N = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][0]);
      const int L = __builtin_amdgcn_readfirstlane(sh_tile[ct - c_lo][1]);
      const bool act0 = t < N, act1 = t >= 1 && t <= N;
      const int tcol = ct * VG_TB;
      // what the finishing threads need of this tile - their previous state, layer 0's residue code - is requested
      // now and lands under the MFMAs
      const int b = tcol + fc;
      const int64_t hoff = ((int64_t)(j4 >> 2) * Lb + b) * 4;
      // (four one-word loads: with the state's float4 as ONE 16-byte load the compiler's packed-f32 gate arithmetic took
      // element 0 for every row - v_pk_add_f32 ... op_sel_hi:[0,1]; found by substitution on the GPU, twice)
      float hp[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        hp[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, (unsigned)(hoff * 4 + 4 * i), 0, VP_SC1));
      const int bm = sh_tile[ct - c_lo][2] + fc;                              // the column in ITS alignment
      const uint8_t* msa = reinterpret_cast<const uint8_t*>(sh_msa[sh_tile[ct - c_lo][3]]);
      const int code = (fl == 0 && act0 && bm < L) ? (int)msa[(int64_t)t * L + bm] : 0;
      vp_f32x4 a0[3][2], a1[4][2];                     // layer 0: r z hn; layer 1: r z hn in; x column half
#pragma unroll
      for (int g = 0; g < 3; ++g) { a0[g][0] = vp_f32x4{0, 0, 0, 0}; a0[g][1] = vp_f32x4{0, 0, 0, 0}; }
#pragma unroll
      for (int g = 0; g < 4; ++g) { a1[g][0] = vp_f32x4{0, 0, 0, 0}; a1[g][1] = vp_f32x4{0, 0, 0, 0}; }
      vp_accumulators_ready(a1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // (unconditional - the row's last tile requests its own pieces again: behind a branch the compiler cannot count
        // the loads in flight and waits for all of them)
        load_ks(r0, r1, ks == 0 ? tcol : ncol, (ks + 3) & 3, b0, b1);
        __builtin_amdgcn_sched_barrier(0);               // the loads stay where they are issued
        if (act0) {
          // small products first (w0 h1, w1 h0, then w0 h0); per product the six accumulators (gate x column half) take
          // their MFMA one after the other, so that a dependent MFMA is six instructions behind the one it waits for
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int g = 0; g < 3; ++g)
                a0[g][nt] = vp_mfma_v(f[pr == 1 ? 3 + g : g], b0[ks][pr == 0 ? 1 : 0][nt], a0[g][nt]);
        }
        // layer 0's weight fragments of the NEXT k-step (of the next tile after the last one: they do not depend on the
        // tile) leave the LDS under layer 1's MFMAs - requested at the head of a k-step they were waited for with the
        // matrix pipe idle, four waves sharing one LDS port
#pragma unroll
        for (int pg = 0; pg < 6; ++pg) f[pg] = wl0[((w * 4 + ((ks + 1) & 3)) * 6 + pg) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        if (act1) {
          // recurrent product (into r, z, hn), then the input product (into r, z, in): the same order of additions per
          // accumulator as one after the other, the MFMAs of a product interleaved over the accumulators
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int g = 0; g < 3; ++g)
                a1[g][nt] = vp_mfma_a(WB[ks * 6 + (pr == 1 ? 3 + g : g)], b1[ks][pr == 0 ? 1 : 0][nt], a1[g][nt]);
#pragma unroll
          for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int g = 0; g < 3; ++g) {
                const int qa = g == 2 ? 3 : g;         // the third gate's input and recurrent sums stay apart
                a1[qa][nt] = vp_mfma_a(WA[ks * 6 + (pr == 1 ? 3 + g : g)], b0[ks][pr == 0 ? 1 : 0][nt], a1[qa][nt]);
              }
        }
      }
      vp_mfma_results_ready(a1);
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
        unsigned short q0[4], q1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rg = vg_sigmoid(sum[0][i] * inv + br[i]);
          const float zg = vg_sigmoid(sum[1][i] * inv + bz[i]);
          const float ng = vg_tanh((sum[3][i] * inv + bi[i]) + rg * (sum[2][i] * inv + bh[i]));
          hn[i] = (hp[i] - ng) * zg + ng;
          const float hs = hn[i] * VGRU_STATE_SCALE;
          const _Float16 p0 = (_Float16)hs;
          const _Float16 p1 = (_Float16)(hs - (float)p0);
          q0[i] = __builtin_bit_cast(unsigned short, p0);
          q1[i] = __builtin_bit_cast(unsigned short, p1);
        }
        *reinterpret_cast<float4*>(hnext + hoff) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        const int64_t goff = ((int64_t)(j4 >> 3) * Lb + b) * 8 + (j4 & 7);
        *reinterpret_cast<uint2*>(gnext + goff) =
            make_uint2((unsigned)q0[0] | ((unsigned)q0[1] << 16), (unsigned)q0[2] | ((unsigned)q0[3] << 16));
        *reinterpret_cast<uint2*>(gnext + (int64_t)64 * Lb * 8 + goff) =
            make_uint2((unsigned)q1[0] | ((unsigned)q1[1] << 16), (unsigned)q1[2] | ((unsigned)q1[3] << 16));
      }
      __syncthreads();                                 // `red` is free for the next tile
      ct = nx;
    }
    // ---- row boundary: every workgroup of this XCD has written its rows of the new state
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have reached the L2
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
    // A workgroup of this XCD is missing for good (not resident, or one XCD got a 33rd workgroup and another is short):
    // the fault bit is raised and the outputs of this prediction become NaN - leave the row loop now, the whole
    // workgroup together, instead of waiting the full bound again at each of the up to 3001 rows that are left
    // (seconds per row = a GPU that looks hung for hours before dmp_sync_faults returns).
    if (sh_abort) break;
  }
}

// ---- group launcher ---------------------------------------------------------------------------------
static int vgru2_nw(int ntiles) { return ntiles >= 24 ? 4 : (ntiles >= 12 ? 2 : 1); }
static int vgru2_cgs(int ntiles) { return ntiles >= 24 ? 2 : 1; }

static int vgru2_graph(dmp_ctx* c, int ntiles, int nw, int cgs, int len, hipGraphExec_t* out) {
  const int grid = vgru2_grid(ntiles, nw, cgs);
  const int64_t key = ((int64_t)1 << 62) | ((int64_t)cgs << 52) | ((int64_t)nw << 48) | ((int64_t)ntiles << 16) | len;
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
  const VGroupRec* rec = reinterpret_cast<const VGroupRec*>(c->vgru_run);
  void* fn = nw == 4 ? (void*)vgru2_step_kernel<4> : (nw == 2 ? (void*)vgru2_step_kernel<2> : (void*)vgru2_step_kernel<1>);
  hipGraph_t g;
  DMP_HIP(hipGraphCreate(&g, 0));
  hipGraphNode_t prev = nullptr;
  for (int idx = 0; idx < len; ++idx) {
    int idx_arg = idx;
    int cgs_arg = cgs, nt_arg = ntiles;
    void* params[5] = {(void*)&st, (void*)&rec, (void*)&idx_arg, (void*)&cgs_arg, (void*)&nt_arg};
    hipKernelNodeParams kp{};
    kp.func = fn;
    kp.gridDim = dim3(grid);
    kp.blockDim = dim3(128 * nw);
    kp.sharedMemBytes = VG2_LDS_BYTES;
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

int vgru_kernel_attrs(dmp_ctx* c) {
  DMP_HIP(hipFuncSetAttribute((const void*)vgru_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, VP_LDS_BYTES));
  {
    // the persistent form is laid out for 256 CUs in 8 XCDs (one workgroup per CU, 32 per XCD)
    int cus = 0;
    DMP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    c->vgru_persist_ok = cus == VP_GRID;
  }
  DMP_HIP(hipFuncSetAttribute((const void*)vgru2_step_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, VG2_LDS_BYTES));
  DMP_HIP(hipFuncSetAttribute((const void*)vgru2_step_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, VG2_LDS_BYTES));
  DMP_HIP(hipFuncSetAttribute((const void*)vgru2_step_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, VG2_LDS_BYTES));
  if (int rc = vgru_f32_kernel_attrs(c)) return rc;
  return vgru_x3_kernel_attrs(c);
}

// Group record of n member alignments (their column tiles one after the other in the LEADER's state buffers)
// and the state cleared; all on stream s.
int vgru_group_setup(dmp_ctx* lead, dmp_ctx* const* members, const uint8_t* const* msas, const int* Ns,
                     const int* Ls, int n, hipStream_t s) {
  if (n < 1 || n > VG_MAX_MEMBERS) { set_error("a vertical-GRU group has 1..%d members, got %d", VG_MAX_MEMBERS, n); return DMP_ERR_ARG; }
  VGroupRec rec{};
  int nt = 0, maxN = 0;
  for (int i = 0; i < n; ++i) {
    if (members[i]->W.hash != lead->W.hash || !members[i]->W.ready) {
      set_error("vertical-GRU group: member %d does not hold the leader's weights", i);
      return DMP_ERR_WEIGHTS;
    }
    rec.mem[i].msa = msas[i];
    rec.mem[i].N = Ns[i];
    rec.mem[i].L = Ls[i];
    rec.mem[i].tile0 = nt;
    lead->vg_tile0[i] = nt;
    nt += vgru_pitch(Ls[i]) / VG_TB;
    maxN = std::max(maxN, Ns[i]);
  }
  if (nt * VG_TB > lead->vg_cap_cols) {
    set_error("vertical-GRU group of %d columns exceeds the leader's capacity %d", nt * VG_TB, lead->vg_cap_cols);
    return DMP_ERR_CAPACITY;
  }
  rec.nmem = n;
  rec.t0 = 0;
  rec.t_end = 0;
  const size_t hbytes = sizeof(float) * WIDTH * (size_t)nt * VG_TB;
  for (int l = 0; l < 2; ++l) {
    DMP_HIP(hipMemsetAsync(lead->hT[l][0], 0, hbytes, s));
    DMP_HIP(hipMemsetAsync(lead->hH[l][0], 0, hbytes, s));     // 2 pieces x 512 x columns x 2 bytes
  }
  hipLaunchKernelGGL(vgru_set_group_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<VGroupRec*>(lead->vgru_run), rec);
  DMP_LAUNCH_CHECK();
  lead->vg_ntiles = nt;
  lead->vg_maxN = maxN;
  return DMP_OK;
}

// time steps [t_lo, t_hi) of the group set up on `lead` (t_hi is cut at max N + 1); t_lo must be even (the buffer
// parity of a row is the parity of its node index in the chain)
int vgru_group_steps(dmp_ctx* lead, int t_lo, int t_hi, hipStream_t s) {
  const int nt = lead->vg_ntiles, nw = vgru2_nw(nt);
  if (t_hi > lead->vg_maxN + 1) t_hi = lead->vg_maxN + 1;
  if (t_lo < 0) t_lo = 0;
  if (t_lo & 1) { set_error("vertical-GRU chunks start at even rows (got %d)", t_lo); return DMP_ERR_ARG; }
  if (vgru_runs_f32(lead) == 2) return vgru_x3_group_steps(lead, t_lo, t_hi, s);
  if (vgru_runs_f32(lead)) return vgru_f32_group_steps(lead, t_lo, t_hi, s);
  VGroupRec* rec = reinterpret_cast<VGroupRec*>(lead->vgru_run);
  if (lead->vgru_persist && lead->vgru_persist_ok) {
    // one launch for all the rows: XCD-local row barriers inside (vgru_persist_kernel)
    const Weights& W = lead->W;
    VStatic st{};
    for (int l = 0; l < 2; ++l) {
      st.wx[l] = reinterpret_cast<const uint4*>(W.v_wx[l]);
      st.wh[l] = reinterpret_cast<const uint4*>(W.v_wh[l]);
      st.inv_scale[l] = W.v_inv_scale[l];
      for (int p = 0; p < 2; ++p) { st.hT[l][p] = lead->hT[l][p]; st.hH[l][p] = lead->hH[l][p]; }
    }
    st.bias[0] = W.v_b0; st.bias[1] = W.v_b1;
    if (t_lo >= t_hi) return DMP_OK;
    VPSync* sync = reinterpret_cast<VPSync*>(lead->vgru_sync);
    // needs every one of its 256 workgroups resident (row barriers): ordered against the process's other persistent
    // launches and cluster kernels on this device (CoResident, common.h)
    {
      CoResident guard(lead, s, true);
      if (guard.status()) return guard.status();
      DMP_HIP(hipMemsetAsync(sync, 0, sizeof(VPSync), s));
      hipLaunchKernelGGL(vgru_persist_kernel, dim3(VP_GRID - (lead->vgru_debug_drop_wg ? 1 : 0)), dim3(256), VP_LDS_BYTES, s, st, (const VGroupRec*)rec, sync,
                         lead->seq_abort, t_lo, t_hi, nt);
      DMP_LAUNCH_CHECK();
      int rc = guard.done();
      if (rc) return rc;
    }
    return DMP_OK;
  }
  for (int t = t_lo; t < t_hi;) {
    // whole 128-row chains, then 16-row chains: an idle node of a chain costs a full step (a step learns its row
    // number only after its main loop)
    const int len = (t_hi - t >= VGRU_CHUNK) ? VGRU_CHUNK : VGRU_CHUNK_SMALL;
    hipGraphExec_t ge;
    int rc = vgru2_graph(lead, nt, nw, vgru2_cgs(nt), len, &ge);
    if (rc) return rc;
    hipLaunchKernelGGL(vgru_set_run2_kernel, dim3(1), dim3(1), 0, s, rec, t, t_hi);
    DMP_HIP(hipGraphLaunch(ge, s));
    t += len;
  }
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// out[l][j] = state[j/4][col0 + l][j%4] with column pitch Lb.  A chain whose row barrier timed out (fault word of the
// LEADER, whose prediction the chain belongs to) hands NaN to every alignment it served: a member's or a rider's
// prediction must never go on from a half-updated state to a plausible-looking wrong structure (round 5: the members
// of a group used to get exactly that - only the leader's prediction was invalidated).
__global__ __launch_bounds__(128) void vgru2_out_kernel(const float* __restrict__ hP, int Lb, int col0,
                                                        float* __restrict__ out, const int* __restrict__ fault,
                                                        int* __restrict__ member_fault) {
  const int l = blockIdx.x, j4 = threadIdx.x;
  float4 v = *reinterpret_cast<const float4*>(hP + ((int64_t)j4 * Lb + col0 + l) * 4);
  if (fault[0] & DMP_FAULT_VGRU_HANDOFF) {
    const float nan = __builtin_nanf("");
    v = make_float4(nan, nan, nan, nan);
    if (member_fault && l == 0 && j4 == 0) atomicOr(member_fault, DMP_FAULT_VGRU_HANDOFF);
  }
  *reinterpret_cast<float4*>(out + (int64_t)l * WIDTH + 4 * j4) = v;
}

// member `mi` of the group on `lead`: its result (top-layer state after its last row) as L x 512; `member` = the context
// whose prediction consumes it (its fault word takes the chain's time-out), or null for a rider's buffer
int vgru_group_output(dmp_ctx* lead, int mi, int N, int L, float* d_out, hipStream_t s, dmp_ctx* member) {
  hipLaunchKernelGGL(vgru2_out_kernel, dim3(L), dim3(128), 0, s, lead->hT[1][N & 1], lead->vg_ntiles * VG_TB,
                     lead->vg_tile0[mi] * VG_TB, d_out, (const int*)lead->seq_abort,
                     (member && member != lead) ? member->seq_abort : (int*)nullptr);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

int gru_vertical_steps(dmp_ctx* c, const uint8_t* d_msa, int N, int L, int t_lo, int t_hi, float* d_out,
                       hipStream_t s) {
  int rc;
  if (t_lo <= 0) {
    t_lo = 0;
    if ((rc = vgru_group_setup(c, &c, &d_msa, &N, &L, 1, s))) return rc;
  }
  if (t_hi > N + 1) t_hi = N + 1;
  if ((rc = vgru_group_steps(c, t_lo, t_hi, s))) return rc;
  if (t_hi == N + 1) return vgru_group_output(c, 0, N, L, d_out, s, c);
  return DMP_OK;
}

int gru_vertical(dmp_ctx* c, const uint8_t* d_msa, int N, int L, float* d_out, hipStream_t s) {
  return gru_vertical_steps(c, d_msa, N, L, 0, N + 1, d_out, s);
}


// ---------------------------------------------------------------------------------------
// CoResident (common.h): per-device launch order of the kernels that wait for their own workgroups
// ---------------------------------------------------------------------------------------
namespace {
struct CoResidentDevice {
  hipEvent_t persist = nullptr;                       // completion of the most recent persistent launch
  std::map<dmp_ctx*, hipEvent_t> cluster;             // per context: completion of its most recent cluster kernel
};
std::mutex co_mu;
std::map<int, CoResidentDevice> co_dev;
}  // namespace

CoResident::CoResident(dmp_ctx* c, hipStream_t s, bool persistent)
    : c_(c), s_(s), persistent_(persistent), locked_(true), rc_(DMP_OK) {
  co_mu.lock();
  CoResidentDevice& d = co_dev[c->device];
  if (d.persist && hipStreamWaitEvent(s, d.persist, 0) != hipSuccess) rc_ = DMP_ERR_HIP;
  if (persistent)
    for (auto& kv : d.cluster)
      if (hipStreamWaitEvent(s, kv.second, 0) != hipSuccess) rc_ = DMP_ERR_HIP;
  if (rc_) set_error("hipStreamWaitEvent failed while ordering a co-resident launch");
}

int CoResident::done() {
  CoResidentDevice& d = co_dev[c_->device];
  hipEvent_t& ev = persistent_ ? d.persist : d.cluster[c_];
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) rc_ = DMP_ERR_HIP;
  if (!rc_ && hipEventRecord(ev, s_) != hipSuccess) rc_ = DMP_ERR_HIP;
  if (rc_) set_error("hipEventRecord failed while ordering a co-resident launch");
  locked_ = false;
  co_mu.unlock();
  return rc_;
}

CoResident::~CoResident() {
  if (locked_) co_mu.unlock();
}

void coresident_forget(dmp_ctx* c) {
  std::lock_guard<std::mutex> lock(co_mu);
  auto it = co_dev.find(c->device);
  if (it == co_dev.end()) return;
  auto e = it->second.cluster.find(c);
  if (e != it->second.cluster.end()) {
    (void)hipEventDestroy(e->second);
    it->second.cluster.erase(e);
  }
}

}  // namespace dmp
