// conv 5x5 (128 -> 512) + bias + 4-way maxout with float32 semantics on the bf16 matrix cores.
//
// Every float32 operand is split EXACTLY into three bf16 pieces (x = x0 + x1 + x2: 3 x 8 significand
// bits = the 24 of a float32; round-to-nearest pieces, so a finite float32 is the exact sum).  Of the nine piece
// products w_i * x_j the six with i + j <= 2 are accumulated in float32 by v_mfma_f32_32x32x16_bf16; the three
// dropped ones are below 2^-24 of the product, i.e. below the rounding of a float32 multiply.  The result equals
// the float32 convolution to float32 rounding error (measured: max error vs float64 2e-6, the plain f32 kernel
// 5e-6) at 16/6 = 2.67x the f32 MFMA rate.  No range limit (bf16 has float32's exponent), no scales.
// This is the convolution of option "precision" = 2: full-width (24-bit) operands on the 16-bit matrix cores.
//
// Round 6: row reuse, the structure of conv_f16.h.  The 32 pixels of the MFMA N axis are 2 rows x 16 columns, so
// accumulator q (rows 2q, 2q+1 of the 16 x 16 tile) at tap (dy, dx) reads the B fragment "row pair r = 2q + dy at
// column offset dx": one fragment (x 3 pieces) serves every (q, dy) with 2q + dy = r.  A tap column dx is two passes -
// taps dy = 0, 2, 4 against r = 0, 2, .., 18 and taps dy = 1, 3 against r = 1, 3, .., 17 - with the pass's weight
// fragments (9 / 6 x 16 bytes per lane) in registers and a per-wave weight buffer of 3 tap slots (9 KB) refilled by
// LDS-DMA for the next pass as soon as the fragments are in registers.  72 LDS reads per tap column and wave instead
// of 135 for the same 240 MFMAs (the round-1 kernel went tap by tap: 3 weight + 24 activation fragments per tap).
//
// Layouts (one 16-byte load = one MFMA operand)
//   activations  xs[piece 3][c/8 16][P][P][8]   bf16, zero border (same geometry as the f32 planes)
//   weights      wq[split 4][cgrp 8][dx 5][wave 4][dy: 0 2 4 1 3][piece 3][cg 2][m 32][8]   bf16
//                (conv channel = split*128 + wave*32 + m, input channel = cgrp*16 + cg*8 + e):
//                one wave's operands for one pass are 9 / 6 KB contiguous = 1 KB LDS-DMA pieces
// Workgroup = 4 waves x (32 conv channels each) x one 16x16 (small L: 8x16, CqShape) pixel tile; 8 input stages of 16 channels, whose 20 x 20
// halo tile (3 pieces) is shared by the waves (2 barriers per stage); no workgroup barrier inside a stage.
// Lane -> pixel of a fragment follows the lane groups ds_read_b128 is served in (conv_f16.h): conflict free at any
// row pitch >= 20.
#pragma once
#include "common.h"
#include <vector>

namespace dmp {

typedef float cq_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cq_bf16x8 __attribute__((ext_vector_type(8)));

#ifndef CQ_PITCH_N
#define CQ_PITCH_N 20        // row pitch of the halo tile in LDS (16-byte slots)
#endif
constexpr int CQ_PITCH = CQ_PITCH_N;
constexpr int CQ_HCOLS = 20;                                          // columns of the halo tile
// Tile shapes as in conv_f16.h (ChShape): NQ = 8 -> 16 x 16 pixels, NQ = 4 -> 8 rows x 16 columns for small L.
template <int NQ> struct CqShape {
  static constexpr int ROWS = 2 * NQ;
  static constexpr int HALO = ROWS + 4;                                 // 20 or 12 halo rows
  static constexpr int RP = ROWS + 3;                                   // row pairs
  static constexpr int PIECE = 2 * HALO * CQ_PITCH;                     // slots of one piece
  static constexpr int IN_SLOTS = 3 * PIECE;                            // 2400 / 1440 16-byte slots at pitch 20
  static constexpr int IN_PAD = (IN_SLOTS + 63) / 64 * 64;              // the tile's DMA goes in whole waves of 64 slots
  static constexpr int IN_BYTES = IN_PAD * 16;                          // 38912 / 23552
};
constexpr int CQ_WSLOT = 3 * 2 * 32;                                  // 192 slots = 3 KB per wave and tap
constexpr int CQ_WCOL = 5 * CQ_WSLOT;                                 // one tap column of one wave in the packed weights
constexpr int CQ_WBUF = 3 * CQ_WSLOT;                                 // per-wave LDS weight buffer: 3 tap slots
template <int NQ> constexpr int convq_lds_bytes() { return CqShape<NQ>::IN_BYTES + 4 * CQ_WBUF * 16; }     // 75776 / 60416
constexpr int CONVQ_LDS_BYTES = convq_lds_bytes<8>();                 // two workgroups per CU

// round-to-nearest-even float32 -> bf16 bits
__host__ __device__ inline uint16_t cq_bf16_rne(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__host__ __device__ inline float cq_bf16_f32(uint16_t h) {
  return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// x == p[0] + p[1] + p[2] exactly (finite x)
__host__ __device__ inline void split3_bf16(float x, uint16_t p[3]) {
  p[0] = cq_bf16_rne(x);
  const float r1 = x - cq_bf16_f32(p[0]);
  p[1] = cq_bf16_rne(r1);
  const float r2 = r1 - cq_bf16_f32(p[1]);
  p[2] = cq_bf16_rne(r2);
}

// w: [512][128][5][5] float32 -> packed bf16 pieces (layout above)
inline std::vector<uint16_t> pack_conv_weights_bf16(const float* w) {
  const int tap_row[5] = {0, 2, 4, 1, 3};              // slot order inside a column: even pass, then odd pass
  std::vector<uint16_t> q((size_t)4 * 8 * 25 * 4 * 3 * 2 * 32 * 8);
  for (int split = 0; split < 4; ++split)
    for (int g = 0; g < 8; ++g)
      for (int dx = 0; dx < 5; ++dx)
        for (int wave = 0; wave < 4; ++wave)
          for (int sl = 0; sl < 5; ++sl)
            for (int cg = 0; cg < 2; ++cg)
              for (int m = 0; m < 32; ++m)
                for (int e = 0; e < 8; ++e) {
                  const int oc = split * 128 + wave * 32 + m, ic = g * 16 + cg * 8 + e;
                  uint16_t p3[3];
                  split3_bf16(w[((size_t)oc * 128 + ic) * 25 + tap_row[sl] * 5 + dx], p3);
                  for (int p = 0; p < 3; ++p)
                    q[(((((((((size_t)split * 8 + g) * 5 + dx) * 4 + wave) * 5 + sl) * 3 + p) * 2 + cg) * 32 + m) * 8) + e] = p3[p];
                }
  return q;
}

#if defined(__HIPCC__) && defined(CONV_BF16_KERNELS)   // kernels: only the unit that launches them
// LDS-DMA of 16 bytes per lane; destination = wave-uniform LDS byte address + lane*16.  Issued
// through inline asm so that hipcc does not serialise later LDS reads behind it; completion is
// waited for with cq_wait_vm<N>().
__device__ __forceinline__ void cq_dma16(const void* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
template <int N> __device__ __forceinline__ void cq_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__device__ __forceinline__ cq_f32x16 cq_mfma(uint4 a, uint4 b, cq_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cq_bf16x8, a),
                                                 __builtin_bit_cast(cq_bf16x8, b), c, 0, 0, 0);
}

// lane (0..31 of a half wave) -> (row, column) of the 2 x 16 pixel fragment (the map of conv_f16.h)
__device__ __forceinline__ void cq_lane_pixel(int li, int& row, int& x) {
  const bool a = li < 4 || (li >= 12 && li < 16) || (li >= 20 && li < 28);
  row = a ? 0 : 1;
  if (a) x = li < 4 ? li : (li < 16 ? li - 8 : li - 12);
  else x = li < 12 ? li - 4 : (li < 20 ? li - 8 : li - 16);
}

// One pass of a tap column: NT taps (rows dy = PAR, PAR + 2, ..) whose weight fragments a[t][piece] sit in registers,
// against the row pairs r = PAR, PAR + 2, .. < 19 read from the halo tile at `il`; the next row pair's three pieces are
// requested before this one's MFMAs.  Per fragment the smallest products go first: w0 x2, w1 x1, w2 x0 (2^-16), then
// w0 x1, w1 x0 (2^-8), then w0 x0 - each product type over the pass's taps, i.e. over different accumulators.
template <int NQ, int NT, int PAR>
__device__ __forceinline__ void cq_column_pass(const uint4 (&a)[NT][3], const uint4* il, cq_f32x16 (&acc)[NQ]) {
  constexpr int PIECE = CqShape<NQ>::PIECE, RP = CqShape<NQ>::RP;
  uint4 bn0 = il[PAR * CQ_PITCH], bn1 = il[PAR * CQ_PITCH + PIECE], bn2 = il[PAR * CQ_PITCH + 2 * PIECE];
#pragma unroll
  for (int r = PAR; r < RP; r += 2) {
    const uint4 b0 = bn0, b1 = bn1, b2 = bn2;
    if (r + 2 < RP) {
      bn0 = il[(r + 2) * CQ_PITCH];
      bn1 = il[(r + 2) * CQ_PITCH + PIECE];
      bn2 = il[(r + 2) * CQ_PITCH + 2 * PIECE];
    }
#define CQ_TAPS(AP, BV)                                                           \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                              \
      const int q = (r - PAR - 2 * t) / 2;                                        \
      if (r - PAR - 2 * t >= 0 && q < NQ) acc[q] = cq_mfma(a[t][AP], BV, acc[q]); \
    }
    CQ_TAPS(0, b2)
    CQ_TAPS(1, b1)
    CQ_TAPS(2, b0)
    CQ_TAPS(0, b1)
    CQ_TAPS(1, b0)
    CQ_TAPS(0, b0)
#undef CQ_TAPS
  }
}

// number of workgroups to launch for tiles x tiles 16 x 16 pixel tiles cut into `bands` row bands each: the XCD-aware
// block map of conv_f16.h (block b runs on XCD b % 8; XCD x works on ONE channel split (x & 3) of a contiguous half of
// the tiles, so its share of the weight pieces - 2.46 MB - stays in its 4 MB L2)
inline int conv_bf16_grid(int tiles, int bands = 1) {
  const int nt = tiles * tiles * bands;
  return 8 * ((nt + 1) / 2);
}

// grid: conv_bf16_grid(tiles, 16 / (2 NQ)) blocks   block: 256   dynamic LDS: convq_lds_bytes<NQ>() (two workgroups per CU)
// part: [tiles * tiles * 2 half tiles][CW][2] float64, as conv5x5_f16x3_kernel
template <int NQ>
__global__ __launch_bounds__(256, 2) void conv5x5_bf16x6_kernel(const uint16_t* __restrict__ xs,
                                                                const uint16_t* __restrict__ wq,
                                                                const float* __restrict__ bias, int L, int P,
                                                                int tiles, int nwork, float* __restrict__ u,
                                                                double* __restrict__ part) {
  using SH = CqShape<NQ>;
  constexpr int BANDS = 8 / NQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char cq_smem[];
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int ntiles = tiles * tiles * BANDS;
  const int tper = (ntiles + 1) >> 1;
  const int tile = (xcd >> 2) * tper + slot;
  const int split = xcd & 3;
  if (slot >= tper || tile >= ntiles) return;
  const int trow = tile / tiles, tcol = tile % tiles;
  const int ty0 = trow * SH::ROWS, tx0 = tcol * CONV_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int64_t PP = (int64_t)P * P;
  const int64_t half0 = ((int64_t)(trow / BANDS) * tiles + tcol) * 2 + (trow % BANDS);
  if (BANDS == 2 && ty0 >= L) {
    // a band below the last row (L % 16 in 1 .. 8): nothing to convolve, but the reduction reads this half tile's sums
    if (li == 0)
      for (int g4 = 0; g4 < 4; ++g4) {
        const int gch = split * 32 + wave * 8 + 2 * g4 + kk;
        part[(half0 * CW + gch) * 2 + 0] = 0.0;
        part[(half0 * CW + gch) * 2 + 1] = 0.0;
      }
    return;
  }

  const uint4* in_l = reinterpret_cast<const uint4*>(cq_smem);
  const uint4* w_l = reinterpret_cast<const uint4*>(cq_smem + SH::IN_BYTES) + wave * CQ_WBUF;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)cq_smem;
  const unsigned w_lds_addr = lds_base + SH::IN_BYTES + wave * (CQ_WBUF * 16);

  // input-tile DMA plan: slot s = e*256 + tid (IN_PAD slots: the pad slots and the slots of the row pitch beyond the
  // 20 halo columns re-read a valid pixel)
  const uint4* xs4 = reinterpret_cast<const uint4*>(xs);
  constexpr int NE = (SH::IN_PAD + 255) / 256;
  int in_src[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int s = e * 256 + tid;
    const int sc = s < SH::IN_SLOTS ? s : 0;
    const int p = sc / SH::PIECE, r = sc % SH::PIECE;
    const int cg = r / (SH::HALO * CQ_PITCH), r2 = r % (SH::HALO * CQ_PITCH);
    const int yy = r2 / CQ_PITCH;
    int xx = r2 % CQ_PITCH;
    xx = xx < CQ_HCOLS ? xx : 0;
    in_src[e] = (int)(((int64_t)(p * 16 + cg) * P + ty0 + yy) * P + tx0 + xx);
  }
  const uint4* wq4 = reinterpret_cast<const uint4*>(wq) + (int64_t)split * 8 * 5 * 4 * CQ_WCOL +
                     (int64_t)wave * CQ_WCOL + lane;
  int prow, px;
  cq_lane_pixel(li, prow, px);
  const int b_base = (kk * SH::HALO + prow) * CQ_PITCH + px;      // + r * CQ_PITCH + dx (+ piece stride)
  const int a_off = kk * 32 + li;

  cq_f32x16 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // pass h = 2 * (g * 5 + dx) + odd: stream its 3 (even) or 2 (odd) tap slots into this wave's buffer
  auto wdma = [&](int h) {
    const int odd = h & 1;
    const uint4* src = wq4 + (int64_t)(h >> 1) * 4 * CQ_WCOL + odd * 3 * CQ_WSLOT;
    if (odd) {
#pragma unroll
      for (int i = 0; i < 6; ++i) cq_dma16(src + 64 * i, w_lds_addr + 1024 * i);
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) cq_dma16(src + 64 * i, w_lds_addr + 1024 * i);
    }
  };
  wdma(0);

  for (int g = 0; g < 8; ++g) {
    __syncthreads();                                   // every wave is done with the previous tile
    {
      const uint4* src = xs4 + (int64_t)g * 2 * PP;
      const unsigned dst = lds_base + (wave * 64) * 16;
#pragma unroll
      for (int e = 0; e < SH::IN_PAD / 256; ++e) cq_dma16(src + in_src[e], dst + e * 4096);
      if (wave * 64 < SH::IN_PAD % 256) cq_dma16(src + in_src[NE - 1], dst + (SH::IN_PAD / 256) * 4096);
    }
    cq_wait_vm<0>();
    __syncthreads();                                   // the tile of every wave has landed
#pragma unroll 1
    for (int dx = 0; dx < 5; ++dx) {
      const uint4* il = in_l + b_base + dx;
      const int h0 = 2 * (g * 5 + dx);
      {
        // this pass's weights: issued one pass ago (at the head of a stage everything was drained with the tile)
        cq_wait_vm<0>();
        uint4 a[3][3];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int p = 0; p < 3; ++p) a[t][p] = w_l[t * CQ_WSLOT + p * 64 + a_off];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wdma(h0 + 1);                                  // the buffer is free: stream the next pass
        cq_column_pass<NQ, 3, 0>(a, il, acc);
      }
      {
        cq_wait_vm<0>();
        uint4 a[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < 3; ++p) a[t][p] = w_l[t * CQ_WSLOT + p * 64 + a_off];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h0 + 2 < 80) wdma(h0 + 2);
        cq_column_pass<NQ, 2, 1>(a, il, acc);
      }
    }
  }

  // ---- epilogue: bias, 4-way max, store, per-channel partial sums per 8-row half tile (no cross-wave reduction:
  // a wave owns its 32 conv channels = 8 maxout channels for all pixels of the tile)
  const float* bsp = bias + split * 128 + wave * 32;
  const int64_t LL = (int64_t)L * L;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int cl = 8 * g4 + 4 * kk;                        // first conv channel of the group in the wave
    const int gch = split * 32 + wave * 8 + 2 * g4 + kk;    // maxout channel
    const float b0 = bsp[cl], b1 = bsp[cl + 1], b2 = bsp[cl + 2], b3 = bsp[cl + 3];
#pragma unroll
    for (int hq = 0; hq < NQ / 4; ++hq) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 4 * hq; q < 4 * hq + 4; ++q) {
        float v = acc[q][4 * g4] + b0;
        v = fmaxf(v, acc[q][4 * g4 + 1] + b1);
        v = fmaxf(v, acc[q][4 * g4 + 2] + b2);
        v = fmaxf(v, acc[q][4 * g4 + 3] + b3);
        const int y = ty0 + 2 * q + prow, x = tx0 + px;
        if (y < L && x < L) {
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
        part[((half0 + hq) * CW + gch) * 2 + 0] = (double)s1;
        part[((half0 + hq) * CW + gch) * 2 + 1] = (double)s2;
      }
    }
  }
}
#endif  // __HIPCC__ && CONV_BF16_KERNELS

}  // namespace dmp
