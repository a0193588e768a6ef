// conv 5x5 (128 -> 512) + bias + 4-way maxout with float32-grade accuracy on the f16 matrix cores.
//
// Every float32 operand is split into two f16 pieces, x = x0 + x1 with x0 = f16(x), x1 = f16(x - x0)
// (22 significand bits; the weights are first scaled by a power of two S so that their low pieces
// stay normal numbers: S*w = w0 + w1).  The three products w0 x0, w0 x1, w1 x0 are accumulated in
// float32 by v_mfma_f32_32x32x16_f16 and the accumulator is scaled back by 1/S (exact).  The dropped
// w1 x1 term is 2^-22 of the product.  Measured against a float64 convolution the result has the
// SAME error as a float32 convolution (CPU emulation: max 4.7e-6 vs 4.9e-6 for PyTorch's float32
// conv on the same data; GPU: tests/test_gpu_parity.py::test_conv_paths_error_vs_float64) because
// the float32 accumulation of the 3200-term sums dominates both.  Cost: 3 f16 MFMAs per float32
// MFMA-equivalent = 16/3 = 5.3x the f32 matrix-core rate.
//
// Range and activation scale (round 4).  The pieces are taken of 2^e x, with e chosen per residual block at
// dmp_weights_finalize (api.hip: the largest e that keeps a bound of the block's input, built from the InstanceNorm
// gamma / beta of the blocks before it, below 32768; undone exactly, together with the weight scale, in this kernel's
// epilogue), so the low pieces of the bulk of the activations are NORMAL f16 numbers whatever scale the trunk sits at
// (unscaled, a value below 0.125 has a subnormal low piece and one below 6e-5 a subnormal high piece: 25 x the
// float32 kernel's error at trunk scale 2^-16, tests/test_gpu_parity.py::
// test_unscaled_pieces_lose_precision_on_small_activations).  |2^e x| must stay below 65504 (f16 max): the producer
// kernels raise the context's fault flag at |2^e x| >= 60000; a trunk that would leave the range unscaled is scaled
// DOWN instead.  Option "act_scaling" = 0 restores unscaled pieces (e = 0).
//
// Workgroup = 4 waves x 32 conv channels x one 16x16 (or, at small L, 8x16: ChShape) pixel tile (one of 4 channel splits);
// 8 input stages of 16 channels, whose 20x20 halo tile (both pieces, 30 KB) is brought in by LDS-DMA.
//
// Row reuse.  The 32 pixels of the MFMA N axis are 2 rows x 16 columns, so accumulator q (rows 2q, 2q+1
// of the tile) at tap (dy, dx) reads the B fragment "row pair r = 2q + dy at column offset dx": one
// fragment serves every (q, dy) with 2q + dy = r.  For a tap column dx the wave therefore loads 19
// row-pair fragments (x 2 pieces) instead of 8 x 5 = 40, and keeps the column's weight fragments in
// registers.  Even r only meets even dy and odd r odd dy, so a column is two passes - taps dy = 0, 2, 4
// against r = 0, 2, .., 18 and taps dy = 1, 3 against r = 1, 3, .., 17 - with 24 and 16 weight registers
// and a 6 KB weight buffer per wave (3 tap slots, refilled by LDS-DMA for the next pass as soon as the
// fragments are in registers).  240 LDS reads per stage and wave instead of 450 for the same 600 MFMAs:
// Sustained (tools/ubench_conv_sustained.hip; the chip sits at its power cap, 1340 W at 1.83 GHz, so a
// 35 ms burst flatters any change): 0.655-0.660 ms per launch at L = 300 against 0.676 for the tap-by-tap
// kernel it replaces, 0.628 against 0.648 with two launches in flight.
//
// Layouts (one 16-byte load = one MFMA operand):
//   activations  xs[piece 2][c/8 16][P][P][8] f16
//   weights      wq[split 4][cgrp 8][dx 5][wave 4][dy: 0 2 4 1 3][piece 2][cg 2][m 32][8] f16
// Lane -> pixel of a fragment follows the lane groups ds_read_b128 is served in ({0-3,12-15,20-27} and
// {4-11,16-19,28-31} per half wave): each group reads the 16 consecutive slots of one row = every LDS
// bank once, whatever the row pitch.
#pragma once
#include "common.h"
#include <cmath>
#include <vector>

namespace dmp {

typedef float ch_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ch_f16x8 __attribute__((ext_vector_type(8)));

#ifndef CH_PITCH_N
#define CH_PITCH_N 24        // row pitch of the halo tile in LDS (16-byte slots); any value >= 20 is conflict free.
                             // 24 -> 54 KB per workgroup: two per CU and room for other targets' kernels.  20 -> 50 KB
                             // lets a third workgroup in (3 waves per SIMD): sustained 0.665 against 0.667 ms - at
                             // the power cap more occupancy buys nothing, and the room beside the two is worth more
#endif
// Tile shapes (round 6).  NQ = accumulators per wave = row pairs of the pixel tile: 8 -> 16 x 16 pixels (the shape of
// rounds 1-5), 4 -> 8 rows x 16 columns.  Measured (profiles/r06_conv_tile_shapes.txt): the duration of a launch is
// ceil(workgroups / 256 CUs) x T1, T1 = 0.115 ms (f16x3) / 0.175 ms (bf16x6) the time of ONE 16 x 16 workgroup on a CU - a
// second workgroup on the CU doubles it, so what is lost at mid L is the rounding up (L = 200: 676 workgroups = 2.64 per
// CU -> 3; L = 300: 5.64 -> 6).  Half-height tiles halve the unit but cost 1.2 x per MFMA (1.8 instead of 2.1 MFMAs per
// LDS fragment, a 12-row halo for 8 rows, the weight stream per workgroup unchanged): ceil(2 x 2.64) x 0.6 = 3.6 > 3 -
// they LOSE wherever more than 256 half-height workgroups exist (L = 200: 0.399 against 0.356 ms; L = 82 .. 128: equal)
// and WIN where the 16 x 16 shape leaves half the CUs idle: L <= 80 (L = 48: 0.064 against 0.114 ms).  conv_split_rows
// picks them there.  The accumulation order of an output element does not depend on the shape (per stage and tap column:
// taps dy ascending, the three products of a tap in the order below), and the InstanceNorm partial sums are formed per
// 8-row half tile in both shapes: the same bits.
constexpr int CH_PITCH = CH_PITCH_N;
template <int NQ> struct ChShape {
  static constexpr int ROWS = 2 * NQ;                                   // pixel rows of the tile
  static constexpr int HALO = ROWS + 4;                                 // rows of the halo tile (20 or 12)
  static constexpr int RP = ROWS + 3;                                   // row pairs r = 0 .. RP - 1 (19 or 11)
  static constexpr int IN_SLOTS = 2 * 2 * HALO * CH_PITCH;              // 1920 / 1152 16-byte slots
  static constexpr int IN_BYTES = IN_SLOTS * 16;                        // 30720 / 18432
  static constexpr int PIECE = 2 * HALO * CH_PITCH;                     // slots of one piece
};
constexpr int CH_HCOLS = 20;                                          // columns of the halo tile
constexpr int CH_WSLOT = 2 * 2 * 32;                                  // 128 slots = 2 KB per wave and tap
constexpr int CH_WCOL = 5 * CH_WSLOT;                                 // one tap column of one wave in the packed weights
constexpr int CH_WBUF = 3 * CH_WSLOT;                                 // per-wave LDS weight buffer: 3 tap slots
template <int NQ> constexpr int convh_lds_bytes() { return ChShape<NQ>::IN_BYTES + 4 * CH_WBUF * 16; }      // 55296 / 43008
constexpr int CONVH_LDS_BYTES = convh_lds_bytes<8>();
// row bands per 16-row tile row for `tiles` x `tiles` 16 x 16 tiles: 2 = the half-height shape, where its workgroups
// (2 x 4 x tiles^2) still find a CU each
inline int conv_split_rows(int tiles) { return 8 * tiles * tiles <= 256 ? 2 : 1; }
// number of workgroups to launch for tiles x tiles 16 x 16 pixel tiles cut into `bands` row bands each (XCD-aware block map)
// block -> (tile, channel split) map (round 3, tools/r03_map.sh, profiles/r03_conv_block_maps.txt; block b runs on XCD
// b % 8).  Sustained ms per launch at L = 300 (one stream / two launches in flight), L2-fabric bytes per launch
// (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included; 98.7 MB algorithmic), scheduler throughput:
//   0  XCD x: splits {0,1} or {2,3} of a contiguous quarter of the tiles (rounds 1-2)   0.686 / 0.652   306 MB   7.22
//   1  XCD x: all four splits of an eighth of the tiles (a tile read by ONE XCD)         0.687 / 0.659   399 MB
//   2  XCD x: ONE split of half of the tiles (1.64 MB of weight pieces per XCD)           0.675 / 0.645   339 MB   7.28
// Fewer XCDs per tile do not lower the traffic - the weight set no longer fits the L2 beside the activations and comes
// back from the Infinity Cache - and the map with the MOST fabric bytes is the fastest: its weights stay in the L2, and
// what the kernel waits for is the weight stream, not the bytes.  Map 2 is the one built; 0 and 1 are history.
inline int conv_f16_grid(int tiles, int bands = 1) {
  const int nt = tiles * tiles * bands;
  return 8 * ((nt + 1) / 2);
}

__host__ __device__ inline uint16_t ch_f16_bits(float f) {
  const _Float16 h = (_Float16)f;                                     // round to nearest even
  return __builtin_bit_cast(uint16_t, h);
}
__host__ __device__ inline float ch_f16_f32(uint16_t b) {
  return (float)__builtin_bit_cast(_Float16, b);
}
// x ~= p[0] + p[1]
__host__ __device__ inline void split2_f16(float x, uint16_t p[2]) {
  p[0] = ch_f16_bits(x);
  p[1] = ch_f16_bits(x - ch_f16_f32(p[0]));
}

// power-of-two scale that puts max|w| into [512, 1024)
inline float conv_weight_scale_f16(const float* w, size_t n) {
  float m = 0.f;
  for (size_t i = 0; i < n; ++i) m = std::fmax(m, std::fabs(w[i]));
  if (!(m > 0.f)) return 1.f;
  int e;
  std::frexp(m, &e);                    // m = f * 2^e, f in [0.5, 1)
  return std::ldexp(1.0f, 10 - e);      // m * scale in [512, 1024)
}

// w: [512][128][5][5] float32 -> packed f16 pieces of scale * w (layout above)
inline std::vector<uint16_t> pack_conv_weights_f16(const float* w, float scale) {
  const int tap_row[5] = {0, 2, 4, 1, 3};              // slot order inside a column: even pass, then odd pass
  std::vector<uint16_t> q((size_t)4 * 8 * 25 * 4 * 2 * 2 * 32 * 8);
  for (int split = 0; split < 4; ++split)
    for (int g = 0; g < 8; ++g)
      for (int dx = 0; dx < 5; ++dx)
        for (int wave = 0; wave < 4; ++wave)
          for (int sl = 0; sl < 5; ++sl)
            for (int cg = 0; cg < 2; ++cg)
              for (int m = 0; m < 32; ++m)
                for (int e = 0; e < 8; ++e) {
                  const int oc = split * 128 + wave * 32 + m, ic = g * 16 + cg * 8 + e;
                  uint16_t p2[2];
                  split2_f16(scale * w[((size_t)oc * 128 + ic) * 25 + tap_row[sl] * 5 + dx], p2);
                  for (int p = 0; p < 2; ++p)
                    q[(((((((((size_t)split * 8 + g) * 5 + dx) * 4 + wave) * 5 + sl) * 2 + p) * 2 + cg) * 32 + m) * 8) + e] = p2[p];
                }
  return q;
}

#if defined(__HIPCC__) && defined(CONV_F16_KERNELS)   // kernels: only the unit that launches them
__device__ __forceinline__ void ch_dma16(const void* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
template <int N> __device__ __forceinline__ void ch_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ ch_f32x16 ch_mfma(uint4 a, uint4 b, ch_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ch_f16x8, a),
                                                __builtin_bit_cast(ch_f16x8, b), c, 0, 0, 0);
}
// lane (0..31 of a half wave) -> (row, column) of the 2 x 16 pixel fragment
__device__ __forceinline__ void ch_lane_pixel(int li, int& row, int& x) {
  const bool a = li < 4 || (li >= 12 && li < 16) || (li >= 20 && li < 28);
  row = a ? 0 : 1;
  if (a) x = li < 4 ? li : (li < 16 ? li - 8 : li - 12);
  else x = li < 12 ? li - 4 : (li < 20 ? li - 8 : li - 16);
}

// Issue efficiency is NOT what bounds this kernel (round 3, gpurun_out r03d, profiles/r03_conv_variants.txt).  The
// compiler sinks a row pair's ds_reads down to their first use (a wait for a read issued two instructions earlier
// before every third to ninth MFMA: the LDS latency shows on every row pair).  Two variants were built and measured in
// round 3 (removed in round 4): "pinned" - a scheduling barrier behind the reads keeps them one row pair ahead (188
// VGPRs, lifting the 168 cap) - and "pipelined" - additionally the next pass's weight fragments and first B fragment
// fetched during the current one (206 VGPRs).  Sustained at L = 300, same box: 0.659 ms per launch as shipped, 0.654
// uncapped, 0.650 pinned, 0.647 pipelined (two launches in flight: 0.646 / 0.644 / 0.642 / 0.637): the stalls go and
// 1.8 % of the time with them - the chip sits at its power cap (section 4 of DESIGN.md) and gives the issue slots
// back as clock.  In the scheduler the shipped form is the fastest (6.95 structures/s against 6.82 .. 6.91): the
// larger footprints take registers from the kernels that run beside the convolutions.
// One pass of a tap column: NT taps (rows dy = PAR, PAR + 2, ..) whose weight fragments a[t][piece] sit in
// registers, against the row pairs r = PAR, PAR + 2, .. < 19 read from the halo tile at `il`.
// Per fragment the small products go first (w0 x1, w1 x0), then w0 x0.
template <int NQ, int NT, int PAR>
__device__ __forceinline__ void ch_column_pass(const uint4 (&a)[NT][2], const uint4* il, ch_f32x16 (&acc)[NQ]) {
  constexpr int PIECE = ChShape<NQ>::PIECE, RP = ChShape<NQ>::RP;
  uint4 bn0 = il[PAR * CH_PITCH], bn1 = il[PAR * CH_PITCH + PIECE];
#pragma unroll
  for (int r = PAR; r < RP; r += 2) {
    const uint4 b0 = bn0, b1 = bn1;
    if (r + 2 < RP) {
      bn0 = il[(r + 2) * CH_PITCH];
      bn1 = il[(r + 2) * CH_PITCH + PIECE];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int q = (r - PAR - 2 * t) / 2;
      if (r - PAR - 2 * t >= 0 && q < NQ) acc[q] = ch_mfma(a[t][0], b1, acc[q]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int q = (r - PAR - 2 * t) / 2;
      if (r - PAR - 2 * t >= 0 && q < NQ) acc[q] = ch_mfma(a[t][1], b0, acc[q]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int q = (r - PAR - 2 * t) / 2;
      if (r - PAR - 2 * t >= 0 && q < NQ) acc[q] = ch_mfma(a[t][0], b0, acc[q]);
    }
  }
}

// Footprint: the register budget is capped at 168 VGPRs (3 waves per SIMD; the input-tile DMA plan is
// recomputed per stage instead of living in registers - kept there, the cap parked it and some epilogue
// constants in 68 bytes of scratch per lane, 25 MB of extra writes per launch in WRITE_SIZE; 8 bytes remain),
// and 54 KB of LDS let exactly two workgroups share a CU (a third does not fit).  Two of them leave 176
// registers per SIMD lane and 52 KB of LDS to the kernels of other targets that run beside the convolutions in
// throughput mode (vertical GRU step 156 registers / 32 KB, norm 108, Gauss-Jordan update 92); uncapped
// (194 registers) the same kernel cost the scheduler 6 % although it is faster alone.
// grid: conv_f16_grid(tiles, 16 / (2 NQ)) blocks (XCD-aware map)   block: 256   dynamic LDS: convh_lds_bytes<NQ>()
// part: [tiles * tiles * 2 half tiles][CW][2] float64 - half tile h of 16 x 16 tile t at index 2 t + h, in both shapes
template <int NQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv5x5_f16x3_kernel(const uint16_t* __restrict__ xs,
                                                               const uint16_t* __restrict__ wq,
                                                               const float* __restrict__ bias, float inv_scale,
                                                               int L, int P, int tiles, int nwork,
                                                               float* __restrict__ u, double* __restrict__ part) {
  using SH = ChShape<NQ>;
  constexpr int BANDS = 8 / NQ;                          // row bands per 16-row tile row
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int ntiles = tiles * tiles * BANDS;
  // XCD x works on ONE channel split (x & 3) of a contiguous half of the tiles: its 1.64 MB of weight pieces stay in
  // its 4 MB L2; a tile's input is read by four XCDs at about the same time (Infinity-Cache hits)
  const int tper = (ntiles + 1) >> 1;
  const int tile = (xcd >> 2) * tper + slot;
  const int split = xcd & 3;
  if (slot >= tper || tile >= ntiles) return;
  const int trow = tile / tiles, tcol = tile % tiles;    // trow counts bands of SH::ROWS rows
  const int ty0 = trow * SH::ROWS, tx0 = tcol * CONV_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int64_t PP = (int64_t)P * P;
  // index of this tile's first half tile in `part`: 16 x 16 tile (trow / BANDS, tcol), half trow % BANDS
  const int64_t half0 = ((int64_t)(trow / BANDS) * tiles + tcol) * 2 + (trow % BANDS);
  int prow, px;
  ch_lane_pixel(li, prow, px);
  if (BANDS == 2 && ty0 >= L) {
    // a band below the last row (L % 16 in 1 .. 8): nothing to convolve, but the reduction reads this half tile's sums
    if (li == 0 && prow == 0 && px == 0)
      for (int g4 = 0; g4 < 4; ++g4) {
        const int gch = split * 32 + wave * 8 + 2 * g4 + kk;
        part[(half0 * CW + gch) * 2 + 0] = 0.0;
        part[(half0 * CW + gch) * 2 + 1] = 0.0;
      }
    return;
  }

  const uint4* in_l = reinterpret_cast<const uint4*>(ch_smem);
  const uint4* w_l = reinterpret_cast<const uint4*>(ch_smem + SH::IN_BYTES) + wave * CH_WBUF;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ch_smem;
  const unsigned w_lds_addr = lds_base + SH::IN_BYTES + wave * (CH_WBUF * 16);

  // input-tile DMA plan: slot s = e*256 + tid; recomputed at every stage
  // (a few dozen integer operations, 8 times per workgroup) rather than kept in registers across the tap loops
  const uint4* xs4 = reinterpret_cast<const uint4*>(xs);
  auto in_src = [&](int e, int t) {
    const int s = e * 256 + t;
    const int sc = s < SH::IN_SLOTS ? s : 0;
    const int p = sc / SH::PIECE, r = sc % SH::PIECE;
    const int cg = r / (SH::HALO * CH_PITCH), r2 = r % (SH::HALO * CH_PITCH);
    const int yy = r2 / CH_PITCH;
    int xx = r2 % CH_PITCH;
    xx = xx < CH_HCOLS ? xx : 0;                    // pad slots re-read a valid pixel
    return (int)(((int64_t)(p * 16 + cg) * P + ty0 + yy) * P + tx0 + xx);
  };
  const uint4* wq4 = reinterpret_cast<const uint4*>(wq) + (int64_t)split * 8 * 5 * 4 * CH_WCOL +
                     (int64_t)wave * CH_WCOL + lane;
  const int b_base = (kk * SH::HALO + prow) * CH_PITCH + px;      // + r * CH_PITCH + dx (+ piece stride)
  const int a_off = kk * 32 + li;

  ch_f32x16 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // pass h = 2 * (g * 5 + dx) + odd: stream its 3 (even) or 2 (odd) tap slots into this wave's buffer
  auto wdma = [&](int h) {
    const int odd = h & 1;
    const uint4* src = wq4 + (int64_t)(h >> 1) * 4 * CH_WCOL + odd * 3 * CH_WSLOT;
    if (odd) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ch_dma16(src + 64 * i, w_lds_addr + 1024 * i);
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) ch_dma16(src + 64 * i, w_lds_addr + 1024 * i);
    }
  };
  wdma(0);

  for (int g = 0; g < 8; ++g) {
    __syncthreads();                                   // every wave is done with the previous tile
    {
      const uint4* src = xs4 + (int64_t)g * 2 * PP;
      const unsigned dst = lds_base + (wave * 64) * 16;
      int t = tid;
      asm volatile("" : "+v"(t));                    // opaque per stage: the plan is not hoisted out of the loop
#pragma unroll
      for (int e = 0; e < SH::IN_SLOTS / 256; ++e) ch_dma16(src + in_src(e, t), dst + e * 4096);
      if (wave * 64 < SH::IN_SLOTS % 256) ch_dma16(src + in_src(SH::IN_SLOTS / 256, t), dst + (SH::IN_SLOTS / 256) * 4096);
    }
    ch_wait_vm<0>();
    __syncthreads();                                   // the tile of every wave has landed
#pragma unroll 1
    for (int dx = 0; dx < 5; ++dx) {
      const uint4* il = in_l + b_base + dx;
      const int h0 = 2 * (g * 5 + dx);
      {
        // this pass's weights: issued one pass ago (at the head of a stage everything was drained with the tile)
        ch_wait_vm<0>();
        uint4 a[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a[t][0] = w_l[t * CH_WSLOT + a_off];
          a[t][1] = w_l[t * CH_WSLOT + 64 + a_off];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wdma(h0 + 1);                                  // the buffer is free: stream the next pass
        ch_column_pass<NQ, 3, 0>(a, il, acc);
      }
      {
        ch_wait_vm<0>();
        uint4 a[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t][0] = w_l[t * CH_WSLOT + a_off];
          a[t][1] = w_l[t * CH_WSLOT + 64 + a_off];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h0 + 2 < 80) wdma(h0 + 2);
        ch_column_pass<NQ, 2, 1>(a, il, acc);
      }
    }
  }

  // epilogue: undo the weight scale, bias, 4-way max, store, per-channel partial sums per 8-row half tile
  const float* bsp = bias + split * 128 + wave * 32;
  const int64_t LL = (int64_t)L * L;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int cl = 8 * g4 + 4 * kk;
    const int gch = split * 32 + wave * 8 + 2 * g4 + kk;
    const float b0 = bsp[cl], b1 = bsp[cl + 1], b2 = bsp[cl + 2], b3 = bsp[cl + 3];
#pragma unroll
    for (int hq = 0; hq < NQ / 4; ++hq) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 4 * hq; q < 4 * hq + 4; ++q) {
        float v = acc[q][4 * g4] * inv_scale + b0;
        v = fmaxf(v, acc[q][4 * g4 + 1] * inv_scale + b1);
        v = fmaxf(v, acc[q][4 * g4 + 2] * inv_scale + b2);
        v = fmaxf(v, acc[q][4 * g4 + 3] * inv_scale + b3);
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
#endif  // __HIPCC__ && CONV_F16_KERNELS

}  // namespace dmp
