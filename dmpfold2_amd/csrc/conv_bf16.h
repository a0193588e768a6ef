// conv 5x5 (128 -> 512) + bias + 4-way maxout with float32 semantics on the bf16 matrix cores.
//
// Every float32 operand is split EXACTLY into three bf16 pieces (x = x0 + x1 + x2: 3 x 8 significand
// bits = the 24 of a float32).  Of the nine piece products w_i * x_j the six with i + j <= 2 are
// accumulated in float32 by v_mfma_f32_32x32x16_bf16; the three dropped ones are below 2^-24 of the
// product, i.e. below the rounding of a float32 multiply.  The result equals the float32 convolution
// to float32 rounding error (measured: max error vs float64 2e-6, the plain f32 kernel 5e-6) at
// 16/6 = 2.67x the f32 MFMA rate.
//
// Layouts
//   activations  xs[piece 3][c/8 16][P][P][8]   bf16, zero border (same geometry as the f32 planes);
//                a pixel's 8 channels are one 16-byte LDS/MFMA operand
//   weights      wq[split 4][cgrp 8][tap 25][wave 4][piece 3][cg 2][m 32][8]   bf16
//                (conv channel = split*128 + wave*32 + m, input channel = cgrp*16 + cg*8 + e):
//                one wave's operands for one tap are 3 KB contiguous = three 1 KB LDS-DMA pieces
// Workgroup = 4 waves x (32 conv channels each) x one 16x16 pixel tile (eight 4x8 patches = MFMA N
// blocks); K = 16 input channels x 25 taps per input stage, 8 stages.  The input halo tile
// (3 pieces x 2 x 20 x 24 slots) is shared by the waves (2 barriers per stage); each wave streams
// its own weights tap by tap through a private 2-slot LDS ring with LDS-DMA issued one tap ahead
// (inline asm, counted vmcnt) - no workgroup barrier inside the 25-tap loop.
#pragma once
#include "common.h"
#include <vector>

namespace dmp {

typedef float cq_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cq_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int CQ_HALO = 20, CQ_PITCH = 24;
constexpr int CQ_IN_SLOTS = 3 * 2 * CQ_HALO * CQ_PITCH;               // 2880 16-byte slots
constexpr int CQ_IN_BYTES = CQ_IN_SLOTS * 16;                         // 46080
constexpr int CQ_WSLOT = 3 * 2 * 32;                                  // 192 slots per wave and tap
constexpr int CQ_RING = 2;
constexpr int CONVQ_LDS_BYTES = CQ_IN_BYTES + 4 * CQ_RING * CQ_WSLOT * 16;   // 70656

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

// w: [512][128][5][5] float32 -> packed bf16 pieces
inline std::vector<uint16_t> pack_conv_weights_bf16(const float* w) {
  std::vector<uint16_t> q((size_t)4 * 8 * 25 * 4 * 3 * 2 * 32 * 8);
  for (int split = 0; split < 4; ++split)
    for (int g = 0; g < 8; ++g)
      for (int tap = 0; tap < 25; ++tap)
        for (int wave = 0; wave < 4; ++wave)
          for (int cg = 0; cg < 2; ++cg)
            for (int m = 0; m < 32; ++m)
              for (int e = 0; e < 8; ++e) {
                const int oc = split * 128 + wave * 32 + m, ic = g * 16 + cg * 8 + e;
                uint16_t p3[3];
                split3_bf16(w[((size_t)oc * 128 + ic) * 25 + tap], p3);
                for (int p = 0; p < 3; ++p)
                  q[((((((((size_t)split * 8 + g) * 25 + tap) * 4 + wave) * 3 + p) * 2 + cg) * 32 + m) * 8) + e] = p3[p];
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

// grid: round_up(tiles*tiles*4, 8) blocks, XCD-aware map as the f32 kernel   block: 256
// dynamic LDS: CONVQ_LDS_BYTES
// Scheduling note: hand software-pipelining of the LDS fragment reads (one patch pair ahead, 2-tap
// DMA distance, with and without sched_group_barrier) measured 5-9 % SLOWER than hipcc's own
// schedule at 2 waves per SIMD; PMC: MFMA pipe 78 % busy while the clock sits at 1.74 GHz - the
// kernel runs into the power limit, not into issue stalls.
__global__ __launch_bounds__(256, 2) void conv5x5_bf16x6_kernel(const uint16_t* __restrict__ xs,
                                                                const uint16_t* __restrict__ wq,
                                                                const float* __restrict__ bias, int L, int P,
                                                                int tiles, int nwork, float* __restrict__ u,
                                                                double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cq_smem[];
  const int id = blockIdx.x;
  const int per = gridDim.x >> 3;
  const int work = (id & 7) * per + (id >> 3);
  if (work >= nwork) return;
  const int tile = work >> 2, split = work & 3;
  const int ty0 = (tile / tiles) * CONV_TILE, tx0 = (tile % tiles) * CONV_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int64_t PP = (int64_t)P * P;

  const uint4* in_l = reinterpret_cast<const uint4*>(cq_smem);
  const uint4* w_l = reinterpret_cast<const uint4*>(cq_smem + CQ_IN_BYTES) + wave * (CQ_RING * CQ_WSLOT);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)cq_smem;
  const unsigned w_lds_addr = lds_base + CQ_IN_BYTES + wave * (CQ_RING * CQ_WSLOT * 16);

  // ---- input-tile DMA plan: slot s = e*256 + tid, e = 0..11 (2880 slots)
  const uint4* xs4 = reinterpret_cast<const uint4*>(xs);
  int64_t in_src[12];
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    const int s = e * 256 + tid;
    const int sc = s < CQ_IN_SLOTS ? s : 0;
    const int p = sc / 960, r = sc % 960;
    const int cg = r / 480, r2 = r % 480;
    const int yy = r2 / CQ_PITCH;
    int xx = r2 % CQ_PITCH;
    xx = xx < CQ_HALO ? xx : 0;                     // pad slots re-read a valid pixel
    in_src[e] = ((int64_t)(p * 16 + cg) * P + ty0 + yy) * P + tx0 + xx;
  }
  const uint4* wq4 = reinterpret_cast<const uint4*>(wq) + (int64_t)split * 8 * 25 * 4 * CQ_WSLOT +
                     (int64_t)wave * CQ_WSLOT + lane;

  // fragment offsets (16-byte slots)
  int b_off[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int y = (q >> 1) * 4 + (li >> 3), x = (q & 1) * 8 + (li & 7);
    b_off[q] = (kk * CQ_HALO + y) * CQ_PITCH + x;
  }
  const int a_off = kk * 32 + li;

  cq_f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  auto wdma = [&](int g, int tap, int slot) {
    const uint4* src = wq4 + ((int64_t)g * 25 + tap) * 4 * CQ_WSLOT;
    const unsigned dst = w_lds_addr + slot * (CQ_WSLOT * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier LDS reads of the slot are complete
    cq_dma16(src, dst);
    cq_dma16(src + 64, dst + 1024);
    cq_dma16(src + 128, dst + 2048);
  };

  for (int g = 0; g < 8; ++g) {
    __syncthreads();                                   // every wave is done with the previous tile
    {
      const uint4* src = xs4 + (int64_t)g * 2 * PP;
      const unsigned dst = lds_base + (wave * 64) * 16;
#pragma unroll
      for (int e = 0; e < 11; ++e) cq_dma16(src + in_src[e], dst + e * 4096);
      if (wave == 0) cq_dma16(src + in_src[11], dst + 11 * 4096);
    }
    wdma(g, 0, 0);
    cq_wait_vm<0>();
    __syncthreads();                                   // the tile of every wave has landed
#pragma unroll 1
    for (int dy = 0; dy < 5; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int tap = dy * 5 + dx;
        const int slot = tap & 1;
        if (tap + 1 < 25) {
          wdma(g, tap + 1, slot ^ 1);
          cq_wait_vm<3>();                             // this tap's three pieces are in LDS
        } else {
          cq_wait_vm<0>();
        }
        const uint4* wl = w_l + slot * CQ_WSLOT + a_off;
        const uint4 a0 = wl[0], a1 = wl[64], a2 = wl[128];
        const uint4* il = in_l + dy * CQ_PITCH + dx;
#pragma unroll
        for (int qp = 0; qp < 4; ++qp) {
          uint4 b[2][3];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[h][p] = il[p * (2 * CQ_HALO * CQ_PITCH) + b_off[2 * qp + h]];
          // smallest terms first; the two accumulators alternate
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = cq_mfma(a0, b[h][2], acc[2 * qp + h]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = cq_mfma(a1, b[h][1], acc[2 * qp + h]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = cq_mfma(a2, b[h][0], acc[2 * qp + h]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = cq_mfma(a0, b[h][1], acc[2 * qp + h]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = cq_mfma(a1, b[h][0], acc[2 * qp + h]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = cq_mfma(a0, b[h][0], acc[2 * qp + h]);
        }
      }
    }
  }

  // ---- epilogue: bias, 4-way max, store, per-channel partial sums (no cross-wave reduction:
  // a wave owns its 32 conv channels = 8 maxout channels for all 256 pixels of the tile)
  const float* bsp = bias + split * 128 + wave * 32;
  const int64_t LL = (int64_t)L * L;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int cl = 8 * g4 + 4 * kk;                        // first conv channel of the group in the wave
    const int gch = split * 32 + wave * 8 + 2 * g4 + kk;    // maxout channel
    const float b0 = bsp[cl], b1 = bsp[cl + 1], b2 = bsp[cl + 2], b3 = bsp[cl + 3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = acc[q][4 * g4] + b0;
      v = fmaxf(v, acc[q][4 * g4 + 1] + b1);
      v = fmaxf(v, acc[q][4 * g4 + 2] + b2);
      v = fmaxf(v, acc[q][4 * g4 + 3] + b3);
      const int y = ty0 + (q >> 1) * 4 + (li >> 3), x = tx0 + (q & 1) * 8 + (li & 7);
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
      part[((int64_t)tile * CW + gch) * 2 + 0] = (double)s1;
      part[((int64_t)tile * CW + gch) * 2 + 1] = (double)s2;
    }
  }
}
#endif  // __HIPCC__ && CONV_BF16_KERNELS

}  // namespace dmp
