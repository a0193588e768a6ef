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
// Range: activations are used unscaled, |x| must stay below 65504 (f16 max); the producer kernels
// raise the context's fault flag if they ever see |x| >= 60000 (InstanceNorm keeps the trunk at
// O(1)..O(100)).  Low pieces of |x| < 0.125 are f16 subnormals: their absolute error <= 3e-8 is far
// below the float32 accumulation error of the sums they enter.
//
// Layouts and workgroup structure are those of conv_bf16.h with 2 pieces instead of 3:
//   activations  xs[piece 2][c/8 16][P][P][8] f16;   weights wq[split 4][cgrp 8][tap 25][wave 4][piece 2][cg 2][m 32][8] f16
//   workgroup = 4 waves x 32 conv channels x one 16x16 pixel tile; 8 input stages of 16 channels;
//   per-wave private weight ring (4 slots of 2 KB, LDS-DMA two taps ahead, counted vmcnt).
#pragma once
#include "common.h"
#include <cmath>
#include <vector>

namespace dmp {

typedef float ch_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ch_f16x8 __attribute__((ext_vector_type(8)));

constexpr int CH_HALO = 20, CH_PITCH = 24;
constexpr int CH_IN_SLOTS = 2 * 2 * CH_HALO * CH_PITCH;               // 1920 16-byte slots
constexpr int CH_IN_BYTES = CH_IN_SLOTS * 16;                         // 30720
constexpr int CH_WSLOT = 2 * 2 * 32;                                  // 128 slots per wave and tap
#ifndef CH_RING_SLOTS
#define CH_RING_SLOTS 4      // per-wave weight ring: 4 slots = DMA two taps ahead, 2 slots = one tap
#endif
#ifndef CH_OCC
#define CH_OCC 2             // workgroups per CU the register budget is compiled for
#endif
#ifndef CH_MAP_PAIR
#define CH_MAP_PAIR 1        // 1: each XCD handles two of the four channel splits (see the kernel);
#endif                       //    same time, fabric reads per launch 361 -> 257 MB (tools/pmc_fetch.sh);
                             //    one split per XCD (a tile's input read by four XCDs) measured 290 MB
#ifndef CH_VGPR_CAP
#define CH_VGPR_CAP          // e.g. __attribute__((amdgpu_waves_per_eu(3, 3))) caps the kernel at 168 VGPRs
#endif
// number of workgroups to launch for tiles x tiles pixel tiles
inline int conv_f16_grid(int tiles) {
  const int ntiles = tiles * tiles;
  return CH_MAP_PAIR ? 8 * 2 * ((ntiles + 3) / 4) : (ntiles * 4 + 7) / 8 * 8;
}
constexpr int CH_RING = CH_RING_SLOTS;
constexpr int CH_DIST = CH_RING >= 4 ? 2 : 1;
constexpr int CONVH_LDS_BYTES = CH_IN_BYTES + 4 * CH_RING * CH_WSLOT * 16;   // 63488 (ring 4) / 47104 (ring 2)

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

// w: [512][128][5][5] float32 -> packed f16 pieces of scale * w
inline std::vector<uint16_t> pack_conv_weights_f16(const float* w, float scale) {
  std::vector<uint16_t> q((size_t)4 * 8 * 25 * 4 * 2 * 2 * 32 * 8);
  for (int split = 0; split < 4; ++split)
    for (int g = 0; g < 8; ++g)
      for (int tap = 0; tap < 25; ++tap)
        for (int wave = 0; wave < 4; ++wave)
          for (int cg = 0; cg < 2; ++cg)
            for (int m = 0; m < 32; ++m)
              for (int e = 0; e < 8; ++e) {
                const int oc = split * 128 + wave * 32 + m, ic = g * 16 + cg * 8 + e;
                uint16_t p2[2];
                split2_f16(scale * w[((size_t)oc * 128 + ic) * 25 + tap], p2);
                for (int p = 0; p < 2; ++p)
                  q[((((((((size_t)split * 8 + g) * 25 + tap) * 4 + wave) * 2 + p) * 2 + cg) * 32 + m) * 8) + e] = p2[p];
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

// grid: conv_f16_grid(tiles) blocks (XCD-aware map)   block: 256   dynamic LDS: CONVH_LDS_BYTES
__global__ __launch_bounds__(256, CH_OCC) CH_VGPR_CAP void conv5x5_f16x3_kernel(const uint16_t* __restrict__ xs,
                                                               const uint16_t* __restrict__ wq,
                                                               const float* __restrict__ bias, float inv_scale,
                                                               int L, int P, int tiles, int nwork,
                                                               float* __restrict__ u, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  const int id = blockIdx.x;
#if CH_MAP_PAIR
  // XCD x works on the channel splits {0,1} (x even) or {2,3} (x odd) only: its share of the weight
  // pieces is 3.3 MB, which fits the 4 MB L2; a tile's input is then read by two XCDs
  const int xcd = id & 7, slot = id >> 3;
  const int ntiles = tiles * tiles, tper = (ntiles + 3) >> 2;
  const int tile = (xcd >> 1) * tper + (slot >> 1);
  const int split = 2 * (xcd & 1) + (slot & 1);
  if ((slot >> 1) >= tper || tile >= ntiles) return;
#else
  const int per = gridDim.x >> 3;
  const int work = (id & 7) * per + (id >> 3);
  if (work >= nwork) return;
  const int tile = work >> 2, split = work & 3;
#endif
  const int ty0 = (tile / tiles) * CONV_TILE, tx0 = (tile % tiles) * CONV_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int64_t PP = (int64_t)P * P;

  const uint4* in_l = reinterpret_cast<const uint4*>(ch_smem);
  const uint4* w_l = reinterpret_cast<const uint4*>(ch_smem + CH_IN_BYTES) + wave * (CH_RING * CH_WSLOT);
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ch_smem;
  const unsigned w_lds_addr = lds_base + CH_IN_BYTES + wave * (CH_RING * CH_WSLOT * 16);

  // input-tile DMA plan: slot s = e*256 + tid, e = 0..7 (1920 slots = 7.5 x 256)
  const uint4* xs4 = reinterpret_cast<const uint4*>(xs);
  int in_src[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int s = e * 256 + tid;
    const int sc = s < CH_IN_SLOTS ? s : 0;
    const int p = sc / 960, r = sc % 960;
    const int cg = r / 480, r2 = r % 480;
    const int yy = r2 / CH_PITCH;
    int xx = r2 % CH_PITCH;
    xx = xx < CH_HALO ? xx : 0;                     // pad slots re-read a valid pixel
#ifdef CH_EXP_SAMETILE      // traffic experiment only: every workgroup reads the first tile's halo
    in_src[e] = (int)(((int64_t)(p * 16 + cg) * P + yy) * P + xx);
#else
    in_src[e] = (int)(((int64_t)(p * 16 + cg) * P + ty0 + yy) * P + tx0 + xx);
#endif
  }
#ifdef CH_EXP_SAMEW         // traffic experiment only: every workgroup streams the first split's weights
  const uint4* wq4 = reinterpret_cast<const uint4*>(wq) +
#else
  const uint4* wq4 = reinterpret_cast<const uint4*>(wq) + (int64_t)split * 8 * 25 * 4 * CH_WSLOT +
#endif
                     (int64_t)wave * CH_WSLOT + lane;

  int b_off[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int y = (q >> 1) * 4 + (li >> 3), x = (q & 1) * 8 + (li & 7);
    b_off[q] = (kk * CH_HALO + y) * CH_PITCH + x;
  }
  const int a_off = kk * 32 + li;

  ch_f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  auto wdma = [&](int g, int tap) {
    const uint4* src = wq4 + ((int64_t)g * 25 + tap) * 4 * CH_WSLOT;
    const unsigned dst = w_lds_addr + (tap & (CH_RING - 1)) * (CH_WSLOT * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier LDS reads of the slot are complete
    ch_dma16(src, dst);
    ch_dma16(src + 64, dst + 1024);
  };

  for (int g = 0; g < 8; ++g) {
    __syncthreads();                                   // every wave is done with the previous tile
    {
      const uint4* src = xs4 + (int64_t)g * 2 * PP;
      const unsigned dst = lds_base + (wave * 64) * 16;
#pragma unroll
      for (int e = 0; e < 7; ++e) ch_dma16(src + in_src[e], dst + e * 4096);
      if (wave < 2) ch_dma16(src + in_src[7], dst + 7 * 4096);
    }
    wdma(g, 0);
    if (CH_DIST == 2) wdma(g, 1);
    ch_wait_vm<0>();
    __syncthreads();                                   // the tile of every wave has landed
#pragma unroll 1
    for (int dy = 0; dy < 5; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int tap = dy * 5 + dx;
        // weights two taps ahead; this tap's two pieces must have landed (2 instructions per tap)
        if (tap + CH_DIST < 25) { wdma(g, tap + CH_DIST); ch_wait_vm<2 * CH_DIST>(); }
        else if (CH_DIST == 2 && tap + 1 < 25) ch_wait_vm<2>();
        else ch_wait_vm<0>();
        const uint4* wl = w_l + (tap & (CH_RING - 1)) * CH_WSLOT + a_off;
        const uint4 a0 = wl[0], a1 = wl[64];
        const uint4* il = in_l + dy * CH_PITCH + dx;
#pragma unroll
        for (int qp = 0; qp < 4; ++qp) {
          uint4 b[2][2];
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[h][p] = il[p * (2 * CH_HALO * CH_PITCH) + b_off[2 * qp + h]];
          // small terms first; the two accumulators alternate
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = ch_mfma(a0, b[h][1], acc[2 * qp + h]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = ch_mfma(a1, b[h][0], acc[2 * qp + h]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[2 * qp + h] = ch_mfma(a0, b[h][0], acc[2 * qp + h]);
        }
      }
    }
  }

  // epilogue: undo the weight scale, bias, 4-way max, store, per-channel partial sums
  const float* bsp = bias + split * 128 + wave * 32;
  const int64_t LL = (int64_t)L * L;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int cl = 8 * g4 + 4 * kk;
    const int gch = split * 32 + wave * 8 + 2 * g4 + kk;
    const float b0 = bsp[cl], b1 = bsp[cl + 1], b2 = bsp[cl + 2], b3 = bsp[cl + 3];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = acc[q][4 * g4] * inv_scale + b0;
      v = fmaxf(v, acc[q][4 * g4 + 1] * inv_scale + b1);
      v = fmaxf(v, acc[q][4 * g4 + 2] * inv_scale + b2);
      v = fmaxf(v, acc[q][4 * g4 + 3] * inv_scale + b3);
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
#endif  // __HIPCC__ && CONV_F16_KERNELS

}  // namespace dmp
