// Sandbox (GPU box): row-reuse variant of the split-f16 5x5 convolution, checked against a float64 CPU
// convolution at a small L and timed against the product kernel (conv_f16.h) at L = 300.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_conv_rr.hip -o tools/_bin/ubench_conv_rr
//
// Idea: with the 32 pixels of an MFMA N axis laid out as 2 rows x 16 columns, the B fragment of row
// pair r (rows r, r+1 of the 20-row halo tile) at column offset dx serves every (accumulator q, tap row
// dy) with 2q + dy = r.  For one dx the wave loads 19 row-pair fragments instead of 8 x 5 = 40, keeps
// the five weight fragments of the taps (0..4, dx) in registers, and issues the same 120 MFMA triples.
#define CONV_F16_KERNELS
#include "../dmpfold2_amd/csrc/conv_f16.h"
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cmath>
#include <vector>

namespace dmp {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); }
int hip_fail(hipError_t e, const char* what, const char*, int line) {
  printf("HIP error %s (%s) line %d\n", hipGetErrorString(e), what, line);
  return -2;
}

constexpr int RR_WGROUP = 5 * CH_WSLOT;                                // 640 slots = 10 KB per wave: taps (0..4, dx)
constexpr int RR_LDS_BYTES = CH_IN_BYTES + 4 * RR_WGROUP * 16;         // 71680
constexpr int EO_WBUF = 3 * CH_WSLOT;                                  // even/odd variant: 3 tap slots = 6 KB per wave
constexpr int EO_LDS_BYTES = CH_IN_BYTES + 4 * EO_WBUF * 16;           // 55296

// w: [512][128][5][5] -> [split 4][g 8][dx 5][wave 4][dy 5][piece 2][cg 2][m 32][8] f16 pieces of scale * w
inline std::vector<uint16_t> pack_conv_weights_rr(const float* w, float scale, bool eo = false) {
  const int order_eo[5] = {0, 2, 4, 1, 3};
  std::vector<uint16_t> q((size_t)4 * 8 * 25 * 4 * 2 * 2 * 32 * 8);
  for (int split = 0; split < 4; ++split)
    for (int g = 0; g < 8; ++g)
      for (int dx = 0; dx < 5; ++dx)
        for (int wave = 0; wave < 4; ++wave)
          for (int dyi = 0; dyi < 5; ++dyi)
            for (int cg = 0; cg < 2; ++cg)
              for (int m = 0; m < 32; ++m)
                for (int e = 0; e < 8; ++e) {
                  const int oc = split * 128 + wave * 32 + m, ic = g * 16 + cg * 8 + e;
                  const int dy = dyi, tapy = eo ? order_eo[dyi] : dyi;     // slot dyi holds tap row tapy
                  uint16_t p2[2];
                  split2_f16(scale * w[((size_t)oc * 128 + ic) * 25 + tapy * 5 + dx], p2);
                  for (int p = 0; p < 2; ++p)
                    q[(((((((((size_t)split * 8 + g) * 5 + dx) * 4 + wave) * 5 + dy) * 2 + p) * 2 + cg) * 32 + m) * 8) + e] = p2[p];
                }
  return q;
}

// lane -> pixel of the 2 x 16 fragment: ds_read_b128 is served in the lane groups {0-3,12-15,20-27} and
// {4-11,16-19,28-31}; each group reads one row's 16 consecutive 16-byte slots = all 64 banks once
__device__ __forceinline__ void rr_lane_pixel(int li, int& row, int& x) {
  const bool a = li < 4 || (li >= 12 && li < 16) || (li >= 20 && li < 28);
  row = a ? 0 : 1;
  if (a) x = li < 4 ? li : (li < 16 ? li - 8 : li - 12);
  else x = li < 12 ? li - 4 : (li < 20 ? li - 8 : li - 16);
}

__global__ __launch_bounds__(256, 2) void conv5x5_f16x3_rr_kernel(const uint16_t* __restrict__ xs,
                                                                    const uint16_t* __restrict__ wq,
                                                                    const float* __restrict__ bias, float inv_scale,
                                                                    int L, int P, int tiles, int nwork,
                                                                    float* __restrict__ u, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int ntiles = tiles * tiles, tper = (ntiles + 3) >> 2;
  const int tile = (xcd >> 1) * tper + (slot >> 1);
  const int split = 2 * (xcd & 1) + (slot & 1);
  if ((slot >> 1) >= tper || tile >= ntiles) return;
  const int ty0 = (tile / tiles) * CONV_TILE, tx0 = (tile % tiles) * CONV_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int64_t PP = (int64_t)P * P;

  const uint4* in_l = reinterpret_cast<const uint4*>(ch_smem);
  const uint4* w_l = reinterpret_cast<const uint4*>(ch_smem + CH_IN_BYTES) + wave * RR_WGROUP;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ch_smem;
  const unsigned w_lds_addr = lds_base + CH_IN_BYTES + wave * (RR_WGROUP * 16);

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
    xx = xx < CH_HALO ? xx : 0;
    in_src[e] = (int)(((int64_t)(p * 16 + cg) * P + ty0 + yy) * P + tx0 + xx);
  }
  const uint4* wq4 = reinterpret_cast<const uint4*>(wq) + (int64_t)split * 8 * 5 * 4 * RR_WGROUP +
                     (int64_t)wave * RR_WGROUP + lane;
  int prow, px;
  rr_lane_pixel(li, prow, px);
  const int b_base = (kk * CH_HALO + prow) * CH_PITCH + px;      // + r * CH_PITCH + dx (+ piece stride)
  const int a_off = kk * 32 + li;

  ch_f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  auto wdma = [&](int grp) {            // weights of tap column grp = g * 5 + dx -> this wave's 10 KB buffer
    const uint4* src = wq4 + (int64_t)grp * 4 * RR_WGROUP;
#pragma unroll
    for (int i = 0; i < 10; ++i) ch_dma16(src + 64 * i, w_lds_addr + 1024 * i);
  };
  wdma(0);

  for (int g = 0; g < 8; ++g) {
    __syncthreads();                                   // every wave is done with the previous tile
    {
      const uint4* src = xs4 + (int64_t)g * 2 * PP;
      const unsigned dst = lds_base + (wave * 64) * 16;
#pragma unroll
      for (int e = 0; e < 7; ++e) ch_dma16(src + in_src[e], dst + e * 4096);
      if (wave < 2) ch_dma16(src + in_src[7], dst + 7 * 4096);
    }
    ch_wait_vm<0>();
    __syncthreads();                                   // the tile of every wave has landed
#pragma unroll 1
    for (int dx = 0; dx < 5; ++dx) {
      ch_wait_vm<0>();                                 // this column's weights (issued one column ago)
      uint4 a[5][2];
#pragma unroll
      for (int dy = 0; dy < 5; ++dy) {
        a[dy][0] = w_l[dy * CH_WSLOT + a_off];
        a[dy][1] = w_l[dy * CH_WSLOT + 64 + a_off];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (g * 5 + dx + 1 < 40) wdma(g * 5 + dx + 1);   // the buffer is free: stream the next column
      const uint4* il = in_l + b_base + dx;
      uint4 bn0 = il[0], bn1 = il[2 * CH_HALO * CH_PITCH];
#pragma unroll
      for (int r = 0; r < 19; ++r) {
        const uint4 b0 = bn0, b1 = bn1;
        if (r + 1 < 19) {
          bn0 = il[(r + 1) * CH_PITCH];
          bn1 = il[(r + 1) * CH_PITCH + 2 * CH_HALO * CH_PITCH];
        }
        // small terms first (w0 x1, w1 x0), then w0 x0; the accumulators of the row's taps alternate
#pragma unroll
        for (int dy = r & 1; dy < 5; dy += 2) {
          const int q = (r - dy) / 2;
          if (r - dy >= 0 && q < 8) acc[q] = ch_mfma(a[dy][0], b1, acc[q]);
        }
#pragma unroll
        for (int dy = r & 1; dy < 5; dy += 2) {
          const int q = (r - dy) / 2;
          if (r - dy >= 0 && q < 8) acc[q] = ch_mfma(a[dy][1], b0, acc[q]);
        }
#pragma unroll
        for (int dy = r & 1; dy < 5; dy += 2) {
          const int q = (r - dy) / 2;
          if (r - dy >= 0 && q < 8) acc[q] = ch_mfma(a[dy][0], b0, acc[q]);
        }
      }
    }
  }

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
      part[((int64_t)tile * CW + gch) * 2 + 0] = (double)s1;
      part[((int64_t)tile * CW + gch) * 2 + 1] = (double)s2;
    }
  }
}

// ---- even/odd half-column variant: 24 + 16 weight registers, 6 KB weight buffer per wave ----
__global__ __launch_bounds__(256, 2) void conv5x5_f16x3_eo_kernel(const uint16_t* __restrict__ xs,
                                                                    const uint16_t* __restrict__ wq,
                                                                    const float* __restrict__ bias, float inv_scale,
                                                                    int L, int P, int tiles, int nwork,
                                                                    float* __restrict__ u, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ch_smem[];
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int ntiles = tiles * tiles, tper = (ntiles + 3) >> 2;
  const int tile = (xcd >> 1) * tper + (slot >> 1);
  const int split = 2 * (xcd & 1) + (slot & 1);
  if ((slot >> 1) >= tper || tile >= ntiles) return;
  const int ty0 = (tile / tiles) * CONV_TILE, tx0 = (tile % tiles) * CONV_TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int64_t PP = (int64_t)P * P;

  const uint4* in_l = reinterpret_cast<const uint4*>(ch_smem);
  const uint4* w_l = reinterpret_cast<const uint4*>(ch_smem + CH_IN_BYTES) + wave * EO_WBUF;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ch_smem;
  const unsigned w_lds_addr = lds_base + CH_IN_BYTES + wave * (EO_WBUF * 16);

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
    xx = xx < CH_HALO ? xx : 0;
    in_src[e] = (int)(((int64_t)(p * 16 + cg) * P + ty0 + yy) * P + tx0 + xx);
  }
  const uint4* wq4 = reinterpret_cast<const uint4*>(wq) + (int64_t)split * 8 * 5 * 4 * RR_WGROUP +
                     (int64_t)wave * RR_WGROUP + lane;
  int prow, px;
  rr_lane_pixel(li, prow, px);
  const int b_base = (kk * CH_HALO + prow) * CH_PITCH + px;      // + r * CH_PITCH + dx (+ piece stride)
  const int a_off = kk * 32 + li;

  ch_f32x16 acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

  // half columns: pass h = 2 * (g * 5 + dx) + odd; even pass = taps dy 0, 2, 4 (3 slots), odd pass = dy 1, 3.
  // The packed order [dx][wave][dy 0 2 4 1 3] makes both passes contiguous.
  auto wdma = [&](int h) {
    const int grp = h >> 1, odd = h & 1;
    const uint4* src = wq4 + (int64_t)grp * 4 * RR_WGROUP + odd * 3 * CH_WSLOT;
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
#pragma unroll
      for (int e = 0; e < 7; ++e) ch_dma16(src + in_src[e], dst + e * 4096);
      if (wave < 2) ch_dma16(src + in_src[7], dst + 7 * 4096);
    }
    ch_wait_vm<0>();
    __syncthreads();                                   // the tile of every wave has landed
#pragma unroll 1
    for (int dx = 0; dx < 5; ++dx) {
      const uint4* il = in_l + b_base + dx;
      const int h0 = 2 * (g * 5 + dx);
      {   // even pass: taps dy = 0, 2, 4 against the row pairs r = 0, 2, ..., 18
        ch_wait_vm<0>();
        uint4 a[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          a[t][0] = w_l[t * CH_WSLOT + a_off];
          a[t][1] = w_l[t * CH_WSLOT + 64 + a_off];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wdma(h0 + 1);
        uint4 bn0 = il[0], bn1 = il[2 * CH_HALO * CH_PITCH];
#pragma unroll
        for (int r = 0; r < 19; r += 2) {
          const uint4 b0 = bn0, b1 = bn1;
          if (r + 2 < 19) {
            bn0 = il[(r + 2) * CH_PITCH];
            bn1 = il[(r + 2) * CH_PITCH + 2 * CH_HALO * CH_PITCH];
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int q = (r - 2 * t) / 2;
            if (r - 2 * t >= 0 && q < 8) acc[q] = ch_mfma(a[t][0], b1, acc[q]);
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int q = (r - 2 * t) / 2;
            if (r - 2 * t >= 0 && q < 8) acc[q] = ch_mfma(a[t][1], b0, acc[q]);
          }
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int q = (r - 2 * t) / 2;
            if (r - 2 * t >= 0 && q < 8) acc[q] = ch_mfma(a[t][0], b0, acc[q]);
          }
        }
      }
      {   // odd pass: taps dy = 1, 3 against the row pairs r = 1, 3, ..., 17
        ch_wait_vm<0>();
        uint4 a[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a[t][0] = w_l[t * CH_WSLOT + a_off];
          a[t][1] = w_l[t * CH_WSLOT + 64 + a_off];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (h0 + 2 < 80) wdma(h0 + 2);
        uint4 bn0 = il[CH_PITCH], bn1 = il[CH_PITCH + 2 * CH_HALO * CH_PITCH];
#pragma unroll
        for (int r = 1; r < 19; r += 2) {
          const uint4 b0 = bn0, b1 = bn1;
          if (r + 2 < 19) {
            bn0 = il[(r + 2) * CH_PITCH];
            bn1 = il[(r + 2) * CH_PITCH + 2 * CH_HALO * CH_PITCH];
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int q = (r - 1 - 2 * t) / 2;
            if (r - 1 - 2 * t >= 0 && q < 8) acc[q] = ch_mfma(a[t][0], b1, acc[q]);
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int q = (r - 1 - 2 * t) / 2;
            if (r - 1 - 2 * t >= 0 && q < 8) acc[q] = ch_mfma(a[t][1], b0, acc[q]);
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int q = (r - 1 - 2 * t) / 2;
            if (r - 1 - 2 * t >= 0 && q < 8) acc[q] = ch_mfma(a[t][0], b0, acc[q]);
          }
        }
      }
    }
  }

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
      part[((int64_t)tile * CW + gch) * 2 + 0] = (double)s1;
      part[((int64_t)tile * CW + gch) * 2 + 1] = (double)s2;
    }
  }
}
}  // namespace dmp
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace dmp;

static void cpu_ref(const std::vector<float>& x, const std::vector<float>& w, const std::vector<float>& b,
                    int L, std::vector<float>& u) {
  std::vector<double> o(512);
  for (int y = 0; y < L; ++y)
    for (int xx = 0; xx < L; ++xx) {
      for (int oc = 0; oc < 512; ++oc) {
        double s = b[oc];
        for (int c = 0; c < 128; ++c)
          for (int dy = 0; dy < 5; ++dy) {
            const int yy = y + dy - 2;
            if (yy < 0 || yy >= L) continue;
            for (int dx = 0; dx < 5; ++dx) {
              const int xq = xx + dx - 2;
              if (xq < 0 || xq >= L) continue;
              s += (double)w[((size_t)oc * 128 + c) * 25 + dy * 5 + dx] * (double)x[((size_t)c * L + yy) * L + xq];
            }
          }
        o[oc] = s;
      }
      for (int g = 0; g < 128; ++g) {
        double m = o[4 * g];
        for (int q = 1; q < 4; ++q) m = o[4 * g + q] > m ? o[4 * g + q] : m;
        u[((size_t)g * L + y) * L + xx] = (float)m;
      }
    }
}

int main(int argc, char** argv) {
  const int Lt = argc > 1 ? atoi(argv[1]) : 24;
  const int Lb = argc > 2 ? atoi(argv[2]) : 300;
  std::vector<float> w((size_t)512 * 128 * 25), b(512);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };
  for (auto& v : w) v = rnd() * 0.04f;
  for (auto& v : b) v = rnd() * 0.1f;
  const char* ub_data = getenv("UB_DATA");
  auto shape = [&](std::vector<float>& a) {
    if (!ub_data) return;
    for (auto& v : a) v = ub_data[0] == 'z' ? 0.f : (float)(_Float16)v;
  };
  shape(w);
  const float scale = conv_weight_scale_f16(w.data(), w.size());
  std::vector<uint16_t> wq = pack_conv_weights_f16(w.data(), scale);
  std::vector<uint16_t> wr = pack_conv_weights_rr(w.data(), scale);
  std::vector<uint16_t> we = pack_conv_weights_rr(w.data(), scale, true);
  uint16_t *d_wq, *d_wr, *d_we; float* d_b;
  CK(hipMalloc(&d_wq, wq.size() * 2)); CK(hipMalloc(&d_wr, wr.size() * 2)); CK(hipMalloc(&d_b, 512 * 4));
  CK(hipMalloc(&d_we, we.size() * 2));
  CK(hipMemcpy(d_we, we.data(), we.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)conv5x5_f16x3_eo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EO_LDS_BYTES));
  const bool use_eo = getenv("RR_EO") && getenv("RR_EO")[0] == '1';
  CK(hipMemcpy(d_wq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wr, wr.data(), wr.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, b.data(), 512 * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)conv5x5_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                         CONVH_LDS_BYTES));
  CK(hipFuncSetAttribute((const void*)conv5x5_f16x3_rr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                         RR_LDS_BYTES));
  for (int L : {Lt, 37, Lb}) {
    const int P = act_pitch(L), tiles = act_tiles(L);
    std::vector<float> x((size_t)128 * L * L);
    for (auto& v : x) v = rnd() * 6.f;
    shape(x);
    std::vector<uint16_t> xs((size_t)2 * 16 * P * P * 8, 0);
    for (int ch = 0; ch < 128; ++ch)
      for (int y = 0; y < L; ++y)
        for (int xx = 0; xx < L; ++xx) {
          uint16_t p2[2];
          split2_f16(x[((size_t)ch * L + y) * L + xx], p2);
          for (int p = 0; p < 2; ++p)
            xs[((((size_t)p * 16 + ch / 8) * P + y + 2) * P + xx + 2) * 8 + ch % 8] = p2[p];
        }
    uint16_t* d_xs; float *d_u, *d_u2; double *d_part, *d_part2;
    CK(hipMalloc(&d_xs, xs.size() * 2)); CK(hipMalloc(&d_u, (size_t)128 * L * L * 4)); CK(hipMalloc(&d_u2, (size_t)128 * L * L * 4));
    CK(hipMalloc(&d_part, (size_t)tiles * tiles * 128 * 2 * 8)); CK(hipMalloc(&d_part2, (size_t)tiles * tiles * 128 * 2 * 8));
    CK(hipMemcpy(d_xs, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
    const int nwork = tiles * tiles * 4, grid = conv_f16_grid(tiles);
    auto launch_old = [&]() {
      hipLaunchKernelGGL(conv5x5_f16x3_kernel, dim3(grid), dim3(256), CONVH_LDS_BYTES, 0, d_xs, d_wq, d_b,
                         1.0f / scale, L, P, tiles, nwork, d_u, d_part);
    };
    auto launch_new = [&]() {
      if (use_eo)
        hipLaunchKernelGGL(conv5x5_f16x3_eo_kernel, dim3(grid), dim3(256), EO_LDS_BYTES, 0, d_xs, d_we, d_b,
                           1.0f / scale, L, P, tiles, nwork, d_u2, d_part2);
      else
        hipLaunchKernelGGL(conv5x5_f16x3_rr_kernel, dim3(grid), dim3(256), RR_LDS_BYTES, 0, d_xs, d_wr, d_b,
                           1.0f / scale, L, P, tiles, nwork, d_u2, d_part2);
    };
    launch_old(); launch_new();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    std::vector<float> u((size_t)128 * L * L), u2(u.size());
    CK(hipMemcpy(u.data(), d_u, u.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(u2.data(), d_u2, u.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> pa((size_t)tiles * tiles * 256), pb(pa.size());
    CK(hipMemcpy(pa.data(), d_part, pa.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(pb.data(), d_part2, pb.size() * 8, hipMemcpyDeviceToHost));
    double md = 0, mr = 0, mp = 0;
    for (size_t i = 0; i < u.size(); ++i) { md = fmax(md, fabs((double)u[i] - u2[i])); mr = fmax(mr, fabs(u[i])); }
    for (size_t i = 0; i < pa.size(); ++i) mp = fmax(mp, fabs(pa[i] - pb[i]) / fmax(1.0, fabs(pa[i])));
    printf("L=%d  row-reuse vs product kernel: max|du| = %.3e (scale %.3e), partial sums rel %.2e\n", L, md, mr, mp);
    if (L == Lt) {
      std::vector<float> ref(u.size());
      cpu_ref(x, w, b, L, ref);
      double m1 = 0, m2 = 0;
      for (size_t i = 0; i < u.size(); ++i) { m1 = fmax(m1, fabs((double)u[i] - ref[i])); m2 = fmax(m2, fabs((double)u2[i] - ref[i])); }
      printf("L=%d  vs float64: product %.3e, row-reuse %.3e\n", L, m1, m2);
    }
    if (L == Lb) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int which = 0; which < 2; ++which) {
        float best = 1e9f, tot = 0;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(e0));
          for (int i = 0; i < 10; ++i) { if (which) launch_new(); else launch_old(); }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          best = fminf(best, ms / 10); tot += ms / 10;
        }
        const double flop = 2.0 * 128 * 512 * 25 * L * L;
        printf("L=%d  %s: %.3f ms avg, %.3f ms best -> %.1f TFLOP/s float32-equivalent\n", L,
               which ? (use_eo ? "even/odd " : "row-reuse") : "product  ", tot / 5, best, flop / (best * 1e-3) / 1e12);
      }
    }
    CK(hipFree(d_xs)); CK(hipFree(d_u)); CK(hipFree(d_u2)); CK(hipFree(d_part)); CK(hipFree(d_part2));
  }
  return 0;
}
