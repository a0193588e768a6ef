// GPU box: what the f16 matrix cores sustain under the chip's power cap, by operand data.
// Register-resident operands, 8 independent accumulators per wave, no memory traffic in the loop:
// the only variable is what the multipliers toggle.  Runs each case for ~2 s and prints TFLOP/s per
// 0.25 s window (the first windows are faster: power management settles after ~0.5 s).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma_power.hip -o tools/_bin/ubench_mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// -DUSE_BF16: the same loop on v_mfma_f32_32x32x16_bf16 with bf16 operand data (round 6: the ceiling of the exact
// three-piece bf16 convolution, conv_bf16.h)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ ab, int iters, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(f16x8, ab[(i * 64 + lane)]);
    b[i] = __builtin_bit_cast(f16x8, ab[((4 + i) * 64 + lane)]);
  }
#ifdef USE_16X16
  // -DUSE_16X16: v_mfma_f32_16x16x32_bf16 (half the MACs per instruction, a quarter of the accumulator registers): does the
  // shape change what the power cap allows?  Two instructions per slot keep the FLOP count of a loop iteration equal.
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc4[16];
  for (int q = 0; q < 16; ++q)
    for (int r = 0; r < 4; ++r) acc4[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      acc4[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[q & 3]), __builtin_bit_cast(bf16x8, b[(q >> 1) & 3]), acc4[q], 0, 0, 0);
  }
  float s4 = 0.f;
  for (int q = 0; q < 16; ++q)
    for (int r = 0; r < 4; ++r) s4 += acc4[q][r];
  if (s4 == 12345.678f) out[0] = s4;
  return;
#endif
  f32x16 acc[8];
  for (int q = 0; q < 8; ++q)
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
#ifdef USE_BF16
#ifdef SAME_B
      // -DSAME_B: eight consecutive MFMAs share their B operand (the convolution's row reuse: one pixel fragment, several taps)
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q & 3]), __builtin_bit_cast(bf16x8, b[0]), acc[q], 0, 0, 0);
#else
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q & 3]), __builtin_bit_cast(bf16x8, b[(q >> 1) & 3]), acc[q], 0, 0, 0);
#endif
#else
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q & 3], b[(q >> 1) & 3], acc[q], 0, 0, 0);
#endif
  }
  float s = 0.f;
  for (int q = 0; q < 8; ++q)
    for (int r = 0; r < 16; ++r) s += acc[q][r];
  if (s == 12345.678f) out[0] = s;       // keep the loop
}

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  std::vector<uint16_t> h(8 * 64 * 8);
  uint4* d_ab; float* d_out;
  CK(hipMalloc(&d_ab, h.size() * 2)); CK(hipMalloc(&d_out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * waves_per_simd;           // 256-thread blocks: 4 waves = one per SIMD
  const int iters = 20000;
  const double flop_per_launch = (double)grid * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
  for (const char* kind : {"zero", "small ints", "random f16", "random low pieces"}) {
    unsigned s = 12345u;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u;
      const float r = ((s >> 8) & 0xffffff) / 16777216.f - 0.5f;
      _Float16 x = kind[0] == 'z' ? (_Float16)0.f : kind[0] == 's' ? (_Float16)(float)((int)(r * 8)) :
                   kind[7] == 'f' ? (_Float16)(r * 6.f) : (_Float16)(r * 6.f * 0.00048828125f);
      v = __builtin_bit_cast(uint16_t, x);
#ifdef USE_BF16
      {
        const float xf = kind[0] == 'z' ? 0.f : kind[0] == 's' ? (float)((int)(r * 8)) : kind[7] == 'f' ? r * 6.f : r * 6.f * 0.00390625f;
        unsigned bits = __builtin_bit_cast(unsigned, xf);
        bits += 0x7FFFu + ((bits >> 16) & 1u);
        v = (uint16_t)(bits >> 16);
      }
#endif
    }
    CK(hipMemcpy(d_ab, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    printf("%-18s %d waves/SIMD: ", kind, waves_per_simd);
    for (int w = 0; w < 8; ++w) {
      CK(hipEventRecord(e0));
      int n = 0;
      float ms = 0;
      do {
        hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, d_ab, iters, d_out);
        ++n;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      } while (ms < 250.f);
      printf("%.0f ", flop_per_launch * n / (ms * 1e-3) / 1e12);
      fflush(stdout);
    }
    printf("TFLOP/s\n");
  }
  return 0;
}
