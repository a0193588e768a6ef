// Developer micro-benchmark (GPU box): practical ceiling of v_mfma_f32_32x32x2_f32 for the conv5x5
// kernel's geometry (1444 workgroups x 4 waves x 12800 MFMAs, 8 accumulators per wave, 2 WG/CU)
// with no memory traffic at all.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma.hip -o tools/_bin/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LDSKB>
__global__ __launch_bounds__(256, 2) void mfma_only(float* out, int iters) {
  __shared__ float pad[LDSKB * 256];
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 25; ++t) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + t, b + i, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  pad[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = pad[0] + pad[255];
}

int main() {
  float* out;
  CK(hipMalloc(&out, 1 << 20));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 1444.0 * 4 * 64 * 25 * 8 * 4096.0;     // = 2*128*512*25*304*304 (padded tiles)
  for (int grid : {1444, 1536, 2048}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((mfma_only<57>), dim3(grid), dim3(256), 0, 0, out, 64);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mfma_only<57>), dim3(grid), dim3(256), 0, 0, out, 64);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= 10;
      printf("MFMA-only grid=%d (58 KB LDS, 2 WG/CU): %.3f ms  -> %.1f TFLOP/s\n", grid, ms,
             flop * grid / 1444.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
