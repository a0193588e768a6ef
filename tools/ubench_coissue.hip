// Does a wave issue VALU / transcendental instructions in the shadow of its own MFMAs?  (round 4: the persistent vertical
// GRU's gate arithmetic placed between groups of three MFMAs did not get any faster.)  One wave per SIMD, 256 CUs;
// per iteration 12 x v_mfma_f32_16x16x32_f16 on 6 accumulators, and V independent VALU instructions (MODE 1: v_fma_f32,
// MODE 2: v_exp_f32) behind every MFMA.    hipcc --offload-arch=gfx950 -O3 tools/ubench_coissue.hip -o /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int V>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[6];
  for (int i = 0; i < 6; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m % 6]) : "v"(a), "v"(b));
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[v % 8]));
        if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[v % 8]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 6; ++i) s += acc[i][0];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int V>
void run(float* d, const char* name) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, V><<<256, 256>>>(d, 100);
  hipEventRecord(e0);
  k<MODE, V><<<256, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %7.3f ms = %6.1f ns per MFMA (+%d)\n", name, ms, ms * 1e6 / (iters * 12.0), V);
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 256 * 4);
  run<0, 0>(d, "MFMA only");
  run<1, 1>(d, "MFMA + 1 v_fma");
  run<1, 2>(d, "MFMA + 2 v_fma");
  run<1, 3>(d, "MFMA + 3 v_fma");
  run<1, 4>(d, "MFMA + 4 v_fma");
  run<2, 1>(d, "MFMA + 1 v_exp");
  run<2, 2>(d, "MFMA + 2 v_exp");
  return 0;
}
