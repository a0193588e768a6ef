// GPU box: sustained rate of conv5x5_f16x3_kernel (whichever conv_f16.h is first on the include path):
// back-to-back launches for several seconds, time per launch reported per window of 200 launches, so
// that the power-management state the benchmark runs in is reached (a 35 ms burst is not).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dmpfold2_amd/csrc tools/ubench_conv_sustained.hip -o tools/_bin/ubench_conv_sus
#define CONV_F16_KERNELS
#include "conv_f16.h"
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <vector>
namespace dmp {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); }
int hip_fail(hipError_t e, const char* what, const char*, int line) { printf("HIP error %s (%s) line %d\n", hipGetErrorString(e), what, line); return -2; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using namespace dmp;
int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 300;
  const int windows = argc > 2 ? atoi(argv[2]) : 20;
  const int streams = argc > 3 ? atoi(argv[3]) : 1;     // 2 = two launches in flight, as the scheduler's lane
  std::vector<float> w((size_t)512 * 128 * 25), b(512);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };
  for (auto& v : w) v = rnd() * 0.04f;
  for (auto& v : b) v = rnd() * 0.1f;
  const float scale = conv_weight_scale_f16(w.data(), w.size());
  std::vector<uint16_t> wq = pack_conv_weights_f16(w.data(), scale);
  uint16_t* d_wq; float* d_b;
  CK(hipMalloc(&d_wq, wq.size() * 2)); CK(hipMalloc(&d_b, 512 * 4));
  CK(hipMemcpy(d_wq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, b.data(), 512 * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)conv5x5_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CONVH_LDS_BYTES));
  const int P = act_pitch(L), tiles = act_tiles(L);
  std::vector<uint16_t> xs((size_t)2 * 16 * P * P * 8, 0);
  for (int ch = 0; ch < 128; ++ch)
    for (int y = 0; y < L; ++y)
      for (int xx = 0; xx < L; ++xx) {
        uint16_t p2[2];
        split2_f16(rnd() * 6.f, p2);
        for (int p = 0; p < 2; ++p) xs[((((size_t)p * 16 + ch / 8) * P + y + 2) * P + xx + 2) * 8 + ch % 8] = p2[p];
      }
  uint16_t* d_xs; CK(hipMalloc(&d_xs, xs.size() * 2));
  CK(hipMemcpy(d_xs, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
  float* d_u[2]; double* d_part[2]; hipStream_t st[2];
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&d_u[i], (size_t)128 * L * L * 4)); CK(hipMalloc(&d_part[i], (size_t)tiles * tiles * 128 * 2 * 8));
    CK(hipStreamCreate(&st[i]));
  }
  const int nwork = tiles * tiles * 4, grid = conv_f16_grid(tiles);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("L=%d grid=%d LDS=%d streams=%d\n", L, grid, (int)CONVH_LDS_BYTES, streams);
  for (int wdw = 0; wdw < windows; ++wdw) {
    CK(hipEventRecord(e0, st[0]));
    for (int i = 0; i < 200; ++i) {
      hipStream_t q = st[streams > 1 ? (i & 1) : 0];
      hipLaunchKernelGGL(conv5x5_f16x3_kernel, dim3(grid), dim3(256), CONVH_LDS_BYTES, q, d_xs, d_wq, d_b,
                         1.0f / scale, L, P, tiles, nwork, d_u[i & 1], d_part[i & 1]);
    }
    if (streams > 1) { CK(hipStreamSynchronize(st[1])); }
    CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%.4f ", ms / 200); fflush(stdout);
  }
  printf("ms per launch\n");
  return 0;
}
