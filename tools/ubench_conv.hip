// Developer micro-benchmark (GPU box): scheduling variants of conv5x5_maxout_kernel at L = 300.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. tools/ubench_conv.hip dmpfold2_amd/csrc/_build/api.o ... (see below)
// Built stand-alone: includes the kernel source and provides the two symbols it needs.
#include "../dmpfold2_amd/csrc/trunk.hip"
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <vector>

namespace dmp {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); }
int hip_fail(hipError_t e, const char* what, const char*, int line) {
  printf("HIP error %s (%s) line %d\n", hipGetErrorString(e), what, line);
  return -2;
}
}  // namespace dmp
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int V>
static void run(const char* tag, const float* x, const float* w, const float* b, int L, float* u, double* part) {
  using namespace dmp;
  const int tiles = act_tiles(L), P = act_pitch(L);
  const int nwork = tiles * tiles * CONV_SPLIT;
  const int grid = round_up(nwork, 8);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL(conv5x5_maxout_kernel<V>, dim3(grid), dim3(256), 0, 0, x, w, b, L, P, tiles, nwork, u, part);
  CK(hipDeviceSynchronize());
  float best = 1e9f, tot = 0.f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL(conv5x5_maxout_kernel<V>, dim3(grid), dim3(256), 0, 0, x, w, b, L, P, tiles, nwork, u, part);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    best = ms < best ? ms : best;
    tot += ms;
  }
  const double flop = 2.0 * 128 * 512 * 25 * L * L;
  printf("variant %d %-46s avg %.3f ms  best %.3f ms  -> %.1f TFLOP/s (best)\n", V, tag, tot / 5, best,
         flop / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  using namespace dmp;
  const int L = argc > 1 ? atoi(argv[1]) : 300;
  const int P = act_pitch(L), tiles = act_tiles(L);
  float *x, *w, *b, *u; double* part;
  const size_t nx = (size_t)CW * P * P, nw = (size_t)512 * 128 * 25;
  CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&b, 512 * 4));
  CK(hipMalloc(&u, (size_t)CW * L * L * 4)); CK(hipMalloc(&part, (size_t)tiles * tiles * CW * 2 * 8));
  std::vector<float> h(nx > nw ? nx : nw);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
  CK(hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), nw * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data(), 512 * 4, hipMemcpyHostToDevice));
  printf("L = %d\n", L);
  run<0>("compiler schedule, guarded prefetch", x, w, b, L, u, part);
  run<1>("branch-free prefetch", x, w, b, L, u, part);
  run<2>("tap-ahead LDS reads", x, w, b, L, u, part);
  run<3>("branch-free + tap-ahead", x, w, b, L, u, part);
  run<6>("tap-ahead + group barriers", x, w, b, L, u, part);
  run<7>("branch-free + tap-ahead + group barriers", x, w, b, L, u, part);
  run<65>("LDS-DMA weight slab", x, w, b, L, u, part);
  run<193>("LDS-DMA weight slab + input tile", x, w, b, L, u, part);
  run<17>("branch-free, NO barrier (timing only)", x, w, b, L, u, part);
  run<33>("branch-free, NO staging (timing only)", x, w, b, L, u, part);
  run<49>("branch-free, NO staging NO barrier (timing only)", x, w, b, L, u, part);
  return 0;
}
