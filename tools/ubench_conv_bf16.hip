// Prototype + micro-benchmark (GPU box): conv5x5 (128->512) + bias + 4-way maxout with float32
// semantics on the bf16 matrix cores (dmpfold2_amd/csrc/conv_bf16.h), all scheduling variants:
// correctness against a float64 CPU convolution at a small L, timing at L = 300.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_conv_bf16.hip -o tools/_bin/ubench_conv_bf16
#define CONV_BF16_KERNELS
#include "../dmpfold2_amd/csrc/conv_bf16.h"
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

struct Ctx { uint16_t *wq, *xs; float *b, *u; double* part; int L, P, tiles; };

static void launch(const Ctx& c) {
  const int nwork = c.tiles * c.tiles * 4, grid = (nwork + 7) / 8 * 8;
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute((const void*)conv5x5_bf16x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                           CONVQ_LDS_BYTES));
    attr = true;
  }
  hipLaunchKernelGGL(conv5x5_bf16x6_kernel, dim3(grid), dim3(256), CONVQ_LDS_BYTES, 0, c.xs, c.wq, c.b, c.L,
                     c.P, c.tiles, nwork, c.u, c.part);
}

static void check(const Ctx& c, const std::vector<float>& ref) {
  CK(hipMemset(c.u, 0, (size_t)128 * c.L * c.L * 4));
  launch(c);
  CK(hipDeviceSynchronize());
  std::vector<float> u((size_t)128 * c.L * c.L);
  CK(hipMemcpy(u.data(), c.u, u.size() * 4, hipMemcpyDeviceToHost));
  double md = 0, mr = 0;
  for (size_t i = 0; i < u.size(); ++i) { md = fmax(md, fabs((double)u[i] - ref[i])); mr = fmax(mr, fabs(ref[i])); }
  printf("L=%d  max|u - ref| = %.3e (scale %.3e, rel %.2e)\n", c.L, md, mr, md / mr);
}

static void timeit(const Ctx& c) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch(c);
  CK(hipDeviceSynchronize());
  float best = 1e9f, tot = 0;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch(c);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms / 10); tot += ms / 10;
  }
  const double flop = 2.0 * 128 * 512 * 25 * c.L * c.L;
  printf("L=%d  %.3f ms avg, %.3f ms best -> %.1f TFLOP/s float32-equivalent (%.0f bf16 executed)\n",
         c.L, tot / 5, best, flop / (best * 1e-3) / 1e12, 6 * flop / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const int Lt = argc > 1 ? atoi(argv[1]) : 24;
  const int Lb = argc > 2 ? atoi(argv[2]) : 300;
  std::vector<float> w((size_t)512 * 128 * 25), b(512);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };
  for (auto& v : w) v = rnd() * 0.04f;
  for (auto& v : b) v = rnd() * 0.1f;
  std::vector<uint16_t> wq = pack_conv_weights_bf16(w.data());
  Ctx c{};
  CK(hipMalloc(&c.wq, wq.size() * 2)); CK(hipMalloc(&c.b, 512 * 4));
  CK(hipMemcpy(c.wq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(c.b, b.data(), 512 * 4, hipMemcpyHostToDevice));
  for (int L : {Lt, Lb}) {
    c.L = L; c.P = act_pitch(L); c.tiles = act_tiles(L);
    const int P = c.P;
    std::vector<float> x((size_t)128 * L * L);
    for (auto& v : x) v = rnd() * 6.f;
    std::vector<uint16_t> xs((size_t)3 * 16 * P * P * 8, 0);
    for (int ch = 0; ch < 128; ++ch)
      for (int y = 0; y < L; ++y)
        for (int xx = 0; xx < L; ++xx) {
          uint16_t p3[3];
          split3_bf16(x[((size_t)ch * L + y) * L + xx], p3);
          for (int p = 0; p < 3; ++p)
            xs[((((size_t)p * 16 + ch / 8) * P + y + 2) * P + xx + 2) * 8 + ch % 8] = p3[p];
        }
    CK(hipMalloc(&c.xs, xs.size() * 2)); CK(hipMalloc(&c.u, (size_t)128 * L * L * 4));
    CK(hipMalloc(&c.part, (size_t)c.tiles * c.tiles * 128 * 2 * 8));
    CK(hipMemcpy(c.xs, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
    if (L == Lt) {
      std::vector<float> ref((size_t)128 * L * L);
      cpu_ref(x, w, b, L, ref);
      check(c, ref);
    } else {
      timeit(c);
    }
    CK(hipFree(c.xs)); CK(hipFree(c.u)); CK(hipFree(c.part));
  }
  return 0;
}
