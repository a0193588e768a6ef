// Micro-benchmark (GPU box): conv5x5_f16x3_kernel (dmpfold2_amd/csrc/conv_f16.h): correctness against a
// float64 CPU convolution at a small L, timing at L = 300.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_conv_f16.hip -o tools/_bin/ubench_conv_f16
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
  const int lds_bytes = argc > 3 ? atoi(argv[3]) : CONVH_LDS_BYTES;   // > 80 KB forces 1 workgroup per CU
  std::vector<float> w((size_t)512 * 128 * 25), b(512);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };
  for (auto& v : w) v = rnd() * 0.04f;
  for (auto& v : b) v = rnd() * 0.1f;
  // UB_DATA=zero: all-zero operands; UB_DATA=f16: operands exactly representable in f16 (zero low pieces).
  // Data-dependent timing = the kernel is limited by power, not by instruction issue.
  const char* ub_data = getenv("UB_DATA");
  auto shape = [&](std::vector<float>& a) {
    if (!ub_data) return;
    for (auto& v : a) v = ub_data[0] == 'z' ? 0.f : (float)(_Float16)v;
  };
  shape(w);
  const float scale = conv_weight_scale_f16(w.data(), w.size());
  std::vector<uint16_t> wq = pack_conv_weights_f16(w.data(), scale);
  uint16_t* d_wq; float* d_b;
  CK(hipMalloc(&d_wq, wq.size() * 2)); CK(hipMalloc(&d_b, 512 * 4));
  CK(hipMemcpy(d_wq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, b.data(), 512 * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)conv5x5_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                         lds_bytes));
  printf("weight scale 2^%d, LDS %d B per workgroup\n", (int)std::log2(scale), lds_bytes);
  for (int L : {Lt, Lb}) {
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
    uint16_t* d_xs; float* d_u; double* d_part;
    CK(hipMalloc(&d_xs, xs.size() * 2)); CK(hipMalloc(&d_u, (size_t)128 * L * L * 4));
    CK(hipMalloc(&d_part, (size_t)tiles * tiles * 128 * 2 * 8));
    CK(hipMemcpy(d_xs, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
    const int nwork = tiles * tiles * 4, grid = conv_f16_grid(tiles);
    auto launch = [&]() {
      hipLaunchKernelGGL(conv5x5_f16x3_kernel, dim3(grid), dim3(256), lds_bytes, 0, d_xs, d_wq, d_b,
                         1.0f / scale, L, P, tiles, nwork, d_u, d_part);
    };
    launch();
    CK(hipDeviceSynchronize());
    if (L == Lt) {
      std::vector<float> u((size_t)128 * L * L), ref(u.size());
      CK(hipMemcpy(u.data(), d_u, u.size() * 4, hipMemcpyDeviceToHost));
      cpu_ref(x, w, b, L, ref);
      double md = 0, mr = 0;
      for (size_t i = 0; i < u.size(); ++i) { md = fmax(md, fabs((double)u[i] - ref[i])); mr = fmax(mr, fabs(ref[i])); }
      printf("L=%d  max|u - ref| = %.3e (scale %.3e, rel %.2e)\n", L, md, mr, md / mr);
    } else {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e9f, tot = 0;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / 10); tot += ms / 10;
      }
      const double flop = 2.0 * 128 * 512 * 25 * L * L;
      printf("L=%d  %.3f ms avg, %.3f ms best -> %.1f TFLOP/s float32-equivalent (%.0f f16 executed)\n", L,
             tot / 5, best, flop / (best * 1e-3) / 1e12, 3 * flop / (best * 1e-3) / 1e12);
    }
    CK(hipFree(d_xs)); CK(hipFree(d_u)); CK(hipFree(d_part));
  }
  return 0;
}
