// Prototype + micro-benchmark (GPU box): conv5x5 (128->512) + bias + 4-way maxout with float32
// semantics on the bf16 matrix cores.  x = x0 + x1 + x2 and w = w0 + w1 + w2 are EXACT 3-way bf16
// splits of the float32 operands; six of the nine cross products are accumulated in float32
// (w0x0, w0x1, w1x0, w0x2, w1x1, w2x0 - the dropped ones are below 2^-24 relative), which
// reproduces the float32 convolution to float32 rounding error at 16/6 = 2.67x the f32 MFMA rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_conv_bf16.hip -o tools/_bin/ubench_conv_bf16
#define CONV_BF16_KERNELS
#include "../dmpfold2_amd/csrc/conv_bf16.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace dmp;

static void cpu_ref(const std::vector<float>& x, const std::vector<float>& w, const std::vector<float>& b,
                    int L, std::vector<float>& u) {
  // x [128][L][L], w [512][128][5][5] -> u [128][L][L] (max over channel quadruples), in double
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
  const int Lt = argc > 1 ? atoi(argv[1]) : 24;      // correctness size
  const int Lb = argc > 2 ? atoi(argv[2]) : 300;     // timing size
  std::vector<float> w((size_t)512 * 128 * 25), b(512);
  unsigned s = 777u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };
  for (auto& v : w) v = rnd() * 0.04f;
  for (auto& v : b) v = rnd() * 0.1f;
  std::vector<uint16_t> wq = pack_conv_weights_bf16(w.data());
  uint16_t* d_wq; float* d_b;
  CK(hipMalloc(&d_wq, wq.size() * 2)); CK(hipMalloc(&d_b, 512 * 4));
  CK(hipMemcpy(d_wq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, b.data(), 512 * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)conv5x5_bf16x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CONVQ_LDS_BYTES));

  for (int L : {Lt, Lb}) {
    const int P = act_pitch(L), tiles = act_tiles(L);
    std::vector<float> x((size_t)128 * L * L);
    for (auto& v : x) v = rnd() * 6.f;
    // split + pack activations on the host: [piece][cgrp 16][P][P][8]
    std::vector<uint16_t> xs((size_t)3 * 16 * P * P * 8, 0);
    for (int c = 0; c < 128; ++c)
      for (int y = 0; y < L; ++y)
        for (int xx = 0; xx < L; ++xx) {
          uint16_t p3[3];
          split3_bf16(x[((size_t)c * L + y) * L + xx], p3);
          for (int p = 0; p < 3; ++p)
            xs[((((size_t)p * 16 + c / 8) * P + y + 2) * P + xx + 2) * 8 + c % 8] = p3[p];
        }
    uint16_t* d_xs; float* d_u; double* d_part;
    CK(hipMalloc(&d_xs, xs.size() * 2)); CK(hipMalloc(&d_u, (size_t)128 * L * L * 4));
    CK(hipMalloc(&d_part, (size_t)tiles * tiles * 128 * 2 * 8));
    CK(hipMemcpy(d_xs, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
    const int nwork = tiles * tiles * 4, grid = (nwork + 7) / 8 * 8;
    auto launch = [&]() {
      hipLaunchKernelGGL(conv5x5_bf16x6_kernel, dim3(grid), dim3(256), CONVQ_LDS_BYTES, 0, d_xs, d_wq, d_b, L, P,
                         tiles, nwork, d_u, d_part);
    };
    launch();
    CK(hipDeviceSynchronize());
    if (L == Lt) {
      std::vector<float> u((size_t)128 * L * L), ref(u.size());
      CK(hipMemcpy(u.data(), d_u, u.size() * 4, hipMemcpyDeviceToHost));
      cpu_ref(x, w, b, L, ref);
      double md = 0, mr = 0;
      for (size_t i = 0; i < u.size(); ++i) { md = fmax(md, fabs((double)u[i] - ref[i])); mr = fmax(mr, fabs(ref[i])); }
      std::vector<double> part((size_t)tiles * tiles * 128 * 2);
      CK(hipMemcpy(part.data(), d_part, part.size() * 8, hipMemcpyDeviceToHost));
      double s1 = 0, r1 = 0;
      for (int t = 0; t < tiles * tiles; ++t) s1 += part[((size_t)t * 128 + 5) * 2];
      for (int i = 0; i < L * L; ++i) r1 += ref[(size_t)5 * L * L + i];
      printf("L=%d  max|u - ref| = %.3e (scale %.3e, rel %.2e)   channel-5 sum %.6f vs %.6f\n", L, md, mr,
             md / mr, s1, r1);
    } else {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / 10);
      }
      const double flop = 2.0 * 128 * 512 * 25 * L * L;
      printf("L=%d  bf16x6 conv: %.3f ms per launch -> %.1f TFLOP/s float32-equivalent (%.1f TFLOP/s bf16 executed)\n",
             L, best, flop / (best * 1e-3) / 1e12, 6 * flop / (best * 1e-3) / 1e12);
    }
    CK(hipFree(d_xs)); CK(hipFree(d_u)); CK(hipFree(d_part));
  }
  return 0;
}
