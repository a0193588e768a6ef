// GPU box: instruction-level investigation of the backbone-kernel hazard.
//
// Loads a code object that contains a backbone kernel (built from dmpfold2_amd/csrc/coords.hip, possibly
// with its assembly edited: tools/bb_hazard_variants.sh), runs it on a fixed C-alpha trace
//   (1) alone, which gives the reference output and shows the variant is deterministic, and
//   (2) beside conv5x5_f16x3_kernel running on another stream,
// and counts the outputs that differ from the reference by wave quarter (lanes 0-15 .. 48-63) and atom
// (N, CA, C, O, CB, confidence).  Launch geometry is the product's: cdiv(L, 256) workgroups of 256.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bb_hazard.hip -o tools/_bin/bb_hazard
//   tools/_bin/bb_hazard <code object> <kernel symbol> [launches 20000] [L 300]
#define CONV_F16_KERNELS
#include "../dmpfold2_amd/csrc/conv_f16.h"
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <cmath>
#include <vector>

namespace dmp {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vprintf(fmt, ap); va_end(ap); }
int hip_fail(hipError_t e, const char* what, int line) {
  printf("HIP error %s (%s) line %d\n", hipGetErrorString(e), what, line);
  return -1;
}
}  // namespace dmp
using namespace dmp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: bb_hazard <code object> <kernel symbol> [launches] [L]\n"); return 2; }
  const int launches = argc > 3 ? atoi(argv[3]) : 20000;
  const int L = argc > 4 ? atoi(argv[4]) : 300;
  const int LC = 300;
  unsigned s = 99u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };

  hipModule_t mod; hipFunction_t fn;
  CK(hipModuleLoad(&mod, argv[1]));
  CK(hipModuleGetFunction(&fn, mod, argv[2]));

  // co-runner
  std::vector<float> w((size_t)512 * 128 * 25), bias(512);
  for (auto& v : w) v = rnd() * 0.04f;
  for (auto& v : bias) v = rnd() * 0.1f;
  const float scale = conv_weight_scale_f16(w.data(), w.size());
  std::vector<uint16_t> wq = pack_conv_weights_f16(w.data(), scale);
  const int P = act_pitch(LC), tiles = act_tiles(LC);
  std::vector<uint16_t> xs((size_t)2 * 16 * P * P * 8, 0);
  for (int ch = 0; ch < 128; ++ch)
    for (int y = 0; y < LC; ++y)
      for (int xx = 0; xx < LC; ++xx) {
        uint16_t p2[2];
        split2_f16(rnd() * 6.f, p2);
        for (int p = 0; p < 2; ++p)
          xs[((((size_t)p * 16 + ch / 8) * P + y + 2) * P + xx + 2) * 8 + ch % 8] = p2[p];
      }
  uint16_t *d_wq, *d_xs; float *d_b, *d_u; double* d_part;
  CK(hipMalloc(&d_wq, wq.size() * 2)); CK(hipMalloc(&d_b, 512 * 4));
  CK(hipMalloc(&d_xs, xs.size() * 2)); CK(hipMalloc(&d_u, (size_t)128 * LC * LC * 4));
  CK(hipMalloc(&d_part, (size_t)tiles * tiles * 128 * 2 * 8));
  CK(hipMemcpy(d_wq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, bias.data(), 512 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_xs, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)conv5x5_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                         CONVH_LDS_BYTES));
  const int grid = conv_f16_grid(tiles), nwork = tiles * tiles * 4;

  // a chain-like C-alpha trace and confidence logits
  std::vector<float> ca((size_t)3 * L), lg(L);
  float p[3] = {0, 0, 0};
  for (int i = 0; i < L; ++i) {
    for (int k = 0; k < 3; ++k) { p[k] += rnd() * 4.4f; ca[3 * i + k] = p[k]; }
    lg[i] = rnd() * 6.f;
  }
  constexpr int SLOTS = 64;
  const size_t per = (size_t)L * 16;                     // 15 coordinates + 1 confidence per residue
  float *d_ca, *d_lg, *d_out;
  CK(hipMalloc(&d_ca, ca.size() * 4)); CK(hipMalloc(&d_lg, lg.size() * 4));
  CK(hipMalloc(&d_out, SLOTS * per * 4));
  CK(hipMemcpy(d_ca, ca.data(), ca.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_lg, lg.data(), lg.size() * 4, hipMemcpyHostToDevice));
  const double ang = 3.14159265358979323846 / 2.0 - asin(1.0 / sqrt(3.0));
  float sxc = (float)(1.5 * cos(ang)), syc = (float)(1.5 * sin(ang));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

  auto launch_bb = [&](int slot) {
    float* coords = d_out + slot * per;
    float* conf = coords + (size_t)L * 15;
    int Lv = L;
    void* args[] = {&d_ca, &d_lg, &Lv, &sxc, &syc, &coords, &conf};
    CK(hipModuleLaunchKernel(fn, (L + 255) / 256, 1, 1, 256, 1, 1, 0, sb, args, nullptr));
  };
  std::vector<float> ref(per), got(SLOTS * per);
  launch_bb(0);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), d_out, per * 4, hipMemcpyDeviceToHost));

  for (int phase = 0; phase < 2; ++phase) {
    long bad_launch = 0, total = 0, byq[4] = {0, 0, 0, 0}, byatom[6] = {0, 0, 0, 0, 0, 0};
    double maxdiff = 0;
    for (int done = 0; done < launches; done += SLOTS) {
      CK(hipMemsetAsync(d_out, 0xff, SLOTS * per * 4, sb));
      for (int k = 0; k < SLOTS; ++k) {
        if (phase == 1 && (k & 7) == 0)
          for (int c = 0; c < 2; ++c)
            hipLaunchKernelGGL(conv5x5_f16x3_kernel, dim3(grid), dim3(256), CONVH_LDS_BYTES, sa, d_xs, d_wq,
                               d_b, 1.0f / scale, LC, P, tiles, nwork, d_u, d_part);
        launch_bb(k);
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), d_out, SLOTS * per * 4, hipMemcpyDeviceToHost));
      for (int k = 0; k < SLOTS; ++k) {
        ++total;
        const float* g = got.data() + k * per;
        if (!memcmp(g, ref.data(), per * 4)) continue;
        ++bad_launch;
        if (bad_launch <= 2 && getenv("BB_DUMP"))
          for (int r = 0, shown = 0; r < L && shown < 3; ++r)
            if (memcmp(&g[(size_t)r * 15], &ref[(size_t)r * 15], 60)) {
              ++shown;
              printf("  residue %d (lane %d):\n", r, r & 63);
              for (int j = -1; j <= 3; ++j)
                if (r + j >= 0 && r + j < L)
                  printf("    ca[%d] = %.9g %.9g %.9g\n", r + j, ca[3 * (r + j)], ca[3 * (r + j) + 1], ca[3 * (r + j) + 2]);
              for (int a = 0; a < 5; ++a)
                printf("    atom %d got %.9g %.9g %.9g   expected %.9g %.9g %.9g\n", a, g[r * 15 + a * 3],
                       g[r * 15 + a * 3 + 1], g[r * 15 + a * 3 + 2], ref[r * 15 + a * 3], ref[r * 15 + a * 3 + 1],
                       ref[r * 15 + a * 3 + 2]);
            }
        for (int r = 0; r < L; ++r)
          for (int a = 0; a < 6; ++a) {
            bool diff = false;
            for (int k3 = 0; k3 < (a < 5 ? 3 : 1); ++k3) {
              const size_t i = a < 5 ? (size_t)r * 15 + a * 3 + k3 : (size_t)L * 15 + r;
              if (memcmp(&g[i], &ref[i], 4)) { diff = true; maxdiff = fmax(maxdiff, fabs((double)g[i] - ref[i])); }
            }
            if (diff) { ++byatom[a]; if (a == 2 || a == 0 || a == 4) ++byq[(r & 63) >> 4]; }
          }
      }
    }
    printf("%-30s %ld of %ld launches differ from the reference; residues by wave quarter %ld %ld %ld %ld; "
           "atoms N %ld CA %ld C %ld O %ld CB %ld conf %ld; max |diff| %.3g\n",
           phase ? "BESIDE conv5x5_f16x3_kernel:" : "ALONE:", bad_launch, total, byq[0], byq[1], byq[2], byq[3],
           byatom[0], byatom[1], byatom[2], byatom[3], byatom[4], byatom[5], maxdiff);
  }
  return 0;
}
