// GPU box: stand-alone characterisation of the hazard of DESIGN section 6.
//
// A small kernel executes packed-f32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 /
// v_pk_mov_b32, with and without op_sel / neg modifiers) and checks every result bit for bit against
// the same arithmetic done with scalar-per-lane instructions.  It runs (1) alone and (2) while another
// stream keeps the CUs busy with conv5x5_f16x3_kernel (the product's f16 split-product convolution).
// Mismatches are counted per instruction form and per quarter of the wave (lanes 0-15, ..., 48-63).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_hazard.hip -o tools/_bin/pk_hazard
//   tools/_bin/pk_hazard [workgroups 2] [iterations 2000] [launches 400]
#define CONV_F16_KERNELS
#include "../dmpfold2_amd/csrc/conv_f16.h"
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
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

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int NFORM = 18;
static const char* FORM_NAME[NFORM] = {
    "v_pk_mul_f32", "v_pk_mul_f32 op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]",
    "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]", "v_pk_fma_f32", "v_pk_mov_b32 op_sel:[1,0]",
    "v_mul_f32 x2 (scalar control)", "v_pk_add_f32 op_sel_hi:[1,0] neg",
    "v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 b, a, b op_sel:[0,1] op_sel_hi:[1,0]",
    "v_pk_mul_f32 a, a, b op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 a, a, b op_sel:[1,0] op_sel_hi:[0,1]",
    "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]",
    "v_pk_mul_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]",
    "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0]", "v_pk_add_f32 op_sel:[0,1]"};

__device__ __forceinline__ float smul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float ssub(float a, float b) {
  float r;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sfma(float a, float b, float c) {
  float r;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ bool same(f2 p, float x, float y) {
  const float px = p[0], py = p[1];
  return __float_as_uint(px) == __float_as_uint(x) && __float_as_uint(py) == __float_as_uint(y);
}

// bad[form][quarter]; first[form][4] = {lane, got.x, got.y bits, iteration} of the first mismatch
__global__ __launch_bounds__(256) void pk_test_kernel(const f2* __restrict__ in, int R,
                                                      unsigned* __restrict__ bad,
                                                      unsigned* __restrict__ first, f2* __restrict__ sink) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int q = (threadIdx.x & 63) >> 4;
  f2 s = in[2 * t];
  const f2 b = in[2 * t + 1];
  f2 acc = {0.f, 0.f};
  for (int i = 0; i < R; ++i) {
    f2 r[NFORM];
    float ex[NFORM], ey[NFORM];
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r[0]) : "v"(s), "v"(b));
    ex[0] = smul(s.x, b.x); ey[0] = smul(s.y, b.y);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r[1]) : "v"(s), "v"(b));
    ex[1] = smul(s.x, b.x); ey[1] = smul(s.y, b.x);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r[2]) : "v"(s), "v"(b));
    ex[2] = smul(s.y, b.x); ey[2] = smul(s.x, b.y);
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r[3]) : "v"(s), "v"(b));
    ex[3] = ssub(s.x, b.x); ey[3] = ssub(s.y, b.y);
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[4]) : "v"(s), "v"(b), "v"(acc));
    ex[4] = sfma(s.x, b.x, acc.x); ey[4] = sfma(s.y, b.y, acc.y);
    asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r[5]) : "v"(s), "v"(b));
    ex[5] = s.y; ey[5] = b.x;
    r[6].x = smul(s.x, b.y); r[6].y = smul(s.y, b.x);
    ex[6] = smul(s.x, b.y); ey[6] = smul(s.y, b.x);
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r[7]) : "v"(s), "v"(b));
    ex[7] = ssub(s.x, b.x); ey[7] = ssub(s.y, b.x);
    // src1 cross-swizzled: separate destination, destination = src1, destination = src0
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r[8]) : "v"(s), "v"(b));
    ex[8] = smul(s.x, b.y); ey[8] = smul(s.y, b.x);
    r[9] = b;
    asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(r[9]) : "v"(s));
    ex[9] = smul(s.x, b.y); ey[9] = smul(s.y, b.x);
    r[10] = s;
    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(r[10]) : "v"(b));
    ex[10] = smul(s.x, b.y); ey[10] = smul(s.y, b.x);
    // src0 cross-swizzled in place
    r[11] = s;
    asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(r[11]) : "v"(b));
    ex[11] = smul(s.y, b.x); ey[11] = smul(s.x, b.y);
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r[12]) : "v"(s), "v"(b));
    ex[12] = ssub(s.x, -b.y); ey[12] = ssub(s.y, -b.x);
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r[13]) : "v"(s), "v"(b), "v"(acc));
    ex[13] = sfma(s.x, b.y, acc.x); ey[13] = sfma(s.y, b.x, acc.y);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r[14]) : "v"(s), "v"(b));
    ex[14] = smul(s.x, b.y); ey[14] = smul(s.y, b.y);
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r[15]) : "v"(s), "v"(b), "v"(acc));
    ex[15] = sfma(s.x, b.x, acc.y); ey[15] = sfma(s.y, b.y, acc.x);
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r[16]) : "v"(s), "v"(b));
    ex[16] = smul(s.y, b.y); ey[16] = smul(s.x, b.x);
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r[17]) : "v"(s), "v"(b));
    ex[17] = ssub(s.x, -b.y); ey[17] = ssub(s.y, -b.y);
#pragma unroll
    for (int f = 0; f < NFORM; ++f)
      if (!same(r[f], ex[f], ey[f])) {
        if (atomicAdd(&bad[f * 4 + q], 1u) == 0u) {
          first[(f * 4 + q) * 4 + 0] = threadIdx.x & 63;
          const float gx = r[f][0], gy = r[f][1];
          const bool xbad = __float_as_uint(gx) != __float_as_uint(ex[f]);
          first[(f * 4 + q) * 4 + 1] = __float_as_uint(xbad ? gx : gy);
          first[(f * 4 + q) * 4 + 2] = __float_as_uint(xbad ? ex[f] : ey[f]);
          first[(f * 4 + q) * 4 + 3] = (unsigned)i;
        }
      }
    // next state from the scalar results (stays O(1): b is close to 1)
    acc.x = ex[3]; acc.y = ey[3];
    s.x = ex[0]; s.y = ey[0];
    if ((i & 255) == 255) s = in[2 * t];
  }
  sink[t] = s + acc;
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 2;
  const int R = argc > 2 ? atoi(argv[2]) : 2000;
  const int launches = argc > 3 ? atoi(argv[3]) : 400;
  const int L = 300;
  unsigned s = 4242u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffffff) / 16777216.f - 0.5f; };

  // the co-runner: the product's f16 split-product convolution on random data
  std::vector<float> w((size_t)512 * 128 * 25), bias(512);
  for (auto& v : w) v = rnd() * 0.04f;
  for (auto& v : bias) v = rnd() * 0.1f;
  const float scale = conv_weight_scale_f16(w.data(), w.size());
  std::vector<uint16_t> wq = pack_conv_weights_f16(w.data(), scale);
  const int P = act_pitch(L), tiles = act_tiles(L);
  std::vector<uint16_t> xs((size_t)2 * 16 * P * P * 8, 0);
  for (int ch = 0; ch < 128; ++ch)
    for (int y = 0; y < L; ++y)
      for (int xx = 0; xx < L; ++xx) {
        uint16_t p2[2];
        split2_f16(rnd() * 6.f, p2);
        for (int p = 0; p < 2; ++p)
          xs[((((size_t)p * 16 + ch / 8) * P + y + 2) * P + xx + 2) * 8 + ch % 8] = p2[p];
      }
  uint16_t *d_wq, *d_xs; float *d_b, *d_u; double* d_part;
  CK(hipMalloc(&d_wq, wq.size() * 2)); CK(hipMalloc(&d_b, 512 * 4));
  CK(hipMalloc(&d_xs, xs.size() * 2)); CK(hipMalloc(&d_u, (size_t)128 * L * L * 4));
  CK(hipMalloc(&d_part, (size_t)tiles * tiles * 128 * 2 * 8));
  CK(hipMemcpy(d_wq, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, bias.data(), 512 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_xs, xs.data(), xs.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)conv5x5_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                         CONVH_LDS_BYTES));
  const int grid = conv_f16_grid(tiles), nwork = tiles * tiles * 4;

  // the test kernel's operands: s in [0.5, 1.5), b in [0.9995, 1.0005)
  const int nthreads = wgs * 256;
  std::vector<f2> in((size_t)2 * nthreads);
  for (int t = 0; t < nthreads; ++t) {
    in[2 * t] = {1.0f + rnd(), 1.0f + rnd()};
    in[2 * t + 1] = {1.0f + rnd() * 1e-3f, 1.0f + rnd() * 1e-3f};
  }
  f2 *d_in, *d_sink; unsigned *d_bad, *d_first;
  CK(hipMalloc(&d_in, in.size() * sizeof(f2))); CK(hipMalloc(&d_sink, nthreads * sizeof(f2)));
  CK(hipMalloc(&d_bad, NFORM * 4 * 4)); CK(hipMalloc(&d_first, NFORM * 4 * 4 * 4));
  CK(hipMemcpy(d_in, in.data(), in.size() * sizeof(f2), hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

  for (int phase = 0; phase < 2; ++phase) {
    CK(hipMemset(d_bad, 0, NFORM * 4 * 4)); CK(hipMemset(d_first, 0, NFORM * 4 * 4 * 4));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, sb));
    for (int l = 0; l < launches; ++l) {
      if (phase == 1)
        hipLaunchKernelGGL(conv5x5_f16x3_kernel, dim3(grid), dim3(256), CONVH_LDS_BYTES, sa, d_xs, d_wq, d_b,
                           1.0f / scale, L, P, tiles, nwork, d_u, d_part);
      hipLaunchKernelGGL(pk_test_kernel, dim3(wgs), dim3(256), 0, sb, d_in, R, d_bad, d_first, d_sink);
    }
    CK(hipEventRecord(e1, sb));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned bad[NFORM * 4], first[NFORM * 16];
    CK(hipMemcpy(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost));
    CK(hipMemcpy(first, d_first, sizeof(first), hipMemcpyDeviceToHost));
    printf("%s: %d launches x %d workgroups x %d iterations (test stream %.1f ms)\n",
           phase ? "BESIDE conv5x5_f16x3_kernel" : "ALONE", launches, wgs, R, ms);
    const double per = (double)launches * wgs * 4 * R;      // wave-instructions per form
    for (int f = 0; f < NFORM; ++f) {
      printf("  %-48s mismatching lanes by wave quarter: %8u %8u %8u %8u   (of %.3g wave-instructions)\n",
             FORM_NAME[f], bad[f * 4], bad[f * 4 + 1], bad[f * 4 + 2], bad[f * 4 + 3], per);
      for (int q = 0; q < 4; ++q)
        if (bad[f * 4 + q])
          printf("      first in quarter %d: lane %u got %08x expected %08x at iteration %u\n", q,
                 first[(f * 4 + q) * 4], first[(f * 4 + q) * 4 + 1], first[(f * 4 + q) * 4 + 2],
                 first[(f * 4 + q) * 4 + 3]);
    }
  }
  return 0;
}
