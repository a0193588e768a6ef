// Micro-benchmark (GPU box): cost of a chain of dependent kernel launches on one stream, plain and
// replayed from a hipGraph - the floor under per-time-step kernels such as vgru_step_kernel.
// Variants: trivial kernel; kernel with a 136-byte by-value struct; kernel that first reads a
// record from device memory; kernel with 64 KB of static LDS.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/_bin/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Big { float* p[16]; int n; int pad; };
struct Rec { int t0, t_end; float* p; };

__global__ __launch_bounds__(512) void k_trivial(float* p, int n) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  if (i < n) p[i] += 1.0f;
}
__global__ __launch_bounds__(512) void k_big(Big b) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  if (i < b.n) b.p[i & 15][i] += 1.0f;
}
__global__ __launch_bounds__(512) void k_rec(Big b, const Rec* r, int idx) {
  const int t = r->t0 + idx;
  if (t >= r->t_end) return;
  const int i = blockIdx.x * 512 + threadIdx.x;
  if (i < b.n) r->p[i] += 1.0f;
}
__global__ __launch_bounds__(512) void k_lds(float* p, int n) {
  __shared__ float red[16384];
  const int i = blockIdx.x * 512 + threadIdx.x;
  red[threadIdx.x] = p[i];
  __syncthreads();
  if (i < n) p[i] = red[threadIdx.x ^ 1] + 1.0f;
}

template <class F>
static void bench(const char* name, hipStream_t s, F launch) {
  const int steps = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best_s = 1e9f, best_g = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int t = 0; t < steps; ++t) launch(t);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best_s = fminf(best_s, ms * 1e3f / steps);
  }
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int t = 0; t < 128; ++t) launch(t);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int c = 0; c < 16; ++c) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best_g = fminf(best_g, ms * 1e3f / (16 * 128));
  }
  printf("%-28s stream %.2f us/kernel   graph (128-node chains) %.2f us/kernel\n", name, best_s, best_g);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  const int grid = 320, n = grid * 512;
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
  Rec* r; CK(hipMalloc(&r, sizeof(Rec)));
  Rec hr{0, 1 << 30, d}; CK(hipMemcpy(r, &hr, sizeof(Rec), hipMemcpyHostToDevice));
  Big b{}; for (int i = 0; i < 16; ++i) b.p[i] = d; b.n = n;
  hipStream_t s; CK(hipStreamCreate(&s));
  bench("trivial", s, [&](int) { hipLaunchKernelGGL(k_trivial, dim3(grid), dim3(512), 0, s, d, n); });
  bench("136-byte struct argument", s, [&](int) { hipLaunchKernelGGL(k_big, dim3(grid), dim3(512), 0, s, b); });
  bench("record read from memory", s, [&](int t) { hipLaunchKernelGGL(k_rec, dim3(grid), dim3(512), 0, s, b, (const Rec*)r, t & 127); });
  bench("64 KB static LDS", s, [&](int) { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(512), 0, s, d, n); });
  return 0;
}
