// Developer micro-benchmark (GPU box): where does the vertical-GRU step kernel spend its time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_vgru.hip -o gpurun_out/ubench_vgru
// Variants of one wave's K=512 x 3-gate contraction, launched with the product kernel's geometry
// (320 workgroups x 256 threads, 64 KB LDS): MFMA only / loads only / both / LDS-staged.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int Lb = 320;

template <int MODE>   // 0 both (hand-staged), 1 MFMA only, 2 loads only, 3 both (naive loop)
__device__ __forceinline__ void part(const float* __restrict__ w, const float* __restrict__ x,
                                     f32x16& a0, f32x16& a1, f32x16& a2, float& sink) {
  if (MODE == 3) {
#pragma unroll 8
    for (int p = 0; p < 64; ++p) {
      const float* wk = w + (int64_t)p * 8 * 1536;
      const float xv = x[(int64_t)p * 8 * Lb];
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[0], xv, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[512], xv, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[1024], xv, a2, 0, 0, 0);
    }
    return;
  }
  float wv[2][8][3], xv[2][8];
  auto load_chunk = [&](int buf, int c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 1) {
        wv[buf][i][0] = 1.0f + c; wv[buf][i][1] = 2.0f + i; wv[buf][i][2] = 3.0f; xv[buf][i] = 0.5f;
      } else {
        const float* wk = w + (int64_t)(c * 8 + i) * 8 * 1536;
        wv[buf][i][0] = wk[0]; wv[buf][i][1] = wk[512]; wv[buf][i][2] = wk[1024];
        xv[buf][i] = x[(int64_t)(c * 8 + i) * 8 * Lb];
      }
    }
  };
  load_chunk(0, 0);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c + 1 < 8) load_chunk((c + 1) & 1, c + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 2) {
        sink += wv[c & 1][i][0] * xv[c & 1][i] + wv[c & 1][i][1] + wv[c & 1][i][2];
      } else {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[c & 1][i][0], xv[c & 1][i], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[c & 1][i][1], xv[c & 1][i], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[c & 1][i][2], xv[c & 1][i], a2, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MODE, int LDSKB>
__global__ __launch_bounds__(256) void step(const float* __restrict__ wx, const float* __restrict__ wh,
                                            const float* __restrict__ h0, const float* __restrict__ h1,
                                            float* __restrict__ out, int heavy_only) {
  __shared__ float red[LDSKB * 256];
  const int nbt = Lb >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rest = slot >> 1;
  const int layer = 1 - rest / nbt;
  if (heavy_only && layer == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int b0 = (rest % nbt) * 32, j0 = (2 * xcd + (slot & 1)) * 32;
  f32x16 ar, az, ai, ah;
#pragma unroll
  for (int r = 0; r < 16; ++r) { ar[r] = 0; az[r] = 0; ai[r] = 0; ah[r] = 0; }
  float sink = 0.f;
  const int64_t ro = (int64_t)(2 * wave + kk);
  if (layer == 1) part<MODE>(wx + j0 + li + ro * 1536, h0 + b0 + li + ro * Lb, ar, az, ai, sink);
  part<MODE>(wh + j0 + li + ro * 1536, (layer ? h1 : h0) + b0 + li + ro * Lb, ar, az, ah, sink);
  float s = sink;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += ar[r] + az[r] + ai[r] + ah[r];
  red[tid] = s;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = red[0] + red[255];
}

// LDS-staged variant: the workgroup streams its W slab (96 gate columns x 512 k) through LDS with
// coalesced float4 loads (double buffered, 32 k rows per stage), all 4 waves then read fragments.
__global__ __launch_bounds__(256) void step_lds(const float* __restrict__ wx, const float* __restrict__ wh,
                                                const float* __restrict__ h0, const float* __restrict__ h1,
                                                float* __restrict__ out, int heavy_only) {
  __shared__ __attribute__((aligned(16))) float wl[2][32][96];
  __shared__ __attribute__((aligned(16))) float xl[2][32][32];
  __shared__ float red[256];
  const int nbt = Lb >> 5;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rest = slot >> 1;
  const int layer = 1 - rest / nbt;
  if (heavy_only && layer == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int b0 = (rest % nbt) * 32, j0 = (2 * xcd + (slot & 1)) * 32;
  f32x16 ar, az, ai, ah;
#pragma unroll
  for (int r = 0; r < 16; ++r) { ar[r] = 0; az[r] = 0; ai[r] = 0; ah[r] = 0; }
  // staging: 32 rows x (3 gates x 32) floats = 768 float4 -> 3 per thread; x: 32 x 32 = 256 float4 -> 1
  auto run = [&](const float* W, const float* X, f32x16& g0, f32x16& g1, f32x16& g2) {
    float4 wr[3], xr;
    auto fetch = [&](int st) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int idx = tid + e * 256;          // 0..767
        const int row = idx / 24, c4 = idx % 24;  // 24 float4 per row (3 gates x 8)
        const int gate = c4 >> 3, q = c4 & 7;
        wr[e] = *reinterpret_cast<const float4*>(W + (int64_t)(st * 32 + row) * 1536 + gate * 512 + j0 + q * 4);
      }
      const int row = tid >> 3, q = tid & 7;
      xr = *reinterpret_cast<const float4*>(X + (int64_t)(st * 32 + row) * Lb + b0 + q * 4);
    };
    auto commit = [&](int buf) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int idx = tid + e * 256;
        const int row = idx / 24, c4 = idx % 24;
        *reinterpret_cast<float4*>(&wl[buf][row][c4 * 4]) = wr[e];
      }
      *reinterpret_cast<float4*>(&xl[buf][tid >> 3][(tid & 7) * 4]) = xr;
    };
    fetch(0); commit(0); __syncthreads();
    for (int st = 0; st < 16; ++st) {
      const int buf = st & 1;
      if (st + 1 < 16) fetch(st + 1);
      // each wave takes 4 of the 16 k-pairs of this stage
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 2 * (wave * 4 + q) + kk;
        const float xv = xl[buf][k][li];
        g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[buf][k][li], xv, g0, 0, 0, 0);
        g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[buf][k][32 + li], xv, g1, 0, 0, 0);
        g2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[buf][k][64 + li], xv, g2, 0, 0, 0);
      }
      if (st + 1 < 16) commit(buf ^ 1);
      __syncthreads();
    }
  };
  if (layer == 1) run(wx, h0, ar, az, ai);
  run(wh, layer ? h1 : h0, ar, az, ah);
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += ar[r] + az[r] + ai[r] + ah[r];
  red[tid] = s;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = red[0] + red[255];
}

__global__ void empty_kernel(float* o) { if (threadIdx.x == 9999) o[0] = 1; }

template <typename F>
static float time_it(const char* tag, int iters, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-44s %8.2f us per launch\n", tag, ms / iters * 1e3);
  return ms / iters;
}

int main() {
  float *wx, *wh, *h0, *h1, *out;
  CK(hipMalloc(&wx, 512 * 1536 * 4)); CK(hipMalloc(&wh, 512 * 1536 * 4));
  CK(hipMalloc(&h0, 512 * Lb * 4)); CK(hipMalloc(&h1, 512 * Lb * 4)); CK(hipMalloc(&out, 4096 * 4));
  std::vector<float> hw(512 * 1536);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(wx, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wh, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(h0, hw.data(), 512 * Lb * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(h1, hw.data(), 512 * Lb * 4, hipMemcpyHostToDevice));
  const int grid = 8 * 2 * (Lb / 32) * 2;
  const int it = 300;
  time_it("empty kernel (launch gap)", it, [&] { hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, out); });
  for (int heavy = 0; heavy < 2; ++heavy) {
    printf("--- %s\n", heavy ? "layer-1 (heavy) workgroups only" : "all 320 workgroups");
    time_it("staged loads + MFMA, 64 KB LDS", it, [&] { hipLaunchKernelGGL((step<0, 64>), dim3(grid), dim3(256), 0, 0, wx, wh, h0, h1, out, heavy); });
    time_it("staged loads + MFMA, 1 KB LDS", it, [&] { hipLaunchKernelGGL((step<0, 1>), dim3(grid), dim3(256), 0, 0, wx, wh, h0, h1, out, heavy); });
    time_it("MFMA only", it, [&] { hipLaunchKernelGGL((step<1, 1>), dim3(grid), dim3(256), 0, 0, wx, wh, h0, h1, out, heavy); });
    time_it("loads only", it, [&] { hipLaunchKernelGGL((step<2, 1>), dim3(grid), dim3(256), 0, 0, wx, wh, h0, h1, out, heavy); });
    time_it("naive loop (compiler-scheduled), 1 KB LDS", it, [&] { hipLaunchKernelGGL((step<3, 1>), dim3(grid), dim3(256), 0, 0, wx, wh, h0, h1, out, heavy); });
    time_it("LDS-staged float4 slabs", it, [&] { hipLaunchKernelGGL(step_lds, dim3(grid), dim3(256), 0, 0, wx, wh, h0, h1, out, heavy); });
  }
  return 0;
}
