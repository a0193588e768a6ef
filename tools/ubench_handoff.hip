// GPU box: one-way latency of a {flag, value} hand-off between two workgroups through global memory, by the cache
// policy bits of the store and of the polling load (gfx950: sc0 / sc1 / nt; read-modify-write atomics execute in L2).
// The cluster kernels (seq_gru_kernel, refine_cluster_kernel) publish 8-byte {epoch, value} granules with agent-scope
// relaxed atomic stores / loads; this measures what that costs against the alternatives, for partners on the same
// XCD (blocks 0 and 8) and on different XCDs (blocks 0 and 1).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_handoff tools/ubench_handoff.hip && ./ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

template <int S> __device__ __forceinline__ void st(u64* p, u64 v) {
  if (S == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  if (S == 2) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (S == 3) asm volatile("global_atomic_swap_x2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (S == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
template <int L> __device__ __forceinline__ u64 ld(u64* p) {
  u64 v = 0;
  if (L == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 3) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 4) { u64 z = 0; asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory"); }
  if (L == 5) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// flags[0]: written by A, polled by B; flags[16] (another 128-byte line): written by B, polled by A
template <int S, int L>
__global__ void pingpong(u64* flags, int iters, int partner, int* stuck, u64* clocks) {
  const int b = blockIdx.x;
  if ((b != 0 && b != partner) || threadIdx.x != 0) return;
  u64* mine = flags + (b == 0 ? 0 : 16);
  u64* theirs = flags + (b == 0 ? 16 : 0);
  const u64 t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (b == 0) st<S>(mine, (u64)i);
    unsigned spins = 0;
    while (ld<L>(theirs) != (u64)i) {
      if (++spins > 4000000u) { *stuck = 1; return; }
    }
    if (b != 0) st<S>(mine, (u64)i);
  }
  if (b == 0) clocks[0] = wall_clock64() - t0;
}

template <int S, int L> void run(const char* sn, const char* ln, u64* flags, int* stuck, u64* clocks) {
  const int iters = 20000;
  for (int partner : {8, 1}) {
    hipMemset(flags, 0, 256 * sizeof(u64));
    hipMemset(stuck, 0, sizeof(int));
    hipMemset(clocks, 0, sizeof(u64));
    hipLaunchKernelGGL((pingpong<S, L>), dim3(16), dim3(64), 0, 0, flags, iters, partner, stuck, clocks);
    hipDeviceSynchronize();
    int hs = 0; u64 hc = 0;
    hipMemcpy(&hs, stuck, sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(&hc, clocks, sizeof(u64), hipMemcpyDeviceToHost);
    // wall_clock64 ticks at 100 MHz
    if (hs) printf("store %-22s load %-26s %s: STUCK (never saw the partner's store)\n", sn, ln, partner == 8 ? "same XCD " : "other XCD");
    else printf("store %-22s load %-26s %s: %7.0f ns one way\n", sn, ln, partner == 8 ? "same XCD " : "other XCD", hc * 10.0 / (2.0 * iters));
  }
}

int main() {
  u64 *flags, *clocks; int* stuck;
  hipMalloc(&flags, 256 * sizeof(u64)); hipMalloc(&clocks, sizeof(u64)); hipMalloc(&stuck, sizeof(int));
#define RUN(S, SN, L, LN) run<S, L>(SN, LN, flags, stuck, clocks)
#define ALL_L(S, SN) RUN(S, SN, 0, "sc1 (agent)"); RUN(S, SN, 1, "sc0 sc1 (system)"); RUN(S, SN, 2, "sc0 (workgroup)"); \
                     RUN(S, SN, 3, "nt"); RUN(S, SN, 4, "atomic_or 0 (RMW in L2)"); RUN(S, SN, 5, "buffer_inv sc1 + plain")
  ALL_L(0, "sc1 (agent)");
  ALL_L(1, "sc0 sc1 (system)");
  ALL_L(2, "plain");
  ALL_L(3, "atomic_swap (RMW in L2)");
  ALL_L(4, "sc0 (workgroup)");
  return 0;
}
