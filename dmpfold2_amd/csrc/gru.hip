// gru_bidir: bidirectional multi-layer GRU along the sequence with batch 1 (hgru, reference
// network.py:190/225; coord_gru, network.py:211/253).  The vertical GRU is in vgru.hip.
// Gate maths follow ATen's gru_cell: r,z = sigmoid(gi+gh), n = tanh(gi_n + r*gh_n),
// h' = (h - n)*z + n.
#include "common.h"

namespace dmp {

// Gates.  Default: the device library's expf / tanhf and an IEEE division, as in rounds 1-3.  SEQ_LIBM_GATES=0 builds
// gate_sigmoid / gate_tanh (common.h: hardware exponential and reciprocal with the product's rounding error recovered
// and a polynomial for small tanh arguments).  Measured against the same recurrence in float64
// (tools/seq_gru_accuracy.py) the two are EQUALLY accurate (rms error 4.9e-8 / 6.0e-8 for hgru / coord_gru either
// way; torch's own float32 CPU GRU: 4.2e-8 / 5.3e-8; the plain hardware forms 5.8e-8 / 7.0e-8).  But the reference
// computes with a 1-ulp expf / tanhf too, so the library forms agree with it bit for bit in most gate evaluations,
// and on the expansive fixtures that shows: L=200, N=1000, 2 iterations against the oracle |dconf| 2.4e-5 (library)
// against 9.6e-5 .. 1.4e-4 (fast forms; the oracle's own thread-count spread: 3.9e-5); headline fixture 8.6e-4
// against 1.03e-3 A.  Parity first.  With one lane evaluating all six gates of its two units the library functions
// cost 0.55 us of a 1.84 us step (fast forms: 1.51 us); with the evaluations spread over six lanes (below) a step is
// 1.18 us with the library functions.
#ifndef SEQ_LIBM_GATES
#define SEQ_LIBM_GATES 1
#endif
#if SEQ_LIBM_GATES
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float sigmoidf_(float x) { return gate_sigmoid(x); }
__device__ __forceinline__ float tanhf_(float x) { return gate_tanh(x); }
#endif

// Weight-stationary cluster: each direction is run by SEQ_G workgroups; workgroup g keeps the
// 96 rows of W_hh that produce hidden units [32g, 32g+32) in registers (96 floats per lane)
// for the whole sequence.  Per time step a workgroup computes its 32 new state values,
// publishes them as 8-byte {epoch, value} granules (one agent-scope store each, the data is
// the flag) and gathers the other 224 by sweeping the granule array until every tag equals
// the step's epoch.  The protocol does not depend on where the workgroups run; the 8 workgroups of a
// direction are put on one XCD (block id % 8), and when they find themselves there (cluster_on_one_xcd,
// common.h) they publish with plain stores that stop in that XCD's L2: 1.29 instead of 2.78 us per step.
// Every wave gathers the state for itself into its own LDS copy (LDS operations of one wave execute in
// order): no workgroup barrier anywhere in the step loop.
constexpr int SEQ_G = 8;
typedef unsigned long long u64;
#ifndef SEQ_SHFL
#define SEQ_SHFL 0
#endif

struct SeqArgs {
  const float* G;        // [T][1536] input projections incl. b_ih (fwd | rev)
  const float* whh[2];   // [768][256]
  const float* bhh[2];   // [768]
  float* out;            // [T][512]
  u64* hx;               // [2 dir][2 parity][256] granules + [2 dir][2] placement header, zeroed before every launch
  int* abort_flag;       // set if a hand-off ever times out
  int T;
  int allow_local;       // 0: never the XCD-local publication
  int xcd0;              // the two directions run on XCDs xcd0, xcd0 + 1 (block id % 8)
};

// grid: 8 * SEQ_G blocks (only ids with (id % 8 - xcd0) % 8 < 2 work: that is the direction)   block: 256
#ifndef SEQ_PRIO
#define SEQ_PRIO 0
#endif
__global__ __launch_bounds__(256) void seq_gru_kernel(SeqArgs a) {
#if SEQ_PRIO
  __builtin_amdgcn_s_setprio(SEQ_PRIO);
#endif
  __shared__ __attribute__((aligned(16))) float hs[4][HID2];
  __shared__ int sh_local;
  const int dir = ((int)(blockIdx.x & 7) - a.xcd0) & 7, g = blockIdx.x >> 3;
  if (dir >= 2) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rgp = lane >> 4, kg = lane & 15;
  const int u0 = 32 * g + 8 * wave + 2 * rgp;            // this lane group's two hidden units
  const float* whh = a.whh[dir];
  float wreg[3][2][16];
  float bh[3][2];
#pragma unroll
  for (int gate = 0; gate < 3; ++gate)
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      const int row = gate * HID2 + u0 + uu;
      const float4* wr = reinterpret_cast<const float4*>(whh + (int64_t)row * HID2 + kg * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w4 = wr[q];
        wreg[gate][uu][q * 4 + 0] = w4.x; wreg[gate][uu][q * 4 + 1] = w4.y;
        wreg[gate][uu][q * 4 + 2] = w4.z; wreg[gate][uu][q * 4 + 3] = w4.w;
      }
      bh[gate][uu] = a.bhh[dir][row];
    }
  float* h = hs[wave];
#pragma unroll
  for (int q = 0; q < 4; ++q) h[lane + 64 * q] = 0.f;
  // do the SEQ_G workgroups of this direction share an XCD (common.h)?  Then granules are published with plain stores
  if (tid == 0) sh_local = cluster_on_one_xcd(a.hx + 4 * HID2 + 2 * dir, SEQ_G, a.allow_local != 0) ? 1 : 0;
  __syncthreads();
  const bool local = sh_local != 0;
  u64* hx_dir = a.hx + (int64_t)dir * 2 * HID2;
  // Gates, one evaluation per lane: after the row reduction all 16 lanes of a row hold the six sums of its two hidden
  // units, so lane kg < 6 evaluates gate kg >> 1 (r, z, n) of unit kg & 1 - the four sigmoids in ONE pass of the
  // (long) library function, then the two tanh, with the r / z / n values moved between the lanes by DPP row shifts -
  // instead of lane 0 evaluating all six one after the other.  Same functions on the same arguments: same bits.
  const int my_gate = kg >> 1, my_uu = kg & 1;
  const float bhsel = my_gate == 0 ? (my_uu ? bh[0][1] : bh[0][0]) : my_gate == 1 ? (my_uu ? bh[1][1] : bh[1][0])
                                                                                   : (my_uu ? bh[2][1] : bh[2][0]);
  float gin1 = 0.f;                                       // the one input projection this lane needs
  auto load_gi = [&](int step) {
    if (kg < 6 && step < a.T) {
      const int tn = dir ? (a.T - 1 - step) : step;
      gin1 = a.G[(int64_t)tn * 1536 + dir * 768 + u0 + my_gate * HID2 + my_uu];
    }
  };
  load_gi(0);
#define SEQ_T(i)
  for (int step = 0; step < a.T; ++step) {
    const int t = dir ? (a.T - 1 - step) : step;
    const unsigned epoch = (unsigned)step + 1u;
    u64* hx = hx_dir + (step & 1) * HID2;
    // input projection of this step for this lane's gate and unit (lanes kg < 6): loaded one step ahead - behind the
    // previous step's publication, while that step's granules were being awaited - so that its L2 latency is off the
    // step's critical path (round 3)
    const float gi1 = gin1;
    float hv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(&h[kg * 16 + q * 4]);
      hv[q * 4 + 0] = v.x; hv[q * 4 + 1] = v.y; hv[q * 4 + 2] = v.z; hv[q * 4 + 3] = v.w;
    }
    const float hp0 = h[u0], hp1 = h[u0 + 1];
    SEQ_T(0)
    float acc[3][2];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) sacc = fmaf(wreg[gate][uu][q], hv[q], sacc);
        acc[gate][uu] = sacc;
      }
    // sum over the 16 lanes of a row: the xor butterfly 8, 4, 2, 1 - as DPP row rotations (after the xor-8 step lanes
    // i and i ^ 8 agree, so "lane i + 4 mod 16" holds what lane i ^ 4 holds, and so on down: same operands, same sums
    // as __shfl_xor, without four dependent ds_bpermute round trips through the LDS pipeline per step)
#if SEQ_SHFL
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
      for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) acc[gate][uu] += __shfl_xor(acc[gate][uu], off, 16);
#else
#define SEQ_ROR(n)                                                                                                   \
    _Pragma("unroll") for (int gate = 0; gate < 3; ++gate) _Pragma("unroll") for (int uu = 0; uu < 2; ++uu)          \
        acc[gate][uu] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[gate][uu]), \
                                                                              0x120 + (n), 0xf, 0xf, false));
    SEQ_ROR(8) SEQ_ROR(4) SEQ_ROR(2) SEQ_ROR(1)
#undef SEQ_ROR
#endif
    SEQ_T(1)
    {
      const float accsel = my_gate == 0 ? (my_uu ? acc[0][1] : acc[0][0]) : my_gate == 1 ? (my_uu ? acc[1][1] : acc[1][0])
                                                                                         : (my_uu ? acc[2][1] : acc[2][0]);
      const float hsum = accsel + bhsel;
      const float sg = sigmoidf_(gi1 + hsum);                                  // lanes 0..3: r0, r1, z0, z1
      const float r_in = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sg), 0x114, 0xf, 0xf, true));   // row_shr:4: lanes 4, 5 get r0, r1
      const float ng = tanhf_(gi1 + r_in * hsum);                              // lanes 4, 5: n0, n1
      const float zg = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sg), 0x102, 0xf, 0xf, true));     // row_shl:2: lanes 0, 1 get z0, z1
      const float nn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ng), 0x104, 0xf, 0xf, true));     // row_shl:4: lanes 0, 1 get n0, n1
      if (kg < 2) {
        const float hp = my_uu ? hp1 : hp0;
        const float hn = (hp - nn) * zg + nn;
        a.out[(int64_t)t * 512 + dir * HID2 + u0 + my_uu] = hn;
        cluster_publish(&hx[u0 + my_uu], ((u64)epoch << 32) | (u64)__float_as_uint(hn), local);
      }
    }
    load_gi(step + 1);               // next step's input projections fly while this step's granules are gathered
    SEQ_T(2)
    {
      // sweep the 256 granules of this step (4 per lane) until all carry this epoch
      unsigned vals[4];
      bool dead = false;
      for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u64 x = __hip_atomic_load(&hx[lane + 64 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          vals[q] = (unsigned)x;
          ok = ok && ((unsigned)(x >> 32) == epoch);
        }
        if (__all(ok)) break;
        if (spins > 2000000u) { dead = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (dead) {                    // every wave of the cluster waits for the same granules: all of them leave
        if (lane == 0) atomicOr(a.abort_flag, 1);
        break;
      }
      __builtin_amdgcn_wave_barrier();   // every lane has read this step's h before it is overwritten
#pragma unroll
      for (int q = 0; q < 4; ++q) h[lane + 64 * q] = __uint_as_float(vals[q]);
      __builtin_amdgcn_wave_barrier();
      SEQ_T(3)
    }
  }
}

int gru_bidir(dmp_ctx* c, int which, const float* d_in, int T, float* d_out, hipStream_t s) {
  const int layers = which == 0 ? 2 : 3;
  const float* in = d_in;
  for (int l = 0; l < layers; ++l) {
    const GruDirW* w = which == 0 ? c->W.hgru[l] : c->W.cgru[l];
    {
      // input projections of both directions in one launch: G[t][dir 2][3H] = in[t] W_ih^T + b_ih
      GemmArgs g{};
      g.A = in; g.sam = w[0].nin; g.sak = 1;
      g.B = w[0].wihT_both; g.sbk = 1536; g.sbn = 1;
      g.C = c->seq_g; g.ldc = 1536;
      g.M = T; g.N = 1536; g.K = w[0].nin;
      g.alpha = 1.f; g.beta = 0.f; g.bias_n = w[0].bih_both;
      int rc = gemm_f32(g, s);
      if (rc) return rc;
    }
    float* out = (l == layers - 1) ? d_out : ((l & 1) ? c->seq_b : c->seq_a);
    SeqArgs a{};
    a.G = c->seq_g;
    a.whh[0] = w[0].whh; a.whh[1] = w[1].whh;
    a.bhh[0] = w[0].bhh; a.bhh[1] = w[1].bhh;
    a.out = out; a.T = T;
    a.hx = (u64*)c->seq_hx;
    a.abort_flag = c->seq_abort;
    a.allow_local = c->cluster_local;
    a.xcd0 = c->seq_xcd0;
    {
      CoResident guard(c, s, false);                   // not beside a persistent vertical-GRU launch (common.h)
      if (guard.status()) return guard.status();
      DMP_HIP(hipMemsetAsync(c->seq_hx, 0, sizeof(u64) * (2 * 2 * HID2 + 4), s));
      hipLaunchKernelGGL(seq_gru_kernel, dim3(8 * SEQ_G), dim3(256), 0, s, a);
      DMP_LAUNCH_CHECK();
      int rc = guard.done();
      if (rc) return rc;
    }
    in = out;
  }
  return DMP_OK;
}

}  // namespace dmp
