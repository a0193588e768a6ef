// Sequence reweighting (reference predict.py:32-37) as integer byte-compare kernels.
//   cnt[n,m] = #{l : min(a_nl,20) == min(a_ml,20)}
//   w[n]     = 1 / #{m : float(cnt[n,m]) > float32(L*0.8)}
// N^2*L byte compares; the alignment (<= 3000 x L bytes) lives in L2, tiles of 64x64 row
// pairs are staged through LDS as packed words and compared four residues per instruction.
#include "common.h"

namespace dmp {

// codes clamped to <= 20 and packed 4 per word; tail bytes are 0xFF in every row, so each
// pair gains exactly (4*Lw - L) spurious matches, removed again in the count kernel.
__global__ void msa_pack_kernel(const uint8_t* __restrict__ msa, int N, int L, int Lw,
                                uint32_t* __restrict__ words, int* __restrict__ fault) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)N * Lw) return;
  const int n = idx / Lw, wd = idx % Lw;
  uint32_t v = 0;
  for (int b = 0; b < 4; ++b) {
    const int l = wd * 4 + b;
    uint32_t c = 0xFF;
    if (l < L) {
      c = msa[(int64_t)n * L + l];
      // the reference's 22-row embedding raises IndexError on such a code (network.py:223)
      if (c > 21) atomicOr(fault, DMP_FAULT_BAD_CODE);
      c = c > 20 ? 20 : c;
    }
    v |= c << (8 * b);
  }
  words[idx] = v;
}

__device__ __forceinline__ int eq_bytes(uint32_t a, uint32_t b) {
  const uint32_t x = a ^ b;                                   // zero byte <=> equal residue
  const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;    // bit 7 set <=> byte non-zero
  return __popc(~t & 0x80808080u);
}

constexpr int MT = 64;   // rows per tile side
constexpr int KW = 16;   // words per K chunk

__global__ __launch_bounds__(256) void msa_count_kernel(const uint32_t* __restrict__ words, int N,
                                                        int Lw, int pad_matches, float id_min,
                                                        int* __restrict__ nbr) {
  __shared__ uint32_t An[MT][KW + 1];
  __shared__ uint32_t Bm[MT][KW + 1];
  const int tid = threadIdx.x;
  const int tn = tid >> 4, tm = tid & 15;
  const int n0 = blockIdx.y * MT, m0 = blockIdx.x * MT;
  int cnt[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) cnt[i][j] = 0;

  for (int k0 = 0; k0 < Lw; k0 += KW) {
    for (int e = tid; e < MT * KW; e += 256) {
      const int r = e / KW, kw = e % KW;
      const int gk = k0 + kw;
      // rows beyond N and words beyond Lw get patterns that never match anything
      An[r][kw] = (n0 + r < N && gk < Lw) ? words[(int64_t)(n0 + r) * Lw + gk] : 0xFEFEFEFEu;
      Bm[r][kw] = (m0 + r < N && gk < Lw) ? words[(int64_t)(m0 + r) * Lw + gk] : 0xFDFDFDFDu;
    }
    __syncthreads();
#pragma unroll 4
    for (int kw = 0; kw < KW; ++kw) {
      uint32_t a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = An[tn * 4 + i][kw];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bm[tm + 16 * j][kw];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) cnt[i][j] += eq_bytes(a[i], b[j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int hits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + tm + 16 * j;
      // the reference compares a float32 count with the float32 threshold (predict.py:35)
      if (m < N && (float)(cnt[i][j] - pad_matches) > id_min) ++hits;
    }
    // the 16 threads sharing tn are 16 adjacent lanes
    for (int off = 8; off > 0; off >>= 1) hits += __shfl_xor(hits, off, 16);
    const int n = n0 + tn * 4 + i;
    if (tm == 0 && n < N && hits) atomicAdd(&nbr[n], hits);
  }
}

__global__ void msa_recip_kernel(const int* __restrict__ nbr, int N, float* __restrict__ w) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) w[n] = 1.0f / (float)nbr[n];
}

int msa_weights(dmp_ctx* c, const uint8_t* d_msa, int N, int L, float* d_w, hipStream_t s) {
  const int Lw = cdiv(L, 4);
  const int64_t nw = (int64_t)N * Lw;
  hipLaunchKernelGGL(msa_pack_kernel, dim3((unsigned)cdiv64(nw, 256)), dim3(256), 0, s, d_msa, N,
                     L, Lw, c->msa_words, c->seq_abort);
  DMP_LAUNCH_CHECK();
  DMP_HIP(hipMemsetAsync(c->nbr_count, 0, sizeof(int) * N, s));
  const float id_min = (float)((double)L * 0.8);
  hipLaunchKernelGGL(msa_count_kernel, dim3(cdiv(N, MT), cdiv(N, MT)), dim3(256), 0, s,
                     c->msa_words, N, Lw, 4 * Lw - L, id_min, c->nbr_count);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(msa_recip_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, c->nbr_count, N, d_w);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

}  // namespace dmp
