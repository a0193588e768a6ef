// Training-side slice (reference train.py:318-344 runs autograd through every ResNet_Block,
// network.py:85-103): the backward of a residual block as two entry points that mirror the forward's two kernels,
//   dmp_block_norm_scse_residual_bwd   d(out) -> d(u), d(gamma, beta, cSE fc, sSE conv)      (this file, second half)
//   dmp_block_conv5x5_maxout_bwd       d(u)   -> d(x), d(W), d(b)                            (this file, first half)
// the residual branch is the identity: d(block input) = d(x) of the convolution + d(out); and the 1x1 head's backward
// (dmp_head_conv_bwd), so that a test can chain the sixteen blocks and the head of net.resnet.
// EVALUATION MODE ONLY: in training mode the reference's ResNet_Block applies Dropout(0.2) / Dropout2d(0.2) ahead of
// layer1 (network.py:85-103); neither the masks nor their backward exist here.
// float32 arithmetic (exact-f32 matrix cores: an MFMA chain is an fmaf chain) with float64 reductions; none of this is on
// the inference path.  One workspace per context, sized for the context's max_L at the first call (no reallocation
// when L changes between calls).
#include "common.h"

#pragma clang fp contract(off)

namespace dmp {

typedef float tr_f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------
// First half (network.py:25-31): backward of a block's convolution + maxout,  u[g] = max_q (conv(x, W)[4g+q] + b[4g+q]).
//   dz[4g+q] = du[g] where q is the FIRST maximal channel of the quadruple (torch.max), 0 elsewhere
//   db[o]        = sum_p dz[o][p]
//   dx[c][p]     = sum_{o,tap} W[o][c,tap] dz[o][p - tap]   (dgrad)
//   dW[o][c,tap] = sum_p dz[o][p] x[c][p + tap]             (wgrad)
// Round 5: implicit GEMMs on the float32 matrix cores in the forward kernel's tile structure, no patch matrix (round 4:
// im2col (3200 + 1024) L^2 floats = 1.5 GB at L = 300 and three library-style GEMMs).  The routed gradient dz is never
// materialised either: it is 75 % structural zeros, so it travels in GATHERED form - du (128 planes) + the winner of
// every quadruple (one byte per maxout channel and pixel, written by the forward kernel's <IDX> instantiation) - and is
// expanded to its dense (channel pair x halo tile) form only in LDS, where the MFMAs read it.  (The FLOPs stay dense:
// which of a quadruple's four weight rows a pixel uses differs from pixel to pixel, and an MFMA's A operand is shared by
// its 32 pixel columns.  The 1-of-4 pattern is a special case of the 2:4 sparsity v_smfmac accepts on its A operand -
// half the dense work, f16 only - not built.)
//   1. winners   conv5x5_maxout_kernel<true> (trunk.hip) on the padded input: the forward in float32 + idx
//   2. db        per maxout channel four float64 sums over the pixels
//   3. dgrad     conv5x5_dgrad_kernel: "a 5x5 convolution 512 -> 128 with flipped taps" - M = 64 of the 128 input
//                channels per workgroup, N = one 16x16 pixel tile, K = 256 stages of (2 convolution channels x 25 taps);
//                722 workgroups at L = 300, four per CU
//   4. wgrad     conv5x5_wgrad_kernel: K = the PIXELS; workgroup = 32 convolution channels x 32 input channels x 25 taps
//                (five waves, one tap row each: five accumulators), looping over 8 x 16 pixel tiles whose dz tile and
//                x halo tile it stages in LDS (next tile requested under this tile's MFMAs); the tiles are split over
//                WG_KSPLIT workgroups whose partial sums are reduced in fixed order (deterministic)
// Workspace: the padded input (128 P^2 floats: 1.06 activation tensors), the winners when the caller did not keep them
// (128 L^2 bytes: a quarter), and an L-independent 85 MB - the flipped weight pack (6.8 MB) and the wgrad partial sums
// (WG_KSPLIT x 6.5 MB): 145 MB at L = 300, 165 MB at the 350 crop (round 4: 1.5 GB / 2.1 GB).
// ---------------------------------------------------------------------------------------
constexpr int DG_MB = 2;                          // 32-row MFMA blocks per workgroup
constexpr int DG_M = 32 * DG_MB;                  // input channels (c) per workgroup: 64 of the 128
constexpr int DG_MSPLIT = CW / DG_M;              // 2
constexpr int DG_NCHUNK = 4 * CW / CONV_CC;       // 256 stages of two convolution channels
constexpr int DG_WSLAB = 3328;                    // floats per stage: 25 taps x 2 x 64 = 3200, padded to 13 KB of LDS-DMA
constexpr int DG_HALO = CONV_TILE + 4, DG_IP = 24;
constexpr int DG_IN = CONV_CC * DG_HALO * DG_IP;  // 960 floats
constexpr int WG_KSPLIT = 12;                     // 64 x 12 = 768 workgroups = three per CU
constexpr int WG_TY = 8, WG_TX = 16, WG_HY = WG_TY + 4, WG_HX = WG_TX + 4;
constexpr int WG_LP = 33;                         // LDS pitch of the wgrad tiles: [position][32 channels | 1 pad]
constexpr int DB_CHUNKS = 16;                     // pixel chunks of the bias gradient's first stage
constexpr int ST_OCH = 32;                        // stem backward: stem channels per panel of the outer-product weight gradient
constexpr int ST_CCH = 64;                        // ... and sequence channels per panel of the mat1d gradient

struct BwdWs { float* norm; float* xpad; float* wd; float* part; uint8_t* idx; };

static int64_t norm_ws_floats(int L) {
  const int64_t LL = (int64_t)L * L, nch = cdiv64(LL, 4096);
  return ((2 * LL + 512 + 1024 + 3) & ~(int64_t)3) + 2 * (int64_t)CW * nch * 8 + 16;
}

// the one workspace of the context, sized for max_L (both halves and every L <= max_L use the same carving)
static int bwd_workspace(dmp_ctx* c, BwdWs* w) {
  const int64_t L = c->max_L, P = act_pitch(c->max_L);
  // the norm region also holds the bias gradient's first-stage partials (conv5x5_maxout_bwd: CW x DB_CHUNKS x 4 doubles),
  // which are larger than norm_ws_floats below max_L = 73 (they ran into xpad at max_L = 64)
  const int64_t n_norm = (std::max<int64_t>(norm_ws_floats(c->max_L), 2 * (int64_t)CW * DB_CHUNKS * 4) + 3) & ~(int64_t)3;
  const int64_t n_xpad = (int64_t)CW * P * P;
  const int64_t n_wd = (int64_t)DG_MSPLIT * DG_NCHUNK * DG_WSLAB, n_part = (int64_t)WG_KSPLIT * 512 * 3200;
  const int64_t n_idx = ((int64_t)CW * L * L + 3) / 4;
  // the stem's backward (below) uses the space behind the norm region differently: DZ (384 L^2) + one GEMM panel
  const int64_t n_stem = (int64_t)STEM_OUT * L * L + std::max<int64_t>((int64_t)ST_OCH * L * WIDTH, (int64_t)ST_CCH * L * L);
  const int64_t need = n_norm + std::max(n_xpad + n_wd + n_part + n_idx, n_stem);
  if (!c->bwd_ws) {
    DMP_HIP(hipMalloc((void**)&c->bwd_ws, sizeof(float) * (size_t)need));
    c->bwd_ws_floats = need;
    c->bwd_w_block = 0;
  }
  w->norm = c->bwd_ws;
  w->xpad = w->norm + n_norm;
  w->wd = w->xpad + n_xpad;
  w->part = w->wd + n_wd;
  w->idx = reinterpret_cast<uint8_t*>(w->part + n_part);
  return DMP_OK;
}

// Wd[ms][chunk][tap][cc][m] = W[o = 2 chunk + cc][c = 64 ms + m][24 - tap] from the exact-f32 forward pack
// [split = o / 128][c / 2][tap][c % 2][o % 128] (the raw tensors are not kept after dmp_weights_finalize)
__global__ __launch_bounds__(256) void bwd_pack_dgrad_weights_kernel(const float* __restrict__ wpack, float* __restrict__ wd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= DG_MSPLIT * DG_NCHUNK * DG_WSLAB) return;
  const int r = i % DG_WSLAB, chunk = (i / DG_WSLAB) % DG_NCHUNK, ms = i / (DG_WSLAB * DG_NCHUNK);
  float v = 0.f;
  if (r < 25 * CONV_CC * DG_M) {
    const int m = r % DG_M, cc = (r / DG_M) % CONV_CC, tap = r / (DG_M * CONV_CC);
    const int o = 2 * chunk + cc, ch = ms * DG_M + m, t = 24 - tap;
    v = wpack[((((int64_t)(o >> 7) * (CW / CONV_CC) + ch / CONV_CC) * 25 + t) * CONV_CC + ch % CONV_CC) * 128 + (o & 127)];
  }
  wd[i] = v;
}

// db[4g + q] = sum over the pixels won by q of du[g]: float64 partial sums per pixel chunk, then in chunk order
// grid: (DB_CHUNKS, 128)   block: 256      part[g][chunk][4]
__global__ __launch_bounds__(256) void bwd_bias_kernel(const float* __restrict__ du, const uint8_t* __restrict__ idx, int LL,
                                                       double* __restrict__ part) {
  __shared__ double red[4][4];
  const int g = blockIdx.y, ch = blockIdx.x;
  const int p0 = (int)((int64_t)LL * ch / DB_CHUNKS), p1 = (int)((int64_t)LL * (ch + 1) / DB_CHUNKS);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int p = p0 + threadIdx.x; p < p1; p += 256) {
    const double d = (double)du[(int64_t)g * LL + p];
    const int w = idx[(int64_t)g * LL + p];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += w == q ? d : 0.0;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = wave_sum_f64(acc[q]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int q = 0; q < 4; ++q) red[threadIdx.x >> 6][q] = acc[q];
  __syncthreads();
  if (threadIdx.x < 4)
    part[((int64_t)g * DB_CHUNKS + ch) * 4 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
// grid: 2   block: 256 (thread = convolution channel o)
__global__ __launch_bounds__(256) void bwd_bias_finish_kernel(const double* __restrict__ part, float* __restrict__ db) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  double s = 0.0;
  for (int ch = 0; ch < DB_CHUNKS; ++ch) s += part[((int64_t)(o >> 2) * DB_CHUNKS + ch) * 4 + (o & 3)];
  db[o] = (float)s;
}

// grid: round_up(tiles^2 x DG_MSPLIT, 8)   block: 256   (the forward kernel's structure: conv5x5_maxout_kernel, trunk.hip)
__global__ __launch_bounds__(256, 4) void conv5x5_dgrad_kernel(const float* __restrict__ du, const uint8_t* __restrict__ idx,
                                                               const float* __restrict__ wd, int L, int tiles, int nwork,
                                                               float* __restrict__ dx) {
  __shared__ __attribute__((aligned(16))) float smem[2 * DG_WSLAB + 2 * DG_IN];
  float (*w_lds)[DG_WSLAB] = reinterpret_cast<float (*)[DG_WSLAB]>(smem);
  float (*in_lds)[DG_IN] = reinterpret_cast<float (*)[DG_IN]>(smem + 2 * DG_WSLAB);
  // XCD-aware remap: block b runs on XCD b % 8; each XCD gets a contiguous range of work items
  const int id = blockIdx.x, per = gridDim.x >> 3;
  const int work = (id & 7) * per + (id >> 3);
  if (work >= nwork) return;
  const int tile = work / DG_MSPLIT, ms = work % DG_MSPLIT;
  const int ty0 = (tile / tiles) * CONV_TILE, tx0 = (tile % tiles) * CONV_TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, li = lane & 31;
  const int64_t LL = (int64_t)L * L;

  // staging of the (2 convolution channels x 20 x 20) halo tile of the ROUTED gradient: slot = (cc, yy, xx); both
  // channels of a stage belong to one maxout channel g (o = 4 g + q), so the slot loads du[g] and the winner at its
  // pixel and keeps the value where the winner is its own q.  Every load is unconditional (clamped offset).
  int in_off[4], in_dst[4], in_q[4];
  bool in_ok[4], in_img[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = tid + e * 256;                          // < 800 valid
    in_ok[e] = i < CONV_CC * DG_HALO * DG_HALO;
    const int ic = in_ok[e] ? i : 0;
    const int cc = ic / (DG_HALO * DG_HALO), rem = ic % (DG_HALO * DG_HALO);
    const int yy = rem / DG_HALO, xx = rem % DG_HALO;
    const int gy = ty0 + yy - 2, gx = tx0 + xx - 2;
    in_img[e] = gy >= 0 && gy < L && gx >= 0 && gx < L;
    in_off[e] = in_img[e] ? gy * L + gx : 0;
    in_dst[e] = cc * DG_HALO * DG_IP + yy * DG_IP + xx;
    in_q[e] = cc;
  }
  const float4* wsrc = reinterpret_cast<const float4*>(wd + (int64_t)ms * DG_NCHUNK * DG_WSLAB);
  float ireg[4];
  int wreg[4];
  auto prefetch = [&](int chunk) {
    const float4* ws = wsrc + (int64_t)chunk * (DG_WSLAB / 4);
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    float* dst = w_lds[chunk & 1] + (size_t)wave * 256;
    // the weight slab of the stage by LDS-DMA: 13 wave-instructions of 1 KB (lane-linear destination)
#pragma unroll
    for (int e = 0; e < 3; ++e)
      __builtin_amdgcn_global_load_lds((gptr_t)(ws + tid + e * 256), (lptr_t)(dst + e * 1024), 16, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_global_load_lds((gptr_t)(ws + tid + 3 * 256), (lptr_t)(dst + 3 * 1024), 16, 0, 0);
    const int64_t gb = (int64_t)(chunk >> 1) * LL;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ireg[e] = du[gb + in_off[e]];
      wreg[e] = idx[gb + in_off[e]];
    }
  };
  auto commit = [&](int buf, int chunk) {
    const int q0 = (chunk & 1) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (in_ok[e]) in_lds[buf][in_dst[e]] = (in_img[e] && wreg[e] == q0 + in_q[e]) ? ireg[e] : 0.f;
  };

  int b_off[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int nb = 2 * wave + q;
    const int y = (nb >> 1) * 4 + (li >> 3), x = (nb & 1) * 8 + (li & 7);
    b_off[q] = kk * DG_HALO * DG_IP + y * DG_IP + x;
  }
  const int a_off = kk * DG_M + li;
  tr_f32x16 acc[DG_MB][2];
#pragma unroll
  for (int mb = 0; mb < DG_MB; ++mb)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][q][r] = 0.f;

  prefetch(0);
  commit(0, 0);
  __syncthreads();
  for (int chunk = 0; chunk < DG_NCHUNK; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < DG_NCHUNK) prefetch(chunk + 1);
    const float* wl = w_lds[buf] + a_off;
    const float* il = in_lds[buf];
#pragma unroll
    for (int tap = 0; tap < 25; ++tap) {
      const int dy = tap / 5, dxx = tap % 5;
      float a[DG_MB], b[2];
#pragma unroll
      for (int mb = 0; mb < DG_MB; ++mb) a[mb] = wl[tap * CONV_CC * DG_M + mb * 32];
#pragma unroll
      for (int q = 0; q < 2; ++q) b[q] = il[b_off[q] + dy * DG_IP + dxx];
#pragma unroll
      for (int mb = 0; mb < DG_MB; ++mb)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[mb][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb], b[q], acc[mb][q], 0, 0, 0);
    }
    if (chunk + 1 < DG_NCHUNK) commit(buf ^ 1, chunk + 1);
    __syncthreads();
  }
  // accumulator register r of lane (kk, li): channel 64 ms + 32 mb + 8 (r / 4) + 4 kk + (r % 4), pixel li of patch nb
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int nb = 2 * wave + q;
    const int y = ty0 + (nb >> 1) * 4 + (li >> 3), x = tx0 + (nb & 1) * 8 + (li & 7);
    if (y < L && x < L) {
#pragma unroll
      for (int mb = 0; mb < DG_MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch = ms * DG_M + mb * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
          dx[(int64_t)ch * LL + (int64_t)y * L + x] = acc[mb][q][r];
        }
    }
  }
}

// grid: 64 x WG_KSPLIT   block: 256      part[ks][o 512][tap 25][c 128]
// Four waves, balanced: wave w owns taps 6 w .. 6 w + 5 (six accumulators over every k) and a quarter of the k steps of
// the 25th tap (a seventh accumulator; the four quarters are added in wave order once, at the end).  Three workgroups per
// CU (48.6 KB of LDS each) = 12 waves = exactly three per SIMD.  (First forms: five waves, one tap row each - 3.46 ms at
// L = 300 with two workgroups per CU, one SIMD carrying twice the others' MFMAs; 3.15 ms with three per CU.)  The next
// tile's operands are requested before this tile's MFMAs and written to LDS behind them: the x halo tile as 16-byte
// loads of the padded planes (rows of 20 = 5 x float4, aligned: P and the tile origin are multiples of 4), the dz tile as
// du + winner byte, expanded at the LDS write.  LDS layout [position][32 channels | 1 pad]: the 32 lanes of an MFMA
// operand read 32 consecutive words.
__global__ __launch_bounds__(256, 3) void conv5x5_wgrad_kernel(const float* __restrict__ xpad, const float* __restrict__ du,
                                                               const uint8_t* __restrict__ idx, int L, int P, int tx_tiles,
                                                               int ntiles, float* __restrict__ part) {
  __shared__ float a_lds[WG_TY * WG_TX * WG_LP];          // dz tile  [pixel 128][o 32 | pad]
  __shared__ float b_lds[WG_HY * WG_HX * WG_LP];          // x halo   [position 240][c 32 | pad]
  const int ks = blockIdx.x >> 6, rb = blockIdx.x & 63, ob = rb >> 2, cb = rb & 3;
  const int t_lo = (int)((int64_t)ntiles * ks / WG_KSPLIT), t_hi = (int)((int64_t)ntiles * (ks + 1) / WG_KSPLIT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 5, li = lane & 31;
  const int64_t LL = (int64_t)L * L, PP = (int64_t)P * P;
  int toff[6];                                           // LDS word offset of this wave's taps: (dy 20 + dx) positions
#pragma unroll
  for (int d = 0; d < 6; ++d) toff[d] = (((6 * w + d) / 5) * WG_HX + (6 * w + d) % 5) * WG_LP;
  // staging slots: up to eight float4 of the halo tile (i = tid + 256 e < 32 x 12 x 5 = 1920: channel i / 60, row, float4
  // of the row) and four (maxout channel, pixel) items of dz (i < 1024).  The slot arithmetic is redone at every use -
  // divisions by constants under 400 MFMAs per wave and tile - instead of being kept in registers (seven accumulators
  // = 112 registers, 40 of the prefetch; the budget of three waves per SIMD is 170).
  const float* xb = xpad + (int64_t)cb * 32 * PP;
  const float* dub = du + (int64_t)ob * 8 * LL;
  const uint8_t* idb = idx + (int64_t)ob * 8 * LL;
  float4 breg[8];
  float areg[4];
  int wreg[4];                                           // the winner, or -1 for a pixel outside the image
  auto prefetch = [&](int t) {
    const int ty0 = (t / tx_tiles) * WG_TY, tx0 = (t % tx_tiles) * WG_TX;
    const float* xt = xb + (int64_t)ty0 * P + tx0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int i = tid + e * 256;
      asm volatile("" : "+v"(i));                        // (keeps the slot arithmetic inside the loop: see above)
      const int ic = i < 1920 ? i : 0;
      const int cl = ic / 60, rem = ic % 60, row = rem / 5, f4 = rem % 5;
      breg[e] = *reinterpret_cast<const float4*>(xt + ((int64_t)cl * PP + (int64_t)row * P + 4 * f4));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int i = tid + e * 256;
      asm volatile("" : "+v"(i));
      const int g = i >> 7, p = i & 127;
      const int y = ty0 + (p >> 4), x = tx0 + (p & 15);
      const bool real = y < L && x < L;
      const int64_t off = (int64_t)g * LL + (real ? y * L + x : 0);
      areg[e] = dub[off];
      const int win = idb[off];
      wreg[e] = real ? win : -1;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int i = tid + e * 256;
      asm volatile("" : "+v"(i));
      if (i < 1920) {
        const int cl = i / 60, rem = i % 60, row = rem / 5, f4 = rem % 5;
        float* d = b_lds + (row * WG_HX + 4 * f4) * WG_LP + cl;
        d[0] = breg[e].x;
        d[WG_LP] = breg[e].y;
        d[2 * WG_LP] = breg[e].z;
        d[3 * WG_LP] = breg[e].w;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int i = tid + e * 256;
      asm volatile("" : "+v"(i));
      float* d = a_lds + (i & 127) * WG_LP + (i >> 7) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = wreg[e] == q ? areg[e] : 0.f;
    }
  };
  tr_f32x16 acc[7];
#pragma unroll
  for (int d = 0; d < 7; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
  if (t_lo < t_hi) prefetch(t_lo);
  for (int t = t_lo; t < t_hi; ++t) {
    __syncthreads();                                     // the previous tile's MFMAs are done with the LDS tiles
    commit();
    __syncthreads();
    if (t + 1 < t_hi) prefetch(t + 1);
#pragma unroll 1
    for (int seg = 0; seg < 4; ++seg) {
      const bool mine = seg == w;                        // this wave's quarter of the 25th tap (dy = dx = 4)
#pragma unroll 4
      for (int s = 16 * seg; s < 16 * seg + 16; ++s) {
        const int p = 2 * s + kk;
        const float a = a_lds[p * WG_LP + li];
        const float* bp = b_lds + ((p >> 4) * WG_HX + (p & 15)) * WG_LP + li;
#pragma unroll
        for (int d = 0; d < 6; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[toff[d]], acc[d], 0, 0, 0);
        if (mine) acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[(4 * WG_HX + 4) * WG_LP], acc[6], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < 6; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
      part[(((int64_t)ks * 512 + o) * 25 + 6 * w + d) * CW + cb * 32 + li] = acc[d][r];
    }
  // the 25th tap: the four waves' quarters, added in wave order (through the halo tile's LDS)
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) b_lds[(w * 16 + r) * 64 + lane] = acc[6][r];
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = 4 * w + rr;
    const float v = ((b_lds[(0 * 16 + r) * 64 + lane] + b_lds[(1 * 16 + r) * 64 + lane]) + b_lds[(2 * 16 + r) * 64 + lane]) +
                    b_lds[(3 * 16 + r) * 64 + lane];
    const int o = ob * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
    part[(((int64_t)ks * 512 + o) * 25 + 24) * CW + cb * 32 + li] = v;
  }
}

// dW[o][c 25 + tap] = sum_ks part[ks][o][tap][c], ks ascending          grid: 512 x 25 x 128 / 256   block: 256
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;           // (o 25 + tap) 128 + c
  const int c = i & 127, ot = i >> 7, tap = ot % 25, o = ot / 25;
  float s = part[i];
#pragma unroll
  for (int ks = 1; ks < WG_KSPLIT; ++ks) s += part[(int64_t)ks * 512 * 3200 + i];
  dw[(int64_t)o * 3200 + c * 25 + tap] = s;
}

// The forward a trainer runs: the block's convolution + maxout in float32 AND the winner of every quadruple (what
// autograd saves for the backward of torch.max, network.py:31): u (128 x L x L), idx (128 x L x L bytes, 0..3).
int conv5x5_maxout_fwd_winners(dmp_ctx* c, int block, const float* d_x, int L, float* d_u, uint8_t* d_idx, hipStream_t s) {
  BwdWs w;
  int rc;
  if ((rc = bwd_workspace(c, &w))) return rc;
  if ((rc = act_pad(d_x, L, w.xpad, s))) return rc;
  return conv5x5_maxout_winners(c, block, w.xpad, L, d_u, d_idx, s);
}

// d_idx: the winners the forward saved (conv5x5_maxout_fwd_winners), or null: the forward is run again here
int conv5x5_maxout_bwd(dmp_ctx* c, int block, const float* d_x, const float* d_du, const uint8_t* d_idx, int L, float* d_dx,
                       float* d_dw, float* d_db, hipStream_t s) {
  const BlockW& B = c->W.blk[block - 1];
  const int LL = L * L, P = act_pitch(L), tiles = act_tiles(L);
  BwdWs w;
  int rc;
  if ((rc = bwd_workspace(c, &w))) return rc;
  if (c->bwd_w_block != block) {                          // the flipped weight pack of this block
    hipLaunchKernelGGL(bwd_pack_dgrad_weights_kernel, dim3(cdiv(DG_MSPLIT * DG_NCHUNK * DG_WSLAB, 256)), dim3(256), 0, s,
                       B.wpack, w.wd);
    DMP_LAUNCH_CHECK();
    c->bwd_w_block = block;
  }
  // 1. the padded input (wgrad's halo tiles) and, unless the caller kept them, the winners: the forward again, in
  //    float32 (its maxout output goes to the context's scratch plane c->u)
  if ((rc = act_pad(d_x, L, w.xpad, s))) return rc;
  if (!d_idx) {
    if ((rc = conv5x5_maxout_winners(c, block, w.xpad, L, c->u, w.idx, s))) return rc;
    d_idx = w.idx;
  }
  // 2. bias gradient
  hipLaunchKernelGGL(bwd_bias_kernel, dim3(DB_CHUNKS, CW), dim3(256), 0, s, d_du, d_idx, LL, reinterpret_cast<double*>(w.norm));
  hipLaunchKernelGGL(bwd_bias_finish_kernel, dim3(2), dim3(256), 0, s, reinterpret_cast<const double*>(w.norm), d_db);
  DMP_LAUNCH_CHECK();
  // 3. input gradient
  const int nwork = tiles * tiles * DG_MSPLIT;
  hipLaunchKernelGGL(conv5x5_dgrad_kernel, dim3(round_up(nwork, 8)), dim3(256), 0, s, d_du, d_idx, w.wd, L, tiles, nwork, d_dx);
  DMP_LAUNCH_CHECK();
  // 4. weight gradient
  const int ty = cdiv(L, WG_TY), tx = cdiv(L, WG_TX);
  hipLaunchKernelGGL(conv5x5_wgrad_kernel, dim3(64 * WG_KSPLIT), dim3(256), 0, s, w.xpad, d_du, d_idx, L, P, tx, ty * tx, w.part);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(512 * 25 * CW / 256), dim3(256), 0, s, w.part, d_dw);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// The 1x1 head (network.py:207: Conv2d(128 -> 2, k 1)), backward: with G = d(head output) [2][L][L] and x its input,
//   dx[c][p] = sum_h W[h][c] G[h][p];   dW[h][c] = sum_p G[h][p] x[c][p];   db[h] = sum_p G[h][p]
// dparams: [dW 2 x 128][db 2]
__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ g, const float* __restrict__ hw, int LL,
                                                          float* __restrict__ dx) {
  const int c = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= LL) return;
  dx[(int64_t)c * LL + p] = fmaf(hw[CW + c], g[LL + p], hw[c] * g[p]);
}
// grid: 130 (c < 128: the two weight gradients of channel c; 128, 129: the bias gradients)   block: 256
__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* __restrict__ g, const float* __restrict__ x, int LL,
                                                          float* __restrict__ dparams) {
  __shared__ double red[4][2];
  const int c = blockIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int p = threadIdx.x; p < LL; p += 256) {
    const double xv = c < CW ? (double)x[(int64_t)c * LL + p] : 1.0;
    a0 += (double)g[p] * xv;
    a1 += (double)g[LL + p] * xv;
  }
  a0 = wave_sum_f64(a0);
  a1 = wave_sum_f64(a1);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a0; red[threadIdx.x >> 6][1] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s0 = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0], s1 = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    if (c < CW) { dparams[c] = (float)s0; dparams[CW + c] = (float)s1; }
    else if (c == CW) { dparams[2 * CW] = (float)s0; dparams[2 * CW + 1] = (float)s1; }
  }
}

int head_conv_bwd(dmp_ctx* c, const float* d_x, const float* d_g, int L, float* d_dx, float* d_dparams, hipStream_t s) {
  const int LL = L * L;
  hipLaunchKernelGGL(head_bwd_dx_kernel, dim3(cdiv(LL, 256), CW), dim3(256), 0, s, d_g, c->W.head_w, LL, d_dx);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(head_bwd_dw_kernel, dim3(CW + 1), dim3(256), 0, s, d_g, d_x, LL, d_dparams);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}


// ---------------------------------------------------------------------------------------
// Second half (network.py:32, 36-83, 99-101): backward of InstanceNorm + scSE + residual add,
//   v = gamma uh + beta, uh = (u - mean) rstd;   out = v (y + s) + x,   y_c = sigma(W2 relu(W1 m)), m = mean_p v,
//   s_p = sigma(a_p), a_p = sum_c ws_c v_cp + bs.
// With D = d(out):
//   d(s)_p = sum_c D v,  d(a) = d(s) s (1 - s);   d(y)_c = sum_p D v -> the two small fc layers -> d(m)
//   d(v) = D (y + s) + ws d(a) + d(m) / P
//   d(beta) = sum_p d(v), d(gamma) = sum_p d(v) uh,  d(u) = gamma rstd (d(v) - mean d(v) - uh mean(d(v) uh))
// (the d(m) / P term is constant per channel: it reaches d(beta) and cancels in d(u), as m = beta does not depend on u).
// Every channel sum is linear in seven pixel sums, so the activations are read three times (pixel gates, channel
// sums, d(u)) and d(v) is never stored:
//   T1 = sum D, T2 = sum D uh, T3 = sum D s, T4 = sum D s uh, T5 = sum d(a) uh, T6 = sum d(a), T7 = sum uh
// ---------------------------------------------------------------------------------------
constexpr int TB_CHUNK = 4096;                           // pixels per workgroup of the reduction kernels
constexpr int TB_NSUM = 8;

template <int NS>
__device__ __forceinline__ void tb_block_sums(double (&v)[NS], double* __restrict__ out) {
  __shared__ double red[4][NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) v[k] = wave_sum_f64(v[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NS; ++k) red[wave][k] = v[k];
  __syncthreads();
  if (threadIdx.x < NS) out[threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// grid: (chunks, 128)   block: 256
__global__ __launch_bounds__(256) void tb_stats_kernel(const float* __restrict__ u, int LL, double* __restrict__ part) {
  const int c = blockIdx.y, nch = gridDim.x;
  const int p1 = min(LL, (int)(blockIdx.x + 1) * TB_CHUNK);
  double v[2] = {0.0, 0.0};
  for (int p = blockIdx.x * TB_CHUNK + threadIdx.x; p < p1; p += 256) {
    const double x = (double)u[(int64_t)c * LL + p];
    v[0] += x;
    v[1] += x * x;
  }
  tb_block_sums<2>(v, part + ((int64_t)c * nch + blockIdx.x) * 2);
}

// coef[c] = {alpha = gamma rstd, beta' = beta - mean alpha, mean, rstd}: the forward's arithmetic (norm_coeff_kernel)
// grid: 1   block: 128
__global__ __launch_bounds__(128) void tb_coef_kernel(const double* __restrict__ part, int nch, double count,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ coef) {
  const int c = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int t = 0; t < nch; ++t) { s1 += part[((int64_t)c * nch + t) * 2]; s2 += part[((int64_t)c * nch + t) * 2 + 1]; }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  const float invstd = (float)(1.0 / sqrt(var + 1e-5));
  const float alpha = invstd * gamma[c];
  coef[c * 4 + 0] = alpha;
  coef[c * 4 + 1] = beta[c] - (float)mean * alpha;
  coef[c * 4 + 2] = (float)mean;
  coef[c * 4 + 3] = invstd;
}

// one thread = one pixel: the spatial gate and its gradient.  The channel sum is formed as the forward kernel forms
// it, (q0 + q1) + (q2 + q3) with each quarter an fmaf chain in channel order.
// grid: ceil(LL / 256)   block: 256
__global__ __launch_bounds__(256) void tb_pixel_gates_kernel(const float* __restrict__ u, const float* __restrict__ dout,
                                                             const float* __restrict__ coef,
                                                             const float* __restrict__ sse_w, float sse_b, int LL,
                                                             float* __restrict__ sg, float* __restrict__ da) {
  __shared__ float sh_a[CW], sh_b[CW], sh_w[CW];
  if (threadIdx.x < CW) {
    sh_a[threadIdx.x] = coef[threadIdx.x * 4];
    sh_b[threadIdx.x] = coef[threadIdx.x * 4 + 1];
    sh_w[threadIdx.x] = sse_w[threadIdx.x];
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= LL) return;
  float dot[4], ds[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dot[q] = 0.f;
    ds[q] = 0.f;
    for (int i = 0; i < CW / 4; ++i) {
      const int c = q * (CW / 4) + i;
      const float v = u[(int64_t)c * LL + p] * sh_a[c] + sh_b[c];
      dot[q] = fmaf(sh_w[c], v, dot[q]);
      ds[q] = fmaf(dout[(int64_t)c * LL + p], v, ds[q]);
    }
  }
  const float s = 1.0f / (1.0f + expf(-(((dot[0] + dot[1]) + (dot[2] + dot[3])) + sse_b)));
  sg[p] = s;
  da[p] = ((ds[0] + ds[1]) + (ds[2] + ds[3])) * (s * (1.0f - s));
}

// grid: (chunks, 128)   block: 256
__global__ __launch_bounds__(256) void tb_channel_sums_kernel(const float* __restrict__ u, const float* __restrict__ dout,
                                                              const float* __restrict__ coef,
                                                              const float* __restrict__ sg, const float* __restrict__ da,
                                                              int LL, double* __restrict__ part) {
  const int c = blockIdx.y, nch = gridDim.x;
  const float mean = coef[c * 4 + 2], rstd = coef[c * 4 + 3];
  const int p1 = min(LL, (int)(blockIdx.x + 1) * TB_CHUNK);
  double v[TB_NSUM] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = blockIdx.x * TB_CHUNK + threadIdx.x; p < p1; p += 256) {
    const double uh = (double)((u[(int64_t)c * LL + p] - mean) * rstd);
    const double d = (double)dout[(int64_t)c * LL + p], s = (double)sg[p], a = (double)da[p];
    v[0] += d;
    v[1] += d * uh;
    v[2] += d * s;
    v[3] += d * s * uh;
    v[4] += a * uh;
    v[5] += a;
    v[6] += uh;
  }
  tb_block_sums<TB_NSUM>(v, part + ((int64_t)c * nch + blockIdx.x) * TB_NSUM);
}

// The per-channel end of the reductions, the two fc layers of the channel gate forwards (as api.hip computes the
// constant gate at dmp_weights_finalize) and backwards, the parameter gradients, and the coefficients of the last
// pass: k[c] = {gamma rstd, y, ws, (d(m) - S1) / P, S2 / P, mean, rstd}.
// dparams: [d gamma 128][d beta 128][d fc.0.weight 8 x 128][d fc.2.weight 128 x 8][d sSE weight 128][d sSE bias 1]
// grid: 1   block: 128 (thread = channel)
__global__ __launch_bounds__(128) void tb_finish_kernel(const double* __restrict__ part, int nch, double count,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ sse_w, const float* __restrict__ fc,
                                                        const float* __restrict__ coef, float* __restrict__ kco,
                                                        float* __restrict__ dparams) {
  __shared__ double sh_m[CW], sh_dz2[CW], sh_h[8], sh_z1[8], sh_dz1[8];
  const int c = threadIdx.x;
  const float* w1 = fc;                                  // [8][128]
  const float* w2 = fc + 8 * CW;                         // [128][8]
  double T[TB_NSUM];
  for (int k = 0; k < TB_NSUM; ++k) T[k] = 0.0;
  for (int t = 0; t < nch; ++t)
    for (int k = 0; k < TB_NSUM; ++k) T[k] += part[((int64_t)c * nch + t) * TB_NSUM + k];
  const double g = (double)gamma[c], b = (double)beta[c], ws = (double)sse_w[c];
  sh_m[c] = b + g * (T[6] / count);                      // mean_p v
  __syncthreads();
  if (c < 8) {
    double a = 0.0;
    for (int q = 0; q < CW; ++q) a += (double)w1[c * CW + q] * sh_m[q];
    sh_z1[c] = a;
    sh_h[c] = a > 0.0 ? a : 0.0;
  }
  __syncthreads();
  double z2 = 0.0;
  for (int r = 0; r < 8; ++r) z2 += (double)w2[c * 8 + r] * sh_h[r];
  const double y = 1.0 / (1.0 + exp(-z2));
  const double dy = g * T[1] + b * T[0];                 // sum_p D v
  const double dz2 = dy * y * (1.0 - y);
  sh_dz2[c] = dz2;
  for (int r = 0; r < 8; ++r) dparams[2 * CW + 8 * CW + c * 8 + r] = (float)(dz2 * sh_h[r]);
  __syncthreads();
  if (c < 8) {
    double a = 0.0;
    for (int q = 0; q < CW; ++q) a += (double)w2[q * 8 + c] * sh_dz2[q];
    sh_dz1[c] = sh_z1[c] > 0.0 ? a : 0.0;
  }
  __syncthreads();
  double dm = 0.0;
  for (int r = 0; r < 8; ++r) {
    dm += (double)w1[r * CW + c] * sh_dz1[r];
    dparams[2 * CW + r * CW + c] = (float)(sh_dz1[r] * sh_m[c]);
  }
  const double S1 = y * T[0] + T[2] + ws * T[5] + dm;
  const double S2 = y * T[1] + T[3] + ws * T[4] + (dm / count) * T[6];
  dparams[c] = (float)S2;                                // d gamma
  dparams[CW + c] = (float)S1;                           // d beta
  dparams[2 * CW + 16 * CW + c] = (float)(g * T[4] + b * T[5]);      // d sSE weight = sum_p d(a) v
  if (c == 0) dparams[2 * CW + 16 * CW + CW] = (float)T[5];          // d sSE bias (T6 is the same in every channel)
  kco[c * 8 + 0] = coef[c * 4 + 0];
  kco[c * 8 + 1] = (float)y;
  kco[c * 8 + 2] = sse_w[c];
  kco[c * 8 + 3] = (float)((dm - S1) / count);
  kco[c * 8 + 4] = (float)(S2 / count);
  kco[c * 8 + 5] = coef[c * 4 + 2];
  kco[c * 8 + 6] = coef[c * 4 + 3];
  kco[c * 8 + 7] = 0.f;
}

// grid: (ceil(LL / 256), 128)   block: 256
__global__ __launch_bounds__(256) void tb_norm_input_kernel(const float* __restrict__ u, const float* __restrict__ dout,
                                                            const float* __restrict__ kco, const float* __restrict__ sg,
                                                            const float* __restrict__ da, int LL, float* __restrict__ du) {
  const int c = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= LL) return;
  const float k0 = kco[c * 8], y = kco[c * 8 + 1], ws = kco[c * 8 + 2], cst = kco[c * 8 + 3], k2 = kco[c * 8 + 4];
  const float uh = (u[(int64_t)c * LL + p] - kco[c * 8 + 5]) * kco[c * 8 + 6];
  const float dv = fmaf(dout[(int64_t)c * LL + p], y + sg[p], ws * da[p]);
  du[(int64_t)c * LL + p] = k0 * ((dv + cst) - uh * k2);
}

int norm_scse_residual_bwd(dmp_ctx* c, int block, const float* d_u, const float* d_dout, int L, float* d_du,
                           float* d_dparams, hipStream_t s) {
  const BlockW& B = c->W.blk[block - 1];
  const int LL = L * L, nch = cdiv(LL, TB_CHUNK);
  // workspace: gate planes 2 LL floats | coef 512 | kco 1024 | partial sums 128 nch 8 doubles
  const int64_t fl = 2 * (int64_t)LL + 512 + 1024;
  int rc;
  BwdWs w;
  if ((rc = bwd_workspace(c, &w))) return rc;
  float* sg = w.norm;
  float* da = sg + LL;
  float* coef = da + LL;
  float* kco = coef + 512;
  double* part = reinterpret_cast<double*>(w.norm + ((fl + 3) & ~(int64_t)3));
  hipLaunchKernelGGL(tb_stats_kernel, dim3(nch, CW), dim3(256), 0, s, d_u, LL, part);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_coef_kernel, dim3(1), dim3(CW), 0, s, part, nch, (double)LL, B.gamma, B.beta, coef);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_pixel_gates_kernel, dim3(cdiv(LL, 256)), dim3(256), 0, s, d_u, d_dout, coef, B.sse_w, B.sse_b, LL,
                     sg, da);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_channel_sums_kernel, dim3(nch, CW), dim3(256), 0, s, d_u, d_dout, coef, sg, da, LL, part);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_finish_kernel, dim3(1), dim3(CW), 0, s, part, nch, (double)LL, B.gamma, B.beta, B.sse_w, B.cse_fc,
                     coef, kco, d_dparams);
  DMP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tb_norm_input_kernel, dim3(cdiv(LL, 256), CW), dim3(256), 0, s, d_u, d_dout, kco, sg, da, LL, d_du);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// ---------------------------------------------------------------------------------------
// The stem, resnet[0] = Maxout2d(955 -> 128, pool 3, kernel 1) + InstanceNorm (network.py:194, 12-34), forwards with
// the winners and backwards.  Its 955-channel input is never materialised, on the prediction path or here:
//   channels 0..511    mat1d[c][i] mat1d[c][j]           (network.py:226-227; mat1d = the sequence trunk's output)
//   channels 512..953  the covariance planes             (the context's c->planes, written by dmp_stem_static)
//   channel  954       the distance / seed channel
// With dz = the gradient at the 1x1 convolution's output (the routed d(u), 384 x L^2, dense: it is only 3 x d(u)):
//   dW[o][c < 512]    = m_c^T DZ_o m_c          -> panels T = DZ_o M^T on the matrix cores (gemm_f32), then a row dot
//   dW[o][512 + q]    = sum_p dz[o][p] planes[q][p]                  -> one GEMM (384 x 442 x L^2)
//   dW[o][954], db[o] = sum_p dz[o][p] dmap[p], sum_p dz[o][p]
//   d mat1d[c][i]     = sum_j (H_c[i][j] + H_c[j][i]) mat1d[c][j],  H = W[:, :512]^T DZ  -> panels of 64 channels (gemm_f32)
//   d(u) from d(y): the InstanceNorm's backward, d gamma, d beta.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_maxout_winners_kernel(const float* __restrict__ z0, const float* __restrict__ wd,
                                                                  const float* __restrict__ dmap, int64_t LL,
                                                                  float* __restrict__ u, uint8_t* __restrict__ idx) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (p >= LL) return;
  const float d = dmap[p];
  float v = z0[(int64_t)(3 * g) * LL + p] + wd[3 * g] * d;
  int win = 0;
#pragma unroll
  for (int q = 1; q < 3; ++q) {
    const float t = z0[(int64_t)(3 * g + q) * LL + p] + wd[3 * g + q] * d;
    if (t > v) { v = t; win = q; }                        // strict: the first maximal channel, as torch.max
  }
  u[(int64_t)g * LL + p] = v;
  idx[(int64_t)g * LL + p] = (uint8_t)win;
}

int stem_maxout_fwd_winners(dmp_ctx* c, const float* d_z0, const float* d_dmap, int L, float* d_u, uint8_t* d_idx,
                            hipStream_t s) {
  const int64_t LL = (int64_t)L * L;
  hipLaunchKernelGGL(stem_maxout_winners_kernel, dim3((unsigned)cdiv64(LL, 256), CW), dim3(256), 0, s, d_z0, c->W.stem_wd,
                     d_dmap, LL, d_u, d_idx);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// per channel and pixel chunk: sum dy, sum dy uh          grid: (chunks, 128)   block: 256
__global__ __launch_bounds__(256) void stem_in_sums_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                                           const float* __restrict__ coef, int LL, double* __restrict__ part) {
  const int c = blockIdx.y, nch = gridDim.x;
  const float mean = coef[c * 4 + 2], rstd = coef[c * 4 + 3];
  const int p1 = min(LL, (int)(blockIdx.x + 1) * 4096);
  double v[2] = {0.0, 0.0};
  for (int p = blockIdx.x * 4096 + threadIdx.x; p < p1; p += 256) {
    const double d = (double)dy[(int64_t)c * LL + p];
    v[0] += d;
    v[1] += d * (double)((u[(int64_t)c * LL + p] - mean) * rstd);
  }
  __shared__ double red[4][2];
  v[0] = wave_sum_f64(v[0]);
  v[1] = wave_sum_f64(v[1]);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = v[0]; red[threadIdx.x >> 6][1] = v[1]; }
  __syncthreads();
  if (threadIdx.x < 2)
    part[((int64_t)c * nch + blockIdx.x) * 2 + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
// kco[c] = {gamma rstd, S1 / P, S2 / P, mean, rstd}; dparams[384 + c] = d gamma, [512 + c] = d beta      grid: 1, block: 128
__global__ __launch_bounds__(128) void stem_in_finish_kernel(const double* __restrict__ part, int nch, double count,
                                                             const float* __restrict__ coef, float* __restrict__ kco,
                                                             float* __restrict__ dparams) {
  const int c = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int t = 0; t < nch; ++t) { s1 += part[((int64_t)c * nch + t) * 2]; s2 += part[((int64_t)c * nch + t) * 2 + 1]; }
  dparams[STEM_OUT + c] = (float)s2;
  dparams[STEM_OUT + CW + c] = (float)s1;
  kco[c * 8 + 0] = coef[c * 4 + 0];
  kco[c * 8 + 1] = (float)(s1 / count);
  kco[c * 8 + 2] = (float)(s2 / count);
  kco[c * 8 + 3] = coef[c * 4 + 2];
  kco[c * 8 + 4] = coef[c * 4 + 3];
}
// d(u) of the InstanceNorm and, routed by the winners, the dense dz (the two losers of a triple get 0)
// grid: (ceil(LL / 256), 128)   block: 256
__global__ __launch_bounds__(256) void stem_route_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                                         const uint8_t* __restrict__ idx, const float* __restrict__ kco,
                                                         int LL, float* __restrict__ dz) {
  const int g = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= LL) return;
  const float uh = (u[(int64_t)g * LL + p] - kco[g * 8 + 3]) * kco[g * 8 + 4];
  const float du = kco[g * 8] * ((dy[(int64_t)g * LL + p] - kco[g * 8 + 1]) - uh * kco[g * 8 + 2]);
  const int w = idx[(int64_t)g * LL + p];
#pragma unroll
  for (int q = 0; q < 3; ++q) dz[(int64_t)(3 * g + q) * LL + p] = w == q ? du : 0.f;
}
// db[o] = sum_p dz[o][p], dW[o][954] = sum_p dz[o][p] dmap[p]        grid: 384   block: 256
__global__ __launch_bounds__(256) void stem_bias_dist_kernel(const float* __restrict__ dz, const float* __restrict__ dmap, int LL,
                                                             float* __restrict__ dparams, float* __restrict__ dw) {
  __shared__ double red[4][2];
  const int o = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (int p = threadIdx.x; p < LL; p += 256) {
    const double v = (double)dz[(int64_t)o * LL + p];
    a += v;
    b += v * (double)dmap[p];
  }
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dparams[o] = (float)(((red[0][0] + red[1][0]) + red[2][0]) + red[3][0]);
    dw[(int64_t)o * STEM_IN + (STEM_IN - 1)] = (float)(((red[0][1] + red[1][1]) + red[2][1]) + red[3][1]);
  }
}
// dW[o0 + o][c] = sum_i mat1d[c][i] T[(o L + i)][c]        grid: (2, ST_OCH)   block: 256 (thread = sequence channel c)
__global__ __launch_bounds__(256) void stem_outer_dw_kernel(const float* __restrict__ T, const float* __restrict__ mat1d, int L,
                                                            int o0, float* __restrict__ dw) {
  const int c = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  double acc = 0.0;
  for (int i = 0; i < L; ++i) acc += (double)mat1d[(int64_t)c * L + i] * (double)T[((int64_t)o * L + i) * WIDTH + c];
  dw[(int64_t)(o0 + o) * STEM_IN + c] = (float)acc;
}
// d mat1d[c0 + cl][i] = sum_j (H[cl][i L + j] + H[cl][j L + i]) mat1d[c0 + cl][j]      grid: (L, ST_CCH)   block: 64
__global__ __launch_bounds__(64) void stem_dmat1d_kernel(const float* __restrict__ H, const float* __restrict__ mat1d, int L,
                                                         int c0, float* __restrict__ dmat1d) {
  const int i = blockIdx.x, cl = blockIdx.y;
  const int64_t LL = (int64_t)L * L;
  double acc = 0.0;
  for (int j = threadIdx.x; j < L; j += 64)
    acc += ((double)H[cl * LL + (int64_t)i * L + j] + (double)H[cl * LL + (int64_t)j * L + i]) * (double)mat1d[(int64_t)(c0 + cl) * L + j];
  acc = wave_sum_f64(acc);
  if (threadIdx.x == 0) dmat1d[(int64_t)(c0 + cl) * L + i] = (float)acc;
}

// d_dw: 384 x 955 (resnet.0.lin.weight.grad); d_dparams: [lin.bias.grad 384][norm.weight.grad 128][norm.bias.grad 128];
// d_dmat1d: 512 x L.  dmp_stem_static must have run for this target on this context (the covariance planes).
int stem_bwd(dmp_ctx* c, const float* d_u, const uint8_t* d_idx, const float* d_dy, const float* d_mat1d,
             const float* d_dmap, int L, float* d_dw, float* d_dparams, float* d_dmat1d, hipStream_t s) {
  const Weights& W = c->W;
  const int LL = L * L, nch = cdiv(LL, 4096);
  BwdWs w;
  int rc;
  if ((rc = bwd_workspace(c, &w))) return rc;
  c->bwd_w_block = 0;                                     // the space behind the norm region is about to be overwritten
  float* coef = w.norm;                                   // [128][4]
  float* kco = coef + 512;                                // [128][8]
  double* part = reinterpret_cast<double*>(w.norm + 2048);
  float* dz = w.xpad;                                     // [384][LL]
  float* panel = dz + (int64_t)STEM_OUT * LL;
  // InstanceNorm: statistics of u (as the forward's), the two sums of its backward, d(u) -> routed dz
  hipLaunchKernelGGL(tb_stats_kernel, dim3(nch, CW), dim3(256), 0, s, d_u, LL, part);
  hipLaunchKernelGGL(tb_coef_kernel, dim3(1), dim3(CW), 0, s, part, nch, (double)LL, W.stem_gamma, W.stem_beta, coef);
  hipLaunchKernelGGL(stem_in_sums_kernel, dim3(nch, CW), dim3(256), 0, s, d_u, d_dy, coef, LL, part);
  hipLaunchKernelGGL(stem_in_finish_kernel, dim3(1), dim3(CW), 0, s, part, nch, (double)LL, coef, kco, d_dparams);
  hipLaunchKernelGGL(stem_route_kernel, dim3(cdiv(LL, 256), CW), dim3(256), 0, s, d_u, d_dy, d_idx, kco, LL, dz);
  hipLaunchKernelGGL(stem_bias_dist_kernel, dim3(STEM_OUT), dim3(256), 0, s, dz, d_dmap, LL, d_dparams, d_dw);
  DMP_LAUNCH_CHECK();
  GemmArgs g{};
  g.alpha = 1.f; g.beta = 0.f; g.bias_n = nullptr; g.lower_tiles = false;
  // covariance channels 512 .. 953: dW[:, 512 + q] = DZ planes^T
  g.A = dz; g.sam = LL; g.sak = 1; g.B = c->planes; g.sbk = 1; g.sbn = LL; g.C = d_dw + WIDTH; g.ldc = STEM_IN;
  g.M = STEM_OUT; g.N = NS * NS + 1; g.K = LL;
  if ((rc = gemm_f32(g, s))) return rc;
  // outer-product channels: panels of ST_OCH stem channels, T[(o, i)][c] = sum_j dz[o][i][j] mat1d[c][j]
  for (int o0 = 0; o0 < STEM_OUT; o0 += ST_OCH) {
    g.A = dz + (int64_t)o0 * LL; g.sam = L; g.sak = 1; g.B = d_mat1d; g.sbk = 1; g.sbn = L; g.C = panel; g.ldc = WIDTH;
    g.M = ST_OCH * L; g.N = WIDTH; g.K = L;
    if ((rc = gemm_f32(g, s))) return rc;
    hipLaunchKernelGGL(stem_outer_dw_kernel, dim3(WIDTH / 256, ST_OCH), dim3(256), 0, s, panel, d_mat1d, L, o0, d_dw);
    DMP_LAUNCH_CHECK();
  }
  // d mat1d: panels of ST_CCH sequence channels, H[cl][p] = sum_o W[o][c0 + cl] dz[o][p]
  for (int c0 = 0; c0 < WIDTH; c0 += ST_CCH) {
    g.A = W.stemT + (int64_t)c0 * STEM_OUT; g.sam = STEM_OUT; g.sak = 1; g.B = dz; g.sbk = LL; g.sbn = 1; g.C = panel; g.ldc = LL;
    g.M = ST_CCH; g.N = LL; g.K = STEM_OUT;
    if ((rc = gemm_f32(g, s))) return rc;
    hipLaunchKernelGGL(stem_dmat1d_kernel, dim3(L, ST_CCH), dim3(64), 0, s, panel, d_mat1d, L, c0, d_dmat1d);
    DMP_LAUNCH_CHECK();
  }
  return DMP_OK;
}


}  // namespace dmp
