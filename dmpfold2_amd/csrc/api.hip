// extern "C" surface of libdmpfold_hip.so (see include/dmpfold_hip.h): context and weight
// management, host-side residue encoding, stage-level entry points and the fused dmp_predict.
#include "common.h"
#include <atomic>
#include "conv_bf16.h"
#include "conv_f16.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>

namespace dmp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  return DMP_ERR_HIP;
}

template <typename T>
static int dev_alloc(std::vector<void*>& pool, int64_t& bytes, T** out, int64_t count) {
  void* p = nullptr;
  const size_t sz = sizeof(T) * (size_t)(count > 0 ? count : 1);
  hipError_t e = hipMalloc(&p, sz);
  if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
  pool.push_back(p);
  bytes += (int64_t)sz;
  *out = (T*)p;
  return DMP_OK;
}

static int upload(std::vector<void*>& pool, int64_t& bytes, float** out, const std::vector<float>& h) {
  int rc = dev_alloc(pool, bytes, out, (int64_t)h.size());
  if (rc) return rc;
  DMP_HIP(hipMemcpy(*out, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  return DMP_OK;
}

struct KeySpec { std::string key; std::vector<int64_t> shape; };

static std::vector<KeySpec> weight_spec() {
  std::vector<KeySpec> s;
  s.push_back({"embed.weight", {22, 22}});
  auto gru = [&](const std::string& p, int nin, int hid, int layers, bool bidir) {
    for (int l = 0; l < layers; ++l) {
      const int lin = l == 0 ? nin : hid * (bidir ? 2 : 1);
      for (int d = 0; d < (bidir ? 2 : 1); ++d) {
        const std::string sfx = std::to_string(l) + (d ? "_reverse" : "");
        s.push_back({p + ".weight_ih_l" + sfx, {3 * hid, lin}});
        s.push_back({p + ".weight_hh_l" + sfx, {3 * hid, hid}});
        s.push_back({p + ".bias_ih_l" + sfx, {3 * hid}});
        s.push_back({p + ".bias_hh_l" + sfx, {3 * hid}});
      }
    }
  };
  gru("vgru", 22, 512, 2, false);
  gru("hgru", 512, 256, 2, true);
  s.push_back({"resnet.0.lin.weight", {STEM_OUT, STEM_IN, 1, 1}});
  s.push_back({"resnet.0.lin.bias", {STEM_OUT}});
  s.push_back({"resnet.0.norm.weight", {CW}});
  s.push_back({"resnet.0.norm.bias", {CW}});
  for (int k = 1; k <= NBLOCK; ++k) {
    const std::string p = "resnet." + std::to_string(k);
    s.push_back({p + ".layer1.lin.weight", {4 * CW, CW, 5, 5}});
    s.push_back({p + ".layer1.lin.bias", {4 * CW}});
    s.push_back({p + ".layer1.norm.weight", {CW}});
    s.push_back({p + ".layer1.norm.bias", {CW}});
    s.push_back({p + ".scSE.cSE.fc.0.weight", {CW / 16, CW}});
    s.push_back({p + ".scSE.cSE.fc.2.weight", {CW, CW / 16}});
    s.push_back({p + ".scSE.sSE.conv.weight", {1, CW, 1, 1}});
    s.push_back({p + ".scSE.sSE.conv.bias", {1}});
  }
  s.push_back({"resnet.17.weight", {2, CW, 1, 1}});
  s.push_back({"resnet.17.bias", {2}});
  gru("coord_gru", 520, 256, 3, true);
  s.push_back({"coord_fc.weight", {3, 512}});
  return s;
}

static std::vector<float> transposed(const std::vector<float>& w, int rows, int cols, int pad_rows = 0) {
  // w is [rows][cols]; returns [cols + pad_rows][rows]
  std::vector<float> t((size_t)(cols + pad_rows) * rows, 0.f);
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = w[(size_t)r * cols + c];
  return t;
}

// Packed weights are reference counted: the owner's buffers are freed when the last context using them lets go.
static void release_weights(dmp_ctx* c) {
  Weights& W = c->W;
  if (W.shared) {
    if (W.shared.use_count() == 1)
      for (void* p : *W.shared) (void)hipFree(p);
    W.shared.reset();
  }
  for (void* p : W.allocs) (void)hipFree(p);
  W.allocs.clear();
  for (auto& kv : c->vgru_graphs) (void)hipGraphExecDestroy((hipGraphExec_t)kv.second);   // they hold weight pointers
  c->vgru_graphs.clear();
  W.ready = false;
}

static int pack_weights(dmp_ctx* c) {
  Weights& W = c->W;
  auto& H = W.host;
  int64_t& bytes = c->bytes;
  auto& pool = W.allocs;
  int rc;
  // vertical GRU
  auto pieces = [&](const std::vector<float>& w, int K, int Kpad, float scale, uint16_t** out) -> int {
    // w is [3*512][K] (rows r | z | n); packed [piece 2][gate 3][Kpad/8][512][8] f16 pieces of scale*w
    const int KQ = Kpad / 8;
    std::vector<uint16_t> q((size_t)2 * 3 * KQ * 512 * 8, 0);
    for (int g = 0; g < 3; ++g)
      for (int j = 0; j < 512; ++j)
        for (int k = 0; k < K; ++k) {
          uint16_t p2[2];
          split2_f16(scale * w[((size_t)g * 512 + j) * K + k], p2);
          for (int p = 0; p < 2; ++p)
            q[((((size_t)p * 3 + g) * KQ + k / 8) * 512 + j) * 8 + k % 8] = p2[p];
        }
    int r = dev_alloc(pool, bytes, out, (int64_t)q.size());
    if (r) return r;
    DMP_HIP(hipMemcpy(*out, q.data(), sizeof(uint16_t) * q.size(), hipMemcpyHostToDevice));
    return DMP_OK;
  };
  // The kernel feeds layer 0 the one-hot residue code, so the embedding (network.py:188, 223: a
  // frozen identity in the reference, but part of the state_dict) is folded into the input weights:
  // W_ih x = W_ih embed[code] = (W_ih embed^T)[:, code].  Exact for the identity.
  std::vector<float> wi0_folded;
  {
    const auto& wi = H["vgru.weight_ih_l0"];     // [1536][22]
    const auto& emb = H["embed.weight"];         // [22 codes][22]
    wi0_folded.resize(wi.size());
    for (int j = 0; j < 3 * 512; ++j)
      for (int code = 0; code < 22; ++code) {
        double acc = 0.0;
        for (int k = 0; k < 22; ++k) acc += (double)wi[(size_t)j * 22 + k] * (double)emb[(size_t)code * 22 + k];
        wi0_folded[(size_t)j * 22 + code] = (float)acc;
      }
  }
  for (int l = 0; l < 2; ++l) {
    const auto& wi = l == 0 ? wi0_folded : H["vgru.weight_ih_l" + std::to_string(l)];
    const auto& wh = H["vgru.weight_hh_l" + std::to_string(l)];
    // one power-of-two scale per layer: the input and the recurrent products share accumulators
    const float scale = std::fmin(conv_weight_scale_f16(wi.data(), wi.size()),
                                  conv_weight_scale_f16(wh.data(), wh.size()));
    W.v_inv_scale[l] = 1.0f / (scale * VGRU_STATE_SCALE);
    if ((rc = pieces(wi, l == 0 ? 22 : 512, l == 0 ? 32 : 512, scale, &W.v_wx[l]))) return rc;
    if ((rc = pieces(wh, 512, 512, scale, &W.v_wh[l]))) return rc;
  }
  {
    // the same weights as float32 operands of v_mfma_f32_16x16x4_f32 (vgru_f32.hip): one 16-byte load = the four k of a
    // k quad for one hidden row
    auto quads = [&](const std::vector<float>& w, float** out) -> int {     // w is [3*512][512]
      std::vector<float> q((size_t)3 * 128 * 512 * 4);
      for (int g = 0; g < 3; ++g)
        for (int j = 0; j < 512; ++j)
          for (int k = 0; k < 512; ++k)
            q[(((size_t)g * 128 + k / 4) * 512 + j) * 4 + k % 4] = w[((size_t)g * 512 + j) * 512 + k];
      return upload(pool, bytes, out, q);
    };
    if ((rc = quads(H["vgru.weight_hh_l0"], &W.v_f32[0]))) return rc;
    if ((rc = quads(H["vgru.weight_ih_l1"], &W.v_f32[1]))) return rc;
    if ((rc = quads(H["vgru.weight_hh_l1"], &W.v_f32[2]))) return rc;
    std::vector<float> x0((size_t)3 * 22 * 512);
    for (int g = 0; g < 3; ++g)
      for (int code = 0; code < 22; ++code)
        for (int j = 0; j < 512; ++j) x0[((size_t)g * 22 + code) * 512 + j] = wi0_folded[((size_t)g * 512 + j) * 22 + code];
    if ((rc = upload(pool, bytes, &W.v_wx0f, x0))) return rc;
  }
  for (int l = 0; l < 2; ++l) {
    const auto& bi = H["vgru.bias_ih_l" + std::to_string(l)];
    const auto& bh = H["vgru.bias_hh_l" + std::to_string(l)];
    std::vector<float> b(4 * 512);
    for (int j = 0; j < 512; ++j) {
      b[j] = bi[j] + bh[j];
      b[512 + j] = bi[512 + j] + bh[512 + j];
      b[1024 + j] = bi[1024 + j];
      b[1536 + j] = bh[1024 + j];
    }
    if ((rc = upload(pool, bytes, l == 0 ? &W.v_b0 : &W.v_b1, b))) return rc;
  }
  // sequence GRUs
  auto seq = [&](const std::string& p, int layers, int nin0, GruDirW (*dst)[2]) -> int {
    for (int l = 0; l < layers; ++l)
      for (int d = 0; d < 2; ++d) {
        const std::string sfx = std::to_string(l) + (d ? "_reverse" : "");
        const int nin = l == 0 ? nin0 : 512;
        GruDirW& g = dst[l][d];
        g.nin = nin;
        int r;
        if ((r = upload(pool, bytes, &g.whh, H[p + ".weight_hh_l" + sfx]))) return r;
        if ((r = upload(pool, bytes, &g.bhh, H[p + ".bias_hh_l" + sfx]))) return r;
      }
    for (int l = 0; l < layers; ++l) {
      const int nin = l == 0 ? nin0 : 512;
      const std::string a = p + ".weight_ih_l" + std::to_string(l), b = p + ".bias_ih_l" + std::to_string(l);
      const auto& wf = H[a];
      const auto& wr = H[a + "_reverse"];
      std::vector<float> both((size_t)nin * 1536), bias(1536);
      for (int k = 0; k < nin; ++k)
        for (int j = 0; j < 768; ++j) {
          both[(size_t)k * 1536 + j] = wf[(size_t)j * nin + k];
          both[(size_t)k * 1536 + 768 + j] = wr[(size_t)j * nin + k];
        }
      for (int j = 0; j < 768; ++j) { bias[j] = H[b][j]; bias[768 + j] = H[b + "_reverse"][j]; }
      int r;
      if ((r = upload(pool, bytes, &dst[l][0].wihT_both, both))) return r;
      if ((r = upload(pool, bytes, &dst[l][0].bih_both, bias))) return r;
    }
    return DMP_OK;
  };
  if ((rc = seq("hgru", 2, 512, W.hgru))) return rc;
  if ((rc = seq("coord_gru", 3, 520, W.cgru))) return rc;
  if ((rc = upload(pool, bytes, &W.fc, H["coord_fc.weight"]))) return rc;
  // stem
  {
    const auto& w = H["resnet.0.lin.weight"];   // [384][955]
    if ((rc = upload(pool, bytes, &W.stemT, transposed(w, STEM_OUT, STEM_IN)))) return rc;
    std::vector<float> wd(STEM_OUT);
    for (int o = 0; o < STEM_OUT; ++o) wd[o] = w[(size_t)o * STEM_IN + (STEM_IN - 1)];
    if ((rc = upload(pool, bytes, &W.stem_wd, wd))) return rc;
    if ((rc = upload(pool, bytes, &W.stem_b, H["resnet.0.lin.bias"]))) return rc;
    if ((rc = upload(pool, bytes, &W.stem_gamma, H["resnet.0.norm.weight"]))) return rc;
    if ((rc = upload(pool, bytes, &W.stem_beta, H["resnet.0.norm.bias"]))) return rc;
  }
  // Activation scales of the split-f16 convolution (conv_f16.h).  The input of block k is the residual stream
  // x_{k-1} = x_{k-2} + y (cSE + sSE) with y = InstanceNorm output (per channel: mean beta, deviation |gamma|) and both
  // gates in (0, 1), so |x_{k-1}| <= B_{k-1} = B_{k-2} + 2 max_c(|beta_c| + R |gamma_c|) as long as no normalised
  // value exceeds R deviations (R = 16; a plane of L^2 near-Gaussian values reaches 4-6).  The pieces are taken of
  // 2^e x with the largest e that keeps B 2^e <= 32768: the bulk of the activations then has normal low pieces
  // whether the weights put the trunk at 1e-3 or at 1e4, the device-side range check (|2^e x| < 60000) keeps its
  // meaning, and a trunk that would have left the f16 range unscaled is scaled DOWN instead of faulting.
  {
    auto reach = [&](const std::vector<float>& g, const std::vector<float>& b) {
      float m = 0.f;
      for (size_t q = 0; q < g.size(); ++q) m = std::fmax(m, std::fabs(b[q]) + 16.0f * std::fabs(g[q]));
      return m;
    };
    float bound = reach(H["resnet.0.norm.weight"], H["resnet.0.norm.bias"]);
    for (int k = 1; k <= NBLOCK; ++k) {
      float sc = 1.f;
      if (bound > 0.f && std::isfinite(bound)) {
        int e;
        std::frexp(bound, &e);                       // bound = f 2^e, f in [0.5, 1)
        sc = std::ldexp(1.0f, std::max(-40, std::min(40, 15 - e)));      // bound * sc in [16384, 32768)
      }
      W.blk[k - 1].x_scale = sc;
      const std::string p = "resnet." + std::to_string(k);
      bound += 2.0f * reach(H[p + ".layer1.norm.weight"], H[p + ".layer1.norm.bias"]);
    }
  }
  // residual blocks
  for (int k = 1; k <= NBLOCK; ++k) {
    const std::string p = "resnet." + std::to_string(k);
    BlockW& B = W.blk[k - 1];
    const auto& w = H[p + ".layer1.lin.weight"];   // [512][128][5][5]
    std::vector<float> pk((size_t)512 * 128 * 25);
    for (int split = 0; split < CONV_SPLIT; ++split)
      for (int chunk = 0; chunk < CW / CONV_CC; ++chunk)
        for (int tap = 0; tap < 25; ++tap)
          for (int cc = 0; cc < CONV_CC; ++cc)
            for (int m = 0; m < 128; ++m) {
              const int och = split * 128 + m, ich = chunk * CONV_CC + cc;
              pk[((((size_t)split * (CW / CONV_CC) + chunk) * 25 + tap) * CONV_CC + cc) * 128 + m] =
                  w[((size_t)och * 128 + ich) * 25 + tap];
            }
    if ((rc = upload(pool, bytes, &B.wpack, pk))) return rc;
    {
      const std::vector<uint16_t> q = pack_conv_weights_bf16(w.data());
      if ((rc = dev_alloc(pool, bytes, &B.wq, (int64_t)q.size()))) return rc;
      DMP_HIP(hipMemcpy(B.wq, q.data(), sizeof(uint16_t) * q.size(), hipMemcpyHostToDevice));
      const float scale = conv_weight_scale_f16(w.data(), w.size());
      const std::vector<uint16_t> h = pack_conv_weights_f16(w.data(), scale);
      if ((rc = dev_alloc(pool, bytes, &B.wh, (int64_t)h.size()))) return rc;
      DMP_HIP(hipMemcpy(B.wh, h.data(), sizeof(uint16_t) * h.size(), hipMemcpyHostToDevice));
      B.wh_inv_scale = 1.0f / scale;
    }
    if ((rc = upload(pool, bytes, &B.bias, H[p + ".layer1.lin.bias"]))) return rc;
    if ((rc = upload(pool, bytes, &B.gamma, H[p + ".layer1.norm.weight"]))) return rc;
    const auto& beta = H[p + ".layer1.norm.bias"];
    if ((rc = upload(pool, bytes, &B.beta, beta))) return rc;
    // cSE gate: avgpool(InstanceNorm(x)) == beta, so the gate is a constant of the weights
    const auto& w1 = H[p + ".scSE.cSE.fc.0.weight"];   // [8][128]
    const auto& w2 = H[p + ".scSE.cSE.fc.2.weight"];   // [128][8]
    std::vector<float> hid(8), gate(128);
    for (int r = 0; r < 8; ++r) {
      float a = 0.f;
      for (int q = 0; q < 128; ++q) a += w1[r * 128 + q] * beta[q];
      hid[r] = a > 0.f ? a : 0.f;
    }
    for (int q = 0; q < 128; ++q) {
      float a = 0.f;
      for (int r = 0; r < 8; ++r) a += w2[q * 8 + r] * hid[r];
      gate[q] = 1.0f / (1.0f + std::exp(-a));
    }
    if ((rc = upload(pool, bytes, &B.cse, gate))) return rc;
    {
      std::vector<float> fc(w1);
      fc.insert(fc.end(), w2.begin(), w2.end());
      if ((rc = upload(pool, bytes, &B.cse_fc, fc))) return rc;
    }
    if ((rc = upload(pool, bytes, &B.sse_w, H[p + ".scSE.sSE.conv.weight"]))) return rc;
    B.sse_b = H[p + ".scSE.sSE.conv.bias"][0];
  }
  if ((rc = upload(pool, bytes, &W.head_w, H["resnet.17.weight"]))) return rc;
  W.head_b[0] = H["resnet.17.bias"][0];
  W.head_b[1] = H["resnet.17.bias"][1];
  return DMP_OK;
}

static int check_ready(dmp_ctx* c, int L, int N) {
  if (!c) { set_error("null context"); return DMP_ERR_ARG; }
  if (L > c->max_L || N > c->max_N) {
    set_error("alignment %d x %d exceeds the context capacity %d x %d", N, L, c->max_N, c->max_L);
    return DMP_ERR_CAPACITY;
  }
  return DMP_OK;
}
static int need_weights(dmp_ctx* c) {
  if (!c->W.ready) { set_error("weights not finalized"); return DMP_ERR_WEIGHTS; }
  return DMP_OK;
}

// A trunk pass = open (stem update, first split), 16 residual blocks, close (head + Gram matrix).
static int trunk_open(dmp_ctx* c, const float* z0, const float* dmap, int L, hipStream_t s) {
  int rc;
  c->trunk_cur = c->xa;
  c->trunk_oth = c->xb;
  c->xsplit_current = false;
  c->ab_current = false;
  // the stem's norm kernel also writes the f16 / bf16 pieces of its output; every block's norm kernel then emits
  // those of its own
  if ((rc = stem_update_padded(c, z0, dmap, L, c->trunk_cur, s, true))) return rc;
  c->xsplit_current = c->conv_mode != 1;
  return DMP_OK;
}

static int trunk_block(dmp_ctx* c, int k, int L, hipStream_t s) {
  int rc;
  {
    // The lane admits two convolutions at a time: a launch waits for the one before the previous one,
    // so the tail of one launch (1444 workgroups on 512 slots at L = 300, fewer at smaller L) fills with
    // the head of the next.  Measured (tools/lane_trace.py, structures/s at depth 1 / 2 / 3 / 4):
    // L = 300: 6.72 / 7.00 / 6.88 / 6.74; L = 200: 12.7 / 13.8-14.1 / 13.8 / 13.4; L = 128: 26.0 / 28.7 / 28.5
    // (round 2, row-reuse convolution: 6.66 / 6.93 / 6.79 at L = 300).  DMP_LANE_DEPTH overrides (tools/lane_trace.py).
    static const int depth = getenv("DMP_LANE_DEPTH") ? std::max(1, atoi(getenv("DMP_LANE_DEPTH"))) : 2;
    if (c->lane && c->lane->count >= depth)
      DMP_HIP(hipStreamWaitEvent(s, (hipEvent_t)c->lane->ev[(c->lane->count - depth) % dmp_lane::RING], 0));
  }
  if (c->prof_on && c->prof_n + 2 <= (int)c->prof_ev.size()) {
    DMP_HIP(hipEventRecord((hipEvent_t)c->prof_ev[c->prof_n], s));
  }
  if ((rc = conv5x5_maxout_padded(c, k, c->trunk_cur, L, c->u, c->stats, s, false))) return rc;
  if (c->lane) {
    dmp_lane* ln = c->lane;
    void* e = ln->ev[ln->next];
    ln->next = (ln->next + 1) % dmp_lane::RING;
    DMP_HIP(hipEventRecord((hipEvent_t)e, s));
    ln->last = e;
    ln->count++;
  }
  if (c->prof_on && c->prof_n + 2 <= (int)c->prof_ev.size()) {
    DMP_HIP(hipEventRecord((hipEvent_t)c->prof_ev[c->prof_n + 1], s));
    c->prof_n += 2;
  }
  if ((rc = conv5x5_reduce_stats(c, L, c->stats, s, k))) return rc;
  if ((rc = norm_scse_residual_padded(c, k, c->u, c->stats, c->trunk_cur, L, c->trunk_oth, s, k == NBLOCK))) return rc;
  std::swap(c->trunk_cur, c->trunk_oth);
  return DMP_OK;
}

static int trunk_close(dmp_ctx* c, int L, float* d_conf, float* d_M, hipStream_t s) {
  c->xsplit_current = false;
  return head_gram_padded(c, c->trunk_cur, L, d_conf, d_M, s);
}

static int trunk_pass(dmp_ctx* c, const float* z0, const float* dmap, int L, float* d_conf,
                      float* d_M, hipStream_t s) {
  struct Reset { dmp_ctx* c; ~Reset() { c->xsplit_current = false; } } reset{c};
  int rc;
  if ((rc = trunk_open(c, z0, dmap, L, s))) return rc;
  for (int k = 1; k <= NBLOCK; ++k)
    if ((rc = trunk_block(c, k, L, s))) return rc;
  return trunk_close(c, L, d_conf, d_M, s);
}

static int coords_from_mds(dmp_ctx* c, const float* mat1d, const float* mds, int L, float* d_ca,
                           hipStream_t s) {
  int rc;
  if ((rc = build_embed(mat1d, mds, L, c->emb, s))) return rc;
  if ((rc = gru_bidir(c, 1, c->emb, L, c->seq_a, s))) return rc;
  return coord_fc(c, c->seq_a, L, d_ca, s);
}

// End of a prediction: if a device-side fault was recorded while it ran, its outputs become NaN (an
// invalid structure can never be mistaken for a result, and a batch can tell WHICH target failed) and
// the fault bits are latched into the word dmp_sync_faults reports.
__global__ void fault_latch_kernel(int* __restrict__ words, float* __restrict__ coords,
                                   float* __restrict__ conf, int L, int* __restrict__ report) {
  const int f = words[0];
  if (!f) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float nan = __builtin_nanf("");
  if (i < 15 * L) coords[i] = nan;
  if (i < L) conf[i] = nan;
  if (i == 0) {
    atomicOr(&words[1], f);
    if (report) *report = f;          // the pipeline's per-ticket fault word (pinned host memory)
  }
}

}  // namespace dmp

using namespace dmp;

extern "C" {

int dmp_abi_version(void) { return DMP_ABI_VERSION; }
const char* dmp_last_error(void) { return g_err; }

int dmp_ctx_create(int device, int max_L, int max_N, dmp_ctx** out) {
  DMP_ARG(out != nullptr, "out is NULL");
  DMP_ARG(max_L >= 8 && max_L <= DMP_MAX_L, "max_L must be in [8, %d], got %d", DMP_MAX_L, max_L);
  DMP_ARG(max_N >= 1, "max_N must be >= 1, got %d", max_N);
  if (max_N > DMP_MAX_SEQS) max_N = DMP_MAX_SEQS;
  DMP_HIP(hipSetDevice(device));
  dmp_ctx* c = new dmp_ctx();
  c->device = device;
  c->max_L = max_L;
  {
    static std::atomic<int> next_xcd{0};             // spread the minimiser clusters of the contexts over the XCDs
    c->refine_xcd = next_xcd.fetch_add(1) & 7;
    // the contexts of a process spread their sequence-GRU clusters over the XCD pairs (four engines: 0-1, 2-3, 4-5, 6-7)
    // instead of all sitting on XCDs 0 and 1 (+0.5 % in the scheduler, round 3)
    c->seq_xcd0 = (2 * c->refine_xcd) & 7;
  }
  c->max_N = max_N;
  c->max_passes = 128;
  const int64_t L = max_L, N = max_N, D = NS * L, LL = L * L;
  // vertical-GRU state: the members and riders of a group side by side (up to 8 alignments of max_L columns,
  // 32-column tiles)
  c->vg_cap_cols = 8 * round_up(max_L, 64);
  const int64_t Lb = c->vg_cap_cols, P = act_pitch(max_L), T = act_tiles(max_L);
  int rc = 0;
#define A_(field, count) if (!rc) rc = dev_alloc(c->allocs, c->bytes, &c->field, (count))
  A_(msa_words, N * cdiv64(L, 4));
  A_(nbr_count, N);
  A_(w, N);
  A_(wsum, 2);
  A_(colmean, D);
  A_(xc, N * D);
  A_(cov, D * D);
  A_(gj_p, 2 * GJ_NB * GJ_NB);                                    // two sets: the look-ahead prepares step k+1
  A_(gj_c, 2 * (int64_t)round_up((int)D, GJ_NB) * GJ_NB);         // beside step k's trailing update (dca.hip)
  A_(gj_rt, 2 * (int64_t)round_up((int)D, GJ_NB) * GJ_NB);
  A_(contacts, LL);
  A_(x3, LL);
  A_(apc_sums, 2 * L + 1);
  for (int l = 0; l < 2; ++l)
    for (int p = 0; p < 2; ++p) {
      A_(hT[l][p], (int64_t)WIDTH * Lb);
      A_(hH[l][p], (int64_t)2 * WIDTH * Lb);
    }
  A_(vgru_run, 256);                 // VRun (legacy) / VGroupRec (vgru.hip)
  A_(vgru_sync, 2048);               // VPSync (vgru.hip)
  A_(vgru_wq, (int64_t)vgru_x3_stream_bytes());
  A_(vout, L * WIDTH);
  A_(seq_g, L * 1536);
  A_(seq_a, L * WIDTH);
  A_(seq_b, L * WIDTH);
  A_(emb, L * (WIDTH + 8));
  A_(mat1d, L * WIDTH);
  A_(seq_hx, 2 * 2 * HID2 + 4);
  A_(seq_abort, 2);      // [0] fault word of the prediction in flight, [1] faults latched by finished ones
  A_(refine_gx, 2 * 3 * L + 2);
  A_(tri_gx, 8 * std::min<int64_t>(L, 640) + 2);
  A_(z0, (int64_t)STEM_OUT * LL);
  A_(planes, (int64_t)(NS * NS + 1) * LL);
  A_(dmap, LL);
  A_(u, (int64_t)CW * LL);
  A_(xa, (int64_t)CW * P * P);
  A_(xb, (int64_t)CW * P * P);
  A_(xdense, (int64_t)CW * LL);
  A_(xsplit, (int64_t)3 * CW * P * P);
  A_(part, std::max<int64_t>(2 * T * T, 8) * CW * 2);
  A_(stats, CW * 2);
  A_(ab, CW * 2);
  A_(head0, LL);
  A_(head1, LL);
  A_(conf, L);
  A_(gram, LL);
  A_(eig_a, LL);
  A_(eig_ws, 3 * L + 16 + 8 * L + 40 * L + LL + 2 * L + 8);
  A_(mds, L * 8);
  A_(ca, L * 3);
  A_(best_ca, L * 3);
  A_(best_ca_snapshot, L * 3);
  A_(best_conf, L);
  A_(best_mean, 1);
  A_(conf_means, c->max_passes);
  A_(ca_pass, (int64_t)c->max_passes * L * 3);
#undef A_
  if (rc) { dmp_ctx_destroy(c); return rc; }
  if (hipMemset(c->seq_abort, 0, 2 * sizeof(int)) != hipSuccess) { dmp_ctx_destroy(c); return DMP_ERR_HIP; }
  if ((rc = trunk_kernel_attrs(c))) { dmp_ctx_destroy(c); return rc; }
  if ((rc = mds_kernel_attrs(c))) { dmp_ctx_destroy(c); return rc; }
  if ((rc = gj_kernel_attrs(c))) { dmp_ctx_destroy(c); return rc; }
  if ((rc = vgru_kernel_attrs(c))) { dmp_ctx_destroy(c); return rc; }
  for (int i = 0; i < 2; ++i) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { dmp_ctx_destroy(c); return DMP_ERR_HIP; }
    c->unit_ev[i] = (void*)e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { dmp_ctx_destroy(c); return DMP_ERR_HIP; }
    c->side_ev[i] = (void*)e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { dmp_ctx_destroy(c); return DMP_ERR_HIP; }
    c->gj_ev[i] = (void*)e;
  }
  {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { dmp_ctx_destroy(c); return DMP_ERR_HIP; }
    c->vg_done_ev = (void*)e;
  }
  *out = c;
  return DMP_OK;
}


int dmp_ctx_set_option(dmp_ctx* ctx, const char* name, int value) {
  DMP_ARG(ctx && name, "null argument");
  const std::string k(name);
  if (k == "conv_f32_exact") { ctx->conv_mode = value ? 1 : 0; return DMP_OK; }
  if (k == "tridiag_single") { ctx->tridiag_single = value ? 1 : 0; return DMP_OK; }
  if (k == "tridiag_cluster") { ctx->tridiag_cluster = value ? 1 : 0; return DMP_OK; }
  if (k == "cluster_local") { ctx->cluster_local = value ? 1 : 0; return DMP_OK; }
  if (k == "refine_single") { ctx->refine_single = value ? 1 : 0; return DMP_OK; }
  if (k == "gj_pairs") { ctx->gj_pairs = value != 0; return DMP_OK; }
  if (k == "gj_diag_blocked") { ctx->gj_diag_blocked = value != 0; return DMP_OK; }
  if (k == "gj_lookahead") { DMP_ARG(value >= 0 && value <= 2, "gj_lookahead must be 0, 1 or 2"); ctx->gj_lookahead = value; return DMP_OK; }
  if (k == "gj_diag_groups") { DMP_ARG(value == 2 || value == 4 || value == 8, "gj_diag_groups must be 2, 4 or 8"); ctx->gj_diag_groups = value; return DMP_OK; }
  if (k == "act_scaling") { ctx->act_scaling = value ? 1 : 0; return DMP_OK; }
  if (k == "conv_tile_bands") { DMP_ARG(value >= 0 && value <= 2, "conv_tile_bands must be 0 (by length), 1 (16 x 16) or 2 (8 x 16)"); ctx->conv_tile_bands = value; return DMP_OK; }
  if (k == "vgru_persistent") { ctx->vgru_persist = value ? 1 : 0; return DMP_OK; }
  if (k == "vgru_debug_drop_wg") { ctx->vgru_debug_drop_wg = value ? 1 : 0; return DMP_OK; }     // tests only
  if (k == "vgru_f32") { DMP_ARG(value >= -1 && value <= 2, "vgru_f32 must be -1 (follow conv_mode), 0, 1 or 2"); ctx->vgru_f32 = value; return DMP_OK; }
  if (k == "precision") {
    // 0: float32-grade split-f16 products in the convolutions and the vertical GRU (22-bit operands; the fast mode);
    // 1: the reference's arithmetic instruction for instruction - float32 MFMAs in both, library gate functions;
    // 2: full-width operands at the 16-bit matrix cores' rate - every float32 operand of the convolutions split EXACTLY
    //    into three bf16 pieces (3 x 8 = 24 significand bits), the six piece products above 2^-24 accumulated in float32
    //    (conv_bf16.h), and the vertical GRU the same way (vgru_x3.hip) with the library gate functions of setting 1
    DMP_ARG(value >= 0 && value <= 2, "precision must be 0 (split f16), 1 (float32 MFMA) or 2 (exact bf16 x 3 split)");
    ctx->conv_mode = value;
    ctx->vgru_f32 = value == 2 ? 2 : -1;
    return DMP_OK;
  }
  if (k == "conv_mode") {
    DMP_ARG(value >= 0 && value <= 2, "conv_mode must be 0 (f16x3), 1 (exact f32) or 2 (bf16x6)");
    ctx->conv_mode = value;
    return DMP_OK;
  }
  set_error("unknown option %s", name);
  return DMP_ERR_ARG;
}

int dmp_ctx_get_option(const dmp_ctx* ctx, const char* name, int* h_value) {
  DMP_ARG(ctx && name && h_value, "null argument");
  const std::string k(name);
  if (k == "conv_mode") { *h_value = ctx->conv_mode; return DMP_OK; }
  if (k == "conv_f32_exact") { *h_value = ctx->conv_mode == 1; return DMP_OK; }
  if (k == "act_scaling") { *h_value = ctx->act_scaling; return DMP_OK; }
  if (k == "conv_tile_bands") { *h_value = ctx->conv_tile_bands; return DMP_OK; }
  if (k == "device_mib") { *h_value = (int)((ctx->bytes + (1 << 20) - 1) >> 20); return DMP_OK; }    // read only
  if (k == "vgru_persistent") { *h_value = ctx->vgru_persist && ctx->vgru_persist_ok; return DMP_OK; }
  if (k == "vgru_debug_drop_wg") { *h_value = ctx->vgru_debug_drop_wg; return DMP_OK; }
  if (k == "vgru_f32") { *h_value = vgru_runs_f32(ctx); return DMP_OK; }          // what the next prediction will run
  if (k == "precision") {                                                          // 0 / 1 / 2, or -1 for a mixed setting
    const int v = vgru_runs_f32(ctx);
    *h_value = ctx->conv_mode == v ? v : -1;
    return DMP_OK;
  }
  // read-only: 1 once every launch of the group chain this context leads has been issued (the scheduler hands the
  // riders' results over from then on)
  if (k == "chain_issued") { *h_value = ctx->vg_leader == ctx && __atomic_load_n(&ctx->vg_done_issued, __ATOMIC_ACQUIRE) ? 1 : 0; return DMP_OK; }
  if (k.rfind("act_scale_log2_block", 0) == 0) {      // read-only: log2 of the piece scale of block 1..16's input
    const int b = atoi(k.c_str() + 20);
    DMP_ARG(b >= 1 && b <= NBLOCK && ctx->W.ready, "act_scale_log2_block<k>: k in 1..16, weights finalized");
    int e;
    std::frexp(ctx->W.blk[b - 1].x_scale, &e);
    *h_value = e - 1;
    return DMP_OK;
  }
  if (k == "tridiag_single") { *h_value = ctx->tridiag_single; return DMP_OK; }
  if (k == "tridiag_cluster") { *h_value = ctx->tridiag_cluster; return DMP_OK; }
  if (k == "cluster_local") { *h_value = ctx->cluster_local; return DMP_OK; }
  if (k == "refine_single") { *h_value = ctx->refine_single; return DMP_OK; }
  if (k == "gj_diag_groups") { *h_value = ctx->gj_diag_groups; return DMP_OK; }
  if (k == "gj_lookahead") { *h_value = ctx->gj_lookahead; return DMP_OK; }
  if (k == "gj_pairs") { *h_value = ctx->gj_pairs; return DMP_OK; }
  if (k == "gj_diag_blocked") { *h_value = ctx->gj_diag_blocked; return DMP_OK; }
  set_error("unknown option %s", name);
  return DMP_ERR_ARG;
}

int dmp_sync_faults(dmp_ctx* ctx, void* stream, int* h_bits) {
  DMP_ARG(ctx && h_bits, "null argument");
  DMP_HIP(hipStreamSynchronize((hipStream_t)stream));
  int words[2] = {0, 0};
  DMP_HIP(hipMemcpy(words, ctx->seq_abort, sizeof(words), hipMemcpyDeviceToHost));
  *h_bits = words[0] | words[1];
  // reported once: a fault of one prediction must not poison the checks of the following ones
  if (*h_bits) DMP_HIP(hipMemset(ctx->seq_abort, 0, sizeof(words)));
  return DMP_OK;
}


void dmp_ctx_destroy(dmp_ctx* c) {
  if (!c) return;
  coresident_forget(c);
  if (c->bwd_ws) (void)hipFree(c->bwd_ws);
  for (void* p : c->allocs) (void)hipFree(p);
  release_weights(c);
  for (void* e : c->prof_ev) (void)hipEventDestroy((hipEvent_t)e);
  for (void* e : c->unit_ev)
    if (e) (void)hipEventDestroy((hipEvent_t)e);
  for (void* e : c->side_ev)
    if (e) (void)hipEventDestroy((hipEvent_t)e);
  for (void* e : c->gj_ev)
    if (e) (void)hipEventDestroy((hipEvent_t)e);
  if (c->vg_done_ev) (void)hipEventDestroy((hipEvent_t)c->vg_done_ev);
  if (c->side_stream) (void)hipStreamDestroy((hipStream_t)c->side_stream);
  for (auto& kv : c->vgru_graphs) (void)hipGraphExecDestroy((hipGraphExec_t)kv.second);
  for (auto& kv : c->tri_graphs) (void)hipGraphExecDestroy((hipGraphExec_t)kv.second);
  delete c;
}


int dmp_weights_set(dmp_ctx* c, const char* key, const float* h_data, const int64_t* shape, int ndim) {
  DMP_ARG(c && key && h_data && shape, "null argument");
  static const std::vector<KeySpec> spec = weight_spec();
  for (const auto& k : spec) {
    if (k.key != key) continue;
    int64_t n = 1;
    bool ok = (int)k.shape.size() == ndim;
    for (int i = 0; ok && i < ndim; ++i) { ok = (shape[i] == k.shape[i]); n *= shape[i]; }
    if (!ok) { c->W.host.clear(); set_error("size mismatch for %s", key); return DMP_ERR_WEIGHTS; }
    c->W.host[key].assign(h_data, h_data + n);
    c->W.ready = false;
    return DMP_OK;
  }
  c->W.host.clear();     // a rejected state_dict leaves nothing staged for the next attempt
  set_error("unexpected key %s in state_dict", key);
  return DMP_ERR_WEIGHTS;
}

int dmp_weights_finalize(dmp_ctx* c) {
  DMP_ARG(c != nullptr, "null context");
  static const std::vector<KeySpec> spec = weight_spec();
  for (const auto& k : spec)
    if (!c->W.host.count(k.key)) {
      c->W.host.clear();
      set_error("missing key %s in state_dict", k.key.c_str());
      return DMP_ERR_WEIGHTS;
    }
  {
    uint64_t h = 1469598103934665603ull;               // FNV-1a over keys and bytes, in spec order
    for (const auto& k : spec) {
      const auto& v = c->W.host[k.key];
      const unsigned char* b = reinterpret_cast<const unsigned char*>(v.data());
      for (size_t i = 0; i < v.size() * sizeof(float); ++i) { h ^= b[i]; h *= 1099511628211ull; }
    }
    c->W.hash = h;
  }
  DMP_HIP(hipSetDevice(c->device));
  release_weights(c);
  c->bwd_w_block = 0;                   // the unpacked weights of the training-side slice belong to the old set
  int rc = pack_weights(c);
  if (rc) return rc;
  c->W.host.clear();
  // hand the buffers to a shared holder so that other contexts of this GPU can use them (dmp_weights_share)
  c->W.shared = std::make_shared<std::vector<void*>>(std::move(c->W.allocs));
  c->W.allocs.clear();
  c->W.ready = true;
  return DMP_OK;
}

int dmp_weights_share(dmp_ctx* dst, const dmp_ctx* src) {
  DMP_ARG(dst && src && dst != src, "two different contexts are needed");
  DMP_ARG(src->W.ready && src->W.shared, "the source context holds no packed weights (dmp_weights_finalize first)");
  DMP_ARG(dst->device == src->device, "the contexts live on different GPUs");
  DMP_HIP(hipSetDevice(dst->device));
  release_weights(dst);
  dst->W.host.clear();
  dst->W.shapes.clear();
  const std::shared_ptr<std::vector<void*>> hold = src->W.shared;
  dst->bwd_w_block = 0;
  dst->W = src->W;                    // pointers, scales, hash
  dst->W.host.clear();
  dst->W.shapes.clear();
  dst->W.allocs.clear();
  dst->W.shared = hold;
  return DMP_OK;
}

int dmp_msa_encode(const uint8_t* h_text, int64_t nbytes, uint8_t* h_codes) {
  DMP_ARG(h_text && h_codes && nbytes >= 0, "bad argument");
  uint8_t tab[256];
  for (int b = 0; b < 256; ++b) tab[b] = (uint8_t)(b - 65);
  const char* aa = "ARNDCQEGHILKMFPSTWYV";
  for (int i = 0; i < 20; ++i) tab[(uint8_t)aa[i]] = (uint8_t)i;
  for (const char* p = "BJOUXZ"; *p; ++p) tab[(uint8_t)*p] = 20;
  tab[(uint8_t)'-'] = 21;
  tab[(uint8_t)'.'] = 21;
  for (int64_t i = 0; i < nbytes; ++i) h_codes[i] = tab[h_text[i]];
  return DMP_OK;
}

#define STREAM ((hipStream_t)stream)
#define CHECK_CAP(L, N) do { int _rc = check_ready(ctx, (L), (N)); if (_rc) return _rc; } while (0)
#define CHECK_W() do { int _rc = need_weights(ctx); if (_rc) return _rc; } while (0)

int dmp_msa_weights(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, float* d_w, void* stream) {
  CHECK_CAP(L, N);
  DMP_ARG(d_msa && d_w && N >= 1 && L >= 1, "bad argument");
  return msa_weights(ctx, d_msa, N, L, d_w, STREAM);
}

int dmp_cov_build(dmp_ctx* ctx, const uint8_t* d_msa, const float* d_w, int N, int L, float* d_cov,
                  void* stream) {
  CHECK_CAP(L, N);
  DMP_ARG(d_msa && d_w && d_cov && N >= 1 && L >= 1, "bad argument");
  return cov_build(ctx, d_msa, d_w, N, L, d_cov, STREAM);
}

// the context's second stream (created on first use): the look-ahead of the inverse where one prediction has the
// device to itself, the covariance features beside the launch-per-row vertical GRU
static int side_stream_of(dmp_ctx* c, hipStream_t* out) {
  if (!c->side_stream) {
    // highest priority: its kernels are small and on the critical path (the look-ahead's one-workgroup sweep must not
    // queue behind the 1200 workgroups of the trailing update it runs beside)
    int least = 0, greatest = 0;
    DMP_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t st;
    DMP_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest));
    c->side_stream = (void*)st;
  }
  *out = (hipStream_t)c->side_stream;
  return DMP_OK;
}

int dmp_spd_inverse(dmp_ctx* ctx, float* d_A, int D, void* stream) {
  DMP_ARG(ctx && d_A && D >= 1, "bad argument");
  if (D > NS * ctx->max_L) { set_error("D=%d exceeds capacity %d", D, NS * ctx->max_L); return DMP_ERR_CAPACITY; }
  hipStream_t la = nullptr;
  int rc;
  if (ctx->gj_lookahead && (rc = side_stream_of(ctx, &la))) return rc;
  return spd_inverse(ctx, d_A, D, STREAM, la);
}

int dmp_dca_contacts(dmp_ctx* ctx, const float* d_inv, int L, float* d_contacts, void* stream) {
  CHECK_CAP(L, 1);
  DMP_ARG(d_inv && d_contacts, "null argument");
  return dca_contacts(ctx, d_inv, L, d_contacts, STREAM);
}

int dmp_dca_features(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, float* d_out, void* stream) {
  CHECK_CAP(L, N);
  DMP_ARG(d_msa && d_out && N >= 1 && L >= 1, "bad argument");
  if (N == 1) {                                   // predict.py:139: a single sequence has zero features
    DMP_HIP(hipMemsetAsync(d_out, 0, sizeof(float) * (size_t)L * L * NUM_DCA, STREAM));
    return DMP_OK;
  }
  int rc;
  if ((rc = msa_weights(ctx, d_msa, N, L, ctx->w, STREAM))) return rc;
  if ((rc = cov_build(ctx, d_msa, ctx->w, N, L, ctx->cov, STREAM, true))) return rc;
  hipStream_t la = nullptr;
  if (ctx->gj_lookahead && (rc = side_stream_of(ctx, &la))) return rc;
  if ((rc = spd_inverse(ctx, ctx->cov, NS * L, STREAM, la))) return rc;
  if ((rc = dca_contacts(ctx, ctx->cov, L, ctx->contacts, STREAM))) return rc;
  return dca_features(ctx->cov, ctx->contacts, L, d_out, STREAM);
}

int dmp_gru_vertical(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, float* d_out, void* stream) {
  CHECK_CAP(L, N);
  CHECK_W();
  DMP_ARG(d_msa && d_out && N >= 1, "bad argument");
  return gru_vertical(ctx, d_msa, N, L, d_out, STREAM);
}

int dmp_gru_vertical_group(dmp_ctx* const* ctxs, int n, const uint8_t* const* d_msas, const int* Ns,
                           const int* Ls, float* const* d_outs, void* stream) {
  DMP_ARG(ctxs && d_msas && Ns && Ls && d_outs && n >= 1 && n <= 8, "bad argument");
  int maxN = 0;
  for (int i = 0; i < n; ++i) {
    dmp_ctx* ctx = ctxs[i];
    DMP_ARG(ctx && d_msas[i] && d_outs[i] && Ns[i] >= 1, "bad argument for member %d", i);
    CHECK_CAP(Ls[i], Ns[i]);
    CHECK_W();
    DMP_ARG(ctx->device == ctxs[0]->device, "the members of a group must live on one GPU");
    maxN = std::max(maxN, Ns[i]);
  }
  int rc;
  if ((rc = vgru_group_setup(ctxs[0], ctxs, d_msas, Ns, Ls, n, STREAM))) return rc;
  if ((rc = vgru_group_steps(ctxs[0], 0, maxN + 1, STREAM))) return rc;
  for (int i = 0; i < n; ++i)
    if ((rc = vgru_group_output(ctxs[0], i, Ns[i], Ls[i], d_outs[i], STREAM))) return rc;
  return DMP_OK;
}

int dmp_gru_bidir(dmp_ctx* ctx, int which, const float* d_in, int T, float* d_out, void* stream) {
  CHECK_CAP(T, 1);
  CHECK_W();
  DMP_ARG((which == 0 || which == 1) && d_in && d_out && T >= 1, "bad argument");
  return gru_bidir(ctx, which, d_in, T, d_out, STREAM);
}

int dmp_stem_static(dmp_ctx* ctx, const float* d_mat1d, const float* d_inv, const float* d_contacts,
                    int L, float* d_z0, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_mat1d && d_z0, "null argument");
  DMP_ARG((d_inv == nullptr) == (d_contacts == nullptr), "d_inv and d_contacts must both be given or both NULL");
  return stem_static(ctx, d_mat1d, d_inv, d_contacts, L, d_z0, STREAM);
}

int dmp_stem_update(dmp_ctx* ctx, const float* d_z0, const float* d_dmap, int L, float* d_x,
                    void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_z0 && d_dmap && d_x, "null argument");
  int rc;
  if ((rc = act_clear(ctx->xa, L, STREAM))) return rc;
  if ((rc = stem_update_padded(ctx, d_z0, d_dmap, L, ctx->xa, STREAM))) return rc;
  return act_unpad(ctx->xa, L, d_x, STREAM);
}

int dmp_block_conv5x5_maxout(dmp_ctx* ctx, int block, const float* d_x, int L, float* d_u,
                             double* d_stats, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(block >= 1 && block <= NBLOCK && d_x && d_u && d_stats, "bad argument");
  int rc;
  if ((rc = act_pad(d_x, L, ctx->xa, STREAM))) return rc;
  return conv5x5_maxout_padded(ctx, block, ctx->xa, L, d_u, d_stats, STREAM, true);
}

int dmp_block_norm_scse_residual(dmp_ctx* ctx, int block, const float* d_u, const double* d_stats,
                                 const float* d_x, int L, float* d_out, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(block >= 1 && block <= NBLOCK && d_u && d_stats && d_x && d_out, "bad argument");
  int rc;
  if ((rc = act_pad(d_x, L, ctx->xa, STREAM))) return rc;
  if ((rc = act_clear(ctx->xb, L, STREAM))) return rc;
  if ((rc = norm_scse_residual_padded(ctx, block, d_u, d_stats, ctx->xa, L, ctx->xb, STREAM))) return rc;
  return act_unpad(ctx->xb, L, d_out, STREAM);
}

int dmp_block_conv5x5_maxout_winners(dmp_ctx* ctx, int block, const float* d_x, int L, float* d_u, uint8_t* d_idx,
                                     void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(block >= 1 && block <= NBLOCK && d_x && d_u && d_idx, "bad argument");
  return conv5x5_maxout_fwd_winners(ctx, block, d_x, L, d_u, d_idx, STREAM);
}

int dmp_block_conv5x5_maxout_bwd(dmp_ctx* ctx, int block, const float* d_x, const float* d_du, const uint8_t* d_idx, int L,
                                 float* d_dx, float* d_dw, float* d_db, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(block >= 1 && block <= NBLOCK && d_x && d_du && d_dx && d_dw && d_db, "bad argument");
  return conv5x5_maxout_bwd(ctx, block, d_x, d_du, d_idx, L, d_dx, d_dw, d_db, STREAM);
}

int dmp_block_norm_scse_residual_bwd(dmp_ctx* ctx, int block, const float* d_u, const float* d_dout, int L,
                                     float* d_du, float* d_dparams, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(block >= 1 && block <= NBLOCK && d_u && d_dout && d_du && d_dparams, "bad argument");
  return norm_scse_residual_bwd(ctx, block, d_u, d_dout, L, d_du, d_dparams, STREAM);
}

int dmp_stem_maxout_winners(dmp_ctx* ctx, const float* d_z0, const float* d_dmap, int L, float* d_u, uint8_t* d_idx, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_z0 && d_dmap && d_u && d_idx, "null argument");
  return stem_maxout_fwd_winners(ctx, d_z0, d_dmap, L, d_u, d_idx, STREAM);
}

int dmp_stem_bwd(dmp_ctx* ctx, const float* d_u, const uint8_t* d_idx, const float* d_dy, const float* d_mat1d,
                 const float* d_dmap, int L, float* d_dw, float* d_dparams, float* d_dmat1d, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_u && d_idx && d_dy && d_mat1d && d_dmap && d_dw && d_dparams && d_dmat1d, "null argument");
  return stem_bwd(ctx, d_u, d_idx, d_dy, d_mat1d, d_dmap, L, d_dw, d_dparams, d_dmat1d, STREAM);
}

int dmp_head_conv_bwd(dmp_ctx* ctx, const float* d_x, const float* d_g, int L, float* d_dx, float* d_dparams, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_x && d_g && d_dx && d_dparams, "null argument");
  return head_conv_bwd(ctx, d_x, d_g, L, d_dx, d_dparams, STREAM);
}

int dmp_head_gram(dmp_ctx* ctx, const float* d_x, int L, float* d_conf, float* d_M, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_x && d_conf && d_M, "null argument");
  int rc;
  if ((rc = act_pad(d_x, L, ctx->xa, STREAM))) return rc;
  return head_gram_padded(ctx, ctx->xa, L, d_conf, d_M, STREAM);
}

int dmp_trunk_pass(dmp_ctx* ctx, const float* d_z0, const float* d_dmap, int L, float* d_conf,
                   float* d_M, void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_z0 && d_dmap && d_conf && d_M, "null argument");
  int rc;
  if ((rc = act_clear(ctx->xa, L, STREAM))) return rc;
  if ((rc = act_clear(ctx->xb, L, STREAM))) return rc;
  return trunk_pass(ctx, d_z0, d_dmap, L, d_conf, d_M, STREAM);
}

int dmp_eigh_top8(dmp_ctx* ctx, const float* d_M, int L, float* d_mds, void* stream) {
  CHECK_CAP(L, 1);
  DMP_ARG(d_M && d_mds && L >= 8, "bad argument (L must be >= 8)");
  return eigh_top8(ctx, d_M, L, d_mds, STREAM);
}

int dmp_coords_from_mds(dmp_ctx* ctx, const float* d_mat1d, const float* d_mds, int L, float* d_ca,
                        void* stream) {
  CHECK_CAP(L, 1);
  CHECK_W();
  DMP_ARG(d_mat1d && d_mds && d_ca, "null argument");
  return coords_from_mds(ctx, d_mat1d, d_mds, L, d_ca, STREAM);
}

int dmp_pair_distances(dmp_ctx* ctx, const float* d_ca, int L, int clamp, float* d_dmap,
                       void* stream) {
  DMP_ARG(ctx && d_ca && d_dmap && L >= 1, "bad argument");
  return pair_distances(d_ca, L, clamp, d_dmap, STREAM);
}

int dmp_refine_coords(dmp_ctx* ctx, float* d_ca, int L, int steps, void* stream) {
  DMP_ARG(ctx && d_ca && L >= 2 && L <= DMP_MAX_L && steps >= 0, "bad argument");
  DMP_ARG(L >= 2 && L <= ctx->max_L, "L = %d outside 2..max_L", L);
  return refine_coords(ctx, d_ca, L, steps, STREAM);
}

int dmp_ca_to_backbone(dmp_ctx* ctx, const float* d_ca, const float* d_conf_logit, int L,
                       float* d_coords, float* d_conf_out, void* stream) {
  DMP_ARG(ctx && d_ca && d_conf_logit && d_coords && d_conf_out && L >= 3, "bad argument (L must be >= 3)");
  return ca_to_backbone(d_ca, d_conf_logit, L, d_coords, d_conf_out, STREAM);
}

static int predict_begin(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, const float* d_template_ca,
                         int Lt, int nloops, int refine_steps, void* stream);

int dmp_predict(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, const float* d_template_ca, int Lt,
                int nloops, int refine_steps, float* d_coords, float* d_conf, void* stream) {
  // begin_units + every front-end unit (the features on the context's side stream) + every unit of every pass + end
  int rc = predict_begin(ctx, d_msa, N, L, d_template_ca, Lt, nloops, refine_steps, stream);
  if (rc) return rc;
  while (ctx->passes_done <= ctx->run_nloops)
    if ((rc = dmp_predict_issue_unit(ctx, stream))) return rc;
  return dmp_predict_end(ctx, d_coords, d_conf, stream);
}

// ---- unit-granular issue ---------------------------------------------------------------------
// A prediction = front-end units (features, sequence trunk, static stem; chunked so that a scheduler
// regains control every ~2 ms of GPU work) followed by 18 units per pass: 0 = recycled distances +
// stem update, 1..16 = residual block k (its convolution takes the lane), 17 = head, Gram matrix,
// MDS, coordinate GRU, best-of update.
static constexpr int FE_INV_BLOCKS = 6;    // Gauss-Jordan block steps per front-end unit
static constexpr int FE_VGRU_STEPS = VGRU_CHUNK;  // vertical-GRU time steps per front-end unit (one graph replay)

static int record_unit(dmp_ctx* c, hipStream_t s) {
  DMP_HIP(hipEventRecord((hipEvent_t)c->unit_ev[c->unit_seq & 1], s));
  c->unit_seq++;
  return DMP_OK;
}

// The covariance features (reweighting, covariance, Gauss-Jordan inverse, contacts: f32 matrix cores,
// serial diagonal blocks) and the vertical GRU (bound by L1 misses) are independent and use different
// units of the chip, so the features run on the context's side stream and their units alternate with
// the GRU's: one unit of each kind is in flight at a time.  They join before the static stem.

// The launch chain of the vertical-GRU group `c` leads, chunk j of c->fe_vgru (a real group's chain is ONE chunk),
// enqueued on s; behind the last chunk the members' results and the event every member's last front-end unit waits for.
static bool vg_done(const dmp_ctx* lead) { return __atomic_load_n(&lead->vg_done_issued, __ATOMIC_ACQUIRE); }

static int issue_group_chain(dmp_ctx* c, int j, hipStream_t s) {
  const int n = (int)c->vg_members.size(), nr = (int)c->vg_riders.size();
  int rc = DMP_OK;
  if (j == 0) {
    const uint8_t* msas[8];
    dmp_ctx* owners[8];
    int Ns[8], Ls[8];
    for (int i = 0; i < n; ++i) {
      owners[i] = c->vg_members[i];
      msas[i] = c->vg_members[i]->run_msa; Ns[i] = c->vg_members[i]->last_N; Ls[i] = c->vg_members[i]->last_L;
    }
    for (int i = 0; i < nr; ++i) {           // riders: state in the leader's buffers like the members', no context
      owners[n + i] = c;
      msas[n + i] = c->vg_riders[i].msa; Ns[n + i] = c->vg_riders[i].N; Ls[n + i] = c->vg_riders[i].L;
    }
    rc = vgru_group_setup(c, owners, msas, Ns, Ls, n + nr, s);
  }
  // a real group's chain is ONE unit (every member waits for its end, and nothing else wants this stream
  // meanwhile): chunked, the chain stood still for 2-3 ms between chunks whenever the scheduler thread was busy
  // issuing the members' inverse units (kernel trace: 40 ms of a 97 ms front-end phase)
  const bool whole = n + nr > 1 || c->fe_vgru == 1;
  const int chunks = whole ? 1 : cdiv(c->vg_maxN + 1, FE_VGRU_STEPS);
  if (!rc) rc = whole ? vgru_group_steps(c, 0, c->vg_maxN + 1, s)
                      : vgru_group_steps(c, j * FE_VGRU_STEPS, (j + 1) * FE_VGRU_STEPS, s);
  if (!rc && j == chunks - 1) {
    for (int i = 0; !rc && i < n; ++i) {
      dmp_ctx* m = c->vg_members[i];
      rc = vgru_group_output(c, i, m->last_N, m->last_L, m->vout, s, m);
    }
    for (int i = 0; !rc && i < nr; ++i)
      rc = vgru_group_output(c, n + i, c->vg_riders[i].N, c->vg_riders[i].L, c->vg_riders[i].out, s);
    if (!rc) {
      DMP_HIP(hipEventRecord((hipEvent_t)c->vg_done_ev, s));
      __atomic_store_n(&c->vg_done_issued, true, __ATOMIC_RELEASE);
    }
  }
  return rc;
}

static int issue_front_end_unit(dmp_ctx* c, hipStream_t s) {
  const int L = c->last_L, N = c->last_N, u = c->fe_next;
  const uint8_t* d_msa = c->run_msa;
  // Only the single-target entry (dmp_predict_begin) forks: a scheduler that drives several contexts has
  // other targets to fill the machine, and a second stream per context would push the process past the
  // hardware queues (with 4 engines the unused side streams alone cost 10 % of the throughput).
  // (not beside the persistent vertical GRU: that launch holds every CU, and the inverse's kernels squeezed in between
  // its row barriers took 27 ms instead of 9 - profiles/r04_single_target_timeline.txt; one after the other then)
  const bool fork = c->fe_side && c->fe_inv > 0 && !(c->vgru_persist && c->vgru_persist_ok);
  // where the front end does not fork, a single prediction's inverse uses the second stream for its look-ahead
  // (spd_inverse_steps): the next step's one-workgroup sweep and panels beside this step's trailing update
  const bool ahead = c->fe_side && !fork && c->gj_lookahead && c->fe_inv > 0;
  hipStream_t side = s, la = nullptr;
  int rc = DMP_OK;
  if (fork && (rc = side_stream_of(c, &side))) return rc;
  if (ahead && (rc = side_stream_of(c, &la))) return rc;
  hipStream_t used = s;
  if (u == 0) {
    // the fault word of the previous prediction was latched by its dmp_predict_end
    DMP_HIP(hipMemsetAsync(c->seq_abort, 0, sizeof(int), s));
    if (fork) {
      DMP_HIP(hipEventRecord((hipEvent_t)c->side_ev[0], s));
      DMP_HIP(hipStreamWaitEvent(side, (hipEvent_t)c->side_ev[0], 0));
    }
    used = side;
    rc = msa_weights(c, d_msa, N, L, c->w, side);
    if (!rc && N > 1) rc = cov_build(c, d_msa, c->w, N, L, c->cov, side, true);
  } else if (u <= c->fe_inv + c->fe_vgru) {
    // alternate GRU chunk / inverse chunk while both kinds remain
    const int k = u - 1, m = std::min(c->fe_inv, c->fe_vgru);
    bool inv;
    int j;
    if (c->vg_leader == c && c->vg_members.size() > 1) {
      // a group leader issues the whole chain first (every member waits for it) and its own inverse afterwards
      inv = k >= c->fe_vgru;
      j = inv ? k - c->fe_vgru : k;
    } else if (k < 2 * m) { inv = (k & 1) != 0; j = k >> 1; }
    else { inv = c->fe_inv > m; j = m + (k - 2 * m); }
    if (inv) {
      used = side;
      rc = spd_inverse_steps(c, c->cov, NS * L, j * c->fe_inv_blocks, (j + 1) * c->fe_inv_blocks, side, la);
      if (!rc && j == c->fe_inv - 1) {
        rc = dca_contacts(c, c->cov, L, c->contacts, side);
        if (!rc && fork) DMP_HIP(hipEventRecord((hipEvent_t)c->side_ev[1], side));
      }
    } else if (c->vg_leader == c) {
      // the chain of the whole group: this context's units serve every member
      rc = issue_group_chain(c, j, s);
    } else {
      rc = c->fe_vgru == 1 ? gru_vertical_steps(c, d_msa, N, L, 0, N + 1, c->vout, s)
                           : gru_vertical_steps(c, d_msa, N, L, j * FE_VGRU_STEPS, std::min((j + 1) * FE_VGRU_STEPS, N + 1),
                                                c->vout, s);
    }
  } else {
    const float* inv = N > 1 ? c->cov : nullptr;
    const float* contacts = N > 1 ? c->contacts : nullptr;
    if (fork) DMP_HIP(hipStreamWaitEvent(s, (hipEvent_t)c->side_ev[1], 0));
    if (c->vg_leader && c->vg_leader != c) {
      // this member's vertical GRU ran in its leader's chain, on the leader's stream
      dmp_ctx* lead = c->vg_leader;
      DMP_ARG(vg_done(lead), "the vertical-GRU chain of this context's group has not been issued to its end yet "
                             "(dmp_predict_next_unit answers DMP_UNIT_WAIT until it has)");
      DMP_HIP(hipStreamWaitEvent(s, (hipEvent_t)lead->vg_done_ev, 0));
      lead->vg_waiters--;
      c->vg_leader = nullptr;
    }
    if (c->ext_vout && c->ext_vout_ev) DMP_HIP(hipStreamWaitEvent(s, (hipEvent_t)c->ext_vout_ev, 0));
    rc = gru_bidir(c, 0, c->ext_vout ? c->ext_vout : c->vout, L, c->seq_b, s);
    if (!rc) rc = transpose_f32(c->seq_b, L, WIDTH, c->mat1d, s);
    if (!rc) rc = stem_static(c, c->mat1d, inv, contacts, L, c->z0, s);
    if (!rc) rc = c->run_template ? pair_distances(c->run_template, L, 0, c->dmap, s)
                                  : fill_f32(c->dmap, (int64_t)L * L, -1.0f, s);
    if (!rc) rc = act_clear(c->xa, L, s);
    if (!rc) rc = act_clear(c->xb, L, s);
  }
  if (rc) return rc;
  c->fe_next = u + 1;
  return record_unit(c, used);
}

int dmp_predict_begin_units(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, const float* d_template_ca,
                            int Lt, int nloops, int refine_steps) {
  CHECK_CAP(L, N);
  CHECK_W();
  DMP_ARG(d_msa != nullptr, "null argument");
  DMP_ARG(N >= 1 && L >= 8, "need N >= 1 and L >= 8 (got N=%d L=%d)", N, L);
  DMP_ARG(d_template_ca == nullptr || Lt == L,
          "template has %d CA atoms but the alignment has %d columns", Lt, L);
  dmp_ctx* c = ctx;
  DMP_ARG(c->vg_waiters == 0, "members of the vertical-GRU group this context led have not taken their results yet");
  DMP_ARG(c->vg_leader == nullptr || c->vg_leader == c, "this context still waits for the vertical GRU of its group");
  c->vg_leader = nullptr;
  c->vg_members.clear();
  c->vg_riders.clear();
  c->vg_done_issued = false;
  c->ext_vout = nullptr;
  c->ext_vout_ev = nullptr;
  c->last_L = L;
  c->last_N = N;
  c->passes_done = 0;
  c->unit_next = 0;
  c->end_refined = false;
  c->run_nloops = nloops < 0 ? 0 : nloops;
  c->run_refine = refine_steps < 0 ? 0 : refine_steps;
  c->run_msa = d_msa;
  c->run_template = d_template_ca;
  c->fe_next = 0;
  c->fe_side = false;
  c->fe_inv_blocks = FE_INV_BLOCKS;
  c->fe_inv = N > 1 ? cdiv(cdiv(NS * L, GJ_NB), FE_INV_BLOCKS) : 0;
  // the vertical GRU in units of 128 rows (launch-per-row form: one graph replay each) - or, as the persistent launch,
  // as ONE unit: every launch of that form loads the CUs' weight slices first
  c->fe_vgru = (c->vgru_persist && c->vgru_persist_ok) ? 1 : cdiv(N + 1, FE_VGRU_STEPS);
  c->fe_total = 1 + c->fe_inv + c->fe_vgru + 1;
  return DMP_OK;
}

static int predict_begin(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, const float* d_template_ca,
                         int Lt, int nloops, int refine_steps, void* stream) {
  int rc = dmp_predict_begin_units(ctx, d_msa, N, L, d_template_ca, Lt, nloops, refine_steps);
  if (!rc) {
    ctx->fe_side = true;
    if (ctx->fe_inv > 0 && ctx->gj_lookahead && ctx->vgru_persist && ctx->vgru_persist_ok) {
      // the whole inverse as one unit: its look-ahead does not reach across unit boundaries
      ctx->fe_inv_blocks = cdiv(NS * L, GJ_NB);
      ctx->fe_inv = 1;
      ctx->fe_total = 1 + ctx->fe_inv + ctx->fe_vgru + 1;
    }
  }
  while (!rc && ctx->fe_next < ctx->fe_total) rc = issue_front_end_unit(ctx, STREAM);
  return rc;
}

int dmp_predict_group_vgru(dmp_ctx* const* ctxs, int n) {
  DMP_ARG(ctxs && n >= 1 && n <= 8, "a group has 1..8 members");
  dmp_ctx* lead = ctxs[0];
  int cols = 0, maxN = 0;
  for (int i = 0; i < n; ++i) {
    dmp_ctx* c = ctxs[i];
    DMP_ARG(c != nullptr, "null context");
    for (int k = 0; k < i; ++k) DMP_ARG(ctxs[k] != c, "a context appears twice in the group");
    DMP_ARG(c->fe_total > 0 && c->fe_next == 0 && c->vg_leader == nullptr,
            "member %d: group its vertical GRU right after dmp_predict_begin_units, before any unit is issued", i);
    DMP_ARG(c->device == lead->device, "the members of a group must live on one GPU");
    DMP_ARG(c->W.ready && c->W.hash == lead->W.hash, "member %d does not hold the leader's weights", i);
    // the chain runs in the LEADER's arithmetic: a member set to another one would silently get the leader's
    DMP_ARG(vgru_runs_f32(c) == vgru_runs_f32(lead), "member %d: its vertical-GRU arithmetic (options precision / vgru_f32) "
            "differs from the leader's", i);
    cols += round_up(c->last_L, 32);
    maxN = std::max(maxN, c->last_N);
  }
  if (cols > lead->vg_cap_cols) {
    set_error("the group's %d alignment columns exceed the leader's capacity %d", cols, lead->vg_cap_cols);
    return DMP_ERR_CAPACITY;
  }
  lead->vg_members.assign(ctxs, ctxs + n);
  lead->vg_done_issued = false;
  lead->vg_waiters = n - 1;
  for (int i = 0; i < n; ++i) {
    dmp_ctx* c = ctxs[i];
    c->vg_leader = lead;
    c->vg_index = i;
    c->fe_vgru = i == 0 ? ((n > 1 || (lead->vgru_persist && lead->vgru_persist_ok)) ? 1 : cdiv(maxN + 1, FE_VGRU_STEPS)) : 0;
    c->fe_total = 1 + c->fe_inv + c->fe_vgru + 1;
  }
  return DMP_OK;
}

int dmp_predict_group_riders(dmp_ctx* lead, int n, const uint8_t* const* d_msas, const int* Ns, const int* Ls,
                             float* const* d_outs) {
  DMP_ARG(lead && d_msas && Ns && Ls && d_outs && n >= 1, "bad argument");
  DMP_ARG(lead->vg_leader == lead && lead->fe_next == 0 && lead->vg_riders.empty(),
          "add riders right after dmp_predict_group_vgru, once, before the leader issues a unit");
  DMP_ARG((int)lead->vg_members.size() + n <= 8, "a chain serves at most 8 alignments (%d members + %d riders)",
          (int)lead->vg_members.size(), n);
  int cols = 0, maxN = 0;
  for (dmp_ctx* m : lead->vg_members) { cols += round_up(m->last_L, 32); maxN = std::max(maxN, m->last_N); }
  for (int i = 0; i < n; ++i) {
    DMP_ARG(d_msas[i] && d_outs[i] && Ns[i] >= 1 && Ls[i] >= 8, "bad argument for rider %d", i);
    if (Ls[i] > lead->max_L || Ns[i] > lead->max_N) {
      set_error("rider %d: alignment %d x %d exceeds the context capacity %d x %d", i, Ns[i], Ls[i], lead->max_N, lead->max_L);
      return DMP_ERR_CAPACITY;
    }
    cols += round_up(Ls[i], 32);
    maxN = std::max(maxN, Ns[i]);
  }
  if (cols > lead->vg_cap_cols) {
    set_error("the chain's %d alignment columns exceed the leader's capacity %d", cols, lead->vg_cap_cols);
    return DMP_ERR_CAPACITY;
  }
  for (int i = 0; i < n; ++i) lead->vg_riders.push_back({d_msas[i], Ns[i], Ls[i], d_outs[i]});
  // with riders even a group of one runs its chain as one unit (issue_group_chain)
  lead->fe_vgru = 1;
  lead->fe_total = 1 + lead->fe_inv + lead->fe_vgru + 1;
  return DMP_OK;
}

int dmp_predict_set_vgru_result(dmp_ctx* ctx, const float* d_vout, void* event) {
  DMP_ARG(ctx && d_vout, "null argument");
  DMP_ARG(ctx->fe_total > 0 && ctx->fe_next == 0 && ctx->vg_leader == nullptr,
          "hand the vertical-GRU result over right after dmp_predict_begin_units, before any unit is issued");
  ctx->ext_vout = d_vout;
  ctx->ext_vout_ev = event;
  ctx->fe_vgru = 0;
  ctx->fe_total = 1 + ctx->fe_inv + 1;
  return DMP_OK;
}

int dmp_predict_next_unit(const dmp_ctx* ctx) {
  if (!ctx) return DMP_UNIT_NONE;
  if (ctx->fe_next == ctx->fe_total - 1 && ctx->vg_leader && ctx->vg_leader != ctx && !vg_done(ctx->vg_leader))
    return DMP_UNIT_WAIT;      // the group's chain has not been issued to its end yet
  if (ctx->fe_next < ctx->fe_total) return DMP_UNIT_LIGHT;
  if (ctx->passes_done > ctx->run_nloops) return DMP_UNIT_NONE;
  return (ctx->unit_next >= 1 && ctx->unit_next <= NBLOCK) ? DMP_UNIT_CONV : DMP_UNIT_LIGHT;
}

int dmp_predict_issue_unit(dmp_ctx* ctx, void* stream) {
  DMP_ARG(ctx != nullptr, "null context");
  dmp_ctx* c = ctx;
  hipStream_t s = STREAM;
  if (c->fe_next < c->fe_total) return issue_front_end_unit(c, s);
  DMP_ARG(c->passes_done <= c->run_nloops, "all passes of this prediction were already issued");
  const int L = c->last_L, pass = c->passes_done, u = c->unit_next;
  int rc = DMP_OK;
  if (u == 0) {
    if (pass > 0) rc = pair_distances(c->ca, L, 1, c->dmap, s);
    if (!rc) rc = trunk_open(c, c->z0, c->dmap, L, s);
  } else if (u <= NBLOCK) {
    rc = trunk_block(c, u, L, s);
  } else {
    rc = trunk_close(c, L, c->conf, c->gram, s);
    if (!rc) rc = eigh_top8(c, c->gram, L, c->mds, s);
    if (!rc) rc = coords_from_mds(c, c->mat1d, c->mds, L, c->ca, s);
    if (!rc && pass == 0 && c->run_refine > 0) rc = refine_coords(c, c->ca, L, c->run_refine, s);
    if (!rc) rc = select_best(c, c->conf, c->ca, L, pass, c->max_passes, s);
  }
  if (rc) { c->xsplit_current = false; return rc; }
  if (u == NBLOCK + 1) { c->unit_next = 0; c->passes_done = pass + 1; }
  else c->unit_next = u + 1;
  return record_unit(c, s);
}

// Units issued through dmp_predict_begin / dmp_predict_issue_unit that have not completed yet:
// 0, 1, or 2 (= two or more).
int dmp_ctx_pending(dmp_ctx* ctx) {
  DMP_ARG(ctx != nullptr, "null context");
  if (ctx->unit_seq == 0) return 0;
  hipError_t e = hipEventQuery((hipEvent_t)ctx->unit_ev[(ctx->unit_seq - 1) & 1]);
  if (e == hipSuccess) return 0;
  if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
  if (ctx->unit_seq < 2) return 1;
  e = hipEventQuery((hipEvent_t)ctx->unit_ev[ctx->unit_seq & 1]);
  if (e == hipSuccess) return 1;
  if (e != hipErrorNotReady) return hip_fail(e, "hipEventQuery", __FILE__, __LINE__);
  return 2;
}


// final refinement of the best trace: first half of dmp_predict_end
static int predict_end_refine(dmp_ctx* ctx, void* stream) {
  DMP_ARG(ctx != nullptr, "null context");
  dmp_ctx* c = ctx;
  DMP_ARG(c->fe_next >= c->fe_total && c->passes_done == c->run_nloops + 1,
          "dmp_predict_end before all passes were issued");
  if (c->end_refined) return DMP_OK;
  hipStream_t s = STREAM;
  const int L = c->last_L;
  int rc;
  DMP_HIP(hipMemcpyAsync(c->best_ca_snapshot, c->best_ca, sizeof(float) * 3 * L,
                         hipMemcpyDeviceToDevice, s));
  if (c->run_refine > 0 && (rc = refine_coords(c, c->best_ca, L, c->run_refine, s))) return rc;
  c->end_refined = true;
  return record_unit(c, s);
}

int dmp_predict_end(dmp_ctx* ctx, float* d_coords, float* d_conf, void* stream) {
  DMP_ARG(ctx && d_coords && d_conf, "null argument");
  dmp_ctx* c = ctx;
  int rc = predict_end_refine(ctx, stream);
  if (rc) return rc;
  hipStream_t s = STREAM;
  const int L = c->last_L;
  // No lane turn: the corruption once seen here (C, O, CB of lanes 48..63 beside another context's
  // split-product convolution) was a packed-f32 instruction form that coords.hip no longer contains
  // (tools/isa_lint.py).
  rc = ca_to_backbone(c->best_ca, c->best_conf, L, d_coords, d_conf, s);
  if (rc) return rc;
  hipLaunchKernelGGL(fault_latch_kernel, dim3(cdiv(15 * L, 256)), dim3(256), 0, s, c->seq_abort, d_coords,
                     d_conf, L, c->end_fault_out);
  DMP_LAUNCH_CHECK();
  return DMP_OK;
}

// ---- heavy lane: serialises the conv launches of the contexts that share it -----------------
static std::shared_ptr<dmp_lane> lane_create() {
  std::shared_ptr<dmp_lane> l(new dmp_lane(), [](dmp_lane* p) {
    for (void* e : p->ev) (void)hipEventDestroy((hipEvent_t)e);
    delete p;
  });
  for (int i = 0; i < dmp_lane::RING; ++i) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    l->ev.push_back((void*)e);
  }
  return l;
}

int dmp_ctx_share_lane(dmp_ctx* ctx, dmp_ctx* other) {
  DMP_ARG(ctx != nullptr, "null context");
  if (!other) { ctx->lane_hold.reset(); ctx->lane = nullptr; return DMP_OK; }
  DMP_ARG(other != ctx && other->device == ctx->device, "a lane is shared by different contexts of one GPU");
  if (!other->lane_hold) {
    other->lane_hold = lane_create();
    if (!other->lane_hold) { set_error("hipEventCreate failed for the lane"); return DMP_ERR_HIP; }
    other->lane = other->lane_hold.get();
  }
  ctx->lane_hold = other->lane_hold;
  ctx->lane = ctx->lane_hold.get();
  return DMP_OK;
}

int64_t dmp_debug_fetch(dmp_ctx* ctx, const char* name, float* d_dst, int64_t capacity, void* stream) {
  DMP_ARG(ctx && name && d_dst, "null argument");
  const int64_t L = ctx->last_L, N = ctx->last_N;
  const int64_t P = std::min(ctx->passes_done, ctx->max_passes);
  const float* src = nullptr;
  int64_t n = 0;
  const std::string k(name);
  if (k == "w") { src = ctx->w; n = N; }
  else if (k == "contacts") { src = ctx->contacts; n = L * L; }
  else if (k == "mat1d") { src = ctx->mat1d; n = WIDTH * L; }
  else if (k == "conf_means") { src = ctx->conf_means; n = P; }
  else if (k == "ca_pass") { src = ctx->ca_pass; n = P * L * 3; }
  else if (k == "best_ca") { src = ctx->best_ca_snapshot; n = L * 3; }
  else if (k == "best_ca_refined") { src = ctx->best_ca; n = L * 3; }
  else if (k == "inv_cov") { src = ctx->cov; n = (int64_t)NS * L * NS * L; }
  else if (k == "mds") { src = ctx->mds; n = L * 8; }
  else if (k == "gram") { src = ctx->gram; n = L * L; }
  else if (k == "vgru_h0" || k == "vgru_h1") {
    // the vertical GRU's float32 state after the chain this context led last: [128][Lb][4], Lb = 32 x (column tiles)
    const int layer = k == "vgru_h1";
    src = ctx->hT[layer][ctx->vg_maxN & 1];
    n = (int64_t)WIDTH * ctx->vg_ntiles * 32;
  }
  else { set_error("unknown debug tensor %s", name); return DMP_ERR_ARG; }
  if (n > capacity) { set_error("capacity %lld too small for %s (%lld)", (long long)capacity, name, (long long)n); return DMP_ERR_ARG; }
  if (n > 0) DMP_HIP(hipMemcpyAsync(d_dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, STREAM));
  return n;
}

int dmp_profile_enable(dmp_ctx* ctx, int on, int max_launches) {
  DMP_ARG(ctx != nullptr, "null context");
  ctx->prof_on = on != 0;
  ctx->prof_n = 0;
  while ((int)ctx->prof_ev.size() < 2 * max_launches) {
    hipEvent_t e;
    DMP_HIP(hipEventCreate(&e));
    ctx->prof_ev.push_back((void*)e);
  }
  return DMP_OK;
}


// developer diagnostic (tools/lane_trace.py): start and end of every recorded conv launch in ms after the
// first recorded launch of `ref`; does not reset the counter
int dmp_profile_conv_intervals(dmp_ctx* ctx, dmp_ctx* ref, float* h_start_ms, float* h_end_ms, int capacity,
                               int* h_launches) {
  DMP_ARG(ctx && ref && h_start_ms && h_end_ms && h_launches, "null argument");
  DMP_ARG(ref->prof_n >= 2, "the reference context has no recorded launch");
  const int n = ctx->prof_n / 2 < capacity ? ctx->prof_n / 2 : capacity;
  hipEvent_t e0 = (hipEvent_t)ref->prof_ev[0];
  for (int i = 0; i < n; ++i) {
    DMP_HIP(hipEventElapsedTime(&h_start_ms[i], e0, (hipEvent_t)ctx->prof_ev[2 * i]));
    DMP_HIP(hipEventElapsedTime(&h_end_ms[i], e0, (hipEvent_t)ctx->prof_ev[2 * i + 1]));
  }
  *h_launches = n;
  return DMP_OK;
}


}  // extern "C"
