// Internal declarations shared by the HIP translation units of libdmpfold_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <map>
#include <algorithm>
#include <memory>
#include "../../include/dmpfold_hip.h"

namespace dmp {

constexpr int WIDTH = 512;     // GRU width
constexpr int HID2 = 256;      // bidirectional GRU hidden per direction
constexpr float VGRU_STATE_SCALE = 1024.f;   // vertical-GRU state is split into f16 pieces of 1024*h
constexpr int VGRU_CHUNK = 128;        // vertical-GRU time steps per hipGraph replay ...
constexpr int VGRU_CHUNK_SMALL = 16;   // ... and for the last <= 64 steps
constexpr int CW = 128;        // pair trunk width
constexpr int NBLOCK = 16;
constexpr int STEM_OUT = 384;  // 128 * pool 3
constexpr int STEM_IN = 955;
constexpr int NS = 21;         // one-hot classes of the DCA features
constexpr int NUM_DCA = NS * NS + 1;   // channels of fast_dca's return value (network.py:10)
constexpr int GJ_NB = 128;     // Gauss-Jordan block size
constexpr int CONV_TILE = 16;  // conv output tile edge
constexpr int CONV_CC = 2;     // input channels per K stage
constexpr int CONV_SPLIT = 4;  // 512 conv channels / 128 per workgroup

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define DMP_HIP(expr)                                                         \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) return ::dmp::hip_fail(_e, #expr, __FILE__, __LINE__); \
  } while (0)
#define DMP_LAUNCH_CHECK() DMP_HIP(hipGetLastError())
#define DMP_ARG(cond, ...)                 \
  do {                                     \
    if (!(cond)) {                         \
      ::dmp::set_error(__VA_ARGS__);       \
      return DMP_ERR_ARG;                  \
    }                                      \
  } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// padded activation geometry: interior [2, 2+L) in both axes, zero elsewhere
inline int act_tiles(int L) { return cdiv(L, CONV_TILE); }
inline int act_pitch(int L) { return act_tiles(L) * CONV_TILE + 4; }

struct GruDirW {          // one direction of one layer of a sequence GRU
  float* whh = nullptr;   // [3H][H]    row-major as in the state_dict
  float* bhh = nullptr;   // [3H]
  int nin = 0;
  // forward direction only: both directions' input weights side by side, [nin][2 x 3H] and [2 x 3H], so
  // that one GEMM launch (twice the workgroups) projects the inputs of both
  float* wihT_both = nullptr;
  float* bih_both = nullptr;
};

struct BlockW {
  float* wpack = nullptr;  // [split 4][chunk 64][tap 25][cc 2][m 128]   (exact-f32 kernel)
  uint16_t* wq = nullptr;  // bf16 pieces, layout in conv_bf16.h          (bf16x6 kernel)
  uint16_t* wh = nullptr;  // f16 pieces of scale*w, layout in conv_f16.h (f16x3 kernel)
  float wh_inv_scale = 1.f;
  // power-of-two scale of the f16 pieces of this block's INPUT activations (conv_mode 0), chosen at
  // dmp_weights_finalize from the InstanceNorm gamma / beta of the blocks before it (api.hip: act_scales)
  float x_scale = 1.f;
  float* bias = nullptr;   // [512]
  float* gamma = nullptr;  // [128]
  float* beta = nullptr;   // [128]
  float* cse = nullptr;    // [128] channel gate sigma(W2 relu(W1 beta))
  float* cse_fc = nullptr; // [8][128] fc.0.weight, [128][8] fc.2.weight as uploaded (the backward slice, train.hip)
  float* sse_w = nullptr;  // [128]
  float sse_b = 0.f;
};

struct Weights {
  bool ready = false;
  uint64_t hash = 0;                                       // FNV-1a of the tensors handed over (equal weights <=> equal hash)
  std::map<std::string, std::vector<float>> host;          // raw tensors by key
  std::map<std::string, std::vector<int64_t>> shapes;
  // vertical GRU: f16 pieces of scale*W, [piece 2][gate 3][k/8][512][8] (see vgru.hip)
  uint16_t* v_wx[2] = {nullptr, nullptr};  // input weights (layer 0: K = 22 padded to 32)
  uint16_t* v_wh[2] = {nullptr, nullptr};  // hidden weights
  float v_inv_scale[2] = {1.f, 1.f};       // 1 / (weight scale * state scale) per layer
  float *v_b0 = nullptr, *v_b1 = nullptr;  // [4][512]: r(bi+bh), z(bi+bh), in(bi), hn(bh)
  // vertical GRU in float32 (vgru_f32.hip, option "precision" = 1): hh_l0, ih_l1, hh_l1 as [gate 3][k/4 128][512 j][4],
  // ih_l0 with the embedding folded in as [gate 3][code 22][512 j]
  float* v_f32[3] = {nullptr, nullptr, nullptr};
  float* v_wx0f = nullptr;
  GruDirW hgru[2][2];                      // [layer][dir]
  GruDirW cgru[3][2];
  float* fc = nullptr;                     // [3][512]
  float* stemT = nullptr;                  // [955][384]
  float* stem_b = nullptr;                 // [384]
  float* stem_wd = nullptr;                // [384] weights of channel 954
  float *stem_gamma = nullptr, *stem_beta = nullptr;
  BlockW blk[NBLOCK];
  float* head_w = nullptr;                 // [2][128]
  float head_b[2] = {0.f, 0.f};
  std::vector<void*> allocs;                               // device buffers of the packed weights (owner only)
  // dmp_weights_share: this context uses the packed weights of `owner` (pointers above copied from it); the device
  // buffers live as long as any context refers to them
  std::shared_ptr<std::vector<void*>> shared;
};

}  // namespace dmp

// Shared by the contexts of one process that run concurrently on different streams: the
// machine-filling conv5x5 launches take turns (cross-stream events), everything else overlaps.
struct dmp_lane {
  static constexpr int RING = 64;
  std::vector<void*> ev;   // hipEvent_t ring
  int next = 0;
  void* last = nullptr;    // event recorded after the most recent conv launch
  long long count = 0;     // conv launches recorded so far (event of launch i: ev[i % RING])
};

struct dmp_ctx {
  int device = 0;
  dmp_lane* lane = nullptr;                // = lane_hold.get()
  std::shared_ptr<dmp_lane> lane_hold;     // dmp_ctx_share_lane: the lane lives as long as a context refers to it
  int run_nloops = 0, run_refine = 0;
  int max_L = 0, max_N = 0;
  int64_t bytes = 0;
  dmp::Weights W;
  std::vector<void*> allocs;

  // feature builder
  uint8_t* msa = nullptr;  // private copy is not needed; kept for packed words
  uint32_t* msa_words = nullptr;
  int* nbr_count = nullptr;
  float* w = nullptr;
  double* wsum = nullptr;   // [2]: sum w, (unused)
  float* colmean = nullptr; // [21L]
  float* xc = nullptr;      // [N][21L]
  float* cov = nullptr;     // [D][D]
  float* gj_p = nullptr;                    // inverse of the diagonal block (two sets of each: step parity)
  float *gj_c = nullptr, *gj_rt = nullptr;  // column / row panels as k quads [32][Dp][4] (dca.hip)
  void* gj_ev[2] = {nullptr, nullptr};      // look-ahead fork / join of the inverse (spd_inverse_steps)
  int gj_pairs = 1;                         // option: block steps of the inverse in pairs, one pass over the tiles per pair (dca.hip)
  int gj_lookahead = 1;                     // option: the next step's sweep and panels beside the trailing update (0 off, 1 from 64 tile rows on, 2 always)
  int fe_inv_blocks = 6;                    // block steps per front-end unit of this prediction
  float* contacts = nullptr;  // [L][L]
  float* x3 = nullptr;
  double* apc_sums = nullptr;  // [2L+1]
  // sequence trunk
  float* hT[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [layer][parity][128][Lb][4] float32 state (Lb: the group's columns)
  uint16_t* hH[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // same state as f16 pieces [2][64][Lb][8]
  uint8_t* vgru_run = nullptr;             // device VRun / VRun2 record read by the graph's step kernels
  uint8_t* vgru_sync = nullptr;            // VPSync of the persistent chain (vgru.hip): XCD arrival counters, row flags
  uint8_t* vgru_wq = nullptr;              // vgru_x3.hip: the weight pieces its waves stream (written at every launch), 24 MiB
  int vgru_persist = 1;                    // option "vgru_persistent": the chain as ONE weight-stationary launch (0: one launch per row)
  bool vgru_persist_ok = false;            // the device has the 256 CUs the persistent form is laid out for
  int vgru_debug_drop_wg = 0;              // TEST option "vgru_debug_drop_wg": launch the persistent chain one workgroup short (its row
                                           // barrier must time out ONCE, raise DMP_FAULT_VGRU_HANDOFF and leave the row loop)
  int vgru_f32 = 2;                        // option "vgru_f32": 2 = full-width operands as three bf16 pieces + library gates
                                           // (vgru_x3.hip), 1 = float32 MFMAs + library gates (vgru_f32.hip), 0 = split-f16
                                           // products, -1 = follow the convolution: float32 with conv_mode 1, else split-f16.
                                           // A context starts in option "precision" 2: conv_mode 2 + vgru_f32 2 (round 6)
  int vg_ntiles = 0, vg_maxN = 0;          // group this context leads (vgru.hip): column tiles, deepest member alignment
  int vg_tile0[8] = {0};                   // ... first column tile of every member
  int vg_cap_cols = 0;                     // columns the state buffers hold (a group's members side by side)
  // dmp_predict_group_vgru: the vertical GRUs of several predictions in flight as one launch chain
  dmp_ctx* vg_leader = nullptr;            // the context whose units run the chain (itself for the leader; null: no group)
  std::vector<dmp_ctx*> vg_members;        // leader: the members, itself first
  struct VgRider { const uint8_t* msa; int N, L; float* out; };
  std::vector<VgRider> vg_riders;          // leader: alignments of LATER predictions whose vertical GRU rides in this chain
  int vg_index = 0;                        // this context's index among its leader's members
  bool vg_done_issued = false;             // leader: the chain's last unit (and the members' outputs) has been enqueued
  int vg_waiters = 0;                      // leader: members that have not yet enqueued their wait for vg_done_ev
  void* vg_done_ev = nullptr;              // hipEvent_t recorded behind the chain's last unit
  const float* ext_vout = nullptr;         // dmp_predict_set_vgru_result: the vertical GRU of this prediction was run ahead
  void* ext_vout_ev = nullptr;             // ... hipEvent_t recorded behind it (not owned)
  std::map<int64_t, void*> vgru_graphs;    // (grid, chain length) -> hipGraphExec_t
  std::map<int, void*> tri_graphs;         // matrix order -> hipGraphExec_t of the tridiagonalisation chain
  int cluster_local = 1;                   // option: 0 = the cluster kernels always publish with agent-scope stores (no XCD-local fast path)
  int tridiag_single = 0;                  // option: 1 = single-workgroup tridiagonalisation
  int tridiag_cluster = 1;                 // option: 1 = all Householder steps in one cluster launch (orders <= 640); 0 = one launch per step
  unsigned long long* tri_gx = nullptr;    // [2][4][min(max_L, 640)] hand-off granules of the tridiagonalisation cluster + [2] placement header
  int refine_single = 0;                   // option: 1 = single-workgroup minimiser
  int gj_diag_groups = 4;                  // option: row groups of the diagonal sweep (2 = 256 threads as in rounds 1-3, 4 = 512 threads)
  int gj_diag_blocked = 0;                 // option: 1 = the diagonal block swept in 8 sub-blocks of 16 pivots (round 5); 0 (default) = the chain of 128 pivots
  float* vout = nullptr;    // [L][512]
  float* seq_g = nullptr;   // [L][1536] input projections, both directions
  float* seq_a = nullptr;   // [L][512]
  float* seq_b = nullptr;   // [L][512]
  float* emb = nullptr;     // [L][520]
  float* mat1d = nullptr;   // [512][L]
  unsigned long long* seq_hx = nullptr;  // [2][2][256] hand-off granules of the sequence GRU + [2 dir][2] placement header
  int* seq_abort = nullptr;              // [2] DMP_FAULT_* bits: [0] of the prediction in flight, [1] latched by finished ones
  unsigned long long* refine_gx = nullptr;  // [2][3 max_L] hand-off granules of the minimiser cluster + [2] placement header
  int refine_xcd = 0;                    // XCD the minimiser cluster of this context runs on
  int seq_xcd0 = 0;                      // XCDs seq_xcd0, seq_xcd0 + 1: the two directions of this context's sequence GRUs
  // pair trunk
  float* z0 = nullptr;      // [384][L][L]
  float* planes = nullptr;  // [442][L][L]: the coupling channels 21a+b and the contact channel as planes (stem_static's B operand)
  float* dmap = nullptr;    // [L][L]
  float* u = nullptr;       // [128][L][L]
  float* xa = nullptr;      // padded activations
  float* xb = nullptr;
  float* xdense = nullptr;  // [128][L][L] scratch for the stage-level API
  uint16_t* xsplit = nullptr;  // [3][16][P][P][8] bf16 pieces of the current activations
  int conv_mode = 2;           // 2: exact 3 x bf16 pieces, 6 products (default: full-width operands), 1: f32 MFMA, 0: f16x3 split products (fast mode)
  int act_scaling = 1;         // conv_mode 0: f16 pieces of x_scale * activation per block (0 = unscaled pieces)
  bool xsplit_current = false; // the producer of the activations already wrote their bf16 pieces
  bool ab_current = false;     // the statistics reduction already wrote this block's InstanceNorm coefficients
  double* part = nullptr;   // [half tiles][128][2] partial sums of the convolution launched last (or the stem's blocks)
  int part_count = 0;       // ... how many of them it wrote (conv5x5_reduce_stats)
  int conv_tile_bands = 0;  // option "conv_tile_bands": 0 = chosen by L (8-row tiles up to L = 240), 1 = 16 x 16, 2 = 8 x 16 pixels
  double* stats = nullptr;  // [128][2]
  float* ab = nullptr;      // [128][2] alpha, beta of the norm
  float* bwd_ws = nullptr;     // training-side slice (train.hip): the one workspace of both backward halves, allocated at
  int64_t bwd_ws_floats = 0;   // the first call for max_L
  int bwd_w_block = 0;         // the block whose flipped weight pack the workspace holds (0: none)
  float* head0 = nullptr;   // [L][L] head channel 0 (distances)
  float* head1 = nullptr;   // [L][L] head channel 1 (confidence logits)
  bool head_current = false;   // the last block's norm kernel already wrote head0 / head1 of the current activations
  float* conf = nullptr;    // [L]
  float* gram = nullptr;    // [L][L]
  // coordinates
  double* eig_a = nullptr;  // [L][L]
  double* eig_ws = nullptr; // d, e, tau, lambda, z ...
  float* mds = nullptr;     // [L][8]
  float* ca = nullptr;      // [L][3]
  float* best_ca = nullptr;
  float* best_conf = nullptr;
  float* best_mean = nullptr;  // [1]
  float* conf_means = nullptr; // [P]
  float* ca_pass = nullptr;    // [P][L][3]
  float* best_ca_snapshot = nullptr;
  int passes_done = 0;
  int* end_fault_out = nullptr;  // pipeline.hip: device-visible host word that the NEXT dmp_predict_end's latch kernel writes this
                                 // prediction's fault bits to (per-ticket status without a synchronising copy); not owned
  bool end_refined = false;    // dmp_predict_end_refine already issued for the prediction in flight
  int unit_next = 0;           // next unit of the current pass (0 open, 1..16 blocks, 17 close + MDS + coordinates)
  float *trunk_cur = nullptr, *trunk_oth = nullptr;   // ping-pong activations of the pass in flight
  void* unit_ev[2] = {nullptr, nullptr};   // hipEvent_t ring: recorded after each unit issued
  void* side_stream = nullptr;             // hipStream_t: dmp_predict_begin runs the covariance inverse here, beside the vertical GRU
  void* side_ev[2] = {nullptr, nullptr};   // fork / join events of the side stream
  bool fe_side = false;                    // this prediction's front end forks onto the side stream
  long unit_seq = 0;           // units issued since the context was created
  int fe_next = 0, fe_total = 0;           // front-end units (features, sequence trunk, static stem)
  int fe_inv = 0, fe_vgru = 0;             // ... of which inverse chunks / vertical-GRU chunks
  const uint8_t* run_msa = nullptr;        // arguments of the prediction in flight
  const float* run_template = nullptr;
  int last_L = 0, last_N = 0;
  int max_passes = 0;
  // optional HIP-event timing of the conv kernel inside dmp_predict
  bool prof_on = false;
  int prof_n = 0;
  std::vector<void*> prof_ev;
};

namespace dmp {
// ---- launchers implemented in the .hip files (all enqueue on `s`) -------------------------
// gemm.hip : C[M x N] = alpha * op(A) * op(B) + beta * C, element strides given explicitly
struct GemmArgs {
  const float* A; int64_t sam, sak;   // A(m,k) = A[m*sam + k*sak]
  const float* B; int64_t sbk, sbn;   // B(k,n) = B[k*sbk + n*sbn]
  float* C; int64_t ldc;              // C(m,n) = C[m*ldc + n]
  int M, N, K;
  float alpha, beta;
  const float* bias_n;                // optional, added per column n
  bool lower_tiles;                   // only the 128 x 128 tiles on / below the diagonal (a symmetric product whose
                                      // consumer reads those alone: the covariance before its in-place inverse)
};
int gemm_f32(const GemmArgs& g, hipStream_t s);

// msa.hip
int msa_weights(dmp_ctx* c, const uint8_t* d_msa, int N, int L, float* d_w, hipStream_t s);
// dca.hip
// lower_only: leave the 128 x 128 tiles above the diagonal uncomputed (spd_inverse never reads them)
int cov_build(dmp_ctx* c, const uint8_t* d_msa, const float* d_w, int N, int L, float* d_cov,
              hipStream_t s, bool lower_only = false);
int trunk_kernel_attrs(dmp_ctx* c);   // once per device, at context creation
int spd_inverse(dmp_ctx* c, float* d_A, int D, hipStream_t s, hipStream_t la = nullptr);
// block steps [blk_lo, blk_hi) of the in-place inverse (GJ_NB columns each); all of them = spd_inverse
// (`la`: a second stream for the look-ahead, or null)
int spd_inverse_steps(dmp_ctx* c, float* d_A, int D, int blk_lo, int blk_hi, hipStream_t s, hipStream_t la = nullptr);
int dca_contacts(dmp_ctx* c, const float* d_inv, int L, float* d_contacts, hipStream_t s);
int dca_features(const float* d_inv, const float* d_contacts, int L, float* d_out, hipStream_t s);
// gru.hip
int gru_vertical(dmp_ctx* c, const uint8_t* d_msa, int N, int L, float* d_out, hipStream_t s);
// launches t in [t_lo, t_hi) of the N + 1 per-row launches (t_lo = 0 clears the state, t_hi = N + 1
// writes d_out)
int gru_vertical_steps(dmp_ctx* c, const uint8_t* d_msa, int N, int L, int t_lo, int t_hi, float* d_out,
                       hipStream_t s);
// Vertical GRUs of several contexts in ONE launch per alignment row (vgru.hip): setup writes the column-tile
// records of all members into the leader's buffer and clears their states, steps runs rows [t_lo, t_hi), output
// hands a member its L x 512 result.
int vgru_kernel_attrs(dmp_ctx* c);
size_t vgru_x3_stream_bytes();            // size of dmp_ctx::vgru_wq (vgru_x3.hip)
// the arithmetic of the vertical GRU of this context's next prediction: 0 split-f16 (vgru.hip), 1 float32 MFMAs
// (vgru_f32.hip), 2 three exact bf16 pieces per operand (vgru_x3.hip) - option "vgru_f32", or following the convolution's
// mode when that option is -1
inline int vgru_runs_f32(const dmp_ctx* c) { return c->vgru_f32 >= 0 ? c->vgru_f32 : (c->conv_mode == 1 ? 1 : 0); }
// Kernels whose workgroups wait for EACH OTHER inside the launch - the cluster kernels (sequence GRU, minimiser,
// cluster tridiagonalisation: 32 workgroups that hand values over) and the persistent vertical GRU (256 workgroups
// with row barriers, one per CU) - must not start while another such kernel holds part of what they need: launched
// at the same moment from two streams, a cluster could get 20 of its 32 CUs and the persistent launch the other 236,
// and both would wait for the rest until their time-outs (seen in round 4 as DMP_FAULT_VGRU_HANDOFF + a hand-off
// time-out of a co-running engine's sequence GRU).  Within a process the launches are therefore ordered per device:
// a persistent launch waits for every cluster kernel launched before it (whatever context and stream) and for the
// persistent launch before it, a cluster kernel waits for the last persistent launch.  Clusters of different contexts
// may still run beside each other (they are placed on different XCDs).  Usage: CoResident g(c, s, persistent); launch;
// g.done().  (Another PROCESS on the same GPU is not covered: time-out, fault bit, the Python layer's fallback.)
class CoResident {
 public:
  CoResident(dmp_ctx* c, hipStream_t s, bool persistent);
  ~CoResident();
  int status() const { return rc_; }
  int done();
 private:
  dmp_ctx* c_; hipStream_t s_; bool persistent_, locked_; int rc_;
};
void coresident_forget(dmp_ctx* c);   // dmp_ctx_destroy
int vgru_group_setup(dmp_ctx* lead, dmp_ctx* const* members, const uint8_t* const* msas, const int* Ns,
                     const int* Ls, int n, hipStream_t s);
int vgru_group_steps(dmp_ctx* lead, int t_lo, int t_hi, hipStream_t s);
int vgru_group_output(dmp_ctx* lead, int member_index, int N, int L, float* d_out, hipStream_t s, dmp_ctx* member = nullptr);
int gru_bidir(dmp_ctx* c, int which, const float* d_in, int T, float* d_out, hipStream_t s);
// trunk.hip
int stem_static(dmp_ctx* c, const float* d_mat1d, const float* d_inv, const float* d_contacts,
                int L, float* d_z0, hipStream_t s);
int stem_update_padded(dmp_ctx* c, const float* d_z0, const float* d_dmap, int L, float* d_xpad,
                       hipStream_t s, bool split = false);
int conv5x5_maxout_padded(dmp_ctx* c, int block, const float* d_xpad, int L, float* d_u,
                          double* d_stats, hipStream_t s, bool reduce);
int conv5x5_reduce_stats(dmp_ctx* c, int L, double* d_stats, hipStream_t s, int block = 0);
int conv5x5_maxout_winners(dmp_ctx* c, int block, const float* d_xpad, int L, float* d_u, uint8_t* d_idx, hipStream_t s);
int norm_scse_residual_padded(dmp_ctx* c, int block, const float* d_u, const double* d_stats,
                              const float* d_xpad_in, int L, float* d_xpad_out, hipStream_t s, bool head = false);
int conv5x5_maxout_fwd_winners(dmp_ctx* c, int block, const float* d_x, int L, float* d_u, uint8_t* d_idx, hipStream_t s);
int conv5x5_maxout_bwd(dmp_ctx* c, int block, const float* d_x, const float* d_du, const uint8_t* d_idx, int L, float* d_dx,
                       float* d_dw, float* d_db, hipStream_t s);
int norm_scse_residual_bwd(dmp_ctx* c, int block, const float* d_u, const float* d_dout, int L, float* d_du,
                           float* d_dparams, hipStream_t s);
int head_conv_bwd(dmp_ctx* c, const float* d_x, const float* d_g, int L, float* d_dx, float* d_dparams, hipStream_t s);
int stem_maxout_fwd_winners(dmp_ctx* c, const float* d_z0, const float* d_dmap, int L, float* d_u, uint8_t* d_idx, hipStream_t s);
int stem_bwd(dmp_ctx* c, const float* d_u, const uint8_t* d_idx, const float* d_dy, const float* d_mat1d, const float* d_dmap,
             int L, float* d_dw, float* d_dparams, float* d_dmat1d, hipStream_t s);
int head_gram_padded(dmp_ctx* c, const float* d_xpad, int L, float* d_conf, float* d_M,
                     hipStream_t s);
int act_pad(const float* d_dense, int L, float* d_xpad, hipStream_t s);
int act_unpad(const float* d_xpad, int L, float* d_dense, hipStream_t s);
int act_clear(float* d_xpad, int L, hipStream_t s);
int act_split(dmp_ctx* c, const float* d_xpad, int L, int block, hipStream_t s);
// mds.hip
// ---- GRU gate functions: float32-accurate (<= 2.5 ulp) on the hardware exponential and reciprocal ---------------------
// v_exp_f32 / v_rcp_f32 are 1 ulp each; what a plain `exp2(x * log2 e)` loses is the rounding of the product (relative
// error |x| 2^-24 in the result), and `1 - 2 / (1 + e^2x)` loses the small tanh values to cancellation (absolute error
// 1.2e-7 whatever x).  Measured on the headline fixture (L=300, N=2000, 10 + 100 against the reference): with the plain
// forms in the sequence GRUs the final structure is 1.32e-3 A from the reference's, with the device library's
// expf / tanhf 8.6e-4 A (the reference's own thread-count spread: 8.2e-4) - and the library costs 0.55 us of the
// 1.3 us a step takes.  Here: the product's rounding error is recovered with one fma and applied as a first-order
// correction; tanh below 0.625 is an odd polynomial (degree 13, 0.8 ulp), above it the exponential form (<= 2.3 ulp).
__device__ __forceinline__ float gate_exp(float x) {
  const float L = 1.442695041f, Ll = 1.925963033e-08f;            // log2(e) = L + Ll
  const float t = x * L;
  const float e = fmaf(x, L, -t) + x * Ll;                        // t + e = x log2(e) to 2^-48
  const float r = __builtin_amdgcn_exp2f(t);
  return fmaf(r, e * 0.6931471806f, r);                           // 2^(t + e) = 2^t (1 + e ln 2)
}
#ifndef GATE_DIV
#define GATE_DIV 0      // 1: IEEE division instead of v_rcp_f32 (0.5 instead of 1 ulp, ten instructions instead of one)
#endif
__device__ __forceinline__ float gate_rcp(float x) { return GATE_DIV ? 1.0f / x : __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float gate_sigmoid(float x) { return gate_rcp(1.0f + gate_exp(-x)); }
__device__ __forceinline__ float gate_tanh(float x) {
  const float ax = fabsf(x), s = x * x;
  float p = 0.0022958183957f;
  p = fmaf(p, s, -0.0083469559308f);
  p = fmaf(p, s, 0.021769951322f);
  p = fmaf(p, s, -0.053959404269f);
  p = fmaf(p, s, 0.13333304266f);
  p = fmaf(p, s, -0.33333333178f);
  const float small = fmaf(x * s, p, x);
  const float big = copysignf(1.0f - 2.0f * gate_rcp(1.0f + gate_exp(2.0f * ax)), x);
  return ax < 0.625f ? small : big;
}

// ---- sum over the 64 lanes of a wave, float64, every lane gets the total ----------------------------------------
// The xor butterfly 32, 16, 8, 4, 2, 1 of `v += __shfl_xor(v, off, 64)` - the same operands in the same order, so
// the same bits - without going through the LDS crossbar: v_permlane32_swap / v_permlane16_swap (gfx950) pair lane i
// with i ^ 32 / i ^ 16, DPP row rotations do the four steps inside a row of 16 (after the xor-8 step lanes i and i ^ 8
// agree, so "lane i + 4 mod 16" holds what lane i ^ 4 holds, and so on down).  Twelve dependent ds_bpermute round
// trips (about 0.7 us) become twelve register moves: the Householder steps, the back-transformation and the
// eigenvector orthogonalisation are chains of such sums.
typedef unsigned dmp_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double wave_sum_f64(double v) {
  {
    const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    const dmp_u32x2 l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const dmp_u32x2 h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __longlong_as_double((long long)(((unsigned long long)h[0] << 32) | l[0])) +
        __longlong_as_double((long long)(((unsigned long long)h[1] << 32) | l[1]));
  }
  {
    const unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    const dmp_u32x2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const dmp_u32x2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __longlong_as_double((long long)(((unsigned long long)h[0] << 32) | l[0])) +
        __longlong_as_double((long long)(((unsigned long long)h[1] << 32) | l[1]));
  }
#define DMP_ROR_F64(n)                                                                                          \
  {                                                                                                             \
    const int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32);                     \
    const unsigned l = (unsigned)__builtin_amdgcn_update_dpp(0, lo, 0x120 + (n), 0xf, 0xf, false);              \
    const unsigned h = (unsigned)__builtin_amdgcn_update_dpp(0, hi, 0x120 + (n), 0xf, 0xf, false);              \
    v += __longlong_as_double((long long)(((unsigned long long)h << 32) | l));                                  \
  }
  DMP_ROR_F64(8) DMP_ROR_F64(4) DMP_ROR_F64(2) DMP_ROR_F64(1)
#undef DMP_ROR_F64
  return v;
}

// ---- hand-off between the workgroups of a cluster kernel (seq_gru_kernel, refine_cluster_kernel) -------------------
// Granules {epoch, value} are published with a store and gathered with sc1 loads (agent scope: past the L1, served by
// the L2).  An agent-scope STORE is written through to the memory side so that every XCD can see it: measured
// (tools/ubench_handoff.hip) 450 ns one way, and 2.78 us per step of the sequence GRU with 256 granules per step.  A
// PLAIN store stops in the L2 of the XCD its workgroup runs on, where the sc1 loads of workgroups on THAT XCD find it:
// 284 ns one way, 1.29 us per step, the same bits - and never seen from another XCD.  The block id -> XCD round robin
// puts a cluster on one XCD, but nothing guarantees it, so the cluster checks where it actually runs: every workgroup
// ORs the bit of its XCC id (hardware register) into a mask word and counts itself in (agent-scope atomics, zeroed
// by the host before the launch); when all `members` have arrived, a mask with one bit set means plain stores are safe.
// Returns true for "one XCD" (call from one thread per workgroup; bounded wait: false on a timeout, the slow path
// then still works wherever the members run).
// `allow` = false (option "cluster_local" = 0) answers false at once: the agent-scope protocol, for tests of that path.
__device__ __forceinline__ bool cluster_on_one_xcd(unsigned long long* hdr, int members, bool allow = true) {
  if (!allow) return false;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  __hip_atomic_fetch_or(&hdr[0], 1ull << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(&hdr[1], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  for (unsigned spins = 0; spins < 2000000u; ++spins) {
    if (__hip_atomic_load(&hdr[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)members) {
      const unsigned long long mask = __hip_atomic_load(&hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return (mask & (mask - 1)) == 0;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
// publication of a granule: `local` = cluster_on_one_xcd
__device__ __forceinline__ void cluster_publish(unsigned long long* p, unsigned long long v, bool local) {
  if (local) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int mds_kernel_attrs(dmp_ctx* c);     // once per device, at context creation
int gj_kernel_attrs(dmp_ctx* c);      // ditto (dca.hip)
int eigh_top8(dmp_ctx* c, const float* d_M, int L, float* d_mds, hipStream_t s);
// coords.hip
int coord_fc(dmp_ctx* c, const float* d_g, int L, float* d_ca, hipStream_t s);
int build_embed(const float* d_mat1d, const float* d_mds, int L, float* d_emb, hipStream_t s);
int transpose_f32(const float* d_in, int R, int C, float* d_out, hipStream_t s);
int pair_distances(const float* d_ca, int L, int clamp, float* d_dmap, hipStream_t s);
int fill_f32(float* d, int64_t n, float v, hipStream_t s);
int select_best(dmp_ctx* c, const float* d_conf, const float* d_ca, int L, int pass, int rec_cap,
                hipStream_t s);
int refine_coords(dmp_ctx* c, float* d_ca, int L, int steps, hipStream_t s);
int ca_to_backbone(const float* d_ca, const float* d_logit, int L, float* d_coords,
                   float* d_conf_out, hipStream_t s);
}  // namespace dmp
