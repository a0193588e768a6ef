/*
 * dmpfold_hip.h - C ABI of libdmpfold_hip.so: the DMPfold2 alignment -> coordinates
 * hot path as hand-written HIP for MI355X (gfx950).
 *
 * The reference (psipred/DMPfold2) has no FFI / plugin interface: its boundary is the
 * Python function dmpfold.aln_to_coords() (reference dmpfold/predict.py:74-158) whose
 * arithmetic lives in PyTorch operator calls.  Every entry point below replaces the
 * operator call sites named in its comment (file:line relative to the reference tree).
 * The Python shim dmpfold2_amd/predict.py binds them with ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; `d_` = device pointer, `h_` = host pointer;
 *   - every function returns 0 on success or a negative dmp_status; the message of the
 *     last failure on the calling thread is dmp_last_error();
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls only
 *     enqueue work, they never synchronise unless stated;
 *   - all floating point buffers are float32 row-major unless stated;
 *   - one dmp_ctx per GPU and per host thread; contexts are independent.
 */
#ifndef DMPFOLD_HIP_H
#define DMPFOLD_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 6): 65 functions.  New: the dmp_pipeline_* family (16 functions) - the throughput scheduler behind the C ABI;
 *    option "precision" accepts 2 (exact three-piece bf16 convolution and vertical GRU: full-width operands at the
 *    16-bit matrix cores' rate; "vgru_f32" accepts 2 for the latter alone).
 * 4 (round 5): 49 functions.  New: dmp_block_conv5x5_maxout_winners (the training slice's forward: maxout output + the
 *    winners autograd saves), dmp_head_conv_bwd, dmp_stem_maxout_winners and dmp_stem_bwd; dmp_block_conv5x5_maxout_bwd takes the saved winners (d_idx, NULL = run
 *    the forward again: the ABI-3 behaviour).  New options: precision (0 split-f16 / 1 the reference's float32 end to end),
 *    vgru_f32, gj_diag_blocked.
 * 3 (round 4): 45 functions instead of 60.  Removed: the experimental scheduler entry points that were measured
 *    slower (detached group chain, chain on its own stream, features ahead: six functions), the convenience wrappers
 *    predict_begin / predict_pass / predict_end_refine - dmp_predict and the unit calls cover them -, clear_faults and
 *    sync_check - dmp_sync_faults reports and clears -, profile_conv_ms and time_conv5x5 - dmp_profile_conv_intervals
 *    holds every launch's interval; the three lane functions became dmp_ctx_share_lane.  Options vgru_legacy and gj_lds are gone;
 *    act_scaling and vgru_persistent are new; fault bit DMP_FAULT_VGRU_HANDOFF is new.
 * 2: dmp_sync_faults clears what it reports, dmp_ctx_get_option and dmp_dca_features added. */
#define DMP_ABI_VERSION 5
#define DMP_MAX_SEQS 3000 /* predict.py:130-132: deeper MSAs are truncated */

typedef struct dmp_ctx dmp_ctx;

typedef enum dmp_status {
  DMP_OK = 0,
  DMP_ERR_ARG = -1,     /* bad argument (sizes, NULL, L < 8 ...) */
  DMP_ERR_HIP = -2,     /* a HIP runtime call failed */
  DMP_ERR_WEIGHTS = -3, /* unknown key, wrong shape, or weights incomplete */
  DMP_ERR_CAPACITY = -4, /* L or N exceeds what the context was created for */
  DMP_ERR_FAULT = -5     /* a device-side fault was recorded: results invalid */
} dmp_status;

/* Device-side fault bits (dmp_sync_faults).  A prediction during which a bit was raised returns NaN
 * coordinates and confidences. */
#define DMP_FAULT_SEQ_HANDOFF 1    /* sequence-GRU workgroup hand-off timed out */
#define DMP_FAULT_F16_RANGE 2      /* conv_mode 0: an activation reached |x| >= 6e4 (re-run in conv_mode 2) */
#define DMP_FAULT_REFINE_HANDOFF 4 /* minimiser workgroup hand-off timed out */
#define DMP_FAULT_EIG_HANDOFF 16   /* tridiagonalisation cluster: workgroup hand-off timed out */
#define DMP_FAULT_VGRU_HANDOFF 32  /* persistent vertical GRU: a row barrier between the workgroups of an XCD timed out */
#define DMP_FAULT_BAD_CODE 8       /* residue code > 21: the reference's embedding raises IndexError
                                      (network.py:223) */

int dmp_abi_version(void);
const char* dmp_last_error(void);

/* ---- context ------------------------------------------------------------------------
 * Allocates every device buffer the path needs for alignments up to max_N x max_L
 * (max_N is clamped to DMP_MAX_SEQS).  No allocation happens after this call.
 * DEVIATION from the reference, which has no length limit (network.py:218-221): 8 <= max_L <= DMP_MAX_L.
 * The eigensolver's kernels are instantiated per range of the order (<= 384: everything of the inverse iteration in
 * one workgroup's LDS; <= 1280: the vectors in LDS; <= 2048: in global memory; 8 x 256 rows per thread block of the
 * Householder step); every element count up to 442 L^2 and (21 L)^2 stays below 2^31 at 2048.  dmp_ctx_create answers
 * DMP_ERR_ARG ("max_L must be in [8, 2048]") for more, and the Python front end raises RuntimeError naming the
 * alignment's length before anything is computed.  The largest configuration of BASELINE.json (L = 1000) needs
 * 8.5 GB of the 288; L = 2048 with 3000 rows about 60 GB. */
#define DMP_MAX_L 2048
int dmp_ctx_create(int device, int max_L, int max_N, dmp_ctx** out);
void dmp_ctx_destroy(dmp_ctx* ctx);
/* (device memory held by the context: option "device_mib" of dmp_ctx_get_option, read only) */

/* Options (additive).  "conv_mode" selects how the 5x5 convolutions form their float32 products:
 *   0            each float32 operand split into two f16 pieces, 3 f16 MFMA products, float32
 *                accumulation - same error against float64 as a float32 convolution, 5.3x the
 *                f32 matrix-core rate; activations must stay inside the f16 range (|x| < 6e4,
 *                checked on the device, reported by dmp_sync_faults);
 *   1            the f32 matrix-core instruction (bitwise an fmaf chain);
 *   2 (default)  exact 3-way bf16 split, 6 bf16 MFMA products (no range limit, 2.7x the f32 rate).
 * "conv_f32_exact" = 1 is shorthand for conv_mode 1 (0 restores the default).
 * "precision" = 0 / 1 / 2 is the end-to-end switch: 0 = the FAST mode (split-f16 products of 22-23-bit operands in the
 * convolutions AND in the vertical GRU, whose gates use the hardware v_exp_f32 / v_rcp_f32; 1.8 x the speed of 2);
 * 1 = the reference's arithmetic instruction for instruction - conv_mode 1 and the float32 vertical GRU
 * (v_mfma_f32_16x16x4_f32 products, the device library's expf / tanhf in the gates: nn.GRU in float32, network.py:189,
 * 223-224); with it no f16 / bf16 matrix-core kernel runs; 2 (round 6) = FULL-WIDTH operands at the 16-bit matrix cores'
 * rate - conv_mode 2 (every float32 operand as three exact bf16 pieces = 24 significand bits, the six piece products
 * above 2^-24 accumulated in float32) and the vertical GRU the same way ("vgru_f32" 2: three bf16 pieces per operand
 * of its three products, float32 accumulation, the library gate functions of setting 1): A CONTEXT'S INITIAL SETTING (the
 * reference computes in float32: predict.py:136, network.py:25-31) and what the drop-in entry points run; 1.7 x the
 * speed of setting 1.
 * Reads back 0 / 1 / 2, or -1 for a mixed setting.
 * "vgru_f32" = -1 / 0 / 1 / 2: the vertical GRU alone (initially 2, with "precision" 2); -1 follows the convolution (float32 exactly when
 * conv_mode is 1, so "conv_mode" 1 and "precision" 1 select the same thing), 0 / 1 / 2 force the split-f16 / the float32
 * / the three-piece bf16 form whatever the convolution does.  Reads back what the next prediction will run (0 / 1 / 2).  The float32 form costs
 * 5.3 x the matrix-core time of the split form (41 against 15 ms for one alignment of 2000 x 300, 196 against 72 ms
 * for a chain of eight); its fallback without the persistent launch ("vgru_persistent" = 0) is the same kernel, one
 * launch per row, the same bits.
 * "tridiag_single" = 1 runs the Householder tridiagonalisation of the MDS eigensolver in a single
 * workgroup (one launch) instead of one multi-workgroup launch per step; same algorithm, different
 * summation order (results agree to float64 rounding).
 * "tridiag_cluster" (default 1): orders up to 640 run
 * every Householder step in ONE launch on a cluster of 32 workgroups of one XCD; 0 = one launch per step (hipGraph
 * chain).  The same algorithm with the float64 partial sums associated differently: the float32 results are the same bits
 * on full-rank matrices, the near-null columns of a rank-deficient Gram matrix can differ in their last bits (a
 * bit-for-bit comparison of two predictions needs the same setting); the cluster is faster for one prediction (0.9 against 1.9 ms at
 * order 300), the launches disturb the convolutions of other contexts less (the multi-engine scheduler sets 0).
 * "cluster_local" (default 1): the cluster kernels (sequence GRU, minimiser, tridiagonalisation) publish their
 * hand-off granules with plain stores when they find all their workgroups on one XCD (run-time check); 0 = always
 * agent-scope stores, the protocol that does not depend on placement.  Same bits, 1.84 against 2.8 us per GRU step.
 * "gj_diag_groups" = 2 / 4 / 8: threads (x 128) of the one-workgroup diagonal sweep of the inverse; same bits; 4 is the
 * default (8.9 against 10.0 ms per inverse at D = 6300 with 2, the form of rounds 1-3).
 * "gj_diag_blocked" (round 5, default 0): 1 = the 128 x 128 diagonal block of a block step is swept in 8 sub-blocks of 16
 * pivots (the pivot block inside one wave, W = T P and the rank-16 update on the f32 matrix cores) instead of as a chain
 * of 128 barrier-synchronised pivots (0: the arithmetic of rounds 1-4; "gj_diag_groups" applies to it).  The blocked form
 * is faster (6.6 against 7.65 ms per inverse at D = 6300, 18.4 against 20.2 at D = 10 500) and closer to the float64
 * inverse (4.5e-7 against 8.7e-7 of max |inv| at D = 630); the two agree to 1.1e-6, not bit for bit - and on three of the
 * reference's recycling fixtures (L = 200, ten passes; the x 4 weight set) that difference, amplified by eleven trunk
 * passes, moves a confidence by 1.4 .. 2.0e-4 from the reference's where the chain's stays below 1e-4, so the chain
 * remains the default of the prediction path.
 * "gj_pairs" (default 1): the inverse takes its 128-column block steps in pairs - one pass of the trailing update over
 * the matrix tiles per two steps (119 against 173 ms at D = 21 000, 20.4 against 26.0 at D = 10 500, equal at D = 6300);
 * 0 = one pass per step.  Same bits.
 * "gj_lookahead" = 0 / 1 / 2 (with gj_pairs = 0 only): the inverse's look-ahead - the next block step's diagonal sweep and panels on the context's
 * second stream beside this step's trailing update - where one prediction has the device to itself (dmp_predict,
 * dmp_spd_inverse, dmp_dca_features): 0 never, 1 (default) from 64 tile rows on (D > 8064: 23.5 against 25.4 ms at
 * D = 10500; at D = 6300 it loses, 8.7 against 7.7 ms), 2 at every size.  Same bits in every mode.
 * "conv_tile_bands" = 0 / 1 / 2 (round 6): pixel tile of the split-product convolutions - 1 = 16 x 16, 2 = 8 rows x 16
 * columns (twice the workgroups), 0 (default) = 8 x 16 where the 16 x 16 shape would leave half the CUs without a workgroup
 * (L <= 80), 16 x 16 above.  The same bits in every setting.
 * "act_scaling" (default 1): conv_mode 0 takes the f16 pieces of 2^e x activation, with e chosen per residual block
 * at dmp_weights_finalize from the InstanceNorm weights of the blocks before it (a bound of the residual stream), so the
 * low pieces of the bulk of the activations are normal f16 numbers whether the trunk sits at 1e-3 or at 1e4, and a trunk
 * that would leave the f16 range unscaled is scaled down instead of faulting; 0 = unscaled pieces (the round-3
 * arithmetic).  "act_scale_log2_block<k>" (k = 1..16, read only) answers e of block k's input.
 * "vgru_persistent" (default 1): the vertical GRU's chain as ONE weight-stationary launch (columns partitioned over the
 * XCDs, hidden units over the CUs of an XCD, XCD-local row barriers); 0 = one launch per alignment row (the round-3
 * group step kernel, also the fallback on a device without 256 CUs).  Reads back 0 where the persistent form is not
 * available.  The two forms sum K in different orders: results agree to float32 rounding.
 * "vgru_debug_drop_wg" = 1 (TESTS ONLY): the persistent chain is launched one workgroup short, so one XCD's row barrier can
 * never complete - the situation of a second process holding CUs.  The kernel must time out ONCE (a quarter of a second),
 * raise DMP_FAULT_VGRU_HANDOFF and leave its row loop; the Python layer then repeats with one launch per row.
 * "refine_single" = 1 runs the minimiser (dmp_refine_coords, dmp_predict*) in one workgroup instead
 * of a cluster of 16 that hands the coordinates over every step; same iteration, different
 * partial-sum slices (results agree to float32 rounding). */
int dmp_ctx_set_option(dmp_ctx* ctx, const char* name, int value);
/* Current value of an option of dmp_ctx_set_option ("conv_f32_exact" reads as conv_mode == 1). */
int dmp_ctx_get_option(const dmp_ctx* ctx, const char* name, int* h_value);

/* ---- weights (the reference's state_dict ABI, network.py:182-215) ---------------------
 * dmp_weights_set: hand over one tensor of GRUResNet(512,128).state_dict() by key, host
 * float32, contiguous, with its shape (replaces load_state_dict, predict.py:98).
 * dmp_weights_finalize: checks that all 184 tensors arrived and builds the packed device
 * layouts the kernels read (transposed GRU matrices, K-major conv slabs, the cSE gate
 * sigma(W2 relu(W1 beta)) of every block).  A rejected tensor (unknown key, wrong shape) or an
 * incomplete set drops everything staged so far: the next state_dict starts from nothing. */
int dmp_weights_set(dmp_ctx* ctx, const char* key, const float* h_data, const int64_t* shape,
                    int ndim);
int dmp_weights_finalize(dmp_ctx* ctx);
/* Several contexts on one GPU with the same weights (the engines of a throughput scheduler): `dst` uses the packed
 * device buffers of `src` instead of packing its own (0.6 s of host work and 0.5 GB per context at the reference's
 * model size).  The buffers are reference counted: either context may be destroyed or re-loaded first. */
int dmp_weights_share(dmp_ctx* dst, const dmp_ctx* src);

/* ---- host-side text -> residue codes (predict.py:124-128) -----------------------------
 * Translates nbytes alignment characters to codes: ARNDCQEGHILKMFPSTWYV -> 0..19,
 * BJOUXZ -> 20, '-' '.' -> 21, any other byte b -> (b - 65) mod 256.  Pure host code. */
int dmp_msa_encode(const uint8_t* h_text, int64_t nbytes, uint8_t* h_codes);

/* ---- feature builder ----------------------------------------------------------------- */
/* reweight(), predict.py:32-37.  d_msa: N x L codes (uint8, 0..21). d_w: N floats.
 * Integer exact: w[n] = 1 / #{m : #{l : min(a_nl,20)==min(a_ml,20)} > float32(L*0.8)}. */
int dmp_msa_weights(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, float* d_w, void* stream);
/* fast_dca() first half, predict.py:45-51: cov_reg (21L x 21L). */
int dmp_cov_build(dmp_ctx* ctx, const uint8_t* d_msa, const float* d_w, int N, int L,
                  float* d_cov, void* stream);
/* torch.inverse, predict.py:53: in-place inverse of a symmetric positive definite D x D
 * matrix (blocked Gauss-Jordan, fp32 MFMA trailing updates). */
int dmp_spd_inverse(dmp_ctx* ctx, float* d_A, int D, void* stream);
/* predict.py:58-60: APC-corrected contact channel (L x L) from the inverse covariance.
 * The 441 coupling channels are never materialised: the stem reads d_inv in place. */
int dmp_dca_contacts(dmp_ctx* ctx, const float* d_inv, int L, float* d_contacts, void* stream);

/* reweight() + fast_dca() in one call, with fast_dca's return value laid out as the reference returns it
 * (predict.py:41-61; train.py:175-190 computes the same tensor in its data loader): d_out (L x L x 442),
 * d_out[i][j][21a+b] = inv_cov[21i+a][21j+b], d_out[i][j][441] = APC-corrected contacts.  N = 1 gives zeros
 * (the glue of predict.py:139).  The prediction path itself never builds this tensor (the stem reads the
 * inverse in place); this export is for consumers that want the features themselves.  Uses the context's
 * covariance workspace: not to be interleaved with a prediction on the same context. */
int dmp_dca_features(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, float* d_out, void* stream);

/* ---- sequence trunk ------------------------------------------------------------------- */
/* embed + vgru, network.py:223-224: 2-layer GRU down the alignment (time = N, batch = L);
 * d_out: L x 512, top-layer state after the last row. */
int dmp_gru_vertical(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, float* d_out,
                     void* stream);
/* The same for n <= 8 alignments at once (contexts of one GPU holding the same weights, their alignments may
 * differ in N and L): ONE launch per alignment row serves the columns of all members, and the weight
 * fragments a workgroup fetches are shared by its column tiles - the independent targets of a throughput job are
 * the batch axis the reference's GRU call (batch = L, network.py:224) does not have.  Each member's result is
 * bit-identical to dmp_gru_vertical on it alone.  Everything is enqueued on `stream`; ctxs[0] leads (its
 * record buffers are used). */
int dmp_gru_vertical_group(dmp_ctx* const* ctxs, int n, const uint8_t* const* d_msas, const int* Ns,
                           const int* Ls, float* const* d_outs, void* stream);
/* hgru (which = 0; network.py:225, in T x 512) or coord_gru (which = 1; network.py:253,
 * in T x 520): multi-layer bidirectional GRU along the sequence, batch 1.
 * d_out: T x 512 (forward | reverse halves). */
int dmp_gru_bidir(dmp_ctx* ctx, int which, const float* d_in, int T, float* d_out, void* stream);

/* ---- pair trunk ------------------------------------------------------------------------ */
/* network.py:227-229 + the input-independent part of the stem 1x1 convolution
 * (network.py:25-27, 194): Z0[o,i,j] = b_o + sum_c W[o,c] m[c,i] m[c,j]
 *   + sum_ab W[o,512+21a+b] inv[21i+a,21j+b] + W[o,953] contacts[i,j]   (o < 384).
 * d_mat1d: 512 x L.  d_inv / d_contacts may both be NULL (single-sequence MSA: zero
 * features, predict.py:139).  d_z0: 384 x L x L. */
int dmp_stem_static(dmp_ctx* ctx, const float* d_mat1d, const float* d_inv,
                    const float* d_contacts, int L, float* d_z0, void* stream);
/* Rest of the stem for one trunk pass: + W[o,954]*dmap, max over channel triples,
 * InstanceNorm (network.py:28-32).  d_dmap: L x L.  d_x: 128 x L x L. */
int dmp_stem_update(dmp_ctx* ctx, const float* d_z0, const float* d_dmap, int L, float* d_x,
                    void* stream);
/* ResNet block `block` (1..16), network.py:94-103, as its two kernels:
 * conv 5x5 (128->512) + bias + max over channel quadruples -> d_u (128 x L x L) and the
 * per-channel sum / sum of squares (d_stats: 128 x 2 doubles) ... */
int dmp_block_conv5x5_maxout(dmp_ctx* ctx, int block, const float* d_x, int L, float* d_u,
                             double* d_stats, void* stream);
/* ... then InstanceNorm, scSE gates and the residual add: d_out = d_x + y*(cSE + sSE). */
int dmp_block_norm_scse_residual(dmp_ctx* ctx, int block, const float* d_u,
                                 const double* d_stats, const float* d_x, int L, float* d_out,
                                 void* stream);
/* Training-side slice: backward of the first half of ResNet block `block` - Maxout2d's convolution and
 * max (network.py:25-31 with kernel 5), the piece of autograd train.py:318-344 runs through ResNet_Block
 * (network.py:85-103).  d_x: the block's input (128 x L x L), d_du: gradient w.r.t. the maxout output (128 x L x L).
 * Outputs: d_dx (128 x L x L) gradient w.r.t. the input, d_dw (512 x 128 x 5 x 5) and d_db (512) gradients of
 * layer1.lin.weight / .bias (overwritten, not accumulated).  Implicit GEMMs on the float32 matrix cores (round 5: no
 * patch matrix): the forward is run again in float32 for the winner of every quadruple (ties go to the first maximal
 * channel, as torch.max), the routed gradient travels as d_du + one winner byte per maxout channel and pixel and is
 * expanded only in LDS; dgrad = the forward's tile structure with flipped taps, wgrad = a pixel-K GEMM over 8 x 16
 * pixel tiles reduced in fixed order (deterministic).  EVALUATION MODE ONLY: the dropouts of network.py:96-97 are
 * identities here; a training-mode block (Dropout / Dropout2d masks ahead of layer1) is not representable.
 * Workspace: one allocation per context at the first call of either backward entry point, sized for the context's
 * max_L (2.6 activation tensors: padded input, winners, flipped weight pack, wgrad partial sums) - L may change between
 * calls without reallocation.  Uses the context's maxout scratch plane (not to be interleaved with a prediction in
 * flight on the same context). */
int dmp_block_conv5x5_maxout_bwd(dmp_ctx* ctx, int block, const float* d_x, const float* d_du, const uint8_t* d_idx, int L,
                                 float* d_dx, float* d_dw, float* d_db, void* stream);
/* The forward a training step runs for that block: convolution + maxout in float32, d_u (128 x L x L), and what autograd
 * saves for the backward of torch.max (network.py:31) - d_idx (128 x L x L bytes): which channel of each quadruple won,
 * 0..3, the FIRST maximal one.  Handed to dmp_block_conv5x5_maxout_bwd as d_idx it saves the backward its own forward
 * (a third of its matrix-core work); with d_idx = NULL the backward runs the forward again.  (Two implementations of the
 * convolution can resolve a near-tie - two channels within float32 rounding of each other - differently, about one
 * decision in a million; both are valid subgradients.  The parity tests substitute the reference's winners at the
 * near-ties their fixtures list.) */
int dmp_block_conv5x5_maxout_winners(dmp_ctx* ctx, int block, const float* d_x, int L, float* d_u, uint8_t* d_idx,
                                     void* stream);
/* ... and of its second half - InstanceNorm (network.py:32), scSE (network.py:36-83) and the residual add
 * (network.py:99-101).  d_u: the maxout output the forward normalised (128 x L x L; its statistics are recomputed),
 * d_dout: gradient w.r.t. the block's output.  Outputs: d_du (128 x L x L) = gradient w.r.t. the maxout output (the
 * d_du argument of the entry point above) and d_dparams (2433 floats, overwritten): layer1.norm.weight 128,
 * layer1.norm.bias 128, scSE.cSE.fc.0.weight 8 x 128, scSE.cSE.fc.2.weight 128 x 8, scSE.sSE.conv.weight 128,
 * scSE.sSE.conv.bias 1.  The residual branch is the identity: the gradient w.r.t. the block's input is d_dx of the
 * first half plus d_dout (the caller's add).  Evaluation-mode block (the dropouts of network.py:96-97 are identities). */
int dmp_block_norm_scse_residual_bwd(dmp_ctx* ctx, int block, const float* d_u, const float* d_dout, int L,
                                     float* d_du, float* d_dparams, void* stream);
/* ... and of the 1x1 head convolution (network.py:207, Conv2d(128 -> 2)): d_x its input (128 x L x L), d_g the gradient
 * w.r.t. its two output planes (2 x L x L).  Outputs: d_dx (128 x L x L) and d_dparams (258 floats: weight 2 x 128,
 * bias 2).  With the two entry points above a caller chains the sixteen blocks and the head of net.resnet
 * (tests/test_gpu_train.py). */
int dmp_head_conv_bwd(dmp_ctx* ctx, const float* d_x, const float* d_g, int L, float* d_dx, float* d_dparams, void* stream);
/* ... and of the stem, resnet[0] = Maxout2d(955 -> 128, pool 3, kernel 1) + InstanceNorm (network.py:194, 12-34).  Its
 * 955-channel input is never materialised: channels 0..511 are the outer product of d_mat1d (512 x L, network.py:226-227),
 * 512..953 the covariance planes the context holds after dmp_stem_static for this target, 954 the distance channel.
 * dmp_stem_maxout_winners: from d_z0 (dmp_stem_static) and d_dmap the maxout output d_u (128 x L x L, before the
 * InstanceNorm) and the winner of every triple, d_idx (128 x L x L bytes, 0..2).  dmp_stem_bwd: d_dy = gradient w.r.t. the
 * stem's output; outputs d_dw (384 x 955 = resnet.0.lin.weight.grad), d_dparams (640 floats: lin.bias.grad 384,
 * norm.weight.grad 128, norm.bias.grad 128) and d_dmat1d (512 x L: what flows on into the sequence trunk).  With these and
 * the block / head entry points above a caller runs autograd's backward through the whole of net.resnet
 * (tests/test_gpu_train.py). */
int dmp_stem_maxout_winners(dmp_ctx* ctx, const float* d_z0, const float* d_dmap, int L, float* d_u, uint8_t* d_idx, void* stream);
int dmp_stem_bwd(dmp_ctx* ctx, const float* d_u, const uint8_t* d_idx, const float* d_dy, const float* d_mat1d,
                 const float* d_dmap, int L, float* d_dw, float* d_dparams, float* d_dmat1d, void* stream);
/* Head 1x1 conv (network.py:207) + network.py:237-246: d_conf (L) = row means of channel 1,
 * d_M (L x L) = Gram matrix 0.5*(dm_0j^2 + dm_i0^2 - dm_ij^2) of dm = |sym(channel 0)|. */
int dmp_head_gram(dmp_ctx* ctx, const float* d_x, int L, float* d_conf, float* d_M,
                  void* stream);
/* One whole trunk pass on internal buffers: stem_update + 16 blocks + head_gram. */
int dmp_trunk_pass(dmp_ctx* ctx, const float* d_z0, const float* d_dmap, int L, float* d_conf,
                   float* d_M, void* stream);

/* ---- coordinates ----------------------------------------------------------------------- */
/* torch.symeig + scaling, network.py:247-250: the 8 algebraically largest eigenpairs of the
 * symmetric d_M (upper triangle used), eigenvalues clamped to >= 1e-8, d_mds (L x 8) =
 * V*sqrt(lambda) in ascending order.  Solved on the device in float64.  Sign rule: each
 * eigenvector's largest-magnitude component is positive. */
int dmp_eigh_top8(dmp_ctx* ctx, const float* d_M, int L, float* d_mds, void* stream);
/* network.py:251-255: coord_gru on [mat1d^T | mds] then coord_fc -> d_ca (L x 3). */
int dmp_coords_from_mds(dmp_ctx* ctx, const float* d_mat1d, const float* d_mds, int L,
                        float* d_ca, void* stream);
/* network.py:272: d_dmap[i,j] = sqrt(max(|ca_i - ca_j|^2, 1e-8)); clamp = 0 gives the
 * template form of predict.py:143 (no clamp, zero diagonal). */
int dmp_pair_distances(dmp_ctx* ctx, const float* d_ca, int L, int clamp, float* d_dmap,
                       void* stream);
/* refine_coords(), network.py:106-137, all steps in one launch; in place on d_ca (L x 3). */
int dmp_refine_coords(dmp_ctx* ctx, float* d_ca, int L, int steps, void* stream);
/* calpha_to_main_chain() + sigmoid, network.py:141-177, 311-312.
 * d_coords: L x 5 x 3 (N, CA, C, O, CB); d_conf_out: L. */
int dmp_ca_to_backbone(dmp_ctx* ctx, const float* d_ca, const float* d_conf_logit, int L,
                       float* d_coords, float* d_conf_out, void* stream);

/* ---- the whole path -------------------------------------------------------------------- */
/* GRUResNet.forward + the feature glue of aln_to_coords (predict.py:134-153,
 * network.py:218-314).  d_msa: N x L codes (N already capped).  d_template_ca: Lt x 3 or NULL
 * (seed distance channel = -1).  nloops = recycling iterations, refine_steps = minimiser
 * steps.  Outputs d_coords (L x 5 x 3) and d_conf (L).  No host synchronisation. */
int dmp_predict(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L, const float* d_template_ca,
                int Lt, int nloops, int refine_steps, float* d_coords, float* d_conf,
                void* stream);

/* The same prediction issued unit by unit, so that one host thread can interleave several contexts (different
 * streams).  dmp_predict_begin_units validates and records the arguments without enqueueing anything; the
 * prediction is then a sequence of units handed out by dmp_predict_issue_unit: first the front end in chunks of about 2 ms of GPU work
 * (sequence weights + covariance; the inverse, 6 block steps at a time, then contacts; the
 * vertical GRU, 128 alignment rows at a time; sequence GRU + static stem), then 18 units per pass:
 * unit 0 = recycled distance map + stem update, units 1..16 = residual block k (its conv5x5 takes
 * the lane), unit 17 = head + Gram matrix + MDS + coordinate GRU + best-of update.
 * dmp_predict = begin_units + every unit + dmp_predict_end.  dmp_predict_next_unit tells what the next dmp_predict_issue_unit would
 * enqueue; dmp_ctx_pending returns how many issued units have not completed on the GPU (0, 1, or 2
 * for "two or more"; negative on error), so a scheduler can hand the lane only to contexts whose
 * next convolution can start at once and keep its own issue loop from running far ahead. */
#define DMP_UNIT_NONE 0   /* every pass of this prediction was issued: call dmp_predict_end */
#define DMP_UNIT_LIGHT 1  /* no convolution in the unit */
#define DMP_UNIT_CONV 2   /* residual block: one lane turn */
#define DMP_UNIT_WAIT 3   /* nothing to issue now: the unit waits for another context's units (dmp_predict_group_vgru) */
int dmp_predict_begin_units(dmp_ctx* ctx, const uint8_t* d_msa, int N, int L,
                            const float* d_template_ca, int Lt, int nloops, int refine_steps);
int dmp_predict_next_unit(const dmp_ctx* ctx);
/* Vertical GRUs of n <= 8 predictions as ONE launch chain (dmp_gru_vertical_group inside the unit machinery): call
 * right after dmp_predict_begin_units on every member, before any of their units is issued.  ctxs[0] leads: its
 * vertical-GRU units serve all members on the stream its units are issued on (which must be ordered behind the
 * producers of every member's alignment), the other members have no vertical-GRU units; their last front-end unit
 * waits (event) for the leader's last vertical-GRU unit, and dmp_predict_next_unit answers DMP_UNIT_WAIT for it
 * until that unit has been issued.  Results are bit-identical to ungrouped predictions.  The members must hold
 * the same weights; the leader cannot begin another prediction before every member has issued that unit. */
int dmp_predict_group_vgru(dmp_ctx* const* ctxs, int n);
/* Riders: the vertical GRUs of n alignments that are NOT being predicted yet (the scheduler's next targets) are
 * computed in the same launch chain as the group led by `lead` (members + riders <= 8; the chain's fixed cost per
 * alignment row - launch boundary, cold L2s - is shared by twice as many columns), their results (L_i x 512 each)
 * written to d_outs[i].  Call once, right after dmp_predict_group_vgru (a group of one is allowed), before the
 * leader issues a unit.  The leader's read-only option "chain_issued" answers
 * 1 once the chain has been enqueued to its end on the stream of the leader's units: an event recorded on that
 * stream from then on is behind the riders' results, which go to their predictions through
 * dmp_predict_set_vgru_result.  Every result is bit-identical to the alignment's own chain. */
int dmp_predict_group_riders(dmp_ctx* lead, int n, const uint8_t* const* d_msas, const int* Ns, const int* Ls,
                             float* const* d_outs);
/* The vertical GRU of this prediction has been (or is being) computed ahead of time by dmp_gru_vertical /
 * dmp_gru_vertical_group on the same alignment: d_vout (L x 512, device) is its result, `event` (hipEvent_t or
 * NULL) was recorded behind it.  Call right after dmp_predict_begin_units, before any unit is issued: the
 * prediction then has no vertical-GRU units and its last front-end unit waits for the event and reads d_vout -
 * a scheduler can run the launch chains of the NEXT targets beside the trunk passes of the current ones (the
 * chain touches none of the buffers the trunk uses).  d_vout must stay valid until that unit has run. */
int dmp_predict_set_vgru_result(dmp_ctx* ctx, const float* d_vout, void* event);
/* After the last unit: final refinement of the best trace + backbone + confidences into d_coords (L x 5 x 3) and
 * d_conf (L); a prediction during which a device-side fault was recorded returns NaN. */
int dmp_predict_end(dmp_ctx* ctx, float* d_coords, float* d_conf, void* stream);
int dmp_predict_issue_unit(dmp_ctx* ctx, void* stream);
int dmp_ctx_pending(dmp_ctx* ctx);

/* Throughput mode: contexts of ONE process that run on different streams may share a lane.  The
 * machine-filling conv5x5 launches of all contexts on a lane then take turns (cross-stream events),
 * two in flight at a time (a launch waits for the one before the previous one, so the tail of one
 * launch fills with the head of the next), while every other kernel of one target overlaps the
 * convolutions of another.  dmp_ctx_share_lane(ctx, other): ctx joins the lane of `other` (made on first use; it
 * lives as long as a context refers to it); other = NULL detaches.  All contexts of a lane must be driven by the same
 * host thread. */
int dmp_ctx_share_lane(dmp_ctx* ctx, dmp_ctx* other);

/* ---- throughput mode behind the C ABI (ABI 5, round 6) ------------------------------------------------------------
 * The batch form of the reference's one call per alignment (predict.py:74-158): N independent alignments through one
 * GPU.  A pipeline owns `engines` contexts (1..8; 4 is the measured optimum on one MI355X), each on a HIP stream of its
 * own, sharing one lane, and ONE host thread inside the library that schedules all of them unit by unit
 * (csrc/pipeline.hip: the units above, the lane, group start of the vertical GRUs, riders, tail stagger - nothing of
 * the issue loop runs in the caller's language).  Use:
 *   dmp_pipeline_create(device, max_L, max_N, 4, &p);
 *   dmp_weights_set(dmp_pipeline_ctx(p, 0), ...) x 184; dmp_weights_finalize(dmp_pipeline_ctx(p, 0));
 *   dmp_pipeline_weights_ready(p);                       (the other engines share that ONE packed copy)
 *   dmp_pipeline_set_option(p, "precision", 2);          (any option of dmp_ctx_set_option, on every engine; idle pipeline only)
 *   t = dmp_pipeline_submit(p, d_msa, N, L, d_template_ca | NULL, iterations, minsteps, d_coords, d_conf, ready_event | NULL);
 *   ... dmp_pipeline_poll(p, tickets, cap, &n) / dmp_pipeline_wait(p, 2) ... dmp_pipeline_status(p, t, &state, &fault_bits);
 *   dmp_pipeline_release(p, t);  dmp_pipeline_destroy(p);
 * Ownership: every buffer is the caller's (device memory; d_coords L x 5 x 3, d_conf L) and must stay valid until the
 * ticket is DMP_TICKET_DONE or DMP_TICKET_FAILED.  `ready_event` (hipEvent_t or NULL): recorded by the caller behind
 * whatever produces d_msa / d_template_ca; the engine that takes the target orders its stream behind it.  submit returns
 * the ticket (>= 0) or a negative dmp_status; it never blocks on the GPU and may be called from any thread.
 * dmp_pipeline_wait(p, what): block until every submitted target has 0 = been started on an engine, 1 = been issued to
 * its end (the engines' streams then hold all the work: dmp_pipeline_stream(p, i) can be waited for on another stream),
 * 2 = completed on the GPU.  dmp_pipeline_poll: tickets that completed (or failed) since the last call, each once.
 * dmp_pipeline_status: *h_state = DMP_TICKET_*; for a DONE ticket *h_fault_bits = the DMP_FAULT_* bits recorded during
 * THAT prediction (non-zero: its outputs are NaN; the caller repeats it, e.g. alone through dmp_predict with
 * "vgru_persistent" 0 or "conv_mode" 2 as the bit suggests); a FAILED ticket (a HIP / capacity error while it was being
 * issued) returns that error.  dmp_pipeline_release forgets a finished ticket (its event and fault word are reused).
 * Scheduler knobs are read from the environment at creation (DMP_VGRU_GROUP, DMP_VGRU_RIDERS, DMP_GROUP_PATIENCE,
 * DMP_TAIL_STAGGER: DESIGN.md section 6); results are bit-identical to dmp_predict on a lone context with
 * "tridiag_cluster" = 0, whatever the grouping. */
typedef struct dmp_pipeline dmp_pipeline;
#define DMP_TICKET_QUEUED 0
#define DMP_TICKET_RUNNING 1
#define DMP_TICKET_ISSUED 2
#define DMP_TICKET_DONE 3
#define DMP_TICKET_FAILED 4
int dmp_pipeline_create(int device, int max_L, int max_N, int engines, dmp_pipeline** out);
/* the same on `engines` streams of the caller's (hipStream_t[engines], non-blocking streams, used by nothing else while the
 * pipeline lives; not destroyed with it) - a host whose memory allocator tracks streams (PyTorch) passes its own */
int dmp_pipeline_create_on(int device, int max_L, int max_N, int engines, void* const* streams, dmp_pipeline** out);
void dmp_pipeline_destroy(dmp_pipeline* p);
int dmp_pipeline_engines(const dmp_pipeline* p);
dmp_ctx* dmp_pipeline_ctx(dmp_pipeline* p, int i);
void* dmp_pipeline_stream(dmp_pipeline* p, int i);
int dmp_pipeline_weights_ready(dmp_pipeline* p);
int dmp_pipeline_set_option(dmp_pipeline* p, const char* name, int value);
int64_t dmp_pipeline_submit(dmp_pipeline* p, const uint8_t* d_msa, int N, int L, const float* d_template_ca, int nloops,
                            int refine_steps, float* d_coords, float* d_conf, void* ready_event);
int dmp_pipeline_wait(dmp_pipeline* p, int what);
int dmp_pipeline_poll(dmp_pipeline* p, int64_t* h_tickets, int capacity, int* h_n);
int dmp_pipeline_status(dmp_pipeline* p, int64_t ticket, int* h_state, int* h_fault_bits);
int dmp_pipeline_release(dmp_pipeline* p, int64_t ticket);
int dmp_pipeline_backlog(dmp_pipeline* p, int* h_queued, int* h_running);
/* dmp_pipeline_pause(p, 1): no NEW target is started until (p, 0) - a caller that submits a batch in a loop pauses around
 * it so that the first vertical-GRU group is formed from the whole batch and not from whatever had arrived when the
 * scheduler looked (results do not depend on it, the first chain's width does). */
int dmp_pipeline_pause(dmp_pipeline* p, int on);
/* counters since creation: [0] vertical-GRU groups formed, [1] largest group, [2] chains that carried riders, [3] most
 * riders in one chain, [4] rider results not yet consumed (0 on an idle pipeline), [5] scheduling rounds that found nothing
 * to issue, [6] scheduling rounds, [7] CPU microseconds of the scheduler thread */
int dmp_pipeline_stats(dmp_pipeline* p, long long* h_stats, int capacity);

/* Synchronise `stream` and report the device-side faults recorded since the last report
 * (DMP_FAULT_* bits in *h_bits; 0 = every result handed out since then is valid).  Reporting clears
 * them: one failed prediction does not poison the checks of later ones. */
int dmp_sync_faults(dmp_ctx* ctx, void* stream, int* h_bits);

/* ---- introspection for tests and the benchmark ------------------------------------------- */
/* After dmp_predict: copy an internal tensor to d_dst (device).  Names: "w", "contacts",
 * "mat1d", "conf_means" (P floats), "ca_pass" (P x L x 3), "best_ca" (L x 3, before the final
 * refinement), "inv_cov" (21L x 21L), "mds" (L x 8) and "gram" (L x L) of the last pass.  Returns the number of floats written or a negative status. */
int64_t dmp_debug_fetch(dmp_ctx* ctx, const char* name, float* d_dst, int64_t capacity,
                        void* stream);
/* Optional HIP-event timing of every conv5x5 launch inside dmp_predict / dmp_trunk_pass / the unit calls (events
 * are recorded on the caller's stream around each launch; up to max_launches).  dmp_profile_enable(ctx, 1, n) starts a
 * fresh record, (ctx, 0, 0) stops recording.  dmp_profile_conv_intervals: start / end of every recorded launch of ctx
 * in ms after the first recorded launch of ref (streams synchronised by the caller); the record is kept. */
int dmp_profile_enable(dmp_ctx* ctx, int on, int max_launches);
int dmp_profile_conv_intervals(dmp_ctx* ctx, dmp_ctx* ref, float* h_start_ms, float* h_end_ms, int capacity,
                               int* h_launches);

#ifdef __cplusplus
}
#endif
#endif /* DMPFOLD_HIP_H */
