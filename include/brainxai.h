/* brainxai.h -- C ABI of libbrainxai.so: hand-written HIP (gfx950 / CDNA4) kernels for the
 * multimodal brain-pattern training + attribution hot path.
 *
 * The reference (KC-decoder/Multimodal-Brain-Pattern-Identification_XAI) has NO native code and no
 * FFI: its "native boundary" for this path is torch -> ATen -> cuDNN/cuBLAS.  Each entry point
 * below therefore cites the torch op chain of the reference it replaces
 * (M  = root/src/models/models.py, NB = root/jupyter_notebooks/XAI_Multimodality.py,
 *  DS = root/src/data/dataset.py, DDP = root/src/training/training_distributed.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless marked host;
 *   - the caller owns every buffer including workspaces; the library never allocates, frees
 *     or retains pointers past a call; no call synchronises the device;
 *   - all launches go to the hipStream_t passed as `stream` (void* to keep hip headers out);
 *   - activations are channels-last: [B, H, W, C] (NHWC) for the 2-D CNN, [B, F, Chans, T]
 *     (NCHW, as the reference) for the EEG branch;
 *   - `dtype` is the STORAGE type of activations: BX_F32 or BX_BF16; arithmetic accumulates in
 *     fp32 always; parameters, parameter gradients and statistics are fp32;
 *   - return 0 on success, a negative BX_E* otherwise; bx_last_error_string() (thread-local)
 *     says why.  Nothing throws or aborts across this boundary.
 */
#ifndef BRAINXAI_H
#define BRAINXAI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BX_VERSION 100

enum { BX_F32 = 0, BX_BF16 = 1 };
enum { BX_POOL_MAX = 0, BX_POOL_AVG = 1 };
enum { BX_OK = 0, BX_EINVAL = -1, BX_EDTYPE = -2, BX_EALIGN = -3, BX_EWORKSPACE = -4, BX_EHIP = -5,
       BX_EUNSUPPORTED = -6 };
/* conv algorithm selector: 0 = library default, 1 = direct VALU (any dtype), 2 = MFMA implicit GEMM */
enum { BX_ALGO_AUTO = 0, BX_ALGO_DIRECT = 1, BX_ALGO_MFMA = 2 };
/* epilogue flags of bx_conv3x3 */
enum { BX_EPI_RELU = 1, BX_EPI_MASK_BITS = 2 };   /* MASK_BITS: relu_mask_src is the bit form written by bx_conv3x3_pair (MFMA path, Ci <= 32) */

typedef void* bxStream;

int bx_version(void);
const char* bx_last_error_string(void);

/* ---- layout ------------------------------------------------------------------------------- */
/* fp32 NCHW [B,C,H,W] -> dtype NHWC [B,H,W,Cp], channels C..Cp-1 zero.  Replaces the implicit
 * NCHW hand-off of DataLoader batches into Block.forward (M:62-63, NB:1595-1598). */
int bx_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cp, int dtype, bxStream stream);
/* dtype NHWC [B,H,W,Cs] -> fp32 NCHW [B,C,H,W] (first C channels). */
int bx_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int Cs, int dtype, bxStream stream);

/* ---- 3x3 convolution, stride 1, pad 1  (nn.Conv2d + F.relu, M:49-51,64-66; autograd bwd) ---- */
/* Pack fp32 OIHW weights [Cout,Cin,3,3] for bx_conv3x3.
 *   transpose_flip = 0: forward operand   Wp[tap][cin][cout]            (cin padded to Cin_p)
 *   transpose_flip = 1: data-gradient operand Wp[tap][cout][cin] = W[cout][cin][8-tap]
 * `packed_f32` ([9][I_p][O_p] fp32) is always written; `packed_mfma` (may be NULL) receives the
 * bf16 MFMA-fragment layout (see csrc/conv3x3_mfma.hip). I_p/O_p = channel counts rounded up to 8/16. */
int bx_conv3x3_pack(const float* w_oihw, float* packed_f32, void* packed_mfma, int Cout, int Cin,
                    int I_p, int O_p, int transpose_flip, bxStream stream);
/* bytes of the MFMA operand for padded dims (0 when the MFMA path does not cover them). */
size_t bx_conv3x3_packed_mfma_bytes(int I_p, int O_p);
/* fp32 STORAGE on the bf16 matrix cores (csrc/conv3x3_split.hip): every fp32 operand is split x = h + m + l (three bf16, 24
 * significand bits) and a product runs as the six MFMAs whose partial products reach 2^-18 of the largest, into the fp32
 * accumulator -- fp32-grade results (see the file's header).  bx_conv3x3_pack_split writes the weight operand in that layout
 * (images h, m, l; bx_conv3x3_packed_split_bytes, 0 when the path does not cover the padded dims); hand it to bx_conv3x3 /
 * bx_conv3x3_carry as `packed_mfma` with dtype BX_F32.  Same torch ops as bx_conv3x3_pack (M:49-51,64-66). */
size_t bx_conv3x3_packed_split_bytes(int I_p, int O_p);
int bx_conv3x3_pack_split(const float* w_oihw, void* packed_split, int Cout, int Cin, int I_p, int O_p, int transpose_flip,
                          bxStream stream);
/* All MFMA weight operands of a model in ONE launch.  `jobs_device` is a device array of njobs descriptors sorted by
 * block_begin (job j owns launch blocks [block_begin_j, block_begin_{j+1})); total_blocks = end of the last job. */
typedef struct {
  const void* w_oihw;      /* fp32 [Cout,Cin,3,3] */
  void* packed_mfma;       /* bx_conv3x3_packed_mfma_bytes(I_p, O_p) bytes (bx_conv3x3_packed_split_bytes with the split bit) */
  int Cout, Cin, I_p, O_p, transpose_flip, block_begin;   /* transpose_flip: bit 0 = data-gradient operand, bit 1 = split (fp32-storage) layout */
} bxPackJob;
int bx_conv3x3_pack_many(const bxPackJob* jobs_device, int njobs, int total_blocks, bxStream stream);
/* The same launch also converts the batch: src fp32 NCHW [B,C,H,W] -> dst bf16 NHWC [B,H,W,Cp] (what bx_nchw_to_nhwc does;
 * the two are independent and both sit at the start of every training step). */
int bx_conv3x3_pack_many_layout(const bxPackJob* jobs_device, int njobs, int total_blocks, const float* src_nchw,
                                void* dst_nhwc_bf16, int B, int C, int H, int W, int Cp, bxStream stream);
/* Evaluation sweeps (Grad-CAM over a dataset, integrated-gradients passes): njobs may be 0 (total_blocks 0, jobs_device ignored) when
 * no parameter changed since the operands were last packed -- the launch is then the layout conversion alone -- and the batch's address
 * may come from a DEVICE slot (src_slot != NULL: *src_slot is read by the kernel, src_nchw is ignored), so that a captured graph can be
 * replayed on the caller's batch without copying it into a static buffer (bx_store_u64x2 fills the slot, stream-ordered). */
int bx_conv3x3_pack_layout_ex(const bxPackJob* jobs_device, int njobs, int total_blocks, const float* src_nchw,
                              const float* const* src_slot, void* dst_nhwc_bf16, int B, int C, int H, int W, int Cp, bxStream stream);
/* The launch that opens a training step of the multimodal model (reference forward NB:1095-1105 with nn.Dropout in both
 * branches): packing, the optional layout conversion (src_nchw / dst_nhwc_bf16 both NULL: none) and bx_seed_next2's work
 * (both branches' dropout counters advance, out_a / out_b receive the seeds of this forward pass) in one launch. */
int bx_conv3x3_pack_many_step(const bxPackJob* jobs_device, int njobs, int total_blocks, const float* src_nchw,
                              void* dst_nhwc_bf16, int B, int C, int H, int W, int Cp, uint64_t* state_a, uint64_t* out_a,
                              uint64_t* state_b, uint64_t* out_b, bxStream stream);
/* conv1 + ReLU and conv2 + ReLU of a stage-1 Block (models.py:64-65) in ONE launch: conv1's output tile stays in LDS and is
 * written to y1 only when y1 != NULL (a backward pass will read it; evaluation-mode passes hand NULL).  Built for bf16 storage and
 * 8 (padded) -> 16 -> 16 channels (bx_conv3x3_pair_supported); results are bit-identical to two bx_conv3x3 calls with BX_EPI_RELU.
 * x [B,H,W,C0_p], y1 / y2 [B,H,W,C1]; packed*_mfma from bx_conv3x3_pack / _pack_many (forward operands).  Also built for
 * 16 -> 32 -> 32 and 32 -> 64 -> 64.  mask1 / mask2 (nullable, only with y1): the ReLU decisions of y1 / y2 as bits, uint8 [B,H,W,C1/4], bit r of byte q =
 * channel 4q + r is positive -- what the layers' data gradients read instead of the activations (bx_conv3x3 with BX_EPI_MASK_BITS:
 * 1/8 of the bytes in the HBM-bound early stages). */
int bx_conv3x3_pair_supported(int C0_p, int C1, int C2, int dtype);
int bx_conv3x3_pair(const void* x, const void* packed1_mfma, const float* bias1, const void* packed2_mfma, const float* bias2,
                    void* y1, void* y2, unsigned char* mask1, unsigned char* mask2, int B, int H, int W, int C0_p, int C1, int C2,
                    int dtype, bxStream stream);
/* y = epi(conv3x3(x, Wp) + bias);  x [B,H,W,Ci] -> y [B,H,W,Co], both `dtype`.
 *   bias (fp32 [Co]) may be NULL; flags & BX_EPI_RELU applies max(.,0);
 *   relu_mask_src (dtype [B,H,W,Co], may be NULL): y *= (relu_mask_src > 0)  -- the ReLU backward
 *     of the layer that produced the tensor whose gradient this call computes;
 *   addend (dtype [B,H,W,Co], may be NULL) is added last (skip-path gradient).
 * Forward uses (bias, RELU); data-gradient uses the transpose_flip pack + relu_mask_src/addend.
 * `packed_mfma` is the MFMA operand in the layout of `dtype`: bx_conv3x3_pack's for BX_BF16, bx_conv3x3_pack_split's for BX_F32
 * (algo AUTO takes the MFMA path whenever it is given and the shape is covered, else the direct kernel on packed_f32). */
int bx_conv3x3(const void* x, const float* packed_f32, const void* packed_mfma, const float* bias,
               const void* relu_mask_src, const void* addend, void* y,
               int B, int H, int W, int Ci, int Co, int dtype, int flags, int algo, bxStream stream);
/* Weight/bias gradient: dW[Cout,Cin,3,3] = sum_p X[p+tap] (x) dZ[p], db = sum_p dZ[p]  (fp32 out,
 * overwritten).  x [B,H,W,Ci_p], dz [B,H,W,Co].  Cin = logical input channels (<= Ci_p). */
size_t bx_conv3x3_wgrad_workspace(int B, int H, int W, int Ci_p, int Co, int dtype, int algo);
int bx_conv3x3_wgrad(const void* x, const void* dz, float* dw_oihw, float* dbias, int B, int H, int W,
                     int Cin, int Ci_p, int Co, int dtype, int algo, void* workspace, size_t workspace_bytes,
                     bxStream stream);
/* Chained form: the fixed-order sum of a layer's weight-gradient partials (a ~6 us launch of its own, 15 per training step)
 * rides in the NEXT layer's launch instead.  *pending (zero-initialised before the first call) describes partials that still
 * have to be summed: a valid *pending is reduced by this call's launch and then overwritten with this layer's; the caller
 * must give consecutive calls DIFFERENT workspaces (the pending partials live in the previous one), keep that workspace and
 * the previous dw/dbias alive until the next call, and end the chain with bx_conv3x3_wgrad_finish (e.g. at the end of
 * backward).  Paths that cannot carry a reduce (fp32 / direct kernels) finish the chain themselves. */
typedef struct {
  const void* partial; float* dw; float* db;
  int nsplit, Cin, Co, ma, nb, ztiles, nfrag4, valid;
} bxWgradPending;
int bx_conv3x3_wgrad_chained(const void* x, const void* dz, float* dw_oihw, float* dbias, int B, int H, int W, int Cin,
                             int Ci_p, int Co, int dtype, int algo, void* workspace, size_t workspace_bytes,
                             bxWgradPending* pending, bxStream stream);
int bx_conv3x3_wgrad_finish(bxWgradPending* pending, bxStream stream);
/* bx_conv3x3 that also ends a weight-gradient chain: a valid *pending is summed by extra workgroups of THIS launch (the
 * last weight gradient of a Block's backward is followed by that layer's data gradient, which then carries the sum
 * instead of a reduce launch of its own); *pending is invalid on return.  Paths that cannot carry it (fp32 / direct
 * kernels) finish the chain with a separate launch first.  Same arguments and results as bx_conv3x3 otherwise. */
int bx_conv3x3_carry(const void* x, const float* packed_f32, const void* packed_mfma, const float* bias,
                     const void* relu_mask_src, const void* addend, void* y, int B, int H, int W, int Ci, int Co,
                     int dtype, int flags, int algo, bxWgradPending* pending, bxStream stream);

/* Measurement hook: the next bx_block_conv3_tail_fwd call of this host thread records the two hipEvent_t (created by the caller,
 * timing enabled) immediately before and after its convolution kernel on the call's stream -- bench.py times the fused conv3 +
 * pool launch of the training step with it (events cannot bracket one kernel of a multi-launch call from outside).  One-shot. */
int bx_profile_next_conv3(void* ev_start, void* ev_stop);

/* ---- Block tail: 2x2 pool -> BatchNorm2d -> Dropout -> + conv1x1(bilinear(x))  (M:67-76) ------ */
typedef struct {
  int B, H, W;          /* conv3 output resolution; pooled map is [B, H/2, W/2] (floor) */
  int Cin_p;            /* channels of the block input x as stored (padded)            */
  int C;                /* block output channels                                        */
  int pool;             /* BX_POOL_MAX | BX_POOL_AVG                                    */
  int training;         /* 1: batch statistics + running-stat update; 0: running stats  */
  float eps, momentum;  /* 1e-5, 0.1 (nn.BatchNorm2d defaults)                          */
  float dropout_p;      /* 0 disables; mask = hash(seed[0], salt, element)              */
  uint32_t salt;
  int dtype;
  uint32_t* sync;       /* NULL, or BX_TAIL_SYNC_WORDS device words owned by THIS block instance, zero before the first call
                         * (the library leaves them zero): where bx_set_tree_max_rows allows it, the batch-statistics
                         * finalizes ride in the kernels that produce the partial sums (last workgroup to arrive).
                         * One block instance must not run on two streams at once with the same words. */
  void* route;          /* NULL, or bx_block_tail_route_bytes(d) bytes: one NIBBLE per pooled element saying which positions of its
                         * 2x2 window (bit q = row-major position q) receive its gradient -- the arg-max if positive (max pool,
                         * first maximum as ATen) or the positive ones (average pool).  That is all the backward needs of conv3's
                         * full-resolution output: bx_block_conv3_tail_fwd writes the nibbles (and may then be given y3 = NULL:
                         * the output is not stored), bx_block_tail_bwd reads them instead of y3 (which may be NULL).  1/16 of the
                         * bytes of y3 in each direction.  bf16 fused path only; ignored by bx_block_tail_fwd. */
} bxTailDesc;
size_t bx_block_tail_route_bytes(const bxTailDesc* d);
#define BX_TAIL_SYNC_WORDS 8192
#define BX_TAIL_SYNC_FWD 0          /* word offsets inside sync: forward statistics | backward statistics */
#define BX_TAIL_SYNC_BWD 4096
size_t bx_block_tail_workspace(const bxTailDesc* d);
/* y3: conv3 output [B,H,W,C]; x: block input [B,H,W,Cin_p]; w1x1 fp32 [C][Cin] OIHW(1x1), Cin logical;
 * bn_* fp32 [C]; num_batches_tracked int64[1]; seed uint64[1] (device, may be NULL if dropout_p==0).
 * Outputs: pooled [B,H/2,W/2,C] (pre-BN, kept for backward), out [B,H/2,W/2,C],
 * save_mean/save_invstd fp32 [C] (statistics actually used). */
int bx_block_tail_fwd(const bxTailDesc* d, const void* y3, const void* x, const float* w1x1, int Cin,
                      const float* b1x1, const float* bn_weight, const float* bn_bias,
                      float* running_mean, float* running_var, int64_t* num_batches_tracked,
                      const uint64_t* seed, void* pooled, void* out, float* save_mean, float* save_invstd,
                      void* workspace, size_t workspace_bytes, bxStream stream);
/* Policy of the in-launch finalizes (bxTailDesc.sync): a statistics pass with at most `rows` rows of partial sums is finished by
 * its last workgroup, a larger one by a separate finalize launch whose workgroups split the channels.  Default 0 = always the
 * separate launch (or the environment variable BX_TREE_MAX_ROWS at first use): measured on MI355X the in-launch form costs
 * every workgroup a drained store + ticket round trip and was 4-10 us SLOWER per pass than the ~6.5 us launch it removes
 * (DESIGN section 6).  Results do not depend on the choice beyond the rounding of double-precision sums. */
int bx_set_tree_max_rows(int rows);
/* Folded finalizes (round 3): the cross-workgroup sum of a statistics pass moves to the START of the kernel that consumes it -- every
 * workgroup of the apply kernel sums the producer's partial rows itself, in one fixed order, so no finalize launch (k_bn_finalize /
 * k_tail_bwd_mid) and no atomics are needed.  mask: bit 0 = backward passes, bit 1 = forward passes; default 3 (or the environment
 * variable BX_TAIL_FOLD at first use); a pass is folded only while rows x C stays small (the backward reduction launches
 * BX_TAIL_FOLD_RC / C rows, default 8192 / C; the forward folds when its producer wrote at most BX_TAIL_FOLD_RC_FWD / C rows,
 * default 16384 / C -- the late stages).  Results do not depend on the choice beyond the rounding of double-precision sums. */
int bx_set_tail_fold(int mask);
/* conv3 + tail forward in two launches (bf16 storage, MFMA-capable C; otherwise BX_EUNSUPPORTED and the caller uses
 * bx_conv3x3 + bx_block_tail_fwd): y3 = relu(conv3x3(y2, w3) + b3) is stored for backward, and conv3's epilogue also
 * writes pooled = pool2x2(y3) and the batch statistics, so the pool never re-reads y3 from HBM.  w3_mfma is conv3's
 * MFMA operand from bx_conv3x3_pack (flip 0).  Same workspace size and the same
 * outputs as bx_block_tail_fwd (M:62-76). */
int bx_block_conv3_tail_fwd(const bxTailDesc* d, const void* y2, const void* w3_mfma, const float* b3, void* y3,
                            const void* x, const float* w1x1, int Cin, const float* b1x1, const float* bn_weight,
                            const float* bn_bias, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                            const uint64_t* seed, void* pooled, void* out, float* save_mean, float* save_invstd,
                            void* workspace, size_t workspace_bytes, bxStream stream);
/* Backward of the tail.  dout [B,H/2,W/2,C].  Produces
 *   dz3 [B,H,W,C]: gradient w.r.t. conv3's pre-activation (pool backward AND conv3's ReLU mask applied),
 *   dx_skip [B,H,W,Cin_p] (may be NULL): gradient reaching the block input through the skip path,
 *   d_bn_weight, d_bn_bias, d_w1x1 [C][Cin], d_b1x1 (fp32, overwritten). */
int bx_block_tail_bwd(const bxTailDesc* d, const void* dout, const void* y3, const void* x, const void* pooled,
                      const float* w1x1, int Cin, const float* bn_weight, const float* save_mean,
                      const float* save_invstd, const uint64_t* seed, void* dz3, void* dx_skip,
                      float* d_bn_weight, float* d_bn_bias, float* d_w1x1, float* d_b1x1,
                      void* workspace, size_t workspace_bytes, bxStream stream);
/* y = max(x,0) elementwise (only used when a pre-ReLU conv output is an attribution target). */
int bx_relu(const void* x, void* y, size_t n, int dtype, bxStream stream);

/* ---- heads ------------------------------------------------------------------------------------ */
/* AdaptiveAvgPool2d(1) -> Linear(C,N) -> LogSoftmax (M:92-94,103-106).  feat [B,HW,C] dtype;
 * gap_out fp32 [B,C] (saved), logp fp32 [B,N].  N <= 32. */
int bx_gap_fc_lsm_fwd(const void* feat, const float* w, const float* b, float* gap_out, float* logp,
                      int B, int HW, int C, int N, int dtype, bxStream stream);
/* dlogp [B,N] -> dfeat [B,HW,C] (dtype), dW [N,C], db [N].  dfeat/dW/db may be NULL individually. */
int bx_gap_fc_lsm_bwd(const float* dlogp, const float* logp, const float* gap_out, const float* w,
                      void* dfeat, float* dw, float* db, int B, int HW, int C, int N, int dtype, bxStream stream);
/* Linear(K,N) -> LogSoftmax on fp32 features [B,K] (EEGNet dense, M:263-269,286-288). */
int bx_linear_lsm_fwd(const float* x, const float* w, const float* b, float* logp, int B, int K, int N,
                      bxStream stream);
int bx_linear_lsm_bwd(const float* dlogp, const float* logp, const float* x, const float* w, float* dx,
                      float* dw, float* db, int B, int K, int N, bxStream stream);
/* Fusion head: cat(eeg_logp, spec_logp) -> Linear(2N,Hd) -> ReLU -> Linear(Hd,N) -> LogSoftmax
 * (NB:1095-1105).  hidden fp32 [B,Hd] is saved for backward. */
int bx_fusion_head_fwd(const float* eeg_logp, const float* spec_logp, const float* w1, const float* b1,
                       const float* w2, const float* b2, float* hidden, float* logp, int B, int N, int Hd,
                       bxStream stream);
int bx_fusion_head_bwd(const float* dlogp, const float* logp, const float* hidden, const float* eeg_logp,
                       const float* spec_logp, const float* w1, const float* w2, float* d_eeg_logp,
                       float* d_spec_logp, float* dw1, float* db1, float* dw2, float* db2, int B, int N, int Hd,
                       bxStream stream);
/* Fused multimodal head: the three ops above in one launch forward / two backward, for MultimodalModel.forward
 * (NB:1095-1105 over models.py:103-106 and :286-288).  feat: block5 output [B,HW,C] channels-last `dtype`; eeg_feat fp32
 * [B,K] (EEGNet features).  Outputs of fwd (all fp32, all read again by bwd): gap_out [B,C], spec_logp / eeg_logp [B,N]
 * (the two branch outputs), hidden [B,Hd], logp [B,N].  bwd: dfeat [B,HW,C] `dtype` and d_eeg_feat [B,K] may be NULL;
 * parameter gradients are written when non-NULL.  Limits: C <= 1024, K <= 4096, N <= 32, Hd <= 256, Hd*2N <= 4096. */
size_t bx_mm_head_workspace(int B, int N, int Hd);
int bx_mm_head_fwd(const void* feat, const float* eeg_feat, const float* fc_w, const float* fc_b, const float* dense_w,
                   const float* dense_b, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* gap_out, float* spec_logp, float* eeg_logp, float* hidden, float* logp, int B, int HW, int C,
                   int K, int N, int Hd, int dtype, bxStream stream);
int bx_mm_head_bwd(const float* dlogp, const float* logp, const float* hidden, const float* spec_logp,
                   const float* eeg_logp, const float* gap, const float* eeg_feat, const float* fc_w,
                   const float* dense_w, const float* w1, const float* w2, void* dfeat, float* d_eeg_feat,
                   float* d_fc_w, float* d_fc_b, float* d_dense_w, float* d_dense_b, float* dw1, float* db1, float* dw2,
                   float* db2, void* workspace, size_t workspace_bytes, int B, int HW, int C, int K, int N, int Hd,
                   int dtype, bxStream stream);
/* nn.KLDivLoss on log-prob input / prob target (NB:1989,1599): loss[1] and dlogp = -t/denom.
 * reduction: 0 'mean' (denom B*N), 1 'batchmean' (denom B), 2 'sum'.  grad_scale multiplies dlogp. */
int bx_kldiv_fwd_bwd(const float* logp, const float* target, float* loss, float* dlogp, int B, int N,
                     int reduction, float grad_scale, bxStream stream);

/* ---- EEGNet branch (M:239-289); activations fp32 or bf16 NCHW ------------------------------------ */
typedef struct {
  int B, Chans, T;      /* input [B,1,Chans,T] fp32                                  */
  int F1, D, F2;        /* 8, 2, 16 (the reference's defaults: register-tiled kernels; any other values: the general kernel set) */
  int K1, K2;           /* temporal kernel lengths 64 and 16 ('same': left pad (K-1)/2) */
  int P1, P2;           /* average-pool widths 4 and 8                               */
  int training;
  float eps, momentum, dropout_p;
  uint32_t salt;        /* bit 31 set: one mask entry per (sample, channel) = nn.Dropout2d (models.py:255) */
  int dtype;            /* storage of the big intermediates (conv1 output)           */
  int collapse;         /* 1: training passes with bf16 storage, K1 = 64, T >= 96, T % 8 == 0 never form the conv1 output
                         * [B,F1,Chans,T]: the electrodes are mixed first, BatchNorm1's statistics come from the input's
                         * autocorrelation, and the gradients of conv1 / bn1 / depthwiseConv from one correlation of dL/du
                         * with the input.  Such a pass has NO input gradient (dx must be NULL);
                         * forward and backward must see the same flag.  0: the layer-by-layer path. */
  float dropout_p2;     /* rate of the SECOND dropout (after pool 2); < 0: the same as dropout_p (EEGNet shares one module,
                         * models.py:255; EEGNetAttentionDeep has dropout1 / dropout2, models.py:152-164) */
  const float* const* x_slot;  /* NULL, or a DEVICE word holding the input's address (bx_store_u64x2): bx_eeg_features_fwd then reads x
                         * through it -- a captured graph is replayed on the caller's batch without a copy.  Collapsed evaluation-mode
                         * path only (collapse = 1, training = 0), where one kernel reads x and nothing is kept for a backward;
                         * the `x` argument must still be a valid pointer of the same shape (it is what an eager run reads). */
} bxEegDesc;
/* Parameter block: pointers to the fp32 tensors of the module, reference names in comments. */
typedef struct {
  const float* conv1_w;      /* conv1.weight        [F1,1,1,K1]      */
  const float* bn1_w; const float* bn1_b; float* bn1_rm; float* bn1_rv; int64_t* bn1_nbt;
  const float* dw_w;         /* depthwiseConv.weight [F1*D,1,Chans,1] */
  const float* bn2_w; const float* bn2_b; float* bn2_rm; float* bn2_rv; int64_t* bn2_nbt;
  const float* sep_w;        /* separableConv.weight [F2,F1*D,1,K2]  */
  const float* bn3_w; const float* bn3_b; float* bn3_rm; float* bn3_rv; int64_t* bn3_nbt;
} bxEegParams;
typedef struct {
  float* conv1_w; float* bn1_w; float* bn1_b; float* dw_w; float* bn2_w; float* bn2_b;
  float* sep_w; float* bn3_w; float* bn3_b;
} bxEegGrads;
/* Saved-for-backward arena layout is private; query its size, pass the same buffer to bwd. */
size_t bx_eeg_saved_bytes(const bxEegDesc* d);
size_t bx_eeg_workspace(const bxEegDesc* d);
/* x fp32 [B,1,Chans,T] -> feat fp32 [B, F2*(T/P1/P2)] (the Flatten() output fed to `dense`). */
int bx_eeg_features_fwd(const bxEegDesc* d, const bxEegParams* p, const float* x, const uint64_t* seed,
                        float* feat, void* saved, void* workspace, size_t workspace_bytes, bxStream stream);
/* dfeat [B, F2*T2] -> parameter gradients and (optional, may be NULL) dx fp32 [B,1,Chans,T]. */
int bx_eeg_features_bwd(const bxEegDesc* d, const bxEegParams* p, const float* x, const float* dfeat,
                        const uint64_t* seed, const void* saved, const bxEegGrads* g, float* dx,
                        void* workspace, size_t workspace_bytes, bxStream stream);

/* ---- EEGNetAttentionDeep head (M:136-235, Attention M:109-134): everything after EEGNet's block 2 ----------
 * feat fp32 [B, F2*T2] (bx_eeg_features_fwd's output for the same input; the class's dropout2 is applied there)
 *   -> conv2 (1x16 'same', F2 -> F3, no bias) -> batchnorm4 -> ELU -> avg_pool3 (1x8) -> dropout3
 *   -> attention_layer over the L = T2/8 time steps (query/key/value = Linear(F3,F3), scale F3^-0.5)
 *   -> flatten (channel-major) -> dense1 (F3*L -> Hd) -> dense2 (Hd -> N) -> LogSoftmax.  All tensors fp32. */
typedef struct {
  int B, T2;            /* feat is [B, F2, T2]                                          */
  int F2, F3, K3, P3;   /* 16, 32, 16, 8                                                */
  int Hd, N;            /* dense1 width (128; power of two in [32,256]), classes (<=16) */
  int training;
  float eps, momentum, dropout_p;
  uint32_t salt;        /* bit 31 set: one mask entry per (sample, channel) = nn.Dropout2d (models.py:255) */
} bxEegDeepDesc;
typedef struct {
  const float* conv2_w;      /* conv2.weight [F3,F2,1,K3] */
  const float* bn4_w; const float* bn4_b; float* bn4_rm; float* bn4_rv; int64_t* bn4_nbt;
  const float* wq; const float* bq;   /* attention_layer.query.weight [F3,F3] / .bias */
  const float* wk; const float* bk;   /* attention_layer.key                          */
  const float* wv; const float* bv;   /* attention_layer.value                        */
  const float* w1; const float* b1;   /* dense1.weight [Hd, F3*L] (16-byte aligned) / .bias */
  const float* w2; const float* b2;   /* dense2.weight [N, Hd] / .bias                */
} bxEegDeepParams;
typedef struct {
  float* conv2_w; float* bn4_w; float* bn4_b; float* wq; float* bq; float* wk; float* bk; float* wv; float* bv;
  float* w1; float* b1; float* w2; float* b2;
} bxEegDeepGrads;
size_t bx_eeg_deep_saved_bytes(const bxEegDeepDesc* d);   /* 0 = unsupported geometry */
size_t bx_eeg_deep_workspace(const bxEegDeepDesc* d);
/* logp fp32 [B,N]; attn fp32 [B,L,L] = the softmax weights (the module's second return value), also read by bwd. */
int bx_eeg_deep_fwd(const bxEegDeepDesc* d, const bxEegDeepParams* p, const float* feat, const uint64_t* seed,
                    float* logp, float* attn, void* saved, void* workspace, size_t workspace_bytes, bxStream stream);
/* dlogp [B,N] -> dfeat [B,F2*T2] (may be NULL) and the parameter gradients (g may be NULL: inputs only). */
int bx_eeg_deep_bwd(const bxEegDeepDesc* d, const bxEegDeepParams* p, const float* feat, const float* dlogp,
                    const float* attn, const uint64_t* seed, const void* saved, const bxEegDeepGrads* g, float* dfeat,
                    void* workspace, size_t workspace_bytes, bxStream stream);

/* Stand-alone Attention module (M:109-134): x fp32 [B,L,D] -> out [B,L,D], attn [B,L,L] (softmax weights); D = 32, L <= 32.
 * qkv_saved fp32 [B,3,L,D] is written by fwd and read by bwd.  bwd: dattn (gradient w.r.t. the returned weights) and dx may
 * be NULL; parameter gradients are written when non-NULL (then workspace >= bx_attention_workspace(B) bytes). */
size_t bx_attention_workspace(int B);
int bx_attention_fwd(const float* x, const float* wq, const float* bq, const float* wk, const float* bk, const float* wv,
                     const float* bv, float* out, float* attn, float* qkv_saved, int B, int L, int D, bxStream stream);
int bx_attention_bwd(const float* dout, const float* dattn, const float* x, const float* attn, const float* qkv_saved,
                     const float* wq, const float* wk, const float* wv, float* dx, float* dwq, float* dbq, float* dwk,
                     float* dbk, float* dwv, float* dbv, void* workspace, size_t workspace_bytes, int B, int L, int D,
                     bxStream stream);

/* ---- attribution ----------------------------------------------------------------------------- */
/* Grad-CAM channel reduce (canonical; the reference has none -- SURVEY.md K18):
 *   w[m,c] = mean_p G[m,p,c];  raw[m,p] = sum_c w[m,c]*A[m/maps_per_act,p,c];  cam = relu ? max(raw,0) : raw.
 * G: [n_maps,HW,C], A: [n_maps/maps_per_act,HW,C], `dtype` channels-last (maps_per_act = classes per
 * sample sharing one activation).  cam fp32 [n_maps,HW]; weights_out fp32 [n_maps,C] may be NULL. */
int bx_gradcam_reduce(const void* A, const void* G, float* cam, float* weights_out, int n_maps, int maps_per_act,
                      int HW, int C, int relu, int dtype, bxStream stream);
/* Grad-CAM at the last stage of the multimodal model in one launch (canonical definition; the reference ships none, SURVEY fact
 * 3; heads = models.py:103-106 and XAI_Multimodality.py:1095-1105).  A: stage output NHWC [B,HW,C] (dtype); eeg_logp fp32 [B,N]
 * (EEG branch's log-probs); fc_* = Spectrogram_Model.fc, w1/b1 = fc1 [Hd,2N], w2/b2 = fc2 [N,Hd].  class_mode -2: every class
 * (nm = N maps per sample), -1: each sample's arg-max class, >= 0: that class (nm = 1).  Outputs: out_logp fp32 [B,N] (nullable),
 * cam fp32 [B*nm,HW] (ReLU'd iff relu), raw fp32 [B*nm,HW] (pre-ReLU, nullable), weights_out fp32 [B*nm,C] (nullable). */
int bx_gradcam_head(const void* A, const float* eeg_logp, const float* fc_w, const float* fc_b, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* out_logp, float* cam, float* raw, float* weights_out, int B, int HW,
                    int C, int N, int Hd, int class_mode, int relu, int dtype, bxStream stream);
/* The same for sweeps over many batches: the EEG branch's own head (EEGNet.dense + LogSoftmax, models.py:287-289: eeg_feat fp32
 * [B,Fe], dense_w [N,Fe], dense_b [N]) and the bilinear up-sampling of the finished maps to H x W (W % 4 == 0) run in the same
 * launch; A is [B,h*w,C].  maps fp32 [B*nm,H,W]. */
int bx_gradcam_head_sweep(const void* A, const float* eeg_feat, const float* dense_w, const float* dense_b, int Fe, const float* fc_w,
                          const float* fc_b, const float* w1, const float* b1, const float* w2, const float* b2, float* out_logp,
                          float* maps, int B, int h, int w, int C, int N, int Hd, int H, int W, int class_mode, int relu, int dtype,
                          bxStream stream);
/* Bilinear resize (align_corners=False) of fp32 maps [N,h,w] -> [N,H,W]  (F.interpolate). */
int bx_resize_bilinear(const float* src, float* dst, int N, int h, int w, int H, int W, bxStream stream);
/* Saliency reduce (NB:3121-3129): out[b,p] = scale * max_c |g[b,p,c]|, g NHWC `dtype` (first C of Cs). */
int bx_saliency_reduce(const void* g, float* out, int B, int HW, int C, int Cs, float scale, int dtype,
                       bxStream stream);
/* y = alpha*x + beta*y over fp32 (integrated-gradients accumulate, baseline interpolation). */
int bx_axpby(const float* x, float* y, size_t n, float alpha, float beta, bxStream stream);
int bx_mul(const float* a, const float* b, float* out, size_t n, bxStream stream);
/* Integrated gradients (Captum semantics): all K interpolants of a pass, out[k][i] = (1-a_k)*base[i] + a_k*x[i],
 * and the weighted accumulation acc[i] += sum_k w_k * grads[k][i] (k ascending); alphas / weights are device float[K]. */
int bx_ig_interpolate(const float* x, const float* base, const float* alphas_device, float* out, size_t n, int K, bxStream stream);
int bx_ig_accumulate(const float* grads, const float* weights_device, float* acc, size_t n, int K, bxStream stream);
/* out = |x| over fp32 (EEG saliency, NB:3121-3122). */
int bx_abs(const float* x, float* out, size_t n, bxStream stream);
/* out = x * scalar[0] with the scalar on the device (loss.backward()'s upstream gradient; no host sync). */
int bx_scale_dev(const float* x, const float* scalar, float* out, size_t n, bxStream stream);

/* ---- EEG stacker (DS:73-104,125-131): select/clip/nan->0,/32 -> Butterworth IIR -> decimate ---- */
/* raw fp32 [B,L,Craw]; channel_index int32[C] (device) selects columns; b,a = host double[order+1];
 * out fp32 [B,1,C,L/step].  fp64 state per (sample, channel) row. */
int bx_eeg_stack_iir(const float* raw, const int* channel_index, float* out, int B, int L, int Craw, int C,
                     const double* b_host, const double* a_host, int order, int step, float clip, float scale,
                     bxStream stream);

/* ---- native-pipeline EEG montage stacker (CombinedDataset.process_eeg, NB:1148-1164,1211-1276) ---- */
/* raw fp32 [B,L,Craw] (eeg.values of each frame); output row r is built from raw column row_a[r], minus column row_b[r]
 * when row_b[r] >= 0 (device int32[R] each, values < Craw).  Per raw column: lfilter(b1,a1) in fp64, NaN -> the row's
 * nanmean; per output row: (difference,) lfilter(b2,a2), mean of 4 consecutive samples at every 4th column of [0,L-1),
 * z-score with population std and eps, zero-padded / truncated to out_len.  out fp32 [B,R,out_len].
 * b*, a* = host double[order+1] (transfer-function coefficients, order <= 12).  L % 4 must be 0 or 1.
 * status: device int32[1], set to 1 when some raw row is NaN from its first sample (the reference drops such rows and
 * then mis-indexes; here the row is treated as all-zero after the first filter and the caller is told).
 * The mirror augmentation (cfg.AUGMENT) is a column permutation the caller applies through row_a/row_b. */
size_t bx_eeg_montage_workspace(int B, int L, int Craw, int R);
int bx_eeg_montage_stack(const float* raw, const int* row_a, const int* row_b, float* out, int B, int L, int Craw, int R,
                         int out_len, const double* b1, const double* a1, int order1, const double* b2, const double* a2,
                         int order2, float eps, int* status, void* workspace, size_t workspace_bytes, bxStream stream);

/* ---- native-pipeline spectrogram pre-processing (CombinedDataset.process_spectrogram, NB:1166-1204) ---- */
/* raw fp32 [B,Trows,Ccols] (the parquet frame's values without the time column, NaNs allowed); offsets: device int32[B]
 * (spectrogram_label_offset_seconds; the reference windows `sel` COLUMNS from offset//2) or NULL; out fp32 [B,3,R,W].
 * Chain: transpose -> pad/truncate to R x W -> NaN -> row nanmean -> subtract column means -> filtfilt(notch_b, notch_a)
 * along the rows (odd extension, 9 samples; notch_zi = scipy.signal.lfilter_zi) -> separable gaussian (gauss_w: 9 symmetric
 * weights, 'reflect' boundary) -> min-max with eps -> 3 identical channels.  The reference's final skimage resize is to
 * the array's own shape (identity).  status: device int32[1]; bit 0 = some row was entirely NaN (the reference drops it
 * and then really resamples: not reproduced), bit 1 = NaN after filtering. */
size_t bx_spec_preprocess_workspace(int B, int R, int W);
int bx_spec_preprocess(const float* raw, const int* offsets, float* out, int B, int Trows, int Ccols, int R, int W, int sel,
                       const double* notch_b, const double* notch_a, const double* notch_zi, const double* gauss_w,
                       float eps, int* status, void* workspace, size_t workspace_bytes, bxStream stream);

/* ---- benchmark-variant spectrogram stacker (spectrogram half): four region planes per sample ----------- */
/* raw fp32 [B,Trows,C] parquet values (C = regions x bins, NaNs allowed); offsets device int32[B] or NULL (window of `win` time
 * rows from offset // 2, zero padded; XAI_Multimodality.py:1178-1183); out fp32 [B,regions,Ho,Wo].  Chain: transpose ->
 * normalize_signal (root/src/utils/data_utils.py:133-136: NaN -> nanmean of the sample, min-max with eps) -> per region
 * resample_spectrogram (data_utils.py:145-147: skimage.transform.resize(mode='reflect', anti_aliasing=True): gaussian_filter
 * with the host-supplied 1-D weights gauss_y[2*radius_y+1] / gauss_x[2*radius_x+1] ('mirror' boundary; radius 0 = no filter),
 * then order-1 zoom with grid_mode coordinates, mirrored indices).  fp64 arithmetic. */
size_t bx_spec_regions_workspace(int B);
int bx_spec_regions(const float* raw, const int* offsets, float* out, int B, int Trows, int C, int regions, int win, int Ho, int Wo,
                    const double* gauss_y, int radius_y, const double* gauss_x, int radius_x, float eps, void* workspace,
                    size_t workspace_bytes, bxStream stream);

/* ---- optimiser over the flat parameter arena (torch.optim.AdamW, NB:1988) ------------------------ */
/* p, g, m, v fp32 [n]; step_count device float[1]: incremented by this call, then used as t. */
int bx_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float grad_scale, float* step_count, bxStream stream);
/* The same step with the hyper-parameters in DEVICE memory, hyper fp32[8] = {lr, beta1, beta2, eps, weight_decay, grad_scale,
 * l2_lambda, unused} (a captured hipGraph follows a learning-rate schedule without re-capture), and with the DDP loop's manual
 * L2 penalty fused in (root/src/training/training_distributed.py:52-57: total_loss = loss + l2_lambda * sum p^2): 2*l2_lambda*p
 * is added to the scaled gradient before the moment updates; when sumsq_partials (fp32[bx_adamw_partials(n)]) and l2_value
 * (fp32[1]) are given, l2_value[0] = l2_lambda * sum p^2 over the parameters BEFORE this update (fixed-order sum).
 * step_count here is a device buffer of bx_adamw_step_words(n) 32-bit words: [0] the step count (float, incremented by this
 * call in the update launch itself and then used as t), the rest ticket counters the launch uses to find its last workgroup --
 * zero on entry, zero again on completion; concurrent launches must not share the buffer. */
size_t bx_adamw_partials(size_t n);
size_t bx_adamw_step_words(size_t n);
int bx_adamw_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper, float* step_count,
                      float* sumsq_partials, float* l2_value, bxStream stream);
/* dst_device[0..7] = values_host[0..7], ordered on `stream`.  The eight values travel as kernel arguments (copied when the call
 * returns), so the host buffer needs no lifetime beyond the call -- how FlatAdamW refreshes `hyper` when a torch LR scheduler
 * changed param_groups (DDP:98-101), also between replays of a captured step. */
int bx_store_f32x8(float* dst_device, const float* values_host, bxStream stream);
/* dst_device[0] = a, dst_device[1] = b (64-bit words, e.g. device addresses for bxEegDesc.x_slot / bx_conv3x3_pack_layout_ex's src_slot);
 * the values travel as kernel arguments, the store is ordered on `stream`. */
int bx_store_u64x2(uint64_t* dst_device, uint64_t a, uint64_t b, bxStream stream);
/* sum of squares of a flat fp32 arena -> out[1] (DDP loop's manual L2 term, DDP:52-53). */
int bx_sumsq(const float* x, size_t n, float* out, bxStream stream);
/* LIME's batched inference (XAI_Multimodality.py:1567-1574): uint8 images [N,H,W,C] -> scale * value in the internal
 * channels-last layout [N,H,W,Cp] (dtype; channels C..Cp-1 zero) = torchvision ToTensor with scale 1/255; row-wise softmax. */
int bx_u8_to_nhwc(const unsigned char* src, void* dst, int N, int H, int W, int C, int Cp, float scale, int dtype, bxStream stream);
int bx_softmax_rows(const float* x, float* y, int rows, int N, bxStream stream);
/* attribution seeds: seed fp32 [rows,N], row r = onehot(class of sample r % B); class_mode >= 0: that class, -1: arg-max of
 * logp fp32 [B,N] (first maximum).  Replaces the reference's output[0, argmax] indexing (XAI_Multimodality.py:3110-3111). */
int bx_class_seed(const float* logp, float* seed, int rows, int B, int N, int class_mode, bxStream stream);
/* dropout seed stream: out[0] = ++state[0] (a forward call and its backward read the same `out`). */
int bx_seed_next(uint64_t* state, uint64_t* out, bxStream stream);
/* two independent counters advanced by one launch (the two branches of the multimodal model draw one seed each per step) */
int bx_seed_next2(uint64_t* state_a, uint64_t* out_a, uint64_t* state_b, uint64_t* out_b, bxStream stream);

#ifdef __cplusplus
}
#endif
#endif /* BRAINXAI_H */
