/* suhpe.h — C ABI of libsuhpe.so: the MI355X (gfx950) hot path of SemiUHPE's teacher-student training step.
 *
 * The reference (hnuzhy/SemiUHPE) has no FFI of its own: its hot path is Python calling torch ops
 * (SURVEY.md §8b).  This header is the boundary a maintainer binds instead (ctypes stub in INTEGRATION.md);
 * each entry point cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless the name ends in _host
 *  - every launch is asynchronous on `stream` (a hipStream_t passed as void*); no allocation, no sync inside
 *  - return 0 on success, negative SUHPE_E* otherwise; suhpe_last_error() gives the message (thread-local)
 *  - activations are NHWC fp32 ("pixel-major": [n][h][w][c]); conv weights are [cout][r][s][cin_per_group]
 *    (what torch stores for a channels_last (cout,cin,r,s) tensor); matrices are row-major
 */
#ifndef SUHPE_H
#define SUHPE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUHPE_OK 0
#define SUHPE_EINVAL (-1)   /* bad argument (shape, alignment, null)            -> ValueError in the binding */
#define SUHPE_ELAUNCH (-2)  /* HIP launch failure                               -> RuntimeError              */
#define SUHPE_EWORKSPACE (-3) /* workspace too small (see *_workspace_bytes)    -> RuntimeError              */

typedef void* suhpe_stream_t;

const char* suhpe_last_error(void);
int suhpe_abi_version(void);

/* activation codes used by the fused normalise+activate kernels */
enum { SUHPE_ACT_NONE = 0, SUHPE_ACT_RELU = 1, SUHPE_ACT_RELU6 = 2, SUHPE_ACT_SILU = 3, SUHPE_ACT_GELU = 4,
       SUHPE_ACT_SIGMOID = 5 };

/* ------------------------------------------------------------------ matrix-Fisher rotation head ---------- */
/* src/fisher/fisher_utils.py:14-36 (vmf_loss / KL_Fisher) + :39-48.  A,R: (n,9).  Writes loss (n), the
 * gradient dloss_i/dA_i (n,9) and the mode rotation Rest (n,9).  Any output pointer may be NULL. */
int suhpe_fisher_nll(const float* A, const float* R, int n, float overreg, float* loss, float* dA, float* Rest,
                     suhpe_stream_t stream);
/* src/fisher/fisher_utils.py:39-48 batch_torch_A_to_R */
int suhpe_fisher_svd_project(const float* A, int n, float* R, suhpe_stream_t stream);
/* src/fisher/fisher_utils.py:70-81 fisher_entropy */
int suhpe_fisher_entropy(const float* A, int n, float* H, suhpe_stream_t stream);
/* src/fisher/fisher_utils.py:84-99 fisher_CE (teacher A1, student A2) and dCE_i/dA2_i.
 * mask (n floats, 0/1) may be NULL; where mask==0 the outputs are exactly 0 (replaces the boolean
 * compaction of src/agent.py:155). */
int suhpe_fisher_ce(const float* A1, const float* A2, const float* mask, int n, float* ce, float* dA2,
                    suhpe_stream_t stream);
/* src/fisher/fisher_utils.py:51-67 fisher_log_pdf */
int suhpe_fisher_log_pdf(const float* A, const float* R, int n, float* logp, suhpe_stream_t stream);
/* src/laplace/rotation_laplace.py:24-34 NLL_loss("RLaplace", ...): discrete normalisation over `grids` (ngrid,9) SO(3)
 * rotations; writes loss (n), dloss_i/dA_i (n,9) and the analytical mode (n,9).  Outputs may be NULL. */
int suhpe_laplace_nll(const float* A, const float* R, const float* grids, int n, int ngrid, float* loss, float* dA,
                      float* mode, suhpe_stream_t stream);
/* src/agent.py:110-122 rotation-consistency adjust.  mode 0: aug*A (DAD3DHeads); mode 1: 300WLP form. */
int suhpe_rot_adjust(const float* A, const float* aug, int n, int mode, float* out, suhpe_stream_t stream);
/* src/utils.py:232-260 -> radians [pitch,yaw,roll] */
int suhpe_euler_from_rot(const float* R, int n, int full_range, float* euler, suhpe_stream_t stream);
/* src/agent.py:447-455: gt_euler_deg NULL -> geodesic degrees; else mean |euler(pred) - gt| degrees */
int suhpe_pose_err_deg(const float* pred, const float* gt, const float* gt_euler_deg, int n, float* err,
                       suhpe_stream_t stream);
/* src/agent.py:148-150: mask_i = H_i < *thres (thres read from device memory so that the dynamic threshold
 * never needs a host round trip); count (1 float) receives sum(mask). */
int suhpe_entropy_mask(const float* H, const float* thres, int n, float* mask, float* count,
                       suhpe_stream_t stream);

/* ------------------------------------------------------------------ tensor augmentations ----------------- */
/* src/augments.py:12-35 + :67-94 in one out-of-place pass, fused with the NCHW->NHWC layout change at the API
 * edge.  src/dst: (n,c,h,w) or (n,h,w,c) element order selected by src_nhwc/dst_nhwc; src != dst.
 * boxes: int32 (n,holes,4) = x0,y0,x1,y1 with the reference's axis use (y slices dim h, x slices dim w).
 * cutout_boxes and/or cutmix_boxes may be NULL.  perm: int32 (n), source image of the CutMix paste
 * (reference: image[rand_index], taken after CutOut). */
int suhpe_strong_aug(const float* src, float* dst, int n, int c, int h, int w, int src_nhwc, int dst_nhwc,
                     const int32_t* cutout_boxes, const int32_t* cutmix_boxes, const int32_t* perm, int holes,
                     suhpe_stream_t stream);
/* plain layout change: (n,c,h,w) contiguous -> (n,h,w,c) */
int suhpe_nchw_to_nhwc(const float* src, int n, int c, int h, int w, float* dst, suhpe_stream_t stream);

/* ------------------------------------------------------------------ convolution / linear ----------------- */
typedef struct {
    int n, h, w, cin;        /* input  (n,h,w,cin)  */
    int ho, wo, cout;        /* output (n,ho,wo,cout) */
    int r, s;                /* filter */
    int stride, pad;
    int groups;              /* 1, or cin==cout==groups for depthwise, or 2 (RepVGG g2) */
} suhpe_conv_t;

/* nn.Conv2d / nn.Linear forward (torch ops dispatched by src/networks.py, src/repvgg.py:28-32,
 * src/tiny_vit.py:28-50).  w: [cout][r][s][cin/groups]; bias NULL or [cout]. */
int suhpe_conv2d_fwd(const suhpe_conv_t* d, const float* x, const float* w, const float* bias, float* y,
                     suhpe_stream_t stream);
/* forward with a folded per-channel affine map + activation (+ residual) in the epilogue: eval-mode conv -> BatchNorm ->
 * act [-> + skip] in one pass over the output (torchvision Conv2dNormActivation / MBConv / RepVGG blocks in eval()).
 * y = act(conv(x,w)*scale[c] + shift[c] (+ residual)) (+ residual); residual (same shape as y) may be NULL;
 * residual_after_act selects which.  scale / shift come from suhpe_bn_eval_affine. */
int suhpe_conv2d_fwd_affine(const suhpe_conv_t* d, const float* x, const float* w, const float* scale, const float* shift,
                            int act, const float* residual, int residual_after_act, float* y, suhpe_stream_t stream);
/* forward without bias that also emits BatchNorm statistics partials from its epilogue (saves the separate read
 * pass of suhpe_bn_stats): stats[(tile*cout + channel)*4 + {0..3}] = (S, Q, k, n): sums of (y-k) and (y-k)^2 over the tile's n rows,
 * shifted by k (the tile's first row); tiles = suhpe_conv2d_fwd_stats_tiles(d), so stats holds cout*tiles*4 floats.  Finish with suhpe_bn_stats_from_partials. */
int suhpe_conv2d_fwd_stats_tiles(const suhpe_conv_t* d);
int suhpe_conv2d_fwd_stats(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats,
                           suhpe_stream_t stream);
/* Batch segments: when one launch carries two sub-batches that BatchNorm must normalise separately (the labeled and the
 * strong-view batch of a training step run as ONE student pass), rows [0, seg_row) are the first segment.  The row tile that
 * contains the boundary writes its first-segment rows to stats[tile] and the remaining rows to stats_extra[channel]
 * (cout records); suhpe_conv2d_fwd_stats_tile_rows(d) is the tile height, so the first segment owns records
 * [0, ceil(seg_row / tile_rows)) and the second stats_extra (if seg_row is not a multiple of the tile height) plus the
 * records from ceil(seg_row / tile_rows) on: finish each with suhpe_bn_stats_from_partials_range. */
int suhpe_conv2d_fwd_stats_tile_rows(const suhpe_conv_t* d);
int suhpe_conv2d_fwd_stats_seg(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, float* stats_extra,
                               int64_t seg_row, suhpe_stream_t stream);
/* Tile menu of the GEMM kernels.  pass: 0 forward, 1 input gradient, 2 filter gradient.  A caller that has timed the
 * menu on a shape pins the winner with suhpe_conv2d_set_tile (tile = -1 unpins: back to the built-in heuristic); the
 * pin is process-wide and keyed by the whole descriptor.  Results of pass 0 / 1 do not depend on the tile among the direct
 * kernels (the Winograd tile below computes the same sums through a different factorisation: equal to fp32 rounding, not
 * bit-equal); pass 2 changes the split-K partition (summation order) with it. */
int suhpe_conv2d_tile_count(int pass);
/* start-up stagger of the GEMM workgroups (0 = off): see stagger_wait in csrc/conv_igemm.hip */
int suhpe_conv2d_set_stagger(int quanta, int first_gen);
const char* suhpe_conv2d_tile_name(int pass, int tile);
int suhpe_conv2d_set_tile(const suhpe_conv_t* d, int pass, int tile);
/* Winograd F(2x2, 3x3) tile (csrc/conv_wino.hip; menu name "wino_nt_kernel<64,64>"): 3x3 / stride 1 / pad 1 forward and input-gradient
 * convolutions on even output sizes with 2.25x fewer multiplies -- what cuDNN does behind the reference's 3x3 convs (src/networks.py:303
 * FusedMBConv, src/repvgg.py:28-32 rbr_dense).  The tile runs fastest on a filter that is already in the Winograd domain:
 *   suhpe_conv2d_wino_pinned(d, pass)        1 when the tile selected for (d, pass) is the Winograd one (and this geometry can run it)
 *   suhpe_conv2d_wino_filter_bytes(d, pass)  size of the transformed filter
 *   suhpe_conv2d_wino_filter(d, pass, w, fw) fw <- G g G^T of the filter AS STORED (w, [cout][3][3][cin/groups]) for pass 0 (forward) or of
 *                                            its transposed / 180-degree rotated form for pass 1 (input gradient: no transposed copy needed),
 *                                            in the kernel's LDS image order; redo after every weight update
 *   suhpe_conv2d_fwd_wino                    suhpe_conv2d_fwd / _fwd_stats / _fwd_stats_seg in one: bias may be NULL; stats NULL = no records;
 *                                            stats_extra / seg_row as suhpe_conv2d_fwd_stats_seg.  fw NULL = the launch falls back to a DIRECT tile
 *                                            on w (direct-kernel rounding, no Winograd arithmetic: the kernel never transforms filters itself);
 *                                            on a (d, pass) whose tile is not the Winograd one, fw is ignored and w is used
 *   suhpe_conv2d_dgrad_wino                  suhpe_conv2d_dgrad_add with the same rule; wt (the transposed filter) may be NULL ONLY when fw is given
 *                                            and suhpe_conv2d_wino_pinned(d, 1) holds -- any other call with wt NULL returns SUHPE_EINVAL
 * Eval-mode passes on database-pinned shapes run the AFFINE instantiations of the same kernel (suhpe_conv2d_fwd_affine_wino), i.e. on
 * Winograd rounding as well.
 * Results differ from the direct tiles by rounding (measured: closer to float64 than the direct sum).  SUHPE_WINOGRAD=0 disables the tile. */
int suhpe_conv2d_wino_pinned(const suhpe_conv_t* d, int pass);
int64_t suhpe_conv2d_wino_filter_bytes(const suhpe_conv_t* d, int pass);
int suhpe_conv2d_wino_filter(const suhpe_conv_t* d, int pass, const float* w, float* fw, suhpe_stream_t stream);
int suhpe_conv2d_fwd_wino(const suhpe_conv_t* d, const float* x, const float* w, const float* fw, const float* bias, float* y, float* stats,
                          float* stats_extra, int64_t seg_row, suhpe_stream_t stream);
/* suhpe_conv2d_fwd_affine (eval-mode conv -> BatchNorm -> activation [-> + skip], RepVGG's re-parameterised blocks) with the same rule:
 * fw from suhpe_conv2d_wino_filter(d, 0, ...) or NULL */
int suhpe_conv2d_fwd_affine_wino(const suhpe_conv_t* d, const float* x, const float* w, const float* fw, const float* scale, const float* shift, int act,
                                 const float* residual, int residual_after_act, float* y, suhpe_stream_t stream);
int suhpe_conv2d_dgrad_wino(const suhpe_conv_t* d, const float* dy, const float* wt, const float* fw, const float* skip, float* dx,
                            suhpe_stream_t stream);
/* Pointwise conv whose INPUT is act(BN(x)) [* per-sample scale] of a tensor x that was never normalised in memory: the loader of
 * the GEMM applies  a <- act(a * a_scale[s][k] + a_shift[s][k]) * a_rowscale[row / a_hw][k]  while it stages the operand
 * (s = batch segment of the row: rows < seg_row -> 0; a_rowscale (n, cin) may be NULL).  Used for torchvision's
 * FusedMBConv (3x3 expand -> BatchNorm -> SiLU -> [1x1 project]) and MBConv (depthwise -> BatchNorm -> SiLU -> SqueezeExcitation
 * scale -> [1x1 project]) blocks (reference src/networks.py:303): the activated / excited tensor is neither written nor read.
 * a_scale / a_shift [segments][cin] come from suhpe_bn_affine_from_stats.  The forward also emits the BatchNorm statistics records
 * of its OUTPUT like suhpe_conv2d_fwd_stats(_seg) (stats may be NULL); the filter gradient recomputes the same input from x. */
int suhpe_bn_affine_from_stats(const float* mean, const float* rstd, const float* gamma, const float* beta, int segments, int c,
                               float* scale, float* shift, suhpe_stream_t stream);
int suhpe_conv2d_fwd_stats_xf(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, float* stats_extra,
                              int64_t seg_row, const float* a_scale, const float* a_shift, const float* a_rowscale, int a_act, int a_hw,
                              suhpe_stream_t stream);
int suhpe_conv2d_wgrad_partial_xf(const suhpe_conv_t* d, const float* x, const float* dy, void* workspace, int64_t workspace_bytes,
                                  const float* a_scale, const float* a_shift, const float* a_rowscale, int a_act, int a_hw,
                                  int64_t seg_row, suhpe_stream_t stream);
/* Linear -> activation chains with the activation in the GEMM epilogues (TinyViT's Mlp: fc1 -> GELU -> fc2, reference
 * src/tiny_vit.py:205-213; pointwise convs / Linear layers only).  Forward: y_pre = x w^T + bias (kept for the backward) and
 * y_act = act(y_pre) from ONE launch -- no elementwise pass re-reads the pre-activation.  Backward: the input gradient of the layer
 * that CONSUMED y_act, multiplied by act'(pre) in its epilogue: dx = dgrad(dy) * act'(pre) -- the gradient of the activated tensor is
 * never stored.  Bit-identical to the unfused chain (same k order, same act / act' arithmetic). */
int suhpe_conv2d_fwd_act2(const suhpe_conv_t* d, const float* x, const float* w, const float* bias, int act, float* y_pre, float* y_act,
                          suhpe_stream_t stream);
/* Linear / 1x1 conv whose output closes a residual connection (TinyViT: x + drop_path(proj(...)), x + drop_path(fc2(...)), reference
 * src/tiny_vit.py:366,378; timm DropPath): y = (x w^T + bias) * rowscale[row / rows_per_sample] + skip from the GEMM epilogue -- no add pass over
 * the activations.  rowscale: one keep scale per sample (mask / keep_prob) or NULL (scale 1); skip has y's shape. */
int suhpe_conv2d_fwd_skip(const suhpe_conv_t* d, const float* x, const float* w, const float* bias, const float* skip, const float* rowscale,
                          int rows_per_sample, float* y, suhpe_stream_t stream);
int suhpe_conv2d_dgrad_actgrad(const suhpe_conv_t* d, const float* dy, const float* wt, const float* pre, int act, float* dx, suhpe_stream_t stream);
/* Input gradient of a pointwise conv whose INPUT was act(BatchNorm(bn_x)) of a train-mode BatchNorm -- torchvision FusedMBConv: 3x3 expand
 * conv -> BatchNorm -> SiLU -> 1x1 project conv (src/networks.py:303), where the forward applies that BatchNorm in the project conv's
 * loader (suhpe_conv2d_fwd_stats_xf).  ONE launch writes g = dgrad(dy) * act'(bn_x * scale + shift), the gradient of the BatchNorm's
 * OUTPUT, and reduces the BatchNorm backward's per-channel sums (sum g, sum g xhat) from the same epilogue (records per 128-row tile and
 * batch segment in `workspace`, finished by the BatchNorm finaliser: sums [c][2] or [2][c][2], dgamma / dbeta accumulated).  What remains
 * of that BatchNorm's backward is the apply pass over (bn_x, g) with activation NONE (suhpe_bn_act_bwd_apply / _seg): the separate
 * reduction pass over the expanded tensors (two reads of the widest activations of the network) disappears.
 * mean / rstd / scale / shift: [c] or, with seg_n > 0 (samples of the first batch segment), [2][c] as suhpe_bn_affine_from_stats writes them. */
int64_t suhpe_conv2d_dgrad_bnbwd_workspace_bytes(const suhpe_conv_t* d);
int suhpe_conv2d_dgrad_bnbwd(const suhpe_conv_t* d, const float* dy, const float* wt, const float* bn_x, const float* mean, const float* rstd,
                             const float* scale, const float* shift, int act, int seg_n, float* g, float* sums, float* dgamma, float* dbeta,
                             int accumulate, void* workspace, int64_t workspace_bytes, suhpe_stream_t stream);
/* input gradient; wt is the transposed+flipped filter made by suhpe_conv2d_transpose_filter */
int suhpe_conv2d_dgrad(const suhpe_conv_t* d, const float* dy, const float* wt, float* dx, suhpe_stream_t stream);
/* the same input gradient with a skip connection's gradient added in the epilogue: dx = dgrad(dy) + skip (skip: the shape of dx,
 * may be NULL) -- the backward of `x + block(x)` (torchvision MBConv / FusedMBConv use_res_connect) without a separate add pass */
int suhpe_conv2d_dgrad_add(const suhpe_conv_t* d, const float* dy, const float* wt, const float* skip, float* dx, suhpe_stream_t stream);
int suhpe_conv2d_transpose_filter(const suhpe_conv_t* d, const float* w, float* wt, suhpe_stream_t stream);
/* The same for many filters in one launch: jobs_dev = device array of njobs records of six int64 (w, wt, groups, cout/groups,
 * cin/groups, r*s), chunk_start_dev[j] = first 1024-element chunk of job j, chunk_start_dev[njobs] = total_chunks. */
int suhpe_transpose_filters_batched(const void* jobs_dev, const int64_t* chunk_start_dev, int njobs, int64_t total_chunks,
                                    suhpe_stream_t stream);
/* name of the kernel instance a call dispatches to (pass 0 fwd, 1 dgrad, 2 wgrad) — for profiling/roofline reports */
const char* suhpe_conv2d_kernel_name(const suhpe_conv_t* d, int pass);
/* name of the dense-conv kernel the LAST suhpe_conv2d_* / suhpe_linear_* call of this thread actually launched (the prediction above is made
 * from the descriptor alone; run-time facts -- a bias, an unaligned skip tensor -- can send a call to another kernel).  "" before the first call. */
const char* suhpe_last_kernel_name(void);
/* filter gradient; dw += sum (accumulate != 0) or dw = sum.  workspace: device scratch */
int64_t suhpe_conv2d_wgrad_workspace_bytes(const suhpe_conv_t* d);
int suhpe_conv2d_wgrad(const suhpe_conv_t* d, const float* x, const float* dy, float* dw, int accumulate,
                       void* workspace, int64_t workspace_bytes, suhpe_stream_t stream);
/* the two launches of suhpe_conv2d_wgrad separately: split-K partial slabs, then the fixed-order slab reduction */
int suhpe_conv2d_wgrad_partial(const suhpe_conv_t* d, const float* x, const float* dy, void* workspace,
                               int64_t workspace_bytes, suhpe_stream_t stream);
int suhpe_conv2d_wgrad_reduce(const suhpe_conv_t* d, const void* workspace, float* dw, int accumulate,
                              suhpe_stream_t stream);
/* Deferred form for a whole backward pass: every layer runs suhpe_conv2d_wgrad_partial into its own slab buffer, then ONE
 * launch reduces them all (dw[i] += sum over splits, same fixed order: bit-identical to suhpe_conv2d_wgrad_reduce with
 * accumulate = 1).  jobs_dev: njobs records of four 64-bit words (slab pointer, dw pointer, element count, splits =
 * suhpe_conv2d_wgrad_splits(d)); chunk_start_dev[njobs + 1]: prefix sum of ceil(count / 1024) per job. */
int suhpe_conv2d_wgrad_splits(const suhpe_conv_t* d);
int suhpe_wgrad_reduce_batched(const void* jobs_dev, const int64_t* chunk_start_dev, int njobs, int64_t total_chunks,
                               suhpe_stream_t stream);
/* depthwise 3x3 (groups == cin == cout, cin % 4 == 0); w in torch layout [c][r*s] */
int suhpe_dwconv_fwd(const suhpe_conv_t* d, const float* x, const float* w, float* y, suhpe_stream_t stream);
int suhpe_dwconv_dgrad(const suhpe_conv_t* d, const float* dy, const float* w, float* dx, suhpe_stream_t stream);
/* Input gradient of a depthwise conv whose INPUT was act(BN(bn_x)) with train-mode statistics (torchvision MBConv: expand conv ->
 * BatchNorm -> SiLU -> depthwise, reference src/networks.py:303): besides dx = the gradient of the activated tensor, the kernel
 * reduces that BatchNorm's backward sums over what it produces, g = dx * act'(gamma xhat + beta): sums[c][2] = (sum g, sum g xhat)
 * (seg_n > 0: two batch segments of seg_n / n - seg_n samples, mean / rstd [2][c], sums [2][c][2]); dgamma / dbeta (+)= their totals.
 * The BatchNorm backward then needs only its apply pass (suhpe_bn_act_bwd_apply / _apply_seg): no reduction pass over (bn_x, dx).
 * Served shapes: suhpe_dwconv_dgrad_bnsums_ok(d) (3x3 / pad 1 / stride 1, the sliding-window kernel); workspace: device scratch.
 * (Measured and dropped: an apply pass that recomputes dx from dy as well -- dx never stored -- made the step 3 ms SLOWER: the stencil +
 * SiLU' twice costs more than the tensor pass it saves.) */
int suhpe_dwconv_dgrad_bnsums_ok(const suhpe_conv_t* d);
int64_t suhpe_dwconv_dgrad_bnsums_workspace_bytes(const suhpe_conv_t* d);
int suhpe_dwconv_dgrad_bnsums(const suhpe_conv_t* d, const float* dy, const float* w, float* dx, const float* bn_x, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, int act, int seg_n, float* sums, float* dgamma,
                              float* dbeta, int accumulate, void* workspace, int64_t workspace_bytes, suhpe_stream_t stream);
/* The depthwise conv's WHOLE backward in one launch (round 6; 3x3 / pad 1 / stride 1 on 14 x 14 or 7 x 7 maps: torchvision MBConv's depthwise
 * Conv2dNormActivation, reached through /root/reference/src/networks.py:301-303): the input gradient dx and the producer BatchNorm's backward sums as
 * suhpe_dwconv_dgrad_bnsums gives them, AND the filter gradient dw (what suhpe_dwconv_bnact_wgrad computes from act(BN(bn_x)): same BatchNorm, so no
 * second parameter set), from ONE staged tile of the output gradient.  Section form (mid_y != NULL): `dy` is then dout = d loss / d (z s) of the
 * SqueezeExcitation section and the conv's output gradient is formed while the tile is filled -- suhpe_se_bn_bwd_apply's arithmetic on (mid_y, dout,
 * mid_s, mid_dpool, mid_chan, the depthwise BatchNorm's mean / rstd [sets][c], gamma, beta), never written to memory.  seg_n = samples of the first
 * batch segment (0: one).  _ok: can this descriptor take the call (SUHPE_DW_BWD_FUSED=0: never).  Results equal the three-kernel path within fp32
 * rounding (other summation order of the filter gradient; one sigmoid per element shared by z and act'). */
int suhpe_dwconv_bwd_fused_ok(const suhpe_conv_t* d);
int64_t suhpe_dwconv_bwd_fused_workspace_bytes(const suhpe_conv_t* d);
int suhpe_dwconv_bwd_fused(const suhpe_conv_t* d, const float* dy, const float* w, float* dx, const float* bn_x, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int act, int seg_n, float* sums, float* dgamma, float* dbeta, int accumulate,
                           float* dw, int dw_accumulate, const float* mid_y, const float* mid_s, const float* mid_dpool, const float* mid_chan,
                           const float* mid_mean, const float* mid_rstd, const float* mid_gamma, const float* mid_beta, int mid_act,
                           int mid_train_stats, void* workspace, int64_t workspace_bytes, suhpe_stream_t stream);
int suhpe_bn_act_bwd_apply_seg(const float* x, const float* dy, int64_t rows, int64_t rows_a, int c, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int act, const float* sums, float* dx, suhpe_stream_t stream);
int64_t suhpe_dwconv_wgrad_workspace_bytes(const suhpe_conv_t* d);
int suhpe_dwconv_wgrad(const suhpe_conv_t* d, const float* x, const float* dy, float* dw, int accumulate,
                       void* workspace, int64_t workspace_bytes, suhpe_stream_t stream);
/* Depthwise forward that also emits the BatchNorm statistics records of its OUTPUT (the depthwise conv of torchvision's MBConv and of
 * tiny_vit.py:91-93,130-131,323-324 is always followed by a train-mode BatchNorm): stats = [segments][blocks][C] float4 (S, Q, k, n),
 * blocks = suhpe_dwconv_fwd_stats_blocks(d) (0: this shape has no statistics epilogue, use suhpe_dwconv_fwd + suhpe_bn_stats);
 * seg_n > 0: samples [0, seg_n) are the first batch segment (two record planes), seg_n = 0: one plane.  The records feed
 * suhpe_bn_stats_from_partials / _seg like the GEMM epilogue's. */
int suhpe_dwconv_fwd_stats_blocks(const suhpe_conv_t* d);
int suhpe_dwconv_fwd_stats(const suhpe_conv_t* d, const float* x, const float* w, float* y, float* stats, int seg_n, suhpe_stream_t stream);
/* The same depthwise forward / filter gradient with the producer's BatchNorm + activation applied while loading the input:
 * x is the raw output of the preceding conv, the kernels use act(gamma*(x-mean)*rstd + beta) (zero at padding taps), so the
 * normalised tensor of an MBConv's expand conv never exists in memory (torchvision MBConv: Conv2dNormActivation 1x1 ->
 * Conv2dNormActivation depthwise).  gamma / beta may be NULL.  seg_n > 0: batch segments -- mean / rstd hold two planes
 * [2][C], the first for the samples [0, seg_n), the second for the rest (seg_n = 0: one plane [C]).  stats (forward, may be
 * NULL): the output's BatchNorm statistics records exactly as suhpe_dwconv_fwd_stats writes them, same seg_n. */
int suhpe_dwconv_bnact_fwd(const suhpe_conv_t* d, const float* x, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, int act, int seg_n, const float* w, float* y, float* stats, suhpe_stream_t stream);
int suhpe_dwconv_bnact_wgrad(const suhpe_conv_t* d, const float* x, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, int act, int seg_n, const float* dy, float* dw, int accumulate, void* workspace,
                             int64_t workspace_bytes, suhpe_stream_t stream);
/* db[c] (+)= sum over rows of dy (rows, c) */
int suhpe_bias_grad(const float* dy, int64_t rows, int c, float* db, int accumulate, void* workspace,
                    suhpe_stream_t stream);   /* workspace >= suhpe_bn_workspace_bytes(c) */

/* ------------------------------------------------------------------ data-side augmentation ---- */
/* The per-sample PIL / torchvision chain of src/datasets/dataset_300WLP.py:103-161 (process_annotated_image;
 * dataset_WildHead.py:76-124 is the same chain) for n decoded 224x224 RGB crops in device memory, src (n,224,224,3) uint8:
 *   flags[n]      bit 0 = flip left-right, bit 1 = ImageFilter.BLUR (applied after the flip)
 *   weak_box[n][4], strong_box[n][4] = (top, left, height, width) of RandomResizedCrop (height, width <= 224)
 *   rot_coef[n][8] = 16.16 fixed-point affine coefficients a0..a5 of Image.rotate(angle, center=(112,112), expand=True)
 *                    followed by the centre 224 crop origin (x0, y0); NULL = no rotation
 *   tables        device copy of suhpe_augment_build_tables' output (Pillow's bilinear resize coefficients)
 * Outputs: weak_out / strong_out (n,224,224,3) fp32 = Normalize(ToTensor(view)) in the pixel-major layout of the backbone
 * (strong_* may be NULL together); weak_u8 / strong_u8 optional uint8 copies of the views (debug / parity).
 * Every pixel operation reproduces Pillow bit for bit.  workspace >= suhpe_augment_workspace_bytes(n). */
int64_t suhpe_augment_tables_bytes(void);
int suhpe_augment_build_tables(int32_t* host_tables);
int64_t suhpe_augment_workspace_bytes(int n);
int suhpe_augment_views(const uint8_t* src, int n, const int32_t* flags, const int32_t* weak_box, const int64_t* rot_coef,
                        const int32_t* strong_box, const int32_t* tables, void* workspace, float* weak_out, float* strong_out,
                        uint8_t* weak_u8, uint8_t* strong_u8, suhpe_stream_t stream);

/* ------------------------------------------------------------------ normalisation + activation ----------- */
/* BatchNorm (train): per-channel statistics of x (rows, c).  Writes mean/rstd (c each) and updates
 * running_mean / running_var (unbiased) / num_batches_tracked (int64, +1) when running_mean != NULL.
 * workspace >= suhpe_bn_workspace_bytes(c). */
int64_t suhpe_bn_workspace_bytes(int c);
int suhpe_bn_stats(const float* x, int64_t rows, int c, float eps, float momentum, float* mean, float* rstd,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, void* workspace,
                   suhpe_stream_t stream);
/* second half of suhpe_bn_stats for partials produced elsewhere (layout [gx][channel][4] = (S, Q, k, n) shifted records) */
int suhpe_bn_stats_from_partials(const float* part, int gx, int64_t rows, int c, float eps, float momentum, float* mean,
                                 float* rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 suhpe_stream_t stream);
/* the same over records [first, first + count) plus one optional extra record per channel (extra: c records or NULL) */
int suhpe_bn_stats_from_partials_range(const float* part, int first, int count, const float* extra, int64_t rows, int c, float eps,
                                       float momentum, float* mean, float* rstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, suhpe_stream_t stream);
/* RepVGG re-parameterisation for eval (src/repvgg.py:104-142 get_equivalent_kernel_bias): 3x3 branch + 1x1 branch + identity
 * BatchNorm -> one 3x3 filter w_eq (layout of w3) and bias b_eq; `ones` (cout) is filled with 1 so that
 * suhpe_conv2d_fwd_affine(d, x, w_eq, ones, b_eq, RELU, ...) is the deploy-mode block.  bn3 / bn1 / bnid are arrays of four
 * device pointers (gamma, beta, running_mean, running_var); bnid = NULL when the block has no identity branch. */
int suhpe_repvgg_fuse(const float* w3, const float* w1, const float* const* bn3, const float* const* bn1, const float* const* bnid,
                      float eps, int cout, int cin_per_group, float* w_eq, float* ones, float* b_eq, suhpe_stream_t stream);
/* eval: scale = gamma/sqrt(running_var+eps), shift = beta - running_mean*scale (gamma / beta may be NULL) */
int suhpe_bn_eval_affine(const float* running_mean, const float* running_var, const float* gamma, const float* beta, int c,
                         float eps, float* scale, float* shift, suhpe_stream_t stream);
/* SyncBN (global-batch BatchNorm statistics over data-parallel ranks with equally many rows; the reference has no multi-GPU code, this is the
 * opt-in --sync_bn mode of semiuhpe_amd/layers.py): moments[0..c) = mean, moments[c..2c) = E[x^2] of this rank's rows -- the pair that an AVG
 * all-reduce turns into the global batch's (suhpe_comm_allreduce(average = 1) / torch.distributed) -- and back: mean / rstd of the global batch of
 * rows_global rows plus the momentum update of the running statistics (unbiased variance; num_batches_tracked += 1; the three may be NULL). */
int suhpe_syncbn_moments(const float* mean, const float* rstd, int c, float eps, float* moments, suhpe_stream_t stream);
int suhpe_syncbn_finish(const float* moments, int c, float eps, float momentum, int64_t rows_global, float* mean, float* rstd,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked, suhpe_stream_t stream);
/* eval: mean/rstd from running stats */
int suhpe_bn_eval_stats(const float* running_mean, const float* running_var, int c, float eps, float* mean,
                        float* rstd, suhpe_stream_t stream);
/* y = act(gamma*(x-mean)*rstd + beta + residual) ; gamma/beta/residual may be NULL (=1/0/0) */
int suhpe_bn_act_fwd(const float* x, int64_t rows, int c, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, int act, const float* residual, float* y, suhpe_stream_t stream);
/* backward of the above in train mode: given dy (grad of the activated output), recomputes the activation
 * derivative from x and returns dx, dgamma (+)=, dbeta (+)=.  train_stats=0 treats mean/rstd as constants
 * (eval mode).  y_out: the forward output, needed only when a residual was added before a ReLU (then the mask is
 * y_out > 0); NULL otherwise.  workspace as suhpe_bn_stats. */
int suhpe_bn_act_bwd(const float* x, const float* dy, const float* y_out, int64_t rows, int c, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int act, int train_stats, float* dx,
                     float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream);
/* The residual block's tail in one pass: y = skip + rowscale[row / hw] * act(BN(x)) (torchvision MBConv / FusedMBConv:
 * x + StochasticDepth('row')(block(x)); rowscale (rows/hw floats: keep mask / keep probability) may be NULL = 1), and
 * its backward: the gradient entering the BatchNorm is dy * rowscale; d(skip) = dy is the caller's. */
int suhpe_bn_act_skip_fwd(const float* x, int64_t rows, int c, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, int act, const float* skip, const float* rowscale, int hw, float* y,
                          suhpe_stream_t stream);
int suhpe_bn_act_skip_bwd(const float* x, const float* dy, int64_t rows, int c, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, int train_stats, const float* rowscale, int hw,
                          float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream);
/* RepVGG block tail in training (src/repvgg.py:67-76): y = relu(BN(x[0]) + BN(x[1]) [+ BN(x[2])]) with per-branch batch
 * statistics, one pass; nb = 2 or 3 branches, every array argument holds nb pointers.  mean / rstd are [c] (rows_a = 0) or
 * [2][c] for two batch segments.  The backward shares g = dout * (y > 0): dx[j] = BatchNorm_j backward of g,
 * dgamma[j] / dbeta[j] += (always accumulated).  workspace >= suhpe_bn_workspace_bytes(c). */
int suhpe_repvgg_bn_fwd(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                        const float* const* beta, int nb, int64_t rows, int64_t rows_a, int c, float* y, suhpe_stream_t stream);
/* The same forward, also emitting the BatchNorm statistics records of its OUTPUT (round 5): the next RepVGG block's identity branch normalises
 * exactly this tensor, so its statistics come from the records (suhpe_bn_stats_from_partials / _seg) instead of a pass of their own.
 * stats: suhpe_repvgg_bn_fwd_stats_blocks(rows, rows_a, c, &first) records of [c] float4 (S, Q, k, n), the first `first` of them covering the rows
 * of the first batch segment (all of them for rows_a == 0); tensors 16-byte aligned when c % 4 == 0. */
int suhpe_repvgg_bn_fwd_stats_blocks(int64_t rows, int64_t rows_a, int c, int32_t* first_records);
int suhpe_repvgg_bn_fwd_stats(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                              const float* const* beta, int nb, int64_t rows, int64_t rows_a, int c, float* y, float* stats,
                              suhpe_stream_t stream);
int suhpe_repvgg_bn_bwd(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma, int nb,
                        const float* dout, const float* y, int64_t rows, int64_t rows_a, int c, float* const* dx,
                        float* const* dgamma, float* const* dbeta, void* workspace, suhpe_stream_t stream);

/* suhpe_repvgg_bn_bwd without the output tensor (round 5): the ReLU mask y > 0 is recomputed from the branch inputs with the forward's own
 * arithmetic (explicit fused multiply-adds in one fixed order: the same bits), one stream less in each of the two passes. */
int suhpe_repvgg_bn_bwd_remask(const float* const* x, const float* const* mean, const float* const* rstd, const float* const* gamma,
                               const float* const* beta, int nb, const float* dout, int64_t rows, int64_t rows_a, int c,
                               float* const* dx, float* const* dgamma, float* const* dbeta, void* workspace, suhpe_stream_t stream);

/* Two batch segments in one launch: rows [0, rows_a) and [rows_a, rows) of x are normalised with separate statistics (the
 * labeled and the strong batch of a step run through the network as one, layers.batch_segments).  mean / rstd are [2][c];
 * the running statistics take the two momentum updates in order and num_batches_tracked += 2; dgamma / dbeta receive the sum
 * over both segments.  Same arithmetic as calling the one-segment entry points on the two halves (tested).  The skip /
 * rowscale forms need whole samples per segment (rows_a % hw == 0). */
int suhpe_bn_stats_seg(const float* x, int64_t rows, int64_t rows_a, int c, float eps, float momentum, float* mean, float* rstd,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, void* workspace,
                       suhpe_stream_t stream);
/* from the records of suhpe_conv2d_fwd_stats_seg: first_records = records of the first segment, boundary = stats_extra or NULL */
int suhpe_bn_stats_from_partials_seg(const float* part, int tiles, int first_records, const float* boundary, int64_t rows,
                                     int64_t rows_a, int c, float eps, float momentum, float* mean, float* rstd,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                     suhpe_stream_t stream);
/* suhpe_bn_stats_from_partials (first_records == tiles, rows_a == rows, boundary NULL, mean / rstd [c]) or _seg (otherwise) for LONG record
 * lists: the finalizers above reduce a channel's records on ceil(c / 8) workgroups, which takes 20-60 us for the 4 000-16 000 records of a
 * 24..192-channel layer at 112 x 112 / 56 x 56; this form first adds runs of records on workgroups all over the chip (fp64 pairs written
 * over each run's first record: `part` is CONSUMED) and finishes from <= 32 pairs per channel and segment.  Same statistics to fp64 rounding. */
int suhpe_bn_stats_from_partials_consume(float* part, int tiles, int first_records, const float* boundary, int64_t rows, int64_t rows_a, int c,
                                         float eps, float momentum, float* mean, float* rstd, float* running_mean, float* running_var,
                                         int64_t* num_batches_tracked, suhpe_stream_t stream);
/* post = 0: act(BN(x) + residual); post = 1: residual + rowscale[sample] * act(BN(x)) */
int suhpe_bn_act_fwd_seg(const float* x, int64_t rows, int64_t rows_a, int c, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int act, const float* residual, int post,
                         const float* rowscale, int hw, float* y, suhpe_stream_t stream);
/* The BatchNorm(+act) backward in two calls, for data-parallel callers that normalise over the GLOBAL batch (SyncBN, SURVEY.md
 * 8e (2)): _sums leaves (sum g, sum g*xhat) per channel in sums[2c] (and accumulates dgamma / dbeta from the local rows); the
 * caller all-reduces sums (SUM); _apply forms dx with the means over total_rows = rows of all ranks.  mean / rstd are the
 * global-batch statistics. */
int suhpe_bn_act_bwd_sums(const float* x, const float* dy, const float* y_out, int64_t rows, int c, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, int act, const float* rowscale, int hw,
                          float* sums, float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream);
int suhpe_bn_act_bwd_apply(const float* x, const float* dy, const float* y_out, int64_t rows, int64_t total_rows, int c,
                           const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                           const float* sums, const float* rowscale, int hw, float* dx, suhpe_stream_t stream);
/* y = act(BN(x)) and pool[sample][c] = mean over the sample's hw rows of y in the same pass (the squeeze of the
 * SqueezeExcitation block behind a depthwise conv's BatchNorm); rows_a = 0: one segment (mean / rstd [c]), else two ([2][c]). */
int suhpe_bn_act_pool_fwd(const float* x, int64_t rows, int64_t rows_a, int c, int hw, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, int act, float* y, float* pool, suhpe_stream_t stream);
/* MBConv middle section fused around the depthwise conv's BatchNorm (torchvision MBConv: dw conv -> BatchNorm -> SiLU ->
 * SqueezeExcitation -> project conv; built by models.efficientnet_v2_s at reference src/networks.py:303).  x = the depthwise conv's
 * raw output (rows = n*hw, c); z = act(BN(x)) is never stored.  Whole samples per segment (rows % hw == 0, rows_a % hw == 0).
 *   suhpe_bn_act_pool_fwd with y == NULL   pool[n][c] = mean_hw z                         (the squeeze alone)
 *   suhpe_bn_act_scale_fwd                 y = z * s[n][c]                                 (excitation applied)
 *   suhpe_se_bn_bwd_reduce                 sums[5][n][c]: per-sample sums over hw of dout z | dout a' | dout a' xhat | a' | a' xhat
 *                                          (a' = act'(u)); sums[0] is ds, the gradient of the excitation (input of suhpe_se_mlp_bwd)
 *   suhpe_se_bn_bwd_combine                chan[segments][2c] = (sum g, sum g xhat), g = (dout s + dpool/hw) a'; dgamma / dbeta (+)=
 *   suhpe_se_bn_bwd_apply                  dy = gamma rstd (g - mean g - xhat mean(g xhat))   (train_stats = 0: gamma rstd g) */
int suhpe_bn_act_scale_fwd(const float* x, int64_t rows, int64_t rows_a, int c, int hw, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int act, const float* s, float* y, suhpe_stream_t stream);
int suhpe_se_bn_bwd_reduce(const float* x, const float* dout, int64_t rows, int64_t rows_a, int c, int hw, const float* mean,
                           const float* rstd, const float* gamma, const float* beta, int act, float* sums, suhpe_stream_t stream);
int suhpe_se_bn_bwd_combine(const float* sums, const float* s, const float* dpool, int n, int n_a, int c, int hw, float* chan,
                            float* dgamma, float* dbeta, int accumulate, suhpe_stream_t stream);
int suhpe_se_bn_bwd_apply(const float* x, const float* dout, int64_t rows, int64_t rows_a, int c, int hw, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, int act, const float* s, const float* dpool,
                          const float* chan, int train_stats, float* dy, suhpe_stream_t stream);
int suhpe_bn_act_bwd_seg(const float* x, const float* dy, const float* y_out, int64_t rows, int64_t rows_a, int c,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                         const float* rowscale, int hw, float* dx, float* dgamma, float* dbeta, int accumulate,
                         void* workspace, suhpe_stream_t stream);

/* LayerNorm over the last dim (rows, c) (src/tiny_vit.py nn.LayerNorm) */
int suhpe_layernorm_fwd(const float* x, int64_t rows, int c, float eps, const float* gamma, const float* beta,
                        float* y, float* mean, float* rstd, suhpe_stream_t stream);
int suhpe_layernorm_bwd(const float* x, const float* dy, int64_t rows, int c, const float* mean, const float* rstd,
                        const float* gamma, float* dx, float* dgamma, float* dbeta, int accumulate,
                        void* workspace, suhpe_stream_t stream);
/* the same backward with the gradient that bypassed the normalised sub-block added to dx (x + drop_path(f(LN(x))), src/tiny_vit.py:366,378):
 * dx = LayerNorm'(dy) + skip, no separate add pass (skip: the shape of dx, or NULL) */
int suhpe_layernorm_bwd_skip(const float* x, const float* dy, int64_t rows, int c, const float* mean, const float* rstd, const float* gamma,
                             const float* skip, float* dx, float* dgamma, float* dbeta, int accumulate, void* workspace, suhpe_stream_t stream);
/* plain activation y = act(x + bias[c]) and its backward dx = dy * act'(x + bias) */
int suhpe_act_fwd(const float* x, int64_t rows, int c, const float* bias, int act, float* y, suhpe_stream_t stream);
int suhpe_act_bwd(const float* x, const float* dy, int64_t rows, int c, const float* bias, int act, float* dx,
                  suhpe_stream_t stream);

/* ------------------------------------------------------------------ pooling, SE, residual, dropout ------- */
/* mean over the hw axis: x (n,hw,c) -> y (n,c) */
/* nn.MaxPool2d(k, stride, pad) on (N,H,W,C) (ResNet stem, torchvision resnet.py): ties go to the first window position, as
 * in torch; idx (N,Ho,Wo,C) uint8 = position inside the window, kept for the backward (gather form, deterministic). */
int suhpe_maxpool_fwd(const float* x, int n, int h, int w, int c, int k, int stride, int pad, float* y, uint8_t* idx,
                      suhpe_stream_t stream);
int suhpe_maxpool_bwd(const float* dy, const uint8_t* idx, int n, int h, int w, int c, int k, int stride, int pad, float* dx,
                      suhpe_stream_t stream);
int suhpe_avgpool_fwd(const float* x, int n, int hw, int c, float* y, suhpe_stream_t stream);
int suhpe_avgpool_bwd(const float* dy, int n, int hw, int c, float* dx, suhpe_stream_t stream);
/* squeeze-excite scale y[n,p,c] = x[n,p,c]*s[n,c] */
int suhpe_se_scale_fwd(const float* x, const float* s, int n, int hw, int c, float* y, suhpe_stream_t stream);
/* ds[n,c] = sum_p dy*x */
int suhpe_se_scale_bwd_s(const float* dy, const float* x, int n, int hw, int c, float* ds, suhpe_stream_t stream);
/* dx = dy*s + dpool/hw (dpool may be NULL) */
int suhpe_se_scale_bwd_x(const float* dy, const float* s, const float* dpool, int n, int hw, int c, float* dx,
                         suhpe_stream_t stream);
/* squeeze-excite MLP per sample: s = sigmoid(W2 silu(W1 p + b1) + b2); w1 [csq][c], w2 [c][csq]; a1 (n,csq) is kept for
 * backward.  c % 4 == 0, c <= 2048, csq % 4 == 0, csq <= 64. */
int suhpe_se_mlp_fwd(const float* p, const float* w1, const float* b1, const float* w2, const float* b2, int n, int c,
                     int csq, float* a1, float* s, suhpe_stream_t stream);
/* backward: ds (n,c) -> dp (n,c); dw1/db1/dw2/db2 += gradients.  workspace >= n*(c+2*csq) floats */
int suhpe_se_mlp_bwd(const float* ds, const float* s, const float* a1, const float* p, const float* w1, const float* w2,
                     int n, int c, int csq, float* dp, float* dw1, float* db1, float* dw2, float* db2, void* workspace,
                     suhpe_stream_t stream);
/* y = a + b*rowscale[n]  (residual + stochastic depth 'row' mode; rowscale NULL = 1); per = elements per sample */
int suhpe_add_rowscale(const float* a, const float* b, const float* rowscale, int n, int64_t per, float* y,
                       suhpe_stream_t stream);
/* y = x*rowscale[n] */
int suhpe_mul_rowscale(const float* x, const float* rowscale, int n, int64_t per, float* y, suhpe_stream_t stream);
/* y = x * mask (elementwise, mask already holds 0 or 1/keep) */
int suhpe_mul(const float* x, const float* m, int64_t count, float* y, suhpe_stream_t stream);
/* out[i] = a[i] + b[i] */
int suhpe_add(const float* a, const float* b, int64_t count, float* y, suhpe_stream_t stream);
int suhpe_add3(const float* a, const float* b, const float* c, int64_t count, float* y, suhpe_stream_t stream);   /* (a + b) + c */
/* Bernoulli(keep)/keep mask from a counter-based generator (Philox4x32-10).  Stream position = counter[1] (device,
 * 2 x uint64: seed, offset) + host_offset.  advance != 0 bumps the device offset after the draw (graph replays then
 * draw fresh numbers); with advance == 0 the caller tracks the position in host_offset and saves the extra launch. */
int suhpe_dropout_mask(uint64_t* counter, uint64_t host_offset, int advance, float keep, int64_t count, float* mask,
                       suhpe_stream_t stream);
/* every mask of a forward pass (StochasticDepth('row') of each residual block, reference src/networks.py:303 / torchvision MBConv) from ONE
 * launch: request k draws count[k] values with keep[k] at stream position counter[1] + host_offset + rel[k] into out + out_off[k]
 * (device arrays of nreq entries).  With rel[k] = the cumulative ceil(count / 4) of the requests before it the values equal those of
 * nreq consecutive suhpe_dropout_mask calls. */
int suhpe_dropout_mask_batch(const uint64_t* counter, uint64_t host_offset, const int64_t* rel, const float* keep, const int64_t* count,
                             const int64_t* out_off, int nreq, int64_t max_count, float* out, suhpe_stream_t stream);

/* ------------------------------------------------------------------ reductions for the loss ------------- */
/* out[0] = scale * sum_i x[i]*(w ? w[i] : 1) */
int suhpe_weighted_sum(const float* x, const float* w, int64_t count, float scale, float* out, suhpe_stream_t stream);
/* out[0] = sum(mask*x)/sum(mask): masked diagnostics of src/agent.py:171-181 without boolean compaction */
int suhpe_masked_mean(const float* x, const float* mask, int64_t count, float* out, suhpe_stream_t stream);
/* out[i] = src[0]*scale  (gradient of a mean: broadcast of a device scalar) */
int suhpe_broadcast_scale(const float* src, float scale, int64_t count, float* out, suhpe_stream_t stream);
/* out = alpha*a + beta*b  (b may be NULL); used for loss_all = loss + lambda*unsup (src/agent.py:206) */
int suhpe_axpby(const float* a, float alpha, const float* b, float beta, int64_t count, float* out,
                suhpe_stream_t stream);
/* y[i,:] = x[i,:] * g[i] * scale (rows, c) */
int suhpe_scale_rows(const float* x, const float* g, int64_t rows, int c, float scale, float* y, suhpe_stream_t stream);

/* ------------------------------------------------------------------ optimiser and EMA teacher ----------- */
/* torch.optim.Adam step (src/agent.py:37,210), flat buffers: p,g,m,v (count); *step_host is the 1-based
 * step number for bias correction. */
int suhpe_adam_step(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, suhpe_stream_t stream);
/* src/agent.py:277-299: ema = alpha*ema + (1-alpha)*p */
int suhpe_ema_update(float* ema, const float* p, int64_t count, float alpha, suhpe_stream_t stream);
int suhpe_fill(float* p, int64_t count, float value, suhpe_stream_t stream);

/* ------------------------------------------------------------------ TinyViT windowed attention ----------- */
/* src/tiny_vit.py:262-286 over windows addressed IN PLACE in a (B,H,W,C) token tensor (no window partition / reverse
 * copies, tiny_vit.py:356-366).  qkv: (B*H*W, heads*3*dh), per head [q(dh) k(dh) v(dh)] as the reference's
 * qkv.view(B,N,heads,-1).split; bias_table: (heads, noff); bias_idx: int32 (ws*ws, ws*ws); out: (B*H*W, heads*dh).
 * probs: (B*nwin*heads, N, N) saved for backward, NULL for inference.  dh must be 32, ws*ws <= 196, ws | H, ws | W.
 * workspace: suhpe_window_attn_fwd_workspace_bytes(ws, heads) bytes (the per-head bias expanded to a dense, padded
 * (heads, NP, NP) matrix by a first small launch); qkv must be smaller than 2 GiB. */
int64_t suhpe_window_attn_fwd_workspace_bytes(int ws, int heads);
int suhpe_window_attn_fwd(const float* qkv, const float* bias_table, const int32_t* bias_idx, int B, int H, int W,
                          int ws, int heads, int dh, int noff, float scale, float* out, float* probs, void* workspace,
                          suhpe_stream_t stream);
int64_t suhpe_window_attn_probs_elems(int B, int H, int W, int ws, int heads);
/* dqkv gets every element written; workspace: probs_elems floats, left holding dS (the score gradient, (windows, heads, N, N)).
 * dbias_table: NULL, or (+)= the bias-table gradient by atomic accumulation (order-dependent rounding); the reproducible form is
 * dbias_table = NULL here followed by suhpe_window_attn_dbias on the same workspace. */
int suhpe_window_attn_bwd(const float* qkv, const float* probs, const float* dout, const int32_t* bias_idx, int B,
                          int H, int W, int ws, int heads, int dh, int noff, float scale, float* dqkv,
                          float* dbias_table, void* workspace, suhpe_stream_t stream);

/* The recomputing form of the same attention (round 5; what the training step runs): the forward keeps TWO floats per query row -- the row max
 * of the scaled, biased scores and the reciprocal of the row's exp sum -- instead of the N x N probabilities, and the backward recomputes the
 * probabilities from q k^T and those two (bit-identical to the forward's) and takes delta_i = sum_j P_ij dP_ij as dO_i . O_i from the forward's
 * output.  Same operands and results as suhpe_window_attn_fwd / _bwd (src/tiny_vit.py:262-286 and its autograd backward); ws*ws <= 196.
 *   rowstats: (windows*heads, NP, 2) floats, NP = 64 for ws*ws <= 64 else 224: suhpe_window_attn_rowstats_elems(); NULL for inference.
 *   workspace / bias_workspace: suhpe_window_attn_rs_workspace_bytes(ws, heads) bytes (two tile-ordered expansions of the bias table, rebuilt
 *     by a small first launch of either call: the two calls need not share it).
 *   fwd_out: the forward's `out`.  dS_workspace: suhpe_window_attn_probs_elems floats, left holding dS for suhpe_window_attn_dbias.
 *   dbias_table: as in suhpe_window_attn_bwd (NULL + suhpe_window_attn_dbias = the reproducible form). */
int64_t suhpe_window_attn_rs_workspace_bytes(int ws, int heads);
int64_t suhpe_window_attn_rowstats_elems(int B, int H, int W, int ws, int heads);
int suhpe_window_attn_fwd_rs(const float* qkv, const float* bias_table, const int32_t* bias_idx, int B, int H, int W,
                             int ws, int heads, int dh, int noff, float scale, float* out, float* rowstats,
                             void* workspace, suhpe_stream_t stream);
int suhpe_window_attn_bwd_rs(const float* qkv, const float* fwd_out, const float* rowstats, const float* dout,
                             const float* bias_table, const int32_t* bias_idx, int B, int H, int W, int ws, int heads,
                             int dh, int noff, float scale, float* dqkv, float* dbias_table, void* dS_workspace,
                             void* bias_workspace, suhpe_stream_t stream);

/* Bias-table gradient of tiny_vit.Attention (attention_biases[:, attention_bias_idxs], src/tiny_vit.py:237-260) without atomics:
 * dbias_table[h][o] += sum over windows and over the (i, j) with bias_idx[i][j] == o of dS, in a fixed order.  inv_ptr (noff + 1) /
 * inv_list (N*N) is the inverse of bias_idx in CSR form (pairs i*N + j grouped by offset, ascending).  dS: suhpe_window_attn_bwd's
 * workspace.  workspace here: suhpe_window_attn_dbias_workspace_bytes(ws, heads). */
int64_t suhpe_window_attn_dbias_workspace_bytes(int ws, int heads);
int suhpe_window_attn_dbias(const float* dS, const int32_t* inv_ptr, const int32_t* inv_list, int B, int H, int W, int ws, int heads,
                            int noff, float* dbias_table, void* workspace, suhpe_stream_t stream);

/* ---- data-parallel exchange (new capability: the reference trains on one GPU, train.py:24-25).
 * One communicator per rank (= per process, = per GPU) over RCCL / xGMI; float32 payloads; every call is enqueued on `stream`
 * and returns without synchronising.  Bring-up: rank 0 calls suhpe_comm_unique_id, hands the 128 bytes to the other ranks by
 * any host channel (the Python host uses torch.distributed's store), every rank calls suhpe_comm_init with its device current.
 * RCCL is loaded at run time (librccl.so.1): suhpe_comm_available() == 0 on a box without it, and only this family fails.
 *   allreduce : buf <- sum (average = 0) or mean (average = 1) over ranks, in place  -- gradient buckets, BatchNorm buffers,
 *               loss scalars (replaces nothing in the reference; DistributedDataParallel semantics)
 *   allgather : recv[r * count_per_rank ...] <- rank r's send                          -- per-rank entropy vectors of
 *               compute_dynamic_entropy_threshold (src/agent.py:357-417)
 *   broadcast : buf <- root's buf                                                      -- initial weights / teacher copy */
typedef struct suhpe_comm_s* suhpe_comm_t;
int suhpe_comm_available(void);
int suhpe_comm_unique_id(void* id128);
int suhpe_comm_init(suhpe_comm_t* comm, const void* id128, int rank, int world);
int suhpe_comm_rank(suhpe_comm_t comm);
int suhpe_comm_world(suhpe_comm_t comm);
int suhpe_comm_allreduce(suhpe_comm_t comm, float* buf, int64_t count, int average, suhpe_stream_t stream);
int suhpe_comm_allgather(suhpe_comm_t comm, const float* send, float* recv, int64_t count_per_rank, suhpe_stream_t stream);
int suhpe_comm_broadcast(suhpe_comm_t comm, float* buf, int64_t count, int root, suhpe_stream_t stream);
int suhpe_comm_destroy(suhpe_comm_t comm);

/* ---- demo renderer: the reference's native Sim3DR functions, bit-identical at any alpha (csrc/raster.hip).
 * suhpe_mesh_normals  <- _get_normal (Sim3DR/lib/rasterize_kernel.cpp:163-221, called by Sim3DR/Sim3DR.py:8-11): per-vertex normals =
 *   un-normalised triangle normals summed over each vertex's triangles in triangle order, normalised.  vt_start[nver + 1] / vt_tris[3 ntri]:
 *   the vertex -> triangle adjacency in CSR form, each vertex's triangles in increasing index (host-built, fixed per mesh).
 * suhpe_rasterize     <- _rasterize (rasterize_kernel.cpp:225-297, Sim3DR/Sim3DR.py:14-29): z-buffer rasterisation of per-vertex colours
 *   into an interleaved uint8 image (h, w, c), triangles in index order, strict depth test against depth_buffer (h, w) (initialise to
 *   -1e8 as Sim3DR.py:24), blend (1 - alpha) * image + alpha * 255 * colour on every accepted triangle, reverse != 0 flips the image
 *   rows.  vertices (nver, 3) = (x, y, depth) in pixels, triangles (ntri, 3) int32, colors (nver, c) in [0, 1].  workspace: device scratch
 *   of suhpe_rasterize_workspace_bytes(ntri). */
int suhpe_mesh_normals(const float* vertices, const int* triangles, const int* vt_start, const int* vt_tris, int nver, int ntri,
                       float* normals, suhpe_stream_t stream);
int64_t suhpe_rasterize_workspace_bytes(int ntri);
int suhpe_rasterize(unsigned char* image, const float* vertices, const int* triangles, const float* colors, float* depth_buffer, int ntri,
                    int h, int w, int c, float alpha, int reverse, void* workspace, int64_t workspace_bytes, suhpe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SUHPE_H */
