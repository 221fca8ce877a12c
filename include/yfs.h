/*
 * yfs.h — C ABI of libyfs.so, the MI355X (gfx950) hot path of yolo-from-scratch.
 *
 * The reference (3141592/yolo-from-scratch) is pure Python on torch.nn and has no FFI of its
 * own; every entry point below names the reference call site (file:line in the upstream repo)
 * whose work it takes over.  Entry points marked "no reference code" implement behaviour the
 * reference only specifies in README.md / assets/yolo_loss_function.png; SPEC.md freezes their
 * conventions.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator in the
 *     Python host); the library never allocates, frees or retains memory.  Scratch is passed in;
 *     its size comes from the matching yfs_*_workspace_bytes().
 *   - tensors are dense in the documented layout.  Activations are NHWC ("channels last").
 *   - `stream` is a hipStream_t passed as void*.  Kernels are enqueued on it; nothing synchronises.
 *   - return value: 0 = success, negative = yfs_status.  yfs_last_error() gives the thread-local
 *     message of the last failure on the calling thread.
 *   - no global mutable state (calls from several host threads on distinct streams are safe) with two documented
 *     exceptions: the tuning knobs below (measurement aids) and the built-in scratch of yfs_yolov1_loss_noalloc_f32.
 */
#ifndef YFS_H
#define YFS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    YFS_OK = 0,
    YFS_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, unsupported shape */
    YFS_ERR_UNSUPPORTED = -2,   /* shape outside what the gfx950 kernels were built for */
    YFS_ERR_WORKSPACE = -3,     /* workspace too small / misaligned */
    YFS_ERR_LAUNCH = -4         /* hipGetLastError() after a launch was not hipSuccess */
} yfs_status;

int yfs_version(void);
/* copies the calling thread's last error message (NUL-terminated, truncated to n); returns its length */
int yfs_last_error(char *buf, size_t n);

/* Tuning knobs: measurement aids for tools/conv_bench.py, tools/probes and the forced-tile tests.  Process-wide
 * integers, 0 = the library's own choice (the default and the only supported production setting).  Their initial
 * values come from the environment variables of the same name (YFS_IGEMM_TILE, YFS_WGRAD_TILE, YFS_WGRAD_SPLITS,
 * YFS_NMS_FUSED, YFS_STEM_KERNEL), read once at the first use of any knob.  Changing a knob while another thread launches kernels
 * affects that thread's tile choice, never its results beyond fp32 summation order.
 *   YFS_KNOB_IGEMM_TILE    forward/dgrad workgroup tile: 128 | 64128 | 224 | 2562 | 256 | 8 (256x256, 8-phase) | 87 (224x256, 8-phase) |
 *                          192 (8-phase, 192-channel tile: masked dgrad with 192 output channels only) |
 *                          84 (8-phase with split-K whatever the shape; needs the _ws entry points)
 *   YFS_KNOB_WGRAD_TILE    weight-gradient tile: 64 | 12864 | 128 (wgrad_kernel, its own tile choice) | 8 (wgrad8_kernel: 256-wide tiles)
 *   YFS_KNOB_WGRAD_SPLITS  split-K factor of the weight gradient (> 0 forces it)
 *   YFS_KNOB_NMS_FUSED     1: yfs_decode_nms_f32 decodes inside the NMS kernel (one launch)
 *   YFS_KNOB_STEM_KERNEL   1: stem forward + pool on the tiled kernel instead of the weight-stationary one (bit-identical)
 *   YFS_KNOB_IGEMM_PERSIST 1: igemm8_kernel as one workgroup per TILE instead of one persistent workgroup per CU
 *                          (bit-identical; the hardware then hands tiles to whichever CU is free, which is what a process
 *                          wants whose CUs are shared with long-running kernels of another queue, e.g. RCCL's) */
typedef enum { YFS_KNOB_IGEMM_TILE = 0, YFS_KNOB_WGRAD_TILE = 1, YFS_KNOB_WGRAD_SPLITS = 2, YFS_KNOB_NMS_FUSED = 3,
               YFS_KNOB_STEM_KERNEL = 4, YFS_KNOB_IGEMM_PERSIST = 5, YFS_KNOB_COUNT = 6 } yfs_knob_id;
int yfs_set_knob(int knob, int value);
int yfs_get_knob(int knob);
/* Dispatch query (no reference counterpart: the reference has one conv code path, torch's): the kernel variant the
 * calling thread's last yfs_conv2d_* / yfs_linear_wgrad_sgd_bf16 call launched, e.g. "igemm8_kernel<224,256,epi0>" or
 * "wgrad8_kernel<256,256,splits7>".  Thread-local like yfs_last_error; the parity tests use it to assert that the
 * auto-dispatched kernel is the one they mean to check.  Copies at most n bytes (NUL-terminated); returns the length. */
int yfs_last_kernel(char *buf, size_t n);

/* ---- kernel set 3: box math, decode, NMS ---------------------------------------------------- */

/* iou_xywh_torch(pred, gt)  yolo_toy_regressor_torch.py:159-182 (numpy twin :121-138).
 * pred, gt [n,4] xywh; iou_out [n] per-pair IoU (may be NULL); mean_out [1] batch mean (may be NULL).
 * workspace: yfs_iou_xywh_workspace_bytes(n) bytes when mean_out != NULL. */
size_t yfs_iou_xywh_workspace_bytes(int64_t n);
int yfs_iou_xywh_f32(const float *pred, const float *gt, int64_t n, float *iou_out, float *mean_out,
                     void *workspace, size_t workspace_bytes, void *stream);

/* box_iou_xyxy(xywh_to_xyxy(a), xywh_to_xyxy(b))  practice/yolo_toy_regressor_3_torch.py:106-119
 * variant: 0 = A4 op order (above), 1 = A4b (side clamps, union+1e-9), 2 = A4c
 * (practice/yolo_toy_regressor_torch_20251015.py:226-278). */
int yfs_iou_variant_f32(const float *pred, const float *gt, int64_t n, int variant, float *iou_out, void *stream);

/* N x M IoU in the op order of iou_xywh_torch (:159-182).  a [na,4], b [nb,4] xywh -> out [na,nb]. */
int yfs_pairwise_iou_f32(const float *a, int64_t na, const float *b, int64_t nb, float *out, void *stream);

/* decode_head(t)  practice/yolo_toy_regressor_3_torch.py:290-295; t,out [n,4]. */
int yfs_decode_head_f32(const float *t, int64_t n, float prior_w, float prior_h, float *out, void *stream);
/* decode_xywh(raw)  practice/yolo_toy_regressor_torch_20251015.py:411-421; t,out [n,4]. */
int yfs_decode_xywh_f32(const float *t, int64_t n, float *out, void *stream);

/* Grid/anchor decode + confidence = objectness x class probability.  No reference code
 * (README.md:44-46); decode_head is its S=1, A=1 case.
 * raw [N,S,S,A,5+C]; anchors [A,2] grid units; boxes [N,S*S*A,4] xywh; score [N,S*S*A] best-class
 * score; cls [N,S*S*A] int32 best class; all_scores [N,S*S*A,C] or NULL. */
int yfs_decode_grid_f32(const float *raw, int64_t N, int S, int A, int C, const float *anchors, float *boxes,
                        float *score, int32_t *cls, float *all_scores, void *stream);
/* YOLOv1 head decode (README.md:91-93, :44-46).  pred [N,S,S,5B+C] -> boxes [N,S*S*B,4], score, cls. */
int yfs_decode_v1_f32(const float *pred, int64_t N, int S, int B, int C, float *boxes, float *score,
                      int32_t *cls, void *stream);

/* Per-image class-wise greedy NMS.  No reference code (README.md:47, :68).
 * boxes [N,A,4] xywh; score [N,A]; cls [N,A] int32; keep [N,A] int32 kept indices in descending
 * score order, -1 padded; count [N] int32.  A <= YFS_NMS_MAX_BOXES; class ids in [0, 2^21). */
#define YFS_NMS_MAX_BOXES 2048
int yfs_nms_classwise_f32(const float *boxes, const float *score, const int32_t *cls, int64_t N, int A,
                          float score_thr, float iou_thr, int32_t *keep, int32_t *count, void *stream);
/* decode_grid + NMS in one launch with the image's boxes resident in LDS (BASELINE config 4). */
int yfs_decode_nms_f32(const float *raw, int64_t N, int S, int A, int C, const float *anchors, float score_thr,
                       float iou_thr, float *boxes, float *score, int32_t *cls, int32_t *keep, int32_t *count,
                       void *stream);

/* ---- kernel set 4: YOLOv2 target assignment -------------------------------------------------- */
/* No reference code (README.md:55-59; nearest: estimate_size_priors practice/yolo_toy_regressor_3_torch.py:278-286).
 * gt [N,G,5] (cx,cy,w,h,class), rows with w<=0 or h<=0 are padding; anchors [A,2] grid units;
 * pred_boxes [N,S,S,A,4] xywh or NULL; obj_mask, ignore_mask uint8 [N,S,S,A]; targets [N,S,S,A,5]. */
int yfs_assign_targets_v2_f32(const float *gt, int64_t N, int G, int S, int A, const float *anchors,
                              const float *pred_boxes, float ignore_thr, uint8_t *obj_mask,
                              uint8_t *ignore_mask, float *targets, void *stream);

/* ---- kernel set 2: composite YOLOv1 loss ------------------------------------------------------ */
/* Slots in at loss_fn(preds, yb)  yolo_toy_regressor_torch.py:329 (nn.MSELoss() at :377); the
 * composite form is specified by assets/yolo_loss_function.png and README.md:97-115 only.
 * pred, target, grad [N,S,S,5B+C] fp32; loss_out [5] = total, coord, obj, noobj, cls (each / N);
 * grad = d total / d pred (written in the same launch; may be NULL). */
size_t yfs_yolov1_loss_workspace_bytes(int64_t N, int S);
int yfs_yolov1_loss_fwd_bwd_f32(const float *pred, const float *target, int64_t N, int S, int B, int C,
                                float l_coord, float l_noobj, float *loss_out, float *grad, void *workspace,
                                size_t workspace_bytes, void *stream);
/* The same loss for callers that cannot allocate device memory.  No counterpart in the reference: it exists
 * for code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments and
 * results (bit for bit) as yfs_yolov1_loss_fwd_bwd_f32 without the workspace: the per-block partial sums go to a
 * 1 MiB array built into the library (one copy per GPU, resolved at the first call on each device), so
 * N*S*S <= yfs_yolov1_loss_noalloc_max_cells() = 16,777,216 grid cells (YFS_ERR_UNSUPPORTED beyond).  Calls no
 * hipMalloc / hipFree.  All calls on one GPU share that array: calls that overlap on one GPU (different streams
 * or threads) race; serialise them or use the workspace form. */
int64_t yfs_yolov1_loss_noalloc_max_cells(void);
int yfs_yolov1_loss_noalloc_f32(const float *pred, const float *target, int64_t N, int S, int B, int C,
                                float l_coord, float l_noobj, float *loss_out, float *grad, void *stream);
/* nn.MSELoss()  yolo_toy_regressor_torch.py:377 (mean over all n elements) + gradient. */
size_t yfs_mse_workspace_bytes(int64_t n);
int yfs_mse_fwd_bwd_f32(const float *pred, const float *target, int64_t n, float *loss_out, float *grad,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- kernel set 1: Conv2D block = Conv2d(k, stride, pad k/2, bias) + LeakyReLU ------------------
 * Conv2D(in_ch, out_ch, k, s)  yolo_toy_regressor_torch.py:206-211, called 24x from :218-254.
 * bf16 NHWC activations, bf16 weights [Cout,K,K,Cin], fp32 accumulate on MFMA. */

/* y = leaky(conv(x, w) + bias).  x [N,H,W,Cin] bf16, w [Cout,K,K,Cin] bf16, bias [Cout] fp32,
 * y [N,Ho,Wo,Cout] bf16.  Cin % 64 == 0 and Cout % 64 == 0 (conv1's Cin=3 goes through the stem entry points
 * below, or through yfs_im2col_bf16 first). */
int yfs_conv2d_fwd_bf16(const void *x, const void *w, const float *bias, void *y, int64_t N, int H, int W,
                        int Cin, int Cout, int K, int stride, float slope, void *stream);
/* Conv + BatchNorm (inference statistics) + LeakyReLU in one launch: y = leaky(conv(x, w) * scale[c] + shift[c]).
 * The reference's block has no BatchNorm (yolo_toy_regressor_torch.py:208-211); it is the YOLOv2 step its roadmap
 * names (README.md:30, what_to_change_next.txt:83).  scale = gamma / sqrt(var + eps), shift = beta + (conv_bias -
 * mean) * scale, both [Cout] fp32, applied in fp32 before the one rounding to bf16.  scale == NULL means 1: that
 * IS yfs_conv2d_fwd_bf16 (shift = bias), bit for bit.  Inference-time folding only: there is no backward through the
 * statistics; to train through a frozen BatchNorm fold scale into the weights (w * scale[co], shift as the bias) and use
 * the ordinary forward / dgrad / wgrad entry points. */
int yfs_conv2d_fwd_affine_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int64_t N,
                               int H, int W, int Cin, int Cout, int K, int stride, float slope, void *stream);
/* dx = conv_transpose(dz, w) [* leaky'(x_act) when x_act != NULL].  dz [N,Ho,Wo,Cout] bf16,
 * wt [Cin,K,K,Cout] bf16 = w transposed and spatially flipped (yfs_weight_prep), dx [N,H,W,Cin] bf16.
 * x_act: this layer's input when it is itself a LeakyReLU output, so the previous block's
 * activation derivative is applied in the epilogue. */
int yfs_conv2d_dgrad_bf16(const void *dz, const void *wt, const void *x_act, void *dx, int64_t N, int H, int W,
                          int Cin, int Cout, int K, int stride, float slope, void *stream);
/* The same two operations with a caller-owned workspace (same reference call sites: Conv2D blocks :206-211 and their
 * autograd).  A GEMM with few pixels and a long reduction (feat.25-27, :250-254: 64 x 7 x 7 pixels, K = 9216) cannot fill
 * 256 CUs with 256-wide tiles; given yfs_conv2d_{fwd,dgrad}_workspace_bytes() bytes of scratch the library splits the
 * reduction over workgroups (fp32 partial tiles in the workspace, summed in split order: deterministic) and applies the
 * epilogue in the reduction.  The byte count is 0 for every shape that does not split; workspace == NULL always runs
 * the unsplit kernels (results differ from the split form only in fp32 summation order before the one bf16 rounding).
 * yfs_conv2d_fwd_ws_bf16 takes the folded-BN form's arguments (scale == NULL: bias-only block). */
size_t yfs_conv2d_fwd_workspace_bytes(int64_t N, int H, int W, int Cin, int Cout, int K, int stride);
size_t yfs_conv2d_dgrad_workspace_bytes(int64_t N, int H, int W, int Cin, int Cout, int K, int stride);
int yfs_conv2d_fwd_ws_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int64_t N,
                           int H, int W, int Cin, int Cout, int K, int stride, float slope, void *workspace,
                           size_t workspace_bytes, void *stream);
int yfs_conv2d_dgrad_ws_bf16(const void *dz, const void *wt, const void *x_act, void *dx, int64_t N, int H, int W,
                             int Cin, int Cout, int K, int stride, float slope, void *workspace, size_t workspace_bytes,
                             void *stream);
/* yfs_conv2d_dgrad_bf16 with x_act = y_pooled, then yfs_maxpool2_bwd_idx_bf16, in one launch: the autograd of a
 * Conv2D block whose input is a pool() output (:221-222, :224-225, :231-232, :245-246 read backwards), carried through
 * that pool.  H, W = the block's input grid = the pool's OUTPUT grid; y_pooled [N,H,W,Cin] bf16 (LeakyReLU derivative of
 * the routed element), pool_idx = the pool's record (layout of yfs_maxpool2_fwd_idx_bf16); dx_full [N,2H,2W,Cin] bf16 is
 * the gradient of the pool's INPUT: the masked value at the recorded window position, zeros at the other three.
 * Stride 1 only.  Bit-identical to the two calls it replaces; the pooled-resolution gradient is never written. */
int yfs_conv2d_dgrad_unpool_bf16(const void *dz, const void *wt, const void *y_pooled, const void *pool_idx, void *dx_full,
                                 int64_t N, int H, int W, int Cin, int Cout, int K, float slope, void *stream);
/* dw [Cout,K,K,Cin] fp32 (+)= sum_pixels dz (x) x ; db [Cout] fp32 (+)= sum_pixels dz.
 * accumulate != 0 adds into dw/db (gradient accumulation), else overwrites. */
size_t yfs_conv2d_wgrad_workspace_bytes(int64_t N, int H, int W, int Cin, int Cout, int K, int stride);
int yfs_conv2d_wgrad_bf16(const void *x, const void *dz, float *dw, float *db, int64_t N, int H, int W, int Cin,
                          int Cout, int K, int stride, int accumulate, void *workspace, size_t workspace_bytes,
                          void *stream);
/* dz = dy * (y > 0 ? 1 : slope): LeakyReLU backward on the saved block output (:210). n elements bf16. */
int yfs_leaky_bwd_bf16(const void *dy, const void *y, void *dz, int64_t n, float slope, void *stream);
/* fp32 master weight [Cout,K,K,Cin] -> bf16 copy w_bf16 (same layout) and wt_bf16 [Cin,K,K,Cout]
 * flipped (for dgrad); either output may be NULL. */
int yfs_weight_prep_bf16(const float *w, void *w_bf16, void *wt_bf16, int Cout, int K, int Cin, void *stream);
/* The same for up to 32 layers in one launch (all Conv2D blocks of feat, :218-254, once per step).  w is the
 * fp32 master weight, or its bf16 copy when w_is_bf16 != 0.  descs is a host array. */
typedef struct yfs_weight_desc {
    const void *w;
    void *w_bf16, *wt_bf16; /* either may be NULL */
    int Cout, K, Cin, w_is_bf16;
} yfs_weight_desc;
int yfs_weight_prep_batch_bf16(const yfs_weight_desc *descs, int count, void *stream);
/* im2col for the Cin=3 stem (feat.0, :220): x [N,H,W,Cin] (fp32 if x_is_f32 else bf16) ->
 * cols [N*Ho*Wo, Kpad] bf16 with (kh,kw,ci) order, zero padded to Kpad. */
int yfs_im2col_bf16(const void *x, int x_is_f32, void *cols, int64_t N, int H, int W, int Cin, int K, int stride,
                    int Kpad, void *stream);

/* Stem Conv2D(3, Cout, 7, s=2)  yolo_toy_regressor_torch.py:220 without an im2col matrix: the image is
 * rearranged once (2x2 space-to-depth of the 3-padded image, 12 -> 16 channels) and the 7x7/s2 convolution
 * becomes a 4x4/s1 one whose kernel rows are 64 contiguous elements.
 * yfs_stem_space_to_depth_bf16: x is [N,3,H,W] with arbitrary element strides (NCHW fp32 as the reference
 *   feeds it, or channels-last bf16); xs [N,Hs,Ws,16] bf16 with 2*Hs >= H+6, 2*Ws >= W+6.
 * w2 [Cout][4][64] bf16: w2[co][a][b*16 + (dy*2+dx)*3 + c] = w[co][c][2a+dy][2b+dx] (0 where 2a+dy or 2b+dx > 6).
 * dw2 [Cout][4][64] fp32 in the same arrangement. */
int yfs_stem_space_to_depth_bf16(const void *x, int x_is_f32, int64_t stride_n, int64_t stride_c, int64_t stride_h,
                                 int64_t stride_w, void *xs, int64_t N, int H, int W, int Hs, int Ws, void *stream);
int yfs_conv2d_stem_fwd_bf16(const void *xs, const void *w2, const float *bias, void *y, int64_t N, int Hs, int Ws,
                             int OH, int OW, int Cout, float slope, void *stream);
size_t yfs_conv2d_stem_wgrad_workspace_bytes(int64_t N, int OH, int OW, int Cout);
int yfs_conv2d_stem_wgrad_bf16(const void *xs, const void *dz, float *dw2, float *db, int64_t N, int Hs, int Ws, int OH,
                               int OW, int Cout, int accumulate, void *workspace, size_t workspace_bytes, void *stream);
/* Backward of [stem conv (:220) -> LeakyReLU -> MaxPool2d(2,2) (:216)] with respect to the stem's weight and bias,
 * taken from the POOLED side: dy_pooled [N,OH/2,OW/2,Cout] bf16 is the gradient of the pool's output, pool_idx
 * the argmax record the fused forward wrote (yfs_conv2d_stem_fwd_pool_bf16), y_pooled the pooled activation
 * (LeakyReLU derivative; NULL = dy_pooled already carries it).  Equals yfs_maxpool2_bwd_idx_bf16(fuse_leaky)
 * followed by yfs_conv2d_stem_wgrad_bf16 (identical bf16 dz values; only the fp32 summation order over pixels
 * differs), without writing and re-reading the full-resolution dz
 * (3/4 zeros).  Workspace: yfs_conv2d_stem_wgrad_workspace_bytes. */
int yfs_conv2d_stem_wgrad_pool_bf16(const void *xs, const void *dy_pooled, const void *pool_idx, const void *y_pooled,
                                    float *dw2, float *db, int64_t N, int Hs, int Ws, int OH, int OW, int Cout,
                                    float slope, int accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* MaxPool2d(2, 2)  yolo_toy_regressor_torch.py:216 on NHWC bf16.  H, W even. */
int yfs_maxpool2_fwd_bf16(const void *x, void *y, int64_t N, int H, int W, int C, void *stream);
/* dx = route(dy) to the first max of each window [* leaky'(x) when fuse_leaky: x is a LeakyReLU output]. */
int yfs_maxpool2_bwd_bf16(const void *x, const void *dy, void *dx, int64_t N, int H, int W, int C,
                          int fuse_leaky, float slope, void *stream);
/* The same pool (:216) for a training chain: the forward also saves the window position of every maximum
 * (idx: uint16 [N*H/2*W/2*C/8], eight 2-bit positions each, first maximum wins as in torch); the backward
 * then reads y (pooled output, the LeakyReLU mask of the routed element), idx and dy instead of x (H, W are
 * the INPUT height and width in both). */
int yfs_maxpool2_fwd_idx_bf16(const void *x, void *y, void *idx, int64_t N, int H, int W, int C, void *stream);
int yfs_maxpool2_bwd_idx_bf16(const void *y, const void *idx, const void *dy, void *dx, int64_t N, int H, int W,
                              int C, int fuse_leaky, float slope, void *stream);
/* Conv2D block followed by pool() (:220-221, :224-225, :231-232, :245-246) in one launch: bias + LeakyReLU +
 * 2x2/2 max pool in the GEMM epilogue.  Writes only the pooled activation y_pooled [N,OH/2,OW/2,Cout] and the
 * window positions pool_idx (layout of yfs_maxpool2_fwd_idx_bf16); bit-identical to yfs_conv2d_fwd_bf16 followed
 * by yfs_maxpool2_fwd_idx_bf16.  The full-resolution activation is never materialised (its only reader, the pool
 * backward, works from y_pooled and pool_idx).  OH, OW must be even. */
int yfs_conv2d_fwd_pool_bf16(const void *x, const void *w, const float *bias, void *y_pooled, void *pool_idx,
                             int64_t N, int H, int W, int Cin, int Cout, int K, int stride, float slope,
                             void *stream);
int yfs_conv2d_stem_fwd_pool_bf16(const void *xs, const void *w2, const float *bias, void *y_pooled, void *pool_idx,
                                  int64_t N, int Hs, int Ws, int OH, int OW, int Cout, float slope, void *stream);

/* Single-box losses of the reference's practice scripts (pinned by tests/golden/g6_boxloss.npz).
 * yfs_ciou_loss_f32: ciou_loss  practice/yolo_toy_regressor_3_torch.py:121-138; pred, gt [n,4] xywh -> loss [n] =
 *   clamp(1 - ciou, 0), optional iou [n], ciou [n] and grad [n,4] = d sum(loss) / d pred (alpha constant, as under the
 *   reference's torch.no_grad()).
 * yfs_single_box_loss_fwd_bwd_f32: yolo_single_box_loss  practice/yolo_toy_regressor_torch_20251015.py:368-383; raw [n,4]
 *   = (x logit, y logit, log w, log h), targets [n,4] xywh -> loss_out[0] (scalar; every smooth-L1 term a mean over the
 *   n boxes) and grad [n,4] (may be NULL).  n = number of boxes.  Deterministic two-stage reduction. */
int yfs_ciou_loss_f32(const float *pred, const float *gt, int64_t n, float *loss, float *iou, float *ciou, float *grad,
                      void *stream);
size_t yfs_single_box_loss_workspace_bytes(int64_t n);
int yfs_single_box_loss_fwd_bwd_f32(const float *raw, const float *targets, int64_t n, float *loss_out, float *grad,
                                    void *workspace, size_t workspace_bytes, void *stream);

/* Epilogue of fc.1  yolo_toy_regressor_torch.py:258-263  Linear(50176, 4096) -> LeakyReLU(0.1) -> Dropout(0.2),
 * fused around a library GEMM.
 * fwd: y[m][n] = drop(leaky(bias[n] + sum_{s<splits} partial[s][m][n])) as bf16; partial fp32 [splits][M][N] (the
 *   split-K slabs of the GEMM, summed in slab order), bias may be NULL; drop keeps an element when its Philox4x32-10
 *   uniform (key = seed, counter = element / 4) is >= p_drop and scales it by 1/(1-p_drop) (nn.Dropout training
 *   semantics; p_drop = 0 = eval mode); keep [M][N] uint8 receives the mask (required when p_drop > 0).
 * bwd: dz = dy * keep/(1-p_drop) * (y > 0 ? 1 : slope) as bf16 (y = the forward's output: a dropped or non-positive
 *   element takes the slope branch, and a dropped one is zero anyway); db[n] = (accumulate ? db[n] : 0) + sum_m dz
 *   (may be NULL). */
int yfs_fc_epilogue_fwd_bf16(const float *partial, int splits, const float *bias, void *y, void *keep, int64_t M, int N,
                             float slope, float p_drop, uint64_t seed, void *stream);
int yfs_fc_epilogue_bwd_bf16(const void *dy, const void *y, const void *keep, void *dz, float *db, int64_t M, int N,
                             float slope, float p_drop, int accumulate, void *stream);

/* clip_grad_norm_ + SGD(momentum, weight_decay)  yolo_toy_regressor_torch.py:331-335, :378-381 on flat
 * fp32 arenas (all parameters / gradients / momentum buffers live in one allocation each).
 * yfs_sumsq_f32: sumsq_out[0] = sum g^2 (deterministic two-stage reduction).
 * yfs_sgd_step_f32: clip = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) (no clipping when sumsq == NULL or
 * max_norm <= 0); d = clip*g + wd*p; buf = first_step ? d : momentum*buf + d; p -= lr*buf.  p_bf16 (may be
 * NULL): bf16 [n] shadow of the updated p, written in the same pass (the next step's conv / linear kernels
 * read it instead of casting the fp32 master weights again; reference counterpart: autocast's per-step
 * weight casts, yolo_toy_regressor_torch.py:323). */
size_t yfs_sumsq_workspace_bytes(int64_t n);
int yfs_sumsq_f32(const float *g, int64_t n, float *sumsq_out, void *workspace, size_t workspace_bytes, void *stream);
int yfs_sgd_step_f32(float *p, const float *g, float *buf, int64_t n, const float *sumsq, float max_norm,
                     float lr, float momentum, float weight_decay, int first_step, void *p_bf16, void *stream);

/* Linear layer, weight gradient fused with the optimiser step (fc.1 = nn.Linear(50176, 4096),
 * yolo_toy_regressor_torch.py:259; its backward at :330 and the update at :331-335, :378-381).
 * x [B,In] bf16 (the layer's input), dy [B,Out] bf16 (gradient of its output); the weight p [Out,In] fp32, its
 * momentum buffer m and bf16 shadow p_bf16 (may be NULL) are updated in place with
 *   g = grad_scale * sum_b dy[b][o] * x[b][i];  clip = min(1, max_norm / (sqrt(*sumsq) + 1e-6)) (1 when sumsq == NULL
 *   or max_norm <= 0);  d = clip*g + wd*p;  m = first_step ? d : momentum*m + d;  p -= lr*m
 * exactly as yfs_sgd_step_f32 would with a materialised gradient — which is never written (4 B/parameter less to
 * write, to sum and to re-read).  *sumsq must already hold the squared gradient norm of the WHOLE model including
 * this layer's own (||dy^T x||_F^2 = sum((dy dy^T) * (x x^T)), two B x B Gram matrices).  In, Out multiples of 128. */
int yfs_linear_wgrad_sgd_bf16(const void *x, const void *dy, int64_t B, int In, int Out, float *p, float *m, void *p_bf16,
                              const float *sumsq, float max_norm, float lr, float momentum, float weight_decay,
                              int first_step, float grad_scale, void *stream);

/* YOLOv1 grid targets from ground-truth boxes: the step between the reference's data path (largest mask box ->
 * box_to_xywh_norm, yolo_toy_regressor_torch.py:98-119, :184-203) and the composite loss (SPEC.md 1; the grid
 * encoding itself has no reference code, README.md:90-93 gives the convention).  gt [N,G,5] = (xc, yc, w, h, class)
 * image-normalised, w <= 0 = padding.  target [N,S,S,5B+C] is written completely: zeros, and for every cell the
 * LOWEST-index gt whose centre falls into it: box slot 0 = (xc*S - col, yc*S - row, w, h, 1), one-hot class. */
int yfs_encode_targets_v1_f32(const float *gt, int64_t N, int G, int S, int B, int C, float *target, void *stream);

/* Segmentation masks -> boxes (the reference's data path: mask_bytes_to_boxes, yolo_toy_regressor_torch.py:98-111 =
 * skimage label() with full connectivity + regionprops().bbox in label order; the largest-box rule of
 * get_samples_from_dataset :195-196; box_to_xywh_norm :113-119).  The PNG decode (:99-100) stays with the caller.
 * mask [N,H,W,CH] u8, a pixel is set when any of its CH channels is non-zero (:101-104).  Per image:
 *   count [N]               number of 8-connected components
 *   boxes [N,max_boxes,4]   (x_min, y_min, x_max, y_max), inclusive pixel coordinates, in label order = raster order of
 *                           each component's first pixel; components past max_boxes are counted but not stored, rows
 *                           past count are left untouched (boxes may be NULL when max_boxes == 0)
 *   best_box [N,4]          the box with the largest inclusive area among ALL components, the first one on ties
 *                           (np.argmax); -1 when the mask is empty (the reference skips such samples, :193-194)
 *   best_xywh [N,4]         its [xc/W, yc/H, w/W, h/H] computed in fp64 and rounded once to fp32 as :113-119 does
 *                           (zeros for an empty mask); dims [N,2] = (H_i, W_i) int32 replaces (H, W) in that
 *                           normalisation for masks that were zero-padded to a common H x W (NULL: H, W)
 * best_box / best_xywh may be NULL.  H*W <= 2^30.  workspace: yfs_mask_to_boxes_workspace_bytes(N, H, W) bytes of
 * device memory (about 20 B per pixel), 16-byte aligned like boxes / best_box / best_xywh. */
size_t yfs_mask_to_boxes_workspace_bytes(int64_t N, int H, int W);
int yfs_mask_to_boxes_u8(const uint8_t *mask, int64_t N, int H, int W, int CH, const int32_t *dims, int32_t *boxes,
                         int max_boxes, int32_t *count, int32_t *best_box, float *best_xywh, void *workspace,
                         size_t workspace_bytes, void *stream);

/* k-means anchor clustering, assignment step (reference: README.md:56 names the feature; nearest code is the k=1
 * median prior estimate_size_priors, practice/yolo_toy_regressor_3_torch.py:278-286).  wh [n,2], anchors [k,2] in
 * the same units, k <= 64 -> assign [n] = argmax_j shape-IoU (ties -> lowest j), best_iou [n] (may be NULL). */
int yfs_anchor_assign_f32(const float *wh, int64_t n, const float *anchors, int k, int32_t *assign, float *best_iou,
                          void *stream);

/* Measurement aid, no reference counterpart: enqueue a register-only bf16 MFMA loop (blocks x 4 waves x iters x
 * 16 MFMAs of 16x16x32) and return the FLOPs it performs in *flop_out; the caller times it.  scratch: >= 4 bytes of
 * device memory (never written in practice).  bench.py prints the resulting rate next to the vendor peak. */
int yfs_probe_mfma_bf16(int iters, int blocks, double *flop_out, void *scratch, void *stream);
/* The same loop with operands that change from one MFMA to the next (toggling != 0: eight pseudo-random fragments per
 * side; 0: all ones) on the 16x16x32 (shape 0) or 32x32x16 (shape 1) instruction: what the matrix pipe alone sustains
 * when its inputs switch the way real data does (tools/probes/clock_under_load.py). */
int yfs_probe_mfma_bf16_ex(int shape, int toggling, int iters, int blocks, double *flop_out, void *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YFS_H */
