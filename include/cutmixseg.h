/*
 * cutmixseg.h -- C ABI of libcutmixseg_hip.so: the MI355X (gfx950) kernels behind the CutMix mean-teacher training
 * step of Britefury/cutmix-semisup-seg (train_seg_semisup_mask_mt.py).
 *
 * The reference has no FFI: its boundary is the Python API. Each entry point below names the
 * reference code (file:line, relative to the upstream repository root) whose device work it replaces; the Python
 * mirror of the reference interface (cutmix-semisup-seg_amd/) binds these with ctypes -- see INTEGRATION.md for
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless its name ends in `_host`
 *   - the caller owns every buffer; the library allocates nothing, never synchronises and enqueues all work on
 *     `stream` (a hipStream_t passed as void*); calls are capturable in a hipGraph
 *   - tensors are dense NCHW unless stated otherwise
 *   - returns 0 on success or a negative CMS_E* code; never throws; cms_last_error() gives a thread-local message
 *   - `dtype` arguments: CMS_F32 or CMS_BF16
 *   - low-resolution logits + (H, W, align_corners) describe an implicit bilinear upsample that is evaluated inside
 *     the kernel (full-resolution logits are never materialised); pass h == H and w == W for plain logits
 */
#ifndef CUTMIXSEG_H
#define CUTMIXSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMS_VERSION 100

#define CMS_OK 0
#define CMS_EINVAL (-1)     /* bad argument */
#define CMS_ELAUNCH (-2)    /* kernel launch / HIP runtime error */
#define CMS_EUNSUPPORTED (-3)

#define CMS_F32 0
#define CMS_BF16 1

#define CMS_LABEL_U8 0
#define CMS_LABEL_I64 1

/* cons_loss_fn, train_seg_semisup_mask_mt.py:428-446 */
#define CMS_LOSS_VAR 0
#define CMS_LOSS_LOGITS_VAR 1
#define CMS_LOSS_LOGITS_SMOOTHL1 2
#define CMS_LOSS_BCE 3
#define CMS_LOSS_KLD 4

/* mask_mode, train_seg_semisup_mask_mt.py:27-33 */
#define CMS_MODE_MIX 0
#define CMS_MODE_CUT 1

int cms_version(void);
const char* cms_last_error(void);
/* number of compute units / name of the current device (device-props cache) */
int cms_device_info(int* n_cu, char* name_out, size_t name_cap);

/* ------------------------------------------------------------------------------------------------------------
 * Box masks + CutMix paste      (mask_gen.py:110-120 ; train_seg_semisup_mask_mt.py:346-351, 363, 385-389)
 *
 * `ranges`: int32 (N, n_boxes, 4) = [y0, y1, x0, x1] half-open, i.e. the reference's float rectangles after its
 * `int(y0):int(y1)` slicing rules (done on the host by the Python side). Boxes XOR; `invert` != 0 means box = 1.
 * ------------------------------------------------------------------------------------------------------------ */

/* mask_out f32 (N,1,H,W): what BoxMaskGenerator.generate_params returns / torch_masks_from_params passes through */
int cms_boxmask_rasterize(const int32_t* ranges, int n, int n_boxes, int h, int w, int invert,
                          float* mask_out, void* stream);

/* out[n,c,y,x] = m ? x1 : x0, with m rasterised in-kernel from `ranges` (mask never touches HBM).
 * x0 == NULL means zeros (cut mode: x * m, :389). Exact for m in {0,1} (x0*(1-m) + x1*m, :350). */
int cms_cutmix_paste(const void* x0, const void* x1, void* out, int dtype, const int32_t* ranges, int n,
                     int n_boxes, int c, int h, int w, int invert, void* stream);

/* same with a materialised f32 mask (N,1,H,W), computed arithmetically as x0*(1-m) + x1*m (any mask values) */
int cms_cutmix_paste_mask(const void* x0, const void* x1, void* out, int dtype, const float* mask, int n, int c,
                          int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Masked consistency loss, fused: bilinear upsample + teacher-logit paste + softmax + confidence + loss
 *                               (train_seg_semisup_mask_mt.py:363-367, 407-459)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct cms_consistency_desc {
    const float* l_stu;    /* (N,C,h,w) student logits (of the pasted / cut image)                          */
    const float* l_tea0;   /* (N,C,h,w) teacher logits of image 0                                            */
    const float* l_tea1;   /* (N,C,h,w) teacher logits of image 1 (mix mode) or NULL (cut mode)              */
    const int32_t* ranges; /* (N,n_boxes,4) box ranges, or NULL when `mask` is given                         */
    const float* mask;     /* (N,1,H,W) materialised box mask, or NULL when `ranges` is given                */
    const float* um0;      /* (N,1,H,W) validity mask of image 0, NULL = all ones                            */
    const float* um1;      /* (N,1,H,W) validity mask of image 1, NULL = all ones (ignored in cut mode)      */
    int n, c, h, w;        /* logits geometry                                                                */
    int H, W;              /* loss geometry (crop size); logits are upsampled h x w -> H x W in-kernel       */
    int align_corners;     /* 1: deeplab2.py:204 ; 0: deeplab3plus.py:77                                     */
    int n_boxes, invert;
    int mode;              /* CMS_MODE_MIX / CMS_MODE_CUT                                                    */
    int loss_fn;           /* CMS_LOSS_*                                                                     */
    float conf_thresh;     /* <= 0 disables confidence thresholding (:408)                                   */
    int conf_per_pixel;    /* --conf_per_pixel (:415)                                                        */
} cms_consistency_desc;

/* bytes of scratch needed by cms_consistency_fwd (per-workgroup partial sums) */
size_t cms_consistency_workspace_bytes(const cms_consistency_desc* d);

/* stats_out: double[4] = { sum(loss*um), sum(loss*um*conf), count(conf >= thresh), P = N*H*W } (device).
 * Deterministic (fixed-order second-stage reduction). Under data parallelism the caller all-reduces
 * stats[2..3] before cms_consistency_finalize. */
int cms_consistency_fwd(const cms_consistency_desc* d, void* workspace, double* stats_out, void* stream);

/* scalars_out: float[4] = { consistency_loss (the value the reference logs, :461), conf_rate (:413, NaN if
 * disabled), grad_scale (d unsup_loss / d per-pixel masked loss), unsup_loss (:458) } (device).
 * `stats_global` = stats used for the confidence rate (== stats_local on one GPU). No host sync. */
int cms_consistency_finalize(const double* stats_local, const double* stats_global, float conf_thresh,
                             int conf_per_pixel, float ramp_val, float cons_weight, float* scalars_out,
                             void* stream);

/* grad_l_stu f32 (N,C,h,w) += d unsup_loss / d l_stu (caller zero-fills when it wants `=`); reads scalars[2] */
int cms_consistency_bwd(const cms_consistency_desc* d, const float* scalars, float* grad_l_stu, void* stream);

/* (round 6) Forward AND backward in one launch (train_seg_semisup_mask_mt.py:363-367, 407-459 and their autograd twins). The
 * backward pass needs ONE scalar of the forward pass -- the confidence rate of the default mode (:415-418) -- and is linear in
 * it: this launch writes stats_out (as cms_consistency_fwd) and ADDS to grad_l_stu the gradient with that scalar left out,
 *     grad_unit * um * [conf >= thresh, --conf_per_pixel only] * d per-pixel loss / d l_stu,   grad_unit = ramp * weight / (N*H*W);
 * behind cms_consistency_finalize the caller multiplies the rows by the rate (default mode with a threshold only):
 *     cms_scale_by_scalar(grad_l_stu, n*c*h*w, scalars, 1, 1.0f, stream).
 * So grad_l_stu must hold nothing else yet (zero-filled rows). cms_consistency_fused_supported: 1 when the launch exists for
 * the descriptor (an upsampling geometry whose tile rectangles fit the LDS, not the deterministic mode); the workspace is
 * cms_consistency_workspace_bytes as before. CMS_LOSS_FUSED=0 in the environment switches both fused launches off. */
int cms_consistency_fused_supported(const cms_consistency_desc* d);
int cms_consistency_fwd_bwd(const cms_consistency_desc* d, float grad_unit, void* workspace, double* stats_out,
                            float* grad_l_stu, void* stream);
/* x[i] *= scalars[index] * factor, i < n: the deferred scalar factor of the fused loss launches (a device scalar, no host sync) */
int cms_scale_by_scalar(float* x, long long n, const float* scalars, int index, float factor, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Supervised cross entropy, fused: bilinear upsample + log-softmax + NLL(ignore_index)
 *                               (nn.CrossEntropyLoss(ignore_index=255), train_seg_semisup_mask_mt.py:126, 299-301)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct cms_ce_desc {
    const float* logits;  /* (N,C,h,w) */
    const void* labels;   /* (N,H,W) uint8 or int64 */
    int label_dtype;      /* CMS_LABEL_U8 / CMS_LABEL_I64 */
    int ignore_index;
    int n, c, h, w, H, W, align_corners;
} cms_ce_desc;

size_t cms_ce_workspace_bytes(const cms_ce_desc* d);
/* stats_out: double[2] = { sum over valid pixels of -log p[label], number of valid pixels } */
int cms_ce_fwd(const cms_ce_desc* d, void* workspace, double* stats_out, void* stream);
/* scalars_out: float[2] = { loss = sum / count, grad_scale = loss_weight / count } from (possibly all-reduced)
 * stats */
int cms_ce_finalize(const double* stats, float loss_weight, float* scalars_out, void* stream);
int cms_ce_bwd(const cms_ce_desc* d, const float* scalars, float* grad_logits, void* stream);
/* (round 6) forward AND backward in one launch (autograd of :126, 299-301): stats_out as cms_ce_fwd; grad_logits += (softmax -
 * onehot) of every valid pixel through the adjoint of the upsample, WITHOUT the factor loss_weight / count -- behind
 * cms_ce_finalize: cms_scale_by_scalar(grad_logits, n*c*h*w, scalars, 1, 1.0f, stream). Rows must be zero on entry. */
int cms_ce_fused_supported(const cms_ce_desc* d);
int cms_ce_fwd_bwd(const cms_ce_desc* d, void* workspace, double* stats_out, float* grad_logits, void* stream);
/* The backward kernels of both losses scatter through tiles that share low-resolution cells with their neighbours. on = 1:
 * the tiles are issued as colour classes that never share a cell (four launches, run-to-run reproducible gradients -- what
 * `--deterministic` asks for); 0 (default; CMS_LOSS_DETERMINISTIC=1 in the environment flips it): one launch, fp32 atomics in a
 * run-dependent order (~1e-7), like the weight gradients' default. */
int cms_loss_set_deterministic(int on);

/* ------------------------------------------------------------------------------------------------------------
 * Bilinear upsample (F.interpolate(mode='bilinear'), architectures/deeplab2.py:204, deeplab3plus.py:54-55,77)
 * Stand-alone form for the `forward(x) -> (N,C,H,W)` contract of the reference models.
 * ------------------------------------------------------------------------------------------------------------ */
int cms_upsample_bilinear_fwd(const float* lo, float* hi, int n, int c, int h, int w, int H, int W,
                              int align_corners, void* stream);
/* grad_lo (N,C,h,w) = adjoint applied to grad_hi (overwrites); deterministic gather */
int cms_upsample_bilinear_bwd(const float* grad_hi, float* grad_lo, int n, int c, int h, int w, int H, int W,
                              int align_corners, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Teacher EMA and fused optimizer + EMA over flat fp32 arenas
 *                               (optim_weight_ema.py:21-25 ; torch.optim.Adam/SGD at
 *                                train_seg_semisup_mask_mt.py:90-100, 465-467)
 * ------------------------------------------------------------------------------------------------------------ */

/* tgt = tgt*alpha + src*one_minus_alpha with the reference's three fp32 roundings (no FMA). Optional bf16 copy
 * of the new target. */
int cms_ema_flat(float* tgt, const float* src, size_t count, float alpha, float one_minus_alpha,
                 uint16_t* tgt_bf16_out, void* stream);

/* One segment of the parameter arena (a state_dict tensor). `k_updates` = how many times the tensor appears in
 * the optimizer's parameter list (deeplab2.py:208-230 yields backbone weights 3x/4x): the update is applied k
 * times in sequence with the same gradient; 0 = not optimised (BN tensors, or tensors that never receive a
 * gradient such as the ASPP d18/d24 branches) -> only the EMA is applied. */
typedef struct cms_param_segment {
    uint64_t offset;     /* first element in the arenas */
    uint64_t count;      /* number of elements */
    int32_t k_updates;
    int32_t lr_group;    /* index into the `lrs` array */
} cms_param_segment;

typedef struct cms_optim_desc {
    float* param;              /* student fp32 master arena */
    const float* grad;         /* gradient arena (same indexing) */
    float* slot0;              /* Adam exp_avg / SGD momentum buffer */
    float* slot1;              /* Adam exp_avg_sq (unused for SGD) */
    float* ema_param;          /* teacher fp32 arena, or NULL (pi model) */
    uint16_t* param_bf16;      /* optional bf16 copy of the updated student arena */
    uint16_t* ema_bf16;        /* optional bf16 copy of the updated teacher arena */
    const cms_param_segment* segments; /* DEVICE array */
    const uint32_t* chunk_seg; /* DEVICE array: segment index of every CMS_OPT_CHUNK-element chunk of work */
    const uint32_t* chunk_off; /* DEVICE array: element offset of the chunk inside its segment */
    uint32_t n_chunks;
    const double* lrs;         /* DEVICE array: current learning rate per group (double, as Python holds it) */
    const int64_t* step_count; /* DEVICE scalar: number of optimizer steps already taken */
    float grad_scale;          /* gradients are multiplied by this first (1/world_size after a sum all-reduce) */
    float ema_alpha;
    float ema_one_minus_alpha; /* (float)(1.0 - (double)alpha), as the reference forms it */
    /* Adam (doubles: torch forms 1-beta and the bias corrections in Python double before the fp32 ops) */
    double beta1, beta2, eps;
    /* SGD */
    float momentum, weight_decay;
    int nesterov;
} cms_optim_desc;

#define CMS_OPT_CHUNK 2048

int cms_adam_ema_step(const cms_optim_desc* d, void* stream);
int cms_sgd_ema_step(const cms_optim_desc* d, void* stream);
/* *counter += 1 on the device (keeps the optimizer step counter graph-replayable) */
int cms_increment_counter(int64_t* counter, void* stream);
/* Frozen BatchNorm of ALL layers of a network folded into the convolution epilogues' affine in one launch
 * (architectures/deeplab2.py:92-107, BatchNorm in eval mode: y = (x - mean) / sqrt(var + eps) * weight + bias):
 *   scale[i] = flat[idx_weight[i]] * rsqrt(flat[idx_var[i]] + eps),  bias[i] = flat[idx_bias[i]] - flat[idx_mean[i]] * scale[i]
 * `flat`: the fp32 parameter arena; idx_*: n element indices into it (int64, device); every product / difference rounded
 * separately, like the tensor expression of the reference's modules. */
int cms_bn_fold(const float* flat, const int64_t* idx_weight, const int64_t* idx_bias, const int64_t* idx_mean,
                const int64_t* idx_var, int n, float eps, float* scale, float* bias, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Evaluation: fused upsample + argmax + confusion matrix   (train_seg_semisup_mask_mt.py:510-514 ; evaluation.py)
 * I = diag(cm), U = rowsum + colsum - diag so one CxC histogram is all that is needed.
 * ------------------------------------------------------------------------------------------------------------ */
/* cm int64 (C,C) += histogram of (truth, argmax) over pixels with truth != ignore_index (ignore_index < 0: none).
 * pred_out (N,H,W) uint8 optional. */
int cms_argmax_confusion(const float* logits, const void* labels, int label_dtype, int ignore_index, int n, int c,
                         int h, int w, int H, int W, int align_corners, int64_t* cm, uint8_t* pred_out,
                         void* stream);
/* fast_cm / per_class_i_and_u_cm on integer maps (evaluation.py:6-37): truth, pred uint8 (count) */
int cms_confusion(const uint8_t* truth, const uint8_t* pred, size_t count, int ignore_index, int c, int64_t* cm,
                  void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * MFMA implicit-GEMM convolution for the backbone   (nn.Conv2d + frozen nn.BatchNorm2d + ReLU (+ residual) of
 *                               architectures/deeplab2.py:89-109, 124-128, and their backward passes)
 * Activations bf16 NHWC, weights bf16 [tap][Cout][Cin], fp32 accumulation.
 * ------------------------------------------------------------------------------------------------------------ */
#define CMS_CONV_MAX_TAPS 18
#define CMS_CONV_FWD 0     /* y  = relu(acc * scale[co] + bias[co] + res)                     */
#define CMS_CONV_DGRAD 1   /* dx = (acc + res) * [mask_src > 0]                               */

typedef struct cms_conv_desc {
    const void* x;         /* bf16 [N][H][W][Cin]                                                                */
    const void* w;         /* bf16 [ntaps][Cout][Cin]                                                            */
    void* y;               /* bf16 [N][out_h][out_w][Cout] or NULL                                               */
    float* y32;            /* fp32 NCHW [N][cout_real][Ho][Wo] or NULL (ASPP head logits)                         */
    const float* scale;    /* [Cout] or NULL: frozen-BN gamma/sqrt(var+eps)                                       */
    const float* bias;     /* [Cout] or NULL: frozen-BN beta - mean*scale, or the conv bias                       */
    const void* res;       /* bf16, indexed like y: residual (forward) / gradient to add (dgrad), or NULL         */
    const void* mask_src;  /* bf16, indexed like y: dgrad output is zeroed where this activation is <= 0, or NULL */
    int n, h, w_in, cin;   /* input geometry                                                                     */
    int ho, wo, cout;      /* GEMM pixel grid (N*ho*wo rows) and channel count (multiple of 32)                   */
    int cout_real;         /* channels actually written to y32 (<= cout); must equal cout for the bf16 output     */
    int ntaps;             /* kernel taps; tap t reads input pixel (oy*stride + tap_dy[t], ox*stride + tap_dx[t]) */
    int tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
    int stride;            /* input gather stride                                                                */
    int out_h, out_w, out_stride; /* output tensor geometry; pixel (oy,ox) is written at (oy*out_stride, ...)      */
    int relu;              /* forward epilogue ReLU                                                              */
    int mode;              /* CMS_CONV_FWD / CMS_CONV_DGRAD                                                      */
    int tile;              /* 0 = auto, else channels per workgroup: 128 / 64 / 32                               */
    int ksplit;            /* <= 1: off; else the taps are split over workgroups (fp32 y32 output only, which must
                              be zero-filled: partial sums are accumulated with atomics; bias added once)        */
    const void* zeros;     /* a run of zero bytes in device memory, at least 2 * cin + 128 long: enables the
                              direct-to-LDS loader (padded / out-of-range rows are fetched from it, advancing through
                              it like a live pixel's channel run); NULL = register-staged loader                 */
    int variant;           /* 0 = auto (direct-to-LDS, 1 stage), 1 = register-staged loader, 4 = direct-to-LDS with
                              two stages, 5 = two stages of 32 K-elements; 2 / 3 = ablation switches (no MFMA / no
                              loads) of variant 0, 6 / 7 = the same of variant 4 (tools/conv_ablate*.py); 10..14 = stage
                              rings, 20..25 = K rotation / staggered starts, 30 = cycle trace (all measured, none
                              faster: DESIGN.md section 4.1); 90 = the eight-phase 256 x 256 kernel, one whole tile
                              per workgroup; 91 = the same, persistent with a stream-K round (needs `workspace`);
                              99 = never the eight-phase kernel                                                 */
    int zeros_bytes;       /* length of the `zeros` run (checked against 2 * cin + 128)                          */
    void* workspace;       /* optional scratch of the eight-phase 256 x 256 kernel (csrc/conv8.hip; variant 90 / 91 or
                              the automatic choice for Cout % 256 == 0 layers with >= 8 K tiles), at least
                              cms_conv_igemm_workspace_bytes() long: 64 KB of arrival counters -- ZERO before the first
                              launch that uses the buffer, left zero by every launch -- followed by the fp32 slabs of
                              the stream-K round (tiles whose K loop is cut across workgroups so that the launch fills
                              every CU; the last arriver of a tile adds the pieces in a fixed order). Launches that may
                              overlap (different streams) need different workspaces. NULL: whole tiles only.        */
    long long workspace_bytes;
    /* ReLU masks as BITS (round 4). A forward launch with `relu` can also write, per output pixel, Cout / 8 bytes whose bit
     * c & 7 of byte c >> 3 says [y[pixel][c] > 0] (of the stored bf16 value): mask_bits_out = uint8 [N][out_h][out_w][cout / 8].
     * The data gradient that would re-read that activation only for its sign (`mask_src`) takes the bits instead
     * (`mask_bits`, mask_src NULL): 1/16 of the bytes -- 69 MB less per layer-3 expansion of BASELINE configs[1] -- and the same
     * result bit for bit. Both kernels of cms_conv_igemm write and read the same layout (round 5: the eight-phase kernel too,
     * whole-tile launches).                                                                                                  */
    uint8_t* mask_bits_out;
    const uint8_t* mask_bits;
    /* Round 5: BatchNorm statistics out of the convolution epilogue (the batch-statistics units of
     * architectures/deeplab2.py:72-84 -- the reference CLI's default, train_seg_semisup_mask_mt.py:587 -- are
     * u = conv(x); y = relu(bn_batch(u) (+ res)): the statistics pass over u, one of the unit's memory passes, disappears).
     * stats_out != NULL (bf16 NHWC output, mode 0): every workgroup also writes the per-channel (sum, sum of squares) of the bf16
     * values it STORES, per pixel tile: float [tiles][2 slots][2][cout], tiles = ceil(n * ho * wo / T), T =
     * cms_conv_igemm_stats_tile_rows(desc) (128 or 256 rows: the kernel that runs the launch; 0 = this launch cannot, use
     * cms_bn_stats). The pixel rows are `groups` equal runs of stats_rows_per_group rows (sample groups normalised separately);
     * slot 0 = the tile's rows of the group its FIRST row belongs to, slot 1 = its rows of the next group (written only by a tile that
     * straddles a boundary). cms_bn_finalize_tiles turns them into mean / rstd / scale / shift and moves the running statistics. */
    float* stats_out;
    int stats_rows_per_group;
    /* Round 5, data gradients with BOTH res and mask_bits: != 0 -> the bits gate the residual only, y = acc + (bit ? res : 0)
     * (the masked gradient of an identity shortcut added to a convolution's data gradient: the batch-statistics bottleneck's
     * `dres` tensor is never written, architectures/deeplab2.py:105-107); 0 -> y = bit ? acc + res : 0 as before.           */
    int mask_gates_res;
    /* Round 5, data gradients (mode 1) with stats_out: the launch writes the gradient dy of the OUTPUT of a batch-statistics unit
     * y = relu(bn(u) (+ res)); with that unit's u (bf16, indexed like this launch's output), ReLU mask bits (NULL: no ReLU) and
     * statistics mean / rstd [groups][cout] it also leaves per-tile (sum d, sum d * xhat), d = bit ? dy : 0, xhat = (u - mean) * rstd --
     * cms_bn_bwd_sums_tiles adds them into the sums cms_bn_bwd_apply_groups takes: the unit's backward reduction over u, dy and the
     * mask (cms_bn_reduce_ws mode 1) is not launched. The stored output is unchanged (unmasked dy).                              */
    const void* bstats_u;
    const uint8_t* bstats_bits;
    const float* bstats_mean;
    const float* bstats_rstd;
} cms_conv_desc;

int cms_conv_igemm(const cms_conv_desc* d, void* stream);
/* Which kernel cms_conv_igemm runs this descriptor on (measurement tooling: algorithmic bytes per KERNEL beside the PMC
 * counters; the choice depends on the geometry and on the CMS_CONV8 / CMS_CONV_MIXED switches of the process). */
#define CMS_ROUTE_OTHER 0     /* an explicit variant / tile request */
#define CMS_ROUTE_TILE128 1   /* conv_igemm_kernel, 128 channels x 128 pixels */
#define CMS_ROUTE_MIXED 2     /* conv_igemm_mixed_kernel: the balanced 128 x 128 launch */
#define CMS_ROUTE_TILE64 3    /* 64-channel tile */
#define CMS_ROUTE_TILE32 4    /* 32-channel tile */
#define CMS_ROUTE_CONV8 8     /* conv8_kernel: eight-phase 256 x 256 */
int cms_conv_igemm_route(const cms_conv_desc* d);
/* pixel rows per statistics tile of a launch with stats_out (see cms_conv_desc), or 0 when the launch cannot write them */
int cms_conv_igemm_stats_tile_rows(const cms_conv_desc* d);
/* bytes of cms_conv_desc.workspace that every launch on this device is satisfied with */
long long cms_conv_igemm_workspace_bytes(void);

/* Diagnostic (tools/conv_trace.py): launches with variant 30 stamp s_memtime at every phase of the K loop of wave 0
 * (own loads landed / barrier / MFMAs issued / barrier / next stage issued) into `buf`: 512 dwords per workgroup for
 * the first `workgroups` workgroups -- dwords 0..12 = HW_ID, XCC_ID, s_memrealtime at start (lo, hi), cycles since
 * start at: prologue done, K loop done, epilogue arithmetic done, stores acknowledged; K steps; tile_m; tile_n;
 * s_memtime at start (lo, hi); from dword 16 on six stamps per K step. While a buffer is set, cms_conv_wgrad launches
 * stamp their stages the same way (tools/wgrad_trace.py). NULL switches it off. Not part of the data path. */
int cms_conv_set_trace(void* buf, int workgroups);

/* dst[tap'][ci][co] = bf16(src[tap][co][ci] * scale[co]) with tap' = ntaps-1-tap when flip != 0: the operand of
 * the dgrad pass (which is cms_conv_igemm on the transposed, tap-flipped, BN-scale-folded weights). */
int cms_conv_pack_transpose(const void* src, int src_dtype, void* dst_bf16, const float* scale, int ntaps, int cout,
                            int cin, int flip, void* stream);

/* The same for many weight tensors in ONE launch. `items_dev` is a device-resident table (the library reads it on the
 * device); item i owns blocks [first_block, first_block + ntaps * ceil(cout/32) * ceil(cin/32)), first_block ascending
 * from 0; total_blocks = the sum. No tap flip. (backbone_hip.py re-packs all dgrad operands after an optimizer step.) */
typedef struct cms_pack_item {
    const void* src;       /* (ntaps, cout, cin) fp32 or bf16 (all items the same dtype)                             */
    void* dst;             /* (ntaps, cin, cout) bf16                                                                 */
    const float* scale;    /* per output channel (cout) or NULL                                                       */
    int ntaps, cout, cin;
    int first_block;
} cms_pack_item;

int cms_conv_pack_transpose_batch(const cms_pack_item* items_dev, int n_items, int total_blocks, int src_dtype,
                                  void* stream);
/* (round 6) The same for bf16 sources whose cout AND cin are multiples of 64 (every body convolution of the DeepLab networks):
 * 64 x 64 tiles, 16-byte accesses on both sides; item i owns blocks [first_block, first_block + ntaps * (cout/64) * (cin/64)). */
int cms_conv_pack_transpose_batch64(const cms_pack_item* items_dev, int n_items, int total_blocks, void* stream);

/* dW[tap][co][ci] (fp32) += scale[co] * sum over pixels of dU[pix][co] * X[pix shifted by tap][ci]; K = pixels,
 * split across workgroups and accumulated with atomics (zero the gradient buffer once per step). */
typedef struct cms_wgrad_desc {
    const void* du;        /* bf16 [N][ho][wo][cout]: gradient wrt the conv+BN output (pre-activation)            */
    const void* x;         /* bf16 [N][h][w_in][cin]: the conv's input                                             */
    float* dw;             /* fp32 [ntaps][cout][cin]                                                              */
    const float* scale;    /* [cout] frozen-BN scale or NULL                                                       */
    int n, h, w_in, cin, ho, wo, cout;
    int cout_real;         /* rows >= cout_real are skipped (padded class axis); 0 = cout                          */
    int ntaps;
    int tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
    int stride;
    int ksplit;            /* 0 = auto                                                                             */
    /* Optional side outputs for a TRAINABLE BatchNorm affine behind the convolution (frozen statistics; torchvision
     * style backbones, architectures/deeplab3plus.py:89-98): with G = the unscaled weight gradient sum_p dU x,
     *   wdot[co]  += <W[.][co][.], G[.][co][.]>  (over taps and input channels; W = the bf16 forward operand)
     *   dbeta[co] += sum over pixels of dU[pix][co]
     * from which  d(bias) = dbeta  and  d(weight) = (wdot - running_mean * dbeta) / sqrt(running_var + eps)  --
     * exact, and no division by the BatchNorm weight. All three NULL = off.                                          */
    const void* w;         /* bf16 [ntaps][cout][cin]                                                              */
    float* wdot;           /* fp32 [cout], accumulated with atomics                                                */
    float* dbeta;          /* fp32 [cout], accumulated with atomics                                                */
    int dw_cout;           /* rows per tap of the dw TENSOR when it is narrower than the GEMM's (padded) cout: the ASPP
                              head computes 64 padded class rows and writes the cout_real live ones straight into the
                              (9, C, 2048) gradient tensor; 0 = cout                                               */
    void* workspace;       /* optional scratch for the split-K partial sums (bf16 entry point), at least
                              cms_conv_wgrad_workspace_bytes(d) long: the pixel slices then write their tiles there with
                              plain stores and a second launch on the same stream adds them to dw IN SLICE ORDER --
                              run-to-run deterministic weight gradients (train_seg_semisup_mask_mt.py:459,465 on the
                              reference's CPU path are deterministic too). NULL or too small: fp32 atomics, whose order
                              varies from run to run (6 % slower alone, 1.5-2 % FASTER inside the two-stream step: the
                              throughput default). The scratch must not be shared by launches that may overlap.       */
    long long workspace_bytes;
    int wg_target;         /* eight-phase kernel only: workgroups (= CUs, one each) this launch should aim at; 0 = the whole
                              machine (a launch that runs alone). A caller that runs weight gradients BESIDE other work says
                              how many CUs are theirs: the DeepLab v2 executor passes 56 per launch on two streams next to
                              the data-gradient chain                                                 */
} cms_wgrad_desc;

int cms_conv_wgrad(const cms_wgrad_desc* d, void* stream);
/* Kernel selection of cms_conv_wgrad (diagnostic / A-B switch; production leaves it alone). Launches with Cout % 256 == 0,
 * Cin % 256 == 0, no BatchNorm-affine side outputs and >= 1024 pixels take the eight-phase 256 x 256 kernel of csrc/wgrad8.hip
 * (few pixel slices, 40-170 K tiles per workgroup), everything else the 128 x 128 kernel of csrc/conv.hip.
 *   mode -1 = the environment decides (CMS_WGRAD8, default 1), 0 = never the eight-phase kernel, 1 = wherever supported. */
int cms_conv_set_wgrad8(int mode);
int cms_conv_wgrad_uses_wgrad8(const cms_wgrad_desc* d);   /* 1: cms_conv_wgrad(d) would take the eight-phase kernel now */

/* GROUPED weight gradients (round 4): the launches of MANY layers -- the bottlenecks of a stretch of the backward pass, autograd
 * of architectures/deeplab2.py:89-109 -- as ONE grid. A per-layer launch has 16 ... 144 output tiles and needs 8 ... 21 pixel
 * slices to occupy the machine; each slice workgroup then runs ~25 pixel stages per 64 KB atomic epilogue. In a group the tiles
 * of all layers fill the machine together (2-3 slices, ~200 stages per epilogue, 3-4 workgroups per CU).
 *   cms_conv_wgrad_group_kind   0 = this launch cannot join a group (channel counts not multiples of 128, side outputs, a
 *                               workspace, padded class axis), 1 = group of pointwise launches, 2 = group of launches with taps
 *   cms_conv_wgrad_group_bytes  size of the item table for n launches
 *   cms_conv_wgrad_group_pack   fills the HOST image of the table for n launches of ONE kind; the caller copies it to the device
 *                               (once per recorded pass) -- target_workgroups: 0 = default; -> grid size in *total_blocks
 *   cms_conv_wgrad_group_run    the launch: table_dev = the device copy                                                        */
int cms_conv_wgrad_group_kind(const cms_wgrad_desc* d);
long long cms_conv_wgrad_group_bytes(int n_items);
int cms_conv_wgrad_group_pack(const cms_wgrad_desc* descs, int n, int target_workgroups, void* host_table, long long bytes,
                              int* total_blocks);
int cms_conv_wgrad_group_run(const void* table_dev, int n_items, int total_blocks, int kind, void* stream);
/* bytes of `workspace` that make the launch deterministic (0: it has a single pixel slice and already is) */
long long cms_conv_wgrad_workspace_bytes(const cms_wgrad_desc* d);

/* ------------------------------------------------------------------------------------------------------------
 * The same convolution family in fp32 on the f32-input MFMA (v_mfma_f32_32x32x2_f32): the PARITY configuration
 * (losses / IoU within 1e-4 of the fp32 reference path) and the precision of the VAT direction
 * pass (train_seg_semisup_vat_mt.py:228-301). Same descriptors; every tensor the bf16 entry points take as bf16 is
 * fp32 here (x, w, y, res, mask_src; du, x of the weight gradient); Cin % 32 == 0; `tile`, `zeros`, `variant` and the
 * BatchNorm-affine side outputs of the weight gradient are not used.
 * ------------------------------------------------------------------------------------------------------------ */
int cms_conv_igemm_f32(const cms_conv_desc* d, void* stream);
int cms_conv_wgrad_f32(const cms_wgrad_desc* d, void* stream);
/* dst[tap'][ci][co] = src[tap][co][ci] * scale[co], fp32 -> fp32 */
int cms_conv_pack_transpose_f32(const float* src, float* dst, const float* scale, int ntaps, int cout, int cin,
                                int flip, void* stream);
/* cms_conv_pack_transpose_batch with fp32 sources AND fp32 destinations */
int cms_conv_pack_transpose_batch_f32(const cms_pack_item* items_dev, int n_items, int total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Device-side input staging (csrc/augment.hip): crop with random scale, flips, ColorJitter / RandomGrayscale,
 * standardisation and NCHW conversion of uint8 source images resident in HBM -- the loader-worker transforms of
 * datapipe/seg_transforms_cv.py:29-133, 169-231, 452-497, 541-623 as wired at train_seg_semisup_mask_mt.py:150-183.
 * params[n][CMS_AUG_PARAMS] (DEVICE, float), drawn on the host in the reference's order:
 *   0 y0, 1 x0   window origin in (unpadded) source pixels, may be negative (padding: image 0 after standardisation,
 *                label 255, mask 0)         2 sc_h, 3 sc_w   window size (== h, w without random scale)
 *   4 flip_x, 5 flip_y, 6 transpose (0/1)   7 brightness, 8 contrast, 9 saturation factors, 10 hue shift (turns)
 *   11 greyscale (0/1)   12 colour jitter applied (0/1)   13 order of the four jitter ops, base-4 digits
 *   14 contrast pivot (mean luminance; fill with cms_augment_luma x brightness when brightness comes first)
 *   15 geometry: 0 = the axis-aligned window of slots 0..3 (crop / Hung scale, above); 1 = AFFINE WARP, the random
 *      rotate + scale crop of datapipe/seg_transforms_cv.py:306-372 (cv2.warpAffine): output pixel (x, y) samples the
 *      source at (a00 x + a01 y + a02, a10 x + a11 y + a12) with slots 16..21 = a00 a01 a02 a10 a11 a12 (the INVERSE of
 *      the reference's local_xf), slot 22 = interpolation of the image (0 nearest: floor(s + 0.5); 1 bilinear), image
 *      border REFLECT_101, labels nearest with 255 outside, mask = in-bounds weight (constant border 0); 23 reserved
 * out0 = geometric transform only (teacher view), out1 = + colour augmentation (student view); either may be NULL.
 * ------------------------------------------------------------------------------------------------------------ */
#define CMS_AUG_PARAMS 24
typedef struct cms_augment_desc {
    const uint8_t* src;         /* uint8 [N][hs][ws][3]                                  */
    const uint8_t* src_labels;  /* uint8 [N][hs][ws] or NULL                             */
    void* out0;                 /* (N,3,h,w) NCHW, out_dtype, or NULL                    */
    void* out1;                 /* (N,3,h,w) NCHW, out_dtype, or NULL                    */
    uint8_t* out_labels;        /* (N,h,w) uint8 or NULL (255 outside the source image)  */
    float* out_mask;            /* (N,1,h,w) fp32 validity mask or NULL                  */
    const float* params;        /* DEVICE [N][CMS_AUG_PARAMS]                            */
    float mean[3], std_[3];     /* standardisation (seg_transforms_cv.py:600-612)        */
    int n, hs, ws, h, w;
    int out_dtype;
} cms_augment_desc;
int cms_augment_batch(const cms_augment_desc* d, void* stream);
/* luma[n] = mean luminance in [0,1] of sample n after its geometric transform (ColorJitter's contrast pivot) */
int cms_augment_luma(const cms_augment_desc* d, float* luma, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Batch-statistics BatchNorm (+ ReLU, + residual) on NHWC activations (csrc/bn.hip): nn.BatchNorm2d in training mode,
 * architectures/deeplab2.py:72-84 without --freeze_bn, architectures/deeplab3plus.py:40-64 (head, always).
 * Statistics are a two-pass protocol so that a data-parallel caller can all-reduce `sums` (and the pixel count) between
 * the passes: SyncBN, SURVEY.md 8(e). `sums` = double[2*C], zero-filled by the caller, accumulated with atomics.
 *   forward : cms_bn_reduce(mode 0) -> sums = (sum x, sum x^2);  cms_bn_finalize -> mean, rstd, scale, shift and the
 *             running statistics (momentum, unbiased variance);  cms_bn_apply: y = relu(x*scale + shift (+ res))
 *   backward: cms_bn_reduce(mode 1) -> sums = (sum dy', sum dy'*xhat), dy' = dy * [y > 0] (y NULL: no ReLU);
 *             cms_bn_bwd_apply: dx = gamma*rstd*(dy' - sums0/count - xhat*sums1/count), optional dres = dy'.
 *             dgamma = sums1, dbeta = sums0 (of the LOCAL pass).
 * ------------------------------------------------------------------------------------------------------------ */
int cms_bn_reduce(const void* x, const void* dy, const void* y, int dtype, const float* mean, const float* rstd, double* sums,
                  size_t n_pixels, int c, int mode, void* stream);
int cms_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum,
                    float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var, int c,
                    void* stream);
/* The same, plus what a REPLAYED pass needs from this launch (cms_program_add_bn): `clear_a` / `clear_b` (double[2*c] each, or
 * NULL) are zeroed after the statistics were read -- normally the forward sums themselves and the sums of the unit's backward
 * pass -- and `counter` (nn.BatchNorm2d.num_batches_tracked, or NULL) is incremented: three tiny launches fewer per layer. */
int cms_bn_finalize_ex(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum,
                       float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var, int c,
                       double* clear_a, double* clear_b, long long* counter, void* stream);
/* Round 3: the reductions without data atomics (inside the step the fp64 adds of cms_bn_reduce queue behind the weight
 * gradients' fp32 atomics in the memory-side units: 52 us per launch in the step against 10-22 alone). Blocks own 64-channel
 * tiles, store partial sums into `ws` and the last block of a tile adds them in fixed order in fp64: bit-reproducible; `sums` is
 * OVERWRITTEN, not accumulated. `ws`: cms_bn_workspace_bytes(n_pixels, c, groups) bytes, zero-filled once by the caller, owned
 * by one call site (launches on different streams must not share it); it is left ready for the next launch.
 * `groups` (>= 1, dividing n_pixels): the pixel rows are `groups` equal runs of consecutive samples whose statistics are kept
 * apart -- ONE launch over [supervised batch; mixed batch] normalises each exactly as the reference's separate forward passes
 * do (train_seg_semisup_mask_mt.py:296-358); the running statistics move once per group, in group order, `counter` by `groups`.
 * Layouts with groups: mean / rstd / scale / shift float[groups][c], sums double[groups][2][c]; `count` = pixels of ONE group.
 *   cms_bn_reduce_ws : cms_bn_reduce's contract (mode 0 / 1) -- the data-parallel protocol all-reduces `sums` after it.
 *   cms_bn_stats     : forward statistics AND cms_bn_finalize_ex's work (count = n_pixels / groups) in the one launch, for
 *                      single-process callers; `sums` optional (NULL: not written).
 *   cms_bn_apply_groups / cms_bn_bwd_apply_groups : the element-wise passes with per-group coefficients. */
size_t cms_bn_workspace_bytes(size_t n_pixels, int c, int groups);
int cms_bn_reduce_ws(const void* x, const void* dy, const void* y, int dtype, const float* mean, const float* rstd, double* sums,
                     size_t n_pixels, int c, int groups, int mode, void* ws, void* stream);
int cms_bn_stats(const void* x, int dtype, size_t n_pixels, int c, int groups, const float* gamma, const float* beta, float eps,
                 float momentum, float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var,
                 long long* counter, double* sums, void* ws, void* stream);
int cms_bn_apply_groups(const void* x, const void* res, void* y, int dtype, const float* scale, const float* shift, int relu,
                        size_t n_pixels, int c, int groups, void* stream);
int cms_bn_bwd_apply_groups(const void* x, const void* dy, const void* y, void* dx, void* dres, int dtype, const float* mean,
                            const float* rstd, const float* gamma, const double* sums, double count, size_t n_pixels, int c,
                            int groups, void* stream);
/* Round 5: the ReLU mask of a unit as BITS beside its output -- uint8 [pixel rows][c / 8], bit e of byte (row, v) = [stored
 * y[row][8 v + e] > 0], the layout of cms_conv_desc.mask_bits_out. cms_bn_apply_groups_bits writes them (mask_bits_out may be NULL),
 * the backward passes read them INSTEAD of y: 1/16 of the bytes of one of the three tensors each of them streams.
 * (cms_bn_bwd_apply_groups_bits: mask_bits == NULL falls back to y.) Results are bit-identical to the y-masked calls. */
int cms_bn_apply_groups_bits(const void* x, const void* res, void* y, int dtype, const float* scale, const float* shift, int relu,
                             size_t n_pixels, int c, int groups, uint8_t* mask_bits_out, void* stream);
/* Backward of y = relu(x * scale + shift (+ res)) with FROZEN statistics (an eval-mode BatchNorm as an affine: the forward is
 * cms_bn_apply with scale = gamma * rstd, shift = beta - mean * scale; architectures/deeplab2.py LayerEngine.bn_act, the teacher of
 * train_seg_semisup_vat_mt.py:237): dx = scale * dy', dres = dy' (or NULL), dy' = dy * [y > 0] (y = the forward's output; NULL: no ReLU). */
int cms_frozen_bn_act_bwd(const void* dy, const void* y, void* dx, void* dres, int dtype, const float* scale, size_t n_pixels, int c,
                          void* stream);
int cms_bn_reduce_ws_bits(const void* x, const void* dy, const uint8_t* mask_bits, int dtype, const float* mean, const float* rstd,
                          double* sums, size_t n_pixels, int c, int groups, void* ws, void* stream);
int cms_bn_bwd_apply_groups_bits(const void* x, const void* dy, const void* y, const uint8_t* mask_bits, void* dx, void* dres,
                                 int dtype, const float* mean, const float* rstd, const float* gamma, const double* sums,
                                 double count, size_t n_pixels, int c, int groups, void* stream);
int cms_bn_apply(const void* x, const void* res, void* y, int dtype, const float* scale, const float* shift, int relu,
                 size_t n_pixels, int c, void* stream);
int cms_bn_bwd_apply(const void* x, const void* dy, const void* y, void* dx, void* dres, int dtype, const float* mean,
                     const float* rstd, const float* gamma, const double* sums, double count, size_t n_pixels, int c,
                     void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * NHWC data movement of the DeepLab v3+ head (csrc/nhwc.hip; reference: architectures/deeplab3plus.py:40-55 -- torch.cat of
 * the ASPP branches, the pooled branch broadcast over the map, F.interpolate(bilinear, align_corners=False) + torch.cat with
 * the low-level features -- and autograd's sum of the gradients that meet at the ASPP input). Pixel rows may sit in wider
 * rows of a concat buffer: `*_pitch` = elements per row of that buffer; rows, pitches and pointers are multiples of 16 bytes.
 *   cms_channel_copy : dst[r][0:channels] = src[r / row_div][0:channels]   (row_div > 1 broadcasts one row per sample)
 *   cms_add_n        : dst = src_0 + ... + src_{k-1}, k <= 6, fp32 accumulation, dense tensors of n elements (n % 8 == 0)
 *   cms_rows_reduce  : dst[n][c] (fp32) = scale * sum_p src[n][p][c]        (global average pool; adjoint of the broadcast)
 *   cms_upsample_nhwc: backward == 0: dst[N,H,W, pitch] <- bilinear(src[N,h,w,C]); backward != 0: the adjoint in gather form,
 *                      src = d(dst) in rows of dst_pitch, dst = d(src) dense (N,h,w,C). No atomics, fixed summation order.
 * ------------------------------------------------------------------------------------------------------------ */
int cms_channel_copy(const void* src, size_t src_pitch, void* dst, size_t dst_pitch, size_t rows, int channels, int dtype,
                     size_t row_div, void* stream);
int cms_add_n(const void* const* srcs, int k, void* dst, size_t n, int dtype, void* stream);
int cms_rows_reduce(const void* src, size_t pitch, int n, size_t rows_per_sample, int channels, int dtype, float* dst,
                    float scale, void* stream);
int cms_upsample_nhwc(const void* src, void* dst, size_t dst_pitch, int n, int h, int w, int H, int W, int channels, int dtype,
                      int align_corners, int backward, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * ASPP head (architectures/deeplab2.py:112-128) with the 2048-channel activation read ONCE (csrc/aspp.hip):
 *   forward   Z = X . Wall^T as a 1x1 cms_conv_igemm (rows of Wall: tap*C + class, fp32 NCHW output), then
 *             logits[n][c][y][x] = bias[c] + sum_t Z[n][t*C + c][y + dy_t][x + dx_t]            (cms_aspp_gather_fwd)
 *   backward  D[n][y][x][t*C + c] = dlogits[n][c][y - dy_t][x - dx_t]                             (cms_aspp_spread_bwd),
 *             then dX = D . Wall (1x1 cms_conv_igemm) and dWall = D^T . X (one cms_conv_wgrad).
 * zc = channel count of Z / D (taps * classes padded to what the GEMMs need); columns >= taps * classes are zero.
 * ------------------------------------------------------------------------------------------------------------ */
int cms_aspp_gather_fwd(const float* z, const float* bias, float* logits, const int* tap_dy, const int* tap_dx, int n_taps,
                        int n, int c, int zc, int h, int w, void* stream);
int cms_aspp_spread_bwd(const float* dlogits, void* d_nhwc, int d_dtype, const int* tap_dy, const int* tap_dx, int n_taps,
                        int n, int c, int zc, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Network stem: 7x7 / stride 2 / pad 3 convolution 3 -> 64 + frozen BatchNorm + ReLU, 3x3 / stride 2 / pad 1 ceil-mode
 * max-pool, and their backward passes (architectures/deeplab2.py:140-146, 183-186; csrc/stem.hip).
 * Images NCHW (the reference's batch layout), activations NHWC; dtypes CMS_F32 / CMS_BF16 per tensor.
 * ------------------------------------------------------------------------------------------------------------ */
/* w_packed[(c*7 + ky)*7 + kx][co] (fp32, 147 x 64) from the convolution weight in its [kh][kw][Cout][Cin] layout,
 * followed by the bf16 (hi, lo) MFMA fragments of the bf16 forward: the buffer holds (147 + 176) x 64 floats. */
int cms_stem_pack_weights(const void* w_khkwcoci, int w_dtype, float* w_packed, void* stream);
/* sizes of the stem convolution output (ho, wo) and of the max-pool output (hp, wp) for an h x w image */
int cms_stem_out_hw(int h, int w, int* ho, int* wo, int* hp, int* wp);
/* y[n][oy][ox][co] = relu(conv(x)[co] * scale[co] + bias[co]) */
int cms_stem_fwd(const void* x_nchw, int x_dtype, void* y_nhwc, int y_dtype, const float* w_packed, const float* scale,
                 const float* bias, int n, int h, int w, void* stream);
/* ceil_mode 1: DeepLab v2 (deeplab2.py:146); 0: torchvision's ResNet stem (DeepLab v3+ backbone).
 * p = maxpool(s); argmax[n][py][px][c] = ky*3+kx of the FIRST maximum of the window (ATen's rule) */
int cms_maxpool3x3s2_fwd(const void* s_nhwc, void* p_nhwc, uint8_t* argmax, int dtype, int n, int hs, int ws, int c,
                         int ceil_mode, void* stream);
/* ds = [s > 0] * scatter(dp through argmax): max-pool backward fused with the ReLU backward of its input */
int cms_maxpool3x3s2_relu_bwd(const void* dp_nhwc, const uint8_t* argmax, const void* s_nhwc, void* ds_nhwc, int dtype,
                              int n, int hs, int ws, int c, int ceil_mode, void* stream);
/* dw[ky][kx][co][c] (fp32, accumulated with atomics) += scale[co] * sum_pixels ds[pix][co] * x[c][2oy-3+ky][2ox-3+kx] */
int cms_stem_wgrad(const void* x_nchw, int x_dtype, const void* ds_nhwc, int ds_dtype, float* dw_khkwcoci,
                   const float* scale, int n, int h, int w, void* stream);
/* The same with an optional scratch buffer (>= cms_stem_wgrad_workspace_bytes): the persistent blocks then write their
 * partial sums there and a second launch adds them to dw in block order -- run-to-run DETERMINISTIC (NULL: fp32 atomics). */
long long cms_stem_wgrad_workspace_bytes(int x_dtype, int ds_dtype, int n, int h, int w);
int cms_stem_wgrad_ws(const void* x_nchw, int x_dtype, const void* ds_nhwc, int ds_dtype, float* dw_khkwcoci,
                      const float* scale, int n, int h, int w, void* workspace, long long workspace_bytes, void* stream);
/* dx (fp32 NCHW) = gradient wrt the image (VAT direction pass, train_seg_semisup_vat_mt.py:244-268) */
int cms_stem_dgrad(const void* ds_nhwc, int ds_dtype, const float* w_packed, const float* scale, float* dx_nchw, int n,
                   int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Launch programs: the host side of a network pass, recorded once per input shape and replayed from C++
 * (csrc/program.hip). Replaces one Python -> ctypes round trip per convolution of architectures/deeplab2.py:89-128
 * (and of its backward pass) by ONE call per pass. Ops refer to caller-owned, persistent device buffers; stream
 * indices are positions in the `streams` array given at run time (index 0 = the caller's current stream).
 * add_* return the op index (>= 0) or a negative error code.
 * ------------------------------------------------------------------------------------------------------------ */
#define CMS_PROGRAM_MAX_STREAMS 4
typedef struct cms_program cms_program;
int cms_program_create(cms_program** out);
int cms_program_destroy(cms_program* p);
int cms_program_add_conv(cms_program* p, const cms_conv_desc* d, int f32, int stream_idx, int group);
int cms_program_add_wgrad(cms_program* p, const cms_wgrad_desc* d, int f32, int stream_idx, int group);
/* a grouped weight-gradient launch (cms_conv_wgrad_group_run) as a program op */
int cms_program_add_wgrad_group(cms_program* p, const void* table_dev, int n_items, int total_blocks, int kind, int stream_idx,
                                int group);
int cms_program_add_memset(cms_program* p, void* ptr, size_t bytes, int stream_idx, int group);
int cms_program_add_aspp_gather(cms_program* p, const float* z, const float* bias, float* logits, const int* tap_dy,
                                const int* tap_dx, int n_taps, int n, int c, int zc, int h, int w, int stream_idx, int group);
int cms_program_add_aspp_spread(cms_program* p, const float* dlogits, void* d_nhwc, int d_dtype, const int* tap_dy,
                                const int* tap_dx, int n_taps, int n, int c, int zc, int h, int w, int stream_idx, int group);
/* work enqueued so far on `from_stream` must finish before anything enqueued later on `to_stream` starts */
int cms_program_add_sync(cms_program* p, int from_stream, int to_stream, int group);
/* Syncs without events (round 6). With a caller-owned, ZEROED device buffer of n_flags >= cms_program_sync_count(p) + 1 ints
 * (persistent like every buffer of a program; the last word counts waiter timeouts and stays 0), a sync is replayed as a one-wave
 * setter kernel on the producing stream + a one-wave polling kernel on the waiting stream instead of hipEventRecord +
 * hipStreamWaitEvent (which costs the producing stream ~12 us per pair of waiters, tools/event_cost_probe.hip). NULL / 0 = events
 * again; CMS_PROG_FLAG_SYNC=0 in the environment keeps the events whatever is set; a replay under stream capture uses events.
 * Opt-in: in the training step the event form is FASTER (profiles/r06ae_*), so ops.Program only sets flags on request.
 * (The reference has no counterpart: its step is one stream, train_seg_semisup_mask_mt.py:287-476.) */
int cms_program_sync_count(const cms_program* p);
int cms_program_set_sync_flags(cms_program* p, int* flags_dev, int n_flags);
/* Batch-statistics BatchNorm launches inside a program (round 3: DeepLab v2 WITHOUT --freeze_bn on the executor,
 * architectures/deeplab2.py:72-84 / train_seg_semisup_mask_mt.py:587). `what`: 0 = cms_bn_reduce(mode 0), 1 = cms_bn_finalize,
 * 2 = cms_bn_apply, 3 = cms_bn_reduce(mode 1), 4 = cms_bn_bwd_apply, 5 = cms_increment_counter(counter), 6 = cms_bn_stats,
 * 7 = cms_bn_finalize_tiles (tile sums in `ws`, tile rows in `reserved`), 8 = cms_bn_bwd_sums_tiles (likewise; writes `sums`); unused pointers NULL. With `ws` set, what 0 / 3
 * run cms_bn_reduce_ws.
 * All buffers are the caller's and persistent (a program is replayed many times). */
/* Statistics of a unit from the tile sums its convolution wrote (cms_conv_desc.stats_out): per group and channel the tiles' sums are
 * added in a fixed order in fp64, then finalised exactly like cms_bn_stats (mean / rstd / scale / shift [groups][c], running statistics
 * moved once per group in group order, *counter += groups). tile_rows = cms_conv_igemm_stats_tile_rows of that launch. */
int cms_bn_finalize_tiles(const float* tile_sums, int tile_rows, size_t n_pixels, int c, int groups, const float* gamma,
                          const float* beta, float eps, float momentum, float* mean, float* rstd, float* scale, float* shift,
                          float* running_mean, float* running_var, long long* counter, void* stream);

/* backward counterpart: tile sums of a data-gradient launch with bstats_* -> sums[groups][2][c] (double) for cms_bn_bwd_apply_groups */
int cms_bn_bwd_sums_tiles(const float* tile_sums, int tile_rows, size_t n_pixels, int c, int groups, double* sums, void* stream);

/* Trainable BatchNorm affine over frozen statistics on the eight-phase weight-gradient kernel (csrc/wfinish.hip; the torchvision
 * backbone of architectures/deeplab3plus.py:96-98 + autograd): the weight-gradient launch writes the UNSCALED gradient G into a
 * scratch tensor of the weight's shape, cms_channel_sum takes d(beta) = sum_p dU, and one finishing launch per backward pass does,
 * per item and output channel co:  grad += scale[co] * G,  wdot[co] += <W[co], G[co]>,  G = 0.  All buffers are the caller's. */
typedef struct cms_wfinish_item {
    float* scratch;            /* fp32 [ntaps][cout][cin]: G of this pass, cleared by the launch                         */
    float* grad;               /* fp32 [ntaps][cout][cin]: the gradient arena's slice of the weight                      */
    const uint16_t* w;         /* bf16 [ntaps][cout][cin]: the weight                                                    */
    const float* scale;        /* [cout] gamma * rstd, or NULL (1)                                                        */
    float* wdot;               /* [cout], accumulated                                                                     */
    int ntaps, cout, cin;      /* cin % 4 == 0                                                                            */
    int first_block;           /* filled by cms_wgrad_finish_pack                                                         */
} cms_wfinish_item;
/* dst[c] (fp32, accumulated with atomics) += sum over `rows` rows of src[row][c]; channels % 64 == 0 */
int cms_channel_sum(const void* src, int dtype, size_t rows, int channels, float* dst, void* stream);
/* host: lays the items out over the grid (first_block) -> total workgroups, or a negative error code */
int cms_wgrad_finish_pack(cms_wfinish_item* items, int n_items);
/* items_dev: the packed table in device memory */
int cms_wgrad_finish_run(const void* items_dev, int n_items, int total_blocks, void* stream);
/* ... as program ops */
int cms_program_add_channel_sum(cms_program* p, const void* src, int dtype, size_t rows, int channels, float* dst, int stream_idx,
                                int group);
int cms_program_add_wgrad_finish(cms_program* p, const void* items_dev, int n_items, int total_blocks, int stream_idx, int group);

typedef struct cms_bn_op {
    int what, dtype, c, relu;
    const void* x;             /* conv output u, NHWC                                                          */
    const void* res;           /* apply: residual or NULL                                                      */
    void* y;                   /* apply: output; reduce(mode 1) / bwd_apply: the stored output (ReLU mask) or NULL */
    const void* dy;            /* backward: incoming gradient                                                  */
    void* dx;                  /* bwd_apply: gradient wrt x                                                    */
    void* dres;                /* bwd_apply: gradient wrt the residual (= masked dy) or NULL                   */
    double* sums;              /* double[2*c] (+1): forward / backward sums                                    */
    const float* gamma;
    const float* beta;
    float* mean;
    float* rstd;
    float* scale;
    float* shift;
    float* running_mean;
    float* running_var;
    long long* counter;        /* what 1 (optional) / what 5: num_batches_tracked                              */
    double* clear_a;           /* what 1: zeroed after the statistics were read (cms_bn_finalize_ex), or NULL  */
    double* clear_b;
    void* ws;                  /* what 0 / 3 (optional), 6: cms_bn_workspace_bytes(n_pixels, c, groups) bytes;
                                * what 7: the tile sums (cms_conv_desc.stats_out of the unit's convolution)   */
    double count;              /* pixels the statistics run over                                               */
    unsigned long long n_pixels;
    float eps, momentum;
    int groups;                /* sample groups (0 / 1: one); needs `ws` for what 0 / 3                       */
    int reserved;              /* what 7: pixel rows per statistics tile (cms_conv_igemm_stats_tile_rows)      */
    void* mask_bits;           /* what 2: written ([y > 0] as bits, or NULL); what 3 (with ws) / 4: read instead of y */
} cms_bn_op;
int cms_program_add_bn(cms_program* p, const cms_bn_op* op, int stream_idx, int group);
int cms_program_size(const cms_program* p);
/* enqueue ops [first, last) (last < 0: to the end); never synchronises the host */
int cms_program_run(cms_program* p, int first, int last, void* const* streams, int n_streams);
/* two programs issued interleaved by `group` (student on one stream, teacher on another) */
int cms_program_run_pair(cms_program* a, void* const* streams_a, int na, cms_program* b, void* const* streams_b, int nb);
/* bracket every k-th convolution launch with HIP events on its stream (0 = off) / read and reset the sums */
int cms_program_set_timing(cms_program* p, int every_k);
/* the ASPP head launches (fp32 NCHW output) are always bracketed while timing is on and summed separately */
int cms_program_read_timing(cms_program* p, double* sum_ms, double* sum_flops, long* launches, double* head_ms,
                            long* head_launches);

#ifdef __cplusplus
}
#endif
#endif /* CUTMIXSEG_H */
