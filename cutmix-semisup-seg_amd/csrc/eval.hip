// Evaluation kernels (gfx950): fused bilinear upsample + argmax + confusion matrix.
//   train_seg_semisup_mask_mt.py:510-514  argmax(logits, 1) -> D2H -> EvaluatorIoU.sample per image
//   evaluation.py:6-37                    fast_cm (bincount of truth*C + pred) and per-class I / U
// The reference ships N*H*W int64 predictions to the host and loops over classes in numpy (0.28 s per 321x321
// image); here a workgroup keeps a C x C histogram in LDS (integer atomics, order-independent => bit exact) and
// flushes it with one 64-bit atomic per non-empty bin. I = diag, U = rowsum + colsum - diag on the host side.
// HBM-bound on the label read (1 or 8 B per pixel); logits are the low-res map (L2 resident).
#include "common.hpp"

namespace cms {

constexpr int kMaxEvalClasses = 64;

__device__ __forceinline__ int eval_label(const void* labels, int label_dtype, size_t pix) {
    if (label_dtype == CMS_LABEL_U8) return (int)((const uint8_t*)labels)[pix];
    const int64_t v = ((const int64_t*)labels)[pix];
    return (v < 0 || v > 0x7fffffff) ? -1 : (int)v;
}

template <bool IDENT>
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float* __restrict__ logits, const void* labels,
                                                               int label_dtype, int ignore_index, int N, int C, int h,
                                                               int w, int H, int W, float sy, float sx, int align,
                                                               unsigned long long* __restrict__ cm,
                                                               uint8_t* __restrict__ pred_out) {
    __shared__ unsigned int hist[kMaxEvalClasses * kMaxEvalClasses];
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const size_t P = (size_t)N * H * W;
    const size_t plane = (size_t)h * w;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const size_t t = idx / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        const float* base = logits + (size_t)n * C * plane;
        Tap ty, tx;
        size_t off = 0;
        if (IDENT) {
            off = (size_t)y * w + x;
        } else {
            ty = bilin_tap(y, sy, h, align != 0);
            tx = bilin_tap(x, sx, w, align != 0);
        }
        // torch.argmax: first index of the maximum; NaN counts as maximal
        int best = 0;
        float bv = IDENT ? base[off] : bilin_gather(base, w, ty, tx);
        for (int c = 1; c < C; ++c) {
            const float v = IDENT ? base[c * plane + off] : bilin_gather(base + c * plane, w, ty, tx);
            if (v > bv || (v != v && bv == bv)) {
                bv = v;
                best = c;
            }
        }
        if (pred_out) pred_out[idx] = (uint8_t)best;
        if (labels) {
            const int tr = eval_label(labels, label_dtype, idx);
            if (tr != ignore_index && tr >= 0 && tr < C) atomicAdd(&hist[tr * C + best], 1u);
        }
    }
    __syncthreads();
    if (cm) {
        for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
            const unsigned int v = hist[i];
            if (v) atomicAdd(&cm[i], (unsigned long long)v);
        }
    }
}

__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t* __restrict__ truth,
                                                        const uint8_t* __restrict__ pred, size_t count,
                                                        int ignore_index, int C,
                                                        unsigned long long* __restrict__ cm) {
    __shared__ unsigned int hist[kMaxEvalClasses * kMaxEvalClasses];
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const int tr = truth[i], pr = pred[i];
        if (tr != ignore_index && tr < C && pr < C) atomicAdd(&hist[tr * C + pr], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const unsigned int v = hist[i];
        if (v) atomicAdd(&cm[i], (unsigned long long)v);
    }
}

}  // namespace cms

using namespace cms;

extern "C" int cms_argmax_confusion(const float* logits, const void* labels, int label_dtype, int ignore_index, int n,
                                    int c, int h, int w, int H, int W, int align_corners, int64_t* cm,
                                    uint8_t* pred_out, void* stream) {
    CMS_REQUIRE(logits, "argmax_confusion: logits NULL");
    CMS_REQUIRE((labels && cm) || pred_out, "argmax_confusion: nothing to produce");
    CMS_REQUIRE(!labels || cm, "argmax_confusion: labels given without cm");
    CMS_REQUIRE(c > 0 && c <= kMaxEvalClasses, "argmax_confusion: 1 <= num_classes <= %d", kMaxEvalClasses);
    CMS_REQUIRE(c <= 256 || !pred_out, "argmax_confusion: uint8 predictions need <= 256 classes");
    CMS_REQUIRE(n > 0 && h > 0 && w > 0 && H > 0 && W > 0, "argmax_confusion: bad geometry");
    CMS_REQUIRE(label_dtype == CMS_LABEL_U8 || label_dtype == CMS_LABEL_I64, "argmax_confusion: bad label dtype");
    const size_t P = (size_t)n * H * W;
    const int grid = grid_for(P, 256, 1024);
    const float sy = bilin_scale(h, H, align_corners != 0), sx = bilin_scale(w, W, align_corners != 0);
    hipStream_t s = (hipStream_t)stream;
    if (h == H && w == W) {
        hipLaunchKernelGGL(argmax_confusion_kernel<true>, dim3(grid), dim3(256), 0, s, logits, labels, label_dtype,
                           ignore_index, n, c, h, w, H, W, sy, sx, align_corners, (unsigned long long*)cm, pred_out);
    } else {
        hipLaunchKernelGGL(argmax_confusion_kernel<false>, dim3(grid), dim3(256), 0, s, logits, labels, label_dtype,
                           ignore_index, n, c, h, w, H, W, sy, sx, align_corners, (unsigned long long*)cm, pred_out);
    }
    return launch_status("cms_argmax_confusion");
}

extern "C" int cms_confusion(const uint8_t* truth, const uint8_t* pred, size_t count, int ignore_index, int c,
                             int64_t* cm, void* stream) {
    CMS_REQUIRE(truth && pred && cm, "confusion: NULL pointer");
    CMS_REQUIRE(c > 0 && c <= kMaxEvalClasses, "confusion: 1 <= num_classes <= %d", kMaxEvalClasses);
    hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(count, 256, 1024)), dim3(256), 0, (hipStream_t)stream, truth,
                       pred, count, ignore_index, c, (unsigned long long*)cm);
    return launch_status("cms_confusion");
}
