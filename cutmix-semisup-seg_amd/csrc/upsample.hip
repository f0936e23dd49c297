// Stand-alone bilinear upsample, forward and adjoint (gfx950).
//   F.interpolate(x, size, mode='bilinear', align_corners=True)    architectures/deeplab2.py:204
//   F.interpolate(..., align_corners=False)                        architectures/deeplab3plus.py:54-55, 77
// Only the `model.forward(x) -> (N,C,H,W)` contract of the reference needs these: the training step and the
// evaluation loop use the fused kernels of losses.hip / eval.hip, which never materialise the hi-res logits.
// HBM-bound: the forward writes C*P*4 bytes and reads the (L2-resident) low-res map; the adjoint reads C*P*4 bytes
// once (each low-res cell gathers its footprint -> deterministic, no atomics).
#include "common.hpp"

namespace cms {

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ lo, float* __restrict__ hi,
                                                           int NC, int h, int w, int H, int W, float sy, float sx,
                                                           int align) {
    const size_t total = (size_t)NC * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t t = i / W;
        const int y = (int)(t % H);
        const size_t nc = t / H;
        const Tap ty = bilin_tap(y, sy, h, align != 0), tx = bilin_tap(x, sx, w, align != 0);
        hi[i] = bilin_gather(lo + nc * (size_t)h * w, w, ty, tx);
    }
}

// range of output indices whose taps can touch input index I (conservative; filtered exactly by the caller)
__device__ __forceinline__ void footprint(int I, float s, int out_size, int& lo, int& hi) {
    if (s <= 0.0f) {
        lo = 0;
        hi = out_size - 1;
        return;
    }
    const float inv = 1.0f / s;
    lo = (int)floorf(((float)I - 1.0f) * inv) - 2;
    hi = (int)ceilf(((float)I + 1.0f) * inv) + 2;
    lo = lo < 0 ? 0 : lo;
    hi = hi > out_size - 1 ? out_size - 1 : hi;
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ ghi, float* __restrict__ glo,
                                                           int NC, int h, int w, int H, int W, float sy, float sx,
                                                           int align) {
    const size_t total = (size_t)NC * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % w);
        const size_t t = i / w;
        const int Y = (int)(t % h);
        const size_t nc = t / h;
        int y_lo, y_hi, x_lo, x_hi;
        footprint(Y, sy, H, y_lo, y_hi);
        footprint(X, sx, W, x_lo, x_hi);
        const float* src = ghi + nc * (size_t)H * W;
        float acc = 0.0f;
        for (int y = y_lo; y <= y_hi; ++y) {
            const Tap ty = bilin_tap(y, sy, h, align != 0);
            const float wy = (ty.i0 == Y ? ty.w0 : 0.0f) + (ty.i1 == Y ? ty.w1 : 0.0f);
            if (wy == 0.0f) continue;
            float row = 0.0f;
            for (int x = x_lo; x <= x_hi; ++x) {
                const Tap tx = bilin_tap(x, sx, w, align != 0);
                const float wx = (tx.i0 == X ? tx.w0 : 0.0f) + (tx.i1 == X ? tx.w1 : 0.0f);
                row += wx * src[(size_t)y * W + x];
            }
            acc += wy * row;
        }
        glo[i] = acc;
    }
}

}  // namespace cms

using namespace cms;

extern "C" int cms_upsample_bilinear_fwd(const float* lo, float* hi, int n, int c, int h, int w, int H, int W,
                                         int align_corners, void* stream) {
    CMS_REQUIRE(lo && hi, "upsample_fwd: NULL pointer");
    CMS_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && H > 0 && W > 0, "upsample_fwd: bad geometry");
    const size_t total = (size_t)n * c * H * W;
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, lo, hi,
                       n * c, h, w, H, W, bilin_scale(h, H, align_corners != 0), bilin_scale(w, W, align_corners != 0),
                       align_corners);
    return launch_status("cms_upsample_bilinear_fwd");
}

extern "C" int cms_upsample_bilinear_bwd(const float* grad_hi, float* grad_lo, int n, int c, int h, int w, int H, int W,
                                         int align_corners, void* stream) {
    CMS_REQUIRE(grad_hi && grad_lo, "upsample_bwd: NULL pointer");
    CMS_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && H > 0 && W > 0, "upsample_bwd: bad geometry");
    const size_t total = (size_t)n * c * h * w;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, grad_hi,
                       grad_lo, n * c, h, w, H, W, bilin_scale(h, H, align_corners != 0),
                       bilin_scale(w, W, align_corners != 0), align_corners);
    return launch_status("cms_upsample_bilinear_bwd");
}
