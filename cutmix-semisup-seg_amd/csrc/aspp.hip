// ASPP head of DeepLab v2 (architectures/deeplab2.py:112-128: conv_d6(x) + conv_d12(x), 2048 -> C classes, 3x3 taps)
// as ONE pass over the 2048-channel activation instead of one per tap.
//
// logits[n][c][y][x] = bias[c] + sum_{t < T} sum_ci W[t][c][ci] * X[n][y + dy_t][x + dx_t][ci]       (T = 18 taps)
//
// The implicit-GEMM formulation reads the activation tile once per tap (18 x 138 MB of L2 -> LDS traffic at cfg 2 for
// 1.3 GFLOP per image) -- SURVEY.md 8(d) lists this layer as HBM-bound: 2048*h*w*s read ONCE. So:
//   forward   Z[n][t*C + c][y][x] = sum_ci W[t][c][ci] * X[n][y][x][ci]     a plain 1x1 GEMM (cms_conv_igemm, Cout = T*C
//             padded to a multiple of 128, fp32 NCHW output), X read once;
//             logits = bias + sum_t shift_t(Z[t])                            cms_aspp_gather_fwd (this file), 2 x T*C*h*w*4 B
//   backward  D[n][y][x][t*C + c] = dlogits[n][c][y - dy_t][x - dx_t]        cms_aspp_spread_bwd (this file)
//             dX = D . Wall  (1x1 GEMM, K = T*C instead of 18 taps x 64 padded classes),  dWall = D^T . X (one weight-
//             gradient GEMM instead of one per branch and tap).
#include "common.hpp"

namespace cms {

struct AsppTaps {
    int n_taps;
    short dy[CMS_CONV_MAX_TAPS], dx[CMS_CONV_MAX_TAPS];
};

__global__ __launch_bounds__(256) void aspp_gather_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              float* __restrict__ logits, AsppTaps tp, int N, int C, int ZC,
                                                              int H, int W) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        size_t t = i / W;
        const int y = (int)(t % H); t /= H;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        float acc = bias ? bias[c] : 0.0f;
#pragma unroll 6
        for (int k = 0; k < tp.n_taps; ++k) {
            const int yy = y + tp.dy[k], xx = x + tp.dx[k];
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                acc += z[(((size_t)n * ZC + k * C + c) * H + yy) * W + xx];
        }
        logits[i] = acc;
    }
}

// D[n][y][x][k*C + c] = dlogits[n][c][y - dy_k][x - dx_k] (zero outside, zero in the padded columns >= T*C)
template <class T>
__global__ __launch_bounds__(256) void aspp_spread_bwd_kernel(const float* __restrict__ dl, T* __restrict__ d, AsppTaps tp,
                                                              int N, int C, int ZC, int H, int W) {
    const size_t total = (size_t)N * H * W * ZC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int zc = (int)(i % ZC);
        size_t t = i / ZC;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        float v = 0.0f;
        const int k = zc / C, c = zc - k * C;
        if (k < tp.n_taps) {
            const int yy = y - tp.dy[k], xx = x - tp.dx[k];
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = dl[(((size_t)n * C + c) * H + yy) * W + xx];
        }
        if constexpr (sizeof(T) == 4) d[i] = v;
        else d[i] = f32_to_bf16(v);
    }
}

}  // namespace cms

using namespace cms;

static int fill_taps(AsppTaps& tp, const int* dy, const int* dx, int n_taps) {
    CMS_REQUIRE(dy && dx && n_taps > 0 && n_taps <= CMS_CONV_MAX_TAPS, "aspp: 1..%d taps", CMS_CONV_MAX_TAPS);
    tp.n_taps = n_taps;
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        tp.dy[i] = (short)(i < n_taps ? dy[i] : 0);
        tp.dx[i] = (short)(i < n_taps ? dx[i] : 0);
    }
    return CMS_OK;
}

extern "C" int cms_aspp_gather_fwd(const float* z, const float* bias, float* logits, const int* tap_dy, const int* tap_dx,
                                   int n_taps, int n, int c, int zc, int h, int w, void* stream) {
    CMS_REQUIRE(z && logits, "aspp_gather_fwd: NULL pointer");
    CMS_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && zc >= n_taps * c, "aspp_gather_fwd: bad geometry (zc >= taps * classes)");
    AsppTaps tp;
    const int rc = fill_taps(tp, tap_dy, tap_dx, n_taps);
    if (rc) return rc;
    const size_t total = (size_t)n * c * h * w;
    hipLaunchKernelGGL(aspp_gather_fwd_kernel, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, z, bias,
                       logits, tp, n, c, zc, h, w);
    return launch_status("cms_aspp_gather_fwd");
}

extern "C" int cms_aspp_spread_bwd(const float* dlogits, void* d_nhwc, int d_dtype, const int* tap_dy, const int* tap_dx,
                                   int n_taps, int n, int c, int zc, int h, int w, void* stream) {
    CMS_REQUIRE(dlogits && d_nhwc, "aspp_spread_bwd: NULL pointer");
    CMS_REQUIRE(d_dtype == CMS_F32 || d_dtype == CMS_BF16, "aspp_spread_bwd: bad dtype");
    CMS_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && zc >= n_taps * c, "aspp_spread_bwd: bad geometry (zc >= taps * classes)");
    AsppTaps tp;
    const int rc = fill_taps(tp, tap_dy, tap_dx, n_taps);
    if (rc) return rc;
    const size_t total = (size_t)n * h * w * zc;
    hipStream_t s = (hipStream_t)stream;
    if (d_dtype == CMS_F32)
        hipLaunchKernelGGL(aspp_spread_bwd_kernel<float>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, dlogits,
                           (float*)d_nhwc, tp, n, c, zc, h, w);
    else
        hipLaunchKernelGGL(aspp_spread_bwd_kernel<uint16_t>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, dlogits,
                           (uint16_t*)d_nhwc, tp, n, c, zc, h, w);
    return launch_status("cms_aspp_spread_bwd");
}
