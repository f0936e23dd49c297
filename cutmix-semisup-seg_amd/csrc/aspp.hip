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

// (round 6) Both kernels read their tap offsets from LDS (filled with constant indices): indexing the by-value AsppTaps argument
// with a run-time tap number puts the struct into scratch memory -- every tap of every element then paid a scratch round trip.
__device__ __forceinline__ void taps_to_lds(const AsppTaps& tp, short* sdy, short* sdx) {
#pragma unroll
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i)
        if ((int)threadIdx.x == i) {
            sdy[i] = tp.dy[i];
            sdx[i] = tp.dx[i];
        }
    __syncthreads();
}

__global__ __launch_bounds__(256) void aspp_gather_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias,
                                                              float* __restrict__ logits, AsppTaps tp, int N, int C, int ZC,
                                                              int H, int W) {
    __shared__ short sdy[CMS_CONV_MAX_TAPS], sdx[CMS_CONV_MAX_TAPS];
    taps_to_lds(tp, sdy, sdx);
    const int n_taps = tp.n_taps;
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned ii = (unsigned)i;                  // (total < 2^32: checked by the launcher)
        const int x = (int)(ii % (unsigned)W);
        unsigned t = ii / (unsigned)W;
        const int y = (int)(t % (unsigned)H); t /= (unsigned)H;
        const int c = (int)(t % (unsigned)C);
        const int n = (int)(t / (unsigned)C);
        float acc = bias ? bias[c] : 0.0f;
        const float* zn = z + ((size_t)n * ZC + c) * H * W;
        const size_t tap_stride = (size_t)C * H * W;
        for (int k = 0; k < n_taps; ++k) {                // (same order of additions as before: bit-identical)
            const int yy = y + sdy[k], xx = x + sdx[k];
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc += zn[(size_t)k * tap_stride + (size_t)yy * W + xx];
        }
        logits[i] = acc;
    }
}

// D[n][y][x][k*C + c] = dlogits[n][c][y - dy_k][x - dx_k] (zero outside, zero in the padded columns >= T*C).
// (round 6) a thread writes 8 consecutive columns of one pixel as ONE 16-byte (bf16) / two 16-byte (fp32) stores -- it was one
// 2-byte store per thread: 47 us for 26 MB at cfg 2 on the chain between the losses and the first data gradient.
template <class T>
__global__ __launch_bounds__(256) void aspp_spread_bwd_kernel(const float* __restrict__ dl, T* __restrict__ d, AsppTaps tp,
                                                              int N, int C, int ZC, int H, int W) {
    __shared__ short sdy[CMS_CONV_MAX_TAPS], sdx[CMS_CONV_MAX_TAPS];
    taps_to_lds(tp, sdy, sdx);
    const int n_taps = tp.n_taps;
    const unsigned zc8 = (unsigned)ZC / 8u;               // (ZC % 8 == 0: checked by the launcher)
    const size_t total = (size_t)N * H * W * zc8;         // 8-column chunks
    const size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned ii = (unsigned)i;                  // (total < 2^32: checked by the launcher)
        const unsigned ch = ii % zc8;
        unsigned t = ii / zc8;
        const int x = (int)(t % (unsigned)W); t /= (unsigned)W;
        const int y = (int)(t % (unsigned)H);
        const int n = (int)(t / (unsigned)H);
        const float* dn = dl + (size_t)n * C * plane;
        float v[8];
        int k = (int)((ch * 8u) / (unsigned)C), c = (int)(ch * 8u) - k * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float val = 0.0f;
            if (k < n_taps) {
                const int yy = y - sdy[k], xx = x - sdx[k];
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) val = dn[(size_t)c * plane + (size_t)yy * W + xx];
            }
            v[e] = val;
            if (++c == C) { c = 0; ++k; }
        }
        if constexpr (sizeof(T) == 4) {
            float4* o = reinterpret_cast<float4*>(d + i * 8);
            o[0] = float4{v[0], v[1], v[2], v[3]};
            o[1] = float4{v[4], v[5], v[6], v[7]};
        } else {
            uint4 o;
            o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
            o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
            *reinterpret_cast<uint4*>(d + i * 8) = o;
        }
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
    CMS_REQUIRE(total < (1ull << 32), "aspp_gather_fwd: more than 2^32 logits");
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
    CMS_REQUIRE(zc % 8 == 0, "aspp_spread_bwd: the stacked tap x class axis (%d) must be a multiple of 8", zc);
    const size_t total = (size_t)n * h * w * (zc / 8);         // 8-column chunks, one per thread and iteration
    CMS_REQUIRE(total < (1ull << 32), "aspp_spread_bwd: more than 2^32 chunks");
    hipStream_t s = (hipStream_t)stream;
    if (d_dtype == CMS_F32)
        hipLaunchKernelGGL(aspp_spread_bwd_kernel<float>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, dlogits,
                           (float*)d_nhwc, tp, n, c, zc, h, w);
    else
        hipLaunchKernelGGL(aspp_spread_bwd_kernel<uint16_t>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, dlogits,
                           (uint16_t*)d_nhwc, tp, n, c, zc, h, w);
    return launch_status("cms_aspp_spread_bwd");
}
