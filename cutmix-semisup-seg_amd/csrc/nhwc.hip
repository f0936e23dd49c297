// Data movement of the DeepLab v3+ head on NHWC activations (architectures/deeplab3plus.py:40-55 of the reference:
// ASPP concat of five branches, global-average-pool branch broadcast back over the map, bilinear upsample of the ASPP
// output + concat with the low-level features, and autograd's sum of the five gradients that meet at the ASPP input).
// Round 2 left these to the library; in a cfg 4 profile its kernels ran 4-10x off the HBM rate on channels-last tensors
// (bilinear upsample 2.5 ms for a 170 MB output, bf16 add 1.0 ms per 346 MB operand, concat 1.05 ms for 216 MB:
// profiles/r03v3_grouped_kernel_stats_before.csv) -- 13.5 % of the step. All kernels here are HBM-bound copies with
// 16-byte lanes along the channel axis; rows (= pixels) may live in wider rows of a concat buffer (`pitch`, in elements).
#include <algorithm>
#include "common.hpp"
#include "pixel_math.hpp"

namespace cms {

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const uint16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4*>(p + 4) = float4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void st8(uint16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
}

// dst[r][0:C] = src[r / row_div][0:C] (row_div > 1: one source row broadcast over the row_div pixels of a sample), rows of
// `pitch` elements; bit copy (16-byte chunks of either dtype)
__global__ __launch_bounds__(256) void channel_copy_kernel(const uint4* __restrict__ src, size_t src_pitch16,
                                                           uint4* __restrict__ dst, size_t dst_pitch16, size_t rows,
                                                           int chunks, size_t row_div) {
    const size_t total = rows * (size_t)chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / chunks;
        const int k = (int)(i - r * chunks);
        dst[r * dst_pitch16 + k] = src[(r / row_div) * src_pitch16 + k];
    }
}

// dst = sum_k src_k (fp32 accumulation), dense tensors of n8 8-element vectors
template <class T>
__global__ __launch_bounds__(256) void add_n_kernel(const T* s0, const T* s1, const T* s2,
                                                    const T* s3, const T* s4, const T* s5, int k, T* __restrict__ dst,
                                                    size_t n8) {
    const T* s[6] = {s0, s1, s2, s3, s4, s5};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        float acc[8];
        ld8(s[0] + i * 8, acc);
#pragma unroll
        for (int j = 1; j < 6; ++j) {
            if (j < k) {
                float v[8];
                ld8(s[j] + i * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            }
        }
        st8(dst + i * 8, acc);
    }
}

// dst[n][c] = scale * sum_p src[n][p][c]  (fp32 out): block = (sample, 64-channel tile), 32 pixel rows side by side, fixed order
template <class T>
__global__ __launch_bounds__(256) void rows_reduce_kernel(const T* __restrict__ src, size_t pitch, size_t rows_per_sample, int C,
                                                          float* __restrict__ dst, float scale) {
    __shared__ float red[32][64 + 1];
    const int tiles = (C + 63) / 64;
    const int n = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int cgl = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const int c0 = tile * 64 + cgl * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        const T* base = src + (size_t)n * rows_per_sample * pitch + c0;
        size_t p = slot;
        for (; p + 96 < rows_per_sample; p += 128) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) ld8(base + (p + 32 * u) * pitch, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[u][e];
        }
        for (; p < rows_per_sample; p += 32) {
            float v[8];
            ld8(base + p * pitch, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[slot][cgl * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = tile * 64 + threadIdx.x;
        if (c < C) {
            float s = 0.0f;
            for (int k = 0; k < 32; ++k) s += red[k][threadIdx.x];
            dst[(size_t)n * C + c] = s * scale;
        }
    }
}

// bilinear upsample (N,h,w,C) -> (N,H,W,C) written into rows of `dst_pitch` elements: thread = (output pixel, 8 channels)
template <class T>
__global__ __launch_bounds__(256) void upsample_nhwc_fwd_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t dst_pitch,
                                                                int N, int h, int w, int H, int W, int C, int align) {
    const int CG = C / 8;
    const size_t total = (size_t)N * H * W * CG;
    const float sy = bilin_scale(h, H, align != 0), sx = bilin_scale(w, W, align != 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const size_t pix = i / CG;
        const int X = (int)(pix % W);
        const size_t t = pix / W;
        const int Y = (int)(t % H), n = (int)(t / H);
        const Tap ty = bilin_tap(Y, sy, h, align != 0), tx = bilin_tap(X, sx, w, align != 0);
        const T* b = src + ((size_t)n * h * w) * C + cg * 8;
        float v00[8], v01[8], v10[8], v11[8], o[8];
        ld8(b + ((size_t)ty.i0 * w + tx.i0) * C, v00);
        ld8(b + ((size_t)ty.i0 * w + tx.i1) * C, v01);
        ld8(b + ((size_t)ty.i1 * w + tx.i0) * C, v10);
        ld8(b + ((size_t)ty.i1 * w + tx.i1) * C, v11);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = ty.w0 * (tx.w0 * v00[e] + tx.w1 * v01[e]) + ty.w1 * (tx.w0 * v10[e] + tx.w1 * v11[e]);
        st8(dst + pix * dst_pitch + cg * 8, o);
    }
}

// adjoint in GATHER form: thread = (source pixel, 8 channels) sums w * d(dst) over the output pixels whose taps touch it --
// no atomics, fixed order. Candidate output rows of source row y: those between the images of y - 1 and y + 1.
__device__ __forceinline__ void adjoint_range(int y, int in_size, int out_size, float scale, bool align, int* lo, int* hi) {
    // dst index d maps to src = scale * (d + 0.5) - 0.5 (or scale * d): taps {floor(src), floor(src) + 1}
    const float inv = scale > 0.0f ? 1.0f / scale : 0.0f;
    float a, b;
    if (align) { a = (float)(y - 1) * inv; b = (float)(y + 1) * inv; }
    else { a = ((float)(y - 1) + 0.5f) * inv - 0.5f; b = ((float)(y + 1) + 0.5f) * inv - 0.5f; }
    int l = (int)floorf(a) - 1, h2 = (int)ceilf(b) + 1;
    if (y == 0) l = 0;                                  // (clamped sources: everything before the first tap lands on row 0)
    if (y == in_size - 1) h2 = out_size - 1;
    *lo = l < 0 ? 0 : l;
    *hi = h2 > out_size - 1 ? out_size - 1 : h2;
}

template <class T>
__global__ __launch_bounds__(256) void upsample_nhwc_bwd_kernel(const T* __restrict__ ddst, size_t dst_pitch, T* __restrict__ dsrc,
                                                                int N, int h, int w, int H, int W, int C, int align) {
    const int CG = C / 8;
    const size_t total = (size_t)N * h * w * CG;
    const float sy = bilin_scale(h, H, align != 0), sx = bilin_scale(w, W, align != 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const size_t pix = i / CG;
        const int x = (int)(pix % w);
        const size_t t = pix / w;
        const int y = (int)(t % h), n = (int)(t / h);
        int Y0, Y1, X0, X1;
        adjoint_range(y, h, H, sy, align != 0, &Y0, &Y1);
        adjoint_range(x, w, W, sx, align != 0, &X0, &X1);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int Y = Y0; Y <= Y1; ++Y) {
            const Tap ty = bilin_tap(Y, sy, h, align != 0);
            const float wy = (ty.i0 == y ? ty.w0 : 0.0f) + (ty.i1 == y ? ty.w1 : 0.0f);
            if (wy == 0.0f) continue;
            for (int X = X0; X <= X1; ++X) {
                const Tap tx = bilin_tap(X, sx, w, align != 0);
                const float wx = (tx.i0 == x ? tx.w0 : 0.0f) + (tx.i1 == x ? tx.w1 : 0.0f);
                if (wx == 0.0f) continue;
                float v[8];
                ld8(ddst + (((size_t)n * H + Y) * W + X) * dst_pitch + cg * 8, v);
                const float wgt = wy * wx;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, v[e], acc[e]);
            }
        }
        st8(dsrc + pix * C + cg * 8, acc);
    }
}

}  // namespace cms

using namespace cms;

extern "C" int cms_channel_copy(const void* src, size_t src_pitch, void* dst, size_t dst_pitch, size_t rows, int channels,
                                int dtype, size_t row_div, void* stream) {
    CMS_REQUIRE(src && dst && rows > 0 && channels > 0 && row_div >= 1, "channel_copy: NULL pointer / bad geometry");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "channel_copy: bad dtype");
    const size_t es = dtype == CMS_F32 ? 4 : 2;
    CMS_REQUIRE((channels * es) % 16 == 0 && (src_pitch * es) % 16 == 0 && (dst_pitch * es) % 16 == 0 &&
                    ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                "channel_copy: rows, pitches and base pointers must be multiples of 16 bytes");
    const int chunks = (int)(channels * es / 16);
    hipLaunchKernelGGL(channel_copy_kernel, dim3(grid_for(rows * chunks, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)src, src_pitch * es / 16, (uint4*)dst, dst_pitch * es / 16, rows, chunks, row_div);
    return launch_status("cms_channel_copy");
}

extern "C" int cms_add_n(const void* const* srcs, int k, void* dst, size_t n, int dtype, void* stream) {
    CMS_REQUIRE(srcs && dst && k >= 1 && k <= 6 && n > 0 && n % 8 == 0, "add_n: 1..6 sources, element count %% 8 == 0");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "add_n: bad dtype");
    const void* s[6];
    for (int i = 0; i < 6; ++i) s[i] = srcs[i < k ? i : 0];
    const dim3 grid(grid_for(n / 8, 256, 256 * 16));
    if (dtype == CMS_F32)
        hipLaunchKernelGGL(add_n_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)s[0], (const float*)s[1],
                           (const float*)s[2], (const float*)s[3], (const float*)s[4], (const float*)s[5], k, (float*)dst, n / 8);
    else
        hipLaunchKernelGGL(add_n_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)s[0],
                           (const uint16_t*)s[1], (const uint16_t*)s[2], (const uint16_t*)s[3], (const uint16_t*)s[4],
                           (const uint16_t*)s[5], k, (uint16_t*)dst, n / 8);
    return launch_status("cms_add_n");
}

extern "C" int cms_rows_reduce(const void* src, size_t pitch, int n, size_t rows_per_sample, int channels, int dtype, float* dst,
                               float scale, void* stream) {
    CMS_REQUIRE(src && dst && n > 0 && rows_per_sample > 0 && channels > 0 && channels % 8 == 0 && pitch % 8 == 0,
                "rows_reduce: NULL pointer / bad geometry (channels, pitch %% 8 == 0)");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "rows_reduce: bad dtype");
    const dim3 grid((unsigned)(n * ((channels + 63) / 64)));
    if (dtype == CMS_F32)
        hipLaunchKernelGGL(rows_reduce_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, pitch,
                           rows_per_sample, channels, dst, scale);
    else
        hipLaunchKernelGGL(rows_reduce_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, pitch,
                           rows_per_sample, channels, dst, scale);
    return launch_status("cms_rows_reduce");
}

extern "C" int cms_upsample_nhwc(const void* src, void* dst, size_t dst_pitch, int n, int h, int w, int H, int W, int channels,
                                 int dtype, int align_corners, int backward, void* stream) {
    CMS_REQUIRE(src && dst && n > 0 && h > 0 && w > 0 && H > 0 && W > 0 && channels > 0 && channels % 8 == 0 && dst_pitch % 8 == 0,
                "upsample_nhwc: NULL pointer / bad geometry (channels, pitch %% 8 == 0)");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "upsample_nhwc: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    // forward: src = (n,h,w,C) dense, dst = (n,H,W,.) rows of dst_pitch. backward: src = d(dst) rows of dst_pitch, dst = d(src) dense
    if (!backward) {
        const dim3 grid(grid_for((size_t)n * H * W * (channels / 8), 256, 256 * 16));
        if (dtype == CMS_F32)
            hipLaunchKernelGGL(upsample_nhwc_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, dst_pitch, n, h,
                               w, H, W, channels, align_corners);
        else
            hipLaunchKernelGGL(upsample_nhwc_fwd_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst,
                               dst_pitch, n, h, w, H, W, channels, align_corners);
    } else {
        const dim3 grid(grid_for((size_t)n * h * w * (channels / 8), 256, 256 * 16));
        if (dtype == CMS_F32)
            hipLaunchKernelGGL(upsample_nhwc_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)src, dst_pitch, (float*)dst, n, h,
                               w, H, W, channels, align_corners);
        else
            hipLaunchKernelGGL(upsample_nhwc_bwd_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)src, dst_pitch,
                               (uint16_t*)dst, n, h, w, H, W, channels, align_corners);
    }
    return launch_status("cms_upsample_nhwc");
}
