// Box-mask rasterisation and CutMix paste (gfx950).
//
//   cms_boxmask_rasterize   mask_gen.py:110-116 -- the (N,1,H,W) canvas the reference builds on the CPU and ships
//                           over PCIe every iteration (train_seg_semisup_mask_mt.py:331)
//   cms_cutmix_paste        train_seg_semisup_mask_mt.py:350-351, 363 (mix) and :389 (cut)
//
// HBM-bound streaming kernels: 16 B per lane per access (4 x f32 / 8 x bf16), tensor treated as one flat array so
// that vector accesses stay aligned for odd crop sizes (321 x 321), box membership evaluated per element from the
// N x n_boxes x 4 int32 range table (scalar-cache resident) -- the mask itself never exists in HBM.
// Algorithmic traffic: 2 reads + 1 write of the image tensor (9*P*s bytes for 3 channels) against 13*P*s for the
// reference's four elementwise kernels + mask.
#include "common.hpp"

namespace cms {

template <class T> struct VecOf;
template <> struct VecOf<float> { using type = float4; static constexpr int N = 4; };
template <> struct VecOf<uint16_t> { using type = uint4; static constexpr int N = 8; };

struct Cursor {
    int n, c, y, x;
};

__device__ __forceinline__ Cursor decode(size_t i, int C, int H, int W) {
    Cursor k;
    k.x = (int)(i % W);
    size_t t = i / W;
    k.y = (int)(t % H);
    t /= H;
    k.c = (int)(t % C);
    k.n = (int)(t / C);
    return k;
}

__device__ __forceinline__ void advance(Cursor& k, int C, int H, int W) {
    if (++k.x == W) {
        k.x = 0;
        if (++k.y == H) {
            k.y = 0;
            if (++k.c == C) {
                k.c = 0;
                ++k.n;
            }
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void paste_ranges_kernel(const T* __restrict__ x0, const T* __restrict__ x1,
                                                           T* __restrict__ out, const int32_t* __restrict__ ranges,
                                                           int nb, int invert, int C, int H, int W, size_t total) {
    using V = typename VecOf<T>::type;
    constexpr int VN = VecOf<T>::N;
    const size_t nvec = total / VN;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = v * VN;
        V a, b;
        b = reinterpret_cast<const V*>(x1)[v];
        if (x0) a = reinterpret_cast<const V*>(x0)[v];
        else memset(&a, 0, sizeof(V));
        T ea[VN], eb[VN], eo[VN];
        memcpy(ea, &a, sizeof(V));
        memcpy(eb, &b, sizeof(V));
        Cursor k = decode(i0, C, H, W);
#pragma unroll
        for (int e = 0; e < VN; ++e) {
            const bool m = box_mask_bit(ranges + (size_t)k.n * nb * 4, nb, k.y, k.x, invert != 0);
            eo[e] = m ? eb[e] : ea[e];
            advance(k, C, H, W);
        }
        V o;
        memcpy(&o, eo, sizeof(V));
        reinterpret_cast<V*>(out)[v] = o;
    }
    // tail (< VN elements)
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(total - nvec * VN)) {
        const size_t i = nvec * VN + threadIdx.x;
        Cursor k = decode(i, C, H, W);
        const bool m = box_mask_bit(ranges + (size_t)k.n * nb * 4, nb, k.y, k.x, invert != 0);
        out[i] = m ? x1[i] : (x0 ? x0[i] : (T)0);
    }
}

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(uint16_t v) { return bf16_to_f32(v); }
__device__ __forceinline__ void from_f(float f, float& o) { o = f; }
__device__ __forceinline__ void from_f(float f, uint16_t& o) { o = f32_to_bf16(f); }

// arithmetic form with a materialised mask: out = x0*(1-m) + x1*m, evaluated in fp32 exactly as the reference's
// elementwise ops do for fp32 tensors
template <class T>
__global__ __launch_bounds__(256) void paste_mask_kernel(const T* __restrict__ x0, const T* __restrict__ x1,
                                                         T* __restrict__ out, const float* __restrict__ mask, int C,
                                                         int H, int W, size_t total) {
    const size_t hw = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / (hw * C);
        const float m = mask[n * hw + i % hw];
        const float a = x0 ? to_f(x0[i]) : 0.0f;
        const float b = to_f(x1[i]);
        float r;
        {
#pragma clang fp contract(off)
            const float t0 = a * (1.0f - m);
            const float t1 = b * m;
            r = t0 + t1;
        }
        from_f(r, out[i]);
    }
}

__global__ __launch_bounds__(256) void rasterize_kernel(const int32_t* __restrict__ ranges, int nb, int invert,
                                                        int H, int W, size_t total, float* __restrict__ mask) {
    const size_t nvec = total / 4;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        Cursor k = decode(v * 4, 1, H, W);
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[j] = box_mask_bit(ranges + (size_t)k.n * nb * 4, nb, k.y, k.x, invert != 0) ? 1.0f : 0.0f;
            advance(k, 1, H, W);
        }
        reinterpret_cast<float4*>(mask)[v] = make_float4(e[0], e[1], e[2], e[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(total - nvec * 4)) {
        const size_t i = nvec * 4 + threadIdx.x;
        Cursor k = decode(i, 1, H, W);
        mask[i] = box_mask_bit(ranges + (size_t)k.n * nb * 4, nb, k.y, k.x, invert != 0) ? 1.0f : 0.0f;
    }
}

}  // namespace cms

using namespace cms;

extern "C" int cms_boxmask_rasterize(const int32_t* ranges, int n, int n_boxes, int h, int w, int invert,
                                     float* mask_out, void* stream) {
    CMS_REQUIRE(ranges && mask_out, "boxmask_rasterize: NULL pointer");
    CMS_REQUIRE(n > 0 && n_boxes >= 0 && h > 0 && w > 0, "boxmask_rasterize: bad geometry");
    const size_t total = (size_t)n * h * w;
    hipLaunchKernelGGL(rasterize_kernel, dim3(grid_for(total / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, ranges,
                       n_boxes, invert, h, w, total, mask_out);
    return launch_status("cms_boxmask_rasterize");
}

extern "C" int cms_cutmix_paste(const void* x0, const void* x1, void* out, int dtype, const int32_t* ranges, int n,
                                int n_boxes, int c, int h, int w, int invert, void* stream) {
    CMS_REQUIRE(x1 && out && ranges, "cutmix_paste: NULL pointer");
    CMS_REQUIRE(n > 0 && n_boxes >= 0 && c > 0 && h > 0 && w > 0, "cutmix_paste: bad geometry");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "cutmix_paste: unknown dtype %d", dtype);
    CMS_REQUIRE(((uintptr_t)x0 % 16 == 0) && ((uintptr_t)x1 % 16 == 0) && ((uintptr_t)out % 16 == 0),
                "cutmix_paste: tensors must be 16-byte aligned");
    const size_t total = (size_t)n * c * h * w;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CMS_F32) {
        hipLaunchKernelGGL(paste_ranges_kernel<float>, dim3(grid_for(total / 4 + 1, 256)), dim3(256), 0, s,
                           (const float*)x0, (const float*)x1, (float*)out, ranges, n_boxes, invert, c, h, w, total);
    } else {
        hipLaunchKernelGGL(paste_ranges_kernel<uint16_t>, dim3(grid_for(total / 8 + 1, 256)), dim3(256), 0, s,
                           (const uint16_t*)x0, (const uint16_t*)x1, (uint16_t*)out, ranges, n_boxes, invert, c, h, w,
                           total);
    }
    return launch_status("cms_cutmix_paste");
}

extern "C" int cms_cutmix_paste_mask(const void* x0, const void* x1, void* out, int dtype, const float* mask, int n,
                                     int c, int h, int w, void* stream) {
    CMS_REQUIRE(x1 && out && mask, "cutmix_paste_mask: NULL pointer");
    CMS_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "cutmix_paste_mask: bad geometry");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "cutmix_paste_mask: unknown dtype %d", dtype);
    const size_t total = (size_t)n * c * h * w;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CMS_F32) {
        hipLaunchKernelGGL(paste_mask_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, s, (const float*)x0,
                           (const float*)x1, (float*)out, mask, c, h, w, total);
    } else {
        hipLaunchKernelGGL(paste_mask_kernel<uint16_t>, dim3(grid_for(total, 256)), dim3(256), 0, s,
                           (const uint16_t*)x0, (const uint16_t*)x1, (uint16_t*)out, mask, c, h, w, total);
    }
    return launch_status("cms_cutmix_paste_mask");
}
