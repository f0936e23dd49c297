// Shared host/device plumbing for libcutmixseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/cutmixseg.h"
#include "pixel_math.hpp"

namespace cms {

void set_error(const char* fmt, ...);

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return CMS_ELAUNCH;
    }
    return CMS_OK;
}

#define CMS_REQUIRE(cond, ...)         \
    do {                               \
        if (!(cond)) {                 \
            cms::set_error(__VA_ARGS__); \
            return CMS_EINVAL;         \
        }                              \
    } while (0)

constexpr int kWave = 64;

// sum over the 64 lanes of a wavefront; result valid in lane 0
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Block-wide sum of K values per thread (blockDim.x multiple of 64, <= 1024). Result valid in thread 0.
template <int K>
__device__ __forceinline__ void block_sum(float (&v)[K], float* smem /* >= K*16 floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) smem[k * 16 + wid] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = 0.0f;
            for (int i = 0; i < nw; ++i) s += smem[k * 16 + i];
            v[k] = s;
        }
    }
}

// csrc/conv8.hip: the eight-phase 256 x 256 convolution (mode 0: one whole tile per workgroup, 1: persistent + stream-K)
size_t conv8_workspace_bytes(int n_cu);
bool conv8_supported(const cms_conv_desc* d);
int conv8_launch(const cms_conv_desc* d, hipStream_t s, int mode, int grid_cap, void* trace, int trace_wgs);
// csrc/wgrad8.hip: the eight-phase 256 x 256 weight gradient
bool wgrad8_supported(const cms_wgrad_desc* d);
int wgrad8_plan(const cms_wgrad_desc* d, int* kt_per_slice);       // -> number of pixel slices
int wgrad8_launch(const cms_wgrad_desc* d, hipStream_t s, void* trace, int trace_wgs);
// csrc/conv.hip: dw += the slices of a split-K slab, in slice order
void wgrad_reduce_launch(const float* slab, float* dw, int ksplit, size_t slice_elems, int ntaps, int cout, int cin, int cout_real,
                         int dw_cout, hipStream_t s);

inline int grid_for(size_t work_items, int block, int max_blocks = 256 * 8) {
    size_t b = (work_items + block - 1) / block;
    if (b > (size_t)max_blocks) b = max_blocks;
    if (b < 1) b = 1;
    return (int)b;
}

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace cms
