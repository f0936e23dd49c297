// Trainable BatchNorm affine over FROZEN statistics (DeepLab v3+'s torchvision backbone: architectures/deeplab3plus.py:96-98 keeps
// gamma / beta trainable, freeze_batchnorm() only freezes the statistics) without the weight-gradient kernel's side outputs.
//
// With y = conv(x, W) * s + t, s = gamma * rstd, t = beta - mean * s and dU = dL/dy:
//     d(beta)  = sum_p dU[p][co]
//     d(gamma) = (<W[co], G[co]> - mean * d(beta)) * rstd,      G = sum_p dU x   (the UNSCALED weight gradient)
//     d(W)     = s[co] * G[co]
// Rounds 2-5 took <W, G> and sum_p dU out of the 128 x 128 weight-gradient kernel as side outputs, which kept the wide layers of
// this backbone off the eight-phase kernel (wgrad8: three times the work per CU and microsecond) -- 23 % of the v3+ step's GPU time.
// Here the eight-phase kernel writes G unscaled into a scratch gradient, `channel_sum` takes sum_p dU, and ONE finishing launch per
// backward pass goes over the scratch: grad += s * G, wdot[co] = <W[co], G[co]> (one workgroup per output channel: no atomics, no
// division by s), scratch cleared for the next pass.
#include "common.hpp"

namespace cms {

__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8f(const uint16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = bf16_to_f32((uint16_t)(u.x & 0xffffu)); v[1] = bf16_to_f32((uint16_t)(u.x >> 16));
    v[2] = bf16_to_f32((uint16_t)(u.y & 0xffffu)); v[3] = bf16_to_f32((uint16_t)(u.y >> 16));
    v[4] = bf16_to_f32((uint16_t)(u.z & 0xffffu)); v[5] = bf16_to_f32((uint16_t)(u.z >> 16));
    v[6] = bf16_to_f32((uint16_t)(u.w & 0xffffu)); v[7] = bf16_to_f32((uint16_t)(u.w >> 16));
}

// dst[c] += sum_rows src[row][c]: block = (slab of rows, TILE-channel tile); LPR = TILE / 8 lanes side by side read one row's
// TILE channels (16 bytes / 32 bytes per lane: 512 contiguous bytes per row of a 256-channel bf16 tile), 256 / LPR rows per pass,
// four passes in flight; one fp32 atomic per block and channel
template <class T, int TILE>
__global__ __launch_bounds__(256) void channel_sum_kernel(const T* __restrict__ src, size_t rows, int C, float* __restrict__ dst, int slabs) {
    constexpr int LPR = TILE / 8, RPP = 256 / LPR;           // lanes per row, rows per pass
    __shared__ float red[RPP][TILE + 1];
    const int tiles = C / TILE;
    const int slab = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int cgl = threadIdx.x % LPR, slot = threadIdx.x / LPR;
    const size_t r0 = rows * (size_t)slab / (size_t)slabs, r1 = rows * (size_t)(slab + 1) / (size_t)slabs;
    const T* base = src + (size_t)tile * TILE + cgl * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    size_t p = r0 + slot;
    for (; p + 3 * RPP < r1; p += 4 * RPP) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) ld8f(base + (p + RPP * u) * (size_t)C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[u][e];
    }
    for (; p < r1; p += RPP) {
        float v[8];
        ld8f(base + p * (size_t)C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[slot][cgl * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < TILE) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < RPP; ++k) s += red[k][threadIdx.x];
        atomicAdd(dst + tile * TILE + threadIdx.x, s);
    }
}

// block = one output channel `co` of one item: its ntaps rows of cin elements in the [tap][cout][cin] layout
__global__ __launch_bounds__(256) void wgrad_finish_batch_kernel(const cms_wfinish_item* __restrict__ items, int n_items) {
    __shared__ float red[4];
    int it = 0;
    while (it + 1 < n_items && (int)blockIdx.x >= items[it + 1].first_block) ++it;      // (a few dozen items: linear scan of scalars)
    const cms_wfinish_item m = items[it];
    const int co = (int)blockIdx.x - m.first_block;
    const float s = m.scale ? m.scale[co] : 1.0f;
    float acc = 0.0f;
    for (int tap = 0; tap < m.ntaps; ++tap) {
        const size_t base = ((size_t)tap * m.cout + co) * m.cin;
        for (int ci = threadIdx.x * 4; ci < m.cin; ci += 256 * 4) {
            float4* gp = reinterpret_cast<float4*>(m.scratch + base + ci);
            const float4 g = *gp;
            const uint2 wv = *reinterpret_cast<const uint2*>(m.w + base + ci);
            acc += g.x * bf16_to_f32((uint16_t)(wv.x & 0xffffu)) + g.y * bf16_to_f32((uint16_t)(wv.x >> 16)) +
                   g.z * bf16_to_f32((uint16_t)(wv.y & 0xffffu)) + g.w * bf16_to_f32((uint16_t)(wv.y >> 16));
            float4* dp = reinterpret_cast<float4*>(m.grad + base + ci);
            float4 d = *dp;
            d.x += s * g.x; d.y += s * g.y; d.z += s * g.z; d.w += s * g.w;
            *dp = d;
            *gp = float4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) m.wdot[co] += red[0] + red[1] + red[2] + red[3];
}

}  // namespace cms

using namespace cms;

extern "C" int cms_channel_sum(const void* src, int dtype, size_t rows, int channels, float* dst, void* stream) {
    CMS_REQUIRE(src && dst && rows > 0 && channels > 0 && channels % 64 == 0, "channel_sum: NULL pointer / channels %% 64 != 0");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "channel_sum: bad dtype");
    const int tile = channels % 256 == 0 ? 256 : 64;            // (the wide layers: 512 contiguous bytes per row and pass)
    const int tiles = channels / tile;
    size_t want = rows / 256;                                   // >= 256 rows per slab
    int slabs = (int)(want < 1 ? 1 : want);
    const int cap = 1024 / tiles > 1 ? 1024 / tiles : 1;        // ~1 k workgroups per launch
    if (slabs > cap) slabs = cap;
    const dim3 grid((unsigned)(slabs * tiles));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CMS_F32 && tile == 256)
        hipLaunchKernelGGL((channel_sum_kernel<float, 256>), grid, dim3(256), 0, s, (const float*)src, rows, channels, dst, slabs);
    else if (dtype == CMS_F32)
        hipLaunchKernelGGL((channel_sum_kernel<float, 64>), grid, dim3(256), 0, s, (const float*)src, rows, channels, dst, slabs);
    else if (tile == 256)
        hipLaunchKernelGGL((channel_sum_kernel<uint16_t, 256>), grid, dim3(256), 0, s, (const uint16_t*)src, rows, channels, dst, slabs);
    else
        hipLaunchKernelGGL((channel_sum_kernel<uint16_t, 64>), grid, dim3(256), 0, s, (const uint16_t*)src, rows, channels, dst, slabs);
    return launch_status("cms_channel_sum");
}

// host: fills first_block of every item; -> total blocks (= sum of cout) or a negative error code
extern "C" int cms_wgrad_finish_pack(cms_wfinish_item* items, int n_items) {
    CMS_REQUIRE(items && n_items > 0, "wgrad_finish_pack: no items");
    int total = 0;
    for (int i = 0; i < n_items; ++i) {
        cms_wfinish_item& m = items[i];
        CMS_REQUIRE(m.scratch && m.grad && m.w && m.wdot && m.ntaps > 0 && m.cout > 0 && m.cin > 0 && m.cin % 4 == 0,
                    "wgrad_finish_pack: item %d: NULL pointer / cin %% 4 != 0", i);
        m.first_block = total;
        total += m.cout;
    }
    return total;
}

extern "C" int cms_wgrad_finish_run(const void* items_dev, int n_items, int total_blocks, void* stream) {
    CMS_REQUIRE(items_dev && n_items > 0 && total_blocks > 0, "wgrad_finish_run: bad arguments");
    hipLaunchKernelGGL(wgrad_finish_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const cms_wfinish_item*)items_dev,
                       n_items);
    return launch_status("cms_wgrad_finish_run");
}
