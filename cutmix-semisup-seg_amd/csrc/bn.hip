// Batch-statistics BatchNorm (+ ReLU, + residual) for NHWC activations: forward, backward, and the hooks a synchronised
// (data-parallel) BatchNorm needs.
//
// The reference's networks run nn.BatchNorm2d in TRAINING mode wherever `--freeze_bn` does not reach: every BatchNorm of
// DeepLab v2 without the flag (architectures/deeplab2.py:72-84 freeze the affine parameters only), the DeepLab v3+ head
// always (architectures/deeplab3plus.py:40-64, 120-121), the U-Nets. Round 1 left those layers to the library.
//
//   forward    sums[c] = (sum_p x, sum_p x^2)             bn_reduce_kernel<.., 0>   (per-thread fp32, block tree in LDS,
//                                                                                    fp64 atomics across blocks)
//              [data parallel: all-reduce sums and the pixel count -- SURVEY.md 8(e), "BN statistics"]
//              mean, var -> scale = gamma * rstd, shift = beta - mean * scale; running statistics   bn_finalize_kernel
//              y = relu(x * scale + shift (+ residual))                                             bn_apply_kernel
//   backward   dy' = dy * [y > 0];  sums[c] = (sum_p dy', sum_p dy' * xhat)                         bn_reduce_kernel<.., 1>
//              [data parallel: all-reduce]
//              dx = gamma * rstd * (dy' - mean(dy') - xhat * mean(dy' * xhat));  dgamma, dbeta      bn_bwd_apply_kernel
//
// HBM-bound: forward reads x twice and writes y once (2 passes, statistics are a global dependency), backward reads
// (dy, x, y) twice and writes dx once. Threads own 8 consecutive channels of a pixel (16-byte bf16 / 32-byte fp32
// vectors, coalesced along the channel axis); the per-channel reduction over pixels is per-thread accumulation, then a
// cross-lane tree (__shfl_xor over the lanes of a wave that hold the same channel group) and LDS across waves.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"

namespace cms {

template <class T>
struct Vec8 {
    float v[8];
};

__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const uint16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4*>(p + 4) = float4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(uint16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
}

// [stored value > 0] of 8 outputs as one byte (bit e = channel e of the vector): the ReLU mask the backward kernels test, taken
// from what store8 writes (a positive fp32 below bf16's smallest subnormal is stored as 0)
__device__ __forceinline__ unsigned stored_positive_bits(const float*, const float (&v)[8]) {
    unsigned b = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) b |= (v[e] > 0.0f ? 1u : 0u) << e;
    return b;
}
__device__ __forceinline__ unsigned stored_positive_bits(const uint16_t*, const float (&v)[8]) {
    unsigned b = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) b |= ((int16_t)f32_to_bf16(v[e]) > 0 ? 1u : 0u) << e;
    return b;
}

// MODE 0: (sum x, sum x^2).  MODE 1: (sum dy', sum dy' * xhat) with dy' = dy * [y > 0] (y == nullptr: no ReLU)
template <class T, int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const T* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, double* __restrict__ sums,
                                                        size_t P, int C, int pitch) {
    // C = channels of THIS launch (<= 2048: layers wider than that -- DenseNet-161's transition3 / denseblock4 / norm5 with
    // 2112 .. 2208 channels -- are reduced as channel slices, one launch each: x, dy, y, mean, rstd and sums arrive
    // offset by the slice's first channel); pitch = channels per pixel row of the tensors = plane stride of `sums`
    extern __shared__ float red[];                    // [slots][CG][16]
    const int CG = C / 8;
    const int slots = CG >= 256 ? 1 : 256 / CG;       // pixels handled side by side by one block
    const int tid = threadIdx.x;
    const int cg = tid % CG, slot = tid / CG;
    const bool active = slot < slots;
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.0f;
    {
        float mu[8], rs[8];
        if (MODE == 1 && active) {
            load8(mean + cg * 8, mu);
            load8(rstd + cg * 8, rs);
        }
        if (active) {
            // four pixels per iteration: four independent 16-byte loads per tensor in flight per thread (one was latency-bound:
            // 1 TB/s on the 34 MB activations of layer3, profiles/r03r_*)
            const size_t stride = (size_t)gridDim.x * slots;
            auto accumulate = [&](const float (&xv)[8], const float (&dv)[8], const float (&yv)[8]) {
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { a0[e] += xv[e]; a1[e] = fmaf(xv[e], xv[e], a1[e]); }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = (y == nullptr || yv[e] > 0.0f) ? dv[e] : 0.0f;
                        a0[e] += d;
                        a1[e] = fmaf(d, (xv[e] - mu[e]) * rs[e], a1[e]);
                    }
                }
            };
            size_t p = (size_t)blockIdx.x * slots + slot;
            for (; p + 3 * stride < P; p += 4 * stride) {
                float xv[4][8], dv[4][8], yv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t o = (p + u * stride) * (size_t)pitch + (size_t)cg * 8;
                    load8(x + o, xv[u]);
                    if (MODE == 1) {
                        load8(dy + o, dv[u]);
                        if (y) load8(y + o, yv[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) accumulate(xv[u], dv[u], yv[u]);
            }
            for (; p < P; p += stride) {
                const size_t o = p * (size_t)pitch + (size_t)cg * 8;
                float xv[8], dv[8], yv[8];
                load8(x + o, xv);
                if (MODE == 1) {
                    load8(dy + o, dv);
                    if (y) load8(y + o, yv);
                }
                accumulate(xv, dv, yv);
            }
        }
        // lanes of a wave that hold the same channel group: CG divides 64 -> tree over the lane bits above log2(CG)
        if (CG < 64 && (64 % CG) == 0) {
            for (int off = 32; off >= CG; off >>= 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    a0[e] += __shfl_xor(a0[e], off, 64);
                    a1[e] += __shfl_xor(a1[e], off, 64);
                }
            }
        }
        if (active) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(slot * CG + cg) * 16 + e] = a0[e];
                red[(slot * CG + cg) * 16 + 8 + e] = a1[e];
            }
        }
        __syncthreads();
        // one thread per (channel group, value): sum the slots that are distinct after the wave tree
        const int step = (CG < 64 && (64 % CG) == 0) ? 64 / CG : 1;       // slots inside one wave hold the same sum
        for (int i = tid; i < CG * 16; i += 256) {
            const int g = i / 16, e = i % 16;
            float s = 0.0f;
            for (int sl = 0; sl < slots; sl += step) s += red[(sl * CG + g) * 16 + e];
            const int c = g * 8 + (e & 7);
            atomicAdd(sums + (size_t)(e >> 3) * pitch + c, (double)s);
        }
    }
}

// what one channel's statistics turn into (shared by bn_finalize_kernel and the fused tail of bn_reduce_tiled_kernel)
struct BnFin {
    const float* gamma;
    const float* beta;
    float* mean;
    float* rstd;
    float* scale;
    float* shift;
    float* running_mean;
    float* running_var;
    long long* counter;
    double count;
    float eps, momentum;
};

__device__ __forceinline__ void bn_finalize_channel(const BnFin& f, int c, double s0, double s1) {
    const double m = s0 / f.count;
    double var = s1 / f.count - m * m;
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)f.eps));
    f.mean[c] = (float)m;
    f.rstd[c] = r;
    const float g = f.gamma ? f.gamma[c] : 1.0f, b = f.beta ? f.beta[c] : 0.0f;
    f.scale[c] = g * r;
    f.shift[c] = b - (float)m * g * r;
    if (f.running_mean) f.running_mean[c] = (1.0f - f.momentum) * f.running_mean[c] + f.momentum * (float)m;
    if (f.running_var) {
        const double unbiased = f.count > 1.0 ? var * f.count / (f.count - 1.0) : var;
        f.running_var[c] = (1.0f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
    }
}

// Round 3: the reduction WITHOUT data atomics. bn_reduce_kernel ends every block with 2 C fp64 atomics; inside the training
// step they queue behind the weight gradients' 2.5 GB of fp32 atomics in the memory-side atomic units (52 / 58 us per launch in
// the step against 10-22 us alone, profiles/r03t_*, r03bn_*). Here a block owns a CHANNEL TILE of 64 channels (one 128-byte line
// per pixel row in bf16), one sample GROUP and one of S pixel splits of that group, stores its 128 partial sums, and the LAST
// block of a tile to arrive (one counter atomic per block, self-resetting) adds the partials of every group in fixed order in
// fp64 -- bit-reproducible -- and either writes sums[] (the data-parallel protocol all-reduces them) or, FIN, finalises its 64
// channels on the spot (no bn_finalize launch).
//   GROUPS (gridDim.y): the pixel rows are G equal runs of consecutive samples whose statistics are kept APART -- one launch
//   normalises the supervised and the mixed batch of the student (or the teacher's two batches) exactly as the reference's
//   separate forward passes do (train_seg_semisup_mask_mt.py:296-358), the running statistics moving once per group, in order.
//   Layouts: mean / rstd / scale / shift [G][C], sums [G][2][C].
//   ws: unsigned counters[tiles] (zero before the first use) padded to 256 bytes, then float part[tiles][G][S][2][64]
constexpr int BN_CT = 64;            // channels per tile

// 16-byte agent-coherent load (global_load_dwordx4 sc1: served past the non-coherent L2 lines of other XCDs)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "bn_reduce_tiled_kernel's write-through / sc1 hand-off is written for gfx942 / gfx950 (sc1 cache-policy bit); build with --offload-arch=gfx950"
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load_sc1_x4(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <class T, int MODE, bool FIN>
__global__ __launch_bounds__(256) void bn_reduce_tiled_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                              const T* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, double* __restrict__ sums,
                                                              unsigned* counters, float* part,
                                                              size_t Pg, int C, int tiles, int S, BnFin fin, int fenced,
                                                              const uint8_t* __restrict__ bits) {
    __shared__ float red[4][2 * BN_CT];
    __shared__ double redd[8][2 * BN_CT];
    __shared__ int last_flag;
    const int tid = threadIdx.x;
    const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
    const int G = gridDim.y, grp = blockIdx.y;
    const int cgl = tid & 7, slot = tid >> 3;          // 8 channel groups of 8 channels x 32 pixel rows side by side
    const int c0 = tile * BN_CT + cgl * 8;
    const bool active = c0 < C;
    float a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.0f;
    if (active) {
        float mu[8], rs[8];
        if (MODE == 1) {
            load8(mean + (size_t)grp * C + c0, mu);
            load8(rstd + (size_t)grp * C + c0, rs);
        }
        // the ReLU mask: the stored output y, or (round 5) the BITS bn_apply wrote beside it -- one byte per 8-channel vector,
        // 1/16 of the bytes of y ([pixel rows][C / 8], bit e = channel e of the vector)
        auto accumulate = [&](const float (&xv)[8], const float (&dv)[8], const float (&yv)[8], unsigned mb) {
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { a0[e] += xv[e]; a1[e] = fmaf(xv[e], xv[e], a1[e]); }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool on = bits ? ((mb >> e) & 1u) != 0 : (y == nullptr || yv[e] > 0.0f);
                    const float d = on ? dv[e] : 0.0f;
                    a0[e] += d;
                    a1[e] = fmaf(d, (xv[e] - mu[e]) * rs[e], a1[e]);
                }
            }
        };
        const size_t base = (size_t)grp * Pg * (size_t)C + (size_t)c0;
        const uint8_t* brow = bits ? bits + (size_t)grp * Pg * (size_t)(C >> 3) + (size_t)(c0 >> 3) : nullptr;
        const size_t bstride = (size_t)(C >> 3);
        const size_t stride = (size_t)S * 32;
        size_t p = (size_t)split * 32 + slot;
        for (; p + 3 * stride < Pg; p += 4 * stride) {
            float xv[4][8], dv[4][8], yv[4][8];
            unsigned mb[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = base + (p + u * stride) * (size_t)C;
                load8(x + o, xv[u]);
                if (MODE == 1) {
                    load8(dy + o, dv[u]);
                    if (bits) mb[u] = brow[(p + u * stride) * bstride];
                    else if (y) load8(y + o, yv[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accumulate(xv[u], dv[u], yv[u], mb[u]);
        }
        for (; p < Pg; p += stride) {
            const size_t o = base + p * (size_t)C;
            float xv[8], dv[8], yv[8];
            unsigned mb = 0u;
            load8(x + o, xv);
            if (MODE == 1) {
                load8(dy + o, dv);
                if (bits) mb = brow[p * bstride];
                else if (y) load8(y + o, yv);
            }
            accumulate(xv, dv, yv, mb);
        }
    }
    // the 8 pixel rows of a wave that share a channel group (lane bits 3..5), then the 4 waves through LDS
#pragma unroll
    for (int off = 32; off >= 8; off >>= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a0[e] += __shfl_xor(a0[e], off, 64);
            a1[e] += __shfl_xor(a1[e], off, 64);
        }
    }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[tid >> 6][cgl * 8 + e] = a0[e];
            red[tid >> 6][BN_CT + cgl * 8 + e] = a1[e];
        }
    }
    __syncthreads();
    // The partials travel through AGENT-SCOPE relaxed atomic stores / `sc1` loads (write-through, L2-bypassing accesses): no
    // __threadfence(), whose release / acquire halves are a writeback / invalidate of the whole XCD L2 on this part -- with
    // them the kernel took 26-93 us alone where the loads need 5-20 (tools/bn_bench.py, profiles/r03bn_*). An explicit
    // s_waitcnt vmcnt(0) (the write-through stores have been acknowledged; a workgroup-scope release fence emits nothing on
    // this target) and the barrier order them before the counter add.
    float* mine = part + (((size_t)tile * G + grp) * S + split) * (2 * BN_CT);
    if (tid < 2 * BN_CT)
        __hip_atomic_store(mine + tid, (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        // `fenced` (CMS_BN_FENCE=1): the textbook release / acquire pair of the HIP memory model on top of the write-through
        // protocol -- an agent-scope release before the ticket (a writeback of the XCD's L2: 26-93 us per launch alone,
        // profiles/r03bn_fenced_variant.log), an agent-scope acquire in the last block. The default relies on what
        // cdna_hip_programming.md (Guideline 16, R1) documents for this part: sc1 stores + drained vmcnt + relaxed agent
        // ticket, sc1 loads on the reading side. tools/bn_stress.py and tests/test_gpu_bn.py run both.
        if (fenced) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned prev = __hip_atomic_fetch_add(counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = prev == (unsigned)(G * S - 1);
        if (fenced && last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!last_flag) return;
    // the tile is complete: per group, S rows of 128 floats; 32 lanes x 16 bytes cover a row, 8 rows side by side, up to 8 loads
    // in flight per thread; fixed order: rows rg, rg + 8, ... per thread, then the 8 row groups
    const int q = tid & 31, rg = tid >> 5;
    for (int g = 0; g < G; ++g) {
        const float* pp = part + ((size_t)tile * G + g) * S * (2 * BN_CT) + q * 4;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        int r = rg;
        for (; r + 56 < S; r += 64) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = load_sc1_x4(pp + (size_t)(r + 8 * u) * (2 * BN_CT));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]),
                         "+v"(v[7]) :: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[0] += (double)v[u].x; acc[1] += (double)v[u].y; acc[2] += (double)v[u].z; acc[3] += (double)v[u].w;
            }
        }
        for (; r < S; r += 8) {
            f32x4 v = load_sc1_x4(pp + (size_t)r * (2 * BN_CT));
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
            acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
        }
        __syncthreads();                                // (previous group's readers of redd are done)
#pragma unroll
        for (int e = 0; e < 4; ++e) redd[rg][q * 4 + e] = acc[e];
        __syncthreads();
        if (tid < BN_CT) {
            const int c = tile * BN_CT + tid;
            if (c < C) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) { s0 += redd[k][tid]; s1 += redd[k][BN_CT + tid]; }
                if (sums) { sums[(size_t)g * 2 * C + c] = s0; sums[(size_t)g * 2 * C + C + c] = s1; }
                if (FIN) {
                    BnFin f = fin;                      // this group's outputs; the running statistics move once per group, in order
                    f.mean += (size_t)g * C; f.rstd += (size_t)g * C; f.scale += (size_t)g * C; f.shift += (size_t)g * C;
                    bn_finalize_channel(f, c, s0, s1);
                }
            }
        }
    }
    if (tid == 0) {
        __hip_atomic_store(counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        if (FIN && tile == 0 && fin.counter) *fin.counter += G;
    }
}

// sums (sum x, sum x^2) over `count` pixels -> mean, rstd, scale, shift; running statistics like nn.BatchNorm2d (momentum,
// unbiased variance)
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ mean,
                                   float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, int C,
                                   double* __restrict__ clear_a, double* __restrict__ clear_b, long long* __restrict__ counter) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && counter) *counter += 1;             // nn.BatchNorm2d.num_batches_tracked
    if (c >= C) return;
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    rstd[c] = r;
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    scale[c] = g * r;
    shift[c] = b - (float)m * g * r;
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)m;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    // recorded passes (cms_program_add_bn): leave the sum buffers of this unit zeroed for their next use -- the forward sums
    // just consumed (every thread has read its own two entries above) and the sums of the unit's backward pass
    if (clear_a) { clear_a[c] = 0.0; clear_a[C + c] = 0.0; }
    if (clear_b) { clear_b[c] = 0.0; clear_b[C + c] = 0.0; }
}

// gridDim.y = sample groups (scale / shift / mean / rstd [G][C], sums [G][2][C]); P = pixel rows of ONE group
template <class T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int relu, size_t P, int C, uint8_t* __restrict__ bits_out) {
    const int CG = C / 8;
    const size_t total = P * CG;
    const size_t gbase = (size_t)blockIdx.y * total * 8;
    if (bits_out) bits_out += (size_t)blockIdx.y * total;
    scale += (size_t)blockIdx.y * C;
    shift += (size_t)blockIdx.y * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const size_t o = gbase + i * 8;
        float xv[8], sc[8], sh[8], rv[8];
        load8(x + o, xv);
        load8(scale + cg * 8, sc);
        load8(shift + cg * 8, sh);
        if (res) load8(res + o, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = fmaf(xv[e], sc[e], sh[e]);
            if (res) v += rv[e];
            xv[e] = relu ? fmaxf(v, 0.0f) : v;
        }
        store8(y + o, xv);
        if (bits_out) bits_out[i] = (uint8_t)stored_positive_bits(y, xv);      // consecutive lanes: consecutive bytes
    }
}

// dx = gamma * rstd * (dy' - sum_dy / n - xhat * sum_dyxhat / n); dres = dy' (gradient of the residual branch, optional)
// The host picks a grid whose thread count is a multiple of C / 8, so a thread keeps ITS channel group over the whole loop and
// loads the per-channel coefficients once: the per-element version fetched sums[c] as 16 scattered 8-byte loads per 16-byte
// vector -- 64 cache lines per wave instruction against the 24 lines of x, dy, y -- and ran at 1.7-1.9 TB/s where bn_apply
// reaches 5.3-6 (tools/bn_bench.py, profiles/r03bn_*: 160 -> 56 us on 16810 x 2048).
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const T* __restrict__ y, T* __restrict__ dx, T* __restrict__ dres,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums,
                                                           double count, size_t P, int C, const uint8_t* __restrict__ bits) {
    const int CG = C / 8;
    const size_t total = P * CG;
    const size_t gbase = (size_t)blockIdx.y * total * 8;
    if (bits) bits += (size_t)blockIdx.y * total;
    mean += (size_t)blockIdx.y * C;
    rstd += (size_t)blockIdx.y * C;
    sums += (size_t)blockIdx.y * 2 * C;
    const size_t stride = (size_t)gridDim.x * blockDim.x;          // multiple of CG (cms_bn_bwd_apply)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = (int)(i % CG);
    float mu[8], rs[8], grs[8], m1[8], m2[8];
    {
        const double inv_count = 1.0 / count;
        load8(mean + cg * 8, mu);
        load8(rstd + cg * 8, rs);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cg * 8 + e;
            m1[e] = (float)(sums[c] * inv_count);
            m2[e] = (float)(sums[C + c] * inv_count);
            grs[e] = (gamma ? gamma[c] : 1.0f) * rs[e];
        }
    }
    auto one = [&](const float (&xv)[8], const float (&dv)[8], const float (&yv)[8], unsigned mb, size_t o) {
        float out[8], dr[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool on = bits ? ((mb >> e) & 1u) != 0 : (y == nullptr || yv[e] > 0.0f);
            const float d = on ? dv[e] : 0.0f;
            const float xh = (xv[e] - mu[e]) * rs[e];
            out[e] = grs[e] * (d - m1[e] - xh * m2[e]);
            dr[e] = d;
        }
        store8(dx + o, out);
        if (dres) store8(dres + o, dr);
    };
    for (; i + stride < total; i += 2 * stride) {                  // two vectors in flight per thread
        float xa[8], da[8], ya[8], xb[8], db[8], yb[8];
        unsigned ma = 0u, mb = 0u;
        const size_t oa = gbase + i * 8, ob = gbase + (i + stride) * 8;
        load8(x + oa, xa); load8(dy + oa, da); if (bits) ma = bits[i]; else if (y) load8(y + oa, ya);
        load8(x + ob, xb); load8(dy + ob, db); if (bits) mb = bits[i + stride]; else if (y) load8(y + ob, yb);
        one(xa, da, ya, ma, oa);
        one(xb, db, yb, mb, ob);
    }
    if (i < total) {
        float xa[8], da[8], ya[8];
        unsigned ma = 0u;
        const size_t oa = gbase + i * 8;
        load8(x + oa, xa); load8(dy + oa, da); if (bits) ma = bits[i]; else if (y) load8(y + oa, ya);
        one(xa, da, ya, ma, oa);
    }
}

}  // namespace cms

using namespace cms;

static int bn_geo_ok(size_t p, int c) { return p > 0 && c > 0 && c % 8 == 0; }

extern "C" int cms_bn_reduce(const void* x, const void* dy, const void* y, int dtype, const float* mean, const float* rstd,
                             double* sums, size_t n_pixels, int c, int mode, void* stream) {
    CMS_REQUIRE(x && sums, "bn_reduce: NULL pointer");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "bn_reduce: bad dtype");
    CMS_REQUIRE(bn_geo_ok(n_pixels, c), "bn_reduce: bad geometry (channels %% 8 == 0)");
    CMS_REQUIRE(mode == 0 || (mode == 1 && dy && mean && rstd), "bn_reduce: mode 1 needs dy, mean, rstd");
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = dtype == CMS_F32 ? 4 : 2;
    // channel slices of <= 2048 (= 256 groups of 8: one thread per group); every network layer but DenseNet-161's widest
    // is one slice
    for (int c0 = 0; c0 < c; c0 += 2048) {
        const int cs = std::min(2048, c - c0);
        const int CG = cs / 8;
        const int slots = CG >= 256 ? 1 : 256 / CG;
        size_t want = (n_pixels + slots - 1) / slots;
        // every block ends with 2 * C fp64 atomics on the SAME 2 * C addresses: the adds of an address serialise, so that phase
        // grows with the grid (1024 blocks: ~15 us of a 32 us launch; 2048: 52 us in total, profiles/r03t_*). Step time of the
        // batch-statistics DeepLab v2 against the cap (profiles/r03v_*): 128 -> 46.1 ms, 256 -> 45.7, 384 -> 46.8, 512 -> 48.9,
        // 768 -> 52.6: one block per CU with 4-pixel-unrolled loads.
        static int env_cap = -1;                    // CMS_BN_GRID: A/B switch, read once
        if (env_cap < 0) {
            const char* e = getenv("CMS_BN_GRID");
            env_cap = e ? atoi(e) : 256;
        }
        if (env_cap < 1) env_cap = 1;               // (CMS_BN_GRID=0 would ask for an empty grid)
        if (want > (size_t)env_cap) want = (size_t)env_cap;
        const dim3 grid((unsigned)want);
        const size_t lds = (size_t)slots * CG * 16 * sizeof(float);
        const char* xs = (const char*)x + (size_t)c0 * esz;
        const char* dys = dy ? (const char*)dy + (size_t)c0 * esz : nullptr;
        const char* ys = y ? (const char*)y + (size_t)c0 * esz : nullptr;
        const float* ms = mean ? mean + c0 : nullptr;
        const float* rs = rstd ? rstd + c0 : nullptr;
#define CMS_BN_RED(T, M) hipLaunchKernelGGL((bn_reduce_kernel<T, M>), grid, dim3(256), lds, s, (const T*)xs, (const T*)dys, (const T*)ys, ms, rs, sums + c0, n_pixels, cs, c)
        if (dtype == CMS_F32) { if (mode == 0) CMS_BN_RED(float, 0); else CMS_BN_RED(float, 1); }
        else { if (mode == 0) CMS_BN_RED(uint16_t, 0); else CMS_BN_RED(uint16_t, 1); }
#undef CMS_BN_RED
    }
    return launch_status("cms_bn_reduce");
}

// ---- atomics-free reduction (bn_reduce_tiled_kernel) ----------------------------------------------------------------------
// CMS_BN_FENCE=1: the fenced (release / acquire) variant of the last-block hand-off. Read at every launch (a getenv), so that
// one process can run both variants (tests/test_gpu_bn.py, tools/bn_stress.py).
static int bn_fenced() {
    const char* e = getenv("CMS_BN_FENCE");
    return (e && atoi(e) != 0) ? 1 : 0;
}

static int bn_groups_ok(size_t n_pixels, int groups) { return groups >= 1 && n_pixels % (size_t)groups == 0; }

static void bn_tiling(size_t n_pixels, int c, int groups, int* tiles, int* splits) {
    static int target = -1;                         // CMS_BN_BLOCKS: blocks per launch aimed at (A/B switch, read once)
    if (target < 0) {
        const char* e = getenv("CMS_BN_BLOCKS");
        target = e ? std::max(1, atoi(e)) : 512;
    }
    const int t = (c + BN_CT - 1) / BN_CT;
    const size_t pg = n_pixels / (size_t)groups;
    size_t s = (size_t)std::max(1, target / (t * groups));
    s = std::min<size_t>(s, (size_t)std::max(1, 128 / groups));     // <= 128 partial rows per tile for its last block to add
    s = std::min<size_t>(s, (pg + 127) / 128);                      // >= 4 rounds of 32 pixel rows per block
    *tiles = t;
    *splits = (int)std::max<size_t>(s, 1);
}

static size_t bn_counter_bytes(int tiles) { return ((size_t)tiles * sizeof(unsigned) + 255) / 256 * 256; }

extern "C" size_t cms_bn_workspace_bytes(size_t n_pixels, int c, int groups) {
    if (n_pixels == 0 || c <= 0 || !bn_groups_ok(n_pixels, groups)) return 0;
    int tiles, S;
    bn_tiling(n_pixels, c, groups, &tiles, &S);
    return bn_counter_bytes(tiles) + (size_t)tiles * groups * S * 2 * BN_CT * sizeof(float);
}

static int bn_reduce_tiled(const void* x, const void* dy, const void* y, int dtype, const float* mean, const float* rstd,
                           double* sums, size_t n_pixels, int c, int groups, int mode, void* ws, const BnFin* fin,
                           hipStream_t s, const uint8_t* bits = nullptr) {
    int tiles, S;
    bn_tiling(n_pixels, c, groups, &tiles, &S);
    unsigned* counters = (unsigned*)ws;
    float* part = (float*)((char*)ws + bn_counter_bytes(tiles));
    const dim3 grid((unsigned)(tiles * S), (unsigned)groups);
    const size_t pg = n_pixels / (size_t)groups;
    BnFin f = fin ? *fin : BnFin{};
#define CMS_BN_TILED(T, M, F) hipLaunchKernelGGL((bn_reduce_tiled_kernel<T, M, F>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, (const T*)y, mean, rstd, sums, counters, part, pg, c, tiles, S, f, bn_fenced(), bits)
    if (dtype == CMS_F32) {
        if (mode == 1) CMS_BN_TILED(float, 1, false);
        else if (fin) CMS_BN_TILED(float, 0, true);
        else CMS_BN_TILED(float, 0, false);
    } else {
        if (mode == 1) CMS_BN_TILED(uint16_t, 1, false);
        else if (fin) CMS_BN_TILED(uint16_t, 0, true);
        else CMS_BN_TILED(uint16_t, 0, false);
    }
#undef CMS_BN_TILED
    return 0;
}

extern "C" int cms_bn_reduce_ws(const void* x, const void* dy, const void* y, int dtype, const float* mean, const float* rstd,
                                double* sums, size_t n_pixels, int c, int groups, int mode, void* ws, void* stream) {
    CMS_REQUIRE(x && sums && ws, "bn_reduce_ws: NULL pointer");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "bn_reduce_ws: bad dtype");
    CMS_REQUIRE(bn_geo_ok(n_pixels, c), "bn_reduce_ws: bad geometry (channels %% 8 == 0)");
    CMS_REQUIRE(bn_groups_ok(n_pixels, groups), "bn_reduce_ws: %d groups do not divide %zu pixel rows", groups, n_pixels);
    CMS_REQUIRE(mode == 0 || (mode == 1 && dy && mean && rstd), "bn_reduce_ws: mode 1 needs dy, mean, rstd");
    bn_reduce_tiled(x, dy, y, dtype, mean, rstd, sums, n_pixels, c, groups, mode, ws, nullptr, (hipStream_t)stream);
    return launch_status("cms_bn_reduce_ws");
}

extern "C" int cms_bn_reduce_ws_bits(const void* x, const void* dy, const uint8_t* mask_bits, int dtype, const float* mean,
                                     const float* rstd, double* sums, size_t n_pixels, int c, int groups, void* ws, void* stream) {
    CMS_REQUIRE(x && dy && mask_bits && mean && rstd && sums && ws, "bn_reduce_ws_bits: NULL pointer");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "bn_reduce_ws_bits: bad dtype");
    CMS_REQUIRE(bn_geo_ok(n_pixels, c), "bn_reduce_ws_bits: bad geometry (channels %% 8 == 0)");
    CMS_REQUIRE(bn_groups_ok(n_pixels, groups), "bn_reduce_ws_bits: %d groups do not divide %zu pixel rows", groups, n_pixels);
    bn_reduce_tiled(x, dy, nullptr, dtype, mean, rstd, sums, n_pixels, c, groups, 1, ws, nullptr, (hipStream_t)stream, mask_bits);
    return launch_status("cms_bn_reduce_ws_bits");
}

extern "C" int cms_bn_stats(const void* x, int dtype, size_t n_pixels, int c, int groups, const float* gamma, const float* beta,
                            float eps, float momentum, float* mean, float* rstd, float* scale, float* shift,
                            float* running_mean, float* running_var, long long* counter, double* sums, void* ws, void* stream) {
    CMS_REQUIRE(x && ws && mean && rstd && scale && shift, "bn_stats: NULL pointer");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "bn_stats: bad dtype");
    CMS_REQUIRE(bn_geo_ok(n_pixels, c), "bn_stats: bad geometry (channels %% 8 == 0)");
    CMS_REQUIRE(bn_groups_ok(n_pixels, groups), "bn_stats: %d groups do not divide %zu pixel rows", groups, n_pixels);
    BnFin f{gamma, beta, mean, rstd, scale, shift, running_mean, running_var, counter, (double)(n_pixels / (size_t)groups), eps,
            momentum};
    bn_reduce_tiled(x, nullptr, nullptr, dtype, nullptr, nullptr, sums, n_pixels, c, groups, 0, ws, &f, (hipStream_t)stream);
    return launch_status("cms_bn_stats");
}

// Statistics from the tile sums a convolution's epilogue wrote (cms_conv_desc.stats_out, csrc/tile_stats.hpp): no pass over the
// activation at all. tile_sums: float [tiles][2 slots][2][C]; tile t covers pixel rows [t T, (t + 1) T), slot 0 = its rows of the
// sample group its first row is in, slot 1 = its rows of the next group (written only by a straddling tile).
// The launch sits on the dependency chain conv -> statistics -> normalise of every unit, and inside the training step a memory round
// trip costs 4-5 us (loaded latency), so it is built around ONE round trip: a block owns 16 channels, 64 lanes per channel; every
// lane issues ALL its loads (tiles k, k + 64, ...; up to 20 of them = 1280 tiles, plus the slot-1 sums of the <= G - 1 straddling
// tiles) before it looks at any, then sorts them into the sample groups in registers. fp64 sums, lanes joined by shuffles and a
// fixed-order pass over the 16 waves -- reproducible; groups finalised in order (the running statistics move once per group).
// (A two-level version -- 264 small blocks + a last-block ticket -- measured the same 19 us in the step as a naive single-level one:
// its chain is loads -> store -> ticket -> loads, profiles/r05s_*.) More tiles than 1280: rounds of loads per group (rare: 512 x 1024 crops).
constexpr int BNT_CH = 16, BNT_LANES = 64, BNT_J = 20;

__global__ __launch_bounds__(BNT_CH * BNT_LANES) void bn_finalize_tiles_kernel(const float* __restrict__ tile_sums, unsigned T, unsigned Pg,
                                                                               int C, int G, unsigned nt, BnFin fin,
                                                                               double* __restrict__ sums_out) {
    __shared__ double red[BNT_LANES / 4][2 * BNT_CH];       // [wave][stat][channel]
    const int tid = threadIdx.x, cl = tid % BNT_CH, k = tid / BNT_CH, wave = tid >> 6;
    const int c = blockIdx.x * BNT_CH + cl;
    const bool live = c < C;
    const size_t row = 2 * (size_t)C;                       // floats per (tile, slot)
    const bool fast = nt <= (unsigned)(BNT_LANES * BNT_J);
    float x[BNT_J], y[BNT_J];
    float ex0 = 0.0f, ex1 = 0.0f;                           // lane k: slot 1 of the tile that straddles the boundary into group k + 1
    if (fast) {
        // uniform base + 32-bit byte offset per lane: one VGPR per address (global_load ... saddr), not a 64-bit pair -- with the 40
        // loads of a lane in flight at once the pairs alone would overflow the 128 registers a 1024-thread block has
        const char* base = reinterpret_cast<const char*>(tile_sums);
        const unsigned cb = (unsigned)(live ? c : 0) * 4u, rowb = (unsigned)C * 16u;       // bytes per tile = 2 slots x 2 x C floats
        const unsigned off0 = (unsigned)k * rowb + cb, stride = (unsigned)BNT_LANES * rowb, off_max = (nt - 1) * rowb + cb;
#pragma unroll
        for (int j = 0; j < BNT_J; ++j) {
            const unsigned off = min(off0 + (unsigned)j * stride, off_max);      // (past the end: a valid address, never counted)
            x[j] = *reinterpret_cast<const float*>(base + (size_t)off);
            y[j] = *reinterpret_cast<const float*>(base + (size_t)(off + (unsigned)C * 4u));
        }
        if (live && k + 1 < G) {
            const unsigned brow = (unsigned)(k + 1) * Pg;   // first row of group k + 1
            if (brow % T != 0) {
                const float* p = tile_sums + ((size_t)(brow / T) * 2 + 1) * row + c;
                ex0 = p[0];
                ex1 = p[C];
            }
        }
    }
    for (int g = 0; g < G; ++g) {
        double s0 = 0.0, s1 = 0.0;
        if (fast) {
            // slot 0 of tile t belongs to the group of its first row: g Pg <= t T < (g + 1) Pg
            const unsigned lo = ((unsigned)g * Pg + T - 1) / T, hi = ((unsigned)(g + 1) * Pg + T - 1) / T;
#pragma unroll
            for (int j = 0; j < BNT_J; ++j) {
                const unsigned t = (unsigned)k + (unsigned)(BNT_LANES * j);
                const bool mine = t >= lo && t < hi;
                float a = mine ? x[j] : 0.0f, b = mine ? y[j] : 0.0f;
                asm volatile("" : "+v"(a), "+v"(b));       // (keeps the 40 fp64 conversions inside the group loop: hoisted, they spill)
                s0 += (double)a;
                s1 += (double)b;
            }
            if (g >= 1 && k == g - 1) { s0 += (double)ex0; s1 += (double)ex1; }
        } else if (live) {
            const unsigned row_lo = (unsigned)g * Pg, row_hi = row_lo + Pg - 1;
            const unsigned t_lo = row_lo / T, t_hi = row_hi / T;
            for (unsigned t = t_lo + (unsigned)k; t <= t_hi; t += BNT_LANES) {
                const unsigned sl = (t * T) / Pg == (unsigned)g ? 0u : 1u;
                const float* p = tile_sums + ((size_t)t * 2 + sl) * row + c;
                s0 += (double)p[0];
                s1 += (double)p[C];
            }
        }
        // lanes k = 4 w .. 4 w + 3 of a channel sit in wave w at lane distance 16
        s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
        s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
        __syncthreads();                                    // (the previous group's readers of red are done)
        if ((tid & 63) < BNT_CH) { red[wave][cl] = s0; red[wave][BNT_CH + cl] = s1; }
        __syncthreads();
        if (tid < BNT_CH && live) {
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int w = 0; w < BNT_LANES / 4; ++w) { a0 += red[w][cl]; a1 += red[w][BNT_CH + cl]; }
            if (sums_out) {                                 // backward statistics: the sums themselves, [G][2][C] (cms_bn_bwd_sums_tiles)
                sums_out[(size_t)g * 2 * C + c] = a0;
                sums_out[(size_t)g * 2 * C + C + c] = a1;
            } else {
                BnFin f = fin;
                f.mean += (size_t)g * C; f.rstd += (size_t)g * C; f.scale += (size_t)g * C; f.shift += (size_t)g * C;
                bn_finalize_channel(f, c, a0, a1);
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0 && fin.counter) *fin.counter += G;
}

extern "C" int cms_bn_finalize_tiles(const float* tile_sums, int tile_rows, size_t n_pixels, int c, int groups, const float* gamma,
                                     const float* beta, float eps, float momentum, float* mean, float* rstd, float* scale,
                                     float* shift, float* running_mean, float* running_var, long long* counter, void* stream) {
    CMS_REQUIRE(tile_sums && mean && rstd && scale && shift, "bn_finalize_tiles: NULL pointer");
    CMS_REQUIRE(n_pixels > 0 && c > 0 && tile_rows > 0, "bn_finalize_tiles: bad geometry");
    CMS_REQUIRE(n_pixels + (size_t)tile_rows < (1ull << 31), "bn_finalize_tiles: more than 2^31 pixel rows");
    CMS_REQUIRE(((n_pixels + (size_t)tile_rows - 1) / (size_t)tile_rows) * 16 * (size_t)c < (1ull << 32), "bn_finalize_tiles: tile sums beyond 4 GB");
    CMS_REQUIRE(bn_groups_ok(n_pixels, groups) && groups <= BNT_LANES, "bn_finalize_tiles: %d groups do not divide %zu pixel rows (or > 64 groups)",
                groups, n_pixels);
    const size_t pg = n_pixels / (size_t)groups;
    CMS_REQUIRE(pg >= (size_t)tile_rows, "bn_finalize_tiles: a sample group (%zu rows) is shorter than a tile (%d rows)", pg, tile_rows);
    const size_t nt = (n_pixels + (size_t)tile_rows - 1) / (size_t)tile_rows;
    BnFin f{gamma, beta, mean, rstd, scale, shift, running_mean, running_var, counter, (double)pg, eps, momentum};
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3((c + BNT_CH - 1) / BNT_CH), dim3(BNT_CH * BNT_LANES), 0, (hipStream_t)stream,
                       tile_sums, (unsigned)tile_rows, (unsigned)pg, c, groups, (unsigned)nt, f, (double*)nullptr);
    return launch_status("cms_bn_finalize_tiles");
}

// Backward: the tile sums (sum d, sum d xhat) a data-gradient launch wrote (cms_conv_desc.bstats_*) -> sums[groups][2][c], what
// cms_bn_reduce_ws(mode 1) leaves for cms_bn_bwd_apply_groups. Same kernel, same fixed order.
extern "C" int cms_bn_bwd_sums_tiles(const float* tile_sums, int tile_rows, size_t n_pixels, int c, int groups, double* sums,
                                     void* stream) {
    CMS_REQUIRE(tile_sums && sums, "bn_bwd_sums_tiles: NULL pointer");
    CMS_REQUIRE(n_pixels > 0 && c > 0 && tile_rows > 0, "bn_bwd_sums_tiles: bad geometry");
    CMS_REQUIRE(n_pixels + (size_t)tile_rows < (1ull << 31), "bn_bwd_sums_tiles: more than 2^31 pixel rows");
    CMS_REQUIRE(((n_pixels + (size_t)tile_rows - 1) / (size_t)tile_rows) * 16 * (size_t)c < (1ull << 32), "bn_bwd_sums_tiles: tile sums beyond 4 GB");
    CMS_REQUIRE(bn_groups_ok(n_pixels, groups) && groups <= BNT_LANES, "bn_bwd_sums_tiles: %d groups do not divide %zu pixel rows (or > 64 groups)",
                groups, n_pixels);
    const size_t pg = n_pixels / (size_t)groups;
    CMS_REQUIRE(pg >= (size_t)tile_rows, "bn_bwd_sums_tiles: a sample group (%zu rows) is shorter than a tile (%d rows)", pg, tile_rows);
    const size_t nt = (n_pixels + (size_t)tile_rows - 1) / (size_t)tile_rows;
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3((c + BNT_CH - 1) / BNT_CH), dim3(BNT_CH * BNT_LANES), 0, (hipStream_t)stream,
                       tile_sums, (unsigned)tile_rows, (unsigned)pg, c, groups, (unsigned)nt, BnFin{}, sums);
    return launch_status("cms_bn_bwd_sums_tiles");
}

extern "C" int cms_bn_finalize_ex(const double* sums, double count, const float* gamma, const float* beta, float eps,
                                  float momentum, float* mean, float* rstd, float* scale, float* shift, float* running_mean,
                                  float* running_var, int c, double* clear_a, double* clear_b, long long* counter,
                                  void* stream) {
    CMS_REQUIRE(sums && mean && rstd && scale && shift && c > 0 && count > 0, "bn_finalize: NULL pointer / bad geometry");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count, gamma, beta,
                       eps, momentum, mean, rstd, scale, shift, running_mean, running_var, c, clear_a, clear_b, counter);
    return launch_status("cms_bn_finalize");
}

extern "C" int cms_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps,
                               float momentum, float* mean, float* rstd, float* scale, float* shift, float* running_mean,
                               float* running_var, int c, void* stream) {
    return cms_bn_finalize_ex(sums, count, gamma, beta, eps, momentum, mean, rstd, scale, shift, running_mean, running_var, c,
                              nullptr, nullptr, nullptr, stream);
}

extern "C" int cms_bn_apply_groups(const void* x, const void* res, void* y, int dtype, const float* scale, const float* shift,
                                   int relu, size_t n_pixels, int c, int groups, void* stream) {
    return cms_bn_apply_groups_bits(x, res, y, dtype, scale, shift, relu, n_pixels, c, groups, nullptr, stream);
}

extern "C" int cms_bn_apply_groups_bits(const void* x, const void* res, void* y, int dtype, const float* scale, const float* shift,
                                        int relu, size_t n_pixels, int c, int groups, uint8_t* mask_bits_out, void* stream) {
    CMS_REQUIRE(x && y && scale && shift, "bn_apply: NULL pointer");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "bn_apply: bad dtype");
    CMS_REQUIRE(bn_geo_ok(n_pixels, c), "bn_apply: bad geometry (channels %% 8 == 0)");
    CMS_REQUIRE(bn_groups_ok(n_pixels, groups), "bn_apply: %d groups do not divide %zu pixel rows", groups, n_pixels);
    const size_t pg = n_pixels / (size_t)groups;
    const size_t total = pg * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)grid_for(total, 256, std::max(1, 256 * 16 / groups)), (unsigned)groups);
    if (dtype == CMS_F32)
        hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)res, (float*)y, scale,
                           shift, relu, pg, c, mask_bits_out);
    else
        hipLaunchKernelGGL(bn_apply_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)x, (const uint16_t*)res,
                           (uint16_t*)y, scale, shift, relu, pg, c, mask_bits_out);
    return launch_status("cms_bn_apply");
}

// Backward of y = relu(x * scale + shift (+ res)) over FROZEN statistics (eval-mode BatchNorm as an affine, the teacher of the VAT
// trainer: architectures/deeplab2.py LayerEngine.bn_act): dx = scale * dy', dres = dy' with dy' = dy * [y > 0] (relu) or dy.
// One launch where the tensor-op form took four (mask, multiply, two casts). y == NULL: no ReLU.
template <class T>
__global__ __launch_bounds__(256) void frozen_bn_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                            T* __restrict__ dres, const float* __restrict__ scale, size_t P, int C) {
    const int CG = C / 8;
    const size_t total = P * CG;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        float d[8], sc[8], yv[8];
        load8(dy + i * 8, d);
        load8(scale + cg * 8, sc);
        if (y) {
            load8(y + i * 8, yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] = yv[e] > 0.0f ? d[e] : 0.0f;
        }
        if (dres) store8(dres + i * 8, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] *= sc[e];
        store8(dx + i * 8, d);
    }
}

extern "C" int cms_frozen_bn_act_bwd(const void* dy, const void* y, void* dx, void* dres, int dtype, const float* scale,
                                     size_t n_pixels, int c, void* stream) {
    CMS_REQUIRE(dy && dx && scale, "frozen_bn_act_bwd: NULL pointer");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "frozen_bn_act_bwd: bad dtype");
    CMS_REQUIRE(bn_geo_ok(n_pixels, c), "frozen_bn_act_bwd: bad geometry (channels %% 8 == 0)");
    const size_t total = n_pixels * (size_t)(c / 8);
    const dim3 grid((unsigned)grid_for(total, 256, 256 * 16));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == CMS_F32)
        hipLaunchKernelGGL(frozen_bn_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)y, (float*)dx, (float*)dres,
                           scale, n_pixels, c);
    else
        hipLaunchKernelGGL(frozen_bn_bwd_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)dx,
                           (uint16_t*)dres, scale, n_pixels, c);
    return launch_status("cms_frozen_bn_act_bwd");
}

extern "C" int cms_bn_apply(const void* x, const void* res, void* y, int dtype, const float* scale, const float* shift, int relu,
                            size_t n_pixels, int c, void* stream) {
    return cms_bn_apply_groups(x, res, y, dtype, scale, shift, relu, n_pixels, c, 1, stream);
}

extern "C" int cms_bn_bwd_apply_groups(const void* x, const void* dy, const void* y, void* dx, void* dres, int dtype,
                                       const float* mean, const float* rstd, const float* gamma, const double* sums, double count,
                                       size_t n_pixels, int c, int groups, void* stream) {
    return cms_bn_bwd_apply_groups_bits(x, dy, y, nullptr, dx, dres, dtype, mean, rstd, gamma, sums, count, n_pixels, c, groups, stream);
}

extern "C" int cms_bn_bwd_apply_groups_bits(const void* x, const void* dy, const void* y, const uint8_t* mask_bits, void* dx, void* dres,
                                            int dtype, const float* mean, const float* rstd, const float* gamma, const double* sums,
                                            double count, size_t n_pixels, int c, int groups, void* stream) {
    CMS_REQUIRE(x && dy && dx && mean && rstd && sums, "bn_bwd_apply: NULL pointer");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "bn_bwd_apply: bad dtype");
    CMS_REQUIRE(bn_geo_ok(n_pixels, c) && count > 0, "bn_bwd_apply: bad geometry (channels %% 8 == 0)");
    CMS_REQUIRE(bn_groups_ok(n_pixels, groups), "bn_bwd_apply: %d groups do not divide %zu pixel rows", groups, n_pixels);
    const size_t pg = n_pixels / (size_t)groups;
    const size_t total = pg * (c / 8);
    hipStream_t s = (hipStream_t)stream;
    // thread count = multiple of the channel groups (a thread keeps its group): blocks in units of CG / gcd(CG, 256)
    const int CG = c / 8;
    int a = CG, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }
    const unsigned unit = (unsigned)(CG / a);
    static int cap = -1;                            // CMS_BN_APPLY_BLOCKS (A/B switch, read once)
    if (cap < 0) {
        const char* e = getenv("CMS_BN_APPLY_BLOCKS");
        cap = e ? std::max(1, atoi(e)) : 1024;
    }
    unsigned want = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)std::max(1, cap / groups));
    want = std::max(unit, want / unit * unit);
    const dim3 grid(want, (unsigned)groups);
    if (dtype == CMS_F32)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (const float*)dy, (const float*)y,
                           (float*)dx, (float*)dres, mean, rstd, gamma, sums, count, pg, c, mask_bits);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)x, (const uint16_t*)dy,
                           (const uint16_t*)y, (uint16_t*)dx, (uint16_t*)dres, mean, rstd, gamma, sums, count, pg, c, mask_bits);
    return launch_status("cms_bn_bwd_apply");
}

extern "C" int cms_bn_bwd_apply(const void* x, const void* dy, const void* y, void* dx, void* dres, int dtype, const float* mean,
                                const float* rstd, const float* gamma, const double* sums, double count, size_t n_pixels, int c,
                                void* stream) {
    return cms_bn_bwd_apply_groups(x, dy, y, dx, dres, dtype, mean, rstd, gamma, sums, count, n_pixels, c, 1, stream);
}
