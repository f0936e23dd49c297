// Per-channel (sum, sum of squares) of the bf16 output tile a convolution workgroup has built in LDS, taken inside its store loop
// (round 5: BatchNorm statistics out of the convolution epilogue, cms_conv_desc.stats_out). Shared by conv.hip and conv8.hip.
//
// The store loop gives thread t the 16-byte chunk ch = t % CPR (8 channels) of the rows r0 = t / CPR, r0 + NT / CPR, ...: it adds
// what it stores. A tile may straddle ONE boundary between sample groups (groups are >= one tile long): rows of the group the
// tile's first row belongs to go to slot 0, rows of the next group to slot 1. After the loop: lanes of a wave that hold the same
// chunk are summed with shuffles, the waves' partial sums meet in LDS (the tile itself is done with), and the workgroup writes
// out[tile][slot][stat][Cout] for its channels -- plain stores, fixed order: reproducible.
#pragma once
#include "common.hpp"

namespace cms {

struct TileStats {
    float s0[8], q0[8], s1[8], q1[8];      // (sum, sum of squares) of slot 0 / slot 1, per channel of the chunk
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] = q0[e] = s1[e] = q1[e] = 0.0f;
    }
    // the 8 bf16 values of a chunk (4 dwords). Branch-free: an if / else over the slot is turned into a dynamically indexed
    // array by the compiler, i.e. into scratch memory.
    __device__ __forceinline__ void add(uint32_t x, uint32_t y, uint32_t z, uint32_t w, bool second) {
        float v[8];
        v[0] = __uint_as_float(x << 16); v[1] = __uint_as_float(x & 0xffff0000u);
        v[2] = __uint_as_float(y << 16); v[3] = __uint_as_float(y & 0xffff0000u);
        v[4] = __uint_as_float(z << 16); v[5] = __uint_as_float(z & 0xffff0000u);
        v[6] = __uint_as_float(w << 16); v[7] = __uint_as_float(w & 0xffff0000u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = second ? 0.0f : v[e], b = second ? v[e] : 0.0f;
            s0[e] += a; q0[e] = fmaf(a, a, q0[e]);
            s1[e] += b; q1[e] = fmaf(b, b, q1[e]);
        }
    }
    // Backward statistics of a batch-statistics unit from the data-gradient launch that writes the gradient dy of its output (round 5):
    // d = bit ? dy : 0 (the unit's ReLU mask; byte 0xff without a ReLU), xhat = (u - mean) * rstd with the statistics of the row's sample
    // group -> (sum d, sum d * xhat), the two sums csrc/bn.hip's backward reduction takes over u, dy and the mask.
    __device__ __forceinline__ void add_bwd(uint32_t dx, uint32_t dy_, uint32_t dz, uint32_t dw, uint32_t ux, uint32_t uy, uint32_t uz,
                                            uint32_t uw, unsigned byte, bool second, const float (&mu0)[8], const float (&rs0)[8],
                                            const float (&mu1)[8], const float (&rs1)[8]) {
        float d[8], u[8];
        d[0] = __uint_as_float(dx << 16); d[1] = __uint_as_float(dx & 0xffff0000u);
        d[2] = __uint_as_float(dy_ << 16); d[3] = __uint_as_float(dy_ & 0xffff0000u);
        d[4] = __uint_as_float(dz << 16); d[5] = __uint_as_float(dz & 0xffff0000u);
        d[6] = __uint_as_float(dw << 16); d[7] = __uint_as_float(dw & 0xffff0000u);
        u[0] = __uint_as_float(ux << 16); u[1] = __uint_as_float(ux & 0xffff0000u);
        u[2] = __uint_as_float(uy << 16); u[3] = __uint_as_float(uy & 0xffff0000u);
        u[4] = __uint_as_float(uz << 16); u[5] = __uint_as_float(uz & 0xffff0000u);
        u[6] = __uint_as_float(uw << 16); u[7] = __uint_as_float(uw & 0xffff0000u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float dd = ((byte >> e) & 1u) ? d[e] : 0.0f;
            const float xh = (u[e] - (second ? mu1[e] : mu0[e])) * (second ? rs1[e] : rs0[e]);     // csrc/bn.hip: (x - mean) * rstd
            const float a = second ? 0.0f : dd, b = second ? dd : 0.0f;
            s0[e] += a; q0[e] = fmaf(a, xh, q0[e]);
            s1[e] += b; q1[e] = fmaf(b, xh, q1[e]);
        }
    }
};

// CPR = chunks per tile row (BN / 8), NW = waves of the workgroup, NT = threads. `scratch` = LDS every wave is done with (the tile;
// the caller has synchronised), >= NW * 4 * BN floats. `dst` = out + ((tile * 2) * 2) * Cout + co0: this tile's [slot][stat][Cout] block.
template <int CPR, int NW, int NT>
__device__ __forceinline__ void tile_stats_finish(TileStats& t, bool straddle, float* scratch, float* dst, int Cout) {
    constexpr int BN = CPR * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // lanes l, l + CPR, l + 2 CPR ... of a wave hold the same chunk
#pragma unroll
    for (int off = 32; off >= CPR; off >>= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            t.s0[e] += __shfl_xor(t.s0[e], off, 64);
            t.q0[e] += __shfl_xor(t.q0[e], off, 64);
            t.s1[e] += __shfl_xor(t.s1[e], off, 64);
            t.q1[e] += __shfl_xor(t.q1[e], off, 64);
        }
    }
    if (lane < CPR) {
        // scratch[wave][slot][stat][BN]
        float* p = scratch + (size_t)wave * 4 * BN + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] = t.s0[e];
            p[BN + e] = t.q0[e];
            p[2 * BN + e] = t.s1[e];
            p[3 * BN + e] = t.q1[e];
        }
    }
    __syncthreads();
    const int n_out = (straddle ? 2 : 1) * 2 * BN;          // slot 1 only for a tile that straddles a group boundary
    for (int i = tid; i < n_out; i += NT) {
        const int c = i % BN, ks = i / BN;                   // ks = slot * 2 + stat
        float acc = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += scratch[(w * 4 + ks) * BN + c];
        dst[(size_t)ks * Cout + c] = acc;
    }
}

}  // namespace cms
