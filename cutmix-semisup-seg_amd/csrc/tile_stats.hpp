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

typedef float ts_f32x2 __attribute__((ext_vector_type(2)));     // packed fp32 pairs: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32

// Cost matters: the loop is pure vector ALU work in the epilogue of EVERY tile (a wave instruction is 4 cycles, two waves share a
// SIMD: the first version's ~7 scalar operations per element were +3-9 us per launch, tools/conv_stats_bench.py). Hence packed
// pairs (a dword of the chunk = two adjacent channels = one pair) and a one-slot path for the tiles that do not straddle a group
// boundary (all but G - 1 of them): 2 + 2 operations per pair.
struct TileStats {
    ts_f32x2 s0[4], q0[4], s1[4], q1[4];   // (sum, second sum) of slot 0 / slot 1, per channel pair of the chunk
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 4; ++i) s0[i] = q0[i] = s1[i] = q1[i] = ts_f32x2{0.0f, 0.0f};
    }
    static __device__ __forceinline__ ts_f32x2 pair(uint32_t d) { return ts_f32x2{__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)}; }
    // Forward: the 8 bf16 values of a chunk (4 dwords) -> (sum, sum of squares). TWO: the tile straddles a boundary, `second` = the row
    // belongs to the next group (branch-free: an if / else over the slot becomes a dynamically indexed array, i.e. scratch memory).
    template <bool TWO>
    __device__ __forceinline__ void add(uint32_t x, uint32_t y, uint32_t z, uint32_t w, bool second) {
        const ts_f32x2 v[4] = {pair(x), pair(y), pair(z), pair(w)};
        if constexpr (!TWO) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] += v[i]; q0[i] = v[i] * v[i] + q0[i]; }
        } else {
            // (v, 0) or (0, v) by a SELECT on the packed dwords, not by a product with 0 / 1: an Inf / NaN of one sample group must
            // not reach the neighbouring group's slot (Inf * 0 = NaN)
            const uint32_t k0 = second ? 0u : 0xffffffffu, k1 = ~k0;
            const uint32_t raw[4] = {x, y, z, w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const ts_f32x2 a = pair(raw[i] & k0), b = pair(raw[i] & k1);
                s0[i] += a; q0[i] = a * a + q0[i];
                s1[i] += b; q1[i] = b * b + q1[i];
            }
        }
    }
    // Backward statistics of a batch-statistics unit from the data-gradient launch that writes the gradient dy of its output (round 5):
    // d = bit ? dy : 0 (the unit's ReLU mask; byte 0xff without a ReLU), xhat = (u - mean) * rstd with the statistics of the row's sample
    // group -> (sum d, sum d * xhat), the two sums csrc/bn.hip's backward reduction takes over u, dy and the mask. nmu = -mean.
    template <bool TWO>
    __device__ __forceinline__ void add_bwd(uint32_t dx, uint32_t dy_, uint32_t dz, uint32_t dw, uint32_t ux, uint32_t uy, uint32_t uz,
                                            uint32_t uw, unsigned byte, bool second, const ts_f32x2 (&nmu0)[4], const ts_f32x2 (&rs0)[4],
                                            const ts_f32x2 (&nmu1)[4], const ts_f32x2 (&rs1)[4]) {
        const uint32_t dd[4] = {dx, dy_, dz, dw}, uu[4] = {ux, uy, uz, uw};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // the mask on the PACKED pair: bit 2 i -> low half, bit 2 i + 1 -> high half
            const uint32_t keep = (0u - ((byte >> (2 * i)) & 1u)) & 0x0000ffffu, keep_hi = (0u - ((byte >> (2 * i + 1)) & 1u)) & 0xffff0000u;
            const ts_f32x2 d = pair(dd[i] & (keep | keep_hi)), u = pair(uu[i]);
            if constexpr (!TWO) {
                const ts_f32x2 xh = (u + nmu0[i]) * rs0[i];         // csrc/bn.hip: (x - mean) * rstd
                s0[i] += d; q0[i] = d * xh + q0[i];
            } else {
                const ts_f32x2 xh = second ? (u + nmu1[i]) * rs1[i] : (u + nmu0[i]) * rs0[i];
                const uint32_t k0 = second ? 0u : 0xffffffffu, dm = dd[i] & (keep | keep_hi);
                // selects, not `d * 0` / `0 * xhat`: the groups stay isolated under Inf / NaN
                const ts_f32x2 a = pair(dm & k0), b = pair(dm & ~k0), z2 = {0.0f, 0.0f};
                s0[i] += a; q0[i] = a * (second ? z2 : xh) + q0[i];
                s1[i] += b; q1[i] = b * (second ? xh : z2) + q1[i];
            }
        }
    }
};

// CPR = chunks per tile row (BN / 8), NW = waves of the workgroup, NT = threads. `scratch` = LDS every wave is done with (the tile;
// the caller has synchronised), >= NW * 4 * BN floats. `dst` = out + ((tile * 2) * 2) * Cout + co0: this tile's [slot][stat][Cout] block.
template <int CPR, int NW, int NT>
__device__ __forceinline__ void tile_stats_finish(TileStats& t, bool straddle, float* scratch, float* dst, int Cout) {
    constexpr int BN = CPR * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // lanes l, l + CPR, l + 2 CPR ... of a wave hold the same chunk (a non-straddling tile carries zeros in slot 1: not exchanged)
#pragma unroll
    for (int off = 32; off >= CPR; off >>= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t.s0[i].x += __shfl_xor(t.s0[i].x, off, 64); t.s0[i].y += __shfl_xor(t.s0[i].y, off, 64);
            t.q0[i].x += __shfl_xor(t.q0[i].x, off, 64); t.q0[i].y += __shfl_xor(t.q0[i].y, off, 64);
        }
        if (straddle) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t.s1[i].x += __shfl_xor(t.s1[i].x, off, 64); t.s1[i].y += __shfl_xor(t.s1[i].y, off, 64);
                t.q1[i].x += __shfl_xor(t.q1[i].x, off, 64); t.q1[i].y += __shfl_xor(t.q1[i].y, off, 64);
            }
        }
    }
    if (lane < CPR) {
        // scratch[wave][slot][stat][BN]
        float* p = scratch + (size_t)wave * 4 * BN + lane * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[2 * i] = t.s0[i].x; p[2 * i + 1] = t.s0[i].y;
            p[BN + 2 * i] = t.q0[i].x; p[BN + 2 * i + 1] = t.q0[i].y;
            p[2 * BN + 2 * i] = t.s1[i].x; p[2 * BN + 2 * i + 1] = t.s1[i].y;
            p[3 * BN + 2 * i] = t.q1[i].x; p[3 * BN + 2 * i + 1] = t.q1[i].y;
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
