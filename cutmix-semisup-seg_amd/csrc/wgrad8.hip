// Eight-phase 256 x 256 weight gradient for the wide layers of the DeepLab backbones (gfx950 / CDNA4).
//
// Same operator and operand layouts as conv_wgrad_kernel in conv.hip (autograd of architectures/deeplab2.py:89-109,
// Bottleneck.forward):  dW[tap][co][ci] (fp32) += scale[co] * sum_p dU[p][co] * X[p shifted by tap][ci]  -- a GEMM whose K
// axis is the PIXEL axis, both operands pixel-major (NHWC), so MFMA fragments are read from LDS with the transposing
// ds_read_b64_tr_b16. What is different is the tile and the schedule -- those of csrc/conv8.hip:
//
//   * workgroup = 8 waves = 256 output channels x 256 input channels of ONE tap, one per CU; wave (wn, wm) owns
//     128 (co) x 64 (ci) = 4 x 2 MFMA tiles. The 128 x 128 tile of conv_wgrad_kernel stages 32 KB per 64-pixel step for
//     512 MFMA cycles per wave -- exactly the 512 cycles the CU's vector-memory path (64 B / clk) needs to deliver them, so
//     it runs at the LOADER's rate whatever else is done; this tile stages 64 KB per 2048 MFMA cycles.
//   * a 64-pixel K tile is consumed in FOUR phases (one accumulator quadrant, 8 MFMAs each): A0 x X0, A0 x X1, A1 x X1,
//     A1 x X0 -- A half h = the h-th 64 of each wave row's 128 output channels (128 channels per half tile), X half h = the
//     h-th 32 of each wave column's 64 input channels. A half tile is [64 pixels][128 channels] bf16 = 16 KB, 256-byte
//     rows whose 64-byte slots are XOR-swizzled with pixel & 3 (the four pixel rows of a transposing read then fall into
//     different bank windows) -- conv_wgrad_kernel's LDS image, filled by 2 buffer_load ... lds per wave (4 pixel rows each).
//   * half tiles are loaded SIX phases ahead of their first read into 8 slots (2 K tiles x {A0, X0, X1, A1}), counted
//     s_waitcnt vmcnt(8) + one raw barrier in front of the MFMAs of a phase and one behind, the two wave groups one barrier
//     apart (conv8.hip's pipeline, unchanged); past the end of a pixel slice the same number of loads is issued with
//     out-of-range offsets (hardware zero fill).
//   * the pixel cursor is branch-free: per K tile and lane two GEMM rows (n, oy, ox) advance by 64 pixels with selects; the
//     tap's bounds test turns into an out-of-range offset (zero padding), rows past M likewise.
//
// Split K. A layer has 4 ... 36 such tiles, so the pixel axis is cut into slices -- but FEW: a workgroup here runs 40 ... 170
// K tiles per 256 KB fp32 epilogue (conv_wgrad_kernel: ~22 per 64 KB), and a launch occupies 40 ... 60 CUs: the weight
// gradients of one bottleneck fill the half of the machine the data-gradient stream leaves free. Slices are combined with
// fp32 atomics, or -- with a workspace -- written as slabs and added in slice order by wgrad_reduce_kernel (deterministic).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"

namespace cms {

namespace w8 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BCO = 256, BCI = 256, BK = 64;     // output channels, input channels, pixels per K tile
constexpr int NT = 512;                          // 8 waves
constexpr uint32_t OOB = 0x80000000u;            // byte offset beyond any (< 2 GB) tensor: the load returns zeros
// LDS map (bytes): A stage 0 {half 0, half 1} | A stage 1 | X stage 0 | X stage 1 | per-row factor of the epilogue
constexpr int A_OFF = 0, X_OFF = 65536, STG = 32768, HALF = 16384, SC_OFF = 131072;
constexpr int LDS_BYTES = SC_OFF + BCO * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

struct Args {
    const uint16_t* du;        // bf16 [N][Ho][Wo][Cout]
    const uint16_t* x;         // bf16 [N][H][W][Cin]
    float* dw;                 // fp32 [ntaps][dw_cout][Cin]
    const float* scale;        // [Cout] or NULL
    float* slab;               // split-K partial sums [ksplit][ntaps][Cout][Cin] (plain stores) or NULL (atomics into dw)
    size_t slab_stride;        // floats per slice
    int N, H, W, Cin, Ho, Wo, Cout, ntaps, stride, M;
    int dw_cout;               // rows per tap of the dw tensor
    int nco, nci, ntiles;      // channel tiles; ntiles = nco * nci * ntaps
    int kt_total, kt_per_slice;// K tiles (64 pixels) of the layer / per pixel slice (even)
    int ksplit, xcd_slices;    // pixel slices; 1 = slice s on XCD s % 8 (padded grid; CMS_WGRAD8_XCD=1), 0 = 8 equal runs (default)
    int q64, r64;              // 64 / Wo, 64 % Wo
    uint32_t* trace;           // diagnostic (cms_conv_set_trace): 16 dwords per workgroup, or NULL
    int trace_wgs;
    short tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
};

struct Rsrc { i32x4 w; };
__device__ __forceinline__ Rsrc make_rsrc(const void* p, uint32_t bytes) {
    const uint64_t addr = (uint64_t)p;
    Rsrc r;
    r.w[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)addr);
    r.w[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(addr >> 32) & 0xffffu));      // stride 0: raw buffer
    r.w[2] = __builtin_amdgcn_readfirstlane((int)bytes);                                    // num_records (bytes)
    r.w[3] = 0x00020000;
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// two wave instructions (1 KB each = 4 pixel rows of a half tile, lane-linear in LDS at lds0 and lds0 + 8 KB = 32 rows on)
__device__ __forceinline__ void dma2(const Rsrc& r, uint32_t lds0, uint32_t v0, uint32_t v1) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds\n\t"
                 "s_add_u32 m0, %1, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds0), "v"(v0), "v"(v1), "s"(r.w)
                 : "memory", "scc");
}
#else
__device__ __forceinline__ void dma2(const Rsrc&, uint32_t, uint32_t, uint32_t) {}
#endif

template <int V>
using IC = std::integral_constant<int, V>;

__global__ __launch_bounds__(NT, 2) void wgrad8_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lds_scale = reinterpret_cast<float*>(smem + SC_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;          // 128-channel half of the co tile, 64-channel quarter of the ci tile
    const int grp = wave >> 2;                        // waves w and w + 4 share a SIMD: the two groups run one barrier apart

    // XCD-aware order. Default (rounds 4-6): the logical ids (slice-major) are cut into 8 equal runs, one per XCD -- the launch's
    // workgroups spread EVENLY over the XCDs, neighbouring tiles of a slice share an L2. With 54 = 6 slices x 9 taps that is 6.75
    // tiles per XCD: every slice straddles two XCDs and is fetched twice (2.06 x the algorithmic bytes by the TCC counters).
    // CMS_WGRAD8_XCD=1 (round 6 experiment, measured and NOT the default): slice s entirely on XCD s % 8 (grid padded to
    // 8 * ceil(slices / 8) * tiles, workgroups of slices that do not exist return at once) -- each slice then leaves HBM once, but
    // the launch's 54-56 one-per-CU workgroups land 9 + 9 + ... on six XCDs instead of 7 on each of eight, and next to the
    // data-gradient stream's 16.5 conv8 tiles per XCD some XCDs are over-subscribed (34.5 workgroups for 32 CUs) while others idle:
    // 546 vs 630 img/s at cfg 2, twice in alternation on one box (profiles/r06a_*). Balance across XCDs is worth more than the bytes.
    int tile, ks;
    if (a.xcd_slices) {
        const int b = (int)blockIdx.x, xcd = b & 7, idx = b >> 3;
        tile = idx % a.ntiles;
        ks = xcd + 8 * (idx / a.ntiles);
        if (ks >= a.ksplit) return;
    } else {
        int b = (int)blockIdx.x;
        const int G = (int)gridDim.x, q = G / 8, r = G % 8, xcd = b % 8, idx = b / 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tile = b % a.ntiles; ks = b / a.ntiles;
    }
    const int tco = tile % a.nco, tci = (tile / a.nco) % a.nci, tap = tile / (a.nco * a.nci);
    const int co0 = tco * BCO, ci0 = tci * BCI;
    int dy = 0, dx = 0;
#pragma unroll
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {       // constant-index scan (no dynamic indexing of kernel arguments)
        if (i == tap) { dy = a.tap_dy[i]; dx = a.tap_dx[i]; }
    }
    const int k0 = ks * a.kt_per_slice;
    const int k1 = min(a.kt_total, k0 + a.kt_per_slice);
    if (k0 >= k1) return;                              // empty slice (uniform)

    const bool tracing = a.trace != nullptr && (int)blockIdx.x < a.trace_wgs;
    auto stamp = [&](int slot) {
        if (tracing && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 16 + slot] = (uint32_t)__builtin_amdgcn_s_memtime();
    };
    stamp(0);
    if (tracing && tid == 0) {
        a.trace[(size_t)blockIdx.x * 16 + 12] = (uint32_t)(k1 - k0);
        a.trace[(size_t)blockIdx.x * 16 + 13] = (uint32_t)tile;
        a.trace[(size_t)blockIdx.x * 16 + 14] = (uint32_t)ks;
    }
    if (tid < BCO) lds_scale[tid] = a.scale ? a.scale[co0 + tid] : 1.0f;

    const Rsrc rsrc_a = make_rsrc(a.du, (uint32_t)a.M * a.Cout * 2u);
    const Rsrc rsrc_x = make_rsrc(a.x, (uint32_t)a.N * a.H * a.W * a.Cin * 2u);

    // ---- loader geometry. A wave instruction fills 4 pixel rows x 256 B of a half tile: lane -> row lane >> 4, PHYSICAL
    // 16-byte chunk lane & 15, and fetches the LOGICAL 8 channels the swizzle puts there. Wave w fills rows 4w..4w+3 and
    // 32 + 4w..+3 of every half tile: per K tile a lane deals with TWO GEMM rows (pixels), the same for all four half tiles.
    const int drow = lane >> 4, dcp = lane & 15;
    const int dchl = ((((dcp >> 2) ^ drow) & 3) << 5) + ((dcp & 3) << 3);       // channel of the half tile (0..127, multiple of 8)
    // channels of this lane in a row of dU / X for half 0 (half 1: + 64 / + 32 channels)
    const uint32_t ca0 = (uint32_t)(co0 + (dchl >> 6) * 128 + (dchl & 63));
    const uint32_t cx0 = (uint32_t)(ci0 + (dchl >> 5) * 64 + (dchl & 31));
    const uint32_t a_lds = (uint32_t)(A_OFF + wave * 1024);      // + stage * STG + half * HALF
    const uint32_t x_lds = (uint32_t)(X_OFF + wave * 1024);

    // Pixel cursor of the two rows, advanced by 64 pixels per K tile WITHOUT a branch and without a multiplication: the byte
    // offsets of the rows in dU and X are carried along -- X's through the wraps of (ox, oy) with the constant corrections
    // of "next image row" and "next image" -- and turned into load offsets (or "out of range") where a half tile is issued.
    const int rowl = 4 * wave + drow;                   // row of the K tile (second row: + 32)
    uint32_t abyte[2], xbyte[2];
    int cix[2], ciy[2];                                 // INPUT coordinates the tap reads for the row: ox * stride + dx, oy * stride + dy
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
        const int m = k0 * BK + rowl + 32 * rg;
        const int ox = m % a.Wo, t = m / a.Wo, oy = t % a.Ho, n = t / a.Ho;
        cix[rg] = ox * a.stride + dx;
        ciy[rg] = oy * a.stride + dy;
        abyte[rg] = ((uint32_t)m * (uint32_t)a.Cout + ca0) * 2u;
        xbyte[rg] = ((uint32_t)((n * a.H + ciy[rg]) * a.W + cix[rg]) * (uint32_t)a.Cin + cx0) * 2u;
    }
    const uint32_t a_step = (uint32_t)(BK * a.Cout * 2);
    const uint32_t x_step = (uint32_t)((a.q64 * a.stride * a.W + a.r64 * a.stride) * a.Cin * 2);
    const uint32_t x_wrap_x = (uint32_t)((a.stride * a.W - a.Wo * a.stride) * a.Cin * 2);         // ox -= Wo, oy += 1
    const uint32_t x_wrap_y = (uint32_t)((a.H * a.W - a.Ho * a.stride * a.W) * a.Cin * 2);        // oy -= Ho, n += 1
    const int ix_step = a.r64 * a.stride, iy_step = a.q64 * a.stride;
    const int ix_span = a.Wo * a.stride, iy_span = a.Ho * a.stride;
    const int ix_lim = ix_span + dx, iy_lim = iy_span + dy;                                        // ox == Wo, oy == Ho
    const int lim = min(k1 * BK, a.M);                  // rows of the slice (and of the tensor) end here
    int rem = lim - k0 * BK;                            // rows left from the cursor's K tile on (scalar; <= 0 past the slice)
    auto advance = [&]() {
        rem -= BK;
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            abyte[rg] += a_step;
            uint32_t xb = xbyte[rg] + x_step;
            const int ix1 = cix[rg] + ix_step;
            const bool wx = ix1 >= ix_lim;
            cix[rg] = ix1 - (wx ? ix_span : 0);
            xb += wx ? x_wrap_x : 0u;
            const int iy1 = ciy[rg] + iy_step + (wx ? a.stride : 0);
            const bool wy = iy1 >= iy_lim;              // (the launcher guarantees Ho > 64 / Wo + 1: at most one wrap)
            ciy[rg] = iy1 - (wy ? iy_span : 0);
            xb += wy ? x_wrap_y : 0u;
            xbyte[rg] = xb;
        }
    };
    auto issue_a = [&](auto S_, auto H_) {
        constexpr int S = decltype(S_)::value, H = decltype(H_)::value;
        const uint32_t v0 = rowl < rem ? abyte[0] + (uint32_t)(H * 128) : OOB;
        const uint32_t v1 = rowl + 32 < rem ? abyte[1] + (uint32_t)(H * 128) : OOB;
        dma2(rsrc_a, a_lds + (uint32_t)(S * STG + H * HALF), v0, v1);
    };
    auto issue_x = [&](auto S_, auto H_) {
        constexpr int S = decltype(S_)::value, H = decltype(H_)::value;
        uint32_t v[2];
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            // (bitwise: a short-circuit && would put a branch into the unrolled eight-phase body; unsigned: covers the negative side)
            const bool ok = (rowl + 32 * rg < rem) & ((uint32_t)ciy[rg] < (uint32_t)a.H) & ((uint32_t)cix[rg] < (uint32_t)a.W);
            v[rg] = ok ? xbyte[rg] + (uint32_t)(H * 64) : OOB;
        }
        dma2(rsrc_x, x_lds + (uint32_t)(S * STG + H * HALF), v[0], v[1]);
    };

    // ---- fragment read geometry (conv_wgrad_kernel's): within a 16-pixel k-step the 16-lane group g, lane li supplies the
    // 8-byte chunk (pixel 8 (g >> 1) + (li >> 2), channels 16 (g & 1) + 4 (li & 3) ..+3) of a 32-channel tile and receives
    // channel (lane & 31)'s four pixels; a second read 4 pixel rows on completes the 8-pixel operand
    const int fg = lane >> 4, li = lane & 15;
    const int ch_in_tile = 16 * (fg & 1) + 4 * (li & 3);
    const int pix_in_blk = 8 * (fg >> 1) + (li >> 2);
    auto frag_off = [&](int ch) -> uint32_t {          // byte offset in a half tile of (pixel pix_in_blk, channel ch)
        const int slot = (ch >> 5) ^ (pix_in_blk & 3);
        return (uint32_t)(pix_in_blk * 256 + slot * 64 + (ch & 31) * 2);
    };
    uint32_t fa[2], fxo;                                // A: the wave's two 32-channel tiles of a half; X: its one
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) fa[ii] = (uint32_t)A_OFF + frag_off(wn * 64 + ii * 32 + ch_in_tile);
    fxo = (uint32_t)X_OFF + frag_off(wm * 32 + ch_in_tile);
    // (opaque: hipcc would otherwise fold X_OFF into the per-read constants, which then exceed the 16-bit offset field of a
    // ds_read -- one address register per read, 32 of them spilled)
    asm volatile("" : "+v"(fxo), "+v"(fa[0]), "+v"(fa[1]));
    auto frag = [&](uint32_t off) -> u32x4 {            // (off: k-step * 4096 folded in by the caller as a constant)
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + off + 1024));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        return u32x4{l2.x, l2.y, h2.x, h2.y};
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    u32x4 fw[2][4], fx0[4], fx1[4];
    auto mfma8 = [&](auto IW_, auto JX_, const u32x4 (&fxq)[4]) {
        constexpr int IW = decltype(IW_)::value, JX = decltype(JX_)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                acc[IW + ii][JX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[ii][kk]),
                                                                           __builtin_bit_cast(bf16x8, fxq[kk]), acc[IW + ii][JX], 0, 0, 0);
    };
    // phase P of the K tile in stage S
    auto phase = [&](auto S_, auto P_) {
        constexpr int S = decltype(S_)::value, P = decltype(P_)::value;
        constexpr uint32_t SS = S * STG;
        if constexpr (P == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fx0[kk] = frag(fxo + SS + kk * 4096);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) fw[ii][kk] = frag(fa[ii] + SS + kk * 4096);
            issue_x(IC<S ^ 1>{}, IC<1>{});           // X1 of the next K tile
        } else if constexpr (P == 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fx1[kk] = frag(fxo + SS + HALF + kk * 4096);
            issue_a(IC<S ^ 1>{}, IC<1>{});           // A1 of the next K tile
        } else if constexpr (P == 2) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) fw[ii][kk] = frag(fa[ii] + SS + HALF + kk * 4096);
            advance();
            issue_a(IC<S>{}, IC<0>{});               // A0 of the K tile after the next
        } else {
            issue_x(IC<S>{}, IC<0>{});               // X0 of the K tile after the next
        }
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (P == 0) mfma8(IC<0>{}, IC<0>{}, fx0);
        else if constexpr (P == 1) mfma8(IC<0>{}, IC<1>{}, fx1);
        else if constexpr (P == 2) mfma8(IC<2>{}, IC<1>{}, fx1);
        else mfma8(IC<2>{}, IC<0>{}, fx0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_barrier" ::: "memory");
    };

    // ---- prologue. Issue order of the first six half tiles: A0 A1 X0 X1 | A0' X0' (the K loop continues with X1', A1',
    // ...): vmcnt(4) retires the first K tile -- what its phases read; from the second phase on the steady-state count applies
    issue_a(IC<0>{}, IC<0>{});
    issue_a(IC<0>{}, IC<1>{});
    issue_x(IC<0>{}, IC<0>{});
    issue_x(IC<0>{}, IC<1>{});
    advance();
    issue_a(IC<1>{}, IC<0>{});
    issue_x(IC<1>{}, IC<0>{});
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");
    stamp(1);
    // two K tiles per trip; a slice of odd length computes one K tile of zeros (its loads are out of range)
    for (int t = k0; t < k1; t += 2) {
        phase(IC<0>{}, IC<0>{});
        phase(IC<0>{}, IC<1>{});
        phase(IC<0>{}, IC<2>{});
        phase(IC<0>{}, IC<3>{});
        phase(IC<1>{}, IC<0>{});
        phase(IC<1>{}, IC<1>{});
        phase(IC<1>{}, IC<2>{});
        phase(IC<1>{}, IC<3>{});
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill loads behind the slice have written their slots
    __syncthreads();
    stamp(2);

    // ---- epilogue: every wave through its OWN 8 KB of LDS ([32 rows][64 columns] fp32, no barrier inside -- the LDS
    // operations of one wave complete in order), then one wave instruction per row: 64 consecutive floats of dW. The rows
    // are walked from a slice-dependent start: the slices of one tile finish together and would otherwise all hit row 0
    // of the same cache lines at once (the memory-side atomic units serialise same-address updates).
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));                     // (pins the address arithmetic below behind the K loop)
    const int lane_e = tid_o & 63, fcol = tid_o & 31, fhalf = (tid_o >> 5) & 1;
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 64);
    const int rot = (ks * 5 + tap * 3) & 31;
    float* dwt = a.dw + (size_t)tap * a.dw_cout * a.Cin;
    float* slab_t = a.slab ? a.slab + (size_t)ks * a.slab_stride + (size_t)tap * a.Cout * a.Cin : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) stg[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * 64 + j * 32 + fcol] = acc[i][j][r];
        const int co_l0 = wn * 128 + i * 32;
        if (slab_t) {
            const int c4 = (lane_e & 15) * 4;
#pragma unroll
            for (int q = 0; q < 8; ++q) {                // 4 rows per wave instruction
                const int row = q * 4 + (lane_e >> 4);
                const float sc = lds_scale[co_l0 + row];
                float4 v = *reinterpret_cast<const float4*>(stg + row * 64 + c4);
                v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                *reinterpret_cast<float4*>(slab_t + (size_t)(co0 + co_l0 + row) * a.Cin + ci0 + wm * 64 + c4) = v;
            }
        } else {
            for (int rr = 0; rr < 32; ++rr) {
                const int row = (rr + rot) & 31;
                atomicAdd(dwt + (size_t)(co0 + co_l0 + row) * a.Cin + ci0 + wm * 64 + lane_e, stg[row * 64 + lane_e] * lds_scale[co_l0 + row]);
            }
        }
    }
    stamp(3);
    if (tracing) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(4);
    }
}

}  // namespace w8

// ---- host side ------------------------------------------------------------------------------------------------------------
static int wgrad8_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

bool wgrad8_supported(const cms_wgrad_desc* d) {
    if (!d || !d->du || !d->x || !d->dw) return false;
    if (d->cout % w8::BCO != 0 || d->cin % w8::BCI != 0) return false;
    if (d->cout_real > 0 && d->cout_real != d->cout) return false;
    // BatchNorm-affine side outputs (<W, G>, sum of dU) stay on conv_wgrad_kernel: an eight-phase variant with them was built
    // and measured on DeepLab v3+ at 513 x 513 (170 K tiles per layer, ONE weight-gradient stream): 123 vs 149 img/s
    // (profiles/r04at_*) -- too few K tiles per slice to pay for its epilogue there
    if (d->wdot || d->dbeta || d->w) return false;
    if (d->ntaps <= 0 || d->ntaps > CMS_CONV_MAX_TAPS || d->n <= 0 || d->ho <= 0 || d->wo <= 0 || d->stride < 1) return false;
    if (d->ho <= 64 / d->wo + 1) return false;                           // the branch-free cursor wraps at most one image row + one image
    const size_t ub = (size_t)d->n * d->ho * d->wo * d->cout * 2, xb = (size_t)d->n * d->h * d->w_in * d->cin * 2;
    if (ub >= (1ull << 31) || xb >= (1ull << 31)) return false;
    const int M = d->n * d->ho * d->wo;
    if (M < 16 * w8::BK) return false;                                    // a pipeline this deep needs a K loop to fill
    // A launch that has the machine to itself (wg_target 0) is bound by the fp32 atomics of its slices (~0.7 TB/s in total,
    // tools/atomic_probe.hip): with FEW 256 x 256 tiles the split that fills 256 CUs adds up 64 x |dW| -- the 128 x 128 kernel's
    // 16-21 slices win there (1 x 1 1024->256 at cfg 2: 47 vs 64 us; 3 x 3 256->256: 87 vs 82; 3 x 3 512->512: 210 vs 163,
    // profiles/r04ar_*). Beside other work (wg_target > 0) CU-time is what counts and the eight-phase kernel always wins.
    const int tiles = (d->cout / w8::BCO) * (d->cin / w8::BCI) * d->ntaps;
    return d->wg_target > 0 || d->ksplit > 0 || tiles >= 8;
}

// pixel slices of a launch: K tiles per slice (even) and the number of slices
int wgrad8_plan(const cms_wgrad_desc* d, int* kt_per_slice) {
    static int alone = -1, override_target = 0;
    if (alone < 0) {
        // workgroups (= CUs) of a launch that has the machine to itself: all of them
        int n_cu = 0;
        if (cms_device_info(&n_cu, nullptr, 0) != CMS_OK || n_cu <= 0) n_cu = 256;
        alone = wgrad8_env("CMS_WGRAD8_ALONE", n_cu);
        // A/B override of what the CALLER asked for (cms_wgrad_desc.wg_target). The training step runs two weight-gradient
        // streams beside the data-gradient chain and asks for 56 per launch: 2 x 56 + the 132 tiles of a data-gradient
        // convolution = the machine (profiles/r04ag-ai_*: 40 -> 545 img/s, 48 -> 575-581, 56 -> 588-601, 64 -> 573, 96 ->
        // 564; "equal K tiles per workgroup" rules -- every launch about as long -- lost on one, two and three streams:
        // 478-564, profiles/r04ah_*, r04am_*)
        override_target = wgrad8_env("CMS_WGRAD8_TARGET", 0);
    }
    const int target = d->wg_target > 0 ? (override_target > 0 ? override_target : d->wg_target) : alone;
    // K tiles per slice below which a slice is not worth its 256 KB epilogue: 24 beside other work (CU-time is the currency),
    // 8 alone (wall time is)
    const int min_kt = d->wg_target > 0 ? 24 : 8;
    const int M = d->n * d->ho * d->wo;
    const int kt = (M + w8::BK - 1) / w8::BK;
    const int tiles = (d->cout / w8::BCO) * (d->cin / w8::BCI) * d->ntaps;
    int ks;
    if (d->ksplit > 0) ks = std::min(d->ksplit, std::max(1, kt / 2));              // the caller's choice
    else ks = std::max(1, std::min(std::max(1, (target + tiles / 2) / tiles), std::max(1, kt / min_kt)));
    int per = (kt + ks - 1) / ks;
    per += per & 1;                                                       // even: the K loop takes two K tiles per trip
    ks = (kt + per - 1) / per;
    if (kt_per_slice) *kt_per_slice = per;
    return ks;
}

int wgrad8_launch(const cms_wgrad_desc* d, hipStream_t s, void* trace, int trace_wgs) {
    CMS_REQUIRE(wgrad8_supported(d), "wgrad8: needs Cout %% 256 == 0, Cin %% 256 == 0, no side outputs, tensors below 2 GB");
    w8::Args a;
    a.du = (const uint16_t*)d->du; a.x = (const uint16_t*)d->x; a.dw = d->dw; a.scale = d->scale;
    a.N = d->n; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout;
    a.ntaps = d->ntaps; a.stride = d->stride; a.M = d->n * d->ho * d->wo;
    a.dw_cout = d->dw_cout > 0 ? d->dw_cout : d->cout;
    CMS_REQUIRE(a.dw_cout >= d->cout, "wgrad8: dw_cout (%d) < cout (%d)", a.dw_cout, d->cout);
    a.nco = d->cout / w8::BCO; a.nci = d->cin / w8::BCI; a.ntiles = a.nco * a.nci * d->ntaps;
    a.kt_total = (a.M + w8::BK - 1) / w8::BK;
    const int ksplit = wgrad8_plan(d, &a.kt_per_slice);
    a.q64 = w8::BK / d->wo; a.r64 = w8::BK % d->wo;
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        a.tap_dy[i] = (short)(i < d->ntaps ? d->tap_dy[i] : 0);
        a.tap_dx[i] = (short)(i < d->ntaps ? d->tap_dx[i] : 0);
    }
    const size_t slice_elems = (size_t)d->ntaps * d->cout * d->cin;
    const bool use_slab = d->workspace != nullptr && ksplit > 1 &&
                          (unsigned long long)d->workspace_bytes >= (unsigned long long)ksplit * slice_elems * sizeof(float);
    a.slab = use_slab ? (float*)d->workspace : nullptr;
    a.slab_stride = slice_elems;
    a.trace = (uint32_t*)trace; a.trace_wgs = trace_wgs;
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(w8::wgrad8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    static int xcd_mode = -1;
    if (xcd_mode < 0) xcd_mode = wgrad8_env("CMS_WGRAD8_XCD", 0);
    a.ksplit = ksplit;
    a.xcd_slices = (xcd_mode != 0 && ksplit > 1) ? 1 : 0;
    const int grid = a.xcd_slices ? 8 * ((ksplit + 7) / 8) * a.ntiles : a.ntiles * ksplit;
    hipLaunchKernelGGL(w8::wgrad8_kernel, dim3(grid), dim3(w8::NT), w8::LDS_BYTES, s, a);
    if (use_slab) wgrad_reduce_launch(a.slab, d->dw, ksplit, slice_elems, d->ntaps, d->cout, d->cin, d->cout, a.dw_cout, s);
    return launch_status("cms_conv_wgrad (8-phase)");
}

}  // namespace cms
