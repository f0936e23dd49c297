// Eight-phase 256 x 256 implicit-GEMM convolution for the wide, K-deep layers of the DeepLab backbones (gfx950 / CDNA4).
//
// Same operator, operand layouts and fused epilogues as conv_igemm_kernel in conv.hip (architectures/deeplab2.py:89-109,
// Bottleneck.forward, and its autograd twin): D[co][pixel] = sum_{tap, ci} W[tap][co][ci] * X[pixel shifted by tap][ci]
// on v_mfma_f32_32x32x16_bf16, bf16 NHWC activations, fp32 accumulation, BN affine / residual / ReLU / ReLU-mask in the
// epilogue. What is different is the schedule (cdna_hip_programming.md, "The 256^2 8-phase template"):
//
//   * workgroup = 8 waves = 256 output channels x 256 pixels, one per CU; wave (wn, wm) owns 128 channels x 64 pixels
//     = 4 x 2 MFMA tiles (128 accumulator registers). Per 64-deep K tile a CU stages 64 KB for 2 x the MFMA work per
//     staged byte of the 128 x 128 tile (whose LDS fill, fragment reads and MFMA time are within 2 x of each other).
//   * a K tile is consumed in FOUR phases, one accumulator quadrant (2 x 1 MFMA tiles x K = 64: 8 MFMAs) each:
//     W0 x X0, W0 x X1, W1 x X1, W1 x X0 -- W-half h = 64 of the wave's 128 channels, X-half h = 32 of its 64 pixels.
//     A phase is  { fragment reads of the half it needs first (12 / 4 / 8 / 0 ds_read_b128), issue ONE 16 KB half tile
//     of a later K tile global -> LDS (2 buffer_load ... lds per wave), counted s_waitcnt vmcnt, s_barrier }
//     { 8 MFMAs under s_setprio 1, s_barrier }.
//   * the two wave groups (waves 0-3 / 4-7 = the two waves of every SIMD) run ONE barrier apart: while one group is
//     in its MFMA half-phase the other issues its LDS reads and loads, so every SIMD always has matrix work queued.
//   * half tiles are loaded SIX phases ahead of their first read into 8 LDS slots (2 K tiles x {W0, X0, X1, W1});
//     s_waitcnt vmcnt(8) before a phase's first barrier leaves the four youngest half tiles in flight and retires
//     exactly what the NEXT phase reads (read one phase after the wait that retires it); a slot is refilled two or three
//     phases after its last read. The queue never drains inside the K loop; past the end of the K range the same
//     number of loads is issued with out-of-range offsets (hardware zero fill, no memory traffic), so the counted
//     waits stay exact.
//   * loads are buffer-addressed direct-to-LDS (inline asm: hipcc would put vmcnt(0) in front of the next ds_read if
//     it knew about them); the implicit-GEMM part is a (tap, K-chunk) cursor: 4 per-lane byte offsets for the pixel
//     rows, rewritten once per TAP (bounds test -> out-of-range offset = zero padding), the K position in a scalar.
//
// Tile-count quantisation. The layers this kernel is for have 132 ... 1052 tiles of 256 x 256 -- 0.5 ... 4.1 rounds of
// the 256 CUs, i.e. up to half the machine idles in the last round of a plain launch. The launch is therefore
// PERSISTENT (one workgroup per CU) and splits the work as "data-parallel rounds + one stream-K round": the first
// (rounds - 1) * G tiles go one per workgroup and round; the K loops of the remaining G ... 2G-1 tiles (or all tiles if
// fewer than G) are cut into G equal runs of K tiles. A run that covers part of a tile stores its fp32 accumulators to a
// slab (write-through stores), takes a ticket on the tile's counter, and the LAST arriver adds the pieces in run order
// (its own included at its position: bit-reproducible) and runs the epilogue -- nobody ever waits for another workgroup.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "tile_stats.hpp"

namespace cms {

namespace c8 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 256, BM = 256, BK = 64;       // channels, pixels, K elements per tile step
constexpr int NT = 512;                          // 8 waves
constexpr uint32_t OOB = 0x80000000u;            // byte offset beyond any (< 2 GB) tensor: the load returns zeros
// LDS map (bytes): W stage 0 | W stage 1 | X stage 0 | X stage 1 | tap table | row table | scale, bias | flags | tap x row
// input offsets.
// A stage of an operand is [256 rows][128 B], 16-byte chunks XOR-swizzled with (row >> 1) & 7 (conv.hip's image).
constexpr int W_OFF = 0, X_OFF = 65536, STG = 32768, TAB_OFF = 131072;
constexpr int TAP_BYTES = 80, ROW_BYTES = BM * 16, SB_BYTES = 2 * BN * 4, FLAG_BYTES = 16;
constexpr int XOFF_OFF = TAB_OFF + TAP_BYTES + ROW_BYTES + SB_BYTES + FLAG_BYTES;     // [ntaps][256 rows] byte offsets of the input pixels
constexpr int LDS_BYTES = XOFF_OFF + CMS_CONV_MAX_TAPS * BM * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int SLAB_FLOATS = BM * BN;             // one partial accumulator tile (fp32)

struct Args {
    const uint16_t* x;         // bf16 [N][H][W][Cin]
    const uint16_t* w;         // bf16 [ntaps][Cout][Cin]
    uint16_t* y;               // bf16 [N][out_H][out_W][Cout]
    const float* scale;        // [Cout] or NULL (forward)
    const float* bias;         // [Cout] or NULL (forward)
    const uint16_t* res;       // bf16, indexed like y, or NULL
    const uint16_t* mask_src;  // bf16, indexed like y, or NULL (dgrad)
    uint8_t* mask_bits_out;    // forward + ReLU: [y > 0] of the stored output as bits, [out pixels][Cout / 8] bytes, or NULL
    const uint8_t* mask_bits;  // dgrad: the ReLU mask as such bits instead of mask_src, or NULL
    float* slab;               // stream-K partial tiles [2 * grid][SLAB_FLOATS] or NULL
    unsigned* counters;        // stream-K arrival counters [tiles of the stream-K round], all zero between launches
    int N, H, W, Cin, Ho, Wo, Cout, ntaps, stride, out_H, out_W, out_stride, relu, mode, M, plain;
    float* stats_out;          // [pixel tiles][2 slots][2][Cout] per-tile (sum, sum of squares) of the stored output, or NULL
    int stats_rpg;             // pixel rows per sample group (>= 256; M for one group)
    int mask_gates_res;        // dgrad with res + mask_bits: out = acc + (bit ? res : 0) instead of bit ? acc + res : 0
    const uint16_t* bstats_u;  // dgrad + stats_out: u / ReLU mask bits / mean / rstd of the unit whose output gradient this launch writes
    const uint8_t* bstats_bits;
    const float* bstats_mean;
    const float* bstats_rstd;
    int nt_store;              // output rows as non-temporal stores (CMS_CONV8_NT): streaming data must not evict the halo rows
                               // and weights neighbouring tiles of the XCD re-read from its L2 (profiles/r05b: 1.30 x over-fetch)
    int ntn;                   // channel tiles (Cout / 256); tile t = (pixel tile t / ntn, channel tile t % ntn)
    int KT, kc_per_tap;        // K tiles per output tile, K tiles per tap
    int ku;                    // K tiles per stream-K unit (2 when KT is even: runs then have even lengths, see the K loop)
    int sk_tiles;              // tiles [0, sk_tiles) are cut into runs of K tiles; 0 = every workgroup does whole tiles
    int dp_rounds;             // whole tiles per workgroup behind them: tile sk_tiles + r * grid + g
    int ntiles;
    uint32_t* trace;           // diagnostic (cms_conv_set_trace): 64 dwords per workgroup, 16 per run (the first 4 runs), or NULL
    int trace_wgs;
    short tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
};

struct RowInfo {            // one per pixel row of the tile
    uint32_t in_off;        // element offset of input pixel (oy*stride, ox*stride), channel 0
    uint32_t yx;            // (oy*stride) << 16 | (ox*stride); 0x70007000 for rows past M (every bounds test fails)
    uint32_t opix;          // output pixel index, 0xffffffff for rows past M
    uint32_t m;             // GEMM row
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

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, p);
}

#if defined(__HIP_DEVICE_COMPILE__)
// two wave instructions (1 KB each, lane-linear in LDS at lds0 and lds0 + 16 KB): 16 bytes per lane from rsrc[v + soff]
__device__ __forceinline__ void dma2(const Rsrc& r, uint32_t lds0, uint32_t v0, uint32_t v1, uint32_t soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %4, %5 offen lds\n\t"      // (5 states behind a v_readfirstlane of %5)
                 "s_add_u32 m0, %1, 0x4000\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds0), "v"(v0), "v"(v1), "s"(r.w), "s"(soff)
                 : "memory", "scc");
}
__device__ __forceinline__ void dma1(const Rsrc& r, uint32_t lds0, uint32_t v0, uint32_t soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds0), "v"(v0), "s"(r.w), "s"(soff)
                 : "memory");
}
// one wave instruction, 4 bytes per lane (64 floats) to LDS bytes [lds0, lds0 + 256)
__device__ __forceinline__ void dma_dword(const Rsrc& r, uint32_t lds0, uint32_t v0) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds0), "v"(v0), "s"(r.w)
                 : "memory");
}
// write-through 16-byte store / L1-bypassing 16-byte load (aux bit 4 = sc1): the slab hand-off between workgroups needs no
// fence (cdna_hip_programming.md, Guideline 16 R1). Compiler-visible buffer operations: hipcc counts and waits for them.
typedef __amdgpu_buffer_rsrc_t brsrc_t;
__device__ __forceinline__ brsrc_t slab_rsrc(float* p) {
    const uint64_t a64 = (uint64_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a64);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a64 >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0, (int)(SLAB_FLOATS * 4), 0x00020000);
}
__device__ __forceinline__ void store_sc1(brsrc_t r, uint32_t voff, const u32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, 16);
}
__device__ __forceinline__ u32x4 load_sc1(brsrc_t r, uint32_t voff) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 16); }
#else
__device__ __forceinline__ void dma2(const Rsrc&, uint32_t, uint32_t, uint32_t, uint32_t) {}
__device__ __forceinline__ void dma1(const Rsrc&, uint32_t, uint32_t, uint32_t) {}
__device__ __forceinline__ void dma_dword(const Rsrc&, uint32_t, uint32_t) {}
typedef int brsrc_t;
__device__ __forceinline__ brsrc_t slab_rsrc(float*) { return 0; }
__device__ __forceinline__ void store_sc1(brsrc_t, uint32_t, const u32x4&) {}
__device__ __forceinline__ u32x4 load_sc1(brsrc_t, uint32_t) { return u32x4{0u, 0u, 0u, 0u}; }
#endif

template <int V>
using IC = std::integral_constant<int, V>;

// first / last run (workgroup) whose K-tile range [g * total / G, (g + 1) * total / G) contains unit u
__device__ __forceinline__ int run_of_unit(uint64_t u, uint64_t total, uint64_t G) {
    return (int)(((u + 1) * G + total - 1) / total) - 1;
}

// STATS: the store loop also takes the per-tile channel sums (Args.stats_out) -- its own instantiation, so that the statistics'
// 32 accumulators cost the default kernel nothing (it sits at 254 VGPRs without scratch)
template <bool SK, bool STATS = false>
__global__ __launch_bounds__(NT, 2) void conv8_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short* lds_tap = reinterpret_cast<short*>(smem + TAB_OFF);
    RowInfo* lds_row = reinterpret_cast<RowInfo*>(smem + TAB_OFF + TAP_BYTES);
    float* lds_sb = reinterpret_cast<float*>(smem + TAB_OFF + TAP_BYTES + ROW_BYTES);
    int* lds_flag = reinterpret_cast<int*>(smem + TAB_OFF + TAP_BYTES + ROW_BYTES + SB_BYTES);
    uint32_t* lds_xoff = reinterpret_cast<uint32_t*>(smem + XOFF_OFF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;          // 128-channel half, 64-pixel quarter of the tile
    const int grp = wave >> 2;                        // waves w and w + 4 share a SIMD: the two groups run one barrier apart

    // XCD-aware order: consecutive logical ids (neighbouring tiles / runs) stay on one XCD's L2
    const int G = (int)gridDim.x;
    int g = (int)blockIdx.x;
    {
        const int q = G / 8, r = G % 8, xcd = g % 8, idx = g / 8;
        g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }

    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
            lds_tap[i] = a.tap_dy[i];
            lds_tap[CMS_CONV_MAX_TAPS + i] = a.tap_dx[i];
        }
    }

    const Rsrc rsrc_x = make_rsrc(a.x, (uint32_t)a.N * a.H * a.W * a.Cin * 2u);
    const Rsrc rsrc_w = make_rsrc(a.w, (uint32_t)a.ntaps * a.Cout * a.Cin * 2u);
    const uint32_t out_bytes = (uint32_t)a.N * a.out_H * a.out_W * a.Cout * 2u;

    // ---- per-lane constants of the loader: a wave instruction fills 8 rows x 128 B; lane -> row lane >> 3, PHYSICAL chunk
    // lane & 7, and fetches the LOGICAL chunk the swizzle puts there. Piece (half h, i) of this wave:
    //   W rows  i * 128 + h * 64 + wave * 8 ..+7          X rows  (2 i + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 ..+7
    const int lrow8 = lane >> 3;
    const int clog = (lane & 7) ^ ((((wave & 1) << 2) + (lrow8 >> 1)) & 7);     // (row >> 1) & 7 is the same for all pieces
    const uint32_t w_lds = (uint32_t)(W_OFF + wave * 1024);                               // + stage, + h * 8192, + i * 16384
    const uint32_t x_lds = (uint32_t)(X_OFF + (wave >> 2) * 8192 + (wave & 3) * 1024);    // + stage, + h * 4096, + i * 16384

    // ---- fragment read addresses: row (lane & 31) of a 32-row MFMA tile, 16-byte chunk (2 kk + (lane >> 5)) ^ swizzle(row)
    const int frow = lane & 31, fhalf = lane >> 5;
    const uint32_t lane_frag = (uint32_t)frow * 128u | (uint32_t)(((fhalf ^ (frow >> 1)) & 7) << 4);
    uint32_t fwk[4], fxk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        fwk[kk] = (uint32_t)(W_OFF + wn * 128 * 128) + (lane_frag ^ (uint32_t)(kk * 32));
        fxk[kk] = (uint32_t)(X_OFF + wm * 64 * 128) + (lane_frag ^ (uint32_t)(kk * 32));
    }
    auto lds16 = [&](uint32_t off) -> u32x4 { return *reinterpret_cast<const u32x4*>(smem + off); };

    f32x16 acc[4][2];
    // cycle stamps of thread 0 (tools/conv8_trace.py): per run  0 start, 1 tables built, 2 first half tiles landed, 3 K loop
    // done, 4 partial tile published, 5 pieces summed, 6 epilogue operands staged, 7 output tile built, 8 stores issued;
    // 12 = K tiles of the run, 13 = tile, 14 = (partial << 1) | last arriver
    const bool tracing = a.trace != nullptr && (int)blockIdx.x < a.trace_wgs;
    int run_idx = 0;
    auto stamp = [&](int slot) {
        if (tracing && threadIdx.x == 0 && run_idx < 4)
            a.trace[(size_t)blockIdx.x * 64 + run_idx * 16 + slot] = (uint32_t)__builtin_amdgcn_s_memtime();
    };
    auto note = [&](int slot, uint32_t v) {
        if (tracing && threadIdx.x == 0 && run_idx < 4) a.trace[(size_t)blockIdx.x * 64 + run_idx * 16 + slot] = v;
    };

    // ================================================================================================================
    // one run of K tiles [k0, k1) of one output tile
    auto process = [&](const int tile, const int k0, const int k1, const int sk_total_units) {
        const int tile_n = tile % a.ntn, tile_m = tile / a.ntn;
        const int co0 = tile_n * BN, m0 = tile_m * BM;

        stamp(0);
        note(12, (uint32_t)(k1 - k0));
        note(13, (uint32_t)tile);
        note(14, (k0 != 0 || k1 != a.KT) ? 2u : 0u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the previous run's output stores: the counted waits assume an empty queue)
        __syncthreads();                     // the previous run's epilogue is done with the tables and the staging area

        // ---- loader state: 4 + 4 per-lane byte offsets [h][i], scalar offsets of the cursor's K tile. The cursor is
        // BRANCH-FREE (selects on scalar conditions): the unrolled eight-phase body stays one basic block
        uint32_t wv[2][2], xv[2][2];
        uint32_t xrow[2][2];                 // LDS byte address of this lane's row in the tap-0 offset table
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wv[h][i] = (uint32_t)(((i * 128 + h * 64 + wave * 8 + lrow8) * a.Cin + clog * 8) * 2);
                xrow[h][i] = (uint32_t)(XOFF_OFF + ((2 * i + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + lrow8) * 4);
            }
        int it = k0, itap = k0 / a.kc_per_tap, ikc = k0 - itap * a.kc_per_tap;
        uint32_t soff_w = 0, soff_x = 0;
        bool live = true;
        auto set_soff = [&]() {              // scalar offsets of K tile `it` (tap itap, chunk ikc)
            const int tap_c = itap < a.ntaps ? itap : a.ntaps - 1;
            soff_x = (uint32_t)__builtin_amdgcn_readfirstlane(ikc * BK * 2);
            soff_w = (uint32_t)__builtin_amdgcn_readfirstlane((((tap_c * a.Cout + co0) * a.Cin) + ikc * BK) * 2);
        };
        auto load_xv = [&]() {               // pixel-row offsets of tap `itap`; past the run: everything out of range
            const int tap_c = itap < a.ntaps ? itap : a.ntaps - 1;
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane(tap_c * BM * 4);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t o = *reinterpret_cast<const uint32_t*>(smem + xrow[h][i] + tb);
                    xv[h][i] = live ? o + (uint32_t)(clog * 16) : OOB;      // (an out-of-range entry stays out of range)
                }
        };
        auto step_cursor = [&]() {
            ++it;
            live = it < k1;
            const bool wrap = ikc + 1 == a.kc_per_tap;
            ikc = wrap ? 0 : ikc + 1;
            itap += wrap ? 1 : 0;
            set_soff();
        };
        auto advance = [&]() {
            step_cursor();
            load_xv();
        };
        auto issue_w = [&](auto S_, auto H_) {
            constexpr int S = decltype(S_)::value, H = decltype(H_)::value;
            dma2(rsrc_w, w_lds + (uint32_t)(S * STG + H * 8192), live ? wv[H][0] : OOB, live ? wv[H][1] : OOB, soff_w);
        };
        auto issue_x = [&](auto S_, auto H_) {
            constexpr int S = decltype(S_)::value, H = decltype(H_)::value;
            dma2(rsrc_x, x_lds + (uint32_t)(S * STG + H * 4096), xv[H][0], xv[H][1], soff_x);
        };

        // ---- prologue, part 1: what does not need the tables -- the BN affine of the tile's channels (asynchronously, 4
        // bytes per lane: it lands long before the epilogue reads it) and the weight half tiles of the first K tiles
        {
            const bool is_scale = wave < 4;
            const float* src = is_scale ? a.scale : a.bias;
            if (src != nullptr && a.mode == 0) {                                 // wave-uniform
                const Rsrc rs = make_rsrc(src, (uint32_t)a.Cout * 4u);
                dma_dword(rs, (uint32_t)(TAB_OFF + TAP_BYTES + ROW_BYTES + wave * 256), (uint32_t)((co0 + (wave & 3) * 64 + lane) * 4));
            } else {
                lds_sb[wave * 64 + lane] = is_scale ? 1.0f : 0.0f;               // (1, 0 in the dgrad epilogue)
            }
        }
        set_soff();
        const int tap_first = itap;
        const uint32_t soff_x_first = soff_x;
        // Pointwise launches (1 x 1, stride 1: the GEMM row IS the input pixel) know the offsets of their pixel rows without
        // the tables: ALL six half tiles of the prologue go out here and fly while the tables are built (round 4: the tables
        // + the wait for the first loads behind them were 7 k of the 70 k cycles of a 16-K-tile run)
        const bool early = a.plain != 0;
        issue_w(IC<0>{}, IC<0>{});
        issue_w(IC<0>{}, IC<1>{});
        if (early) {
            uint32_t xd[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + (2 * i + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + lrow8;
                    xd[h][i] = m < a.M ? (uint32_t)(m * a.Cin * 2 + clog * 16) : OOB;
                    xv[h][i] = xd[h][i];
                }
            issue_x(IC<0>{}, IC<0>{});
            issue_x(IC<0>{}, IC<1>{});
            step_cursor();
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) xv[h][i] = live ? xd[h][i] : OOB;
            issue_w(IC<1>{}, IC<0>{});
            issue_x(IC<1>{}, IC<0>{});
        } else {
            step_cursor();
            issue_w(IC<1>{}, IC<0>{});
        }

        // ---- tables. Per row: geometry (for the epilogue); per (tap, row): byte offset of the input pixel that tap reads for
        // that GEMM row, or out of range (zero padding, rows past M) -- the K loop then needs no bounds test, no row table and
        // no branch when its cursor changes tap. Both halves of the workgroup compute the row's geometry in registers.
        {
            const int r = tid & (BM - 1);
            const int m = m0 + r;
            RowInfo ri;
            ri.m = (uint32_t)m;
            if (m >= a.M) {
                ri.in_off = 0;
                ri.yx = 0x70007000u;
                ri.opix = 0xffffffffu;
            } else if (a.plain) {
                ri.in_off = (uint32_t)(m * a.Cin);
                ri.yx = 0u;
                ri.opix = (uint32_t)m;
            } else {
                const int ox = m % a.Wo;
                const int t = m / a.Wo;
                const int oy = t % a.Ho;
                const int n = t / a.Ho;
                const int iy = oy * a.stride, ix = ox * a.stride;
                ri.in_off = (uint32_t)(((n * a.H + iy) * a.W + ix) * a.Cin);
                ri.yx = ((uint32_t)iy << 16) | (uint32_t)ix;
                ri.opix = (uint32_t)((n * a.out_H + oy * a.out_stride) * a.out_W + ox * a.out_stride);
            }
            if (tid < BM) lds_row[r] = ri;
            for (int tap = tid >> 8; tap < a.ntaps; tap += NT / BM) {
                const int dy = lds_tap[tap], dx = lds_tap[CMS_CONV_MAX_TAPS + tap];
                const uint32_t iy = (ri.yx >> 16) + (uint32_t)dy, ix = (ri.yx & 0xffffu) + (uint32_t)dx;
                const bool ok = iy < (uint32_t)a.H && ix < (uint32_t)a.W;       // unsigned compare covers the negative side
                lds_xoff[tap * BM + r] = ok ? (ri.in_off + (uint32_t)((dy * a.W + dx) * a.Cin)) * 2u : OOB;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        __syncthreads();
        stamp(1);

        u32x4 fw[2][4], fx0[4], fx1[4];
        auto mfma8 = [&](auto IW_, auto JX_, const u32x4 (&fxq)[4]) {
            constexpr int IW = decltype(IW_)::value, JX = decltype(JX_)::value;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
                    acc[IW + ii][JX] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[ii][kk]),
                                                                               __builtin_bit_cast(bf16x8, fxq[kk]),
                                                                               acc[IW + ii][JX], 0, 0, 0);
        };
        // phase P of the K tile in stage S
        auto phase = [&](auto S_, auto P_) {
            constexpr int S = decltype(S_)::value, P = decltype(P_)::value;
            constexpr uint32_t WS = S * STG, XS = S * STG;
            if constexpr (P == 0) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fx0[kk] = lds16(fxk[kk] + XS);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) fw[ii][kk] = lds16(fwk[kk] + WS + ii * 4096);
                issue_x(IC<S ^ 1>{}, IC<1>{});           // X1 of the next K tile
            } else if constexpr (P == 1) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fx1[kk] = lds16(fxk[kk] + XS + 4096);
                issue_w(IC<S ^ 1>{}, IC<1>{});           // W1 of the next K tile
            } else if constexpr (P == 2) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) fw[ii][kk] = lds16(fwk[kk] + WS + 8192 + ii * 4096);
                advance();
                issue_w(IC<S>{}, IC<0>{});               // W0 of the K tile after the next
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

        // ---- prologue, part 2: the pixel half tiles of the first K tiles. Issue order of the first six half tiles: W0 W1 | W0' |
        // X0 X1 | X0' (the K loop continues with X1', W1', ...): vmcnt(2) retires everything but X0' -- what phases 0 and 1 of
        // the first K tile read; from the second phase on the steady-state count applies. (Pointwise launches issued
        // W0 W1 X0 X1 | W0' X0' above: vmcnt(4).)
        if (!early) {
            const int tap_now = itap;
            const uint32_t soff_x_now = soff_x;
            const bool live_now = live;
            itap = tap_first; soff_x = soff_x_first; live = true;
            load_xv();
            issue_x(IC<0>{}, IC<0>{});
            issue_x(IC<0>{}, IC<1>{});
            itap = tap_now; soff_x = soff_x_now; live = live_now;
            load_xv();
            issue_x(IC<1>{}, IC<0>{});
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        }
        if (grp == 1) asm volatile("s_barrier" ::: "memory");
        stamp(2);
        // two K tiles per trip; a run of odd length computes one K tile of zeros (its loads are out of range) -- the
        // launcher cuts runs at even K tiles wherever the tile's K-tile count is even
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-fill loads behind the run have written their slots
        __syncthreads();
        stamp(3);

        // Everything below is invariant across the runs of a persistent workgroup as far as the compiler can see, and it
        // hoists the epilogue's ~100 per-lane addresses out of the work loop, i.e. keeps them alive (spilled) across the K
        // loop. An opaque copy of the thread id pins their computation here.
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int tid = tid_o, lane = tid_o & 63, frow = tid_o & 31, fhalf = (tid_o >> 5) & 1;
        // ---- stream-K: a partial run publishes its accumulators; the last arriver of the tile sums the pieces in run order
        if (SK && (k0 != 0 || k1 != a.KT)) {
            const uint64_t total = (uint64_t)sk_total_units, GG = (uint64_t)G;
            const uint64_t tb = (uint64_t)tile * (uint64_t)(a.KT / a.ku);          // first unit of the tile
            const int g_first = run_of_unit(tb, total, GG), g_last = run_of_unit(tb + (uint64_t)(a.KT / a.ku) - 1, total, GG);
            // slab slot of a piece: 2 * run + (0: the run BEGINS in this tile, 1: it began in an earlier tile)
            auto slot_of = [&](int gg) -> uint32_t {
                const uint64_t u0 = (uint64_t)gg * total / GG;
                return (uint32_t)(2 * gg + (u0 >= tb ? 0 : 1));
            };
            {
                const brsrc_t rs = slab_rsrc(a.slab + (size_t)slot_of(g) * SLAB_FLOATS);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const u32x4 v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]),
                                             __float_as_uint(acc[i][j][4 * q + 2]), __float_as_uint(acc[i][j][4 * q + 3])};
                            store_sc1(rs, (uint32_t)((((i * 2 + j) * 4 + q) * NT + tid) * 16), v);
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave: its write-through stores have left
            __syncthreads();
            if (tid == 0) {
                const unsigned ticket = __hip_atomic_fetch_add(a.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = ticket == (unsigned)(g_last - g_first);
                if (last) __hip_atomic_store(a.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lds_flag[0] = last;
            }
            __syncthreads();
            stamp(4);
            if (lds_flag[0] == 0) { ++run_idx; return; }
            note(14, 3u);
            // ordered sum, a quarter of the accumulators (one channel tile i: 32 registers) at a time
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 sum[8];
                bool first = true;
                for (int gg = g_first; gg <= g_last; ++gg) {
                    f32x4 v[8];
                    if (gg == g) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            v[e] = f32x4{acc[i][e >> 2][4 * (e & 3)], acc[i][e >> 2][4 * (e & 3) + 1], acc[i][e >> 2][4 * (e & 3) + 2],
                                         acc[i][e >> 2][4 * (e & 3) + 3]};
                    } else {
                        const brsrc_t rs = slab_rsrc(a.slab + (size_t)slot_of(gg) * SLAB_FLOATS);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            v[e] = __builtin_bit_cast(f32x4, load_sc1(rs, (uint32_t)(((i * 8 + e) * NT + tid) * 16)));
                    }
                    if (first) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum[e] = v[e];
                        first = false;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum[e] += v[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][e >> 2][4 * (e & 3) + c] = sum[e][c];
            }
        }

        stamp(5);
        // ---- epilogue (conv.hip's): every lane owns runs of 4 consecutive channels of one pixel = 8 bytes of NHWC. Residual /
        // mask tiles are fetched global -> LDS as whole 512-byte rows (direct-to-LDS, rows past M out of range = zeros) and
        // picked up from there in accumulator layout; the bf16 output tile is built IN PLACE and written with 16 bytes per
        // lane, 512 contiguous bytes per row. LDS tile: [256 pixels][256 channels] bf16, 16-byte chunks XOR-swizzled with
        // the row -- exactly the 128 KB of the staging area.
        constexpr int CPR = BN / 8;                 // 32 chunks per tile row
        constexpr int EROW = BN * 2;                // 512 B
        auto stage_tile = [&](const uint16_t* src) {
            const Rsrc rs = make_rsrc(src, out_bytes);
#pragma unroll
            for (int i = 0; i < BM / 16; ++i) {     // a wave instruction fills 2 rows
                const int row = (8 * i + wave) * 2 + (lane >> 5);
                const int cl = (lane & 31) ^ (row & (CPR - 1));
                const uint32_t op = lds_row[row].opix;
                const uint32_t vo = op != 0xffffffffu ? (uint32_t)((op * (uint32_t)a.Cout + (uint32_t)(co0 + cl * 8)) * 2u) : OOB;
                dma1(rs, (uint32_t)((8 * i + wave) * 1024), vo, 0u);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        auto slot = [&](int prow_l, int co_l) -> uint32_t {
            return (uint32_t)(prow_l * EROW + ((((co_l >> 3) ^ prow_l) & (CPR - 1)) << 4) + ((co_l & 4) << 1));
        };
        uint64_t mbits[2] = {0, 0};                 // ReLU mask of this lane's 128 elements when BOTH operands are present
        const bool both = a.res && a.mask_src;
        // (round 5) the ReLU mask of a data gradient as BITS the producing forward launch wrote (conv.hip's layout: one dword
        // per pixel row and 32-channel MFMA tile, bit = channel): 1/16 of the bytes of the bf16 activation, no staged tile
        const bool bits_in = a.mask_bits != nullptr;
        if (bits_in) {
            const int bpr = a.Cout >> 3;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t op = lds_row[(wm * 2 + j) * 32 + frow].opix;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t wbits = 0u;
                    if (op != 0xffffffffu)
                        wbits = *reinterpret_cast<const uint32_t*>(a.mask_bits + (size_t)op * bpr + ((co0 + (wn * 4 + i) * 32) >> 3));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int bit = ((i * 2 + j) * 4 + q) * 4;
                        const uint64_t m4 = (wbits >> (8 * q + 4 * fhalf)) & 0xfu;
                        mbits[bit >> 6] |= m4 << (bit & 63);
                    }
                }
            }
        }
        if (both) {
            stage_tile(a.mask_src);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint2 mk = *reinterpret_cast<const uint2*>(
                            smem + slot((wm * 2 + j) * 32 + frow, (wn * 4 + i) * 32 + 8 * q + 4 * fhalf));
                        const int bit = ((i * 2 + j) * 4 + q) * 4;
                        const uint64_t m4 = ((int16_t)(mk.x & 0xffffu) > 0 ? 1u : 0u) | ((int16_t)(mk.x >> 16) > 0 ? 2u : 0u) |
                                            ((int16_t)(mk.y & 0xffffu) > 0 ? 4u : 0u) | ((int16_t)(mk.y >> 16) > 0 ? 8u : 0u);
                        mbits[bit >> 6] |= m4 << (bit & 63);
                    }
            __syncthreads();                        // everybody has its bits: the tile may be overwritten
        }
        if (a.res || a.mask_src) stage_tile(a.res ? a.res : a.mask_src);
        stamp(6);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        // (round 4) two values per vector instruction wherever the ISA has one: v_pk_fma_f32 for the BN affine, v_pk_add_f32 for
        // the residual, and the ReLU on the PACKED bf16 pair (v_pk_max_i16 with 0: a bf16 is negative iff its bit pattern is a
        // negative int16; rounding first and clamping then gives the bits of clamping first) -- the residual + ReLU nest was 730
        // vector instructions per wave, the whole epilogue VALU-bound. The data-gradient nests skip the (1, 0) affine.
        auto nest = [&](auto AFF_, auto RES_, auto MSK_, auto RELU_, auto BITS_) {
            constexpr int AFF = decltype(AFF_)::value;      // 1: y = acc * scale + bias (forward); 0: the accumulator itself
            constexpr int RES = decltype(RES_)::value;      // 1: residual / gradient add from the staged tile
            constexpr int MSK = decltype(MSK_)::value;      // 1: ReLU mask bits (both operands / mask_bits), 2: mask from the staged tile
            constexpr int RELU = decltype(RELU_)::value;
            constexpr int BITS = decltype(BITS_)::value;    // 1: also write [y > 0] of the STORED output as bits (mask_bits_out)
            uint32_t obits[BITS ? 4 : 1][BITS ? 2 : 1];
#pragma unroll
            for (int i = 0; i < (BITS ? 4 : 1); ++i)
#pragma unroll
                for (int j = 0; j < (BITS ? 2 : 1); ++j) obits[i][j] = 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co_l = (wn * 4 + i) * 32 + 8 * q + 4 * fhalf;
                    f32x2 sc01 = {1.0f, 1.0f}, sc23 = {1.0f, 1.0f}, b01 = {0.0f, 0.0f}, b23 = {0.0f, 0.0f};
                    if constexpr (AFF) {
                        const float4 sc = *reinterpret_cast<const float4*>(lds_sb + co_l);
                        const float4 b = *reinterpret_cast<const float4*>(lds_sb + BN + co_l);
                        sc01 = f32x2{sc.x, sc.y}; sc23 = f32x2{sc.z, sc.w};
                        b01 = f32x2{b.x, b.y}; b23 = f32x2{b.z, b.w};
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int prow_l = (wm * 2 + j) * 32 + frow;
                        unsigned char* cell = smem + slot(prow_l, co_l);
                        f32x2 v01 = {acc[i][j][4 * q], acc[i][j][4 * q + 1]}, v23 = {acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        uint2 keep = {0u, 0u};
                        if constexpr (AFF) {
                            v01 = v01 * sc01 + b01;
                            v23 = v23 * sc23 + b23;
                        }
                        if constexpr (RES == 1 || MSK == 2) {
                            const uint2 rr = *reinterpret_cast<const uint2*>(cell);
                            if constexpr (RES == 1 && MSK == 3) {
                                // the mask bits gate the RESIDUAL (gradient of a shortcut whose ReLU mask they are), not the sum
                                const int bit = ((i * 2 + j) * 4 + q) * 4;
                                const uint32_t m4 = (uint32_t)(mbits[bit >> 6] >> (bit & 63));
                                v01 += f32x2{(m4 & 1u) ? __uint_as_float(rr.x << 16) : 0.0f, (m4 & 2u) ? __uint_as_float(rr.x & 0xffff0000u) : 0.0f};
                                v23 += f32x2{(m4 & 4u) ? __uint_as_float(rr.y << 16) : 0.0f, (m4 & 8u) ? __uint_as_float(rr.y & 0xffff0000u) : 0.0f};
                            } else if constexpr (RES == 1) {
                                v01 += f32x2{__uint_as_float(rr.x << 16), __uint_as_float(rr.x & 0xffff0000u)};
                                v23 += f32x2{__uint_as_float(rr.y << 16), __uint_as_float(rr.y & 0xffff0000u)};
                            } else {
                                keep = rr;
                            }
                        }
                        if constexpr (MSK == 1) {
                            const int bit = ((i * 2 + j) * 4 + q) * 4;
                            const uint32_t m4 = (uint32_t)(mbits[bit >> 6] >> (bit & 63));
                            v01.x = (m4 & 1u) ? v01.x : 0.0f; v01.y = (m4 & 2u) ? v01.y : 0.0f;
                            v23.x = (m4 & 4u) ? v23.x : 0.0f; v23.y = (m4 & 8u) ? v23.y : 0.0f;
                        }
                        uint2 o;
                        o.x = pack_bf16x2(v01.x, v01.y);
                        o.y = pack_bf16x2(v23.x, v23.y);
                        if constexpr (MSK == 2) {
                            // ReLU mask from the staged bf16 pair, on the PACKED result: a bf16 is > 0 iff its bits are a
                            // positive int16 -> max(x, 0) is non-zero -> min(.., 1) * 0xffff = all ones in that half
                            uint32_t mx, my;
                            asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, %2\n\tv_pk_mul_lo_u16 %0, %0, %3"
                                : "=&v"(mx) : "v"(keep.x), "s"(0x00010001u), "s"(0xffffffffu));
                            asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, %2\n\tv_pk_mul_lo_u16 %0, %0, %3"
                                : "=&v"(my) : "v"(keep.y), "s"(0x00010001u), "s"(0xffffffffu));
                            o.x &= mx;
                            o.y &= my;
                        }
                        if constexpr (RELU) {
                            // (inline: as a vector max the compiler converts the two halves separately and re-packs them)
                            asm("v_pk_max_i16 %0, %1, 0" : "=v"(o.x) : "v"(o.x));
                            asm("v_pk_max_i16 %0, %1, 0" : "=v"(o.y) : "v"(o.y));
                        }
                        *reinterpret_cast<uint2*>(cell) = o;
                        if constexpr (BITS) {
                            // of the STORED value: exactly what a data gradient reading the activation back would test
                            const uint32_t m4 = ((int16_t)(o.x & 0xffffu) > 0 ? 1u : 0u) | ((int16_t)(o.x >> 16) > 0 ? 2u : 0u) |
                                                ((int16_t)(o.y & 0xffffu) > 0 ? 4u : 0u) | ((int16_t)(o.y >> 16) > 0 ? 8u : 0u);
                            obits[i][j] |= m4 << (8 * q + 4 * fhalf);
                        }
                    }
                }
            }
            if constexpr (BITS) {
                const int bpr = a.Cout >> 3;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t op = lds_row[(wm * 2 + j) * 32 + frow].opix;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // lanes l and l + 32 hold the two nibbles of every byte of the pixel row's 32 channels
                        const uint32_t wbits = obits[i][j] | (uint32_t)__shfl_xor((int)obits[i][j], 32, 64);
                        if (fhalf == 0 && op != 0xffffffffu)
                            *reinterpret_cast<uint32_t*>(a.mask_bits_out + (size_t)op * bpr + ((co0 + (wn * 4 + i) * 32) >> 3)) = wbits;
                    }
                }
            }
        };
        if (a.mode == 0) {
            if (a.relu && a.mask_bits_out) {
                if (a.res) nest(IC<1>{}, IC<1>{}, IC<0>{}, IC<1>{}, IC<1>{}); else nest(IC<1>{}, IC<0>{}, IC<0>{}, IC<1>{}, IC<1>{});
            }
            else if (a.res) { if (a.relu) nest(IC<1>{}, IC<1>{}, IC<0>{}, IC<1>{}, IC<0>{}); else nest(IC<1>{}, IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{}); }
            else { if (a.relu) nest(IC<1>{}, IC<0>{}, IC<0>{}, IC<1>{}, IC<0>{}); else nest(IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}); }
        } else {
            if (bits_in && a.res && a.mask_gates_res) nest(IC<0>{}, IC<1>{}, IC<3>{}, IC<0>{}, IC<0>{});
            else if (both || (bits_in && a.res)) nest(IC<0>{}, IC<1>{}, IC<1>{}, IC<0>{}, IC<0>{});
            else if (bits_in) nest(IC<0>{}, IC<0>{}, IC<1>{}, IC<0>{}, IC<0>{});
            else if (a.mask_src) nest(IC<0>{}, IC<0>{}, IC<2>{}, IC<0>{}, IC<0>{});
            else if (a.res) nest(IC<0>{}, IC<1>{}, IC<0>{}, IC<0>{}, IC<0>{});
            else nest(IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{}, IC<0>{});
        }
        __syncthreads();
        stamp(7);
        {
            // thread -> LOGICAL chunk ch (8 channels) of rows r0, r0 + 16, ...: the same channels in every pass (tile_stats.hpp)
            const int ch = tid % CPR, r0 = tid / CPR;       // 16 rows per pass
            auto rows = [&](auto KIND_, auto TWO_, TileStats& ts, int boundary, const ts_f32x2 (&mu0)[4], const ts_f32x2 (&rs0)[4],
                            const ts_f32x2 (&mu1)[4], const ts_f32x2 (&rs1)[4]) {
                constexpr int KIND = decltype(KIND_)::value;              // 0: store, 1: + forward statistics, 2: + backward statistics
                constexpr bool TWO = decltype(TWO_)::value != 0;          // the tile straddles a sample-group boundary (both slots)
#pragma unroll
                for (int r = r0; r < BM; r += NT / CPR) {
                    const uint32_t op = lds_row[r].opix;
                    if (op != 0xffffffffu) {
                        const u32x4 val = *reinterpret_cast<const u32x4*>(smem + r * EROW + ((ch ^ (r & (CPR - 1))) << 4));
                        u32x4* dst = reinterpret_cast<u32x4*>(a.y + (size_t)op * a.Cout + co0 + ch * 8);
#if defined(__HIP_DEVICE_COMPILE__)
                        if (a.nt_store) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(val) : "memory");
                        else
#endif
                            *dst = val;
                        if constexpr (KIND == 1) ts.template add<TWO>(val.x, val.y, val.z, val.w, (m0 + r) >= boundary);
                        if constexpr (KIND == 2) {
                            const u32x4 uv = *reinterpret_cast<const u32x4*>(a.bstats_u + (size_t)op * a.Cout + co0 + ch * 8);
                            const unsigned byte = a.bstats_bits ? a.bstats_bits[(size_t)op * (a.Cout >> 3) + (co0 >> 3) + ch] : 0xffu;
                            ts.template add_bwd<TWO>(val.x, val.y, val.z, val.w, uv.x, uv.y, uv.z, uv.w, byte, (m0 + r) >= boundary, mu0, rs0, mu1, rs1);
                        }
                    }
                }
            };
            ts_f32x2 mu0[4] = {}, rs0[4] = {}, mu1[4] = {}, rs1[4] = {};      // (mu = MINUS the mean)
            if constexpr (!STATS) {
                TileStats none;
                rows(IC<0>{}, IC<0>{}, none, 0, mu0, rs0, mu1, rs1);
            } else {
                TileStats ts;
                // BatchNorm statistics out of the epilogue (round 5): per-channel sums over what this tile stores, [tile][slot][stat][Cout]:
                // forward launches (sum, sum of squares); data gradients with bstats_u (sum d, sum d xhat) -- see conv.hip
                const int g0 = m0 / a.stats_rpg;
                const int boundary = (g0 + 1) * a.stats_rpg;        // first row of the next sample group
                const int m_end = m0 + BM < a.M ? m0 + BM : a.M;
                ts.zero();
                if (a.bstats_u == nullptr) {
                    if (boundary < m_end) rows(IC<1>{}, IC<1>{}, ts, boundary, mu0, rs0, mu1, rs1);
                    else rows(IC<1>{}, IC<0>{}, ts, boundary, mu0, rs0, mu1, rs1);
                } else {
                    const int g1 = boundary < a.M ? g0 + 1 : g0;
                    const float* p0 = a.bstats_mean + (size_t)g0 * a.Cout + co0 + ch * 8;
                    const float* p1 = a.bstats_mean + (size_t)g1 * a.Cout + co0 + ch * 8;
                    const float* q0 = a.bstats_rstd + (size_t)g0 * a.Cout + co0 + ch * 8;
                    const float* q1 = a.bstats_rstd + (size_t)g1 * a.Cout + co0 + ch * 8;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        mu0[i] = ts_f32x2{-p0[2 * i], -p0[2 * i + 1]}; mu1[i] = ts_f32x2{-p1[2 * i], -p1[2 * i + 1]};
                        rs0[i] = ts_f32x2{q0[2 * i], q0[2 * i + 1]}; rs1[i] = ts_f32x2{q1[2 * i], q1[2 * i + 1]};
                    }
                    if (boundary < m_end) rows(IC<2>{}, IC<1>{}, ts, boundary, mu0, rs0, mu1, rs1);
                    else rows(IC<2>{}, IC<0>{}, ts, boundary, mu0, rs0, mu1, rs1);
                }
                __syncthreads();                    // every wave has read the tile: its LDS is the scratch now
                tile_stats_finish<CPR, NT / 64, NT>(ts, boundary < m_end, reinterpret_cast<float*>(smem),
                                                    a.stats_out + (size_t)tile_m * 4 * a.Cout + co0, a.Cout);
                __syncthreads();                    // (a persistent workgroup stages its next tile into the same LDS)
            }
        }
        stamp(8);
        ++run_idx;
    };

    // ================================================================================================================
    // work list of this workgroup: its run of the stream-K round first, then its whole tiles (ONE call site: the body is large)
    const int KU = a.KT / a.ku;                      // units per tile
    const uint64_t total = (uint64_t)a.sk_tiles * KU;
    uint64_t u = (SK && a.sk_tiles > 0) ? (uint64_t)g * total / (uint64_t)G : 0;
    const uint64_t u1 = (SK && a.sk_tiles > 0) ? (uint64_t)(g + 1) * total / (uint64_t)G : 0;
    int r = 0;
    for (;;) {
        int tile, k0, k1;
        if (SK && u < u1) {
            tile = (int)(u / (uint64_t)KU);
            k0 = (int)(u - (uint64_t)tile * KU);
            const uint64_t rest = u1 - u;
            k1 = rest < (uint64_t)(KU - k0) ? k0 + (int)rest : KU;
            u += (uint64_t)(k1 - k0);
        } else if (r < a.dp_rounds) {
            tile = a.sk_tiles + r * G + g;
            ++r;
            if (tile >= a.ntiles) break;
            k0 = 0;
            k1 = KU;
        } else {
            break;
        }
        process(tile, k0 * a.ku, k1 * a.ku, (int)total);
    }
}

}  // namespace c8

// Launch. mode 0: one whole tile per workgroup (grid = tiles); 1: persistent, data-parallel rounds + one stream-K round
// on `grid_cap` workgroups (0 = the CU count). The workspace holds 64 KB of arrival counters (zero between launches: the
// caller clears them once, the kernel leaves them zero) followed by 2 * grid slabs of 256 KB.
constexpr size_t CONV8_COUNTER_BYTES = 65536;      // at the START of the workspace: the same place for every grid size
size_t conv8_workspace_bytes(int n_cu) { return CONV8_COUNTER_BYTES + (size_t)2 * n_cu * c8::SLAB_FLOATS * 4; }

bool conv8_supported(const cms_conv_desc* d) {
    if (!d->y || d->y32 || d->ksplit > 1) return false;
    if (d->cout % c8::BN != 0 || d->cin % c8::BK != 0 || d->cout_real != d->cout) return false;
    const size_t xb = (size_t)d->n * d->h * d->w_in * d->cin * 2, wb = (size_t)d->ntaps * d->cout * d->cin * 2;
    const size_t ob = (size_t)d->n * d->out_h * d->out_w * d->cout * 2;
    return xb < (1ull << 31) && wb < (1ull << 31) && ob < (1ull << 31);
}

int conv8_launch(const cms_conv_desc* d, hipStream_t s, int mode, int grid_cap, void* trace, int trace_wgs) {
    CMS_REQUIRE(conv8_supported(d), "conv8: needs the bf16 NHWC output, Cout %% 256 == 0, Cin %% 64 == 0 and tensors below 2 GB");
    c8::Args a;
    a.x = (const uint16_t*)d->x; a.w = (const uint16_t*)d->w; a.y = (uint16_t*)d->y;
    a.scale = d->scale; a.bias = d->bias; a.res = (const uint16_t*)d->res; a.mask_src = (const uint16_t*)d->mask_src;
    a.mask_bits_out = d->mask_bits_out; a.mask_bits = d->mask_bits;
    CMS_REQUIRE(d->mask_bits_out == nullptr || (d->mode == 0 && d->relu != 0), "conv8: mask_bits_out is written by forward + ReLU launches");
    CMS_REQUIRE(d->mask_bits == nullptr || (d->mode == 1 && d->mask_src == nullptr), "conv8: mask_bits replaces mask_src of a data gradient");
    CMS_REQUIRE((d->mask_bits_out == nullptr && d->mask_bits == nullptr) || mode == 0,
                "conv8: ReLU mask bits with whole tiles per workgroup only (not the stream-K launch)");
    a.mask_gates_res = d->mask_gates_res;
    CMS_REQUIRE(d->mask_gates_res == 0 || (d->mode == 1 && d->res && d->mask_bits), "conv8: mask_gates_res belongs to data-gradient launches with a residual and mask bits");
    a.stats_out = (float*)d->stats_out;
    a.stats_rpg = d->stats_rows_per_group > 0 ? d->stats_rows_per_group : d->n * d->ho * d->wo;
    a.bstats_u = (const uint16_t*)d->bstats_u; a.bstats_bits = d->bstats_bits; a.bstats_mean = d->bstats_mean; a.bstats_rstd = d->bstats_rstd;
    CMS_REQUIRE(d->stats_out == nullptr ||
                    ((d->mode == 0 ? d->bstats_u == nullptr : (d->bstats_u && d->bstats_mean && d->bstats_rstd)) && d->out_stride == 1 && d->out_h == d->ho && d->out_w == d->wo && a.stats_rpg >= c8::BM &&
                     (d->n * d->ho * d->wo) % a.stats_rpg == 0),
                "conv8: stats_out needs sample groups of whole runs of >= 256 pixel rows (forward launches; data gradients with bstats_u / _mean / _rstd)");
    a.N = d->n; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout;
    a.ntaps = d->ntaps; a.stride = d->stride; a.out_H = d->out_h; a.out_W = d->out_w; a.out_stride = d->out_stride;
    a.relu = d->relu; a.mode = d->mode;
    a.M = d->n * d->ho * d->wo;
    a.plain = (d->ntaps == 1 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 && d->stride == 1 && d->h == d->ho && d->w_in == d->wo &&
               d->out_stride == 1 && d->out_h == d->ho && d->out_w == d->wo) ? 1 : 0;
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        a.tap_dy[i] = (short)(i < d->ntaps ? d->tap_dy[i] : 0);
        a.tap_dx[i] = (short)(i < d->ntaps ? d->tap_dx[i] : 0);
    }
    a.ntn = d->cout / c8::BN;
    const int ntm = (a.M + c8::BM - 1) / c8::BM;
    a.ntiles = ntm * a.ntn;
    a.kc_per_tap = d->cin / c8::BK;
    a.KT = d->ntaps * a.kc_per_tap;
    a.ku = a.KT % 2 == 0 ? 2 : 1;
    a.slab = nullptr; a.counters = nullptr;
    a.trace = (uint32_t*)trace; a.trace_wgs = trace_wgs;
    static int env_nt = -1;
    if (env_nt < 0) {
        const char* e = getenv("CMS_CONV8_NT");       // A/B switch, read once
        env_nt = e ? atoi(e) : 0;
    }
    a.nt_store = env_nt;
    int grid = a.ntiles;
    a.sk_tiles = 0;
    a.dp_rounds = 1;
    if (mode == 1) {
        int n_cu = 0;
        if (cms_device_info(&n_cu, nullptr, 0) != CMS_OK || n_cu <= 0) n_cu = 256;
        int G = grid_cap > 0 ? std::min(grid_cap, n_cu) : n_cu;
        const size_t units = (size_t)a.ntiles * (a.KT / a.ku);
        if (a.ntiles % G == 0 || d->workspace == nullptr || units < (size_t)G) {
            // whole tiles only: tiles g, g + G, ... per workgroup (a launch without a workspace cannot split tiles)
            G = std::min(G, a.ntiles);
            a.dp_rounds = (a.ntiles + G - 1) / G;
        } else {
            const int rounds = a.ntiles / G;                      // full rounds
            a.dp_rounds = rounds >= 1 ? rounds - 1 : 0;
            a.sk_tiles = a.ntiles - a.dp_rounds * G;              // G .. 2G-1 tiles (all of them when there are fewer than G)
            CMS_REQUIRE((size_t)a.sk_tiles * 4 <= CONV8_COUNTER_BYTES &&
                            d->workspace_bytes >= (long long)(CONV8_COUNTER_BYTES + (size_t)2 * G * c8::SLAB_FLOATS * 4),
                        "conv8: workspace of %lld bytes is too small", d->workspace_bytes);
            a.counters = (unsigned*)d->workspace;
            a.slab = (float*)((char*)d->workspace + CONV8_COUNTER_BYTES);
        }
        grid = G;
    }
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(c8::conv8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(c8::conv8_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(c8::conv8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        raised = true;
    }
    CMS_REQUIRE(a.stats_out == nullptr || a.sk_tiles == 0, "conv8: stats_out with whole tiles per workgroup only (not the stream-K launch)");
    if (a.stats_out) hipLaunchKernelGGL((c8::conv8_kernel<false, true>), dim3(grid), dim3(c8::NT), c8::LDS_BYTES, s, a);
    else if (a.sk_tiles > 0) hipLaunchKernelGGL(c8::conv8_kernel<true>, dim3(grid), dim3(c8::NT), c8::LDS_BYTES, s, a);
    else hipLaunchKernelGGL(c8::conv8_kernel<false>, dim3(grid), dim3(c8::NT), c8::LDS_BYTES, s, a);
    return launch_status("cms_conv_igemm (8-phase)");
}

}  // namespace cms
