// MFMA implicit-GEMM convolution family for the DeepLab v2 backbone (gfx950 / CDNA4, wave64).
//
// Replaces the cuDNN/MIOpen calls the reference issues through nn.Conv2d + nn.BatchNorm2d + ReLU (+ residual add) in
// architectures/deeplab2.py:89-109 (Bottleneck.forward), :124-128 (ASPP head) and their autograd twins.
//
// Data layout: activations bf16 NHWC; weights bf16 [tap][Cout][Cin] (Cin contiguous) -- the physical layout the
// parameter arena keeps conv weights in, so the bf16 copy written by the fused optimizer IS the forward operand;
// fp32 accumulation in the MFMA accumulators.
//
// GEMM view:  D[co][pixel] = sum_{tap, ci} W[tap][co][ci] * X[pixel shifted by tap][ci]
//   MFMA "A" operand = weight tile (rows = co), "B" operand = pixel tile (columns = pixels), K = taps * Cin.
//   v_mfma_f32_32x32x16_bf16: lane l holds A[i = l&31][k = 8*(l>>5)..+7] and B[k = 8*(l>>5)..+7][j = l&31]; both are
//   one ds_read_b128 of a [row][k] LDS image. Accumulator: column j = l&31 (pixel), row i = (r&3) + 8*(r>>2) + 4*(l>>5)
//   (co) -> every lane owns runs of 4 consecutive channels of one pixel = 8 contiguous bytes of the NHWC output.
//
// Workgroup: 256 threads = 4 waves, tile BN (co) x BM (pixels) x BK=64; global -> registers -> LDS staging with the
// next tile's global loads issued before the MFMA phase of the current one; LDS rows are 128 B, 16-byte chunks
// XOR-swizzled with (row>>1)&7 so that the 16-lane groups of ds_read_b128 hit 16 distinct slots (no bank conflicts);
// zero padding / tile tails are handled in the loader (out-of-range rows load zeros). Each global load instruction of
// a wave covers 8 rows x 128 contiguous bytes (full cache lines).
//
// Epilogues (fused, in registers):
//   forward   y = relu(acc * scale[co] + bias[co] + residual)      (frozen BN folded to scale/bias, deeplab2.py:92-107)
//   dgrad     dx = (acc + add_in) * [mask_src > 0]                 (ReLU backward of the producer of this conv's input)
//   optional fp32 NCHW output for the ASPP head logits.
//
// Roofline: dilated 3x3 (layer3: K = 2304, AI ~ 680 FLOP/B) is MFMA-bound; 1x1 (AI ~ 180 FLOP/B) is HBM-bound on the
// activation stream; DESIGN.md section 4 lists algorithmic FLOPs / bytes per layer shape.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "common.hpp"
#include "tile_stats.hpp"

namespace cms {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector type for 16-byte staging registers (HIP's uint4 struct copies become memcpy calls that the compiler
// leaves in scratch memory)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CONV_BK = 64;          // K elements per stage = 128 B per LDS row
constexpr int CONV_ROW_BYTES = 128;

struct ConvArgs {
    const uint16_t* x;         // bf16 [N][H][W][Cin]
    const uint16_t* w;         // bf16 [ntaps][Cout][Cin]
    uint16_t* y;               // bf16 [N][out_H][out_W][Cout] or NULL
    float* y32;                // fp32 NCHW [N][cout_real][Ho][Wo] or NULL
    const float* scale;        // [Cout] or NULL (forward)
    const float* bias;         // [Cout] or NULL (forward)
    const uint16_t* res;       // bf16, indexed like y, or NULL
    const uint16_t* mask_src;  // bf16, indexed like y, or NULL (dgrad)
    const uint16_t* zeros;     // >= 2*Cin + 128 B of zeros (source of padded / out-of-range rows for the direct-to-LDS loader)
    int N, H, W, Cin;
    int Ho, Wo, Cout, cout_real;
    int ntaps, stride;
    int out_H, out_W, out_stride;
    int relu, mode;            // mode 0 = forward epilogue, 1 = dgrad epilogue
    int M;                     // N*Ho*Wo
    int ksplit;                // > 1: the taps are split across workgroups, fp32 output accumulated with atomics
    int dbg;                   // profiling experiments only: 2 = skip the MFMA phase, 3 = skip the loads after the first
    short tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
    uint32_t* trace;        // diagnostic (variant 30): per-workgroup s_memtime stamps, CONV_TRACE_DWORDS each, or NULL
    int trace_wgs;          // workgroups the trace buffer holds
    int stagger;            // != 0: co-resident workgroups of the first dispatch round start 1/4 K-step period apart (24 / 25)
    int n_main, rem_tile_base;   // conv_igemm_mixed_kernel: workgroups of the main tile shape, first pixel tile of the rest
    int krot;               // != 0: workgroup (tile_m) starts its K loop krot * tile_m steps in and wraps (variant 20)
    int plain;              // 1x1, stride 1, tap (0, 0), output grid == input grid: GEMM row m IS input and output pixel m
    int nt_store;           // bf16 output rows as non-temporal stores (CMS_CONV_NT, A/B: streaming stores evicting re-read operands)
    uint8_t* mask_bits_out;       // forward + ReLU: bit (pixel, channel) = [y > 0], [out pixels][Cout / 8] bytes, or NULL
    const uint8_t* mask_bits;     // dgrad: the ReLU mask as such bits instead of mask_src, or NULL
    float* stats_out;             // [pixel tiles][2 slots][2][Cout] per-tile (sum, sum of squares) of the stored output, or NULL
    int stats_rpg;                // pixel rows per sample group (>= the tile's rows; M for one group)
    int mask_gates_res;           // dgrad with res + mask_bits: out = acc + (bit ? res : 0) instead of bit ? acc + res : 0
    const uint16_t* bstats_u;     // dgrad + stats_out: u of the batch-statistics unit whose output gradient this launch writes, or NULL
    const uint8_t* bstats_bits;   // ... its ReLU mask bits, or NULL (no ReLU)
    const float* bstats_mean;     // ... its statistics [groups][Cout]
    const float* bstats_rstd;
};


// Branch-free pointer select for the direct-to-LDS loads: `ok ? p : z` written as a ternary makes the compiler
// duplicate the (side-effecting) load into both arms of a divergent branch, i.e. up to two load instructions where
// the counted s_waitcnt of the ring kernels expects exactly one.
__device__ __forceinline__ const uint16_t* select_ptr(bool ok, const uint16_t* p, const uint16_t* z) {
    const uint64_t a = (uint64_t)p, b = (uint64_t)z;
    const uint64_t m = (uint64_t)0 - (uint64_t)ok;
    return (const uint16_t*)(b ^ ((a ^ b) & m));
}

__device__ __forceinline__ uint32_t swz(int row, int chunk) {
    return (uint32_t)row * CONV_ROW_BYTES + (uint32_t)((chunk ^ ((row >> 1) & 7)) << 4);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    // plain conversions: the compiler emits v_cvt_pk_bf16_f32 (round-to-nearest-even) instead of ~8 integer ops each
    bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, p);
}

constexpr int CONV_TRACE_DWORDS = 512;      // per workgroup: 16 header dwords + 6 stamps per K step (<= 82 steps)

struct RowInfo {            // one per pixel row of the workgroup tile, computed once (2 integer divisions per row)
    uint32_t in_off;        // element offset of input pixel (oy*stride, ox*stride), channel 0
    uint32_t yx;            // (oy*stride) << 16 | (ox*stride); 0x70007000 for rows past M (every bounds test fails)
    uint32_t opix;          // output pixel index, 0xffffffff for rows past M
    uint32_t m;             // GEMM row
};

// GLDS = true: operands go global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave instruction, no staging
// registers, no ds_write); the XOR swizzle is applied on the SOURCE side (the LDS image of a wave instruction is
// lane-linear), out-of-range rows read a zero page. One LDS buffer per workgroup, up to 4 workgroups per CU: the
// load latency of a workgroup is covered by the MFMA phases of its neighbours.
// Buffer-addressed direct-to-LDS load of 16 bytes per lane. The descriptor type and the builtins exist in the device
// compilation only; the host pass (which still has to instantiate the kernel to emit its launch stub) sees dummies.
//
// The loads are issued from INLINE ASSEMBLY, on purpose: hipcc counts a direct-to-LDS load it knows about (the
// __builtin_amdgcn_*_load_lds forms) as a pending LDS WRITE that any later ds_read may alias, and puts
// `s_waitcnt vmcnt(0)` in front of the first LDS read that follows it in program order -- with more than one stage in
// flight that drains the whole pipeline every K step (seen in the ISA of every stage-ring variant of rounds 1 and 2,
// which is why none of them ever beat the single-stage kernel). Loads issued from asm are invisible to that
// bookkeeping; completion is counted by the kernels' own `s_waitcnt vmcnt(N)` statements + barrier.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
struct buf_rsrc_t { i32x4_t w; };
__device__ __forceinline__ buf_rsrc_t make_buf_rsrc(const void* p, int bytes) {
    const uint64_t addr = (uint64_t)p;
    buf_rsrc_t r;
    r.w[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)addr);
    r.w[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(addr >> 32) & 0xffffu));      // stride 0: raw buffer
    r.w[2] = __builtin_amdgcn_readfirstlane(bytes);                                         // num_records (bytes)
    r.w[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
#if defined(__HIP_DEVICE_COMPILE__)
// one wave instruction: 16 bytes per lane from rsrc[voff + soff] to LDS bytes [lds, lds + 1024) lane-linearly
__device__ __forceinline__ void buf_load_lds16(const buf_rsrc_t& r, const unsigned char* lds, uint32_t voff, uint32_t soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(__builtin_amdgcn_readfirstlane((int)lds_addr_of(lds))), "v"(voff), "s"(r.w),
                   "s"(__builtin_amdgcn_readfirstlane((int)soff))
                 : "memory");
}
// four of them: LDS destinations lds + k * step (k = 0..3)
__device__ __forceinline__ void buf_load_lds16_x4(const buf_rsrc_t& r, const unsigned char* lds, uint32_t step, uint32_t v0,
                                                  uint32_t v1, uint32_t v2, uint32_t v3, uint32_t soff) {
    unsigned keep;
    const uint32_t l0 = lds_addr_of(lds);
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %9, %10 offen lds\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %9, %10 offen lds\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %9, %10 offen lds\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %8, %9, %10 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(l0), "s"(l0 + step), "s"(l0 + 2 * step), "s"(l0 + 3 * step), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(r.w),
                   "s"(soff)
                 : "memory");
}
// flat form, 4 bytes per lane (64 floats per wave instruction)
__device__ __forceinline__ void glds4_asm(const void* gsrc, const void* lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr_of(lds))
                 : "memory");
}
#else
__device__ __forceinline__ void buf_load_lds16(const buf_rsrc_t&, const unsigned char*, uint32_t, uint32_t) {}
__device__ __forceinline__ void buf_load_lds16_x4(const buf_rsrc_t&, const unsigned char*, uint32_t, uint32_t, uint32_t, uint32_t,
                                                  uint32_t, uint32_t) {}
__device__ __forceinline__ void glds4_asm(const void*, const void*) {}
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt_lds() {
    // this wave's direct-to-LDS loads except the N youngest have landed; LDS reads of the previous phase are done
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(N) : "memory");
}

// OCC = workgroups per CU the register budget is declared for (0: the round-1 defaults). PF = the fragment reads of
// K sub-step kk+1 are issued BEFORE the MFMAs of sub-step kk (register double buffer): without it every sub-step is
// "4 ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 4 MFMAs", i.e. the LDS latency of each sub-step is exposed to the wave.
// BUFA = the direct-to-LDS loads use buffer addressing (buffer_load_dwordx4 ... offen lds): descriptor in SGPRs, one
// 32-bit byte offset per lane and row piece (rewritten once per TAP), the K position in a scalar offset -- a K step then
// issues its 8 pieces per wave with NO vector ALU instruction (the flat-address form spends ~5 per piece on 64-bit
// pointer arithmetic, which tools/conv_trace.py shows as a 700..1450-cycle issue phase). Padding and rows past M are
// out-of-range offsets: the hardware bounds check writes zeros, no zero page.
// The kernel body is a device function of (logical block id, number of blocks, first pixel tile), so that ONE launch
// can run two tile shapes (conv_igemm_mixed_kernel below).
// ST = the store loop can also take BatchNorm statistics (ConvArgs.stats_out, tile_stats.hpp): separate instantiations of the
// default tiles, so that the statistics' ~40 registers cost the plain kernels nothing (the 32-channel tile went from 6 to 4
// workgroups per CU with the code folded in).
template <int WN, int WM, int TN, int TM, bool GLDS, int NS, int BK = CONV_BK, bool PF = false, int OCC = 0, bool BUFA = false,
          bool ST = false>
__device__ __forceinline__ void conv_body(const ConvArgs& a, const int bid_raw, const int nblk_in, const int m_tile_base) {
    // NS = LDS stages of the direct-to-LDS loader. 1: load -> barrier -> MFMA -> barrier; memory and MFMA phases only
    // overlap ACROSS the (up to 4) workgroups of a CU. 2: the loads of K-step k+1 are in flight during the MFMAs of
    // step k inside one workgroup -- what the DeepLab shapes need, whose grids are only ~2 workgroups per CU.
    constexpr int NW = WN * WM;         // waves per workgroup (4 or 8)
    constexpr int NT = 64 * NW;
    constexpr int BN = WN * TN * 32;    // output channels per workgroup
    constexpr int BM = WM * TM * 32;    // pixels per workgroup
    // BK = K elements per stage. 64: 128-byte LDS rows (8 chunks of 16 B, swizzle (row>>1)&7). 32: 64-byte rows
    // (4 chunks, swizzle (row>>2)&3) -- two stages of it fit the LDS footprint of one 64-wide stage, i.e. the loads of
    // step k+1 overlap the MFMAs of step k WITHOUT giving up the 4 workgroups per CU.
    constexpr int ROWB = BK * 2;        // LDS row pitch (bytes)
    constexpr int CH = BK / 8;          // 16-byte chunks per row
    constexpr int LRPI = 64 / CH;       // rows per wave-wide direct-to-LDS instruction (8 or 16)
    constexpr int WSH = CH == 8 ? 1 : 2;                 // rows per 256-byte bank window = 1 << WSH
    static_assert(BK == 64 || (BK == 32 && GLDS), "BK = 32 only with the direct-to-LDS loader");
    constexpr int PA = BM / (LRPI * NW);   // loader passes over the pixel tile
    constexpr int PB = BN / (LRPI * NW);
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(BM <= NT && PA >= 1 && PB >= 1, "tile too small for the loader");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE_BYTES = (BM + BN) * ROWB;
    constexpr int EPI_BYTES = BM * BN * 2;                // epilogue tile (bf16 output / staged residual or mask)
    static_assert(NS == 1 || (NS >= 2 && NS <= 4 && GLDS), "stage rings only with the direct-to-LDS loader");
    constexpr int UNION_BYTES = NS * STAGE_BYTES > EPI_BYTES ? NS * STAGE_BYTES : EPI_BYTES;
    unsigned char* lds_x = smem;                          // [NS][BM][128 B]   (stage s at + s * STAGE_BYTES)
    unsigned char* lds_w = smem + BM * ROWB;              // [NS][BN][128 B]
    short* lds_tap = reinterpret_cast<short*>(smem + UNION_BYTES);                       // [2][CMS_CONV_MAX_TAPS]
    RowInfo* lds_row = reinterpret_cast<RowInfo*>(smem + UNION_BYTES + 80);              // [BM]
    float* lds_sb = reinterpret_cast<float*>(smem + UNION_BYTES + 80 + BM * 16);         // [2][BN]: BN scale, bias of the tile
    uint32_t* lds_trace = reinterpret_cast<uint32_t*>(smem + UNION_BYTES + 80 + BM * 16 + 2 * BN * 4); // [CONV_TRACE_DWORDS], variant 30 only
    const bool tracing = a.trace != nullptr && (int)blockIdx.x < a.trace_wgs;
    const uint64_t t_start = tracing ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t rt_start = tracing ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int slot) {
        if (tracing) {
            const uint32_t t = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
            if (threadIdx.x == 0 && slot < CONV_TRACE_DWORDS) lds_trace[slot] = t;
        }
    };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;

    // XCD-aware tile order: consecutive logical ids (same pixel tile, different co tiles) stay on one XCD's L2
    const int ntn = a.Cout / BN;
    const int nblk = nblk_in;
    int bid = bid_raw;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntiles = nblk / a.ksplit;
    const int split = bid / ntiles;
    bid -= split * ntiles;
    const int tile_n = bid % ntn, tile_m = bid / ntn + m_tile_base;
    const int co0 = tile_n * BN, m0 = tile_m * BM;

    // ---- one-time tables in LDS: tap offsets (a dynamically indexed by-value kernel argument would go to scratch)
    // and the per-row geometry
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
            lds_tap[i] = a.tap_dy[i];
            lds_tap[CMS_CONV_MAX_TAPS + i] = a.tap_dx[i];
        }
    }
    // PLAIN (pointwise convolution over an unstrided grid -- two thirds of the network's launches): the row geometry
    // needs no division, the loader no table, and the first stage is issued BEFORE the tables are written; the first
    // barrier of the K loop publishes them (tools/conv_trace.py: 2100 cycles of table arithmetic + 1100 of barrier and
    // geometry + 1200 of tap offsets stood in front of the first load of every workgroup)
    const bool plain = GLDS && BUFA && NS == 1 && a.plain != 0;
    if (tid < BM) {
        const int m = m0 + tid;
        RowInfo ri;
        ri.m = (uint32_t)m;
        if (plain) {
            const bool live = m < a.M;
            ri.in_off = live ? (uint32_t)(m * a.Cin) : 0u;
            ri.yx = live ? 0u : 0x70007000u;
            ri.opix = live ? (uint32_t)m : 0xffffffffu;
        } else if (m < a.M) {
            const int ox = m % a.Wo;
            const int t = m / a.Wo;
            const int oy = t % a.Ho;
            const int n = t / a.Ho;
            const int iy = oy * a.stride, ix = ox * a.stride;
            ri.in_off = (uint32_t)(((n * a.H + iy) * a.W + ix) * a.Cin);
            ri.yx = ((uint32_t)iy << 16) | (uint32_t)ix;
            ri.opix = (uint32_t)((n * a.out_H + oy * a.out_stride) * a.out_W + ox * a.out_stride);
        } else {
            ri.in_off = 0;
            ri.yx = 0x70007000u;
            ri.opix = 0xffffffffu;
        }
        lds_row[tid] = ri;
    }
    stamp(13);                                             // tables written (before the barrier)
    // BN affine of the tile's channels: fetched here, so that the epilogue has no global load of its own in front of
    // its arithmetic (tools/conv_trace.py: the per-element float4 loads cost ~10k cycles per workgroup)
    static_assert(BN <= NT, "scale / bias staging");
    const bool use_scale = a.scale && a.mode == 0, use_bias = a.bias && a.mode == 0 && split == 0;   // (1, 0) in the dgrad epilogue
    if constexpr (GLDS && BN >= 64) {
        // asynchronously (4-byte direct-to-LDS loads, 64 floats per wave instruction): they land with the first stage
        for (int wave_u = __builtin_amdgcn_readfirstlane(wave); wave_u < 2 * BN / 64; wave_u += NW) {      // wave-uniform
            const int e = wave_u * 64 + lane;                        // element of [scale BN | bias BN]
            const bool is_scale = e < BN;
            if ((is_scale && use_scale) || (!is_scale && use_bias)) {
                const float* src = is_scale ? a.scale + co0 + e : a.bias + co0 + (e - BN);
                glds4_asm(src, lds_sb + wave_u * 64);
            } else {
                lds_sb[e] = is_scale ? 1.0f : 0.0f;
            }
        }
    } else if (tid < BN) {
        lds_sb[tid] = use_scale ? a.scale[co0 + tid] : 1.0f;
        lds_sb[BN + tid] = use_bias ? a.bias[co0 + tid] : 0.0f;
    }
    if (!plain) __syncthreads();

    // ---- loader geometry (fixed for the whole K loop)
    // register-staged: thread -> 16-byte chunk (tid & 7) of rows (tid >> 3) + 32*i, swizzled on the LDS side.
    // direct-to-LDS:   wave instruction (pass i, wave w) fills rows (4i+w)*8 .. +7 lane-linearly, so lane l lands on
    //                  row (4i+w)*8 + (l>>3), physical chunk l&7 and must FETCH logical chunk (l&7) ^ swizzle(row).
    const int chunk = tid & 7, lrow = tid >> 3;      // register-staged mapping: rows lrow + (NT/8)*i
    constexpr int RSTEP = NT / 8;
    uint32_t xoff[GLDS ? 1 : PA], xyx[GLDS ? 1 : PA], woff[PB];
    if constexpr (!GLDS) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const RowInfo ri = lds_row[lrow + RSTEP * i];
            xoff[i] = ri.in_off + (uint32_t)(chunk * 8);
            xyx[i] = ri.yx;
        }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = GLDS ? (NW * i + wave) * LRPI + lane / CH : lrow + RSTEP * i;
        const int c = GLDS ? ((lane % CH) ^ ((row >> WSH) & (CH - 1))) : chunk;
        woff[i] = (uint32_t)(row * a.Cin + c * 8) * (BUFA ? 2u : 1u);     // elements (bytes under buffer addressing)
    }
    const uint32_t st_off = swz(lrow, chunk);            // rows lrow + RSTEP*i share the swizzle term (RSTEP % 16 == 0)

    const int kc_per_tap = a.Cin / BK;
    const int taps_per_split = (a.ntaps + a.ksplit - 1) / a.ksplit;
    const int tap_begin = split * taps_per_split;
    const int tap_end = min(a.ntaps, tap_begin + taps_per_split);
    const int ks_begin = tap_begin * kc_per_tap;
    const int ksteps = tap_end * kc_per_tap;        // exclusive end of this workgroup's K range

    u32x4 rx[GLDS ? 1 : PA], rw[GLDS ? 1 : PB];
    auto load_tile = [&](int ks, int) {        // register-staged loader only (the direct-to-LDS cursor is below)
        if constexpr (!GLDS) {
            const int tap = ks / kc_per_tap;                                  // wave-uniform
            const int c0 = (ks - tap * kc_per_tap) * BK;
            const int dy = __builtin_amdgcn_readfirstlane((int)lds_tap[tap]);
            const int dx = __builtin_amdgcn_readfirstlane((int)lds_tap[CMS_CONV_MAX_TAPS + tap]);
            const int delta = (dy * a.W + dx) * a.Cin + c0;                   // scalar element offset of this tap / K chunk
            const uint16_t* wtp = a.w + ((size_t)tap * a.Cout + co0) * a.Cin + c0;   // scalar base
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const uint32_t iy = (xyx[i] >> 16) + (uint32_t)dy, ix = (xyx[i] & 0xffffu) + (uint32_t)dx;
                const bool ok = iy < (uint32_t)a.H && ix < (uint32_t)a.W;    // unsigned compare covers the negative side
                if (ok) rx[i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)(xoff[i] + (uint32_t)delta));
                else rx[i] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) rw[i] = *reinterpret_cast<const u32x4*>(wtp + woff[i]);
        }
    };
    auto store_tile = [&]() {
        if constexpr (!GLDS) {
#pragma unroll
            for (int i = 0; i < PA; ++i) *reinterpret_cast<u32x4*>(lds_x + st_off + i * RSTEP * CONV_ROW_BYTES) = rx[i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<u32x4*>(lds_w + st_off + i * RSTEP * CONV_ROW_BYTES) = rw[i];
        }
    };

    // ---- direct-to-LDS loader as a (tap, K-chunk) cursor. Everything that depends on the tap only -- the tap offsets
    // (one LDS read), the bounds test of each fetched row, its 64-bit source address or the zero page -- is set up
    // ONCE per tap; a K step then costs one 64-bit add per wave instruction (the zero page is as long as a pixel's
    // channel run, so dead rows advance like live ones). This took the load phase from ~125 to ~40 instructions per
    // stage and removed an LDS round trip from every stage's critical path.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);            // scalar: LDS destinations live in SGPRs / M0
    const uint16_t* xaddr[(GLDS && !BUFA) ? PA : 1];
    const uint16_t* wt = a.w;
    static_assert(!BUFA || GLDS, "buffer addressing belongs to the direct-to-LDS loader");
    uint32_t xvoff[BUFA ? PA : 1];                   // BUFA: byte offset of this lane's 16 bytes of row piece i (tap applied)
    uint32_t soff_x = 0, soff_w = 0;                 // BUFA: scalar byte offsets (K position; tap and channel tile of W)
    const buf_rsrc_t rsrc_x = make_buf_rsrc(a.x, BUFA ? a.N * a.H * a.W * a.Cin * 2 : 0);
    const buf_rsrc_t rsrc_w = make_buf_rsrc(a.w, BUFA ? a.ntaps * a.Cout * a.Cin * 2 : 0);
    if (a.stagger != 0) {
        // the first 1024 workgroups start at the same instant, 4 per CU, and would run their load and MFMA phases in
        // lockstep; dispatch is round-robin over 8 XCDs x 32 CUs, so blockIdx / 256 is the slot on the CU
        const int slot = (blockIdx.x >> 8) & 3;
        for (int i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(27);
    }
    int cur_tap = tap_begin, cur_kc = 0;
    if (a.krot != 0 && ksteps > ks_begin) {
        // K rotation: workgroups of different pixel tiles walk the (tap, chunk) sequence from different starting
        // points, so at any instant they request DIFFERENT weight lines from L2 (fp32 accumulation: the sum only
        // changes its order)
        const int r = (int)(((unsigned)tile_m * (unsigned)a.krot) % (unsigned)(ksteps - ks_begin));
        cur_tap = tap_begin + r / kc_per_tap;
        cur_kc = r - (r / kc_per_tap) * kc_per_tap;
    }
    auto setup_tap = [&]() {
        if constexpr (BUFA) {
            // (one formula for the scalar offsets on both paths: a value merged from two branches is no longer provably
            // wave-uniform for the "s" operand of the inline-asm loads)
            soff_x = (uint32_t)(cur_kc * BK * 2);
            soff_w = (uint32_t)((((cur_tap * a.Cout + co0) * a.Cin) + cur_kc * BK) * 2);
            if (plain) {                       // no table read: nothing here depends on the (not yet published) LDS tables
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    const int row = (NW * i + wave) * LRPI + lane / CH;
                    const int c = (lane % CH) ^ ((row >> WSH) & (CH - 1));
                    const int m = m0 + row;
                    xvoff[i] = m < a.M ? (uint32_t)(m * a.Cin + c * 8) * 2u : 0x80000000u;
                }
            } else {
                const int dy = __builtin_amdgcn_readfirstlane((int)lds_tap[cur_tap]);
                const int dx = __builtin_amdgcn_readfirstlane((int)lds_tap[CMS_CONV_MAX_TAPS + cur_tap]);
                const int delta = (dy * a.W + dx) * a.Cin;                     // scalar element offset of this tap
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    const int row = (NW * i + wave) * LRPI + lane / CH;
                    const int c = (lane % CH) ^ ((row >> WSH) & (CH - 1));
                    const RowInfo ri = lds_row[row];
                    const uint32_t iy = (ri.yx >> 16) + (uint32_t)dy, ix = (ri.yx & 0xffffu) + (uint32_t)dx;
                    const bool ok = iy < (uint32_t)a.H && ix < (uint32_t)a.W;    // unsigned compare covers the negative side
                    xvoff[i] = ok ? (ri.in_off + (uint32_t)(c * 8) + (uint32_t)delta) * 2u : 0x80000000u;   // out of range: zeros
                }
            }
            return;
        }
        const int dy = __builtin_amdgcn_readfirstlane((int)lds_tap[cur_tap]);
        const int dx = __builtin_amdgcn_readfirstlane((int)lds_tap[CMS_CONV_MAX_TAPS + cur_tap]);
        const int delta = (dy * a.W + dx) * a.Cin;                     // scalar element offset of this tap
        wt = a.w + ((size_t)cur_tap * a.Cout + co0) * a.Cin + cur_kc * BK;   // scalar base (cur_kc != 0 only when rotated)
#pragma unroll
        for (int i = 0; i < ((GLDS && !BUFA) ? PA : 0); ++i) {
            const int row = (NW * i + wave) * LRPI + lane / CH;
            const int c = (lane % CH) ^ ((row >> WSH) & (CH - 1));
            const RowInfo ri = lds_row[row];
            const uint32_t iy = (ri.yx >> 16) + (uint32_t)dy, ix = (ri.yx & 0xffffu) + (uint32_t)dx;
            const bool ok = iy < (uint32_t)a.H && ix < (uint32_t)a.W;    // unsigned compare covers the negative side
            xaddr[i] = select_ptr(ok, a.x + (size_t)(ri.in_off + (uint32_t)(c * 8) + (uint32_t)delta),
                                  a.zeros + (lane & 7) * 8) + cur_kc * BK;
        }
    };
    auto issue_loads = [&](int buf) {
        if constexpr (BUFA) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                buf_load_lds16(rsrc_x, lds_x + buf * STAGE_BYTES + (NW * i + wave_s) * 1024, xvoff[i], soff_x);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                buf_load_lds16(rsrc_w, lds_w + buf * STAGE_BYTES + (NW * i + wave_s) * 1024, woff[i], soff_w);
            return;
        }
#pragma unroll
        for (int i = 0; i < ((GLDS && !BUFA) ? PA : 0); ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xaddr[i],
                                             (__attribute__((address_space(3))) void*)(lds_x + buf * STAGE_BYTES + (NW * i + wave_s) * 1024),
                                             16, 0, 0);
#pragma unroll
        for (int i = 0; i < (GLDS ? PB : 0); ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wt + woff[i]),
                                             (__attribute__((address_space(3))) void*)(lds_w + buf * STAGE_BYTES + (NW * i + wave_s) * 1024),
                                             16, 0, 0);
    };
    auto advance = [&]() {
        if constexpr (BUFA) {
            soff_x += BK * 2;
            soff_w += BK * 2;
        } else {
            wt += BK;
#pragma unroll
            for (int i = 0; i < (GLDS ? PA : 0); ++i) xaddr[i] += BK;
        }
        if (++cur_kc == kc_per_tap) {
            cur_kc = 0;
            if (++cur_tap == tap_end) cur_tap = tap_begin;       // wraps only under K rotation (else this is past the last step)
            setup_tap();
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read address: row (lane & 31) of a 32-row tile, 16-byte chunk (kk*2 + lane>>5) ^ swizzle(row)
    //   = lane_frag ^ (kk * 32)   (the K sub-step only flips bits 5..6)
    const int frow = lane & 31, fhalf = lane >> 5;
    const uint32_t lane_frag = (uint32_t)frow * ROWB | (uint32_t)(((fhalf ^ (frow >> WSH)) & (CH - 1)) << 4);
    const unsigned char* fw_base = lds_w + wn * TN * 32 * ROWB;
    const unsigned char* fx_base = lds_x + wm * TM * 32 * ROWB;

    auto mfma_phase = [&](int buf) {
        if constexpr (PF) {
            // register double buffer over the K sub-steps: reads of kk+1 in flight during the MFMAs of kk
            constexpr int KK = BK / 16;
            u32x4 fw[2][TN], fx[2][TM];
            auto frag_read = [&](int kk, u32x4* w_, u32x4* x_) {
                const uint32_t fo = (lane_frag ^ (uint32_t)(kk * 32)) + (uint32_t)(buf * STAGE_BYTES);
#pragma unroll
                for (int i = 0; i < TN; ++i) w_[i] = *reinterpret_cast<const u32x4*>(fw_base + fo + i * 32 * ROWB);
#pragma unroll
                for (int j = 0; j < TM; ++j) x_[j] = *reinterpret_cast<const u32x4*>(fx_base + fo + j * 32 * ROWB);
            };
            frag_read(0, fw[0], fx[0]);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + 1 < KK) frag_read(kk + 1, fw[(kk + 1) & 1], fx[(kk + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);           // keep the reads of kk+1 in front of the MFMAs of kk
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[kk & 1][i]),
                                                                            __builtin_bit_cast(bf16x8, fx[kk & 1][j]),
                                                                            acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            u32x4 fw[TN], fx[TM];     // (arrays of __bf16 vectors are not promoted to registers by the compiler)
            const uint32_t fo = (lane_frag ^ (uint32_t)(kk * 32)) + (uint32_t)(buf * STAGE_BYTES);
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const u32x4*>(fw_base + fo + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < TM; ++j) fx[j] = *reinterpret_cast<const u32x4*>(fx_base + fo + j * 32 * ROWB);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[i]),
                                                                        __builtin_bit_cast(bf16x8, fx[j]), acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (GLDS) {
        if (ks_begin < ksteps) {
            stamp(14);                                     // barrier passed, loader geometry / accumulators set up
            setup_tap();
            stamp(15);                                     // first tap's offsets computed
            issue_loads(0);
        }
    } else {
        if (ks_begin < ksteps) load_tile(ks_begin, 0);
    }
    if constexpr (GLDS && NS >= 3) {
        // Ring of NS stages, NS-1 of them in flight: counted s_waitcnt vmcnt (never 0 in the steady state) + a raw
        // s_barrier, ONE barrier per K step. At the barrier of step k every wave has (a) waited for its own loads of
        // stage k and (b) finished the fragment reads of step k-1, so the buffer refilled right after the barrier
        // (stage k+NS-1 -> buffer (k-1) % NS) is free and the one the MFMAs read (k % NS) is complete.
        constexpr int LPS = PA + PB;                           // direct-to-LDS instructions per wave per stage
        static_assert(LPS * (NS - 2) <= 63, "vmcnt field");
        const int total = ksteps - ks_begin;
        for (int s = 1; s < NS - 1; ++s)
            if (s < total) {
                advance();
                issue_loads(s);
            }
        int buf = 0;
        for (int ks = ks_begin; ks < ksteps; ++ks) {
            const int behind = ksteps - 1 - ks;                 // stages issued after stage ks that may stay in flight
            if (behind >= NS - 2) wait_vmcnt_lds<LPS * (NS - 2)>();
            else if (NS == 4 && behind == 1) wait_vmcnt_lds<LPS>();
            else wait_vmcnt_lds<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ks + NS - 1 < ksteps) {
                advance();
                int nb = buf + NS - 1;
                if (nb >= NS) nb -= NS;
                issue_loads(nb);
            }
            mfma_phase(buf);
            if (++buf == NS) buf = 0;
        }
        __syncthreads();                                        // the epilogue reuses the staging area
    } else if constexpr (GLDS && NS == 2) {
        int buf = 0;
        for (int ks = ks_begin; ks < ksteps; ++ks, buf ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of stage `buf` has landed in LDS
            __syncthreads();                                    // ... everybody else's too, and all fragment reads of
                                                                // the previous step (the other buffer) are done
            if (ks + 1 < ksteps && a.dbg != 3) {                // in flight during the MFMAs below
                advance();
                issue_loads(buf ^ 1);
            }
            if (a.dbg != 2) mfma_phase(buf);
        }
        __syncthreads();                                        // the epilogue reuses the staging area
    } else if constexpr (GLDS) {
        stamp(4);                                               // prologue done (tables, first stage issued)
        for (int ks = ks_begin; ks < ksteps; ++ks) {
            const int tb = 16 + (ks - ks_begin) * 6;
            stamp(tb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the stage has landed in LDS
            stamp(tb + 1);
            __syncthreads();                                    // ... and everybody else's
            stamp(tb + 2);
            if (a.dbg != 2) mfma_phase(0);
            stamp(tb + 3);
            __syncthreads();                                    // all fragment reads done: the buffer may be refilled
            stamp(tb + 4);
            if (ks + 1 < ksteps && a.dbg != 3) {
                advance();
                issue_loads(0);
            }
            stamp(tb + 5);
        }
        stamp(5);                                               // K loop done
    } else {
        for (int ks = ks_begin; ks < ksteps; ++ks) {
            __syncthreads();            // previous stage's fragment reads are done
            store_tile();
            __syncthreads();
            load_tile(ks + 1 < ksteps ? ks + 1 : ks, 0);   // in flight during the MFMA phase (last one: harmless re-load)
            mfma_phase(0);
        }
    }

    if (ks_begin >= ksteps) {                       // empty K range (a tap split past the last tap): nothing waited yet
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // ---- epilogue. The accumulator layout gives every lane runs of 4 consecutive channels of one pixel (8 bytes of
    // NHWC). BN affine, residual / gradient add, ReLU or ReLU-mask are applied in registers (one rounding to bf16).
    // Global traffic of the epilogue is row-contiguous on BOTH sides: the bf16 output tile is transposed through LDS
    // and written with 16 bytes per lane, and the residual / mask tiles are fetched global -> LDS with the
    // direct-to-LDS loader (16 bytes per lane, whole 256-byte rows) and picked up from there in accumulator layout --
    // 8-byte global reads per lane in that layout cost +50 % on the 256 -> 1024 convolutions (tools/epi_probe.py).
    // LDS tile: [BM][BN] bf16, 16-byte chunks XOR-swizzled with the row; the staged operand and the output tile share
    // the layout, so a lane reads its 8 bytes of residual / mask and writes its 8 bytes of output IN PLACE.
    constexpr int CPR = BN / 8;                 // 16-byte chunks per tile row
    constexpr int EROW = BN * 2;                // tile row pitch (bytes)
    constexpr int RPI = 64 / CPR;               // tile rows filled by one wave-wide direct-to-LDS instruction
    static_assert(BM % (NW * RPI) == 0, "epilogue staging passes");
    const bool to_lds = a.y != nullptr;
    const bool staged = GLDS && to_lds;         // residual / mask through LDS (needs the zero page for dead rows)
    if (to_lds && !GLDS) __syncthreads();       // all fragment reads of the last stage are done
    auto stage_tile = [&](const uint16_t* src) {
#pragma unroll
        for (int i = 0; i < BM / (NW * RPI); ++i) {
            const int row = (NW * i + wave) * RPI + lane / CPR;
            const int clog = (lane % CPR) ^ (row & (CPR - 1));
            const uint32_t op = lds_row[row].opix;
            const uint16_t* p = select_ptr(op != 0xffffffffu, src + (size_t)op * a.Cout + co0 + clog * 8,
                                           a.zeros + (lane & 7) * 8);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                             (__attribute__((address_space(3))) void*)(smem + (NW * i + wave) * 1024),
                                             16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    auto slot = [&](int prow_l, int co_l) -> uint32_t {
        return (uint32_t)(prow_l * EROW + ((((co_l >> 3) ^ prow_l) & (CPR - 1)) << 4) + ((co_l & 4) << 1));
    };
    uint64_t mbits[2] = {0, 0};                  // ReLU mask of this lane's (<= 128) elements when BOTH operands are staged
    static_assert(TN * TM * 16 <= 128, "mask bit field");
    // (round 4) the ReLU mask of a data gradient can come as BITS the producing forward launch wrote (mask_bits: 1/16 of the
    // bytes of the bf16 activation it would otherwise re-read -- 69 MB per layer-3 expansion at cfg 2): one dword per lane and
    // (i, j) = the 32 channels of an MFMA tile of its pixel row, straight from global memory into the same bit field
    const bool bits_in = staged && a.mask_bits != nullptr;
    const bool both = staged && a.res && a.mask_src && !bits_in;
    if constexpr (GLDS) {
        if (bits_in) {
            const int bpr = a.Cout >> 3;                 // bytes per pixel row
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const uint32_t op = lds_row[(wm * TM + j) * 32 + frow].opix;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    uint32_t wbits = 0u;
                    if (op != 0xffffffffu)
                        wbits = *reinterpret_cast<const uint32_t*>(a.mask_bits + (size_t)op * bpr + ((co0 + (wn * TN + i) * 32) >> 3));
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int bit = ((i * TM + j) * 4 + q) * 4;
                        const uint64_t m4 = (wbits >> (8 * q + 4 * fhalf)) & 0xfu;
                        mbits[bit >> 6] |= m4 << (bit & 63);
                    }
                }
            }
        }
        if (both) {
            stage_tile(a.mask_src);
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint2 mk = *reinterpret_cast<const uint2*>(
                            smem + slot((wm * TM + j) * 32 + frow, (wn * TN + i) * 32 + 8 * q + 4 * fhalf));
                        const int bit = ((i * TM + j) * 4 + q) * 4;
                        uint64_t m4 = ((int16_t)(mk.x & 0xffffu) > 0 ? 1u : 0u) | ((int16_t)(mk.x >> 16) > 0 ? 2u : 0u) |
                                      ((int16_t)(mk.y & 0xffffu) > 0 ? 4u : 0u) | ((int16_t)(mk.y >> 16) > 0 ? 8u : 0u);
                        mbits[bit >> 6] |= m4 << (bit & 63);
                    }
            __syncthreads();                     // everybody has its bits: the tile may be overwritten
        }
        if (staged && (a.res || (a.mask_src && !bits_in))) stage_tile(a.res ? a.res : a.mask_src);
    }
    // value of this lane's 4 consecutive channels (i, q) of pixel row j: BN affine, residual / gradient add, ReLU or
    // ReLU mask. Reads the staged operand from `cell` when there is one (so it must run before the cell is overwritten).
    auto element = [&](int i, int j, int q, bool valid, size_t obase, const unsigned char* cell, float (&v)[4]) {
        const int co_l = (wn * TN + i) * 32 + 8 * q + 4 * fhalf;
        const int co = co0 + co_l;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
        if (a.mode == 0) {
            const float4 sc = *reinterpret_cast<const float4*>(lds_sb + co_l);
            const float4 b = *reinterpret_cast<const float4*>(lds_sb + BN + co_l);
            v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (a.res && (staged || valid)) {
            const uint2 rr = staged ? *reinterpret_cast<const uint2*>(cell)
                                    : *reinterpret_cast<const uint2*>(a.res + obase + co);
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
        }
        if (a.mode == 0) {
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
            }
        } else if (both || bits_in) {
            const int bit = ((i * TM + j) * 4 + q) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((mbits[bit >> 6] >> ((bit & 63) + e)) & 1u) ? v[e] : 0.0f;
        } else if (a.mask_src && (staged || valid)) {
            const uint2 mk = staged ? *reinterpret_cast<const uint2*>(cell)
                                    : *reinterpret_cast<const uint2*>(a.mask_src + obase + co);
            // bf16 > 0  <=>  as a signed 16-bit integer it is > 0 (NaNs with the sign bit clear count as > 0,
            // like the float comparison the reference's ReLU backward makes on NaN-free activations)
            v[0] = (int16_t)(mk.x & 0xffffu) > 0 ? v[0] : 0.0f;
            v[1] = (int16_t)(mk.x >> 16) > 0 ? v[1] : 0.0f;
            v[2] = (int16_t)(mk.y & 0xffffu) > 0 ? v[2] : 0.0f;
            v[3] = (int16_t)(mk.y >> 16) > 0 ? v[3] : 0.0f;
        }
    };
    // Two loop nests instead of one with both outputs in its body: the fp32 NCHW output (ASPP head / classifier
    // logits; divisions, per-element stores or atomics, per-lane branches) kept the compiler from scheduling the bf16
    // path -- every (i, j, q) iteration became its own load / wait / branch island.
    if (a.y32) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow_l = (wm * TM + j) * 32 + frow;
            const RowInfo ri = lds_row[prow_l];
            const bool valid = ri.opix != 0xffffffffu;
            const size_t obase = (size_t)ri.opix * a.Cout;
            const int m = (int)ri.m;
            const int ox = m % a.Wo;
            const int t = m / a.Wo;
            const int oy = t % a.Ho;
            const int n = t / a.Ho;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co_l = (wn * TN + i) * 32 + 8 * q + 4 * fhalf;
                    const int co = co0 + co_l;
                    float v[4];
                    element(i, j, q, valid, obase, smem + slot(prow_l, co_l), v);
                    if (valid) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (co + e < a.cout_real) {
                                float* dst = a.y32 + (((size_t)n * a.cout_real + co + e) * a.Ho + oy) * a.Wo + ox;
                                if (a.ksplit > 1) atomicAdd(dst, v[e]);
                                else *dst = v[e];
                            }
                        }
                    }
                }
            }
        }
    }
    // bf16 output, operands (if any) staged in LDS: one straight-line nest per epilogue kind, chosen once per
    // workgroup (wave-uniform), so that the compiler sees 32 independent LDS read -> arithmetic -> LDS write chains
    auto nest = [&](auto RES_, auto MSK_, auto RELU_, auto BITS_) {
        constexpr int RES = decltype(RES_)::value;      // 1: residual / gradient add from the staged tile
        constexpr int MSK = decltype(MSK_)::value;      // 1: ReLU mask bits (both operands / mask_bits), 2: mask from the staged tile,
                                                        // 3: mask_bits gate the residual only (mask_gates_res)
        constexpr int RELU = decltype(RELU_)::value;
        constexpr int BITS = decltype(BITS_)::value;    // 1: also write [y > 0] of the ROUNDED output as bits (mask_bits_out)
        uint32_t obits[BITS ? TN : 1][BITS ? TM : 1];
#pragma unroll
        for (int i = 0; i < (BITS ? TN : 1); ++i)
#pragma unroll
            for (int j = 0; j < (BITS ? TM : 1); ++j) obits[i][j] = 0u;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co_l = (wn * TN + i) * 32 + 8 * q + 4 * fhalf;
                const float4 sc = *reinterpret_cast<const float4*>(lds_sb + co_l);
                const float4 b = *reinterpret_cast<const float4*>(lds_sb + BN + co_l);
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int prow_l = (wm * TM + j) * 32 + frow;
                    unsigned char* cell = smem + slot(prow_l, co_l);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                    v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    if constexpr (RES == 1 || MSK == 2) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(cell);
                        if constexpr (RES == 1 && MSK == 3) {
                            // the mask bits gate the RESIDUAL (gradient of a shortcut whose ReLU mask they are), not the sum
                            const int bit = ((i * TM + j) * 4 + q) * 4;
                            const uint32_t m4 = (uint32_t)(mbits[bit >> 6] >> (bit & 63));
                            v[0] += (m4 & 1u) ? __uint_as_float(rr.x << 16) : 0.0f; v[1] += (m4 & 2u) ? __uint_as_float(rr.x & 0xffff0000u) : 0.0f;
                            v[2] += (m4 & 4u) ? __uint_as_float(rr.y << 16) : 0.0f; v[3] += (m4 & 8u) ? __uint_as_float(rr.y & 0xffff0000u) : 0.0f;
                        } else if constexpr (RES == 1) {
                            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
                            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
                        } else {
                            v[0] = (int16_t)(rr.x & 0xffffu) > 0 ? v[0] : 0.0f;
                            v[1] = (int16_t)(rr.x >> 16) > 0 ? v[1] : 0.0f;
                            v[2] = (int16_t)(rr.y & 0xffffu) > 0 ? v[2] : 0.0f;
                            v[3] = (int16_t)(rr.y >> 16) > 0 ? v[3] : 0.0f;
                        }
                    }
                    if constexpr (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    if constexpr (MSK == 1) {
                        const int bit = ((i * TM + j) * 4 + q) * 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ((mbits[bit >> 6] >> ((bit & 63) + e)) & 1u) ? v[e] : 0.0f;
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
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
            for (int j = 0; j < TM; ++j) {
                const uint32_t op = lds_row[(wm * TM + j) * 32 + frow].opix;
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    // lanes l and l + 32 hold the two nibbles of every byte of the pixel row's 32 channels
                    const uint32_t wbits = obits[i][j] | (uint32_t)__shfl_xor((int)obits[i][j], 32, 64);
                    if (fhalf == 0 && op != 0xffffffffu)
                        *reinterpret_cast<uint32_t*>(a.mask_bits_out + (size_t)op * bpr + ((co0 + (wn * TN + i) * 32) >> 3)) = wbits;
                }
            }
        }
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    if (to_lds && (staged || (!a.res && !a.mask_src))) {
        if (a.mode == 0) {
            if (a.relu && a.mask_bits_out) { if (a.res) nest(K1{}, K0{}, K1{}, K1{}); else nest(K0{}, K0{}, K1{}, K1{}); }
            else if (a.res) { if (a.relu) nest(K1{}, K0{}, K1{}, K0{}); else nest(K1{}, K0{}, K0{}, K0{}); }
            else { if (a.relu) nest(K0{}, K0{}, K1{}, K0{}); else nest(K0{}, K0{}, K0{}, K0{}); }
        } else {
            if (bits_in && a.res && a.mask_gates_res) nest(K1{}, std::integral_constant<int, 3>{}, K0{}, K0{});
            else if (both || (bits_in && a.res)) nest(K1{}, K1{}, K0{}, K0{});
            else if (bits_in) nest(K0{}, K1{}, K0{}, K0{});
            else if (a.mask_src) nest(K0{}, K2{}, K0{}, K0{});
            else if (a.res) nest(K1{}, K0{}, K0{}, K0{});
            else nest(K0{}, K0{}, K0{}, K0{});
        }
    } else if (to_lds) {            // register-staged loader: operands come from global memory in accumulator layout
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int prow_l = (wm * TM + j) * 32 + frow;
            const uint32_t opix = lds_row[prow_l].opix;
            const bool valid = opix != 0xffffffffu;
            const size_t obase = (size_t)opix * a.Cout;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co_l = (wn * TN + i) * 32 + 8 * q + 4 * fhalf;
                    unsigned char* cell = smem + slot(prow_l, co_l);
                    float v[4];
                    element(i, j, q, valid, obase, cell, v);
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(cell) = o;
                }
            }
        }
    }
    stamp(6);                                       // operands staged, epilogue arithmetic done
    if (to_lds) {
        __syncthreads();
        constexpr int RPP = NT / CPR;               // rows per pass
        // thread -> LOGICAL chunk ch of rows r0, r0 + RPP, ... (its place in the swizzled LDS row changes with the row): the
        // 8 channels a thread stores are the same in every pass, which is what the statistics below accumulate over
        const int ch = tid % CPR, r0 = tid / CPR;
        if constexpr (!ST) {
#pragma unroll
            for (int r = r0; r < BM; r += RPP) {
                const uint32_t op = lds_row[r].opix;
                if (op != 0xffffffffu) {
                    const u32x4 val = *reinterpret_cast<const u32x4*>(smem + r * EROW + ((ch ^ (r & (CPR - 1))) << 4));
                    u32x4* dst = reinterpret_cast<u32x4*>(a.y + (size_t)op * a.Cout + co0 + ch * 8);
#if defined(__HIP_DEVICE_COMPILE__)
                    if (a.nt_store) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(val) : "memory");
                    else
#endif
                        *dst = val;
                }
            }
        } else {
            // BatchNorm statistics out of the epilogue (round 5, tile_stats.hpp): per-channel sums over what this tile stores --
            // forward launches (sum, sum of squares) of the output; data-gradient launches with bstats_u (sum d, sum d xhat) of the
            // batch-statistics unit whose output gradient the launch writes
            auto rows = [&](auto KIND_, auto TWO_, TileStats& ts, int boundary, const ts_f32x2 (&mu0)[4], const ts_f32x2 (&rs0)[4],
                            const ts_f32x2 (&mu1)[4], const ts_f32x2 (&rs1)[4]) {
                constexpr int KIND = decltype(KIND_)::value;              // 0: store, 1: + forward statistics, 2: + backward statistics
                constexpr bool TWO = decltype(TWO_)::value != 0;          // the tile straddles a sample-group boundary (both slots)
#pragma unroll
                for (int r = r0; r < BM; r += RPP) {
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
            TileStats ts;
            ts_f32x2 mu0[4] = {}, rs0[4] = {}, mu1[4] = {}, rs1[4] = {};      // (mu = MINUS the mean)
            if (a.stats_out == nullptr) {
                rows(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ts, 0, mu0, rs0, mu1, rs1);
            } else {
                const int g0 = m0 / a.stats_rpg;
                const int boundary = (g0 + 1) * a.stats_rpg;                    // first row of the next sample group
                const int m_end = m0 + BM < a.M ? m0 + BM : a.M;
                ts.zero();
                if (a.bstats_u == nullptr) {
                    if (boundary < m_end) rows(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, ts, boundary, mu0, rs0, mu1, rs1);
                    else rows(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, ts, boundary, mu0, rs0, mu1, rs1);
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
                    if (boundary < m_end) rows(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, ts, boundary, mu0, rs0, mu1, rs1);
                    else rows(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, ts, boundary, mu0, rs0, mu1, rs1);
                }
                __syncthreads();                    // the tile has been read by every wave: its LDS is the scratch now
                tile_stats_finish<CPR, NW, NT>(ts, boundary < m_end, reinterpret_cast<float*>(smem),
                                               a.stats_out + (size_t)tile_m * 4 * a.Cout + co0, a.Cout);
            }
        }
    }
    if (tracing) {                                  // diagnostic dump: header + stamps of thread 0 (tools/conv_trace.py)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(7);                                   // output stores acknowledged
        __syncthreads();
        uint32_t* out = a.trace + (size_t)blockIdx.x * CONV_TRACE_DWORDS;
        if (tid == 0) {
            out[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID: wave / simd / cu / sh / se
            out[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
            out[2] = (uint32_t)rt_start; out[3] = (uint32_t)(rt_start >> 32);
            out[8] = (uint32_t)(ksteps - ks_begin);
            out[9] = (uint32_t)tile_m; out[10] = (uint32_t)tile_n;
            out[11] = (uint32_t)t_start; out[12] = (uint32_t)(t_start >> 32);
        }
        for (int i = 4 + tid; i < CONV_TRACE_DWORDS; i += NT)
            if (i < 8 || i >= 13) out[i] = lds_trace[i];
    }
}

template <int WN, int WM, int TN, int TM, bool GLDS, int NS, int BK = CONV_BK, bool PF = false, int OCC = 0, bool BUFA = false,
          bool ST = false>
__global__ __launch_bounds__(64 * WN * WM, OCC > 0 ? (OCC * WN * WM + 3) / 4
                                           : ((GLDS && WN * WM == 4 && TN * TM == 4) ? ((NS == 1 || BK == 32) ? 4 : 2)
                                              : ((WN * WM == 4 && TN * TM == 8) ? 2 : 1))) void conv_igemm_kernel(ConvArgs a) {
    conv_body<WN, WM, TN, TM, GLDS, NS, BK, PF, OCC, BUFA, ST>(a, (int)blockIdx.x, (int)gridDim.x, 0);
}

// Two tile shapes in one launch. A layer of T = pixel tiles x channel tiles workgroups of the 128 x 128 tile leaves
// T mod 256 of them for a last, nearly empty round of the 256 CUs (DeepLab v2 at 321 x 321: 526 = 2 * 256 + 14, the
// CUs that get a third workgroup finish 26 % after the others; tools/conv_trace.py). Here the first n_main = a multiple
// of 256 workgroups run the 128 x 128 tile over pixel tiles [0, rem_tile_base), and the pixel tiles behind them are cut
// into 32-channel slices (128 pixels x 32 channels, a quarter of the work each): four times as many, four times
// shorter, spread over four times as many CUs.
template <bool BUFA, bool ST = false>
__global__ __launch_bounds__(256, 4) void conv_igemm_mixed_kernel(ConvArgs a) {
    if ((int)blockIdx.x < a.n_main)
        conv_body<2, 2, 2, 2, true, 1, CONV_BK, false, 0, BUFA, ST>(a, (int)blockIdx.x, a.n_main, 0);
    else
        conv_body<1, 4, 1, 1, true, 1, CONV_BK, false, 0, BUFA, ST>(a, (int)blockIdx.x - a.n_main, (int)gridDim.x - a.n_main,
                                                                   a.rem_tile_base);
}

// ---------------------------------------------------------------------------------------------------------------
// dgrad operand: wT[tap'][ci][co] = bf16( w[tap][co][ci] * scale[co] ), tap' = ntaps-1-tap when `flip`
// (32x32 LDS tile transpose; coalesced on both sides). src may be fp32 or bf16.
template <class T>
__global__ __launch_bounds__(256) void pack_transpose_kernel(const T* __restrict__ src, uint16_t* __restrict__ dst,
                                                             const float* __restrict__ scale, int ntaps, int Cout,
                                                             int Cin, int flip) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const T* s = src + (size_t)tap * Cout * Cin;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.0f;
        if (co < Cout && ci < Cin) {
            if constexpr (sizeof(T) == 4) v = (float)s[(size_t)co * Cin + ci];
            else v = bf16_to_f32((uint16_t)s[(size_t)co * Cin + ci]);
            if (scale) v *= scale[co];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    const int otap = flip ? ntaps - 1 - tap : tap;
    uint16_t* d = dst + (size_t)otap * Cin * Cout;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) d[(size_t)ci * Cout + co] = f32_to_bf16(tile[tx][r]);
    }
}


// The same transpose for MANY weight tensors in one launch (the backward pass re-packs all 104 convolutions after
// every optimizer step): block -> item by binary search over the first_block column of a device-resident table.
__global__ __launch_bounds__(256) void pack_transpose_batch_kernel(const cms_pack_item* __restrict__ items, int n_items,
                                                                   int src_is_f32) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n_items - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                                   // last item with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const cms_pack_item it = items[lo];
    const int Cout = it.cout, Cin = it.cin;
    const int nbx = (Cin + 31) / 32, nby = (Cout + 31) / 32;
    int r0 = b - it.first_block;
    const int bx = r0 % nbx; r0 /= nbx;
    const int by = r0 % nby;
    const int tap = r0 / nby;
    const int co0 = by * 32, ci0 = bx * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t base = (size_t)tap * Cout * Cin;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.0f;
        if (co < Cout && ci < Cin) {
            const size_t e = base + (size_t)co * Cin + ci;
            v = src_is_f32 ? ((const float*)it.src)[e] : bf16_to_f32(((const uint16_t*)it.src)[e]);
            if (it.scale) v *= it.scale[co];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    uint16_t* d = (uint16_t*)it.dst + base;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) d[(size_t)ci * Cout + co] = f32_to_bf16(tile[tx][r]);
    }
}

// (round 6) The same for tensors whose channel counts are multiples of 64 (every body convolution of the DeepLab networks + the
// head's stacked operand), bf16 -> bf16: a 64 x 64 tile per workgroup, 16-byte global accesses on both sides (8 lanes = one 128-byte
// row segment; the 32 x 32 kernel above moves 2 bytes per lane in 64-byte segments: 166 us for the 170 MB of a ResNet-101 at 1 TB/s,
// on the critical path between the losses and the backward pass once the loss kernels got shorter). Padded fp32 LDS tile, the
// classic two-way-conflict transpose. first_block counts 64 x 64 tiles here.
__global__ __launch_bounds__(256) void pack_transpose_batch64_kernel(const cms_pack_item* __restrict__ items, int n_items) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n_items - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                                   // last item with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const cms_pack_item it = items[lo];
    const int Cout = it.cout, Cin = it.cin;
    const int nbx = Cin >> 6, nby = Cout >> 6;
    int r0 = b - it.first_block;
    const int bx = r0 % nbx; r0 /= nbx;
    const int by = r0 % nby;
    const int tap = r0 / nby;
    const int co0 = by * 64, ci0 = bx * 64;
    const size_t base = (size_t)tap * Cout * Cin;
    const uint16_t* src = (const uint16_t*)it.src + base;
    uint16_t* dst = (uint16_t*)it.dst + base;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = (int)threadIdx.x + 256 * p;
        const int r = q >> 3, c8 = q & 7;               // source row (co), 8-channel chunk of the row (ci)
        const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(co0 + r) * Cin + ci0 + c8 * 8);
        const float sc = it.scale ? it.scale[co0 + r] : 1.0f;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[r][c8 * 8 + 2 * e] = __uint_as_float(w[e] << 16) * sc;
            tile[r][c8 * 8 + 2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u) * sc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int q = (int)threadIdx.x + 256 * p;
        const int ci = q >> 3, c8 = q & 7;              // destination row (ci), 8-channel chunk of the row (co)
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = (uint32_t)f32_to_bf16(tile[c8 * 8 + 2 * e][ci]) | ((uint32_t)f32_to_bf16(tile[c8 * 8 + 2 * e + 1][ci]) << 16);
        *reinterpret_cast<uint4*>(dst + (size_t)(ci0 + ci) * Cout + co0 + c8 * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
}

}  // namespace cms

using namespace cms;

static int conv_check(const cms_conv_desc* d) {
    CMS_REQUIRE(d != nullptr, "conv: null descriptor");
    CMS_REQUIRE(d->x && d->w && (d->y || d->y32), "conv: NULL tensor");
    CMS_REQUIRE(d->n > 0 && d->h > 0 && d->w_in > 0 && d->cin > 0 && d->ho > 0 && d->wo > 0 && d->cout > 0,
                "conv: bad geometry");
    CMS_REQUIRE(d->cin % CONV_BK == 0, "conv: Cin (%d) must be a multiple of %d", d->cin, CONV_BK);
    CMS_REQUIRE(d->cout % 32 == 0, "conv: Cout (%d) must be a multiple of 32 (pad the weight tensor)", d->cout);
    CMS_REQUIRE(d->ntaps > 0 && d->ntaps <= CMS_CONV_MAX_TAPS, "conv: 1..%d taps", CMS_CONV_MAX_TAPS);
    CMS_REQUIRE(d->stride >= 1 && d->out_stride >= 1, "conv: bad stride");
    CMS_REQUIRE(d->y == nullptr || d->cout_real == d->cout, "conv: bf16 NHWC output needs cout_real == cout");
    CMS_REQUIRE((size_t)d->n * d->h * d->w_in * d->cin < (1u << 31) && (size_t)d->n * d->ho * d->wo < (1u << 31) &&
                    (size_t)d->n * d->out_h * d->out_w < (1u << 31) && d->h < 0x7000 && d->w_in < 0x7000,
                "conv: too many pixels");
    return CMS_OK;
}

// Pipelined variants of the direct-to-LDS kernel (cms_conv_desc.variant 10..14): NS ring stages of BK K-elements,
// fragment double buffer, register budget declared for OCC workgroups per CU.
template <int WN, int WM, int TN, int TM, int NS, int BK, int OCC, bool BUFA = false>
static void conv_launch_ring(const ConvArgs& a, hipStream_t s) {
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32, NT = 64 * WN * WM;
    const int grid = (a.Cout / BN) * ((a.M + BM - 1) / BM) * a.ksplit;
    const size_t stage = (size_t)(BN + BM) * BK * 2 * NS, epi = (size_t)BM * BN * 2;
    const size_t lds = (stage > epi ? stage : epi) + 80 + BM * 16 + 2 * BN * 4;
    auto kern = conv_igemm_kernel<WN, WM, TN, TM, true, NS, BK, true, OCC, BUFA>;
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);
}

template <int WN, int WM, int TN, int TM, bool STATS_TILE = false>
static void conv_launch(const ConvArgs& a, hipStream_t s, int loader) {      // loader: 0 registers, 1 / 2 = glds stages,
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;                       //         3 = two glds stages of BK = 32
    const int grid = (a.Cout / BN) * ((a.M + BM - 1) / BM) * a.ksplit;
    const size_t stage = (size_t)(BN + BM) * CONV_ROW_BYTES * (loader == 2 ? 2 : 1), epi = (size_t)BM * BN * 2;
    const size_t lds = (stage > epi ? stage : epi) + 80 + BM * 16 + 2 * BN * 4 + (a.trace ? CONV_TRACE_DWORDS * 4 : 0);
    constexpr int NT = 64 * WN * WM;
    if (loader == 2) {
        // two stages of the 128 x 128 tile need 66 KB of dynamic LDS: above the 64 KB a kernel gets without asking
        static bool raised = false;
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<WN, WM, TN, TM, true, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised = true;
        }
        hipLaunchKernelGGL((conv_igemm_kernel<WN, WM, TN, TM, true, 2>), dim3(grid), dim3(NT), lds, s, a);
    } else if (loader == 3 && BN % (16 * WN * WM) == 0 && BM % (16 * WN * WM) == 0) {
        if constexpr (BN % (16 * WN * WM) == 0 && BM % (16 * WN * WM) == 0)       // 64-byte rows: 16 rows per wave load
            hipLaunchKernelGGL((conv_igemm_kernel<WN, WM, TN, TM, true, 2, 32>), dim3(grid), dim3(NT), lds, s, a);
    } else if (STATS_TILE && a.stats_out != nullptr && (loader == 4 || loader == 1)) {     // (the caller has checked the loader)
        if constexpr (STATS_TILE) {
            if (loader == 4)
                hipLaunchKernelGGL((conv_igemm_kernel<WN, WM, TN, TM, true, 1, CONV_BK, false, 0, true, true>), dim3(grid), dim3(NT), lds, s, a);
            else
                hipLaunchKernelGGL((conv_igemm_kernel<WN, WM, TN, TM, true, 1, CONV_BK, false, 0, false, true>), dim3(grid), dim3(NT), lds, s, a);
        }
    } else if (loader == 4) {                    // direct-to-LDS with buffer addressing
        hipLaunchKernelGGL((conv_igemm_kernel<WN, WM, TN, TM, true, 1, CONV_BK, false, 0, true>), dim3(grid), dim3(NT), lds, s, a);
    } else if (loader == 1 || loader == 3) {
        hipLaunchKernelGGL((conv_igemm_kernel<WN, WM, TN, TM, true, 1>), dim3(grid), dim3(NT), lds, s, a);
    } else {
        hipLaunchKernelGGL((conv_igemm_kernel<WN, WM, TN, TM, false, 1>), dim3(grid), dim3(NT), lds, s, a);
    }
}

static uint32_t* g_conv_trace = nullptr;
static int g_conv_trace_wgs = 0;

extern "C" int cms_conv_set_trace(void* buf, int workgroups) {
    g_conv_trace = (uint32_t*)buf;
    g_conv_trace_wgs = buf ? workgroups : 0;
    return 0;
}

// A/B switch for whole-step measurements (bench.py under CMS_CONV_DEFAULT_VARIANT=43 etc.): the variant used by
// descriptors that ask for 0 = auto. Diagnostic only; read once.
static int conv_default_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("CMS_CONV_DEFAULT_VARIANT");
        v = e ? atoi(e) : 0;
    }
    return v;
}

// The eight-phase 256 x 256 kernel (csrc/conv8.hip) takes the wide, K-deep layers. CMS_CONV8 (read once): 0 = never,
// 1 = one whole tile per workgroup (default: +3.5 / +5 % on the two-stream step at 321 x 321 / 512 x 1024, profiles/r04d_*),
// 2 = persistent launch with a stream-K round where the descriptor carries a workspace (slower: the 256 KB partial tiles
// move at the ~10 B/clk a CU gets from memory), CMS_CONV8_MIN_KT = fewest 64-deep K tiles per output tile it is used for
// (default 16), CMS_CONV8_GRID = workgroup cap of mode 2, CMS_CONV8_PASSES = bit 0 forward / bit 1 data-gradient launches.
static int conv8_env(int which) {
    static int mode = -1, min_kt = 0, grid = 0, passes = 3, min_tiles = 0;
    if (mode < 0) {
        const char* mt = getenv("CMS_CONV8_MIN_TILES");        // fewest 256 x 256 output tiles of a launch it is used for (round 6)
        min_tiles = mt ? atoi(mt) : 0;
        const char* e = getenv("CMS_CONV8");
        const char* k = getenv("CMS_CONV8_MIN_KT");
        const char* g = getenv("CMS_CONV8_GRID");
        const char* p = getenv("CMS_CONV8_PASSES");            // bit 0: forward epilogue launches, bit 1: data gradients
        min_kt = k ? atoi(k) : 16;
        grid = g ? atoi(g) : 0;
        passes = p ? atoi(p) : 3;
        mode = e ? atoi(e) : 1;
    }
    return which == 0 ? mode : (which == 1 ? min_kt : (which == 2 ? grid : (which == 3 ? passes : min_tiles)));
}

extern "C" long long cms_conv_igemm_workspace_bytes(void) {
    int n_cu = 0;
    if (cms_device_info(&n_cu, nullptr, 0) != CMS_OK || n_cu <= 0) n_cu = 256;
    return (long long)cms::conv8_workspace_bytes(n_cu);
}

// Would the default dispatch (variant 0, tile 0) send this launch to the eight-phase kernel?
static bool conv_takes_conv8(const cms_conv_desc* d) {
    return d->variant == 0 && d->tile == 0 && conv8_env(0) > 0 && cms::conv8_supported(d) &&
           ((d->mask_bits == nullptr && d->mask_bits_out == nullptr) || conv8_env(0) == 1) &&     // (bits: whole tiles only)
           d->ntaps * (d->cin / 64) >= conv8_env(1) && conv_default_variant() == 0 &&
           ((conv8_env(3) >> (d->mode == 0 ? 0 : 1)) & 1) &&
           (conv8_env(4) <= 0 || (long long)((d->n * d->ho * d->wo + 255) / 256) * (d->cout / 256) >= conv8_env(4));
}

static int conv_mixed_env() {                       // balanced launch; CMS_CONV_MIXED=0 switches it off (A/B, read once)
    static int env_mixed = -1;
    if (env_mixed < 0) {
        const char* e = getenv("CMS_CONV_MIXED");
        env_mixed = e ? atoi(e) : 1;
    }
    return env_mixed;
}

// Which kernel cms_conv_igemm runs a descriptor on (measurement tooling: per-kernel algorithmic bytes beside the PMC
// counters): CMS_ROUTE_CONV8 = conv8_kernel, CMS_ROUTE_MIXED = conv_igemm_mixed_kernel (balanced 128 x 128
// launch), CMS_ROUTE_TILE128 = conv_igemm_kernel<2,2,2,2>, CMS_ROUTE_TILE64 / _TILE32 = the 64- / 32-channel tiles,
// CMS_ROUTE_OTHER = an explicit variant / tile request. Negative: the descriptor is invalid.
extern "C" int cms_conv_igemm_route(const cms_conv_desc* d) {
    int rc = conv_check(d);
    if (rc) return rc;
    if (d->variant != 0 || d->tile != 0 || conv_default_variant() != 0) return CMS_ROUTE_OTHER;
    if (conv_takes_conv8(d)) return CMS_ROUTE_CONV8;
    if (d->cout % 128 == 0) {
        const bool small = (size_t)d->n * d->h * d->w_in * d->cin * 2 < (1ull << 31) &&
                           (size_t)d->ntaps * d->cout * d->cin * 2 < (1ull << 31);
        const int M = d->n * d->ho * d->wo;
        const int ntn = d->cout / 128, mtiles = (M + 127) / 128, total = mtiles * ntn;
        const int rounds = total / 256, rem = total % 256;
        if (conv_mixed_env() != 0 && d->zeros != nullptr && small && d->ksplit <= 1 && 256 % ntn == 0 && rounds >= 1 && rounds <= 9 &&
            rem > 0 && rem <= 100)
            return CMS_ROUTE_MIXED;
        return CMS_ROUTE_TILE128;
    }
    return d->cout % 64 == 0 ? CMS_ROUTE_TILE64 : CMS_ROUTE_TILE32;
}

// Pixel rows per tile of the per-tile statistics a launch of this descriptor would write to stats_out ([ceil(M / rows)][2][2][Cout]
// floats), 0 when it cannot (then cms_bn_stats reads the output back instead), negative for an invalid descriptor.
extern "C" int cms_conv_igemm_stats_tile_rows(const cms_conv_desc* d) {
    int rc = conv_check(d);
    if (rc) return rc;
    if (d->y == nullptr || d->zeros == nullptr || d->ksplit > 1 || d->out_stride != 1 || d->out_h != d->ho || d->out_w != d->wo) return 0;
    // forward launches: statistics of the output; data gradients: only with the unit's u / mean / rstd (backward statistics)
    if (d->mode == 0 ? d->bstats_u != nullptr : (d->bstats_u == nullptr || d->bstats_mean == nullptr || d->bstats_rstd == nullptr)) return 0;
    const int route = cms_conv_igemm_route(d);
    if (route < 0) return route;
    if (route == CMS_ROUTE_CONV8 && conv8_env(0) != 1) return 0;        // (whole tiles per workgroup only, not the stream-K launch)
    const int rows = route == CMS_ROUTE_CONV8 ? 256 : (route == CMS_ROUTE_OTHER ? 0 : 128);
    if (rows == 0) return 0;
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long rpg = d->stats_rows_per_group > 0 ? d->stats_rows_per_group : M;
    if (M % rpg != 0 || rpg < rows) return 0;       // a tile may straddle ONE group boundary
    return rows;
}

extern "C" int cms_conv_igemm(const cms_conv_desc* d_in, void* stream) {
    int rc = conv_check(d_in);
    if (rc) return rc;
    if (d_in->variant >= 90 && d_in->variant <= 93)     // 92 / 93: 90 / 91 with cycle stamps into the cms_conv_set_trace buffer
        return cms::conv8_launch(d_in, (hipStream_t)stream, (d_in->variant - 90) & 1, conv8_env(2),
                                 d_in->variant >= 92 ? g_conv_trace : nullptr, g_conv_trace_wgs);
    cms_conv_desc d_no8;
    if (d_in->variant == 99) {                 // the 128 x 128 family, whatever CMS_CONV8 says (reference of the conv8 tests)
        d_no8 = *d_in;
        d_no8.variant = 0;
        d_in = &d_no8;
    } else if (conv_takes_conv8(d_in))
        return cms::conv8_launch(d_in, (hipStream_t)stream, conv8_env(0) - 1, conv8_env(2), nullptr, 0);
    cms_conv_desc d_copy;
    const cms_conv_desc* d = d_in;
    if (d_in->variant == 0 && conv_default_variant() != 0) {
        const int v = conv_default_variant();
        // ring variants exist for the 128-channel tile of the bf16 output only
        const bool ring = (v >= 10 && v <= 14) || (v >= 50 && v <= 54) || (v >= 60 && v <= 85);
        const int need = (v >= 70 && v <= 85) ? 256 : 128;
        const bool small_ok = (size_t)d_in->n * d_in->h * d_in->w_in * d_in->cin * 2 < (1ull << 31) &&
                              (size_t)d_in->ntaps * d_in->cout * d_in->cin * 2 < (1ull << 31);
        if (!ring || (d_in->cout % need == 0 && (d_in->tile == 0 || d_in->tile == 128) && d_in->zeros != nullptr && small_ok &&
                      d_in->y != nullptr && (d_in->ksplit <= 1))) {
            d_copy = *d_in;
            d_copy.variant = v;
            d = &d_copy;
        }
    }
    ConvArgs a;
    a.x = (const uint16_t*)d->x; a.w = (const uint16_t*)d->w; a.y = (uint16_t*)d->y; a.y32 = d->y32;
    a.scale = d->scale; a.bias = d->bias; a.res = (const uint16_t*)d->res; a.mask_src = (const uint16_t*)d->mask_src;
    a.mask_bits_out = d->mask_bits_out; a.mask_bits = d->mask_bits;
    a.mask_gates_res = d->mask_gates_res;
    CMS_REQUIRE(d->mask_gates_res == 0 || (d->mode == 1 && d->res && d->mask_bits && d->zeros && d->y),
                "conv: mask_gates_res belongs to data-gradient launches with a residual and mask bits on the direct-to-LDS kernel");
    a.stats_out = (float*)d->stats_out;
    a.stats_rpg = d->stats_rows_per_group > 0 ? d->stats_rows_per_group : d->n * d->ho * d->wo;
    a.bstats_u = (const uint16_t*)d->bstats_u; a.bstats_bits = d->bstats_bits; a.bstats_mean = d->bstats_mean; a.bstats_rstd = d->bstats_rstd;
    CMS_REQUIRE(d->stats_out == nullptr || cms_conv_igemm_stats_tile_rows(d) == 128,
                "conv: stats_out needs a launch cms_conv_igemm_stats_tile_rows() accepts (bf16 output of a forward launch on a default "
                "tile, sample groups of whole rows no shorter than a tile)");
    CMS_REQUIRE((d->mask_bits_out == nullptr && d->mask_bits == nullptr) ||
                    (d->y != nullptr && d->zeros != nullptr && d->variant == 0 && d->ksplit <= 1),
                "conv: ReLU mask bits need the bf16 output on the default (direct-to-LDS) kernel");
    CMS_REQUIRE(d->mask_bits_out == nullptr || (d->mode == 0 && d->relu != 0), "conv: mask_bits_out is written by forward + ReLU launches");
    CMS_REQUIRE(d->mask_bits == nullptr || (d->mode == 1 && d->mask_src == nullptr), "conv: mask_bits replaces mask_src of a data gradient");
    a.N = d->n; a.H = d->h; a.W = d->w_in; a.Cin = d->cin;
    a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout; a.cout_real = d->cout_real;
    a.ntaps = d->ntaps; a.stride = d->stride;
    a.out_H = d->out_h; a.out_W = d->out_w; a.out_stride = d->out_stride;
    a.relu = d->relu; a.mode = d->mode;
    a.M = d->n * d->ho * d->wo;
    a.ksplit = d->ksplit > 1 ? d->ksplit : 1;
    CMS_REQUIRE(a.ksplit == 1 || (d->y == nullptr && d->relu == 0 && d->res == nullptr && d->mode == 0),
                "conv: ksplit needs the fp32 output without residual / ReLU (pre-zeroed, accumulated with atomics)");
    CMS_REQUIRE(a.ksplit <= d->ntaps, "conv: ksplit (%d) > taps (%d)", a.ksplit, d->ntaps);
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        a.tap_dy[i] = (short)(i < d->ntaps ? d->tap_dy[i] : 0);
        a.tap_dx[i] = (short)(i < d->ntaps ? d->tap_dx[i] : 0);
    }
    hipStream_t s = (hipStream_t)stream;
    a.zeros = (const uint16_t*)d->zeros;
    CMS_REQUIRE(d->zeros == nullptr || d->zeros_bytes >= 2 * d->cin + 128,
                "conv: the zero run (%d bytes) must be at least 2 * Cin + 128 = %d bytes long", d->zeros_bytes, 2 * d->cin + 128);
    // 20..23: the default kernel with K rotation (stride 1 / 3 / 5 / 11 K steps per pixel tile)
    // 24: the default kernel with staggered starts of co-resident workgroups; 25: stagger + rotation by 3
    a.krot = d->variant == 20 ? 1 : (d->variant == 21 ? 3 : (d->variant == 22 ? 5 : (d->variant == 23 ? 11 : (d->variant == 25 ? 3 : 0))));
    a.stagger = (d->variant == 24 || d->variant == 25) ? 1 : 0;
    a.n_main = 0; a.rem_tile_base = 0;
    // pointwise fast path of the prologue (conv_body: `plain`); CMS_CONV_PLAIN=0 switches it off (A/B, read once)
    static int env_plain = -1;
    if (env_plain < 0) {
        const char* e = getenv("CMS_CONV_PLAIN");
        env_plain = e ? atoi(e) : 1;
    }
    static int env_nt = -1;
    if (env_nt < 0) {
        const char* e = getenv("CMS_CONV_NT");          // A/B switch, read once
        env_nt = e ? atoi(e) : 0;
    }
    a.nt_store = env_nt;
    a.plain = (env_plain != 0 && d->ntaps == 1 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 && d->stride == 1 && d->h == d->ho &&
               d->w_in == d->wo && d->out_stride == 1 && d->out_h == d->ho && d->out_w == d->wo && a.ksplit == 1 && a.krot == 0 &&
               (size_t)a.M * d->cin * 2 < (1ull << 31)) ? 1 : 0;
    // 30: the default kernel with per-workgroup cycle stamps into the buffer given to cms_conv_set_trace
    a.trace = (d->variant == 30 || d->variant == 41) ? g_conv_trace : nullptr;      // 41: trace of variant 40
    a.trace_wgs = g_conv_trace_wgs;
    a.dbg = (d->variant == 2 || d->variant == 3) ? d->variant : (d->variant == 6 ? 2 : (d->variant == 7 ? 3 : 0));
    // variant 0: direct-to-LDS, one stage, up to 4 workgroups per CU (default); 1: register-staged loader;
    // 4: direct-to-LDS, two stages, 2 workgroups per CU -- measured 10 % faster on grids of exactly <= 2 workgroups per
    // CU, 20 % slower on everything else (tools/tail_probe.py): co-resident workgroups hide more than the second stage;
    // 2 / 3: ablation switches of the default kernel (no MFMA / no loads after the first stage)
    // 5: direct-to-LDS, two stages of 32 K-elements each (same LDS footprint and occupancy as the default)
    // 6 / 7: the ablation switches applied to the two-stage kernel (variant 4)
    // default (and 40 / 41): direct-to-LDS with buffer addressing when both tensors are below 2 GB; 43 forces the flat
    // 64-bit addresses of round 1 (A/B: profiles/r02o_*)
    const bool small = (size_t)d->n * d->h * d->w_in * d->cin * 2 < (1ull << 31) &&
                       (size_t)d->ntaps * d->cout * d->cin * 2 < (1ull << 31);
    const int glds = (d->zeros == nullptr || d->variant == 1) ? 0
                     : ((d->variant == 4 || d->variant == 6 || d->variant == 7) ? 2
                        : (d->variant == 5 ? 3 : ((small && d->variant != 43) ? 4 : 1)));      // 43: flat addresses (round-1 loader)
    const int tile = d->tile;   // 0 = auto
    CMS_REQUIRE(d->stats_out == nullptr || glds == 4 || glds == 1, "conv: stats_out needs the direct-to-LDS kernels (a zero run in the descriptor)");
    if (d->variant >= 80 && d->variant <= 85) {
        // Round 3 experiment: 256 (co) x 128 (pixels) on FOUR waves (each 128 co x 64 pixels): 48 KB staged and 96 KB of fragment
        // reads per 2x the MFMA work of the default tile, one workgroup per CU for layers of ~263 pixel tiles
        CMS_REQUIRE(d->zeros != nullptr && small && d->cout % 256 == 0, "conv: variants 80..85 need the zero run, tensors below 2 GB, Cout %% 256 == 0");
        switch (d->variant) {
        case 80: conv_launch_ring<2, 2, 4, 2, 1, 64, 1, true>(a, s); break;
        case 81: conv_launch_ring<2, 2, 4, 2, 2, 64, 1, true>(a, s); break;
        case 82: conv_launch_ring<2, 2, 4, 2, 3, 64, 1, true>(a, s); break;
        case 83: conv_launch_ring<2, 2, 4, 2, 3, 32, 1, true>(a, s); break;
        case 84: conv_launch_ring<2, 2, 4, 2, 4, 32, 1, true>(a, s); break;
        default: conv_launch_ring<2, 2, 4, 2, 2, 64, 2, true>(a, s); break;
        }
        return launch_status("cms_conv_igemm");
    }
    if (d->variant >= 60 && d->variant <= 75) {
        // Round 3 experiment: the stage rings on WIDE tiles with buffer-addressed asm loads (the wide rings 10..14 of round 2
        // used the builtin loads the compiler drains): 60..63 = 128 (co) x 256 (pixels) on 8 waves; 70..73 = 256 x 256 on 8
        // waves (each wave 128 co x 64 pixels: 6 fragment reads per 8 MFMAs, 64 KB staged per 4x the work of the default tile)
        CMS_REQUIRE(d->zeros != nullptr && small, "conv: variants 60..75 need the zero run and tensors below 2 GB");
        if (d->variant < 70) {
            CMS_REQUIRE(d->cout % 128 == 0, "conv: variants 60..63 need Cout %% 128 == 0");
            switch (d->variant) {
            case 60: conv_launch_ring<2, 4, 2, 2, 1, 64, 2, true>(a, s); break;
            case 61: conv_launch_ring<2, 4, 2, 2, 2, 64, 1, true>(a, s); break;
            case 62: conv_launch_ring<2, 4, 2, 2, 3, 32, 2, true>(a, s); break;
            default: conv_launch_ring<2, 4, 2, 2, 4, 32, 1, true>(a, s); break;
            }
        } else {
            CMS_REQUIRE(d->cout % 256 == 0, "conv: variants 70..73 need Cout %% 256 == 0");
            switch (d->variant) {
            case 70: conv_launch_ring<2, 4, 4, 2, 1, 64, 1, true>(a, s); break;
            case 71: conv_launch_ring<2, 4, 4, 2, 2, 64, 1, true>(a, s); break;
            case 72: conv_launch_ring<2, 4, 4, 2, 3, 32, 1, true>(a, s); break;
            default: conv_launch_ring<2, 4, 4, 2, 4, 32, 1, true>(a, s); break;
            }
        }
        return launch_status("cms_conv_igemm");
    }
    if (d->variant >= 50 && d->variant <= 54) {
        // 50..54: the stage rings 10..14 of the 128 x 128 tile with buffer addressing
        CMS_REQUIRE(d->zeros != nullptr && small && d->cout % 128 == 0 && (tile == 0 || tile == 128),
                    "conv: variants 50..54 need the zero run, tensors below 2 GB and the 128-channel tile");
        switch (d->variant) {
        case 50: conv_launch_ring<2, 2, 2, 2, 1, 64, 4, true>(a, s); break;
        case 51: conv_launch_ring<2, 2, 2, 2, 2, 64, 2, true>(a, s); break;
        case 52: conv_launch_ring<2, 2, 2, 2, 3, 32, 3, true>(a, s); break;
        case 53: conv_launch_ring<2, 2, 2, 2, 4, 32, 2, true>(a, s); break;
        default: conv_launch_ring<2, 2, 2, 2, 3, 64, 1, true>(a, s); break;
        }
        return launch_status("cms_conv_igemm");
    }
    if (d->variant >= 10 && d->variant <= 14) {
        // pipelined kernels (ring of LDS stages with counted vmcnt, fragment double buffer), 128 x 128 tile on 4 waves or
        // 128 (co) x 256 (pixels) on 8 waves:  10: 1 stage of 64   11: 2 x 64   12: 3 x 32   13: 4 x 32   14: 3 x 64
        CMS_REQUIRE(d->zeros != nullptr, "conv: variants 10..14 need the zero run (direct-to-LDS loader)");
        CMS_REQUIRE(d->cout % 128 == 0 && (tile == 0 || tile == 128 || tile == 256),
                    "conv: variants 10..14 exist for the 128-channel tiles (tile 0 / 128 / 256)");
        if (tile == 256) {
            switch (d->variant) {
            case 10: conv_launch_ring<2, 4, 2, 2, 1, 64, 2>(a, s); break;
            case 11: conv_launch_ring<2, 4, 2, 2, 2, 64, 1>(a, s); break;
            case 12: conv_launch_ring<2, 4, 2, 2, 3, 32, 2>(a, s); break;
            case 13: conv_launch_ring<2, 4, 2, 2, 4, 32, 1>(a, s); break;
            default: conv_launch_ring<2, 4, 2, 2, 3, 64, 1>(a, s); break;
            }
        } else {
            switch (d->variant) {
            case 10: conv_launch_ring<2, 2, 2, 2, 1, 64, 4>(a, s); break;
            case 11: conv_launch_ring<2, 2, 2, 2, 2, 64, 2>(a, s); break;
            case 12: conv_launch_ring<2, 2, 2, 2, 3, 32, 3>(a, s); break;
            case 13: conv_launch_ring<2, 2, 2, 2, 4, 32, 2>(a, s); break;
            default: conv_launch_ring<2, 2, 2, 2, 3, 64, 1>(a, s); break;
            }
        }
        return launch_status("cms_conv_igemm");
    }
    if (tile == 256) {                         // 8 waves: 128 co x 256 pixels (more reuse of the weight tile)
        CMS_REQUIRE(d->cout % 128 == 0, "conv: tile 256 needs Cout %% 128 == 0");
        conv_launch<2, 4, 2, 2>(a, s, glds);
    } else if (tile == 2256) {                 // 4 waves, each 64 co x 128 pixels: 6 fragment reads per 8 MFMAs
        CMS_REQUIRE(d->cout % 128 == 0, "conv: tile 2256 needs Cout %% 128 == 0");
        conv_launch<2, 2, 2, 4>(a, s, glds);
    } else if (tile == 1128) {                 // 8 waves on the 128 x 128 tile (each wave 64 co x 32 pixels)
        CMS_REQUIRE(d->cout % 128 == 0, "conv: tile 1128 needs Cout %% 128 == 0");
        conv_launch<2, 4, 2, 1>(a, s, glds);
    } else if ((tile == 0 && d->cout % 128 == 0) || tile == 128) {
        // default: 128 co x 128 pixels, 4 waves of 64 x 64 (the 8-wave layouts above measured within +-5 % of it on
        // the DeepLab v2 layer shapes and no better end to end, tools/conv_ablate.py)
        CMS_REQUIRE(d->cout % 128 == 0, "conv: tile 128 needs Cout %% 128 == 0");
        // balanced launch (conv_igemm_mixed_kernel): when the 128 x 128 grid is a few workgroups more than a multiple of
        // the 256 CUs, those few are cut into 32-channel slices. CMS_CONV_MIXED=0 switches it off (A/B).
        const int env_mixed = conv_mixed_env();
        const int ntn = d->cout / 128, mtiles = (a.M + 127) / 128, total = mtiles * ntn;
        const int rounds = total / 256, rem = total % 256;
        if (env_mixed != 0 && tile == 0 && d->variant == 0 && (glds == 4 || glds == 1) && a.ksplit == 1 && 256 % ntn == 0 &&
            rounds >= 1 && rounds <= 9 && rem > 0 && rem <= 100) {
            a.n_main = rounds * 256;
            a.rem_tile_base = a.n_main / ntn;
            const int n_rem = (mtiles - a.rem_tile_base) * (d->cout / 32);
            const size_t lds = 32768 + 80 + 128 * 16 + 2 * 128 * 4 + (a.trace ? CONV_TRACE_DWORDS * 4 : 0);
            if (a.stats_out) {
                if (glds == 4) hipLaunchKernelGGL((conv_igemm_mixed_kernel<true, true>), dim3(a.n_main + n_rem), dim3(256), lds, s, a);
                else hipLaunchKernelGGL((conv_igemm_mixed_kernel<false, true>), dim3(a.n_main + n_rem), dim3(256), lds, s, a);
            } else if (glds == 4) hipLaunchKernelGGL(conv_igemm_mixed_kernel<true>, dim3(a.n_main + n_rem), dim3(256), lds, s, a);
            else hipLaunchKernelGGL(conv_igemm_mixed_kernel<false>, dim3(a.n_main + n_rem), dim3(256), lds, s, a);
            return launch_status("cms_conv_igemm");
        }
        conv_launch<2, 2, 2, 2, true>(a, s, glds);
    } else if ((tile == 0 && d->cout % 64 == 0) || tile == 64) {
        CMS_REQUIRE(d->cout % 64 == 0, "conv: tile 64 needs Cout %% 64 == 0");
        conv_launch<1, 4, 2, 1, true>(a, s, glds);     // 64 co x 128 pixels
    } else {
        conv_launch<1, 4, 1, 1, true>(a, s, glds);     // 32 co x 128 pixels
    }
    return launch_status("cms_conv_igemm");
}

extern "C" int cms_conv_pack_transpose(const void* src, int src_dtype, void* dst_bf16, const float* scale, int ntaps,
                                       int cout, int cin, int flip, void* stream) {
    CMS_REQUIRE(src && dst_bf16, "conv_pack_transpose: NULL pointer");
    CMS_REQUIRE(ntaps > 0 && cout > 0 && cin > 0, "conv_pack_transpose: bad geometry");
    CMS_REQUIRE(src_dtype == CMS_F32 || src_dtype == CMS_BF16, "conv_pack_transpose: bad dtype");
    dim3 grid((cin + 31) / 32, (cout + 31) / 32, ntaps);
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == CMS_F32)
        hipLaunchKernelGGL(pack_transpose_kernel<float>, grid, dim3(256), 0, s, (const float*)src, (uint16_t*)dst_bf16,
                           scale, ntaps, cout, cin, flip);
    else
        hipLaunchKernelGGL(pack_transpose_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)src,
                           (uint16_t*)dst_bf16, scale, ntaps, cout, cin, flip);
    return launch_status("cms_conv_pack_transpose");
}

extern "C" int cms_conv_pack_transpose_batch(const cms_pack_item* items_dev, int n_items, int total_blocks,
                                             int src_dtype, void* stream) {
    CMS_REQUIRE(items_dev && n_items > 0 && total_blocks > 0, "conv_pack_transpose_batch: empty table");
    CMS_REQUIRE(src_dtype == CMS_F32 || src_dtype == CMS_BF16, "conv_pack_transpose_batch: bad dtype");
    hipLaunchKernelGGL(pack_transpose_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev,
                       n_items, src_dtype == CMS_F32 ? 1 : 0);
    return launch_status("cms_conv_pack_transpose_batch");
}

extern "C" int cms_conv_pack_transpose_batch64(const cms_pack_item* items_dev, int n_items, int total_blocks, void* stream) {
    CMS_REQUIRE(items_dev && n_items > 0 && total_blocks > 0, "conv_pack_transpose_batch64: empty table");
    hipLaunchKernelGGL(pack_transpose_batch64_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n_items);
    return launch_status("cms_conv_pack_transpose_batch64");
}

// =================================================================================================================
// Weight gradient:  dW[tap][co][ci] += scale[co] * sum_pixels dU[pix][co] * X[pix shifted by tap][ci]
//
// GEMM with K = pixels: both operands are stored pixel-major (NHWC), i.e. K-major, while the MFMA wants each lane to
// hold 8 consecutive k for one row. gfx950's LDS transpose read does that conversion: ds_read_b64_tr_b16 over a
// 16-lane group turns a [4 pixels][16 channels] block (source lane s supplies the 8-byte chunk
// pixel = s>>2, channels 4*(s&3)..+3) into "lane i holds channel i for the 4 pixels" (probed on hardware:
// result(lane i, j) = chunk[4j + (i>>2)][i&3], tools/tr_probe.hip). Two such reads give one MFMA operand fragment.
//
// LDS image: [64 pixels][128 channels] bf16 per operand (256-byte rows, 64-byte slots XOR-swizzled with pixel&3 so the
// four pixel rows of a half-wave's read land in different bank windows). Workgroup = 4 waves, 128 (co) x 128 (ci)
// outputs of one tap over one slice of the pixel axis (split-K); results are accumulated into the fp32 gradient arena
// with atomics (the arena is zeroed once per step, so both backward passes of an iteration simply add up).
// MFMA-bound for Cin*Cout >= 256*256 (AI = 2*64*128*128 / 32 KB per stage = 64 FLOP/B from L2/HBM per workgroup stage).
namespace cms {

struct WgradArgs {
    const uint16_t* du;    // bf16 [N][Ho][Wo][Cout]
    const uint16_t* x;     // bf16 [N][H][W][Cin]
    float* dw;             // fp32 [ntaps][Cout][Cin]
    const float* scale;    // [Cout] or NULL
    int N, H, W, Cin, Ho, Wo, Cout;
    int ntaps, stride;
    int M;                 // N*Ho*Wo
    int ksplit, pix_per_split;   // pixel slice per workgroup (multiple of 64)
    int cout_real;         // rows >= cout_real are not written (padded class axis)
    int dw_cout;           // rows per tap of the dw tensor
    const uint16_t* w;     // bf16 [ntaps][Cout][Cin] or NULL   } side outputs for a trainable BN affine:
    float* wdot;           // [Cout] += <W, G> per output channel } see cms_wgrad_desc
    float* dbeta;          // [Cout] += sum_p dU[p][co]
    uint32_t* trace;       // diagnostic: cycle stamps (cms_conv_set_trace), NULL in production
    int trace_wgs;
    float* slab;           // split-K partial sums [ksplit][ntaps][Cout][Cin] (plain stores) or NULL (atomics into dw)
    size_t slab_stride;    // floats per slice
    short tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
};

// dw[tap][co][ci] += sum over slices of slab[s][tap][co][ci], co < cout_real. 64 float4 outputs per block, the slices dealt
// over 4 thread groups with 4 independent loads in flight each, combined through LDS in a FIXED order (round 3: one thread
// per output walking all slices serially made this launch 35 us for the 64 KB gradients of layer1, whose 2-9 tiles are cut
// into ~190 slices -- the whole in-step cost of the deterministic mode, profiles/r03s_*).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int ksplit,
                                                            size_t slab_stride, int ntaps, int Cout, int Cin, int cout_real,
                                                            int dw_cout) {
    __shared__ float4 part[3][64];
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t q = (size_t)blockIdx.x * 64 + j;                      // float4 index over [ntaps][cout_real][Cin / 4]
    const int c4 = Cin >> 2;
    const size_t total = (size_t)ntaps * cout_real * c4;
    const bool valid = q < total;
    float4 acc = float4{0.f, 0.f, 0.f, 0.f};
    int ci = 0, co = 0, tap = 0;
    if (valid) {
        ci = (int)(q % c4) * 4;
        const size_t r = q / c4;
        co = (int)(r % cout_real);
        tap = (int)(r / cout_real);
        const float* src = slab + ((size_t)tap * Cout + co) * Cin + ci;
        const float4 zero = float4{0.f, 0.f, 0.f, 0.f};
        for (int s = g; s < ksplit; s += 16) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)s * slab_stride);
            const float4 v1 = s + 4 < ksplit ? *reinterpret_cast<const float4*>(src + (size_t)(s + 4) * slab_stride) : zero;
            const float4 v2 = s + 8 < ksplit ? *reinterpret_cast<const float4*>(src + (size_t)(s + 8) * slab_stride) : zero;
            const float4 v3 = s + 12 < ksplit ? *reinterpret_cast<const float4*>(src + (size_t)(s + 12) * slab_stride) : zero;
            acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
            acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
    }
    if (g > 0) part[g - 1][j] = acc;
    __syncthreads();
    if (g == 0 && valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 v = part[k][j];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        float4* dst = reinterpret_cast<float4*>(dw + ((size_t)tap * dw_cout + co) * Cin + ci);
        float4 o = *dst;
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *dst = o;
    }
}

}  // namespace cms (reopened below)
void cms::wgrad_reduce_launch(const float* slab, float* dw, int ksplit, size_t slice_elems, int ntaps, int cout, int cin, int cout_real,
                              int dw_cout, hipStream_t s) {
    const size_t total4 = (size_t)ntaps * cout_real * (cin / 4);
    hipLaunchKernelGGL(cms::wgrad_reduce_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, s, slab, dw, ksplit, slice_elems, ntaps,
                       cout, cin, cout_real, dw_cout);
}
namespace cms {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wg_off(int pix, int ch) {     // byte offset of channel `ch` (multiple of 4) of row `pix`
    const int slot = (ch >> 5) ^ (pix & 3);
    return (uint32_t)pix * 256u + (uint32_t)slot * 64u + (uint32_t)(ch & 31) * 2u;
}

// PLAIN = 1x1, stride 1, no tap offset: the input pixel IS the output pixel, so the loader needs no (n, y, x) cursor and
// no bounds test -- two thirds of the network's weight-gradient launches.
// BETA = also accumulate sum_p dU[p][co] (cms_wgrad_desc.dbeta). A compile-time switch: as a run-time one its extra
// accumulators sat in the main loop of EVERY launch (32 v_accvgpr moves per stage, +3-5 % on the DeepLab v2 step).
// DMA = both operands go global -> LDS with buffer-addressed direct-to-LDS loads (the swizzle applied on the source
// side), no staging registers and no ds_write; one LDS stage, up to 4 workgroups per CU overlap each other's phases.
// PLAIN launches then advance a SCALAR offset per stage (no vector ALU in the loader at all); rows past the slice are
// out-of-range offsets (hardware zero fill). tools/wgrad_trace.py: the register loader spent 1450..1900 of a
// 3700..4100-cycle stage issuing its 8 loads (64-bit address arithmetic) and 750 waiting for them + ds_write.
// (the body is a device function of the LOGICAL block id -- XCD-remapped by the caller -- so that a grouped launch can run the
// weight gradients of many layers in one grid: conv_wgrad_group_kernel below)
template <int TCO, int TCI, bool PLAIN, bool BETA, bool DMA = false, int DNS = 1>   // 32x32 MFMA tiles per wave along co / ci; waves are 2 x 2
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, int b) {
    constexpr int BCO = 2 * TCO * 32, BCI = 2 * TCI * 32;       // <= 128 each
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WG_STAGE = 2 * 64 * 256;        // one stage: dU tile + X tile (32 KB)
    constexpr int WG_NS = DMA ? DNS : 1;          // DMA, DNS = 2: the loads of stage k+1 fly during the MFMAs of stage k
    static_assert(DNS == 1 || DNS == 2, "one or two stages");
    unsigned char* lds_u = smem;                  // [64][256 B]  dU tile (only the first BCO channels are used)
    unsigned char* lds_x = smem + 64 * 256;       // [64][256 B]  X tile
    float* lds_scale = reinterpret_cast<float*>(smem + WG_NS * WG_STAGE);   // [BCO] per-row factor of the epilogue
    uint32_t* lds_trace = reinterpret_cast<uint32_t*>(smem + WG_NS * WG_STAGE + 128 * 4);   // [CONV_TRACE_DWORDS] when tracing
    const bool tracing = a.trace != nullptr && (int)blockIdx.x < a.trace_wgs;
    const uint64_t t_start = tracing ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t rt_start = tracing ? __builtin_amdgcn_s_memrealtime() : 0;
    auto stamp = [&](int slot) {
        if (tracing) {
            const uint32_t t = (uint32_t)(__builtin_amdgcn_s_memtime() - t_start);
            if (threadIdx.x == 0 && slot < CONV_TRACE_DWORDS) lds_trace[slot] = t;
        }
    };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave & 1, wci = wave >> 1;
    const int nco = a.Cout / BCO, nci = a.Cin / BCI;
    const int tco = b % nco; b /= nco;
    const int tci = b % nci; b /= nci;
    const int tap = b % a.ntaps; b /= a.ntaps;
    const int ks = b;
    const int co0 = tco * BCO, ci0 = tci * BCI;
    int dy = 0, dx = 0;
#pragma unroll
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {       // constant-index scan (no dynamic indexing of kernel arguments)
        if (i == tap) { dy = a.tap_dy[i]; dx = a.tap_dx[i]; }
    }
    const int p_begin = ks * a.pix_per_split;
    const int p_end = min(a.M, p_begin + a.pix_per_split);
    if (p_begin >= p_end) return;      // empty slice (uniform for the whole workgroup)
    // the epilogue's per-row factor, fetched up front (a global load per accumulator row in front of the atomics
    // serialised 16 * TCO memory round trips; made visible by the first barrier of the pixel loop)
    if (tid < BCO) lds_scale[tid] = a.scale ? a.scale[co0 + tid] : 1.0f;

    // loader: 64 pixels x 16 chunks of 16 B per operand; thread -> chunk (tid & 15), pixel rows (tid >> 4) + 16*i
    const int c16 = tid & 15, prow = tid >> 4;
    const bool load_u = c16 * 8 < BCO, load_x = c16 * 8 < BCI;
    // pixel cursor of the 4 rows this thread loads: decoded once (2 divisions), then advanced by 64 pixels per stage
    int cm[4], cn[4], cy[4], cx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = p_begin + prow + 16 * i;
        cm[i] = m;
        if constexpr (!PLAIN) {
            cx[i] = m % a.Wo;
            const int t = m / a.Wo;
            cy[i] = t % a.Ho;
            cn[i] = t / a.Ho;
        }
    }
    const uint32_t uoff = (uint32_t)(co0 + c16 * 8), xoff = (uint32_t)(ci0 + c16 * 8);
    u32x4 ru[4], rxx[4];
    auto load_tile = [&]() {           // loads the stage the cursors point at
        if constexpr (PLAIN) {
            if (cm[0] - prow + 64 <= p_end) {      // whole 64-pixel stage inside the slice (uniform): straight loads
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (load_u) ru[i] = *reinterpret_cast<const u32x4*>(a.du + (size_t)((uint32_t)(cm[i] * a.Cout) + uoff));
                    if (load_x) rxx[i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)((uint32_t)(cm[i] * a.Cin) + xoff));
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ru[i] = u32x4{0u, 0u, 0u, 0u};
            rxx[i] = u32x4{0u, 0u, 0u, 0u};
            if (cm[i] < p_end) {
                if (load_u) ru[i] = *reinterpret_cast<const u32x4*>(a.du + (size_t)((uint32_t)(cm[i] * a.Cout) + uoff));
                if constexpr (PLAIN) {
                    if (load_x) rxx[i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)((uint32_t)(cm[i] * a.Cin) + xoff));
                } else {
                    const uint32_t iy = (uint32_t)(cy[i] * a.stride + dy), ix = (uint32_t)(cx[i] * a.stride + dx);
                    if (load_x && iy < (uint32_t)a.H && ix < (uint32_t)a.W)
                        rxx[i] = *reinterpret_cast<const u32x4*>(
                            a.x + (size_t)((uint32_t)(((cn[i] * a.H + (int)iy) * a.W + (int)ix) * a.Cin) + xoff));
                }
            }
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cm[i] += 64;
            if constexpr (!PLAIN) {
                cx[i] += 64;
                while (cx[i] >= a.Wo) { cx[i] -= a.Wo; cy[i] += 1; }
                while (cy[i] >= a.Ho) { cy[i] -= a.Ho; cn[i] += 1; }
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = prow + 16 * i;
            *reinterpret_cast<u32x4*>(lds_u + wg_off(p, c16 * 8)) = ru[i];
            *reinterpret_cast<u32x4*>(lds_x + wg_off(p, c16 * 8)) = rxx[i];
        }
    };

    // ---- DMA loader: wave instruction (pass i, wave w) fills the 4 pixel rows of piece 4i + w (1 KB, lane-linear):
    // lane -> row lane >> 4, PHYSICAL 16-byte chunk lane & 15, and fetches the LOGICAL chunk the swizzle puts there
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int drow = lane >> 4, dcp = lane & 15;
    const int dchl = ((((dcp >> 2) ^ drow) & 3) << 5) + ((dcp & 3) << 3);       // logical channel of this lane's 16 bytes
    constexpr uint32_t OOB = 0x80000000u;                                       // beyond any (< 2 GB) tensor: zero fill
    uint32_t uvoff[DMA ? 4 : 1], xvoff[DMA ? 4 : 1];
    uint32_t soff_u = 0, soff_x = 0;                                            // scalar: stage * 64 pixels (PLAIN: both)
    const buf_rsrc_t rsrc_u = make_buf_rsrc(a.du, DMA ? a.M * a.Cout * 2 : 0);
    const buf_rsrc_t rsrc_x = make_buf_rsrc(a.x, DMA ? a.N * a.H * a.W * a.Cin * 2 : 0);
    int dn[(DMA && !PLAIN) ? 4 : 1], dyy[(DMA && !PLAIN) ? 4 : 1], dxx[(DMA && !PLAIN) ? 4 : 1];   // (n, oy, ox) cursors
    auto dma_x_offsets = [&](int p0) {             // !PLAIN: X offsets of the stage starting at pixel p0 from the cursors
        if constexpr (DMA && !PLAIN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = p0 + 4 * (4 * i + wave) + drow;
                const uint32_t iy = (uint32_t)(dyy[i] * a.stride + dy), ix = (uint32_t)(dxx[i] * a.stride + dx);
                const bool ok = m < p_end && dchl < BCI && iy < (uint32_t)a.H && ix < (uint32_t)a.W;
                xvoff[i] = ok ? (uint32_t)((((dn[i] * a.H + (int)iy) * a.W + (int)ix) * a.Cin + ci0 + dchl) * 2) : OOB;
            }
        }
    };
    auto dma_tail_mask = [&](int p0) {             // last, partial stage of the slice: rows >= p_end read zeros
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = p0 + 4 * (4 * i + wave) + drow;
                if (m >= p_end) {
                    uvoff[i] = OOB;
                    xvoff[i] = OOB;
                }
            }
        }
    };
    auto dma_issue = [&](int buf) {
        if constexpr (DMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                buf_load_lds16(rsrc_u, lds_u + buf * WG_STAGE + (4 * i + wave_s) * 1024, uvoff[i], soff_u);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                buf_load_lds16(rsrc_x, lds_x + buf * WG_STAGE + (4 * i + wave_s) * 1024, xvoff[i], soff_x);
        }
    };
    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = p_begin + 4 * (4 * i + wave) + drow;
            uvoff[i] = dchl < BCO ? (uint32_t)((m * a.Cout + co0 + dchl) * 2) : OOB;
            if constexpr (PLAIN) {
                xvoff[i] = dchl < BCI ? (uint32_t)((m * a.Cin + ci0 + dchl) * 2) : OOB;
            } else {
                dxx[i] = m % a.Wo;
                const int t = m / a.Wo;
                dyy[i] = t % a.Ho;
                dn[i] = t / a.Ho;
            }
        }
        dma_x_offsets(p_begin);
        if (p_begin + 64 > p_end) dma_tail_mask(p_begin);
    }

    f32x16 acc[TCO][TCI];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // sum_p dU[p][co] as one more GEMM column: the dU fragments are multiplied with an all-ones operand (bf16 1.0) in
    // the workgroups of ci-tile 0 / tap 0 (every pixel slice once), waves wci == 0; all 32 columns come out equal
    const bool do_beta = BETA && tci == 0 && tap == 0 && wci == 0;
    f32x16 accb[BETA ? TCO : 1];
#pragma unroll
    for (int i = 0; i < (BETA ? TCO : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.0f;
    const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    // transpose-read geometry of this lane: 16-lane group g, lane-in-group li
    const int g = lane >> 4, li = lane & 15;
    const int ch_in_tile = 16 * (g & 1) + 4 * (li & 3);     // channel chunk this lane SUPPLIES (within a 32-wide tile)
    const int pix_in_blk = 8 * (g >> 1) + (li >> 2);        // pixel row this lane supplies (within a 16-pixel k-step)

    auto mfma_phase = [&](int buf) {
        unsigned char* su = smem + buf * WG_STAGE;                      // this stage's dU / X tiles
        unsigned char* sx = smem + buf * WG_STAGE + 64 * 256;
        auto frag_read = [&](int kk, u32x4* fu, u32x4* fx) {
            const int pr0 = kk * 16 + pix_in_blk;           // first read: pixels +0..3 of this lane group's 8
#pragma unroll
            for (int i = 0; i < TCO; ++i) {
                const int ch = (wco * TCO + i) * 32 + ch_in_tile;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(su + wg_off(pr0, ch)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(su + wg_off(pr0 + 4, ch)));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fu[i] = u32x4{l2.x, l2.y, h2.x, h2.y};
            }
#pragma unroll
            for (int j = 0; j < TCI; ++j) {
                const int ch = (wci * TCI + j) * 32 + ch_in_tile;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sx + wg_off(pr0, ch)));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(sx + wg_off(pr0 + 4, ch)));
                uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                fx[j] = u32x4{l2.x, l2.y, h2.x, h2.y};
            }
        };
        // register double buffer over the four 16-pixel sub-steps: the transpose reads of sub-step kk+1 are in flight
        // during the MFMAs of kk (a wgrad grid has 1..2 waves per SIMD: nobody else hides the LDS latency)
        u32x4 fu[2][TCO], fx[2][TCI];
        frag_read(0, fu[0], fx[0]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {                    // 16 pixels per MFMA
            if (kk + 1 < 4) frag_read(kk + 1, fu[(kk + 1) & 1], fx[(kk + 1) & 1]);
            // (without the fences hipcc sinks the reads of kk+1 below the MFMAs of kk and waits lgkmcnt(0) in front
            // of every MFMA pair: the phase then takes ~1100 cycles instead of ~600)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TCI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fu[kk & 1][i]),
                                                                        __builtin_bit_cast(bf16x8, fx[kk & 1][j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BETA) {
                if (do_beta) {                              // wave-uniform
#pragma unroll
                    for (int i = 0; i < TCO; ++i)
                        accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fu[kk & 1][i]),
                                                                          __builtin_bit_cast(bf16x8, ones), accb[i], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (DMA) {
        dma_issue(0);
        stamp(4);                                           // prologue done (first stage's loads issued)
        int buf = 0;
        auto next_stage = [&](int p0, int nbuf) {
            if (p0 + 64 < p_end) {
                soff_u += 64 * a.Cout * 2;
                if constexpr (PLAIN) {
                    soff_x += 64 * a.Cin * 2;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        dxx[i] += 64;
                        while (dxx[i] >= a.Wo) { dxx[i] -= a.Wo; dyy[i] += 1; }
                        while (dyy[i] >= a.Ho) { dyy[i] -= a.Ho; dn[i] += 1; }
                    }
                    dma_x_offsets(p0 + 64);
                }
                if (p0 + 128 > p_end) dma_tail_mask(p0 + 64);
                dma_issue(nbuf);
            }
        };
        for (int p0 = p_begin; p0 < p_end; p0 += 64) {
            const int tb = 16 + ((p0 - p_begin) >> 6) * 6;
            stamp(tb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of stage `buf` have landed in LDS
            stamp(tb + 1);
            __syncthreads();                                    // ... everybody else's too (two stages: and all fragment
            stamp(tb + 2);                                      // reads of the previous stage, the other buffer, are done)
            if constexpr (WG_NS == 2) next_stage(p0, buf ^ 1);  // in flight during the MFMAs below
            stamp(tb + 3);
            mfma_phase(buf);
            stamp(tb + 4);
            if constexpr (WG_NS == 1) {
                __syncthreads();                                // all fragment reads done: the stage may be refilled
                next_stage(p0, 0);
            } else {
                buf ^= 1;
            }
            stamp(tb + 5);
        }
    } else {
        load_tile();
        stamp(4);                                           // prologue done (first stage's loads issued)
        for (int p0 = p_begin; p0 < p_end; p0 += 64) {
            const int tb = 16 + ((p0 - p_begin) >> 6) * 6;
            stamp(tb);
            __syncthreads();
            stamp(tb + 1);
            store_tile();                                   // (waits for the stage's global loads)
            stamp(tb + 2);
            __syncthreads();
            stamp(tb + 3);
            advance();
            load_tile();                                    // next stage (all-zero past the end of the slice)
            stamp(tb + 4);
            mfma_phase(0);
            stamp(tb + 5);
        }
    }
    stamp(5);                                               // pixel loop done

    // epilogue. The side outputs (trainable-BN launches only) come straight from the accumulator registers. The
    // gradient itself goes through LDS, every wave through its OWN 8 KB (32 rows x TCI*32 columns, no barrier inside):
    // the split-K workgroups of one output tile finish at about the same time and would all start their atomics at
    // row 0 of the same cache lines -- the memory-side atomic units serialise same-address updates (tools/wgrad_trace.py:
    // 13.6k / 21.7k / 34k cycles of epilogue at 11 / 24 / 88 slices for the same 25 MB of atomics). From LDS the rows can
    // be walked from a slice-dependent starting row, and one wave instruction covers 64 consecutive floats of a row.
    float* dwt = a.dw + (size_t)tap * a.dw_cout * a.Cin;
    const int fcol = lane & 31, fhalf = lane >> 5;
    if (a.wdot || (BETA && do_beta)) {
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            if (a.wdot) {
                // <W, G> per output channel: first ALL products of this 32-row block (32 independent 2-byte loads in
                // flight, one wait), then the sum over the 32 lanes of a row with DPP adds (row_shr 1, 2, 4, 8 +
                // row_bcast 15: 5 vector-ALU instructions, result in lane 31 of each half-wave). The first version
                // paid a load round trip + 5 ds_bpermute round trips PER ROW: ~40 000 cycles per workgroup, more than
                // the K loop of a DeepLab v3+ launch (10 890 pixels).
                float dots[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wco * TCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    const int co_c = min(co, a.Cout - 1);
                    float dot = 0.0f;
#pragma unroll
                    for (int j = 0; j < TCI; ++j) {
                        const int ci = ci0 + (wci * TCI + j) * 32 + fcol;
                        dot += acc[i][j][r] * bf16_to_f32(a.w[((size_t)tap * a.Cout + co_c) * a.Cin + ci]);
                    }
                    dots[r] = dot;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int v = __builtin_bit_cast(int, dots[r]);
                    auto addf = [](int x, int y) { return __builtin_bit_cast(int, __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y)); };
                    v = addf(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));     // row_shr:1
                    v = addf(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));     // row_shr:2
                    v = addf(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false));     // row_shr:4, banks 1..3
                    v = addf(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false));     // row_shr:8, banks 2..3
                    v = addf(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1, 3
                    const int co = co0 + (wco * TCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                    if (fcol == 31 && co < a.cout_real) atomicAdd(a.wdot + co, __builtin_bit_cast(float, v));
                }
            }
            if constexpr (BETA) {
                if (do_beta && fcol == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + (wco * TCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
                        if (co < a.cout_real) atomicAdd(a.dbeta + co, accb[i][r]);
                    }
                }
            }
        }
    }
    __syncthreads();                                        // every wave is done with the stage buffers
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * TCI * 32);   // this wave's [32][TCI*32] floats
    const int rot = (ks * 5 + tap * 3) & 31;
    float* slab_t = a.slab ? a.slab + (size_t)ks * a.slab_stride + (size_t)tap * a.Cout * a.Cin : nullptr;
#pragma unroll
    for (int i = 0; i < TCO; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < TCI; ++j)
                stg[((r & 3) + 8 * (r >> 2) + 4 * fhalf) * (TCI * 32) + j * 32 + fcol] = acc[i][j][r];
        // (same wave writes and reads: LDS operations of one wave complete in order, no barrier)
        if (slab_t) {
            // split-K partial sums as plain 16-byte stores: one wave instruction = 4 (TCI = 2) or 8 rows of the tile
            constexpr int RW = TCI * 32, LPR = RW / 4, RPI = 64 / LPR;      // floats per row, lanes per row, rows / instr
            const int c4 = (lane % LPR) * 4;
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int row = it * RPI + lane / LPR;
                const int co_l = (wco * TCO + i) * 32 + row;
                const int co = co0 + co_l;
                if (co < a.cout_real) {
                    const float sc = lds_scale[co_l];
                    float4 v = *reinterpret_cast<const float4*>(stg + row * RW + c4);
                    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
                    *reinterpret_cast<float4*>(slab_t + (size_t)co * a.Cin + ci0 + wci * RW + c4) = v;
                }
            }
            continue;
        }
        for (int rr = 0; rr < 32; ++rr) {
            const int row = (rr + rot) & 31;
            if constexpr (TCI == 2) {
                const int co_l = (wco * TCO + i) * 32 + row;
                const int co = co0 + co_l;
                if (co >= a.cout_real) continue;             // wave-uniform
                atomicAdd(dwt + (size_t)co * a.Cin + ci0 + wci * 64 + lane, stg[row * 64 + lane] * lds_scale[co_l]);
            } else {
                // 32 columns per row: the two half-waves take two rows (row, row ^ 16) per instruction
                if (rr >= 16) break;
                const int row2 = (row + 16 * fhalf) & 31;
                const int co2 = co0 + (wco * TCO + i) * 32 + row2;
                if (co2 < a.cout_real)
                    atomicAdd(dwt + (size_t)co2 * a.Cin + ci0 + wci * 32 + fcol,
                              stg[row2 * 32 + fcol] * lds_scale[(wco * TCO + i) * 32 + row2]);
            }
        }
    }
    if (tracing) {
        stamp(6);                                           // atomics issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(7);                                           // ... and acknowledged
        __syncthreads();
        uint32_t* out = a.trace + (size_t)blockIdx.x * CONV_TRACE_DWORDS;
        if (tid == 0) {
            out[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            out[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            out[2] = (uint32_t)rt_start; out[3] = (uint32_t)(rt_start >> 32);
            out[8] = (uint32_t)((p_end - p_begin + 63) >> 6);
            out[9] = (uint32_t)ks; out[10] = (uint32_t)(tap * 256 + tci * 16 + tco);
        }
        for (int i = 4 + tid; i < CONV_TRACE_DWORDS; i += 256)
            if (i < 8 || i >= 16) out[i] = lds_trace[i];
    }
}

// XCD-aware order: all (co, ci, tap) tiles of one pixel slice are consecutive logical ids and therefore run on one XCD, which
// then fetches that slice of dU / X from HBM once instead of once per XCD
__device__ __forceinline__ int xcd_logical_block() {
    const int b = blockIdx.x, nblk = gridDim.x, q = nblk / 8, r = nblk % 8, xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int TCO, int TCI, bool PLAIN, bool BETA, bool DMA = false, int DNS = 1>
__global__ __launch_bounds__(256, DMA ? ((BETA || !PLAIN) ? 3 : 4) : 1) void conv_wgrad_kernel(WgradArgs a) {
    wgrad_body<TCO, TCI, PLAIN, BETA, DMA, DNS>(a, xcd_logical_block());
}

// GROUPED launch: the weight gradients of MANY layers (the bottlenecks of a stretch of the backward pass) in one grid. A
// per-layer launch has 16 ... 144 output tiles and needs 8 ... 21 pixel slices to put even 1.3 workgroups on a CU; each of
// those workgroups then runs ~25 pixel stages and pays a 64 KB atomic epilogue (21-34 k cycles) for them. In a group the
// tiles of all layers fill the machine together: 2-3 slices suffice, a workgroup runs ~200 stages per epilogue, and 3-4
// workgroups per CU overlap each other's load and MFMA phases. The table is device-resident (built once per recorded pass):
// item i owns logical blocks [first_block, first_block + blocks).
struct WgradGroupItem {
    WgradArgs a;
    int first_block;
    int pad_[3];
};

template <bool PLAIN>
__global__ __launch_bounds__(256, PLAIN ? 4 : 3) void conv_wgrad_group_kernel(const WgradGroupItem* __restrict__ items, int n_items) {
    const int b = xcd_logical_block();
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {                                   // last item with first_block <= b (block-uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    const WgradArgs a = items[lo].a;
    wgrad_body<2, 2, PLAIN, false, true, 1>(a, b - __builtin_amdgcn_readfirstlane(items[lo].first_block));
}

}  // namespace cms

template <int TCO, int TCI, bool PLAIN, bool BETA, bool DMA, int DNS>
static void wgrad_launch(dim3 grid, size_t lds, hipStream_t s, const WgradArgs& a) {
    auto kern = conv_wgrad_kernel<TCO, TCI, PLAIN, BETA, DMA, DNS>;
    static bool raised = false;
    if (DMA && !raised) {           // two stages = 64 KB + tables: above what a kernel gets without asking
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
}

// Number of pixel slices (split-K) of a weight-gradient launch and the pixels per slice -- shared by the launch and by
// cms_conv_wgrad_workspace_bytes (a caller sizing the slab of the deterministic combine).
static int wgrad_plan(const cms_wgrad_desc* d, int* per_out, bool* dma_out, int* stages_out) {
    const int M = d->n * d->ho * d->wo;
    const int bco = d->cout % 128 == 0 ? 128 : 64, bci = d->cin % 128 == 0 ? 128 : 64;
    const int tiles = (d->cout / bco) * (d->cin / bci) * d->ntaps;
    // split the pixel axis so that the grid has ~1.5 workgroups per CU (more slices only add atomic traffic and
    // per-workgroup prologue / epilogue); each slice is a multiple of 64 pixels
    // direct-to-LDS loader (buffer addressing: both tensors below 2 GB); CMS_WGRAD_DMA=0 forces the register loader and
    // CMS_WGRAD_TARGET the workgroup count the automatic split aims at (A/B switches, read once)
    static int env_dma = -1, env_target = -1, env_stages = 1;
    if (env_dma < 0) {
        const char* e = getenv("CMS_WGRAD_DMA");
        env_dma = e ? atoi(e) : 1;
        const char* g = getenv("CMS_WGRAD_STAGES");
        env_stages = g ? atoi(g) : 1;
        const char* t = getenv("CMS_WGRAD_TARGET");
        env_target = t ? atoi(t) : 0;
    }
    const bool dma = env_dma != 0 && (size_t)M * d->cout * 2 < (1ull << 31) &&
                     (size_t)d->n * d->h * d->w_in * d->cin * 2 < (1ull << 31);
    const int target = env_target > 0 ? env_target : 384;
    // one stage by default: alone on the chip two stages are ~8 % faster (5.5 vs 6.0 ms over the DeepLab v2 layer
    // list), inside the step -- where the weight gradients share the CUs with the data-gradient convolutions of the
    // other stream -- the 64 KB of LDS per workgroup cost 3-4 % of the step (profiles/r02u_*)
    const int stages = env_stages == 2 ? 2 : 1;
    int ksplit = d->ksplit > 0 ? d->ksplit : (target + tiles - 1) / tiles;
    const bool plain_shape = d->ntaps == 1 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 && d->stride == 1 && d->h == d->ho &&
                             d->w_in == d->wo;
    if (d->ksplit <= 0 && dma && env_target <= 0) {
        // Number of pixel slices from a two-term model (env CMS_WGRAD_MODEL=0: the sweep-tuned rule of profiles/r02r):
        //   the slices' K loops run side by side (<= `cap` resident workgroups): nst / ks stages of t_stage each;
        //   their atomics are served by the memory-side units at ~0.7 TB/s IN TOTAL (tools/atomic_probe.hip): ks * |dW|.
        // The sum is smallest at ks = sqrt(nst * t_stage * 0.7e12 / |dW| bytes). It reproduces the tuned values at
        // the 321 x 321 shapes (21 / 16 / 8 / 10 slices for the 1x1 1024->256, 3x3 256->256, 3x3 512->512, 1x1 2048->512
        // layers) and follows M where the tuning did not go: DeepLab v3+ at 513 x 513 has 10 890 pixels per layer-3/4
        // launch, where 24 slices of 7 stages each spent their time in the atomics.
        static int env_model = -1;
        if (env_model < 0) {
            const char* e = getenv("CMS_WGRAD_MODEL");
            env_model = e ? atoi(e) : 1;
        }
        const int nst = (M + 63) / 64;
        // resident workgroups: 2 per CU with two stages (LDS), 3 where the register budget is 168 (taps / side outputs)
        const int cap = stages == 2 ? 512 : ((plain_shape && d->dbeta == nullptr) ? 864 : 720);
        if (env_model != 0) {
            const double t_stage = plain_shape ? 1.25e-6 : 1.75e-6;
            const double dw_bytes = (double)d->ntaps * d->cout * d->cin * 4.0;
            int ks = (int)(std::sqrt((double)nst * t_stage * 0.7e12 / dw_bytes) + 0.5);
            ks = std::max(1, std::min(ks, std::min(std::max(1, cap / tiles), nst)));
            ksplit = ks;
        } else {
            const int hi = std::min(cap / tiles, nst / 32);
            if (hi > ksplit) ksplit = hi;
        }
    }
    int per = ((M + ksplit - 1) / ksplit + 63) / 64 * 64;
    if (per < 64) per = 64;
    ksplit = (M + per - 1) / per;
    if (per_out) *per_out = per;
    if (dma_out) *dma_out = dma;
    if (stages_out) *stages_out = stages;
    return ksplit;
}

// The eight-phase 256 x 256 kernel (csrc/wgrad8.hip) takes the launches it supports unless switched off: CMS_WGRAD8=0 in the
// environment, or cms_conv_set_wgrad8 (A/B legs and the tests that compare the two kernels in one process).
static int g_wgrad8_mode = -1;                   // -1 = environment (default on), 0 = off, 1 = on
extern "C" int cms_conv_set_wgrad8(int mode) {
    g_wgrad8_mode = mode;
    return CMS_OK;
}
static bool wgrad8_selected(const cms_wgrad_desc* d) {
    static int env = -1;
    if (env < 0) {
        const char* e = getenv("CMS_WGRAD8");
        env = e ? (atoi(e) != 0) : 1;
    }
    const int on = g_wgrad8_mode >= 0 ? g_wgrad8_mode : env;
    return on != 0 && wgrad8_supported(d);
}

extern "C" int cms_conv_wgrad_uses_wgrad8(const cms_wgrad_desc* d) { return wgrad8_selected(d) ? 1 : 0; }

// Bytes of caller-owned scratch that make this launch DETERMINISTIC: with a workspace of at least this size the pixel
// slices write their partial sums as plain stores ([slice][tap][Cout][Cin] fp32) and a second launch on the same stream
// adds them to dw in slice order; without one (NULL) they are combined with fp32 atomics, whose order varies from run to
// run. 0 = the launch has one slice and is deterministic as it is.
extern "C" long long cms_conv_wgrad_workspace_bytes(const cms_wgrad_desc* d) {
    if (!d || d->cin % 64 != 0 || d->cout % 64 != 0 || d->ntaps <= 0 || d->ntaps > CMS_CONV_MAX_TAPS || d->n <= 0 || d->ho <= 0 ||
        d->wo <= 0 || d->cin % 4 != 0)
        return 0;
    const int ks = wgrad8_selected(d) ? wgrad8_plan(d, nullptr) : wgrad_plan(d, nullptr, nullptr, nullptr);
    return ks > 1 ? (long long)ks * d->ntaps * d->cout * d->cin * (long long)sizeof(float) : 0;
}

static int wgrad_fill_args(const cms_wgrad_desc* d, WgradArgs& a) {
    CMS_REQUIRE(d && d->du && d->x && d->dw, "conv_wgrad: NULL pointer");
    CMS_REQUIRE(d->cin % 64 == 0 && d->cout % 64 == 0, "conv_wgrad: Cin (%d) and Cout (%d) must be multiples of 64", d->cin,
                d->cout);
    CMS_REQUIRE(d->ntaps > 0 && d->ntaps <= CMS_CONV_MAX_TAPS, "conv_wgrad: 1..%d taps", CMS_CONV_MAX_TAPS);
    CMS_REQUIRE(d->n > 0 && d->h > 0 && d->w_in > 0 && d->ho > 0 && d->wo > 0 && d->stride >= 1, "conv_wgrad: bad geometry");
    a.du = (const uint16_t*)d->du; a.x = (const uint16_t*)d->x; a.dw = d->dw; a.scale = d->scale;
    a.N = d->n; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout;
    a.ntaps = d->ntaps; a.stride = d->stride;
    a.M = d->n * d->ho * d->wo;
    a.cout_real = d->cout_real > 0 ? d->cout_real : d->cout;
    a.dw_cout = d->dw_cout > 0 ? d->dw_cout : d->cout;
    CMS_REQUIRE(a.dw_cout >= a.cout_real, "conv_wgrad: dw_cout (%d) < cout_real (%d)", a.dw_cout, a.cout_real);
    a.w = (const uint16_t*)d->w; a.wdot = d->wdot; a.dbeta = d->dbeta;
    CMS_REQUIRE((d->wdot == nullptr) == (d->w == nullptr), "conv_wgrad: wdot needs the bf16 weights (w) and vice versa");
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        a.tap_dy[i] = (short)(i < d->ntaps ? d->tap_dy[i] : 0);
        a.tap_dx[i] = (short)(i < d->ntaps ? d->tap_dx[i] : 0);
    }
    CMS_REQUIRE((size_t)d->n * d->h * d->w_in * d->cin < (1u << 31) && (size_t)d->n * d->ho * d->wo * d->cout < (1u << 31),
                "conv_wgrad: tensors must have < 2^31 elements");
    a.slab = nullptr; a.slab_stride = 0; a.trace = nullptr; a.trace_wgs = 0;
    a.ksplit = 1; a.pix_per_split = 64;
    return CMS_OK;
}

static bool wgrad_is_plain(const cms_wgrad_desc* d) {
    return d->ntaps == 1 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 && d->stride == 1 && d->h == d->ho && d->w_in == d->wo;
}

extern "C" int cms_conv_wgrad(const cms_wgrad_desc* d, void* stream) {
    WgradArgs a;
    const int rc0 = wgrad_fill_args(d, a);
    if (rc0) return rc0;
    if (wgrad8_selected(d)) return wgrad8_launch(d, (hipStream_t)stream, g_conv_trace, g_conv_trace_wgs);
    const int bco = d->cout % 128 == 0 ? 128 : 64, bci = d->cin % 128 == 0 ? 128 : 64;
    const int tiles = (d->cout / bco) * (d->cin / bci) * d->ntaps;
    int per = 64, stages = 1;
    bool dma = false;
    int ksplit = wgrad_plan(d, &per, &dma, &stages);
    a.ksplit = ksplit;
    a.pix_per_split = per;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(tiles * ksplit);
    // split-K combine: fp32 atomics (default), or -- when the caller hands over a workspace (cms_conv_wgrad_workspace_bytes)
    // -- slabs + a reduce launch in slice order: DETERMINISTIC sums. Measured (profiles/r02ah_*, r03a_*): alone the layer
    // list is 6 % faster with slabs (5.71 vs 6.06 ms); inside the two-stream step 1.5-2 % slower (497 vs 507 img/s): the
    // atomics are fire-and-forget work for the otherwise idle memory-side units while the other stream computes, the slabs
    // cost HBM bandwidth and one more launch per layer. Throughput runs pass no workspace.
    const size_t slice_elems = (size_t)d->ntaps * d->cout * d->cin;
    const bool use_slab = d->workspace != nullptr && ksplit > 1 && d->cin % 4 == 0 &&
                          (unsigned long long)d->workspace_bytes >= (unsigned long long)ksplit * slice_elems * sizeof(float);
    a.slab = use_slab ? (float*)d->workspace : nullptr;
    a.slab_stride = slice_elems;
    a.trace = g_conv_trace;                      // diagnostic (cms_conv_set_trace); NULL in production
    a.trace_wgs = g_conv_trace_wgs;
    const size_t lds = (dma ? stages : 1) * 2 * 64 * 256 + 128 * 4 + (a.trace ? CONV_TRACE_DWORDS * 4 : 0);
    const bool beta = d->dbeta != nullptr;
    const bool plain = d->ntaps == 1 && d->tap_dy[0] == 0 && d->tap_dx[0] == 0 && d->stride == 1 && d->h == d->ho &&
                       d->w_in == d->wo;
#define CMS_WGRAD_LAUNCH2(TCO, TCI, DMA, DNS)                                                         \
    do {                                                                                              \
        if (plain && beta) wgrad_launch<TCO, TCI, true, true, DMA, DNS>(grid, lds, s, a);             \
        else if (plain) wgrad_launch<TCO, TCI, true, false, DMA, DNS>(grid, lds, s, a);               \
        else if (beta) wgrad_launch<TCO, TCI, false, true, DMA, DNS>(grid, lds, s, a);                \
        else wgrad_launch<TCO, TCI, false, false, DMA, DNS>(grid, lds, s, a);                         \
    } while (0)
#define CMS_WGRAD_LAUNCH(TCO, TCI)                                                                    \
    do {                                                                                              \
        if (dma && stages == 2) CMS_WGRAD_LAUNCH2(TCO, TCI, true, 2);                                 \
        else if (dma) CMS_WGRAD_LAUNCH2(TCO, TCI, true, 1);                                           \
        else CMS_WGRAD_LAUNCH2(TCO, TCI, false, 1);                                                   \
    } while (0)
    if (bco == 128 && bci == 128) CMS_WGRAD_LAUNCH(2, 2);
    else if (bco == 128) CMS_WGRAD_LAUNCH(2, 1);
    else if (bci == 128) CMS_WGRAD_LAUNCH(1, 2);
    else CMS_WGRAD_LAUNCH(1, 1);
#undef CMS_WGRAD_LAUNCH
#undef CMS_WGRAD_LAUNCH2
    if (use_slab) {
        const size_t total4 = (size_t)d->ntaps * a.cout_real * (d->cin / 4);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, s, a.slab, d->dw, ksplit,
                           slice_elems, d->ntaps, d->cout, d->cin, a.cout_real, a.dw_cout);
    }
    return launch_status("cms_conv_wgrad");
}

// ---- grouped weight gradients ------------------------------------------------------------------------------------------
// 0 = this launch cannot join a group; 1 = the group of pointwise (1 x 1, stride 1) launches; 2 = the group of launches with
// taps / strides. Grouped launches take the 128 x 128 tile on the direct-to-LDS loader and combine their slices with fp32
// atomics: no side outputs, no slab.
extern "C" int cms_conv_wgrad_group_kind(const cms_wgrad_desc* d) {
    if (!d || !d->du || !d->x || !d->dw || d->cin % 128 != 0 || d->cout % 128 != 0 || d->ntaps <= 0 || d->ntaps > CMS_CONV_MAX_TAPS ||
        d->wdot || d->dbeta || d->w || d->workspace || d->stride < 1)
        return 0;
    if ((d->cout_real > 0 && d->cout_real != d->cout) || (d->dw_cout > 0 && d->dw_cout != d->cout)) return 0;
    const size_t M = (size_t)d->n * d->ho * d->wo;
    if (M * d->cout * 2 >= (1ull << 31) || (size_t)d->n * d->h * d->w_in * d->cin * 2 >= (1ull << 31)) return 0;
    return wgrad_is_plain(d) ? 1 : 2;
}

extern "C" long long cms_conv_wgrad_group_bytes(int n_items) { return (long long)n_items * (long long)sizeof(WgradGroupItem); }

// Fills the HOST image of the item table (the caller copies it to the device once and keeps it; cms_conv_wgrad_group_run reads
// the device copy). All launches must be of one kind. `target_workgroups` (0 = default) is what the split of the pixel axis aims
// at for the WHOLE group. -> total blocks of the grid in *total_blocks.
extern "C" int cms_conv_wgrad_group_pack(const cms_wgrad_desc* descs, int n, int target_workgroups, void* host_table, long long bytes,
                                         int* total_blocks) {
    CMS_REQUIRE(descs && n > 0 && host_table && total_blocks, "conv_wgrad_group_pack: NULL pointer / empty group");
    CMS_REQUIRE(bytes >= cms_conv_wgrad_group_bytes(n), "conv_wgrad_group_pack: table buffer too small");
    const int kind = cms_conv_wgrad_group_kind(&descs[0]);
    CMS_REQUIRE(kind != 0, "conv_wgrad_group_pack: launch 0 cannot be grouped");
    long long tiles_all = 0;
    for (int i = 0; i < n; ++i) {
        CMS_REQUIRE(cms_conv_wgrad_group_kind(&descs[i]) == kind, "conv_wgrad_group_pack: launch %d is of another kind", i);
        tiles_all += (long long)(descs[i].cout / 128) * (descs[i].cin / 128) * descs[i].ntaps;
    }
    static int env_target = -1;
    if (env_target < 0) {
        const char* e = getenv("CMS_WGRAD_GROUP_TARGET");
        env_target = e ? atoi(e) : 0;
    }
    const long long target = target_workgroups > 0 ? target_workgroups : (env_target > 0 ? env_target : 2048);
    WgradGroupItem* items = (WgradGroupItem*)host_table;
    long long first = 0;
    for (int i = 0; i < n; ++i) {
        const cms_wgrad_desc* d = &descs[i];
        WgradGroupItem& it = items[i];
        memset(&it, 0, sizeof(it));
        const int rc = wgrad_fill_args(d, it.a);
        if (rc) return rc;
        const int M = d->n * d->ho * d->wo;
        const int tiles = (d->cout / 128) * (d->cin / 128) * d->ntaps;
        // the group's slices: about target / (all tiles) per tile, at least 8 pixel stages each
        int ks = (int)std::max<long long>(1, (target + tiles_all / 2) / tiles_all);
        ks = std::min(ks, std::max(1, M / 512));
        int per = ((M + ks - 1) / ks + 63) / 64 * 64;
        if (per < 64) per = 64;
        ks = (M + per - 1) / per;
        it.a.ksplit = ks;
        it.a.pix_per_split = per;
        it.first_block = (int)first;
        first += (long long)tiles * ks;
        CMS_REQUIRE(first < (1ll << 30), "conv_wgrad_group_pack: too many blocks");
    }
    *total_blocks = (int)first;
    return CMS_OK;
}

extern "C" int cms_conv_wgrad_group_run(const void* table_dev, int n_items, int total_blocks, int kind, void* stream) {
    CMS_REQUIRE(table_dev && n_items > 0 && total_blocks > 0 && (kind == 1 || kind == 2), "conv_wgrad_group_run: bad arguments");
    const size_t lds = 2 * 64 * 256 + 128 * 4;
    hipStream_t s = (hipStream_t)stream;
    if (kind == 1)
        hipLaunchKernelGGL(conv_wgrad_group_kernel<true>, dim3(total_blocks), dim3(256), lds, s, (const WgradGroupItem*)table_dev, n_items);
    else
        hipLaunchKernelGGL(conv_wgrad_group_kernel<false>, dim3(total_blocks), dim3(256), lds, s, (const WgradGroupItem*)table_dev, n_items);
    return launch_status("cms_conv_wgrad_group_run");
}
