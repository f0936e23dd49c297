// fp32 implicit-GEMM convolution family on the f32-input MFMA (v_mfma_f32_32x32x2_f32, gfx950): the PARITY
// configuration of the DeepLab bodies and the precision the VAT direction pass needs.
//
// Same GEMM view, tap tables, epilogues and descriptor as csrc/conv.hip (architectures/deeplab2.py:89-109, 124-128 and
// their autograd twins), with activations fp32 NHWC and weights fp32 [tap][Cout][Cin] -- the fp32 MASTER arena itself is
// the forward operand, no bf16 copy is involved. The f32 MFMA is bit-for-bit a k-ordered fmaf chain (one rounding per
// product, MI355X guide section 3), i.e. the arithmetic of a plain fp32 convolution up to summation order; peak is the
// fp32 vector rate (157 TFLOP/s), 1/16 of the bf16 path -- this is the configuration whose losses / IoU are held to
// the 1e-4 of BASELINE.json's north star, the bf16 path is the throughput configuration.
//
// v_mfma_f32_32x32x2_f32 operands: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31], ONE float each.
// The LDS image is the bf16 kernel's: 128-byte rows (32 floats of K), 16-byte chunks XOR-swizzled with (row>>1)&7. A
// lane reads the 16-byte chunk (2*kk + (l>>5)) of row l&31 with one ds_read_b128 and feeds its four floats to four
// successive MFMAs: MFMA e of sub-step kk multiplies k = 8*kk + 4*(l>>5) + e on both operands, so the four MFMAs cover
// the 8 K-values of the chunk pair (any consistent K permutation is a valid GEMM).
#include "common.hpp"

namespace cms {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int F32_BK = 32;            // K elements per stage = 128 B per LDS row
constexpr int F32_ROWB = 128;

struct ConvArgsF32 {
    const float* x;          // [N][H][W][Cin]
    const float* w;          // [ntaps][Cout][Cin]
    float* y;                // [N][out_H][out_W][Cout] or NULL
    float* y32;              // NCHW [N][cout_real][Ho][Wo] or NULL
    const float* scale;
    const float* bias;
    const float* res;
    const float* mask_src;
    int N, H, W, Cin;
    int Ho, Wo, Cout, cout_real;
    int ntaps, stride;
    int out_H, out_W, out_stride;
    int relu, mode;
    int M;
    int ksplit;
    short tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
};

struct RowInfoF32 {
    uint32_t in_off;        // element offset of input pixel (oy*stride, ox*stride), channel 0
    uint32_t yx;            // (oy*stride) << 16 | (ox*stride); 0x70007000 for rows past M
    uint32_t opix;          // output pixel index, 0xffffffff for rows past M
    uint32_t m;
};

__device__ __forceinline__ uint32_t swz32(int row, int chunk) {
    return (uint32_t)row * F32_ROWB + (uint32_t)((chunk ^ ((row >> 1) & 7)) << 4);
}

// WN x WM waves (WN * WM == 4), each TN x TM MFMA tiles of 32 (co) x 32 (pixels)
template <int WN, int TN, int TM>
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvArgsF32 a) {
    constexpr int WM = 4 / WN;
    constexpr int BN = WN * TN * 32;
    constexpr int BM = WM * TM * 32;
    constexpr int PA = BM / 32, PB = BN / 32;       // loader passes (32 rows of 8 chunks per pass of 256 threads)
    static_assert(BM <= 256, "row table is filled by one pass of the workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* lds_x = smem;                          // [BM][128 B]
    unsigned char* lds_w = smem + BM * F32_ROWB;          // [BN][128 B]
    short* lds_tap = reinterpret_cast<short*>(smem + (BM + BN) * F32_ROWB);                    // [2][MAX_TAPS]
    RowInfoF32* lds_row = reinterpret_cast<RowInfoF32*>(smem + (BM + BN) * F32_ROWB + 80);     // [BM]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;

    const int ntn = a.Cout / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    const int ntiles = nblk / a.ksplit;
    const int split = bid / ntiles;
    bid -= split * ntiles;
    const int tile_n = bid % ntn, tile_m = bid / ntn;
    const int co0 = tile_n * BN, m0 = tile_m * BM;

    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
            lds_tap[i] = a.tap_dy[i];
            lds_tap[CMS_CONV_MAX_TAPS + i] = a.tap_dx[i];
        }
    }
    if (tid < BM) {
        const int m = m0 + tid;
        RowInfoF32 ri;
        ri.m = (uint32_t)m;
        if (m < a.M) {
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
    __syncthreads();

    // register-staged loader: thread -> 16-byte chunk (tid & 7) of rows (tid >> 3) + 32*i
    const int chunk = tid & 7, lrow = tid >> 3;
    uint32_t xoff[PA], xyx[PA], woff[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const RowInfoF32 ri = lds_row[lrow + 32 * i];
        xoff[i] = ri.in_off + (uint32_t)(chunk * 4);
        xyx[i] = ri.yx;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) woff[i] = (uint32_t)((lrow + 32 * i) * a.Cin + chunk * 4);
    const uint32_t st_off = swz32(lrow, chunk);           // rows lrow + 32*i share the swizzle term (32 % 16 == 0)

    const int kc_per_tap = a.Cin / F32_BK;
    const int taps_per_split = (a.ntaps + a.ksplit - 1) / a.ksplit;
    const int tap_begin = split * taps_per_split;
    const int tap_end = min(a.ntaps, tap_begin + taps_per_split);
    const int ks_begin = tap_begin * kc_per_tap;
    const int ks_end = tap_end * kc_per_tap;

    f32x4 rx[PA], rw[PB];
    auto load_tile = [&](int ks) {
        const int tap = ks / kc_per_tap;                                  // wave-uniform
        const int c0 = (ks - tap * kc_per_tap) * F32_BK;
        const int dy = __builtin_amdgcn_readfirstlane((int)lds_tap[tap]);
        const int dx = __builtin_amdgcn_readfirstlane((int)lds_tap[CMS_CONV_MAX_TAPS + tap]);
        const int delta = (dy * a.W + dx) * a.Cin + c0;
        const float* wtp = a.w + ((size_t)tap * a.Cout + co0) * a.Cin + c0;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const uint32_t iy = (xyx[i] >> 16) + (uint32_t)dy, ix = (xyx[i] & 0xffffu) + (uint32_t)dx;
            const bool ok = iy < (uint32_t)a.H && ix < (uint32_t)a.W;    // unsigned compare covers the negative side
            if (ok) rx[i] = *reinterpret_cast<const f32x4*>(a.x + (size_t)(xoff[i] + (uint32_t)delta));
            else rx[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) rw[i] = *reinterpret_cast<const f32x4*>(wtp + woff[i]);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(lds_x + st_off + i * 32 * F32_ROWB) = rx[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(lds_w + st_off + i * 32 * F32_ROWB) = rw[i];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fhalf = lane >> 5;
    const uint32_t lane_frag = (uint32_t)frow * F32_ROWB | (uint32_t)(((fhalf ^ (frow >> 1)) & 7) << 4);
    const unsigned char* fw_base = lds_w + wn * TN * 32 * F32_ROWB;
    const unsigned char* fx_base = lds_x + wm * TM * 32 * F32_ROWB;

    if (ks_begin < ks_end) load_tile(ks_begin);
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        __syncthreads();            // previous stage's fragment reads are done
        store_tile();
        __syncthreads();
        if (ks + 1 < ks_end) load_tile(ks + 1);        // in flight during the MFMA phase
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fw[TN], fx[TM];
            const uint32_t fo = lane_frag ^ (uint32_t)(kk * 32);
#pragma unroll
            for (int i = 0; i < TN; ++i) fw[i] = *reinterpret_cast<const f32x4*>(fw_base + fo + i * 32 * F32_ROWB);
#pragma unroll
            for (int j = 0; j < TM; ++j) fx[j] = *reinterpret_cast<const f32x4*>(fx_base + fo + j * 32 * F32_ROWB);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[i][e], fx[j][e], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: every lane owns runs of 4 consecutive channels of one pixel = 16 bytes of the fp32 NHWC tensors
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int prow_l = (wm * TM + j) * 32 + frow;
        const RowInfoF32 ri = lds_row[prow_l];
        const bool valid = ri.opix != 0xffffffffu;
        const size_t obase = (size_t)ri.opix * a.Cout;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = co0 + (wn * TN + i) * 32 + 8 * q + 4 * fhalf;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (a.mode == 0) {
                    if (a.scale) {
                        const float4 sc = *reinterpret_cast<const float4*>(a.scale + co);
                        v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w;
                    }
                    if (a.bias && split == 0) {
                        const float4 b = *reinterpret_cast<const float4*>(a.bias + co);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                }
                if (a.res && valid) {
                    const float4 rr = *reinterpret_cast<const float4*>(a.res + obase + co);
                    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                }
                if (a.mode == 0) {
                    if (a.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                } else if (a.mask_src && valid) {
                    const float4 mk = *reinterpret_cast<const float4*>(a.mask_src + obase + co);
                    v[0] = mk.x > 0.0f ? v[0] : 0.0f;
                    v[1] = mk.y > 0.0f ? v[1] : 0.0f;
                    v[2] = mk.z > 0.0f ? v[2] : 0.0f;
                    v[3] = mk.w > 0.0f ? v[3] : 0.0f;
                }
                if (a.y && valid) *reinterpret_cast<float4*>(a.y + obase + co) = float4{v[0], v[1], v[2], v[3]};
                if (a.y32 && valid) {
                    const int m = (int)ri.m;
                    const int ox = m % a.Wo;
                    const int t = m / a.Wo;
                    const int oy = t % a.Ho;
                    const int n = t / a.Ho;
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

// ---------------------------------------------------------------------------------------------------------------
// fp32 weight gradient: dW[tap][co][ci] += scale[co] * sum_pixels dU[pix][co] * X[pix shifted by tap][ci]
// K = pixels. The f32 MFMA wants ONE float per lane per operand: lane l reads dU[pixel 2*kk + (l>>5)][co (l&31)] --
// 32 consecutive floats per half-wave straight out of a [pixel][channel] LDS image, no transposition (the bf16 kernel
// needs ds_read_b64_tr_b16 for this). Workgroup = 4 waves (2 x 2), up to 128 (co) x 128 (ci) outputs of one tap over one
// slice of the pixel axis; fp32 atomics into the gradient arena.
struct WgradArgsF32 {
    const float* du;
    const float* x;
    float* dw;
    const float* scale;
    int N, H, W, Cin, Ho, Wo, Cout;
    int ntaps, stride;
    int M;
    int ksplit, pix_per_split;
    int cout_real;
    int dw_cout;
    short tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
};

template <int TCO, int TCI>
__global__ __launch_bounds__(256) void conv_wgrad_f32_kernel(WgradArgsF32 a) {
    constexpr int BCO = 2 * TCO * 32, BCI = 2 * TCI * 32;
    constexpr int KP = 32;                                   // pixels per stage
    constexpr int CRU = BCO / 4, CRX = BCI / 4;              // 16-byte chunks per LDS row
    constexpr int PU = KP * CRU / 256, PX = KP * CRX / 256;  // loader passes
    __shared__ __attribute__((aligned(16))) float lds_u[KP * BCO];
    __shared__ __attribute__((aligned(16))) float lds_x[KP * BCI];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wco = wave & 1, wci = wave >> 1;
    const int nco = a.Cout / BCO, nci = a.Cin / BCI;
    int b = blockIdx.x;
    const int tco = b % nco; b /= nco;
    const int tci = b % nci; b /= nci;
    const int tap = b % a.ntaps; b /= a.ntaps;
    const int ks = b;
    const int co0 = tco * BCO, ci0 = tci * BCI;
    int dy = 0, dx = 0;
#pragma unroll
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        if (i == tap) { dy = a.tap_dy[i]; dx = a.tap_dx[i]; }
    }
    const int p_begin = ks * a.pix_per_split;
    const int p_end = min(a.M, p_begin + a.pix_per_split);
    if (p_begin >= p_end) return;

    f32x4 ru[PU], rxx[PX];
    auto load_tile = [&](int p0) {
#pragma unroll
        for (int i = 0; i < PU; ++i) {
            const int c = (tid + 256 * i) % CRU, row = (tid + 256 * i) / CRU;
            const int m = p0 + row;
            ru[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (m < p_end) ru[i] = *reinterpret_cast<const f32x4*>(a.du + (size_t)m * a.Cout + co0 + c * 4);
        }
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int c = (tid + 256 * i) % CRX, row = (tid + 256 * i) / CRX;
            const int m = p0 + row;
            rxx[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (m < p_end) {
                const int ox = m % a.Wo;
                const int t = m / a.Wo;
                const int oy = t % a.Ho;
                const int n = t / a.Ho;
                const uint32_t iy = (uint32_t)(oy * a.stride + dy), ix = (uint32_t)(ox * a.stride + dx);
                if (iy < (uint32_t)a.H && ix < (uint32_t)a.W)
                    rxx[i] = *reinterpret_cast<const f32x4*>(
                        a.x + ((size_t)(n * a.H + (int)iy) * a.W + (int)ix) * a.Cin + ci0 + c * 4);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PU; ++i) *reinterpret_cast<f32x4*>(lds_u + (size_t)(tid + 256 * i) * 4) = ru[i];
#pragma unroll
        for (int i = 0; i < PX; ++i) *reinterpret_cast<f32x4*>(lds_x + (size_t)(tid + 256 * i) * 4) = rxx[i];
    };

    f32x16 acc[TCO][TCI];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int fcol = lane & 31, fhalf = lane >> 5;
    load_tile(p_begin);
    for (int p0 = p_begin; p0 < p_end; p0 += KP) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (p0 + KP < p_end) load_tile(p0 + KP);
#pragma unroll
        for (int kk = 0; kk < KP / 2; ++kk) {
            const int p = 2 * kk + fhalf;
            float fu[TCO], fx[TCI];
#pragma unroll
            for (int i = 0; i < TCO; ++i) fu[i] = lds_u[p * BCO + (wco * TCO + i) * 32 + fcol];
#pragma unroll
            for (int j = 0; j < TCI; ++j) fx[j] = lds_x[p * BCI + (wci * TCI + j) * 32 + fcol];
#pragma unroll
            for (int i = 0; i < TCO; ++i)
#pragma unroll
                for (int j = 0; j < TCI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fu[i], fx[j], acc[i][j], 0, 0, 0);
        }
    }

    float* dwt = a.dw + (size_t)tap * a.dw_cout * a.Cin;
#pragma unroll
    for (int i = 0; i < TCO; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wco * TCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            if (co >= a.cout_real) continue;
            const float s = a.scale ? a.scale[co] : 1.0f;
#pragma unroll
            for (int j = 0; j < TCI; ++j) {
                const int ci = ci0 + (wci * TCI + j) * 32 + fcol;
                atomicAdd(dwt + (size_t)co * a.Cin + ci, acc[i][j][r] * s);
            }
        }
    }
}

// dgrad operand in fp32: wT[tap'][ci][co] = w[tap][co][ci] * scale[co]
__global__ __launch_bounds__(256) void pack_transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                 const float* __restrict__ scale, int ntaps, int Cout,
                                                                 int Cin, int flip) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* s = src + (size_t)tap * Cout * Cin;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.0f;
        if (co < Cout && ci < Cin) {
            v = s[(size_t)co * Cin + ci];
            if (scale) v *= scale[co];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    const int otap = flip ? ntaps - 1 - tap : tap;
    float* d = dst + (size_t)otap * Cin * Cout;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) d[(size_t)ci * Cout + co] = tile[tx][r];
    }
}

__global__ __launch_bounds__(256) void pack_transpose_batch_f32_kernel(const cms_pack_item* __restrict__ items, int n_items) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n_items - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
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
            v = ((const float*)it.src)[base + (size_t)co * Cin + ci];
            if (it.scale) v *= it.scale[co];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    float* d = (float*)it.dst + base;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) d[(size_t)ci * Cout + co] = tile[tx][r];
    }
}

}  // namespace cms

using namespace cms;

template <int WN, int TN, int TM>
static void conv_f32_launch(const ConvArgsF32& a, hipStream_t s) {
    constexpr int BN = WN * TN * 32, BM = (4 / WN) * TM * 32;
    const int grid = (a.Cout / BN) * ((a.M + BM - 1) / BM) * a.ksplit;
    const size_t lds = (size_t)(BN + BM) * F32_ROWB + 80 + BM * 16;
    hipLaunchKernelGGL((conv_f32_kernel<WN, TN, TM>), dim3(grid), dim3(256), lds, s, a);
}

extern "C" int cms_conv_igemm_f32(const cms_conv_desc* d, void* stream) {
    CMS_REQUIRE(d != nullptr, "conv_f32: null descriptor");
    CMS_REQUIRE(d->x && d->w && (d->y || d->y32), "conv_f32: NULL tensor");
    CMS_REQUIRE(d->n > 0 && d->h > 0 && d->w_in > 0 && d->cin > 0 && d->ho > 0 && d->wo > 0 && d->cout > 0,
                "conv_f32: bad geometry");
    CMS_REQUIRE(d->stats_out == nullptr && d->bstats_u == nullptr && d->mask_gates_res == 0 && d->mask_bits == nullptr &&
                    d->mask_bits_out == nullptr,
                "conv_f32: ReLU mask bits and epilogue statistics exist on the bf16 entry point only (cms_conv_igemm)");
    CMS_REQUIRE(d->cin % F32_BK == 0, "conv_f32: Cin (%d) must be a multiple of %d", d->cin, F32_BK);
    CMS_REQUIRE(d->cout % 32 == 0, "conv_f32: Cout (%d) must be a multiple of 32 (pad the weight tensor)", d->cout);
    CMS_REQUIRE(d->ntaps > 0 && d->ntaps <= CMS_CONV_MAX_TAPS, "conv_f32: 1..%d taps", CMS_CONV_MAX_TAPS);
    CMS_REQUIRE(d->stride >= 1 && d->out_stride >= 1, "conv_f32: bad stride");
    CMS_REQUIRE(d->y == nullptr || d->cout_real == d->cout, "conv_f32: NHWC output needs cout_real == cout");
    CMS_REQUIRE((size_t)d->n * d->h * d->w_in * d->cin < (1u << 31) && (size_t)d->n * d->ho * d->wo < (1u << 31) &&
                    (size_t)d->n * d->out_h * d->out_w < (1u << 31) && d->h < 0x7000 && d->w_in < 0x7000,
                "conv_f32: too many pixels");
    ConvArgsF32 a;
    a.x = (const float*)d->x; a.w = (const float*)d->w; a.y = (float*)d->y; a.y32 = d->y32;
    a.scale = d->scale; a.bias = d->bias; a.res = (const float*)d->res; a.mask_src = (const float*)d->mask_src;
    a.N = d->n; a.H = d->h; a.W = d->w_in; a.Cin = d->cin;
    a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout; a.cout_real = d->cout_real;
    a.ntaps = d->ntaps; a.stride = d->stride;
    a.out_H = d->out_h; a.out_W = d->out_w; a.out_stride = d->out_stride;
    a.relu = d->relu; a.mode = d->mode;
    a.M = d->n * d->ho * d->wo;
    a.ksplit = d->ksplit > 1 ? d->ksplit : 1;
    CMS_REQUIRE(a.ksplit == 1 || (d->y == nullptr && d->relu == 0 && d->res == nullptr && d->mode == 0),
                "conv_f32: ksplit needs the NCHW output without residual / ReLU (pre-zeroed, accumulated with atomics)");
    CMS_REQUIRE(a.ksplit <= d->ntaps, "conv_f32: ksplit (%d) > taps (%d)", a.ksplit, d->ntaps);
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        a.tap_dy[i] = (short)(i < d->ntaps ? d->tap_dy[i] : 0);
        a.tap_dx[i] = (short)(i < d->ntaps ? d->tap_dx[i] : 0);
    }
    hipStream_t s = (hipStream_t)stream;
    if (d->cout % 128 == 0) conv_f32_launch<2, 2, 2>(a, s);        // 128 co x 128 pixels
    else if (d->cout % 64 == 0) conv_f32_launch<2, 1, 2>(a, s);    // 64 co x 128 pixels
    else conv_f32_launch<1, 1, 1>(a, s);                           // 32 co x 128 pixels
    return launch_status("cms_conv_igemm_f32");
}

extern "C" int cms_conv_wgrad_f32(const cms_wgrad_desc* d, void* stream) {
    CMS_REQUIRE(d && d->du && d->x && d->dw, "conv_wgrad_f32: NULL pointer");
    CMS_REQUIRE(d->cin % 64 == 0 && d->cout % 64 == 0, "conv_wgrad_f32: Cin (%d) and Cout (%d) must be multiples of 64",
                d->cin, d->cout);
    CMS_REQUIRE(d->ntaps > 0 && d->ntaps <= CMS_CONV_MAX_TAPS, "conv_wgrad_f32: 1..%d taps", CMS_CONV_MAX_TAPS);
    CMS_REQUIRE(d->n > 0 && d->h > 0 && d->w_in > 0 && d->ho > 0 && d->wo > 0 && d->stride >= 1, "conv_wgrad_f32: bad geometry");
    CMS_REQUIRE(d->w == nullptr && d->wdot == nullptr && d->dbeta == nullptr,
                "conv_wgrad_f32: the BatchNorm-affine side outputs exist on the bf16 path only");
    CMS_REQUIRE((size_t)d->n * d->h * d->w_in * d->cin < (1u << 31) && (size_t)d->n * d->ho * d->wo * d->cout < (1u << 31),
                "conv_wgrad_f32: tensors must have < 2^31 elements");
    WgradArgsF32 a;
    a.du = (const float*)d->du; a.x = (const float*)d->x; a.dw = d->dw; a.scale = d->scale;
    a.N = d->n; a.H = d->h; a.W = d->w_in; a.Cin = d->cin; a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout;
    a.ntaps = d->ntaps; a.stride = d->stride;
    a.M = d->n * d->ho * d->wo;
    a.cout_real = d->cout_real > 0 ? d->cout_real : d->cout;
    a.dw_cout = d->dw_cout > 0 ? d->dw_cout : d->cout;
    CMS_REQUIRE(a.dw_cout >= a.cout_real, "conv_wgrad_f32: dw_cout (%d) < cout_real (%d)", a.dw_cout, a.cout_real);
    for (int i = 0; i < CMS_CONV_MAX_TAPS; ++i) {
        a.tap_dy[i] = (short)(i < d->ntaps ? d->tap_dy[i] : 0);
        a.tap_dx[i] = (short)(i < d->ntaps ? d->tap_dx[i] : 0);
    }
    const int bco = d->cout % 128 == 0 ? 128 : 64, bci = d->cin % 128 == 0 ? 128 : 64;
    const int tiles = (d->cout / bco) * (d->cin / bci) * d->ntaps;
    int ksplit = d->ksplit > 0 ? d->ksplit : (384 + tiles - 1) / tiles;
    int per = ((a.M + ksplit - 1) / ksplit + 31) / 32 * 32;
    if (per < 32) per = 32;
    ksplit = (a.M + per - 1) / per;
    a.ksplit = ksplit;
    a.pix_per_split = per;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(tiles * ksplit);
    if (bco == 128 && bci == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<2, 2>), grid, dim3(256), 0, s, a);
    else if (bco == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<2, 1>), grid, dim3(256), 0, s, a);
    else if (bci == 128) hipLaunchKernelGGL((conv_wgrad_f32_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_f32_kernel<1, 1>), grid, dim3(256), 0, s, a);
    return launch_status("cms_conv_wgrad_f32");
}

extern "C" int cms_conv_pack_transpose_f32(const float* src, float* dst, const float* scale, int ntaps, int cout,
                                           int cin, int flip, void* stream) {
    CMS_REQUIRE(src && dst, "conv_pack_transpose_f32: NULL pointer");
    CMS_REQUIRE(ntaps > 0 && cout > 0 && cin > 0, "conv_pack_transpose_f32: bad geometry");
    dim3 grid((cin + 31) / 32, (cout + 31) / 32, ntaps);
    hipLaunchKernelGGL(pack_transpose_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, scale, ntaps, cout,
                       cin, flip);
    return launch_status("cms_conv_pack_transpose_f32");
}

extern "C" int cms_conv_pack_transpose_batch_f32(const cms_pack_item* items_dev, int n_items, int total_blocks,
                                                 void* stream) {
    CMS_REQUIRE(items_dev && n_items > 0 && total_blocks > 0, "conv_pack_transpose_batch_f32: empty table");
    hipLaunchKernelGGL(pack_transpose_batch_f32_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev,
                       n_items);
    return launch_status("cms_conv_pack_transpose_batch_f32");
}
