// The network stem on hand-written kernels: 7x7 / stride 2 / pad 3 convolution (3 -> 64 channels) + frozen BatchNorm +
// ReLU, the 3x3 / stride 2 / pad 1 ceil-mode max-pool, and their backward passes
// (architectures/deeplab2.py:140-146 construction, :183-186 forward; the reference runs them through cuDNN).
//
// 0.3 % of the network's FLOPs with Cin = 3: not GEMM-shaped (K = 147), so this is VALU work with fp32 accumulation:
//   stem_fwd      thread = one output pixel x all 64 channels; the 37 x 37 x 3 input patch of a 16 x 16 pixel tile is
//                 staged through LDS (converted to fp32 once), the weights -- wave-uniform -- come in through the SCALAR
//                 cache ([tap][co] packed copy, s_load + v_fmac with an SGPR operand: no LDS / VGPR traffic for them);
//                 epilogue scale / bias / ReLU, one NHWC row of 64 channels per thread (full cache lines).
//   maxpool_fwd   thread = one output pixel x 8 channels (16-byte vectors), first-maximum-wins like ATen
//                 (`val > max || isnan(val)`, scan order ky, kx), argmax position kept in one byte per element.
//   maxpool_bwd   gather form (no atomics): thread = one stem-output pixel x 8 channels, looks at the <= 4 windows that
//                 contain it, takes their gradient where the stored argmax is this pixel; fused with the ReLU mask.
//   stem_wgrad    dW[co][c][ky][kx] += scale[co] * sum_pixels dS[pix][co] * x[c][2*oy-3+ky][2*ox-3+kx]; thread = one
//                 output channel x 5-6 (c, ky) rows x 7 kx; dS tile and input patch in LDS, patch reads are broadcasts;
//                 persistent blocks, one round of fp32 atomics per block at the end.
//   stem_dgrad    gradient wrt the image (VAT direction pass only): thread = one input pixel x 3 channels.
// Input images are NCHW (the reference's batch layout, A0), activations NHWC.
#include <cstdlib>
#include "common.hpp"

namespace cms {

constexpr int STEM_K = 7, STEM_TAPS = 147, STEM_CO = 64;

template <class T>
__device__ __forceinline__ float ld_f32(const T* p);
template <>
__device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_f32<uint16_t>(const uint16_t* p) { return bf16_to_f32(*p); }

template <class T>
__device__ __forceinline__ void st_f32(T* p, float v);
template <>
__device__ __forceinline__ void st_f32<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st_f32<uint16_t>(uint16_t* p, float v) { *p = f32_to_bf16(v); }

// w_packed[(c*7 + ky)*7 + kx][co] = w[ky][kx][co][c]   (source: the arena's physical [kh][kw][Cout][Cin] layout)
template <class T>
__global__ __launch_bounds__(256) void stem_pack_kernel(const T* __restrict__ w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= STEM_TAPS * STEM_CO) return;
    const int co = i % STEM_CO, q = i / STEM_CO;
    const int c = q / 49, ky = (q % 49) / 7, kx = q % 7;
    out[i] = ld_f32(w + ((size_t)(ky * 7 + kx) * STEM_CO + co) * 3 + c);
}

constexpr int SF_T = 16;                       // output tile edge
constexpr int SF_P = (SF_T - 1) * 2 + STEM_K;  // 37: input patch edge
constexpr int SF_PW = SF_P + 1;                // row pitch

template <class TX, class TY>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const TX* __restrict__ x, TY* __restrict__ y,
                                                       const float* __restrict__ wp, const float* __restrict__ scale,
                                                       const float* __restrict__ bias, int N, int H, int W, int Ho,
                                                       int Wo) {
    __shared__ float patch[3][SF_P][SF_PW];
    const int n = blockIdx.z, ty0 = blockIdx.y * SF_T, tx0 = blockIdx.x * SF_T;
    const int tid = threadIdx.x;
    const int iy0 = ty0 * 2 - 3, ix0 = tx0 * 2 - 3;
    for (int i = tid; i < 3 * SF_P * SF_P; i += 256) {
        const int px = i % SF_P, r = i / SF_P, py = r % SF_P, c = r / SF_P;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.0f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld_f32(x + (((size_t)n * 3 + c) * H + iy) * W + ix);
        patch[c][py][px] = v;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    const int oy = ty0 + ty, ox = tx0 + tx;
    float acc[STEM_CO];
#pragma unroll
    for (int co = 0; co < STEM_CO; ++co) acc[co] = 0.0f;
    for (int r = 0; r < 21; ++r) {                      // (c, ky) rows; kx unrolled
        const int c = r / 7, ky = r % 7;
        const float* prow = &patch[c][ty * 2 + ky][tx * 2];
        const float* wrow = wp + (size_t)r * 7 * STEM_CO;    // wave-uniform -> scalar loads
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float xv = prow[kx];
#pragma unroll
            for (int co = 0; co < STEM_CO; ++co) acc[co] = fmaf(xv, wrow[kx * STEM_CO + co], acc[co]);
        }
    }
    if (oy < Ho && ox < Wo) {
        TY* dst = y + (((size_t)n * Ho + oy) * Wo + ox) * STEM_CO;
#pragma unroll
        for (int co = 0; co < STEM_CO; ++co) {
            float v = acc[co] * scale[co] + bias[co];
            st_f32(dst + co, fmaxf(v, 0.0f));
        }
    }
}

// ---- the same forward on the matrix cores (bf16 in, bf16 out) -----------------------------------------------------
// K is ordered (c, ky) x kx with kx padded from 7 to 8: 21 rows of 8 = 168 -> 11 MFMA K steps of 16 (two rows each, the
// 22nd row has zero weights). A row's 8 operands of pixel (ty, tx) are patch[c][2 ty + ky][2 tx .. 2 tx + 7]: 16
// CONTIGUOUS bytes of the LDS patch, so the pixel operand of v_mfma_f32_32x32x16_bf16 is read with two ds_read2_b32 (4-byte
// aligned) -- no im2col buffer. The weights keep fp32 precision as a bf16 pair hi + lo (two MFMAs per step): the step's
// numerics stay those of the VALU kernel (fp32 weights x bf16 pixels, fp32 accumulation) to 2^-17 relative.
// Workgroup = 16 x 16 output pixels x 64 channels; wave (wc, wp) = channel half wc x pixel half wp: its 11 x 2 weight
// fragments (hi, lo) live in registers for the whole (persistent) workgroup, accumulators 4 x [32 co x 32 px].
typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef float sf32x16 __attribute__((ext_vector_type(16)));
constexpr int SM_ROWS = 22;                    // (c, ky) rows incl. the zero row
constexpr int SM_PITCH = 40;                   // patch row pitch in elements (80 B: rows stay 4-byte aligned)

template <class T>
__global__ __launch_bounds__(256) void stem_pack_frag_kernel(const T* __restrict__ w, uint16_t* __restrict__ frag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= STEM_CO * SM_ROWS * 8) return;
    const int kx = i & 7, r = (i >> 3) % SM_ROWS, co = i / (8 * SM_ROWS);
    float v = 0.0f;
    if (r < 21 && kx < 7) v = ld_f32(w + ((size_t)((r % 7) * 7 + kx) * STEM_CO + co) * 3 + r / 7);
    const uint16_t hi = f32_to_bf16(v);
    frag[i] = hi;
    frag[STEM_CO * SM_ROWS * 8 + i] = f32_to_bf16(v - bf16_to_f32(hi));
}

__global__ __launch_bounds__(256, 2) void stem_fwd_mfma_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                               const uint16_t* __restrict__ frag,
                                                               const float* __restrict__ scale, const float* __restrict__ bias,
                                                               int N, int H, int W, int Ho, int Wo) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SF_P * SM_PITCH * 2 + 256 * STEM_CO * 2];
    uint16_t* patch = reinterpret_cast<uint16_t*>(smem);                           // [3][37][40] bf16
    unsigned char* stage = smem + 3 * SF_P * SM_PITCH * 2;                         // [256 px][128 B], 16-byte chunks swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1;
    const int fcol = lane & 31, fhalf = lane >> 5;

    // this wave's weight fragments: row 2s + fhalf of channel wc*32 + fcol, hi and lo
    su32x4 whi[11], wlo[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) {
        const size_t e = ((size_t)(wc * 32 + fcol) * SM_ROWS + 2 * s + fhalf) * 8;
        whi[s] = *reinterpret_cast<const su32x4*>(frag + e);
        wlo[s] = *reinterpret_cast<const su32x4*>(frag + STEM_CO * SM_ROWS * 8 + e);
    }
    // the pad columns 37..39 of the patch are read (times a zero weight): keep them finite
    for (int i = tid; i < 3 * SF_P * 3; i += 256) patch[(i / 3) * SM_PITCH + SF_P + i % 3] = 0;

    const int tiles_x = (Wo + SF_T - 1) / SF_T, tiles_y = (Ho + SF_T - 1) / SF_T;
    const int ntiles = N * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int txi = t % tiles_x, tyi = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int ty0 = tyi * SF_T, tx0 = txi * SF_T;
        const int iy0 = ty0 * 2 - 3, ix0 = tx0 * 2 - 3;
        __syncthreads();                                   // the previous tile's patch reads and stage reads are done
        for (int i = tid; i < 3 * SF_P * SF_P; i += 256) {
            const int px = i % SF_P, r = i / SF_P, py = r % SF_P, c = r / SF_P;
            const int iy = iy0 + py, ix = ix0 + px;
            uint16_t v = 0;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)n * 3 + c) * H + iy) * W + ix];
            patch[(c * SF_P + py) * SM_PITCH + px] = v;
        }
        __syncthreads();

        sf32x16 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 11; ++s) {
            const int r = min(2 * s + fhalf, 20);          // row 21 does not exist: re-read row 20 (its weights are zero)
            const int c = r / 7, ky = r - c * 7;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = (4 * wp + q) * 32 + fcol;    // pixel of the tile: row p >> 4, column p & 15
                const uint32_t* src = reinterpret_cast<const uint32_t*>(
                    patch + (c * SF_P + 2 * (p >> 4) + ky) * SM_PITCH + 2 * (p & 15));
                const su32x4 b = su32x4{src[0], src[1], src[2], src[3]};
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sbf16x8, whi[s]),
                                                                 __builtin_bit_cast(sbf16x8, b), acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sbf16x8, wlo[s]),
                                                                 __builtin_bit_cast(sbf16x8, b), acc[q], 0, 0, 0);
            }
        }
        // epilogue: lane holds, per accumulator, pixel column fcol and channels (r & 3) + 8 (r >> 2) + 4 fhalf of its
        // half: BN affine + ReLU, 8-byte cells into the staging tile (16-byte chunks XOR-swizzled with the pixel)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = (4 * wp + q) * 32 + fcol;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = wc * 32 + 8 * g + 4 * fhalf;
                const float4 sc = *reinterpret_cast<const float4*>(scale + co);
                const float4 bi = *reinterpret_cast<const float4*>(bias + co);
                const float v0 = fmaxf(acc[q][4 * g + 0] * sc.x + bi.x, 0.0f), v1 = fmaxf(acc[q][4 * g + 1] * sc.y + bi.y, 0.0f);
                const float v2 = fmaxf(acc[q][4 * g + 2] * sc.z + bi.z, 0.0f), v3 = fmaxf(acc[q][4 * g + 3] * sc.w + bi.w, 0.0f);
                uint2 o;
                o.x = (uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16);
                o.y = (uint32_t)f32_to_bf16(v2) | ((uint32_t)f32_to_bf16(v3) << 16);
                *reinterpret_cast<uint2*>(stage + p * 128 + ((((co >> 3) ^ p) & 7) << 4) + ((co & 4) << 1)) = o;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {                      // 256 pixels x 8 chunks of 16 bytes
            const int idx = i * 256 + tid, p = idx >> 3, ch = idx & 7;
            const int oy = ty0 + (p >> 4), ox = tx0 + (p & 15);
            if (oy < Ho && ox < Wo) {
                const su32x4 v = *reinterpret_cast<const su32x4*>(stage + p * 128 + (((ch ^ p) & 7) << 4));
                *reinterpret_cast<su32x4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * STEM_CO + ch * 8) = v;
            }
        }
    }
}

// ---- max-pool 3x3 / 2 / pad 1, NHWC, C % 8 == 0
// 8 channels per thread as ONE 16-byte (bf16) / two 16-byte (fp32) vector accesses and one 8-byte access to the argmax bytes
// (round 4: the element-wise 2-byte loads and stores of the first version made the backward 158 us for 91 MB at 321 x 321 --
// it sits alone at the end of the step, between the last data gradient and the optimizer)
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const uint16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4*>(p + 4) = float4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void st8(uint16_t* p, const float (&v)[8]) {
    uint4 a;
    a.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
    a.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
    a.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
    a.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = a;
}

template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ s, T* __restrict__ p,
                                                          uint8_t* __restrict__ idx, int N, int Hs, int Ws, int Hp,
                                                          int Wp, int C) {
    const int cv = C / 8;
    const size_t total = (size_t)N * Hp * Wp * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv);
        size_t t = i / cv;
        const int px = (int)(t % Wp); t /= Wp;
        const int py = (int)(t % Hp);
        const int n = (int)(t / Hp);
        float best[8];
        uint32_t bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = py * 2 - 1 + ky;
            if (yy < 0 || yy >= Hs) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = px * 2 - 1 + kx;
                if (xx < 0 || xx >= Ws) continue;
                float v[8];
                ld8(s + (((size_t)n * Hs + yy) * Ws + xx) * C + c8 * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = (uint32_t)(ky * 3 + kx); }
            }
        }
        const size_t o = (((size_t)n * Hp + py) * Wp + px) * C + c8 * 8;
        st8(p + o, best);
        uint2 pk;
        pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
        *reinterpret_cast<uint2*>(idx + o) = pk;
    }
}

// dS[n,y,x,c] = [s > 0] * sum over windows (py,px) containing (y,x) with argmax == (y,x) of dP[n,py,px,c]
template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dp, const uint8_t* __restrict__ idx,
                                                          const T* __restrict__ s, T* __restrict__ ds, int N, int Hs,
                                                          int Ws, int Hp, int Wp, int C) {
    const int cv = C / 8;
    const size_t total = (size_t)N * Hs * Ws * cv;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv);
        size_t t = i / cv;
        const int x = (int)(t % Ws); t /= Ws;
        const int y = (int)(t % Hs);
        const int n = (int)(t / Hs);
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.0f;
        // windows: py*2-1 <= y <= py*2+1
        const int py_lo = max(0, (y) / 2), py_hi = min(Hp - 1, (y + 1) / 2);
        const int px_lo = max(0, (x) / 2), px_hi = min(Wp - 1, (x + 1) / 2);
        for (int py = py_lo; py <= py_hi; ++py) {
            const int ky = y - (py * 2 - 1);
            if (ky < 0 || ky > 2) continue;
            for (int px = px_lo; px <= px_hi; ++px) {
                const int kx = x - (px * 2 - 1);
                if (kx < 0 || kx > 2) continue;
                const size_t o = (((size_t)n * Hp + py) * Wp + px) * C + c8 * 8;
                const uint32_t want = (uint32_t)(ky * 3 + kx);
                const uint2 pk = *reinterpret_cast<const uint2*>(idx + o);
                float v[8];
                ld8(dp + o, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t who = ((e < 4 ? pk.x : pk.y) >> (8 * (e & 3))) & 0xffu;
                    g[e] += who == want ? v[e] : 0.0f;
                }
            }
        }
        const size_t so = (((size_t)n * Hs + y) * Ws + x) * C + c8 * 8;
        float sv[8];
        ld8(s + so, sv);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = sv[e] > 0.0f ? g[e] : 0.0f;
        st8(ds + so, g);
    }
}

// ---- weight gradient of the stem convolution
constexpr int SW_TH = 8, SW_TW = 16;                       // output tile (rows x cols) = 128 pixels
constexpr int SW_PH = (SW_TH - 1) * 2 + STEM_K;            // 21
constexpr int SW_PW = (SW_TW - 1) * 2 + STEM_K;            // 37

template <class TX, class TS>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const TX* __restrict__ x, const TS* __restrict__ ds,
                                                         float* __restrict__ dw, const float* __restrict__ scale, int N,
                                                         int H, int W, int Ho, int Wo, float* __restrict__ slab) {
    __shared__ float dst[SW_TH * SW_TW][STEM_CO];          // 32 KB
    __shared__ float patch[3][SW_PH][SW_PW + 1];
    const int tid = threadIdx.x, co = tid & 63, g = tid >> 6;     // g = wave index: wave-uniform
    const int r0 = g * 5, nr = g == 3 ? 6 : 5;                    // (c, ky) rows of this wave: 5 + 5 + 5 + 6 = 21
    float acc[6][7];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[i][k] = 0.0f;
    const int tiles_x = (Wo + SW_TW - 1) / SW_TW, tiles_y = (Ho + SW_TH - 1) / SW_TH;
    const int ntiles = N * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int txi = t % tiles_x, tyi = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int oy0 = tyi * SW_TH, ox0 = txi * SW_TW;
        __syncthreads();                                    // the previous tile's reads are done
        for (int i = tid; i < SW_TH * SW_TW * STEM_CO; i += 256) {
            const int c = i & 63, p = i >> 6;
            const int oy = oy0 + p / SW_TW, ox = ox0 + p % SW_TW;
            float v = 0.0f;
            if (oy < Ho && ox < Wo) v = ld_f32(ds + (((size_t)n * Ho + oy) * Wo + ox) * STEM_CO + c);
            dst[p][c] = v;
        }
        const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
        for (int i = tid; i < 3 * SW_PH * SW_PW; i += 256) {
            const int px = i % SW_PW, r = i / SW_PW, py = r % SW_PH, c = r / SW_PH;
            const int iy = iy0 + py, ix = ix0 + px;
            float v = 0.0f;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld_f32(x + (((size_t)n * 3 + c) * H + iy) * W + ix);
            patch[c][py][px] = v;
        }
        __syncthreads();
        for (int p = 0; p < SW_TH * SW_TW; ++p) {
            const float dy = dst[p][co];
            const int py = (p / SW_TW) * 2, px = (p % SW_TW) * 2;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (i < nr) {
                    const int r = r0 + i;
                    const float* prow = &patch[r / 7][py + r % 7][px];       // uniform address: LDS broadcast
#pragma unroll
                    for (int k = 0; k < 7; ++k) acc[i][k] = fmaf(dy, prow[k], acc[i][k]);
                }
            }
        }
    }
    const float sc = scale ? scale[co] : 1.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (i < nr) {
            const int r = r0 + i, c = r / 7, ky = r % 7;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const size_t e = ((size_t)(ky * 7 + k) * STEM_CO + co) * 3 + c;
                if (slab) slab[(size_t)blockIdx.x * (49 * STEM_CO * 3) + e] = acc[i][k] * sc;     // deterministic combine
                else atomicAdd(dw + e, acc[i][k] * sc);
            }
        }
    }
}

// ---- the same weight gradient on the matrix cores (bf16 image, bf16 dS): both operands are exact in bf16, so the
// products are exact in fp32 and only the summation order differs from the VALU kernel.
// GEMM dW[co][k] = sum_pixels dS[pixel][co] * P[pixel][k] with the forward's K order k = (c, ky) * 8 + kx (22 rows of 8,
// 176 columns, padded to 6 MFMA column tiles of 32). Per stage of 64 pixels (4 rows of the 16 x 16 tile) the workgroup
// builds the im2col tile P[64][176] in LDS -- one 16-byte copy per (pixel, row), the same contiguous 8-operand run the
// forward reads -- and stages dS[64][64]; both are pixel-major = K-major, so the MFMA operands (8 consecutive PIXELS
// per lane) come out of ds_read_b64_tr_b16 exactly as in conv_wgrad_kernel (csrc/conv.hip). Row pitches 144 / 400 B keep
// the four pixel rows of a transpose read in different banks. Wave w: channel half w & 1, column tiles w>>1, +2, +4.
// Persistent over the tiles, one round of fp32 atomics per workgroup at the end.
typedef short ss16x4 __attribute__((ext_vector_type(4)));
constexpr int SWM_DS_PITCH = 144;              // bytes per pixel row of the dS stage (64 channels + pad)
constexpr int SWM_P_PITCH = 400;               // bytes per pixel row of the im2col stage (192 columns + pad)

__global__ __launch_bounds__(256, 2) void stem_wgrad_mfma_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ ds,
                                                                 float* __restrict__ dw, const float* __restrict__ scale, int N,
                                                                 int H, int W, int Ho, int Wo, float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SF_P * SM_PITCH * 2 + 64 * SWM_DS_PITCH + 64 * SWM_P_PITCH];
    uint16_t* patch = reinterpret_cast<uint16_t*>(smem);                           // [3][37][40] bf16
    unsigned char* lds_ds = smem + 3 * SF_P * SM_PITCH * 2;                        // [64 px][144 B]
    unsigned char* lds_p = lds_ds + 64 * SWM_DS_PITCH;                             // [64 px][400 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wk = wave >> 1;
    const int g = lane >> 4, li = lane & 15;
    const int ch_in_tile = 16 * (g & 1) + 4 * (li & 3);     // column chunk this lane SUPPLIES to the transpose read
    const int pix_in_blk = 8 * (g >> 1) + (li >> 2);        // pixel row this lane supplies (within a 16-pixel K step)

    sf32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
    // patch pad columns 37..39 and the im2col columns 176..191 are read (into outputs that are never written): finite
    for (int i = tid; i < 3 * SF_P * 3; i += 256) patch[(i / 3) * SM_PITCH + SF_P + i % 3] = 0;
    for (int i = tid; i < 64 * 4; i += 256)
        *reinterpret_cast<uint2*>(lds_p + (i >> 2) * SWM_P_PITCH + 352 + (i & 3) * 8) = uint2{0u, 0u};

    const int tiles_x = (Wo + SF_T - 1) / SF_T, tiles_y = (Ho + SF_T - 1) / SF_T;
    const int ntiles = N * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int txi = t % tiles_x, tyi = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int ty0 = tyi * SF_T, tx0 = txi * SF_T;
        const int iy0 = ty0 * 2 - 3, ix0 = tx0 * 2 - 3;
        __syncthreads();                                   // the previous tile's patch reads are done
        for (int i = tid; i < 3 * SF_P * SF_P; i += 256) {
            const int px = i % SF_P, r = i / SF_P, py = r % SF_P, c = r / SF_P;
            const int iy = iy0 + py, ix = ix0 + px;
            uint16_t v = 0;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)n * 3 + c) * H + iy) * W + ix];
            patch[(c * SF_P + py) * SM_PITCH + px] = v;
        }
        for (int st = 0; st < 4; ++st) {                   // 4 stages of 64 pixels = tile rows 4 st .. 4 st + 3
            __syncthreads();                               // patch complete (st = 0) / previous stage's reads done
            // dS stage: 64 pixels x 8 chunks of 16 bytes (zero outside the image)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = i * 256 + tid, p = idx >> 3, ch = idx & 7;
                const int oy = ty0 + 4 * st + (p >> 4), ox = tx0 + (p & 15);
                su32x4 v = su32x4{0u, 0u, 0u, 0u};
                if (oy < Ho && ox < Wo) v = *reinterpret_cast<const su32x4*>(ds + (((size_t)n * Ho + oy) * Wo + ox) * STEM_CO + ch * 8);
                *reinterpret_cast<su32x4*>(lds_ds + p * SWM_DS_PITCH + ch * 16) = v;
            }
            // im2col stage: 64 pixels x 22 rows of 8 operands (row 21 = zeros)
            for (int idx = tid; idx < 64 * SM_ROWS; idx += 256) {
                const int p = idx / SM_ROWS, r = idx - p * SM_ROWS;
                su32x4 v = su32x4{0u, 0u, 0u, 0u};
                if (r < 21) {
                    const int c = r / 7, ky = r - c * 7;
                    const uint32_t* src = reinterpret_cast<const uint32_t*>(
                        patch + (c * SF_P + 2 * (4 * st + (p >> 4)) + ky) * SM_PITCH + 2 * (p & 15));
                    v = su32x4{src[0], src[1], src[2], src[3]};
                }
                *reinterpret_cast<su32x4*>(lds_p + p * SWM_P_PITCH + r * 16) = v;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {               // 16 pixels per MFMA
                const int pr0 = kk * 16 + pix_in_blk;
                const int cha = wc * 32 + ch_in_tile;
                const ss16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) ss16x4*)(lds_ds + pr0 * SWM_DS_PITCH + cha * 2));
                const ss16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) ss16x4*)(lds_ds + (pr0 + 4) * SWM_DS_PITCH + cha * 2));
                const uint2 al = __builtin_bit_cast(uint2, alo), ah = __builtin_bit_cast(uint2, ahi);
                const su32x4 fa = su32x4{al.x, al.y, ah.x, ah.y};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int chb = (wk + 2 * q) * 32 + ch_in_tile;
                    const ss16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) ss16x4*)(lds_p + pr0 * SWM_P_PITCH + chb * 2));
                    const ss16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) ss16x4*)(lds_p + (pr0 + 4) * SWM_P_PITCH + chb * 2));
                    const uint2 bl = __builtin_bit_cast(uint2, blo), bh = __builtin_bit_cast(uint2, bhi);
                    const su32x4 fb = su32x4{bl.x, bl.y, bh.x, bh.y};
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sbf16x8, fa), __builtin_bit_cast(sbf16x8, fb),
                                                                     acc[q], 0, 0, 0);
                }
            }
        }
    }
    // acc[q]: rows = channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of half wc, column = lane & 31 of column tile wk + 2 q.
    // The gradient tensor is [ky][kx][co][c]: a lane's elements are 12 bytes apart there, and an fp32 atomic is a
    // memory-side request per cache line touched (6.3 M requests per launch from 512 workgroups: 465 us, r02z kernel
    // stats). So the workgroup first lays its 9408 results out in LDS in the tensor's order, then adds them with
    // consecutive lanes on consecutive addresses (147 wave instructions of 256 bytes).
    __syncthreads();                                       // the stage buffers are free
    float* lds_out = reinterpret_cast<float*>(smem);       // [49][64][3] floats = 37.6 KB
    const int fcol = lane & 31, fhalf = lane >> 5;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int k = (wk + 2 * q) * 32 + fcol, r = k >> 3, kx = k & 7;
        if (r < 21 && kx < 7) {
            const int c = r / 7, ky = r - c * 7;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = wc * 32 + (i & 3) + 8 * (i >> 2) + 4 * fhalf;
                lds_out[((ky * 7 + kx) * STEM_CO + co) * 3 + c] = acc[q][i] * (scale ? scale[co] : 1.0f);
            }
        }
    }
    __syncthreads();
    if (slab) {                                            // deterministic combine: this block's partial sums, plain stores
        for (int i = tid; i < 49 * STEM_CO * 3; i += 256) slab[(size_t)blockIdx.x * (49 * STEM_CO * 3) + i] = lds_out[i];
    } else {
        for (int i = tid; i < 49 * STEM_CO * 3; i += 256) atomicAdd(dw + i, lds_out[i]);
    }
}

// dw[i] += sum over blocks of slab[b][i], in block order (the deterministic combine of the two kernels above)
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nblocks) {
    // 64 outputs per block, the slabs dealt over 4 thread groups (4 loads in flight each), fixed combine order
    __shared__ float part[3][64];
    constexpr int TOTAL = 49 * STEM_CO * 3;
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + j;
    float s = 0.0f;
    if (i < TOTAL) {
        for (int b = g; b < nblocks; b += 16) {
            const float v0 = slab[(size_t)b * TOTAL + i];
            const float v1 = b + 4 < nblocks ? slab[(size_t)(b + 4) * TOTAL + i] : 0.0f;
            const float v2 = b + 8 < nblocks ? slab[(size_t)(b + 8) * TOTAL + i] : 0.0f;
            const float v3 = b + 12 < nblocks ? slab[(size_t)(b + 12) * TOTAL + i] : 0.0f;
            s += (v0 + v1) + (v2 + v3);
        }
    }
    if (g > 0) part[g - 1][j] = s;
    __syncthreads();
    if (g == 0 && i < TOTAL) dw[i] += ((s + part[0][j]) + part[1][j]) + part[2][j];
}

// ---- gradient wrt the image: dx[n,c,iy,ix] = sum_{ky,kx,co} dS[n,(iy+3-ky)/2,(ix+3-kx)/2,co] * scale[co]-folded w
// wp2[(ky*7+kx)*3 + c][co] = w[ky][kx][co][c] * scale[co]
template <class TS>
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const TS* __restrict__ ds, const float* __restrict__ wp,
                                                         const float* __restrict__ scale, float* __restrict__ dx, int N,
                                                         int H, int W, int Ho, int Wo) {
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ix = (int)(i % W);
        size_t t = i / W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        for (int ky = (iy + 3) & 1; ky < 7; ky += 2) {
            const int oy = (iy + 3 - ky) >> 1;
            if (oy < 0 || oy >= Ho) continue;
            for (int kx = (ix + 3) & 1; kx < 7; kx += 2) {
                const int ox = (ix + 3 - kx) >> 1;
                if (ox < 0 || ox >= Wo) continue;
                const TS* row = ds + (((size_t)n * Ho + oy) * Wo + ox) * STEM_CO;
                const float* w0 = wp + (size_t)((0 * 7 + ky) * 7 + kx) * STEM_CO;
                const float* w1 = wp + (size_t)((1 * 7 + ky) * 7 + kx) * STEM_CO;
                const float* w2 = wp + (size_t)((2 * 7 + ky) * 7 + kx) * STEM_CO;
#pragma unroll 8
                for (int co = 0; co < STEM_CO; ++co) {
                    const float d = ld_f32(row + co) * scale[co];
                    a0 = fmaf(d, w0[co], a0);
                    a1 = fmaf(d, w1[co], a1);
                    a2 = fmaf(d, w2[co], a2);
                }
            }
        }
        dx[(((size_t)n * 3 + 0) * H + iy) * W + ix] = a0;
        dx[(((size_t)n * 3 + 1) * H + iy) * W + ix] = a1;
        dx[(((size_t)n * 3 + 2) * H + iy) * W + ix] = a2;
    }
}

}  // namespace cms

using namespace cms;

static bool dt_ok(int d) { return d == CMS_F32 || d == CMS_BF16; }

extern "C" int cms_stem_pack_weights(const void* w_khkwcoci, int w_dtype, float* w_packed, void* stream) {
    CMS_REQUIRE(w_khkwcoci && w_packed && dt_ok(w_dtype), "stem_pack_weights: NULL pointer / bad dtype");
    const int n = STEM_TAPS * STEM_CO;
    hipStream_t s = (hipStream_t)stream;
    if (w_dtype == CMS_F32)
        hipLaunchKernelGGL(stem_pack_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)w_khkwcoci, w_packed);
    else
        hipLaunchKernelGGL(stem_pack_kernel<uint16_t>, dim3((n + 255) / 256), dim3(256), 0, s, (const uint16_t*)w_khkwcoci,
                           w_packed);
    // bf16 (hi, lo) MFMA fragments behind the 147 x 64 floats (stem_fwd_mfma_kernel)
    uint16_t* frag = reinterpret_cast<uint16_t*>(w_packed + n);
    const int nf = STEM_CO * SM_ROWS * 8;
    if (w_dtype == CMS_F32)
        hipLaunchKernelGGL(stem_pack_frag_kernel<float>, dim3((nf + 255) / 256), dim3(256), 0, s, (const float*)w_khkwcoci, frag);
    else
        hipLaunchKernelGGL(stem_pack_frag_kernel<uint16_t>, dim3((nf + 255) / 256), dim3(256), 0, s, (const uint16_t*)w_khkwcoci,
                           frag);
    return launch_status("cms_stem_pack_weights");
}

static int pool_out(int s, int ceil_mode) {        // kernel 3, stride 2, padding 1 (ATen's rule for ceil_mode)
    if (!ceil_mode) return (s + 2 - 3) / 2 + 1;
    int o = (s + 2 - 3 + 1) / 2 + 1;
    if ((o - 1) * 2 >= s + 1) --o;
    return o;
}

extern "C" int cms_stem_out_hw(int h, int w, int* ho, int* wo, int* hp, int* wp) {
    CMS_REQUIRE(h > 0 && w > 0, "stem_out_hw: bad size");
    const int Ho = (h + 6 - 7) / 2 + 1, Wo = (w + 6 - 7) / 2 + 1;
    if (ho) *ho = Ho;
    if (wo) *wo = Wo;
    if (hp) *hp = pool_out(Ho, 1);            // (DeepLab v2's ceil-mode pool, deeplab2.py:146)
    if (wp) *wp = pool_out(Wo, 1);
    return CMS_OK;
}

extern "C" int cms_stem_fwd(const void* x_nchw, int x_dtype, void* y_nhwc, int y_dtype, const float* w_packed,
                            const float* scale, const float* bias, int n, int h, int w, void* stream) {
    CMS_REQUIRE(x_nchw && y_nhwc && w_packed && scale && bias, "stem_fwd: NULL pointer");
    CMS_REQUIRE(dt_ok(x_dtype) && dt_ok(y_dtype), "stem_fwd: bad dtype");
    CMS_REQUIRE(n > 0 && h > 0 && w > 0 && n <= 65535, "stem_fwd: bad geometry");
    int Ho, Wo;
    cms_stem_out_hw(h, w, &Ho, &Wo, nullptr, nullptr);
    const dim3 grid((Wo + SF_T - 1) / SF_T, (Ho + SF_T - 1) / SF_T, n);
    hipStream_t s = (hipStream_t)stream;
    static int env_mfma = -1;                   // CMS_STEM_MFMA=0: the VALU kernel also for bf16 (A/B switch)
    if (env_mfma < 0) {
        const char* e = getenv("CMS_STEM_MFMA");
        env_mfma = e ? atoi(e) : 1;
    }
    if (x_dtype == CMS_BF16 && y_dtype == CMS_BF16 && env_mfma != 0) {
        const int ntiles = (int)(grid.x * grid.y * grid.z);
        hipLaunchKernelGGL(stem_fwd_mfma_kernel, dim3(ntiles < 512 ? ntiles : 512), dim3(256), 0, s, (const uint16_t*)x_nchw,
                           (uint16_t*)y_nhwc, reinterpret_cast<const uint16_t*>(w_packed + STEM_TAPS * STEM_CO), scale, bias, n,
                           h, w, Ho, Wo);
        return launch_status("cms_stem_fwd");
    }
#define CMS_STEM_FWD(TX, TY) \
    hipLaunchKernelGGL((stem_fwd_kernel<TX, TY>), grid, dim3(256), 0, s, (const TX*)x_nchw, (TY*)y_nhwc, w_packed, scale, bias, n, h, w, Ho, Wo)
    if (x_dtype == CMS_F32 && y_dtype == CMS_F32) CMS_STEM_FWD(float, float);
    else if (x_dtype == CMS_F32) CMS_STEM_FWD(float, uint16_t);
    else if (y_dtype == CMS_F32) CMS_STEM_FWD(uint16_t, float);
    else CMS_STEM_FWD(uint16_t, uint16_t);
#undef CMS_STEM_FWD
    return launch_status("cms_stem_fwd");
}

extern "C" int cms_maxpool3x3s2_fwd(const void* s_nhwc, void* p_nhwc, uint8_t* argmax, int dtype, int n, int hs, int ws,
                                    int c, int ceil_mode, void* stream) {
    CMS_REQUIRE(s_nhwc && p_nhwc && argmax && dt_ok(dtype), "maxpool_fwd: NULL pointer / bad dtype");
    CMS_REQUIRE(n > 0 && hs > 0 && ws > 0 && c > 0 && c % 8 == 0, "maxpool_fwd: bad geometry (C %% 8 == 0)");
    const int hp = pool_out(hs, ceil_mode), wp = pool_out(ws, ceil_mode);
    const size_t total = (size_t)n * hp * wp * (c / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CMS_F32)
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, st, (const float*)s_nhwc,
                           (float*)p_nhwc, argmax, n, hs, ws, hp, wp, c);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<uint16_t>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, st,
                           (const uint16_t*)s_nhwc, (uint16_t*)p_nhwc, argmax, n, hs, ws, hp, wp, c);
    return launch_status("cms_maxpool3x3s2_fwd");
}

extern "C" int cms_maxpool3x3s2_relu_bwd(const void* dp_nhwc, const uint8_t* argmax, const void* s_nhwc, void* ds_nhwc,
                                         int dtype, int n, int hs, int ws, int c, int ceil_mode, void* stream) {
    CMS_REQUIRE(dp_nhwc && argmax && s_nhwc && ds_nhwc && dt_ok(dtype), "maxpool_bwd: NULL pointer / bad dtype");
    CMS_REQUIRE(n > 0 && hs > 0 && ws > 0 && c > 0 && c % 8 == 0, "maxpool_bwd: bad geometry (C %% 8 == 0)");
    const int hp = pool_out(hs, ceil_mode), wp = pool_out(ws, ceil_mode);
    const size_t total = (size_t)n * hs * ws * (c / 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == CMS_F32)
        hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, st, (const float*)dp_nhwc,
                           argmax, (const float*)s_nhwc, (float*)ds_nhwc, n, hs, ws, hp, wp, c);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<uint16_t>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, st,
                           (const uint16_t*)dp_nhwc, argmax, (const uint16_t*)s_nhwc, (uint16_t*)ds_nhwc, n, hs, ws, hp, wp, c);
    return launch_status("cms_maxpool3x3s2_relu_bwd");
}

static int stem_wgrad_blocks(int x_dtype, int ds_dtype, int n, int h, int w, bool* mfma_out) {
    int Ho, Wo;
    cms_stem_out_hw(h, w, &Ho, &Wo, nullptr, nullptr);
    static int env_mfma = -1;                   // CMS_STEM_MFMA=0: the VALU kernel also for bf16 (A/B switch)
    if (env_mfma < 0) {
        const char* e = getenv("CMS_STEM_MFMA");
        env_mfma = e ? atoi(e) : 1;
    }
    const bool mfma = x_dtype == CMS_BF16 && ds_dtype == CMS_BF16 && env_mfma != 0;
    if (mfma_out) *mfma_out = mfma;
    if (mfma) {
        const int nt = n * ((Ho + SF_T - 1) / SF_T) * ((Wo + SF_T - 1) / SF_T);
        return nt < 256 ? nt : 256;
    }
    const int ntiles = n * ((Ho + SW_TH - 1) / SW_TH) * ((Wo + SW_TW - 1) / SW_TW);
    return ntiles < 768 ? ntiles : 768;          // 3 blocks per CU (41 KB of LDS each), persistent over the tiles
}

// Bytes of scratch that make cms_stem_wgrad_ws deterministic: one [49][64][3] fp32 slab per (persistent) block.
extern "C" long long cms_stem_wgrad_workspace_bytes(int x_dtype, int ds_dtype, int n, int h, int w) {
    if (n <= 0 || h <= 0 || w <= 0 || !dt_ok(x_dtype) || !dt_ok(ds_dtype)) return 0;
    return (long long)stem_wgrad_blocks(x_dtype, ds_dtype, n, h, w, nullptr) * 49 * STEM_CO * 3 * (long long)sizeof(float);
}

// `workspace` (>= cms_stem_wgrad_workspace_bytes, or NULL): with it the blocks write their partial sums there with plain
// stores and a second launch adds them to dw in block order -- run-to-run deterministic; without it, fp32 atomics.
extern "C" int cms_stem_wgrad_ws(const void* x_nchw, int x_dtype, const void* ds_nhwc, int ds_dtype, float* dw_khkwcoci,
                                 const float* scale, int n, int h, int w, void* workspace, long long workspace_bytes,
                                 void* stream) {
    CMS_REQUIRE(x_nchw && ds_nhwc && dw_khkwcoci, "stem_wgrad: NULL pointer");
    CMS_REQUIRE(dt_ok(x_dtype) && dt_ok(ds_dtype), "stem_wgrad: bad dtype");
    CMS_REQUIRE(n > 0 && h > 0 && w > 0, "stem_wgrad: bad geometry");
    int Ho, Wo;
    cms_stem_out_hw(h, w, &Ho, &Wo, nullptr, nullptr);
    bool mfma = false;
    const int nblk = stem_wgrad_blocks(x_dtype, ds_dtype, n, h, w, &mfma);
    const dim3 grid(nblk);
    hipStream_t s = (hipStream_t)stream;
    CMS_REQUIRE(workspace == nullptr || workspace_bytes >= cms_stem_wgrad_workspace_bytes(x_dtype, ds_dtype, n, h, w),
                "stem_wgrad: workspace of %lld bytes is too small", workspace_bytes);
    float* slab = (float*)workspace;
    if (mfma) {
        hipLaunchKernelGGL(stem_wgrad_mfma_kernel, grid, dim3(256), 0, s, (const uint16_t*)x_nchw,
                           (const uint16_t*)ds_nhwc, dw_khkwcoci, scale, n, h, w, Ho, Wo, slab);
    } else {
#define CMS_STEM_WG(TX, TS) \
    hipLaunchKernelGGL((stem_wgrad_kernel<TX, TS>), grid, dim3(256), 0, s, (const TX*)x_nchw, (const TS*)ds_nhwc, dw_khkwcoci, scale, n, h, w, Ho, Wo, slab)
        if (x_dtype == CMS_F32 && ds_dtype == CMS_F32) CMS_STEM_WG(float, float);
        else if (x_dtype == CMS_F32) CMS_STEM_WG(float, uint16_t);
        else if (ds_dtype == CMS_F32) CMS_STEM_WG(uint16_t, float);
        else CMS_STEM_WG(uint16_t, uint16_t);
#undef CMS_STEM_WG
    }
    if (slab)
        hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((49 * STEM_CO * 3 + 63) / 64), dim3(256), 0, s, slab, dw_khkwcoci, nblk);
    return launch_status("cms_stem_wgrad");
}

extern "C" int cms_stem_wgrad(const void* x_nchw, int x_dtype, const void* ds_nhwc, int ds_dtype, float* dw_khkwcoci,
                              const float* scale, int n, int h, int w, void* stream) {
    return cms_stem_wgrad_ws(x_nchw, x_dtype, ds_nhwc, ds_dtype, dw_khkwcoci, scale, n, h, w, nullptr, 0, stream);
}

extern "C" int cms_stem_dgrad(const void* ds_nhwc, int ds_dtype, const float* w_packed, const float* scale, float* dx_nchw,
                              int n, int h, int w, void* stream) {
    CMS_REQUIRE(ds_nhwc && w_packed && scale && dx_nchw && dt_ok(ds_dtype), "stem_dgrad: NULL pointer / bad dtype");
    CMS_REQUIRE(n > 0 && h > 0 && w > 0, "stem_dgrad: bad geometry");
    int Ho, Wo;
    cms_stem_out_hw(h, w, &Ho, &Wo, nullptr, nullptr);
    const size_t total = (size_t)n * h * w;
    hipStream_t s = (hipStream_t)stream;
    if (ds_dtype == CMS_F32)
        hipLaunchKernelGGL(stem_dgrad_kernel<float>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, (const float*)ds_nhwc,
                           w_packed, scale, dx_nchw, n, h, w, Ho, Wo);
    else
        hipLaunchKernelGGL(stem_dgrad_kernel<uint16_t>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s,
                           (const uint16_t*)ds_nhwc, w_packed, scale, dx_nchw, n, h, w, Ho, Wo);
    return launch_status("cms_stem_dgrad");
}
