// Device-side input staging: crop (+ random scale), flip, colour augmentation, standardisation -- the per-sample CPU work
// of the reference's loader workers (datapipe/seg_transforms_cv.py:29-133 pad + crop, :169-231 random-scale crop,
// :452-497 flips, :541-585 torchvision ColorJitter / RandomGrayscale through PIL, :587-623 standardise + NCHW;
// wiring: train_seg_semisup_mask_mt.py:150-183) as ONE gather kernel over uint8 source images that already sit in HBM.
//
// Per output pixel: undo the flips, map into the source window (bilinear, cv2.INTER_LINEAR's half-pixel convention;
// nearest = floor for labels, cv2.INTER_NEAREST), zero / 255 / 0 outside the source image (the reference pads with an alpha
// channel so that padding is exactly 0 after standardisation, :46-52, 600-608), colour operations on the interpolated RGB,
// standardise, write NCHW. The paired layout of the unsupervised stream (SegTransformToPair + colour on sample 1 only)
// comes out of the same pass: `out0` = weakly augmented (teacher), `out1` = colour-augmented (student), same geometry.
//
// HBM-bound: reads <= 4 source pixels x 3 bytes per output pixel (L2-local), writes 3 * s bytes per output (x2 when
// paired). Random parameters are drawn on the host in the reference's order (device_pipeline.py) and arrive as a small
// table; nothing else crosses PCIe.
#include "common.hpp"

namespace cms {

struct AugArgs {
    const uint8_t* src;         // [N][Hs][Ws][3]
    const uint8_t* src_labels;  // [N][Hs][Ws] or NULL
    void* out0;                 // (N,3,H,W) or NULL
    void* out1;                 // (N,3,H,W) colour-augmented copy or NULL
    uint8_t* out_labels;        // (N,H,W) or NULL
    float* out_mask;            // (N,1,H,W) or NULL
    const float* params;        // [N][CMS_AUG_PARAMS]
    float mean[3], inv_std[3];
    int N, Hs, Ws, H, W;
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float dh) {
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float d = mx - mn;
    float h = 0.0f;
    if (d > 0.0f) {
        if (mx == r) h = (g - b) / d;
        else if (mx == g) h = 2.0f + (b - r) / d;
        else h = 4.0f + (r - g) / d;
        h *= (1.0f / 6.0f);
        if (h < 0.0f) h += 1.0f;
    }
    const float s = mx > 0.0f ? d / mx : 0.0f, v = mx;
    h += dh;
    h -= floorf(h);
    const float hf = h * 6.0f;
    const int i = (int)hf % 6;
    const float f = hf - floorf(hf);
    const float p = v * (1.0f - s), q = v * (1.0f - s * f), t = v * (1.0f - s * (1.0f - f));
    switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
    }
}

template <class T>
__device__ __forceinline__ void put(void* base, size_t idx, float v) {
    if constexpr (sizeof(T) == 4) reinterpret_cast<float*>(base)[idx] = v;
    else reinterpret_cast<uint16_t*>(base)[idx] = f32_to_bf16(v);
}

// cv2.BORDER_REFLECT_101: ... 2 1 | 0 1 2 ... n-1 | n-2 n-3 ... (period 2n - 2), any distance outside
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * n - 2;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - i;
}

// AFFINE WARP geometry (params slot 15 == 1): source position of (flip-undone) output pixel (cx, cy)
__device__ __forceinline__ void warp_src(const float* p, int cx, int cy, float& sx, float& sy) {
    sx = fmaf(p[16], (float)cx, fmaf(p[17], (float)cy, p[18]));
    sy = fmaf(p[19], (float)cx, fmaf(p[20], (float)cy, p[21]));
}

// The geometric half of the transform for ONE (flip-undone) output pixel (cx, cy) of sample `img`: interpolated source colour
// (0..255), validity weight `alpha`, `img_alpha` = factor of the mean in the standardisation (window mode: zero padding),
// (ny, nx) = nearest source pixel for the labels. Shared by the image kernel and the luminance pre-pass, so that the contrast
// pivot is the mean of exactly the pixels the image kernel produces (same taps, same weights).
__device__ __forceinline__ void sample_source(const AugArgs& a, const float* p, const uint8_t* img, int cx, int cy, float (&rgb)[3],
                                              float& alpha, float& img_alpha, int& ny, int& nx) {
    const float y0 = p[0], x0 = p[1], sh = p[2], sw = p[3];
    const bool warp = p[15] != 0.0f;
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    alpha = 0.0f;
    img_alpha = 1.0f;
    ny = nx = 0;
    if (warp) {
        // datapipe/seg_transforms_cv.py:344-362: cv2.warpAffine(image, local_xf, crop, flags=interp, BORDER_REFLECT_101),
        // labels INTER_NEAREST / constant 255, mask constant 0
        float sx, sy;
        warp_src(p, cx, cy, sx, sy);
        nx = (int)floorf(sx + 0.5f);
        ny = (int)floorf(sy + 0.5f);
        if (p[22] == 0.0f) {
            const uint8_t* q = img + ((size_t)reflect101(ny, a.Hs) * a.Ws + reflect101(nx, a.Ws)) * 3;
            rgb[0] = (float)q[0]; rgb[1] = (float)q[1]; rgb[2] = (float)q[2];
            alpha = ((unsigned)ny < (unsigned)a.Hs && (unsigned)nx < (unsigned)a.Ws) ? 1.0f : 0.0f;
        } else {
            const int ix0 = (int)floorf(sx), iy0 = (int)floorf(sy);
            const float wx = sx - (float)ix0, wy = sy - (float)iy0;
            auto wtap = [&](int Y, int X, float w) {
                const uint8_t* q = img + ((size_t)reflect101(Y, a.Hs) * a.Ws + reflect101(X, a.Ws)) * 3;
                rgb[0] += w * (float)q[0];
                rgb[1] += w * (float)q[1];
                rgb[2] += w * (float)q[2];
                if ((unsigned)Y < (unsigned)a.Hs && (unsigned)X < (unsigned)a.Ws) alpha += w;
            };
            wtap(iy0, ix0, (1.0f - wy) * (1.0f - wx));
            wtap(iy0, ix0 + 1, (1.0f - wy) * wx);
            wtap(iy0 + 1, ix0, wy * (1.0f - wx));
            wtap(iy0 + 1, ix0 + 1, wy * wx);
        }
    } else {
        // bilinear tap positions inside the source window (cv2.INTER_LINEAR: half-pixel centres, border replicated)
        float fy = ((float)cy + 0.5f) * (sh / (float)a.H) - 0.5f, fx = ((float)cx + 0.5f) * (sw / (float)a.W) - 0.5f;
        fy = fminf(fmaxf(fy, 0.0f), sh - 1.0f);
        fx = fminf(fmaxf(fx, 0.0f), sw - 1.0f);
        const int iy0 = (int)floorf(fy), ix0 = (int)floorf(fx);
        const float wy = fy - (float)iy0, wx = fx - (float)ix0;
        const int iy1 = min(iy0 + 1, (int)sh - 1), ix1 = min(ix0 + 1, (int)sw - 1);
        const int Y0 = iy0 + (int)y0, Y1 = iy1 + (int)y0, X0 = ix0 + (int)x0, X1 = ix1 + (int)x0;
        auto tap = [&](int Y, int X, float w) {
            if (w != 0.0f && (unsigned)Y < (unsigned)a.Hs && (unsigned)X < (unsigned)a.Ws) {
                const uint8_t* q = img + ((size_t)Y * a.Ws + X) * 3;
                rgb[0] += w * (float)q[0];
                rgb[1] += w * (float)q[1];
                rgb[2] += w * (float)q[2];
                alpha += w;
            }
        };
        tap(Y0, X0, (1.0f - wy) * (1.0f - wx));
        tap(Y0, X1, (1.0f - wy) * wx);
        tap(Y1, X0, wy * (1.0f - wx));
        tap(Y1, X1, wy * wx);
        img_alpha = alpha;
        // cv2.INTER_NEAREST: floor(dst * scale)
        ny = min((int)((float)cy * (sh / (float)a.H)), (int)sh - 1) + (int)y0;
        nx = min((int)((float)cx * (sw / (float)a.W)), (int)sw - 1) + (int)x0;
    }
}

template <class T>
__global__ __launch_bounds__(256) void augment_kernel(AugArgs a) {
    const size_t total = (size_t)a.N * a.H * a.W;
    const size_t plane = (size_t)a.H * a.W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % a.W);
        const size_t t0 = i / a.W;
        const int oy = (int)(t0 % a.H);
        const int n = (int)(t0 / a.H);
        const float* p = a.params + (size_t)n * CMS_AUG_PARAMS;
        // undo the flips (applied after the crop in the reference: x flip, y flip, transpose)
        int cy = oy, cx = ox;
        if (p[6] != 0.0f) { const int t = cy; cy = cx; cx = t; }
        if (p[5] != 0.0f) cy = a.H - 1 - cy;
        if (p[4] != 0.0f) cx = a.W - 1 - cx;
        float rgb[3];
        float alpha, img_alpha;
        int ny, nx;
        sample_source(a, p, a.src + (size_t)n * a.Hs * a.Ws * 3, cx, cy, rgb, alpha, img_alpha, ny, nx);
        float r = rgb[0] * (1.0f / 255.0f), g = rgb[1] * (1.0f / 255.0f), b = rgb[2] * (1.0f / 255.0f);
        const size_t o = (size_t)n * 3 * plane + (size_t)oy * a.W + ox;
        if (a.out0) {
            put<T>(a.out0, o, (r - a.mean[0] * img_alpha) * a.inv_std[0]);
            put<T>(a.out0, o + plane, (g - a.mean[1] * img_alpha) * a.inv_std[1]);
            put<T>(a.out0, o + 2 * plane, (b - a.mean[2] * img_alpha) * a.inv_std[2]);
        }
        if (a.out1) {
            if (p[12] != 0.0f) {                       // ColorJitter applied (RandomApply, p = aug_colour_prob)
                const int order = (int)p[13];          // permutation index of (brightness, contrast, saturation, hue)
                // decode the permutation: order = ((i0 * 4 + i1) * 4 + i2) * 4 + i3
                const int ops[4] = {(order >> 6) & 3, (order >> 4) & 3, (order >> 2) & 3, order & 3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    switch (ops[k]) {
                    case 0: r = clamp01(r * p[7]); g = clamp01(g * p[7]); b = clamp01(b * p[7]); break;
                    case 1: {
                        const float m = p[14];          // mean luminance at the time contrast is applied (pre-pass)
                        r = clamp01((r - m) * p[8] + m); g = clamp01((g - m) * p[8] + m); b = clamp01((b - m) * p[8] + m);
                        break;
                    }
                    case 2: {
                        const float gr = gray_of(r, g, b);
                        r = clamp01((r - gr) * p[9] + gr); g = clamp01((g - gr) * p[9] + gr); b = clamp01((b - gr) * p[9] + gr);
                        break;
                    }
                    default: if (p[10] != 0.0f) hue_shift(r, g, b, p[10]); break;
                    }
                }
            }
            if (p[11] != 0.0f) {                       // RandomGrayscale
                const float gr = gray_of(r, g, b);
                r = g = b = gr;
            }
            put<T>(a.out1, o, (r - a.mean[0] * img_alpha) * a.inv_std[0]);
            put<T>(a.out1, o + plane, (g - a.mean[1] * img_alpha) * a.inv_std[1]);
            put<T>(a.out1, o + 2 * plane, (b - a.mean[2] * img_alpha) * a.inv_std[2]);
        }
        if (a.out_mask) a.out_mask[(size_t)n * plane + (size_t)oy * a.W + ox] = alpha;
        if (a.out_labels) {
            uint8_t lab = 255;
            if (a.src_labels && (unsigned)ny < (unsigned)a.Hs && (unsigned)nx < (unsigned)a.Ws)
                lab = a.src_labels[((size_t)n * a.Hs + ny) * a.Ws + nx];
            a.out_labels[(size_t)n * plane + (size_t)oy * a.W + ox] = lab;
        }
    }
}

// mean luminance of the geometrically transformed image (the pivot of ColorJitter's contrast), one block per sample
__global__ __launch_bounds__(256) void augment_luma_kernel(AugArgs a, float* __restrict__ luma) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    const float* p = a.params + (size_t)n * CMS_AUG_PARAMS;
    const uint8_t* img = a.src + (size_t)n * a.Hs * a.Ws * 3;
    float acc = 0.0f;
    for (int i = threadIdx.x; i < a.H * a.W; i += blockDim.x) {
        const int cy = i / a.W, cx = i % a.W;           // (flips do not change the mean)
        float rgb[3], alpha, img_alpha;
        int ny, nx;
        sample_source(a, p, img, cx, cy, rgb, alpha, img_alpha, ny, nx);      // the image kernel's own taps and weights
        acc += gray_of(rgb[0], rgb[1], rgb[2]) * (1.0f / 255.0f);
    }
    float v[1] = {acc};
    block_sum<1>(v, red);
    if (threadIdx.x == 0) luma[n] = v[0] / (float)(a.H * a.W);
}

}  // namespace cms

using namespace cms;

static int aug_fill(AugArgs& a, const cms_augment_desc* d) {
    CMS_REQUIRE(d && d->src && d->params && (d->out0 || d->out1), "augment: NULL pointer");
    CMS_REQUIRE(d->n > 0 && d->hs > 0 && d->ws > 0 && d->h > 0 && d->w > 0, "augment: bad geometry");
    CMS_REQUIRE(d->out_dtype == CMS_F32 || d->out_dtype == CMS_BF16, "augment: bad output dtype");
    CMS_REQUIRE(d->std_[0] > 0 && d->std_[1] > 0 && d->std_[2] > 0, "augment: std must be positive");
    a.src = d->src; a.src_labels = d->src_labels; a.out0 = d->out0; a.out1 = d->out1; a.out_labels = d->out_labels;
    a.out_mask = d->out_mask; a.params = d->params;
    for (int i = 0; i < 3; ++i) { a.mean[i] = d->mean[i]; a.inv_std[i] = 1.0f / d->std_[i]; }
    a.N = d->n; a.Hs = d->hs; a.Ws = d->ws; a.H = d->h; a.W = d->w;
    return CMS_OK;
}

extern "C" int cms_augment_batch(const cms_augment_desc* d, void* stream) {
    AugArgs a;
    const int rc = aug_fill(a, d);
    if (rc) return rc;
    const size_t total = (size_t)a.N * a.H * a.W;
    hipStream_t s = (hipStream_t)stream;
    if (d->out_dtype == CMS_F32) hipLaunchKernelGGL(augment_kernel<float>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(augment_kernel<uint16_t>, dim3(grid_for(total, 256, 256 * 16)), dim3(256), 0, s, a);
    return launch_status("cms_augment_batch");
}

extern "C" int cms_augment_luma(const cms_augment_desc* d, float* luma, void* stream) {
    AugArgs a;
    CMS_REQUIRE(luma != nullptr, "augment_luma: NULL pointer");
    cms_augment_desc dd = *d;
    if (!dd.out0 && !dd.out1) dd.out0 = (void*)luma;      // (geometry check only)
    const int rc = aug_fill(a, &dd);
    if (rc) return rc;
    hipLaunchKernelGGL(augment_luma_kernel, dim3(a.N), dim3(256), 0, (hipStream_t)stream, a, luma);
    return launch_status("cms_augment_luma");
}
