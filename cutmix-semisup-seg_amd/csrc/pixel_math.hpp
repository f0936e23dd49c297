// Per-pixel arithmetic of the CutMix mean-teacher loss path, shared by every kernel in losses.hip / eval.hip /
// upsample.hip. Everything here is `__host__ __device__` so that the exact same code can be driven on the host by
// tests/hostcheck (CPU-only check of the formulas against the oracle before any GPU time is spent); the product
// never runs it on the host.
//
// Reference behaviour restated here (paths relative to the upstream repository):
//   bilinear taps        torch F.interpolate(mode='bilinear') as called at architectures/deeplab2.py:204
//                        (align_corners=True) and architectures/deeplab3plus.py:54-55,77 (align_corners=False)
//   box membership       mask_gen.py:110-116 (boxes XOR into a zeros/ones canvas)
//   softmax/confidence   train_seg_semisup_mask_mt.py:366-367, 407-418
//   consistency losses   train_seg_semisup_mask_mt.py:428-446; robust_binary_crossentropy at
//                        architectures/network_architectures.py:115-118
//   cross entropy        nn.CrossEntropyLoss(ignore_index=255), train_seg_semisup_mask_mt.py:126,300
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CMS_HD __host__ __device__ __forceinline__
#else
#define CMS_HD inline
#endif

namespace cms {

enum LossFn : int { LOSS_VAR = 0, LOSS_LOGITS_VAR = 1, LOSS_LOGITS_SMOOTHL1 = 2, LOSS_BCE = 3, LOSS_KLD = 4 };
enum MaskMode : int { MODE_MIX = 0, MODE_CUT = 1 };

// ---------------------------------------------------------------------------------------------- bilinear taps
struct Tap {
    int i0, i1;
    float w0, w1;
};

CMS_HD float bilin_scale(int in_size, int out_size, bool align_corners) {
    if (align_corners) return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
    return (float)in_size / (float)out_size;
}

CMS_HD Tap bilin_tap(int dst, float scale, int in_size, bool align_corners) {
    float src;
    if (align_corners) {
        src = scale * (float)dst;
    } else {
        src = scale * ((float)dst + 0.5f) - 0.5f;
        if (src < 0.0f) src = 0.0f;
    }
    int i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    Tap t;
    t.i0 = i0;
    t.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    float l1 = src - (float)i0;
    l1 = l1 < 0.0f ? 0.0f : (l1 > 1.0f ? 1.0f : l1);
    t.w1 = l1;
    t.w0 = 1.0f - l1;
    return t;
}

// value of the upsampled map at one output pixel: wy0*(wx0*v00 + wx1*v01) + wy1*(wx0*v10 + wx1*v11)
CMS_HD float bilin_gather(const float* plane, int w_in, const Tap& ty, const Tap& tx) {
    const float* r0 = plane + (size_t)ty.i0 * w_in;
    const float* r1 = plane + (size_t)ty.i1 * w_in;
    float a = tx.w0 * r0[tx.i0] + tx.w1 * r0[tx.i1];
    float b = tx.w0 * r1[tx.i0] + tx.w1 * r1[tx.i1];
    return ty.w0 * a + ty.w1 * b;
}

// ---------------------------------------------------------------------------------------------- box membership
// ranges: nb x [y0, y1, x0, x1] half-open (numpy slice semantics already applied on the host).
CMS_HD bool box_mask_bit(const int32_t* ranges, int nb, int y, int x, bool invert) {
    bool parity = false;
    for (int b = 0; b < nb; ++b) {
        const int32_t* r = ranges + 4 * b;
        parity ^= (y >= r[0]) & (y < r[1]) & (x >= r[2]) & (x < r[3]);
    }
    return invert ? parity : !parity;
}

// ---------------------------------------------------------------------------------------------- softmax helpers
// `L` is any callable int -> float giving the logit of class c. With CT > 0 the class count is a compile-time
// constant and callers back `L` with a register array; with CT == 0 it is the run-time `crt` and `L` may re-gather.
template <int CT, class L>
CMS_HD void softmax_stats(L l, int crt, float& mx, float& z) {
    const int C = CT > 0 ? CT : crt;
    mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, l(c));
    z = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) z += expf(l(c) - mx);
}

struct PixelFwd {
    float loss;   // per-pixel consistency value, already summed over classes (and / sqrt(C) where applicable)
    float conf;   // max_c softmax(teacher)_c
};

CMS_HD float smooth_l1(float d) {
    float a = fabsf(d);
    return a < 1.0f ? 0.5f * a * a : a - 0.5f;
}

// forward: per-pixel loss + teacher confidence
template <int CT, class LS, class LT>
CMS_HD PixelFwd consistency_pixel_fwd(LS ls, LT lt, int crt, int loss_fn, float inv_root_c) {
    const int C = CT > 0 ? CT : crt;
    float ms, zs, mt, zt;
    softmax_stats<CT>(ls, crt, ms, zs);
    softmax_stats<CT>(lt, crt, mt, zt);
    PixelFwd out;
    out.conf = 1.0f / zt;
    float acc = 0.0f;
    const float log_zs = logf(zs);
    const float log_zt = logf(zt);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float a = ls(c), b = lt(c);
        if (loss_fn == LOSS_VAR) {
            float d = expf(a - ms) / zs - expf(b - mt) / zt;
            acc += d * d;
        } else if (loss_fn == LOSS_LOGITS_VAR) {
            float d = a - b;
            acc += d * d;
        } else if (loss_fn == LOSS_LOGITS_SMOOTHL1) {
            acc += smooth_l1(a - b);
        } else if (loss_fn == LOSS_BCE) {
            const float eps = 1e-6f;
            float p = expf(a - ms) / zs, t = expf(b - mt) / zt;
            acc += -(t * logf(p + eps) + (1.0f - t) * logf(1.0f - p + eps));
        } else {  // LOSS_KLD: t*(log t - log_softmax(ls)); 0 where t == 0
            float t = expf(b - mt) / zt;
            float logp = (a - ms) - log_zs;
            float logt = (b - mt) - log_zt;
            acc += t > 0.0f ? t * (logt - logp) : 0.0f;
        }
    }
    if (loss_fn == LOSS_LOGITS_VAR || loss_fn == LOSS_LOGITS_SMOOTHL1) acc *= inv_root_c;
    out.loss = acc;
    return out;
}

// backward: d(per-pixel loss)/d(student logit k) for every k, emitted through `emit(k, value)`; also returns conf
template <int CT, class LS, class LT, class E>
CMS_HD float consistency_pixel_bwd(LS ls, LT lt, int crt, int loss_fn, float inv_root_c, E emit) {
    const int C = CT > 0 ? CT : crt;
    float ms, zs, mt, zt;
    softmax_stats<CT>(ls, crt, ms, zs);
    softmax_stats<CT>(lt, crt, mt, zt);
    const float conf = 1.0f / zt;
    if (loss_fn == LOSS_LOGITS_VAR) {
#pragma unroll
        for (int k = 0; k < C; ++k) emit(k, 2.0f * (ls(k) - lt(k)) * inv_root_c);
        return conf;
    }
    if (loss_fn == LOSS_LOGITS_SMOOTHL1) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            float d = ls(k) - lt(k);
            float g = fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
            emit(k, g * inv_root_c);
        }
        return conf;
    }
    // softmax-based losses: loss = sum_c f(p_c, t_c);  dl_k = p_k * (f'_k - sum_c f'_c p_c)
    float dot = 0.0f;
    float tsum = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float p = expf(ls(c) - ms) / zs, t = expf(lt(c) - mt) / zt;
        float fp;
        if (loss_fn == LOSS_VAR) {
            fp = 2.0f * (p - t);
        } else if (loss_fn == LOSS_BCE) {
            const float eps = 1e-6f;
            fp = -t / (p + eps) + (1.0f - t) / (1.0f - p + eps);
        } else {
            fp = 0.0f;
        }
        dot += fp * p;
        tsum += t;
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        float p = expf(ls(k) - ms) / zs, t = expf(lt(k) - mt) / zt;
        float g;
        if (loss_fn == LOSS_VAR) {
            g = p * (2.0f * (p - t) - dot);
        } else if (loss_fn == LOSS_BCE) {
            const float eps = 1e-6f;
            g = p * ((-t / (p + eps) + (1.0f - t) / (1.0f - p + eps)) - dot);
        } else {  // KLD: -t_k + p_k * sum_c t_c
            g = p * tsum - t;
        }
        emit(k, g);
    }
    return conf;
}

// ---------------------------------------------------------------------------------------------- cross entropy
// returns -log_softmax(l)[label]
template <int CT, class L>
CMS_HD float ce_pixel_fwd(L l, int crt, int label) {
    float mx, z;
    softmax_stats<CT>(l, crt, mx, z);
    return -((l(label) - mx) - logf(z));
}

template <int CT, class L, class E>
CMS_HD void ce_pixel_bwd(L l, int crt, int label, E emit) {
    const int C = CT > 0 ? CT : crt;
    float mx, z;
    softmax_stats<CT>(l, crt, mx, z);
#pragma unroll
    for (int k = 0; k < C; ++k) emit(k, expf(l(k) - mx) / z - (k == label ? 1.0f : 0.0f));
}

// ---------------------------------------------------------------------------------------------- EMA (3 roundings)
// optim_weight_ema.py:23-25: t.mul_(alpha); t.add_(s * (1 - alpha)) -- two products and one sum, each rounded.
CMS_HD float ema_update(float t, float s, float alpha, float one_minus_alpha) {
#if defined(__clang__)
#pragma clang fp contract(off)
    // HIP's __fmul_rn / __fadd_rn are plain operators that the optimiser may fuse into an FMA; contraction must be
    // off for the three roundings of the reference to survive
    const float a = t * alpha;
    const float b = s * one_minus_alpha;
    return a + b;
#else
    volatile float a = t * alpha;
    volatile float b = s * one_minus_alpha;
    return a + b;
#endif
}

}  // namespace cms
