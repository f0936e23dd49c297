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
    // Explicit roundings (one product, then fused multiply-adds in a fixed order): left to the compiler's contraction, two
    // inlined instances of this expression can be fused differently -- the student's and the teacher's gathers of the SAME
    // logits then differ in the last bit and "identical distributions -> zero loss" no longer holds exactly
    // (tests/test_gpu_parity.py::test_consistency_properties_full_size caught it when the surrounding code changed).
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float* r0 = plane + (size_t)ty.i0 * w_in;
    const float* r1 = plane + (size_t)ty.i1 * w_in;
    const float a = fmaf(tx.w1, r0[tx.i1], tx.w0 * r0[tx.i0]);
    const float b = fmaf(tx.w1, r1[tx.i1], tx.w0 * r1[tx.i0]);
    return fmaf(ty.w1, b, ty.w0 * a);
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

// Softmax of a logit vector with the exponentials kept (round 4): the loss formulas below need exp(l_c - max) of every class
// once for the normaliser and once more per class -- with a compile-time class count the 2 x C exponentials of a pixel live in
// registers and the C divisions by the normaliser become multiplications with ONE reciprocal (the consistency forward spent
// most of its 160 us in 84 expf + 42 divisions per pixel). CT == 0 (run-time class count) keeps the re-evaluating form.
template <int CT>
struct SoftmaxRegs {
    float e[CT > 0 ? CT : 1];
    float mx, z, rz;
};

template <int CT, class L>
CMS_HD void softmax_regs(L l, int crt, SoftmaxRegs<CT>& s) {
    // (no contraction: the statistics of two identical logit vectors must come out identical whatever surrounds the call)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (CT > 0) {
        s.mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CT; ++c) s.mx = fmaxf(s.mx, l(c));
        s.z = 0.0f;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            s.e[c] = expf(l(c) - s.mx);
            s.z += s.e[c];
        }
    } else {
        softmax_stats<CT>(l, crt, s.mx, s.z);
    }
    s.rz = 1.0f / s.z;
}

// probability of class c
template <int CT, class L>
CMS_HD float softmax_prob(const SoftmaxRegs<CT>& s, L l, int c) {
    // a ROUNDED product: contracted into the caller's subtraction (fma(e_s, rz_s, -t)) the difference of two identical
    // distributions would no longer be exactly zero (tests/test_gpu_parity.py::test_consistency_properties_full_size)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float e = CT > 0 ? s.e[CT > 0 ? c : 0] : expf(l(c) - s.mx);
    const float p = e * s.rz;
    return p;
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
    SoftmaxRegs<CT> ss, st;
    softmax_regs<CT>(ls, crt, ss);
    softmax_regs<CT>(lt, crt, st);
    PixelFwd out;
    out.conf = st.rz;
    float acc = 0.0f;
    if (loss_fn == LOSS_VAR) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = softmax_prob<CT>(ss, ls, c) - softmax_prob<CT>(st, lt, c);
            acc += d * d;
        }
    } else if (loss_fn == LOSS_LOGITS_VAR) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float d = ls(c) - lt(c);
            acc += d * d;
        }
    } else if (loss_fn == LOSS_LOGITS_SMOOTHL1) {
#pragma unroll
        for (int c = 0; c < C; ++c) acc += smooth_l1(ls(c) - lt(c));
    } else if (loss_fn == LOSS_BCE) {
        const float eps = 1e-6f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float p = softmax_prob<CT>(ss, ls, c), t = softmax_prob<CT>(st, lt, c);
            acc += -(t * logf(p + eps) + (1.0f - t) * logf(1.0f - p + eps));
        }
    } else {  // LOSS_KLD: t*(log t - log_softmax(ls)); 0 where t == 0
        const float log_zs = logf(ss.z);
        const float log_zt = logf(st.z);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float t = softmax_prob<CT>(st, lt, c);
            const float logp = (ls(c) - ss.mx) - log_zs;
            const float logt = (lt(c) - st.mx) - log_zt;
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
    SoftmaxRegs<CT> ss, st;
    softmax_regs<CT>(lt, crt, st);
    const float conf = st.rz;
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
    softmax_regs<CT>(ls, crt, ss);
    // softmax-based losses: loss = sum_c f(p_c, t_c);  dl_k = p_k * (f'_k - sum_c f'_c p_c)
    float dot = 0.0f;
    float tsum = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float p = softmax_prob<CT>(ss, ls, c), t = softmax_prob<CT>(st, lt, c);
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
        const float p = softmax_prob<CT>(ss, ls, k), t = softmax_prob<CT>(st, lt, k);
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

// forward AND backward of one pixel from ONE pair of softmaxes (round 6: the one-launch loss kernels; compile-time class count).
// Every formula is the one of consistency_pixel_fwd / consistency_pixel_bwd, evaluated on the same e[] / rz registers in the same
// order, so loss, confidence and gradient come out bit for bit as from the two functions -- what is saved is the second pair of
// softmaxes (2 C expf, two reciprocals) and, in the caller, the second gather of both logit vectors.
template <int CT, int LF, class LS, class LT, class F, class E>
CMS_HD PixelFwd consistency_pixel_fwd_bwd(LS ls, LT lt, int loss_fn_rt, float inv_root_c, F factor_of_conf, E emit) {
    // LF >= 0: the loss function as a compile-time constant (the default `var` gets its own instantiation: with a run-time switch
    // the register allocation is that of the hungriest branch -- KLD keeps both logit vectors alive beside both probability
    // vectors: 176 VGPRs, two waves per SIMD); LF < 0: `loss_fn_rt` decides
    const int loss_fn = LF >= 0 ? LF : loss_fn_rt;
    // `factor_of_conf(conf)` -> the pixel's gradient factor (known as soon as the teacher's softmax is); `emit(k, factor * g_k)`
    // goes straight to its destination: no gradient vector is kept (register budget: 128 for four waves per SIMD)
    static_assert(CT > 0, "compile-time class count only (the generic path calls the two functions)");
    SoftmaxRegs<CT> ss, st;
    softmax_regs<CT>(lt, CT, st);
    PixelFwd out;
    out.conf = st.rz;
    const float f = factor_of_conf(out.conf);
    float acc = 0.0f;
    if (loss_fn == LOSS_LOGITS_VAR) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float d = ls(c) - lt(c);
            acc += d * d;
            emit(c, f * (2.0f * (ls(c) - lt(c)) * inv_root_c));
        }
        out.loss = acc * inv_root_c;
        return out;
    }
    if (loss_fn == LOSS_LOGITS_SMOOTHL1) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float d = ls(c) - lt(c);
            acc += smooth_l1(d);
            const float g = fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f);
            emit(c, f * (g * inv_root_c));
        }
        out.loss = acc * inv_root_c;
        return out;
    }
    softmax_regs<CT>(ls, CT, ss);
    // probabilities IN PLACE of the exponentials: p_c = e_c * (1 / z), the rounded product softmax_prob returns
    float log_zs = 0.0f, log_zt = 0.0f;
    if (loss_fn == LOSS_KLD) {
        log_zs = logf(ss.z);
        log_zt = logf(st.z);
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        ss.e[c] = softmax_prob<CT>(ss, ls, c);
        st.e[c] = softmax_prob<CT>(st, lt, c);
    }
    const float* p = ss.e;
    const float* t = st.e;
    // ---- the per-pixel loss (consistency_pixel_fwd) and the two sums of the gradient (consistency_pixel_bwd)
    float dot = 0.0f;
    float tsum = 0.0f;
    if (loss_fn == LOSS_VAR) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float d = p[c] - t[c];
            acc += d * d;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float fp = 2.0f * (p[c] - t[c]);
            dot += fp * p[c];
        }
    } else if (loss_fn == LOSS_BCE) {
        const float eps = 1e-6f;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc += -(t[c] * logf(p[c] + eps) + (1.0f - t[c]) * logf(1.0f - p[c] + eps));
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float fp = -t[c] / (p[c] + eps) + (1.0f - t[c]) / (1.0f - p[c] + eps);
            dot += fp * p[c];
        }
    } else {  // LOSS_KLD
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const float logp = (ls(c) - ss.mx) - log_zs;
            const float logt = (lt(c) - st.mx) - log_zt;
            acc += t[c] > 0.0f ? t[c] * (logt - logp) : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            dot += 0.0f * p[c];
            tsum += t[c];
        }
    }
    out.loss = acc;
    // ---- dl_k = p_k * (f'_k - sum_c f'_c p_c)
#pragma unroll
    for (int k = 0; k < CT; ++k) {
        float g;
        if (loss_fn == LOSS_VAR) {
            g = p[k] * (2.0f * (p[k] - t[k]) - dot);
        } else if (loss_fn == LOSS_BCE) {
            const float eps = 1e-6f;
            g = p[k] * ((-t[k] / (p[k] + eps) + (1.0f - t[k]) / (1.0f - p[k] + eps)) - dot);
        } else {  // KLD: -t_k + p_k * sum_c t_c
            g = p[k] * tsum - t[k];
        }
        emit(k, f * g);
    }
    return out;
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
    SoftmaxRegs<CT> s;
    softmax_regs<CT>(l, crt, s);
#pragma unroll
    for (int k = 0; k < C; ++k) emit(k, softmax_prob<CT>(s, l, k) - (k == label ? 1.0f : 0.0f));
}

// -log_softmax(l)[label] and its gradient from ONE softmax (round 6, compile-time class count): the value as ce_fwd computes it
// (max, sum of exponentials in class order, one logf), the gradient as ce_pixel_bwd (e_k * (1 / z) - onehot).
template <int CT, class L, class E>
CMS_HD float ce_pixel_fwd_bwd(L l, float l_label, int label, E emit) {
    static_assert(CT > 0, "compile-time class count only");
    SoftmaxRegs<CT> s;
    softmax_regs<CT>(l, CT, s);
#pragma unroll
    for (int k = 0; k < CT; ++k) emit(k, softmax_prob<CT>(s, l, k) - (k == label ? 1.0f : 0.0f));
    return -((l_label - s.mx) - logf(s.z));
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
