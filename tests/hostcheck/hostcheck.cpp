// TEST INFRASTRUCTURE (never shipped, never imported by the product package).
//
// Drives the `__host__ __device__` per-pixel arithmetic of cutmix-semisup-seg_amd/csrc/pixel_math.hpp -- the exact
// code the HIP kernels inline -- in plain host loops, so that the formulas (bilinear taps, box membership, softmax /
// confidence, the five consistency losses and their analytic gradients, cross entropy, the 3-rounding EMA) can be
// checked against the oracle on a CPU-only machine before GPU time is spent. The kernels' indexing, reductions and
// LDS tiling are NOT covered here; the `-m gpu` tests cover those through the C ABI.
//
// Build: g++ -O1 -ffp-contract=off -shared -fPIC hostcheck.cpp -o _build/libhostcheck.so
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../cutmix-semisup-seg_amd/csrc/pixel_math.hpp"

using namespace cms;

struct HostGather {
    const float* base;
    size_t plane;
    int w_in;
    Tap ty, tx;
    float operator()(int c) const { return bilin_gather(base + c * plane, w_in, ty, tx); }
};

extern "C" {

void hc_box_mask(const int32_t* ranges, int n, int nb, int H, int W, int invert, float* out) {
    for (int i = 0; i < n; ++i)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                out[((size_t)i * H + y) * W + x] = box_mask_bit(ranges + (size_t)i * nb * 4, nb, y, x, invert != 0) ? 1.f : 0.f;
}

void hc_upsample(const float* lo, float* hi, int nc, int h, int w, int H, int W, int align) {
    const float sy = bilin_scale(h, H, align != 0), sx = bilin_scale(w, W, align != 0);
    for (int p = 0; p < nc; ++p)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                Tap ty = bilin_tap(y, sy, h, align != 0), tx = bilin_tap(x, sx, w, align != 0);
                hi[((size_t)p * H + y) * W + x] = bilin_gather(lo + (size_t)p * h * w, w, ty, tx);
            }
}

// stats[3] = {sum loss*um, sum loss*um*conf, count conf}; if grad != NULL also accumulates gscale * d/dl_stu
void hc_consistency(const float* l_stu, const float* l_t0, const float* l_t1, const float* mask, const float* um0,
                    const float* um1, int n, int c, int h, int w, int H, int W, int align, int mode, int loss_fn,
                    float tau, int per_pixel, double* stats, float gscale, float* grad) {
    const float sy = bilin_scale(h, H, align != 0), sx = bilin_scale(w, W, align != 0);
    const size_t plane = (size_t)h * w;
    const float inv_root_c = (float)(1.0 / sqrt((double)c));
    stats[0] = stats[1] = stats[2] = 0.0;
    std::vector<float> gv(c);
    for (int i = 0; i < n; ++i)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = ((size_t)i * H + y) * W + x;
                const bool m = mask[pix] >= 0.5f;
                const float* tea;
                float um;
                if (mode == MODE_MIX) {
                    tea = (m ? l_t1 : l_t0) + (size_t)i * c * plane;
                    const float* u = m ? um1 : um0;
                    um = u ? u[pix] : 1.0f;
                } else {
                    tea = l_t0 + (size_t)i * c * plane;
                    um = m ? (um0 ? um0[pix] : 1.0f) : 0.0f;
                }
                HostGather gs, gt;
                gs.base = l_stu + (size_t)i * c * plane;
                gt.base = tea;
                gs.plane = gt.plane = plane;
                gs.w_in = gt.w_in = w;
                gs.ty = gt.ty = bilin_tap(y, sy, h, align != 0);
                gs.tx = gt.tx = bilin_tap(x, sx, w, align != 0);
                PixelFwd r = consistency_pixel_fwd<0>(gs, gt, c, loss_fn, inv_root_c);
                const float cf = (tau > 0.0f && r.conf >= tau) ? 1.0f : 0.0f;
                stats[0] += (double)(r.loss * um);
                stats[1] += (double)(r.loss * um * cf);
                stats[2] += cf;
                if (grad) {
                    const float conf = consistency_pixel_bwd<0>(gs, gt, c, loss_fn, inv_root_c,
                                                                [&](int k, float v) { gv[k] = v; });
                    float f = gscale * um;
                    if (tau > 0.0f && per_pixel && !(conf >= tau)) f = 0.0f;
                    float* gp = grad + (size_t)i * c * plane;
                    for (int k = 0; k < c; ++k) {
                        const float g = f * gv[k];
                        gp[k * plane + (size_t)gs.ty.i0 * w + gs.tx.i0] += gs.ty.w0 * gs.tx.w0 * g;
                        gp[k * plane + (size_t)gs.ty.i0 * w + gs.tx.i1] += gs.ty.w0 * gs.tx.w1 * g;
                        gp[k * plane + (size_t)gs.ty.i1 * w + gs.tx.i0] += gs.ty.w1 * gs.tx.w0 * g;
                        gp[k * plane + (size_t)gs.ty.i1 * w + gs.tx.i1] += gs.ty.w1 * gs.tx.w1 * g;
                    }
                }
            }
}

// stats[2] = {sum nll, count}; grad (optional) accumulates gscale * (softmax - onehot) through the upsample adjoint
void hc_ce(const float* logits, const int64_t* labels, int ignore_index, int n, int c, int h, int w, int H, int W,
           int align, double* stats, float gscale, float* grad) {
    const float sy = bilin_scale(h, H, align != 0), sx = bilin_scale(w, W, align != 0);
    const size_t plane = (size_t)h * w;
    stats[0] = stats[1] = 0.0;
    for (int i = 0; i < n; ++i)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const size_t pix = ((size_t)i * H + y) * W + x;
                const int label = (int)labels[pix];
                if (label == ignore_index || label < 0 || label >= c) continue;
                HostGather gl;
                gl.base = logits + (size_t)i * c * plane;
                gl.plane = plane;
                gl.w_in = w;
                gl.ty = bilin_tap(y, sy, h, align != 0);
                gl.tx = bilin_tap(x, sx, w, align != 0);
                stats[0] += (double)ce_pixel_fwd<0>(gl, c, label);
                stats[1] += 1.0;
                if (grad) {
                    float* gp = grad + (size_t)i * c * plane;
                    ce_pixel_bwd<0>(gl, c, label, [&](int k, float v) {
                        const float g = gscale * v;
                        gp[k * plane + (size_t)gl.ty.i0 * w + gl.tx.i0] += gl.ty.w0 * gl.tx.w0 * g;
                        gp[k * plane + (size_t)gl.ty.i0 * w + gl.tx.i1] += gl.ty.w0 * gl.tx.w1 * g;
                        gp[k * plane + (size_t)gl.ty.i1 * w + gl.tx.i0] += gl.ty.w1 * gl.tx.w0 * g;
                        gp[k * plane + (size_t)gl.ty.i1 * w + gl.tx.i1] += gl.ty.w1 * gl.tx.w1 * g;
                    });
                }
            }
}

void hc_ema(float* tgt, const float* src, size_t count, float alpha, float oma) {
    for (size_t i = 0; i < count; ++i) tgt[i] = ema_update(tgt[i], src[i], alpha, oma);
}

}  // extern "C"
