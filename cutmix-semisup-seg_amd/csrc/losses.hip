// Fused loss kernels of the CutMix mean-teacher step (gfx950).
//
//   consistency:  bilinear upsample of low-res logits + paste of the two teacher predictions with the box mask
//                 (rasterised in-kernel) + softmax x2 + confidence threshold + one of five per-pixel losses +
//                 masked mean, forward and backward.           train_seg_semisup_mask_mt.py:363-367, 407-459
//   supervised:   bilinear upsample + log-softmax + NLL(ignore_index), forward and backward.    :126, 299-301
//
// Design (MI355X-first): the (N,C,H,W) full-resolution logits / probabilities / per-pixel loss maps of the
// reference (12+ elementwise kernels and their autograd twins, (5C+4)*P*4 bytes of HBM traffic) are never
// materialised. One thread owns one output pixel, lanes of a wave are consecutive x so the hi-res validity masks
// are read fully coalesced, the low-res logits (a few MB, L2-resident) are gathered with the 4 bilinear taps, the
// class axis lives in registers (compile-time C for 2/5/19/21) so the channel reductions need no cross-lane
// traffic; the only cross-lane work is the wave-shuffle + LDS block reduction of the three loss sums.
// Backward: the per-pixel gradient vector is scattered to the low-res logits through an LDS-tiled separable
// adjoint of the bilinear upsample (x-reduce, then y-reduce, fixed order inside a tile), so global atomics are
// issued per low-res cell per tile instead of per pixel per tap.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"

namespace cms {

// ------------------------------------------------------------------------------------------------ accessors
template <bool IDENT>
struct Gather {
    const float* base;  // class-0 plane of sample n
    size_t plane;       // h*w
    int w_in;
    Tap ty, tx;
    size_t off;  // y*w + x, IDENT only
    __device__ __forceinline__ float operator()(int c) const {
        if (IDENT) return base[c * plane + off];
        return bilin_gather(base + c * plane, w_in, ty, tx);
    }
};

template <int CT>
struct RegVec {
    float v[CT > 0 ? CT : 1];
    __device__ __forceinline__ float operator()(int c) const { return v[c]; }
};

template <int CT, bool IDENT>
__device__ __forceinline__ void fill(RegVec<CT>& r, const Gather<IDENT>& g) {
#pragma unroll
    for (int c = 0; c < CT; ++c) r.v[c] = g(c);
}

struct Geo {
    int n, c, h, w, H, W, align;
    float sy, sx;
    // tiled backward: this launch covers the tiles (col_x + i * col_kx, col_y + j * col_ky) -- one COLOUR class of the tile
    // grid, chosen on the host so that no two tiles of a launch touch the same low-resolution cell (see tiled_launches)
    int col_kx, col_ky, col_x, col_y;
};

// ------------------------------------------------------------------------------------------------ consistency
struct ConsArgs {
    cms_consistency_desc d;
    Geo g;
    float tau, inv_root_c;
};

struct ConsPixel {
    bool m;
    float um;
    const float* tea;
    int which;          // 0: l_tea0, 1: l_tea1 -- for callers that read the teacher logits from an LDS copy
};

__device__ __forceinline__ ConsPixel cons_pixel_inputs(const ConsArgs& a, int n, int y, int x) {
    const cms_consistency_desc& d = a.d;
    const size_t pix = ((size_t)n * d.H + y) * d.W + x;
    ConsPixel p;
    if (d.mask) {
        p.m = d.mask[pix] >= 0.5f;
    } else {
        p.m = box_mask_bit(d.ranges + (size_t)n * d.n_boxes * 4, d.n_boxes, y, x, d.invert != 0);
    }
    const size_t sample = (size_t)n * d.c * d.h * d.w;
    if (d.mode == MODE_MIX) {
        // paste of teacher logits and of the validity masks with the same box mask (:351, :363)
        p.tea = (p.m ? d.l_tea1 : d.l_tea0) + sample;
        p.which = p.m ? 1 : 0;
        const float* um = p.m ? d.um1 : d.um0;
        p.um = um ? um[pix] : 1.0f;
    } else {
        // cut mode: loss_mask = cut_mask * um (:401)
        p.tea = d.l_tea0 + sample;
        p.which = 0;
        p.um = p.m ? (d.um0 ? d.um0[pix] : 1.0f) : 0.0f;
    }
    return p;
}

// ---- LDS-staged low-resolution patches (round 5) ---------------------------------------------------------------
// A workgroup owns a tile of output pixels of ONE sample; the bilinear taps of the whole tile fall into a small rectangle
// of low-resolution cells (3 rows x 11 columns at the 1/8 scale of the DeepLab heads). The kernels used to gather every tap
// of every class of every pixel from global memory (168 four-byte loads per pixel for 21 classes and two tensors: latency-
// bound, 0.07-0.08 of the HBM rate on moved bytes); now the rectangle of each logit tensor is copied to LDS once
// per workgroup, [class][row][column], and the per-pixel gathers read it there with the taps rebased -- the same arithmetic
// (bilin_gather) on the same values, so every per-pixel result is bit-identical.
constexpr int TILE_W = 64;
constexpr int FWD_TILE_H = 8;               // forward kernels: 64 x 8 pixels per workgroup, 2 per thread
struct Patch {
    int x_lo, n_cols, y_lo, n_rows;
};

__device__ __forceinline__ Patch tile_patch(const Geo& g, int x0, int y0, int tw, int th) {
    // i0 / i1 are monotone in the output coordinate: the first pixel's i0 and the last pixel's i1 bound the rectangle
    const Tap xa = bilin_tap(x0, g.sx, g.w, g.align != 0), xb = bilin_tap(x0 + tw - 1, g.sx, g.w, g.align != 0);
    const Tap ya = bilin_tap(y0, g.sy, g.h, g.align != 0), yb = bilin_tap(y0 + th - 1, g.sy, g.h, g.align != 0);
    Patch p;
    p.x_lo = xa.i0; p.n_cols = xb.i1 - xa.i0 + 1;
    p.y_lo = ya.i0; p.n_rows = yb.i1 - ya.i0 + 1;
    return p;
}

// dst[c][r][j] = src[c][y_lo + r][x_lo + j] for the C class planes of one sample (`src` = its class-0 plane)
__device__ __forceinline__ void stage_patch(float* __restrict__ dst, const float* __restrict__ src, int C, size_t plane, int w,
                                            const Patch& p) {
    const int per = p.n_rows * p.n_cols, total = C * per;
    for (int i = (int)threadIdx.x; i < total; i += (int)blockDim.x) {
        const int c = i / per, rj = i - c * per;
        const int r = rj / p.n_cols, j = rj - r * p.n_cols;
        dst[i] = src[(size_t)c * plane + (size_t)(p.y_lo + r) * w + (p.x_lo + j)];
    }
}

__device__ __forceinline__ void rebase(Tap& ty, Tap& tx, const Patch& p) {
    ty.i0 -= p.y_lo; ty.i1 -= p.y_lo;
    tx.i0 -= p.x_lo; tx.i1 -= p.x_lo;
}

inline int tile_max_cols(float sx) { return (int)((TILE_W - 1) * sx) + 3; }
inline int tile_max_rows_of(float sy, int tile_h) { return (int)((tile_h - 1) * sy) + 3; }
inline size_t patch_floats(int C, float sy, float sx, int tile_h) {
    return (size_t)C * tile_max_rows_of(sy, tile_h) * tile_max_cols(sx);
}
constexpr size_t FWD_PATCH_LDS_MAX = 96 * 1024;      // beyond (scales near 1 with many classes): the direct-gather kernels

template <int CT>
__global__ __launch_bounds__(256) void cons_fwd_tiled_kernel(ConsArgs a, float* __restrict__ partials, int patch_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Geo& g = a.g;
    const int tiles_x = (g.W + TILE_W - 1) / TILE_W, tiles_y = (g.H + FWD_TILE_H - 1) / FWD_TILE_H;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x;
    b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int x0 = tx_i * TILE_W, y0 = ty_i * FWD_TILE_H;
    const int tw = min(TILE_W, g.W - x0), th = min(FWD_TILE_H, g.H - y0);
    const Patch p = tile_patch(g, x0, y0, tw, th);
    const size_t plane = (size_t)g.h * g.w;
    float* Ps = smem;
    float* Pt0 = smem + patch_stride;
    float* Pt1 = smem + 2 * patch_stride;
    const size_t sample = (size_t)n * g.c * plane;
    stage_patch(Ps, a.d.l_stu + sample, g.c, plane, g.w, p);
    stage_patch(Pt0, a.d.l_tea0 + sample, g.c, plane, g.w, p);
    if (a.d.mode == MODE_MIX) stage_patch(Pt1, a.d.l_tea1 + sample, g.c, plane, g.w, p);
    __syncthreads();
    float acc[3] = {0.0f, 0.0f, 0.0f};
    const int col = threadIdx.x & (TILE_W - 1);
#pragma unroll 1
    for (int rr = 0; rr < FWD_TILE_H / 4; ++rr) {
        const int row = (threadIdx.x >> 6) + rr * 4;
        if (col < tw && row < th) {
            const int y = y0 + row, x = x0 + col;
            const ConsPixel px = cons_pixel_inputs(a, n, y, x);
            Gather<false> gs, gt;
            gs.base = Ps;
            gt.base = px.which ? Pt1 : Pt0;
            gs.plane = gt.plane = (size_t)p.n_rows * p.n_cols;
            gs.w_in = gt.w_in = p.n_cols;
            Tap ty = bilin_tap(y, g.sy, g.h, g.align != 0), tx = bilin_tap(x, g.sx, g.w, g.align != 0);
            rebase(ty, tx, p);
            gs.ty = gt.ty = ty;
            gs.tx = gt.tx = tx;
            PixelFwd r;
            if (CT > 0) {
                RegVec<CT> rs, rt;
                fill<CT, false>(rs, gs);
                fill<CT, false>(rt, gt);
                r = consistency_pixel_fwd<CT>(rs, rt, g.c, a.d.loss_fn, a.inv_root_c);
            } else {
                r = consistency_pixel_fwd<0>(gs, gt, g.c, a.d.loss_fn, a.inv_root_c);
            }
            const float lm = r.loss * px.um;
            const float cf = (a.tau > 0.0f && r.conf >= a.tau) ? 1.0f : 0.0f;
            acc[0] += lm;
            acc[1] += lm * cf;
            acc[2] += cf;
        }
    }
    __shared__ float red[3 * 16];
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 3 + 0] = acc[0];
        partials[blockIdx.x * 3 + 1] = acc[1];
        partials[blockIdx.x * 3 + 2] = acc[2];
    }
}

template <int CT, bool IDENT>
__global__ __launch_bounds__(256) void cons_fwd_kernel(ConsArgs a, float* __restrict__ partials) {
    const Geo& g = a.g;
    const size_t P = (size_t)g.n * g.H * g.W;
    const size_t plane = (size_t)g.h * g.w;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % g.W);
        const size_t t = idx / g.W;
        const int y = (int)(t % g.H);
        const int n = (int)(t / g.H);
        const ConsPixel px = cons_pixel_inputs(a, n, y, x);
        Gather<IDENT> gs, gt;
        gs.base = a.d.l_stu + (size_t)n * g.c * plane;
        gt.base = px.tea;
        gs.plane = gt.plane = plane;
        gs.w_in = gt.w_in = g.w;
        if (IDENT) {
            gs.off = gt.off = (size_t)y * g.w + x;
        } else {
            gs.ty = gt.ty = bilin_tap(y, g.sy, g.h, g.align != 0);
            gs.tx = gt.tx = bilin_tap(x, g.sx, g.w, g.align != 0);
        }
        PixelFwd r;
        if (CT > 0) {
            RegVec<CT> rs, rt;
            fill<CT, IDENT>(rs, gs);
            fill<CT, IDENT>(rt, gt);
            r = consistency_pixel_fwd<CT>(rs, rt, g.c, a.d.loss_fn, a.inv_root_c);
        } else {
            r = consistency_pixel_fwd<0>(gs, gt, g.c, a.d.loss_fn, a.inv_root_c);
        }
        const float lm = r.loss * px.um;
        const float cf = (a.tau > 0.0f && r.conf >= a.tau) ? 1.0f : 0.0f;
        acc[0] += lm;
        acc[1] += lm * cf;
        acc[2] += cf;
    }
    __shared__ float red[3 * 16];
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 3 + 0] = acc[0];
        partials[blockIdx.x * 3 + 1] = acc[1];
        partials[blockIdx.x * 3 + 2] = acc[2];
    }
}

// second stage: fixed-order sum of the per-workgroup partials in double
template <int K>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partials, int nblocks,
                                                              double* __restrict__ out, double extra, int extra_slot) {
    __shared__ double sm[K][256];
    double loc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) loc[k] = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
#pragma unroll
        for (int k = 0; k < K; ++k) loc[k] += (double)partials[i * K + k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) sm[k][threadIdx.x] = loc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < K; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = sm[k][0];
        if (extra_slot >= 0) out[extra_slot] = extra;
    }
}

__global__ void cons_finalize_kernel(const double* __restrict__ sl, const double* __restrict__ sg, float tau,
                                     int per_pixel, float ramp, float weight, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double Pl = sl[3];
    double closs, gs, rate;
    if (tau > 0.0f) {
        rate = sg[2] / sg[3];
        if (per_pixel) {
            closs = sl[1] / Pl;
            gs = 1.0 / Pl;
        } else {
            // default mode: the confidence mask is replaced by its scalar mean (:415-418)
            closs = rate * (sl[0] / Pl);
            gs = rate / Pl;
        }
    } else {
        rate = NAN;
        closs = sl[0] / Pl;
        gs = 1.0 / Pl;
    }
    closs *= (double)ramp;                      // :454-455
    out[0] = (float)closs;                      // logged value, :461
    out[1] = (float)rate;                       // :413
    out[2] = (float)(gs * (double)ramp * (double)weight);
    out[3] = (float)(closs * (double)weight);   // :458
}

// ---- backward, identity geometry (h == H, w == W): gradients land directly on their own pixel
template <int CT>
__global__ __launch_bounds__(256) void cons_bwd_ident_kernel(ConsArgs a, const float* __restrict__ scalars,
                                                             float* __restrict__ grad) {
    const Geo& g = a.g;
    const size_t P = (size_t)g.n * g.H * g.W;
    const size_t plane = (size_t)g.h * g.w;
    const float gscale = scalars[2];
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(idx % g.W);
        const size_t t = idx / g.W;
        const int y = (int)(t % g.H);
        const int n = (int)(t / g.H);
        const ConsPixel px = cons_pixel_inputs(a, n, y, x);
        Gather<true> gs, gt;
        gs.base = a.d.l_stu + (size_t)n * g.c * plane;
        gt.base = px.tea;
        gs.plane = gt.plane = plane;
        gs.w_in = gt.w_in = g.w;
        gs.off = gt.off = (size_t)y * g.w + x;
        float* gp = grad + (size_t)n * g.c * plane + gs.off;
        const float base_f = gscale * px.um;
        const bool pp = a.tau > 0.0f && a.d.conf_per_pixel;
        // no early-out on base_f == 0 (confidence rate 0 at random init, invalid pixels): the masked-consistency
        // backward always does its full work, SURVEY.md 8(d) -- a zero factor simply contributes zeros
        if (CT > 0) {
            RegVec<CT> rs, rt;
            fill<CT, true>(rs, gs);
            fill<CT, true>(rt, gt);
            float gv[CT > 0 ? CT : 1];
            const float conf = consistency_pixel_bwd<CT>(rs, rt, g.c, a.d.loss_fn, a.inv_root_c,
                                                        [&](int k, float v) { gv[k] = v; });
            const float f = (pp && !(conf >= a.tau)) ? 0.0f : base_f;
#pragma unroll
            for (int k = 0; k < CT; ++k) gp[k * plane] += f * gv[k];
        } else {
            float ms, zs, mt, zt;
            softmax_stats<0>(gt, g.c, mt, zt);
            (void)ms; (void)zs;
            const float conf = 1.0f / zt;
            const float f = (pp && !(conf >= a.tau)) ? 0.0f : base_f;
            consistency_pixel_bwd<0>(gs, gt, g.c, a.d.loss_fn, a.inv_root_c,
                                     [&](int k, float v) { gp[k * plane] += f * v; });
        }
    }
}

// ---- backward with upsampling: LDS-tiled adjoint of the bilinear interpolation ---------------------------------
// Workgroup = 256 threads = one tile of TILE_H x TILE_W output pixels of one sample.
//   phase 1  every thread computes the gradient vector of its 2 pixels -> G[row][class][col]        (LDS)
//   phase 2  x-adjoint:  R[row][class][j] = sum_col wx(col -> cell j) * G[row][class][col]          (LDS)
//   phase 3  y-adjoint:  out[class][i][j] = sum_row wy(row -> cell i) * R[row][class][j]  -> one global atomicAdd
// Summation order inside a tile is fixed. Neighbouring tiles share low-resolution cells; so that their sums do not meet in
// an order that changes from run to run, the tile grid is issued as kx * ky COLOUR classes, one launch each, in a fixed
// order on the stream: within a launch every cell receives exactly one add (round 3: the logit gradients, and with them
// every weight gradient downstream, are run-to-run reproducible).
#ifndef CMS_LOSS_TILE_H
#define CMS_LOSS_TILE_H 4
#endif
constexpr int TILE_H = CMS_LOSS_TILE_H;     // 4 or 8 (256 threads = 64 columns x 4 rows, TILE_H / 4 pixels per thread). Measured with 4
                                            // (half the LDS, twice the workgroups): consistency / CE backward 205 -> 160 / 170 -> 120 us,
                                            // the step unchanged (620.4 vs 619.4 img/s, profiles/r04zw_*): these launches run beside the
                                            // operand re-pack and are no longer what the step waits for. Round 5: 4 is the default --
                                            // with the logit rectangles staged in LDS next to G and R a 4-row tile keeps 4 workgroups per CU
constexpr int G_LD = TILE_W + 1;  // +1 float: conflict-free column access for class-major readers

struct TileTables {
    int xi0[TILE_W], xi1[TILE_W];
    float xw1[TILE_W];
    int yi0[TILE_H], yi1[TILE_H];
    float yw1[TILE_H];
    int xbeg[TILE_W + 2], xend[TILE_W + 2];
};

inline int tile_max_rows(float sy) { return tile_max_rows_of(sy, TILE_H); }

// G + R + `n_patches` staged logit rectangles (student / teacher 0 / teacher 1, or the one tensor of the cross entropy)
inline size_t tile_lds_bytes(int C, float sy, float sx, int n_patches) {
    size_t g = (size_t)TILE_H * C * G_LD;
    size_t r = (size_t)TILE_H * C * tile_max_cols(sx);
    return (g + r + (size_t)n_patches * patch_floats(C, sy, sx, TILE_H)) * sizeof(float);
}

template <class Stage, class PixelGrad>
__device__ __forceinline__ void tiled_scatter(const Geo& g, Stage stage, PixelGrad pixel_grad, float* __restrict__ grad_lo,
                                              float* smem) {
    __shared__ TileTables tb;
    const int tiles_x = (g.W + TILE_W - 1) / TILE_W;
    const int tiles_y = (g.H + TILE_H - 1) / TILE_H;
    const int ctx = (tiles_x - g.col_x + g.col_kx - 1) / g.col_kx;      // tiles of this colour class per row / column
    const int cty = (tiles_y - g.col_y + g.col_ky - 1) / g.col_ky;
    int b = blockIdx.x;
    const int tx_i = g.col_x + (b % ctx) * g.col_kx;
    b /= ctx;
    const int ty_i = g.col_y + (b % cty) * g.col_ky;
    const int n = b / cty;
    const int x0 = tx_i * TILE_W, y0 = ty_i * TILE_H;
    const int tw = min(TILE_W, g.W - x0), th = min(TILE_H, g.H - y0);
    const int C = g.c;
    const int tid = threadIdx.x;

    if (tid < TILE_W) {
        Tap t = bilin_tap(min(x0 + tid, g.W - 1), g.sx, g.w, g.align != 0);
        tb.xi0[tid] = t.i0;
        tb.xi1[tid] = t.i1;
        tb.xw1[tid] = t.w1;
    } else if (tid < TILE_W + TILE_H) {
        const int r = tid - TILE_W;
        Tap t = bilin_tap(min(y0 + r, g.H - 1), g.sy, g.h, g.align != 0);
        tb.yi0[r] = t.i0;
        tb.yi1[r] = t.i1;
        tb.yw1[r] = t.w1;
    }
    __syncthreads();
    // the tile's rectangle of low-resolution cells: every thread reads it off the tables (four broadcast LDS reads) -- no
    // single-thread section and no second barrier in front of the staging (round 6)
    const int r_x_lo = tb.xi0[0], r_n_cols = tb.xi1[tw - 1] - r_x_lo + 1;
    const int r_y_lo = tb.yi0[0], r_n_rows = tb.yi1[th - 1] - r_y_lo + 1;
    // contiguous range [xbeg, xend) of tile columns that touch low-res column x_lo + j (i0 is monotone in x). (round 6) every tile
    // column reports itself to its two cells with LDS atomics -- one step for 64 lanes -- instead of n_cols lanes scanning all 64
    // columns in a dependent loop while the other three waves of the workgroup wait (same table, bit-identical results)
    if (tid < r_n_cols) {
        tb.xbeg[tid] = tw;
        tb.xend[tid] = 0;
    }
    __syncthreads();
    if (tid < tw) {
        const int j0 = tb.xi0[tid] - r_x_lo, j1 = tb.xi1[tid] - r_x_lo;
        atomicMin(&tb.xbeg[j0], tid);
        atomicMax(&tb.xend[j0], tid + 1);
        atomicMin(&tb.xbeg[j1], tid);
        atomicMax(&tb.xend[j1], tid + 1);
    }

    float* G = smem;                                 // [TILE_H][C][G_LD]
    float* R = smem + (size_t)TILE_H * C * G_LD;     // [TILE_H][C][n_cols]
    // (round 5) phase 0: the tile's rectangle of every logit tensor -> LDS behind G and R; phase 1 gathers from there
    Patch patch;
    patch.x_lo = r_x_lo; patch.n_cols = r_n_cols; patch.y_lo = r_y_lo; patch.n_rows = r_n_rows;
    float* P = R + (size_t)TILE_H * C * r_n_cols;
    stage(n, patch, P);
    // (round 6) x-adjoint weights as a table: wtab[j][u] = weight of tile column xbeg[j] + u in low-res column j -- phase 2's inner
    // loop is then two LDS reads and an FMA per term instead of three table reads, two compares and two selects (same values, same
    // order: bit-identical). Small tables only (the DeepLab scales: 11-19 columns, <= 18 terms); otherwise the comparing loop.
    constexpr int WT_COLS = 24, WT_SPAN = 20;
    __shared__ float wtab[WT_COLS * WT_SPAN];
    const bool wide = tid < r_n_cols && (tb.xend[tid] - tb.xbeg[tid]) > WT_SPAN;
    const bool table = __syncthreads_or(wide ? 1 : 0) == 0 && r_n_cols <= WT_COLS;      // (also the barrier behind the staging)
    if (table) {
        for (int e = tid; e < r_n_cols * WT_SPAN; e += blockDim.x) {
            const int j = e / WT_SPAN, u = e - j * WT_SPAN;
            const int c = tb.xbeg[j] + u;
            float wv = 0.0f;
            if (c < tb.xend[j]) {
                const int X = r_x_lo + j;
                const float w1 = tb.xw1[c];
                wv = (tb.xi0[c] == X ? 1.0f - w1 : 0.0f) + (tb.xi1[c] == X ? w1 : 0.0f);
            }
            wtab[e] = wv;
        }
    }

    // phase 1
    const int col = tid & (TILE_W - 1);
#pragma unroll
    for (int rr = 0; rr < TILE_H / 4; ++rr) {
        const int row = (tid >> 6) + rr * 4;
        float* gcol = G + ((size_t)row * C) * G_LD + col;
        const bool valid = col < tw && row < th;
        bool wrote = false;
        if (valid) {
            Tap ty, tx;
            ty.i0 = tb.yi0[row]; ty.i1 = tb.yi1[row]; ty.w1 = tb.yw1[row]; ty.w0 = 1.0f - ty.w1;
            tx.i0 = tb.xi0[col]; tx.i1 = tb.xi1[col]; tx.w1 = tb.xw1[col]; tx.w0 = 1.0f - tx.w1;
            rebase(ty, tx, patch);
            wrote = pixel_grad(n, y0 + row, x0 + col, ty, tx, patch, P, [&](int k, float v) { gcol[(size_t)k * G_LD] = v; });
        }
        if (!wrote) {
            for (int k = 0; k < C; ++k) gcol[(size_t)k * G_LD] = 0.0f;
        }
    }
    __syncthreads();

    // phase 2: items (row, j, class), class fastest across lanes -> stride G_LD reads, conflict-free
    const int n_cols = r_n_cols, n_rows = r_n_rows;
    const int items2 = th * n_cols * C;
    for (int it = tid; it < items2; it += blockDim.x) {
        const int k = it % C;
        const int rj = it / C;
        const int j = rj % n_cols;
        const int row = rj / n_cols;
        const int X = r_x_lo + j;
        const float* gr = G + ((size_t)row * C + k) * G_LD;
        float s = 0.0f;
        if (table) {
            const int cb = tb.xbeg[j], nterm = tb.xend[j] - cb;
            const float* wt = wtab + j * WT_SPAN;
            for (int u = 0; u < nterm; ++u) s += wt[u] * gr[cb + u];
        } else {
            for (int c = tb.xbeg[j]; c < tb.xend[j]; ++c) {
                const float w1 = tb.xw1[c];
                float wgt = (tb.xi0[c] == X ? 1.0f - w1 : 0.0f) + (tb.xi1[c] == X ? w1 : 0.0f);
                s += wgt * gr[c];
            }
        }
        R[((size_t)row * C + k) * n_cols + j] = s;
    }
    __syncthreads();

    // phase 3: items (class, i, j), j fastest -> coalesced atomics
    const int items3 = C * n_rows * n_cols;
    const size_t plane = (size_t)g.h * g.w;
    float* out_n = grad_lo + (size_t)n * C * plane;
    for (int it = tid; it < items3; it += blockDim.x) {
        const int j = it % n_cols;
        const int ki = it / n_cols;
        const int i = ki % n_rows;
        const int k = ki / n_rows;
        const int Y = r_y_lo + i;
        float s = 0.0f;
        for (int row = 0; row < th; ++row) {
            const float w1 = tb.yw1[row];
            float wgt = (tb.yi0[row] == Y ? 1.0f - w1 : 0.0f) + (tb.yi1[row] == Y ? w1 : 0.0f);
            s += wgt * R[((size_t)row * C + k) * n_cols + j];
        }
        if (s != 0.0f) atomicAdd(out_n + (size_t)k * plane + (size_t)Y * g.w + (r_x_lo + j), s);
    }
}

// Smallest tile distance k along one axis such that tiles t and t + k never touch the same low-resolution cell.
inline int tile_colour_period(int full, int low, float scale, bool align, int tile) {
    const int T = (full + tile - 1) / tile;
    int k = 1;
    for (;;) {
        bool clash = false;
        for (int t = 0; t + k < T && !clash; ++t) {
            const Tap hi = bilin_tap(std::min(t * tile + tile - 1, full - 1), scale, low, align);
            const Tap lo = bilin_tap((t + k) * tile, scale, low, align);
            clash = lo.i0 <= hi.i1;
        }
        if (!clash || k >= T) return k;
        ++k;
    }
}

// Run-to-run reproducible backward of the two losses (cms_loss_set_deterministic / CMS_LOSS_DETERMINISTIC=1): the tiles go
// out as colour classes (below). Off (the throughput default, like the fp32 atomics of the weight gradients): ALL tiles in one
// launch -- tiles that share a low-resolution cell then add into it in a run-dependent order (fp32 atomics, ~1e-7), and the
// launch has four times the workgroups of a colour class: the four launches of a class each kept < 1 round of the machine
// busy and sat on the critical path between the forward and the backward pass (4 x 45-65 us per loss at 321 x 321).
static int g_loss_deterministic = -1;
static bool loss_deterministic() {
    if (g_loss_deterministic < 0) {
        const char* e = getenv("CMS_LOSS_DETERMINISTIC");
        g_loss_deterministic = e ? (atoi(e) != 0) : 0;
    }
    return g_loss_deterministic != 0;
}

// Issues `launch(geo, tiles)` once per colour class of the tile grid, in a fixed order (or once for all tiles, see above).
template <class F>
inline void tiled_launches(Geo g, F launch) {
    const int tiles_x = (g.W + TILE_W - 1) / TILE_W, tiles_y = (g.H + TILE_H - 1) / TILE_H;
    if (!loss_deterministic()) {
        g.col_kx = g.col_ky = 1;
        g.col_x = g.col_y = 0;
        launch(g, tiles_x * tiles_y * g.n);
        return;
    }
    g.col_kx = tile_colour_period(g.W, g.w, g.sx, g.align != 0, TILE_W);
    g.col_ky = tile_colour_period(g.H, g.h, g.sy, g.align != 0, TILE_H);
    for (int cy = 0; cy < g.col_ky; ++cy)
        for (int cx = 0; cx < g.col_kx; ++cx) {
            const int ctx = (tiles_x - cx + g.col_kx - 1) / g.col_kx, cty = (tiles_y - cy + g.col_ky - 1) / g.col_ky;
            if (ctx <= 0 || cty <= 0) continue;
            g.col_x = cx;
            g.col_y = cy;
            launch(g, ctx * cty * g.n);
        }
}

template <int CT>
__global__ __launch_bounds__(256) void cons_bwd_tiled_kernel(ConsArgs a, const float* __restrict__ scalars,
                                                             float* __restrict__ grad, int patch_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Geo& g = a.g;
    const float gscale = scalars[2];
    const size_t plane = (size_t)g.h * g.w;
    const bool pp = a.tau > 0.0f && a.d.conf_per_pixel;
    // LDS copies of the tile's logit rectangles: student | teacher 0 | teacher 1, `pstride` floats apart
    const int pstride = patch_stride;
    auto stage = [&](int n, const Patch& p, float* P) {
        const size_t sample = (size_t)n * g.c * plane;
        stage_patch(P, a.d.l_stu + sample, g.c, plane, g.w, p);
        stage_patch(P + pstride, a.d.l_tea0 + sample, g.c, plane, g.w, p);
        if (a.d.mode == MODE_MIX) stage_patch(P + 2 * pstride, a.d.l_tea1 + sample, g.c, plane, g.w, p);
    };
    auto pixel_grad = [&](int n, int y, int x, const Tap& ty, const Tap& tx, const Patch& p, const float* P, auto emit) -> bool {
        const ConsPixel px = cons_pixel_inputs(a, n, y, x);
        const float base_f = gscale * px.um;
        // (no early-out on base_f == 0, see cons_bwd_ident_kernel)
        Gather<false> gs, gt;
        gs.base = P;
        gt.base = P + (px.which ? 2 * pstride : pstride);
        gs.plane = gt.plane = (size_t)p.n_rows * p.n_cols;
        gs.w_in = gt.w_in = p.n_cols;
        gs.ty = gt.ty = ty;                      // (taps already rebased to the rectangle)
        gs.tx = gt.tx = tx;
        if (CT > 0) {
            RegVec<CT> rs, rt;
            fill<CT, false>(rs, gs);
            fill<CT, false>(rt, gt);
            float gv[CT > 0 ? CT : 1];
            const float conf = consistency_pixel_bwd<CT>(rs, rt, g.c, a.d.loss_fn, a.inv_root_c,
                                                        [&](int k, float v) { gv[k] = v; });
            const float f = (pp && !(conf >= a.tau)) ? 0.0f : base_f;
#pragma unroll
            for (int k = 0; k < CT; ++k) emit(k, f * gv[k]);
        } else {
            float mt, zt;
            softmax_stats<0>(gt, g.c, mt, zt);
            const float conf = 1.0f / zt;
            const float f = (pp && !(conf >= a.tau)) ? 0.0f : base_f;
            consistency_pixel_bwd<0>(gs, gt, g.c, a.d.loss_fn, a.inv_root_c, [&](int k, float v) { emit(k, f * v); });
        }
        return true;
    };
    tiled_scatter(g, stage, pixel_grad, grad, smem);
}

// ---- forward + backward in ONE launch (round 6) -----------------------------------------------------------------------
// The backward kernel recomputes everything the forward kernel computed (the tile's logit rectangles, both softmaxes of every
// pixel) and needs from it only ONE scalar: the factor of the gradient. That factor is  ramp * weight / P * um  [* per-pixel
// confidence]  -- known up front -- times, in the default confidence mode, the scalar RATE (train_seg_semisup_mask_mt.py:415-418:
// the confidence mask is replaced by its mean), in which the gradient is LINEAR. So one pass computes the loss partial sums AND
// the gradient with the rate left out (`grad_unit` = ramp * weight / P), the finalising launch derives the scalars, and
// cms_scale_by_scalar multiplies the gradient rows by the rate afterwards (a 350 k-float pass). The chain between the forward and
// the backward pass of the step loses a launch of 0.10-0.12 ms and a second staging of all rectangles; under data parallelism
// the gradient no longer waits for the all-reduce of the confidence count either. Per-pixel values are those of the two
// kernels (same functions on the same registers).
template <int CT, int LF>
__global__ __launch_bounds__(256, (LF >= 0 && CT > 0) ? 4 : 1) void cons_fused_tiled_kernel(ConsArgs a, float grad_unit, float* __restrict__ grad,
                                                               int patch_stride, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Geo& g = a.g;
    const size_t plane = (size_t)g.h * g.w;
    const bool pp = a.tau > 0.0f && a.d.conf_per_pixel;
    const int pstride = patch_stride;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    auto stage = [&](int n, const Patch& p, float* P) {
        const size_t sample = (size_t)n * g.c * plane;
        stage_patch(P, a.d.l_stu + sample, g.c, plane, g.w, p);
        stage_patch(P + pstride, a.d.l_tea0 + sample, g.c, plane, g.w, p);
        if (a.d.mode == MODE_MIX) stage_patch(P + 2 * pstride, a.d.l_tea1 + sample, g.c, plane, g.w, p);
    };
    auto pixel_grad = [&](int n, int y, int x, const Tap& ty, const Tap& tx, const Patch& p, const float* P, auto emit) -> bool {
        const ConsPixel px = cons_pixel_inputs(a, n, y, x);
        const float base_f = grad_unit * px.um;
        // (no early-out on base_f == 0, see cons_bwd_ident_kernel)
        Gather<false> gs, gt;
        gs.base = P;
        gt.base = P + (px.which ? 2 * pstride : pstride);
        gs.plane = gt.plane = (size_t)p.n_rows * p.n_cols;
        gs.w_in = gt.w_in = p.n_cols;
        gs.ty = gt.ty = ty;                      // (taps already rebased to the rectangle)
        gs.tx = gt.tx = tx;
        PixelFwd r;
        if (CT > 0) {
            RegVec<CT> rs, rt;
            fill<CT, false>(rs, gs);
            fill<CT, false>(rt, gt);
            r = consistency_pixel_fwd_bwd<(CT > 0 ? CT : 1), LF>(
                rs, rt, a.d.loss_fn, a.inv_root_c,
                [&](float conf) -> float { return (pp && !(conf >= a.tau)) ? 0.0f : base_f; },
                [&](int k, float v) { emit(k, v); });
        } else {
            r = consistency_pixel_fwd<0>(gs, gt, g.c, a.d.loss_fn, a.inv_root_c);
            const float f = (pp && !(r.conf >= a.tau)) ? 0.0f : base_f;
            consistency_pixel_bwd<0>(gs, gt, g.c, a.d.loss_fn, a.inv_root_c, [&](int k, float v) { emit(k, f * v); });
        }
        const float lm = r.loss * px.um;
        const float cf = (a.tau > 0.0f && r.conf >= a.tau) ? 1.0f : 0.0f;
        acc[0] += lm;
        acc[1] += lm * cf;
        acc[2] += cf;
        return true;
    };
    tiled_scatter(g, stage, pixel_grad, grad, smem);
    __shared__ float red[3 * 16];
    block_sum<3>(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 3 + 0] = acc[0];
        partials[blockIdx.x * 3 + 1] = acc[1];
        partials[blockIdx.x * 3 + 2] = acc[2];
    }
}

// x[i] *= scalars[idx] * factor for i < n (the deferred factor of the fused loss launches: a device scalar)
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(float* __restrict__ x, size_t n, const float* __restrict__ scalars,
                                                              int idx, float factor) {
    const float f = scalars[idx] * factor;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= f;
}

// ------------------------------------------------------------------------------------------------ cross entropy
struct CeArgs {
    cms_ce_desc d;
    Geo g;
};

__device__ __forceinline__ int load_label(const CeArgs& a, size_t pix) {
    if (a.d.label_dtype == CMS_LABEL_U8) return (int)((const uint8_t*)a.d.labels)[pix];
    const int64_t v = ((const int64_t*)a.d.labels)[pix];
    return (v < 0 || v > 0x7fffffff) ? -1 : (int)v;
}

template <int CT, bool IDENT>
__global__ __launch_bounds__(256) void ce_fwd_kernel(CeArgs a, float* __restrict__ partials) {
    const Geo& g = a.g;
    const size_t P = (size_t)g.n * g.H * g.W;
    const size_t plane = (size_t)g.h * g.w;
    float acc[2] = {0.0f, 0.0f};
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += (size_t)gridDim.x * blockDim.x) {
        const int label = load_label(a, idx);
        if (label == a.d.ignore_index || label < 0 || label >= g.c) continue;
        const int x = (int)(idx % g.W);
        const size_t t = idx / g.W;
        const int y = (int)(t % g.H);
        const int n = (int)(t / g.H);
        Gather<IDENT> gl;
        gl.base = a.d.logits + (size_t)n * g.c * plane;
        gl.plane = plane;
        gl.w_in = g.w;
        if (IDENT) {
            gl.off = (size_t)y * g.w + x;
        } else {
            gl.ty = bilin_tap(y, g.sy, g.h, g.align != 0);
            gl.tx = bilin_tap(x, g.sx, g.w, g.align != 0);
        }
        float v;
        if (CT > 0) {
            RegVec<CT> r;
            fill<CT, IDENT>(r, gl);
            float mx, z;
            softmax_stats<CT>(r, g.c, mx, z);
            v = -((gl(label) - mx) - logf(z));   // label is a run-time index: re-gather instead of indexing registers
        } else {
            v = ce_pixel_fwd<0>(gl, g.c, label);
        }
        acc[0] += v;
        acc[1] += 1.0f;
    }
    __shared__ float red[2 * 16];
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = acc[0];
        partials[blockIdx.x * 2 + 1] = acc[1];
    }
}

template <int CT>
__global__ __launch_bounds__(256) void ce_fwd_tiled_kernel(CeArgs a, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Geo& g = a.g;
    const int tiles_x = (g.W + TILE_W - 1) / TILE_W, tiles_y = (g.H + FWD_TILE_H - 1) / FWD_TILE_H;
    int b = blockIdx.x;
    const int tx_i = b % tiles_x;
    b /= tiles_x;
    const int ty_i = b % tiles_y;
    const int n = b / tiles_y;
    const int x0 = tx_i * TILE_W, y0 = ty_i * FWD_TILE_H;
    const int tw = min(TILE_W, g.W - x0), th = min(FWD_TILE_H, g.H - y0);
    const Patch p = tile_patch(g, x0, y0, tw, th);
    const size_t plane = (size_t)g.h * g.w;
    stage_patch(smem, a.d.logits + (size_t)n * g.c * plane, g.c, plane, g.w, p);
    __syncthreads();
    float acc[2] = {0.0f, 0.0f};
    const int col = threadIdx.x & (TILE_W - 1);
#pragma unroll
    for (int rr = 0; rr < FWD_TILE_H / 4; ++rr) {
        const int row = (threadIdx.x >> 6) + rr * 4;
        if (col < tw && row < th) {
            const int y = y0 + row, x = x0 + col;
            const int label = load_label(a, ((size_t)n * g.H + y) * g.W + x);
            if (!(label == a.d.ignore_index || label < 0 || label >= g.c)) {
                Gather<false> gl;
                gl.base = smem;
                gl.plane = (size_t)p.n_rows * p.n_cols;
                gl.w_in = p.n_cols;
                Tap ty = bilin_tap(y, g.sy, g.h, g.align != 0), tx = bilin_tap(x, g.sx, g.w, g.align != 0);
                rebase(ty, tx, p);
                gl.ty = ty;
                gl.tx = tx;
                float v;
                if (CT > 0) {
                    RegVec<CT> r;
                    fill<CT, false>(r, gl);
                    float mx, z;
                    softmax_stats<CT>(r, g.c, mx, z);
                    v = -((gl(label) - mx) - logf(z));   // label is a run-time index: re-gather instead of indexing registers
                } else {
                    v = ce_pixel_fwd<0>(gl, g.c, label);
                }
                acc[0] += v;
                acc[1] += 1.0f;
            }
        }
    }
    __shared__ float red[2 * 16];
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = acc[0];
        partials[blockIdx.x * 2 + 1] = acc[1];
    }
}

__global__ void ce_finalize_kernel(const double* __restrict__ stats, float weight, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    out[0] = (float)(stats[0] / stats[1]);
    out[1] = (float)((double)weight / stats[1]);
}

template <int CT>
__global__ __launch_bounds__(256) void ce_bwd_ident_kernel(CeArgs a, const float* __restrict__ scalars,
                                                           float* __restrict__ grad) {
    const Geo& g = a.g;
    const size_t P = (size_t)g.n * g.H * g.W;
    const size_t plane = (size_t)g.h * g.w;
    const float gscale = scalars[1];
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < P; idx += (size_t)gridDim.x * blockDim.x) {
        const int label = load_label(a, idx);
        if (label == a.d.ignore_index || label < 0 || label >= g.c) continue;
        const int x = (int)(idx % g.W);
        const size_t t = idx / g.W;
        const int y = (int)(t % g.H);
        const int n = (int)(t / g.H);
        Gather<true> gl;
        gl.base = a.d.logits + (size_t)n * g.c * plane;
        gl.plane = plane;
        gl.w_in = g.w;
        gl.off = (size_t)y * g.w + x;
        float* gp = grad + (size_t)n * g.c * plane + gl.off;
        if (CT > 0) {
            RegVec<CT> r;
            fill<CT, true>(r, gl);
            ce_pixel_bwd<CT>(r, g.c, label, [&](int k, float v) { gp[k * plane] += gscale * v; });
        } else {
            ce_pixel_bwd<0>(gl, g.c, label, [&](int k, float v) { gp[k * plane] += gscale * v; });
        }
    }
}

template <int CT>
__global__ __launch_bounds__(256) void ce_bwd_tiled_kernel(CeArgs a, const float* __restrict__ scalars,
                                                           float* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Geo& g = a.g;
    const float gscale = scalars[1];
    const size_t plane = (size_t)g.h * g.w;
    auto stage = [&](int n, const Patch& p, float* P) {
        stage_patch(P, a.d.logits + (size_t)n * g.c * plane, g.c, plane, g.w, p);
    };
    auto pixel_grad = [&](int n, int y, int x, const Tap& ty, const Tap& tx, const Patch& p, const float* P, auto emit) -> bool {
        const size_t pix = ((size_t)n * g.H + y) * g.W + x;
        const int label = load_label(a, pix);
        if (label == a.d.ignore_index || label < 0 || label >= g.c) return false;
        Gather<false> gl;
        gl.base = P;
        gl.plane = (size_t)p.n_rows * p.n_cols;
        gl.w_in = p.n_cols;
        gl.ty = ty;                              // (taps already rebased to the rectangle)
        gl.tx = tx;
        if (CT > 0) {
            RegVec<CT> r;
            fill<CT, false>(r, gl);
            ce_pixel_bwd<CT>(r, g.c, label, [&](int k, float v) { emit(k, gscale * v); });
        } else {
            ce_pixel_bwd<0>(gl, g.c, label, [&](int k, float v) { emit(k, gscale * v); });
        }
        return true;
    };
    tiled_scatter(g, stage, pixel_grad, grad, smem);
}

// forward + backward of the cross entropy in one launch (round 6): the gradient is (softmax - onehot) * weight / n_valid, linear
// in the one scalar the forward pass contributes (the count of valid labels, global under data parallelism) -- computed here
// with that factor left out, scaled by cms_scale_by_scalar behind cms_ce_finalize.
template <int CT>
__global__ __launch_bounds__(256) void ce_fused_tiled_kernel(CeArgs a, float* __restrict__ grad, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Geo& g = a.g;
    const size_t plane = (size_t)g.h * g.w;
    float acc[2] = {0.0f, 0.0f};
    auto stage = [&](int n, const Patch& p, float* P) {
        stage_patch(P, a.d.logits + (size_t)n * g.c * plane, g.c, plane, g.w, p);
    };
    auto pixel_grad = [&](int n, int y, int x, const Tap& ty, const Tap& tx, const Patch& p, const float* P, auto emit) -> bool {
        const size_t pix = ((size_t)n * g.H + y) * g.W + x;
        const int label = load_label(a, pix);
        if (label == a.d.ignore_index || label < 0 || label >= g.c) return false;
        Gather<false> gl;
        gl.base = P;
        gl.plane = (size_t)p.n_rows * p.n_cols;
        gl.w_in = p.n_cols;
        gl.ty = ty;                              // (taps already rebased to the rectangle)
        gl.tx = tx;
        float v;
        if (CT > 0) {
            RegVec<CT> r;
            fill<CT, false>(r, gl);
            // (the label is a run-time index: its logit is re-gathered instead of indexing registers, as in ce_fwd_tiled_kernel)
            v = ce_pixel_fwd_bwd<(CT > 0 ? CT : 1)>(r, gl(label), label, [&](int k, float gv) { emit(k, gv); });
        } else {
            v = ce_pixel_fwd<0>(gl, g.c, label);
            ce_pixel_bwd<0>(gl, g.c, label, [&](int k, float gv) { emit(k, gv); });
        }
        acc[0] += v;
        acc[1] += 1.0f;
        return true;
    };
    tiled_scatter(g, stage, pixel_grad, grad, smem);
    __shared__ float red[2 * 16];
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 2 + 0] = acc[0];
        partials[blockIdx.x * 2 + 1] = acc[1];
    }
}

// ------------------------------------------------------------------------------------------------ host side
static Geo make_geo(int n, int c, int h, int w, int H, int W, int align) {
    Geo g;
    g.n = n; g.c = c; g.h = h; g.w = w; g.H = H; g.W = W; g.align = align;
    g.col_kx = g.col_ky = 1;
    g.col_x = g.col_y = 0;
    g.sy = bilin_scale(h, H, align != 0);
    g.sx = bilin_scale(w, W, align != 0);
    return g;
}

static int check_cons(const cms_consistency_desc* d) {
    CMS_REQUIRE(d != nullptr, "consistency: null descriptor");
    CMS_REQUIRE(d->l_stu && d->l_tea0, "consistency: l_stu / l_tea0 must not be NULL");
    CMS_REQUIRE(d->mode == CMS_MODE_MIX || d->mode == CMS_MODE_CUT, "consistency: unknown mode %d", d->mode);
    CMS_REQUIRE(d->mode != CMS_MODE_MIX || d->l_tea1, "consistency: mix mode needs l_tea1");
    CMS_REQUIRE((d->ranges != nullptr) != (d->mask != nullptr), "consistency: give exactly one of ranges / mask");
    CMS_REQUIRE(d->ranges == nullptr || d->n_boxes >= 0, "consistency: n_boxes < 0");
    CMS_REQUIRE(d->n > 0 && d->c > 0 && d->h > 0 && d->w > 0 && d->H > 0 && d->W > 0, "consistency: bad geometry");
    CMS_REQUIRE(d->h <= d->H && d->w <= d->W, "consistency: logits larger than the loss geometry");
    CMS_REQUIRE(d->loss_fn >= CMS_LOSS_VAR && d->loss_fn <= CMS_LOSS_KLD, "Unknown consistency loss function %d",
                d->loss_fn);
    return CMS_OK;
}

static ConsArgs make_cons_args(const cms_consistency_desc* d) {
    ConsArgs a;
    a.d = *d;
    a.g = make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners);
    a.tau = d->conf_thresh;
    a.inv_root_c = (float)(1.0 / sqrt((double)d->c));
    return a;
}

static int fwd_grid(size_t P) { return grid_for(P, 256, 2048); }
// the LDS-staged forward kernels: one workgroup per 64 x 8 tile of one sample (0 = use the direct-gather kernels: identity
// geometry, or rectangles beyond FWD_PATCH_LDS_MAX)
static int fwd_tiles(const Geo& g, int n_patches, size_t* lds_out, int* stride_out) {
    if (g.h == g.H && g.w == g.W) return 0;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("CMS_LOSS_FWD_TILED");     // A/B switch, read once
        on = e ? (atoi(e) != 0) : 1;
    }
    if (!on) return 0;
    const size_t pf = patch_floats(g.c, g.sy, g.sx, FWD_TILE_H);
    const size_t lds = pf * n_patches * sizeof(float);
    if (lds > FWD_PATCH_LDS_MAX) return 0;
    if (lds_out) *lds_out = lds;
    if (stride_out) *stride_out = (int)pf;
    return ((g.W + TILE_W - 1) / TILE_W) * ((g.H + FWD_TILE_H - 1) / FWD_TILE_H) * g.n;
}
static int fwd_blocks(const Geo& g, int n_patches) {
    const int t = fwd_tiles(g, n_patches, nullptr, nullptr);
    return t > 0 ? t : fwd_grid((size_t)g.n * g.H * g.W);
}

// the fused forward + backward launches: one workgroup per 64 x TILE_H tile of one sample, all tiles in ONE launch (0 = not
// available: identity geometry, rectangles beyond the LDS, or the deterministic mode, whose colour-class launches stay unfused)
static int fused_tiles(const Geo& g, int n_patches) {
    if (g.h == g.H && g.w == g.W) return 0;
    if (loss_deterministic()) return 0;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("CMS_LOSS_FUSED");         // A/B switch, read once
        on = e ? (atoi(e) != 0) : 1;
    }
    if (!on) return 0;
    if (tile_lds_bytes(g.c, g.sy, g.sx, n_patches) > 160 * 1024 - 4096) return 0;
    return ((g.W + TILE_W - 1) / TILE_W) * ((g.H + TILE_H - 1) / TILE_H) * g.n;
}

#define CMS_DISPATCH_C(C, ...)                    \
    switch (C) {                                  \
        case 2: { constexpr int CT = 2; __VA_ARGS__; } break;   \
        case 5: { constexpr int CT = 5; __VA_ARGS__; } break;   \
        case 19: { constexpr int CT = 19; __VA_ARGS__; } break; \
        case 21: { constexpr int CT = 21; __VA_ARGS__; } break; \
        default: { constexpr int CT = 0; __VA_ARGS__; } break;  \
    }

}  // namespace cms

using namespace cms;

extern "C" size_t cms_consistency_workspace_bytes(const cms_consistency_desc* d) {
    if (!d) return 0;
    const Geo g = make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners);
    return (size_t)std::max(fwd_blocks(g, 3), fused_tiles(g, 3)) * 3 * sizeof(float);
}

extern "C" int cms_consistency_fused_supported(const cms_consistency_desc* d) {
    if (check_cons(d)) return 0;
    return fused_tiles(make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners), 3) > 0 ? 1 : 0;
}

extern "C" int cms_consistency_fwd_bwd(const cms_consistency_desc* d, float grad_unit, void* workspace, double* stats_out,
                                       float* grad_l_stu, void* stream) {
    int rc = check_cons(d);
    if (rc) return rc;
    CMS_REQUIRE(workspace && stats_out && grad_l_stu, "consistency_fwd_bwd: workspace / stats_out / grad NULL");
    ConsArgs a = make_cons_args(d);
    const int tiles = fused_tiles(a.g, 3);
    CMS_REQUIRE(tiles > 0, "consistency_fwd_bwd: not available for this geometry / mode (ask cms_consistency_fused_supported)");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = tile_lds_bytes(d->c, a.g.sy, a.g.sx, 3);
    const int pstride = (int)patch_floats(d->c, a.g.sy, a.g.sx, TILE_H);
    float* partials = (float*)workspace;
    CMS_DISPATCH_C(d->c, {
        // the default loss (`var`) as a compile-time constant: its own register allocation (see consistency_pixel_fwd_bwd)
        auto kern = d->loss_fn == CMS_LOSS_VAR ? cons_fused_tiled_kernel<CT, LOSS_VAR> : cons_fused_tiled_kernel<CT, -1>;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, s, a, grad_unit, grad_l_stu, pstride, partials);
    });
    hipLaunchKernelGGL((reduce_partials_kernel<3>), dim3(1), dim3(256), 0, s, partials, tiles, stats_out,
                       (double)((size_t)d->n * d->H * d->W), 3);
    return launch_status("cms_consistency_fwd_bwd");
}

extern "C" int cms_scale_by_scalar(float* x, long long n, const float* scalars, int index, float factor, void* stream) {
    CMS_REQUIRE(x && scalars && n >= 0 && index >= 0, "scale_by_scalar: bad argument");
    if (n == 0) return CMS_OK;
    const int grid = (int)std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, scalars, index, factor);
    return launch_status("cms_scale_by_scalar");
}

extern "C" int cms_consistency_fwd(const cms_consistency_desc* d, void* workspace, double* stats_out, void* stream) {
    int rc = check_cons(d);
    if (rc) return rc;
    CMS_REQUIRE(workspace && stats_out, "consistency_fwd: workspace / stats_out NULL");
    ConsArgs a = make_cons_args(d);
    const size_t P = (size_t)d->n * d->H * d->W;
    hipStream_t s = (hipStream_t)stream;
    const bool ident = d->h == d->H && d->w == d->W;
    float* partials = (float*)workspace;
    size_t lds = 0;
    int pstride = 0;
    const int tiles = fwd_tiles(a.g, 3, &lds, &pstride);
    const int grid = tiles > 0 ? tiles : fwd_grid(P);
    CMS_DISPATCH_C(d->c, {
        if (tiles > 0) {
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute((const void*)cons_fwd_tiled_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cons_fwd_tiled_kernel<CT>), dim3(grid), dim3(256), lds, s, a, partials, pstride);
        }
        else if (ident) hipLaunchKernelGGL((cons_fwd_kernel<CT, true>), dim3(grid), dim3(256), 0, s, a, partials);
        else hipLaunchKernelGGL((cons_fwd_kernel<CT, false>), dim3(grid), dim3(256), 0, s, a, partials);
    });
    hipLaunchKernelGGL((reduce_partials_kernel<3>), dim3(1), dim3(256), 0, s, partials, grid, stats_out, (double)P, 3);
    return launch_status("cms_consistency_fwd");
}

extern "C" int cms_consistency_finalize(const double* stats_local, const double* stats_global, float conf_thresh,
                                        int conf_per_pixel, float ramp_val, float cons_weight, float* scalars_out,
                                        void* stream) {
    CMS_REQUIRE(stats_local && stats_global && scalars_out, "consistency_finalize: NULL argument");
    hipLaunchKernelGGL(cons_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats_local, stats_global,
                       conf_thresh, conf_per_pixel, ramp_val, cons_weight, scalars_out);
    return launch_status("cms_consistency_finalize");
}

extern "C" int cms_loss_set_deterministic(int on) {
    cms::g_loss_deterministic = on ? 1 : 0;
    return CMS_OK;
}

extern "C" int cms_consistency_bwd(const cms_consistency_desc* d, const float* scalars, float* grad_l_stu,
                                   void* stream) {
    int rc = check_cons(d);
    if (rc) return rc;
    CMS_REQUIRE(scalars && grad_l_stu, "consistency_bwd: scalars / grad NULL");
    ConsArgs a = make_cons_args(d);
    hipStream_t s = (hipStream_t)stream;
    const bool ident = d->h == d->H && d->w == d->W;
    if (ident) {
        const int grid = fwd_grid((size_t)d->n * d->H * d->W);
        CMS_DISPATCH_C(d->c, {
            hipLaunchKernelGGL((cons_bwd_ident_kernel<CT>), dim3(grid), dim3(256), 0, s, a, scalars, grad_l_stu);
        });
    } else {
        const size_t lds = tile_lds_bytes(d->c, a.g.sy, a.g.sx, 3);
        const int pstride = (int)patch_floats(d->c, a.g.sy, a.g.sx, TILE_H);
        CMS_REQUIRE(lds <= 160 * 1024 - 4096, "consistency_bwd: %d classes at this scale need %zu B of LDS", d->c, lds);
        CMS_DISPATCH_C(d->c, {
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute((const void*)cons_bwd_tiled_kernel<CT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            tiled_launches(a.g, [&](const Geo& gc, int tiles) {
                ConsArgs ac = a;
                ac.g = gc;
                hipLaunchKernelGGL((cons_bwd_tiled_kernel<CT>), dim3(tiles), dim3(256), lds, s, ac, scalars, grad_l_stu, pstride);
            });
        });
    }
    return launch_status("cms_consistency_bwd");
}

static int check_ce(const cms_ce_desc* d) {
    CMS_REQUIRE(d != nullptr, "ce: null descriptor");
    CMS_REQUIRE(d->logits && d->labels, "ce: logits / labels NULL");
    CMS_REQUIRE(d->label_dtype == CMS_LABEL_U8 || d->label_dtype == CMS_LABEL_I64, "ce: bad label dtype");
    CMS_REQUIRE(d->n > 0 && d->c > 0 && d->h > 0 && d->w > 0 && d->H > 0 && d->W > 0, "ce: bad geometry");
    CMS_REQUIRE(d->h <= d->H && d->w <= d->W, "ce: logits larger than the label geometry");
    return CMS_OK;
}

extern "C" size_t cms_ce_workspace_bytes(const cms_ce_desc* d) {
    if (!d) return 0;
    const Geo g = make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners);
    return (size_t)std::max(fwd_blocks(g, 1), fused_tiles(g, 1)) * 2 * sizeof(float);
}

extern "C" int cms_ce_fused_supported(const cms_ce_desc* d) {
    if (check_ce(d)) return 0;
    return fused_tiles(make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners), 1) > 0 ? 1 : 0;
}

extern "C" int cms_ce_fwd_bwd(const cms_ce_desc* d, void* workspace, double* stats_out, float* grad_logits, void* stream) {
    int rc = check_ce(d);
    if (rc) return rc;
    CMS_REQUIRE(workspace && stats_out && grad_logits, "ce_fwd_bwd: workspace / stats_out / grad NULL");
    CeArgs a;
    a.d = *d;
    a.g = make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners);
    const int tiles = fused_tiles(a.g, 1);
    CMS_REQUIRE(tiles > 0, "ce_fwd_bwd: not available for this geometry / mode (ask cms_ce_fused_supported)");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = tile_lds_bytes(d->c, a.g.sy, a.g.sx, 1);
    float* partials = (float*)workspace;
    CMS_DISPATCH_C(d->c, {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute((const void*)ce_fused_tiled_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((ce_fused_tiled_kernel<CT>), dim3(tiles), dim3(256), lds, s, a, grad_logits, partials);
    });
    hipLaunchKernelGGL((reduce_partials_kernel<2>), dim3(1), dim3(256), 0, s, partials, tiles, stats_out, 0.0, -1);
    return launch_status("cms_ce_fwd_bwd");
}

extern "C" int cms_ce_fwd(const cms_ce_desc* d, void* workspace, double* stats_out, void* stream) {
    int rc = check_ce(d);
    if (rc) return rc;
    CMS_REQUIRE(workspace && stats_out, "ce_fwd: workspace / stats_out NULL");
    CeArgs a;
    a.d = *d;
    a.g = make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners);
    hipStream_t s = (hipStream_t)stream;
    const bool ident = d->h == d->H && d->w == d->W;
    float* partials = (float*)workspace;
    size_t lds = 0;
    const int tiles = fwd_tiles(a.g, 1, &lds, nullptr);
    const int grid = tiles > 0 ? tiles : fwd_grid((size_t)d->n * d->H * d->W);
    CMS_DISPATCH_C(d->c, {
        if (tiles > 0) {
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute((const void*)ce_fwd_tiled_kernel<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((ce_fwd_tiled_kernel<CT>), dim3(grid), dim3(256), lds, s, a, partials);
        }
        else if (ident) hipLaunchKernelGGL((ce_fwd_kernel<CT, true>), dim3(grid), dim3(256), 0, s, a, partials);
        else hipLaunchKernelGGL((ce_fwd_kernel<CT, false>), dim3(grid), dim3(256), 0, s, a, partials);
    });
    hipLaunchKernelGGL((reduce_partials_kernel<2>), dim3(1), dim3(256), 0, s, partials, grid, stats_out, 0.0, -1);
    return launch_status("cms_ce_fwd");
}

extern "C" int cms_ce_finalize(const double* stats, float loss_weight, float* scalars_out, void* stream) {
    CMS_REQUIRE(stats && scalars_out, "ce_finalize: NULL argument");
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, loss_weight, scalars_out);
    return launch_status("cms_ce_finalize");
}

extern "C" int cms_ce_bwd(const cms_ce_desc* d, const float* scalars, float* grad_logits, void* stream) {
    int rc = check_ce(d);
    if (rc) return rc;
    CMS_REQUIRE(scalars && grad_logits, "ce_bwd: scalars / grad NULL");
    CeArgs a;
    a.d = *d;
    a.g = make_geo(d->n, d->c, d->h, d->w, d->H, d->W, d->align_corners);
    hipStream_t s = (hipStream_t)stream;
    const bool ident = d->h == d->H && d->w == d->W;
    if (ident) {
        const int grid = fwd_grid((size_t)d->n * d->H * d->W);
        CMS_DISPATCH_C(d->c, {
            hipLaunchKernelGGL((ce_bwd_ident_kernel<CT>), dim3(grid), dim3(256), 0, s, a, scalars, grad_logits);
        });
    } else {
        const size_t lds = tile_lds_bytes(d->c, a.g.sy, a.g.sx, 1);
        CMS_REQUIRE(lds <= 160 * 1024 - 4096, "ce_bwd: %d classes at this scale need %zu B of LDS", d->c, lds);
        CMS_DISPATCH_C(d->c, {
            if (lds > 48 * 1024)
                (void)hipFuncSetAttribute((const void*)ce_bwd_tiled_kernel<CT>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            tiled_launches(a.g, [&](const Geo& gc, int tiles) {
                CeArgs ac = a;
                ac.g = gc;
                hipLaunchKernelGGL((ce_bwd_tiled_kernel<CT>), dim3(tiles), dim3(256), lds, s, ac, scalars, grad_logits);
            });
        });
    }
    return launch_status("cms_ce_bwd");
}
