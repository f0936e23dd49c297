// Teacher EMA and fused optimizer + EMA over flat fp32 parameter arenas (gfx950).
//
//   cms_ema_flat          optim_weight_ema.py:21-25   t.mul_(alpha); t.add_(s * (1 - alpha))  -- per tensor, 528 x 3
//                         launches + 528 temporaries in the reference; here one launch over the whole arena with the
//                         same three fp32 roundings (bit exact)
//   cms_adam_ema_step     torch.optim.Adam (train_seg_semisup_mask_mt.py:90-93, :465) + the EMA above (:466-467)
//   cms_sgd_ema_step      torch.optim.SGD with momentum / nesterov / weight decay (:94-98)
//
// The reference's parameter generator yields backbone conv weights 3x / 4x (architectures/deeplab2.py:208-230), so
// torch applies that many sequential updates per step with the same gradient and advances Adam's `step` by k;
// `k_updates` of the segment reproduces this inside one pass over memory.
//
// HBM-bound streaming: per element 5 fp32 reads (p, g, m, v, teacher) + 4 fp32 writes + 2 optional bf16 copies
// = 40 B; 16 B per lane per access; one workgroup per 2048-element chunk with wave-uniform segment parameters.
#include "common.hpp"

namespace cms {

constexpr int kMaxK = 8;

__global__ __launch_bounds__(256) void ema_flat_kernel(float* __restrict__ tgt, const float* __restrict__ src,
                                                       size_t count, float alpha, float oma,
                                                       uint16_t* __restrict__ tgt_bf16) {
    const size_t nvec = count / 4;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        float4 t = reinterpret_cast<float4*>(tgt)[v];
        const float4 s = reinterpret_cast<const float4*>(src)[v];
        t.x = ema_update(t.x, s.x, alpha, oma);
        t.y = ema_update(t.y, s.y, alpha, oma);
        t.z = ema_update(t.z, s.z, alpha, oma);
        t.w = ema_update(t.w, s.w, alpha, oma);
        reinterpret_cast<float4*>(tgt)[v] = t;
        if (tgt_bf16) {
            ushort4 b;
            b.x = f32_to_bf16(t.x); b.y = f32_to_bf16(t.y); b.z = f32_to_bf16(t.z); b.w = f32_to_bf16(t.w);
            reinterpret_cast<ushort4*>(tgt_bf16)[v] = b;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(count - nvec * 4)) {
        const size_t i = nvec * 4 + threadIdx.x;
        const float t = ema_update(tgt[i], src[i], alpha, oma);
        tgt[i] = t;
        if (tgt_bf16) tgt_bf16[i] = f32_to_bf16(t);
    }
}

struct AdamCoef {
    float step_size[kMaxK];   // lr / (1 - beta1^t)
    float bc2_sqrt[kMaxK];    // sqrt(1 - beta2^t)
};

template <bool ADAM>
__global__ __launch_bounds__(256) void optim_ema_kernel(cms_optim_desc d) {
    __shared__ AdamCoef coef;
    __shared__ float lr_sh;
    const uint32_t chunk = blockIdx.x;
    const cms_param_segment seg = d.segments[d.chunk_seg[chunk]];
    const uint64_t coff = d.chunk_off[chunk];
    const int k = seg.k_updates;
    const int64_t steps_done = *d.step_count;
    if (threadIdx.x == 0 && k > 0) {
        const double lr = d.lrs[seg.lr_group];
        lr_sh = (float)lr;
        if (ADAM) {
            for (int j = 0; j < k; ++j) {
                const double t = (double)(steps_done * k + j + 1);
                coef.step_size[j] = (float)(lr / (1.0 - pow(d.beta1, t)));
                coef.bc2_sqrt[j] = (float)sqrt(1.0 - pow(d.beta2, t));
            }
        }
    }
    __syncthreads();
    const bool first_step = steps_done == 0;
    const float b2 = (float)d.beta2, eps = (float)d.eps;
    const float w1 = (float)(1.0 - d.beta1), w2 = (float)(1.0 - d.beta2);
    const float lr_f = lr_sh;

#pragma unroll
    for (int rep = 0; rep < CMS_OPT_CHUNK / (256 * 4); ++rep) {
        const uint64_t e = coff + (uint64_t)(rep * 256 + threadIdx.x) * 4;
        if (e >= seg.count) continue;
        const uint64_t gidx = seg.offset + e;
        const int nvalid = (seg.count - e) >= 4 ? 4 : (int)(seg.count - e);
        float p[4], g[4], m[4], v[4], t[4];
        if (nvalid == 4) {
            float4 q = *reinterpret_cast<const float4*>(d.param + gidx);
            p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
            if (k > 0) {
                q = *reinterpret_cast<const float4*>(d.grad + gidx);
                g[0] = q.x; g[1] = q.y; g[2] = q.z; g[3] = q.w;
                q = *reinterpret_cast<const float4*>(d.slot0 + gidx);
                m[0] = q.x; m[1] = q.y; m[2] = q.z; m[3] = q.w;
                if (ADAM) {
                    q = *reinterpret_cast<const float4*>(d.slot1 + gidx);
                    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
                }
            }
            if (d.ema_param) {
                q = *reinterpret_cast<const float4*>(d.ema_param + gidx);
                t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
            }
        } else {
            for (int i = 0; i < 4; ++i) {
                const bool ok = i < nvalid;
                p[i] = ok ? d.param[gidx + i] : 0.0f;
                g[i] = (ok && k > 0) ? d.grad[gidx + i] : 0.0f;
                m[i] = (ok && k > 0) ? d.slot0[gidx + i] : 0.0f;
                v[i] = (ok && k > 0 && ADAM) ? d.slot1[gidx + i] : 0.0f;
                t[i] = (ok && d.ema_param) ? d.ema_param[gidx + i] : 0.0f;
            }
        }
        if (k > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float gi = g[i] * d.grad_scale;
                if (ADAM) {
                    for (int j = 0; j < k; ++j) {
                        m[i] = m[i] + w1 * (gi - m[i]);
                        v[i] = v[i] * b2 + (w2 * gi) * gi;
                        const float denom = sqrtf(v[i]) / coef.bc2_sqrt[j] + eps;
                        p[i] = p[i] - (coef.step_size[j] * m[i]) / denom;
                    }
                } else {
                    for (int j = 0; j < k; ++j) {
                        float dd = gi;
                        if (d.weight_decay != 0.0f) dd = gi + d.weight_decay * p[i];
                        if (d.momentum != 0.0f) {
                            m[i] = first_step ? dd : m[i] * d.momentum + dd;
                            dd = d.nesterov ? dd + d.momentum * m[i] : m[i];
                        }
                        p[i] = p[i] - lr_f * dd;
                    }
                }
            }
        }
        if (d.ema_param) {
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = ema_update(t[i], p[i], d.ema_alpha, d.ema_one_minus_alpha);
        }
        if (nvalid == 4) {
            if (k > 0) {
                *reinterpret_cast<float4*>(d.param + gidx) = make_float4(p[0], p[1], p[2], p[3]);
                *reinterpret_cast<float4*>(d.slot0 + gidx) = make_float4(m[0], m[1], m[2], m[3]);
                if (ADAM) *reinterpret_cast<float4*>(d.slot1 + gidx) = make_float4(v[0], v[1], v[2], v[3]);
                if (d.param_bf16) {
                    ushort4 b;
                    b.x = f32_to_bf16(p[0]); b.y = f32_to_bf16(p[1]); b.z = f32_to_bf16(p[2]); b.w = f32_to_bf16(p[3]);
                    *reinterpret_cast<ushort4*>(d.param_bf16 + gidx) = b;
                }
            }
            if (d.ema_param) {
                *reinterpret_cast<float4*>(d.ema_param + gidx) = make_float4(t[0], t[1], t[2], t[3]);
                if (d.ema_bf16) {
                    ushort4 b;
                    b.x = f32_to_bf16(t[0]); b.y = f32_to_bf16(t[1]); b.z = f32_to_bf16(t[2]); b.w = f32_to_bf16(t[3]);
                    *reinterpret_cast<ushort4*>(d.ema_bf16 + gidx) = b;
                }
            }
        } else {
            for (int i = 0; i < nvalid; ++i) {
                if (k > 0) {
                    d.param[gidx + i] = p[i];
                    d.slot0[gidx + i] = m[i];
                    if (ADAM) d.slot1[gidx + i] = v[i];
                    if (d.param_bf16) d.param_bf16[gidx + i] = f32_to_bf16(p[i]);
                }
                if (d.ema_param) {
                    d.ema_param[gidx + i] = t[i];
                    if (d.ema_bf16) d.ema_bf16[gidx + i] = f32_to_bf16(t[i]);
                }
            }
        }
    }
}

__global__ void increment_kernel(int64_t* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1;
}

// Frozen BatchNorm folded into the convolution epilogues' (scale, bias) for ALL layers of a network in one launch
// (architectures/deeplab2.py:92-107 with the statistics frozen): the operands are gathered from the flat parameter arena
// through element-index tables. Same operations and roundings as the tensor expression of the CPU oracle -- 1 / sqrt(var + eps)
// with a correctly rounded square root and division (ATen's CPU rsqrt; the GPU library's rsqrt is an approximation that may
// differ from it by an ulp), one product, then product and difference rounded separately (no contraction).
__global__ void bn_fold_kernel(const float* __restrict__ flat, const int64_t* __restrict__ iw, const int64_t* __restrict__ ib,
                               const int64_t* __restrict__ im, const int64_t* __restrict__ iv, int n, float eps,
                               float* __restrict__ scale, float* __restrict__ bias) {
    // (plain operators under `fp contract(off)`: HIP's __fmul_rn / __fsqrt_rn are a contractable product and the NATIVE square
    // root unless OCML's rounded operations are compiled in; sqrtf and / are correctly rounded by default)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float var_eps = flat[iv[i]] + eps;
        const float r = 1.0f / sqrtf(var_eps);
        const float sc = flat[iw[i]] * r;
        const float prod = flat[im[i]] * sc;
        scale[i] = sc;
        bias[i] = flat[ib[i]] - prod;
    }
}

static int check_optim(const cms_optim_desc* d) {
    CMS_REQUIRE(d != nullptr, "optim: null descriptor");
    CMS_REQUIRE(d->param && d->grad && d->slot0 && d->segments && d->chunk_seg && d->chunk_off && d->lrs &&
                    d->step_count,
                "optim: NULL pointer in descriptor");
    CMS_REQUIRE(d->n_chunks > 0, "optim: no chunks");
    CMS_REQUIRE(((uintptr_t)d->param % 16 == 0) && ((uintptr_t)d->grad % 16 == 0) && ((uintptr_t)d->slot0 % 16 == 0),
                "optim: arenas must be 16-byte aligned");
    return CMS_OK;
}

}  // namespace cms

using namespace cms;

extern "C" int cms_ema_flat(float* tgt, const float* src, size_t count, float alpha, float one_minus_alpha,
                            uint16_t* tgt_bf16_out, void* stream) {
    CMS_REQUIRE(tgt && src, "ema_flat: NULL pointer");
    CMS_REQUIRE(((uintptr_t)tgt % 16 == 0) && ((uintptr_t)src % 16 == 0), "ema_flat: arenas must be 16-byte aligned");
    if (count == 0) return CMS_OK;
    hipLaunchKernelGGL(ema_flat_kernel, dim3(grid_for(count / 4 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       tgt, src, count, alpha, one_minus_alpha, tgt_bf16_out);
    return launch_status("cms_ema_flat");
}

extern "C" int cms_adam_ema_step(const cms_optim_desc* d, void* stream) {
    int rc = check_optim(d);
    if (rc) return rc;
    CMS_REQUIRE(d->slot1, "adam: slot1 (exp_avg_sq) NULL");
    hipLaunchKernelGGL(optim_ema_kernel<true>, dim3(d->n_chunks), dim3(256), 0, (hipStream_t)stream, *d);
    return launch_status("cms_adam_ema_step");
}

extern "C" int cms_sgd_ema_step(const cms_optim_desc* d, void* stream) {
    int rc = check_optim(d);
    if (rc) return rc;
    hipLaunchKernelGGL(optim_ema_kernel<false>, dim3(d->n_chunks), dim3(256), 0, (hipStream_t)stream, *d);
    return launch_status("cms_sgd_ema_step");
}

extern "C" int cms_increment_counter(int64_t* counter, void* stream) {
    CMS_REQUIRE(counter, "increment_counter: NULL pointer");
    hipLaunchKernelGGL(increment_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter);
    return launch_status("cms_increment_counter");
}

extern "C" int cms_bn_fold(const float* flat, const int64_t* idx_weight, const int64_t* idx_bias, const int64_t* idx_mean,
                           const int64_t* idx_var, int n, float eps, float* scale, float* bias, void* stream) {
    CMS_REQUIRE(flat && idx_weight && idx_bias && idx_mean && idx_var && scale && bias, "bn_fold: NULL pointer");
    if (n <= 0) return CMS_OK;
    hipLaunchKernelGGL(bn_fold_kernel, dim3(grid_for((size_t)n, 256, 256)), dim3(256), 0, (hipStream_t)stream, flat, idx_weight,
                       idx_bias, idx_mean, idx_var, n, eps, scale, bias);
    return launch_status("cms_bn_fold");
}
