// Launch programs: the native executor of the network passes.
//
// The reference drives its step from Python, one cuDNN call at a time (architectures/deeplab2.py:89-109 expands to ~450
// library ops per pass, train_seg_semisup_mask_mt.py:296-459 issues ~2 k launches per iteration). Round 1 of this build
// kept one Python -> ctypes round trip per convolution (descriptor filled in Python, ~30 us of host time per launch,
// 16 ms per iteration -- close to the 23 ms the GPU needed). Here the host side of a pass is recorded ONCE per input
// shape as a list of launch descriptors over persistent buffers -- convolutions, weight gradients, zero fills, stream
// fork / join points -- and replayed from C++: one call enqueues the whole pass on up to CMS_PROGRAM_MAX_STREAMS HIP
// streams (student || teacher, data gradients || weight gradients), cross-stream order through hipEventRecord /
// hipStreamWaitEvent. Nothing is allocated, nothing synchronises the host; the same call can be stream-captured
// into a hipGraph by the caller since all it does is launch work and record / wait events.
//
// Optional: every k-th convolution launch is bracketed with HIP events on its own stream (bench.py's `roofline`).
//
// Cross-stream order WITHOUT events (round 6, cms_program_set_sync_flags). hipEventRecord + hipStreamWaitEvent cost the
// PRODUCING stream ~11-13 us per pair of waiters (tools/event_cost_probe.hip: a 50 us kernel chain goes from 51.0 to 61.9-63.7 us
// per link; the marker packet drains the queue and signals through the host-visible path) -- the backward pass has one such point
// per bottleneck on its data-gradient chain (profiles/r06w_step_timeline.txt: 23 holes of ~12 us in layer 3 alone). With a flag
// buffer a sync is two one-wave kernels instead: `sync_set_kernel` on the producing stream (release store of a sequence number
// to the op's flag word, in order behind the producer) and `sync_wait_kernel` on the waiting stream (polls the word through L2
// with s_sleep, then ends: the next kernel of that stream starts behind it with the usual acquire): 53.0 us per link.
// Consecutive syncs from one point of a stream share ONE setter. No deadlock: the setter is enqueued before its waiter and
// whatever precedes it in its hardware queue was enqueued earlier still, i.e. depends on nothing behind the waiter; a waiter
// holds one wave. Stream capture (hipGraph) keeps the event form: a graph has no queue order to rely on.
// MEASURED IN THE STEP AND NOT THE DEFAULT: 626.8 / 628.9 img/s with flags against 634.2 / 633.8 with events at cfg 2, alternating legs
// on one box (profiles/r06ae_*). The hole on the data-gradient stream is not idle machine time -- the weight-gradient streams run in it
// -- and releasing them ~10 us earlier takes CUs from the chain's next launch. ops.Program hands the flags over only on request.
#include <cstdlib>
#include "common.hpp"
#include <new>
#include <vector>

namespace cms {

enum OpKind { OP_CONV = 0, OP_WGRAD = 1, OP_MEMSET = 2, OP_SYNC = 3, OP_ASPP_GATHER = 4, OP_ASPP_SPREAD = 5, OP_BN = 6, OP_WGRAD_GROUP = 7, OP_CHANNEL_SUM = 8, OP_WGRAD_FINISH = 9 };

struct AsppOp {            // arguments of cms_aspp_gather_fwd / cms_aspp_spread_bwd
    const float* src;      // z / dlogits
    const float* bias;
    void* dst;             // logits / D
    int d_dtype;
    int tap_dy[CMS_CONV_MAX_TAPS], tap_dx[CMS_CONV_MAX_TAPS];
    int n_taps, n, c, zc, h, w;
};

__global__ __launch_bounds__(64) void sync_set_kernel(int* flag, int v) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// polls until the flag has reached sequence number v (wrap-safe); gives up after ~4 s of wall clock (a lost setter must not hang
// the device: the word behind the flags then counts the timeouts -- tests read it)
__global__ __launch_bounds__(64) void sync_wait_kernel(const int* flag, int v, int* timeouts) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while ((int)((unsigned)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)v) < 0) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 400000000LL) {            // 100 MHz
            atomicAdd(timeouts, 1);
            return;
        }
    }
}

struct Op {
    int kind;
    int stream;        // OP_SYNC: the stream that WAITS
    int from;          // OP_SYNC: the stream whose work is waited for
    int group;         // interleaving key of cms_program_run_pair (bottleneck index)
    int f32;
    cms_conv_desc conv;
    cms_wgrad_desc wg;
    AsppOp aspp;
    cms_bn_op bn;
    void* ptr;
    size_t bytes;
    hipEvent_t ev;     // OP_SYNC
    int flag_slot;     // OP_SYNC: index of this op among the program's syncs = its word in the flag buffer
    double flops;
};

struct Timed {
    hipEvent_t e0, e1;
    double flops;
    bool used;
    bool head;         // the ASPP head convolution (fp32 NCHW logits): HBM-bound, accounted separately
};

}  // namespace cms

struct cms_program {
    std::vector<cms::Op> ops;
    int timing_every = 0;
    long conv_seen = 0;
    std::vector<cms::Timed> timed;
    size_t timed_used = 0;
    double acc_ms = 0.0, acc_flops = 0.0;
    long acc_launches = 0;
    long last_head = -1;       // index into `timed` of the head GEMM bracket the next gather op extends
    int last_head_stream = -1;
    hipEvent_t prev_sync_ev = nullptr;   // the event the PREVIOUS issued op recorded, if that op was a sync from `prev_sync_from`:
    void* prev_sync_from = nullptr;      // consecutive syncs from one stream (both weight-gradient streams waiting for the main
                                         // stream, once per bottleneck) share ONE event record (round 5)
    int n_syncs = 0;
    int* flags = nullptr;                // caller-owned, zero-initialised device words: one per sync op + the timeout counter
    int n_flags = 0;
    unsigned seq = 0;                    // sequence number of the last setter issued (a flag word only ever grows, modulo 2^32)
    int prev_flag_slot = -1;             // the setter the PREVIOUS issued op launched from `prev_sync_from` (shared by the next waiter)
    unsigned prev_flag_val = 0;
    bool use_flags = false;              // this replay: flags given, switched on, not capturing
};

using namespace cms;

static int issue(cms_program* p, Op& o, void* const* streams, int n_streams) {
    CMS_REQUIRE(o.stream >= 0 && o.stream < n_streams, "program: op on stream %d but only %d streams given", o.stream,
                n_streams);
    hipStream_t s = (hipStream_t)streams[o.stream];
    hipEvent_t shared_ev = nullptr;
    static int share_events = -1;
    if (share_events < 0) {
        // A/B switch, read once. OFF by default: measured SLOWER (629.3 / 630.8 against 634.9 / 635.9 img/s at cfg 2,
        // profiles/r05p_*) -- with one record both weight-gradient streams are released at the same instant
        const char* e = getenv("CMS_PROG_SHARE_EVENTS");
        share_events = e ? atoi(e) : 0;
    }
    if (share_events && o.kind == OP_SYNC && p->prev_sync_ev != nullptr && o.from >= 0 && o.from < n_streams &&
        p->prev_sync_from == streams[o.from])
        shared_ev = p->prev_sync_ev;         // nothing was issued since that record: the same point of the `from` stream
    static int share_flags = -1;
    if (share_flags < 0) {
        const char* e = getenv("CMS_PROG_FLAG_SHARE");      // A/B switch, read once: 0 = every waiter gets its own setter
        share_flags = e ? atoi(e) : 1;
    }
    const int shared_slot = (share_flags && o.kind == OP_SYNC && p->prev_flag_slot >= 0 && o.from >= 0 && o.from < n_streams &&
                             p->prev_sync_from == streams[o.from]) ? p->prev_flag_slot : -1;
    p->prev_sync_ev = nullptr;
    p->prev_flag_slot = -1;
    switch (o.kind) {
    case OP_CONV: {
        Timed* t = nullptr;
        const bool head = o.conv.y32 != nullptr;
        if (p->timing_every > 0 && (head || (p->conv_seen++ % p->timing_every) == 0)) {
            if (p->timed_used == p->timed.size()) {
                Timed nt;
                nt.used = false;
                nt.head = false;
                nt.flops = 0.0;
                if (hipEventCreate(&nt.e0) != hipSuccess || hipEventCreate(&nt.e1) != hipSuccess) {
                    set_error("program: hipEventCreate failed");
                    return CMS_ELAUNCH;
                }
                p->timed.push_back(nt);
            }
            t = &p->timed[p->timed_used++];
            t->flops = o.flops;
            t->used = true;
            t->head = head;
            (void)hipEventRecord(t->e0, s);
        }
        const int rc = o.f32 ? cms_conv_igemm_f32(&o.conv, s) : cms_conv_igemm(&o.conv, s);
        if (t) (void)hipEventRecord(t->e1, s);
        p->last_head = (t && head) ? (long)(t - &p->timed[0]) : -1;
        p->last_head_stream = o.stream;
        return rc;
    }
    case OP_ASPP_GATHER: {
        const AsppOp& g = o.aspp;
        const int rc = cms_aspp_gather_fwd(g.src, g.bias, (float*)g.dst, g.tap_dy, g.tap_dx, g.n_taps, g.n, g.c, g.zc, g.h,
                                           g.w, s);
        // the head = GEMM + gather: move the closing event of the GEMM's bracket behind the gather
        if (p->last_head >= 0 && p->last_head_stream == o.stream) (void)hipEventRecord(p->timed[p->last_head].e1, s);
        p->last_head = -1;
        return rc;
    }
    case OP_ASPP_SPREAD: {
        const AsppOp& g = o.aspp;
        return cms_aspp_spread_bwd(g.src, g.dst, g.d_dtype, g.tap_dy, g.tap_dx, g.n_taps, g.n, g.c, g.zc, g.h, g.w, s);
    }
    case OP_WGRAD:
        return o.f32 ? cms_conv_wgrad_f32(&o.wg, s) : cms_conv_wgrad(&o.wg, s);
    case OP_WGRAD_GROUP:        // ptr = the device-resident item table, bytes = items, from = grid size, f32 = kind
        return cms_conv_wgrad_group_run(o.ptr, (int)o.bytes, o.from, o.f32, s);
    case OP_CHANNEL_SUM:        // ptr = src, bytes = rows, from = channels, f32 = dtype code, aspp.dst = the fp32 sums
        return cms_channel_sum(o.ptr, o.f32, o.bytes, o.from, (float*)o.aspp.dst, s);
    case OP_WGRAD_FINISH:       // ptr = the device-resident item table, bytes = items, from = grid size
        return cms_wgrad_finish_run(o.ptr, (int)o.bytes, o.from, s);
    case OP_BN: {
        const cms_bn_op& b = o.bn;
        const int g = b.groups > 1 ? b.groups : 1;
        switch (b.what) {
        case 0:
            if (b.ws) return cms_bn_reduce_ws(b.x, nullptr, nullptr, b.dtype, nullptr, nullptr, b.sums, (size_t)b.n_pixels, b.c, g, 0, b.ws, s);
            return cms_bn_reduce(b.x, nullptr, nullptr, b.dtype, nullptr, nullptr, b.sums, (size_t)b.n_pixels, b.c, 0, s);
        case 1: return cms_bn_finalize_ex(b.sums, b.count, b.gamma, b.beta, b.eps, b.momentum, b.mean, b.rstd, b.scale, b.shift,
                                          b.running_mean, b.running_var, b.c, b.clear_a, b.clear_b, b.counter, s);
        case 2: return cms_bn_apply_groups_bits(b.x, b.res, b.y, b.dtype, b.scale, b.shift, b.relu, (size_t)b.n_pixels, b.c, g,
                                                (uint8_t*)b.mask_bits, s);
        case 3:
            if (b.ws && b.mask_bits)
                return cms_bn_reduce_ws_bits(b.x, b.dy, (const uint8_t*)b.mask_bits, b.dtype, b.mean, b.rstd, b.sums, (size_t)b.n_pixels,
                                             b.c, g, b.ws, s);
            if (b.ws) return cms_bn_reduce_ws(b.x, b.dy, b.y, b.dtype, b.mean, b.rstd, b.sums, (size_t)b.n_pixels, b.c, g, 1, b.ws, s);
            return cms_bn_reduce(b.x, b.dy, b.y, b.dtype, b.mean, b.rstd, b.sums, (size_t)b.n_pixels, b.c, 1, s);
        case 4: return cms_bn_bwd_apply_groups_bits(b.x, b.dy, b.y, (const uint8_t*)b.mask_bits, b.dx, b.dres, b.dtype, b.mean, b.rstd,
                                                    b.gamma, b.sums, b.count, (size_t)b.n_pixels, b.c, g, s);
        case 5: return cms_increment_counter((int64_t*)b.counter, s);
        case 6: return cms_bn_stats(b.x, b.dtype, (size_t)b.n_pixels, b.c, g, b.gamma, b.beta, b.eps, b.momentum, b.mean, b.rstd,
                                    b.scale, b.shift, b.running_mean, b.running_var, b.counter, b.sums, b.ws, s);
        case 7: return cms_bn_finalize_tiles((const float*)b.ws, b.reserved, (size_t)b.n_pixels, b.c, g, b.gamma, b.beta, b.eps, b.momentum,
                                             b.mean, b.rstd, b.scale, b.shift, b.running_mean, b.running_var, b.counter, s);
        case 8: return cms_bn_bwd_sums_tiles((const float*)b.ws, b.reserved, (size_t)b.n_pixels, b.c, g, b.sums, s);
        }
        set_error("program: unknown BatchNorm op %d", b.what);
        return CMS_EINVAL;
    }
    case OP_MEMSET:
        if (hipMemsetAsync(o.ptr, 0, o.bytes, s) != hipSuccess) {
            set_error("program: hipMemsetAsync failed");
            return CMS_ELAUNCH;
        }
        return CMS_OK;
    case OP_SYNC: {
        CMS_REQUIRE(o.from >= 0 && o.from < n_streams, "program: sync from stream %d but only %d streams given", o.from,
                    n_streams);
        if (streams[o.from] == streams[o.stream]) return CMS_OK;      // same stream: already ordered
        if (p->use_flags && o.flag_slot < p->n_flags - 1) {
            int slot = shared_slot;
            unsigned val = p->prev_flag_val;
            if (slot < 0) {                                           // a new point of the producing stream: its own setter
                slot = o.flag_slot;
                val = ++p->seq;
                hipLaunchKernelGGL(sync_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)streams[o.from], p->flags + slot, (int)val);
            }
            hipLaunchKernelGGL(sync_wait_kernel, dim3(1), dim3(64), 0, s, (const int*)(p->flags + slot), (int)val,
                               p->flags + (p->n_flags - 1));
            p->prev_flag_slot = slot;
            p->prev_flag_val = val;
            p->prev_sync_from = streams[o.from];
            return launch_status("program: flag sync");
        }
        if (shared_ev != nullptr) {
            if (hipStreamWaitEvent(s, shared_ev, 0) != hipSuccess) {
                set_error("program: event wait failed");
                return CMS_ELAUNCH;
            }
            p->prev_sync_ev = shared_ev;
            p->prev_sync_from = streams[o.from];
            return CMS_OK;
        }
        if (hipEventRecord(o.ev, (hipStream_t)streams[o.from]) != hipSuccess ||
            hipStreamWaitEvent(s, o.ev, 0) != hipSuccess) {
            set_error("program: event record / wait failed");
            return CMS_ELAUNCH;
        }
        p->prev_sync_ev = o.ev;
        p->prev_sync_from = streams[o.from];
        return CMS_OK;
    }
    }
    set_error("program: unknown op kind %d", o.kind);
    return CMS_EINVAL;
}

extern "C" int cms_program_create(cms_program** out) {
    CMS_REQUIRE(out != nullptr, "program_create: NULL out pointer");
    *out = new (std::nothrow) cms_program();
    CMS_REQUIRE(*out != nullptr, "program_create: out of host memory");
    return CMS_OK;
}

extern "C" int cms_program_destroy(cms_program* p) {
    if (!p) return CMS_OK;
    for (auto& o : p->ops)
        if (o.kind == OP_SYNC && o.ev) (void)hipEventDestroy(o.ev);
    for (auto& t : p->timed) {
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    delete p;
    return CMS_OK;
}

static int push(cms_program* p, const Op& o) {
    p->ops.push_back(o);
    return (int)p->ops.size() - 1;
}

extern "C" int cms_program_add_conv(cms_program* p, const cms_conv_desc* d, int f32, int stream_idx, int group) {
    CMS_REQUIRE(p && d, "program_add_conv: NULL pointer");
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_conv: stream index %d", stream_idx);
    Op o = {};
    o.kind = OP_CONV; o.stream = stream_idx; o.group = group; o.f32 = f32 ? 1 : 0;
    o.conv = *d;
    o.flops = 2.0 * (double)d->n * d->ho * d->wo * (double)d->cout * d->cin * d->ntaps;
    return push(p, o);
}

extern "C" int cms_program_add_wgrad(cms_program* p, const cms_wgrad_desc* d, int f32, int stream_idx, int group) {
    CMS_REQUIRE(p && d, "program_add_wgrad: NULL pointer");
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_wgrad: stream index %d", stream_idx);
    Op o = {};
    o.kind = OP_WGRAD; o.stream = stream_idx; o.group = group; o.f32 = f32 ? 1 : 0;
    o.wg = *d;
    o.flops = 2.0 * (double)d->n * d->ho * d->wo * (double)d->cout * d->cin * d->ntaps;
    return push(p, o);
}

extern "C" int cms_program_add_wgrad_group(cms_program* p, const void* table_dev, int n_items, int total_blocks, int kind, int stream_idx,
                                           int group) {
    CMS_REQUIRE(p && table_dev && n_items > 0 && total_blocks > 0 && (kind == 1 || kind == 2), "program_add_wgrad_group: bad arguments");
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_wgrad_group: stream index %d", stream_idx);
    Op o = {};
    o.kind = OP_WGRAD_GROUP; o.stream = stream_idx; o.group = group;
    o.ptr = const_cast<void*>(table_dev); o.bytes = (size_t)n_items; o.from = total_blocks; o.f32 = kind;
    return push(p, o);
}

extern "C" int cms_program_add_channel_sum(cms_program* p, const void* src, int dtype, size_t rows, int channels, float* dst,
                                           int stream_idx, int group) {
    CMS_REQUIRE(p && src && dst && rows > 0 && channels > 0 && channels % 64 == 0, "program_add_channel_sum: bad arguments");
    CMS_REQUIRE(dtype == CMS_F32 || dtype == CMS_BF16, "program_add_channel_sum: bad dtype");
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_channel_sum: stream index %d", stream_idx);
    Op o = {};
    o.kind = OP_CHANNEL_SUM; o.stream = stream_idx; o.group = group;
    o.ptr = const_cast<void*>(src); o.bytes = rows; o.from = channels; o.f32 = dtype; o.aspp.dst = dst;
    return push(p, o);
}

extern "C" int cms_program_add_wgrad_finish(cms_program* p, const void* items_dev, int n_items, int total_blocks, int stream_idx,
                                            int group) {
    CMS_REQUIRE(p && items_dev && n_items > 0 && total_blocks > 0, "program_add_wgrad_finish: bad arguments");
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_wgrad_finish: stream index %d", stream_idx);
    Op o = {};
    o.kind = OP_WGRAD_FINISH; o.stream = stream_idx; o.group = group;
    o.ptr = const_cast<void*>(items_dev); o.bytes = (size_t)n_items; o.from = total_blocks;
    return push(p, o);
}

extern "C" int cms_program_add_memset(cms_program* p, void* ptr, size_t bytes, int stream_idx, int group) {
    CMS_REQUIRE(p && ptr && bytes > 0, "program_add_memset: NULL pointer / empty range");
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_memset: stream index %d", stream_idx);
    Op o = {};
    o.kind = OP_MEMSET; o.stream = stream_idx; o.group = group; o.ptr = ptr; o.bytes = bytes;
    return push(p, o);
}

extern "C" int cms_program_add_sync(cms_program* p, int from_stream, int to_stream, int group) {
    CMS_REQUIRE(p, "program_add_sync: NULL program");
    CMS_REQUIRE(from_stream >= 0 && from_stream < CMS_PROGRAM_MAX_STREAMS && to_stream >= 0 &&
                    to_stream < CMS_PROGRAM_MAX_STREAMS, "program_add_sync: stream indices %d -> %d", from_stream, to_stream);
    Op o = {};
    o.kind = OP_SYNC; o.stream = to_stream; o.from = from_stream; o.group = group;
    o.flag_slot = p->n_syncs++;
    if (hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) {
        set_error("program_add_sync: hipEventCreate failed");
        return CMS_ELAUNCH;
    }
    return push(p, o);
}

static int add_aspp(cms_program* p, int kind, const float* src, const float* bias, void* dst, int d_dtype, const int* dy,
                    const int* dx, int n_taps, int n, int c, int zc, int h, int w, int stream_idx, int group) {
    CMS_REQUIRE(p && src && dst && dy && dx, "program_add_aspp: NULL pointer");
    CMS_REQUIRE(n_taps > 0 && n_taps <= CMS_CONV_MAX_TAPS, "program_add_aspp: 1..%d taps", CMS_CONV_MAX_TAPS);
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_aspp: stream index %d", stream_idx);
    Op o = {};
    o.kind = kind; o.stream = stream_idx; o.group = group;
    o.aspp.src = src; o.aspp.bias = bias; o.aspp.dst = dst; o.aspp.d_dtype = d_dtype;
    for (int i = 0; i < n_taps; ++i) { o.aspp.tap_dy[i] = dy[i]; o.aspp.tap_dx[i] = dx[i]; }
    o.aspp.n_taps = n_taps; o.aspp.n = n; o.aspp.c = c; o.aspp.zc = zc; o.aspp.h = h; o.aspp.w = w;
    return push(p, o);
}

extern "C" int cms_program_add_aspp_gather(cms_program* p, const float* z, const float* bias, float* logits, const int* tap_dy,
                                           const int* tap_dx, int n_taps, int n, int c, int zc, int h, int w, int stream_idx,
                                           int group) {
    return add_aspp(p, OP_ASPP_GATHER, z, bias, logits, CMS_F32, tap_dy, tap_dx, n_taps, n, c, zc, h, w, stream_idx, group);
}

extern "C" int cms_program_add_aspp_spread(cms_program* p, const float* dlogits, void* d_nhwc, int d_dtype, const int* tap_dy,
                                           const int* tap_dx, int n_taps, int n, int c, int zc, int h, int w, int stream_idx,
                                           int group) {
    return add_aspp(p, OP_ASPP_SPREAD, dlogits, nullptr, d_nhwc, d_dtype, tap_dy, tap_dx, n_taps, n, c, zc, h, w, stream_idx,
                    group);
}

extern "C" int cms_program_add_bn(cms_program* p, const cms_bn_op* op, int stream_idx, int group) {
    CMS_REQUIRE(p && op, "program_add_bn: NULL pointer");
    CMS_REQUIRE(op->what >= 0 && op->what <= 8, "program_add_bn: unknown op %d", op->what);
    CMS_REQUIRE((op->what != 7 && op->what != 8) || (op->ws && op->reserved > 0),
                "program_add_bn: finalize_tiles / sums_tiles need the tile sums (ws) and the tile rows");
    CMS_REQUIRE(op->what != 8 || op->sums, "program_add_bn: sums_tiles writes `sums`");
    CMS_REQUIRE(op->groups <= 1 || op->ws || (op->what != 0 && op->what != 3), "program_add_bn: grouped reductions need a workspace");
    CMS_REQUIRE(op->what != 3 || !op->mask_bits || op->ws, "program_add_bn: the backward reduction reads mask bits on the workspace kernels only");
    CMS_REQUIRE(stream_idx >= 0 && stream_idx < CMS_PROGRAM_MAX_STREAMS, "program_add_bn: stream index %d", stream_idx);
    Op o = {};
    o.kind = OP_BN; o.stream = stream_idx; o.group = group;
    o.bn = *op;
    return push(p, o);
}

extern "C" int cms_program_size(const cms_program* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int cms_program_sync_count(const cms_program* p) { return p ? p->n_syncs : 0; }

extern "C" int cms_program_set_sync_flags(cms_program* p, int* flags_dev, int n_flags) {
    CMS_REQUIRE(p, "program_set_sync_flags: NULL program");
    CMS_REQUIRE((flags_dev == nullptr && n_flags == 0) || (flags_dev != nullptr && n_flags >= 2),
                "program_set_sync_flags: a device buffer of >= 2 zeroed ints (one per sync op + the timeout counter), or NULL / 0");
    p->flags = flags_dev;
    p->n_flags = n_flags;
    return CMS_OK;
}

// flags given + CMS_PROG_FLAG_SYNC != 0 (read once) + the first stream is not being captured into a graph
static void begin_replay(cms_program* p, void* const* streams) {
    static int flag_sync = -1;
    if (flag_sync < 0) {
        const char* e = getenv("CMS_PROG_FLAG_SYNC");
        flag_sync = e ? atoi(e) : 1;        // (the Python side only hands flags over with CMS_PROG_FLAG_SYNC=1: measured slower)
    }
    p->prev_sync_ev = nullptr;               // (the host may have enqueued anything since the last call)
    p->prev_flag_slot = -1;
    p->use_flags = false;
    if (flag_sync != 0 && p->flags != nullptr && p->n_syncs > 0) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)streams[0], &st) != hipSuccess) {
            (void)hipGetLastError();
            st = hipStreamCaptureStatusNone;
        }
        p->use_flags = st == hipStreamCaptureStatusNone;
    }
}

extern "C" int cms_program_run(cms_program* p, int first, int last, void* const* streams, int n_streams) {
    CMS_REQUIRE(p && streams && n_streams > 0, "program_run: NULL program / streams");
    const int n = (int)p->ops.size();
    if (last < 0 || last > n) last = n;
    CMS_REQUIRE(first >= 0 && first <= last, "program_run: bad range [%d, %d) of %d ops", first, last, n);
    begin_replay(p, streams);
    for (int i = first; i < last; ++i) {
        const int rc = issue(p, p->ops[i], streams, n_streams);
        if (rc != CMS_OK) return rc;
    }
    return CMS_OK;
}

// Two programs issued interleaved, group by group (ops of `a` with group g, then ops of `b` with group g, ...): keeps
// two kernels in flight for the whole pass when the programs run on different streams (student || teacher).
extern "C" int cms_program_run_pair(cms_program* a, void* const* streams_a, int na, cms_program* b,
                                    void* const* streams_b, int nb) {
    CMS_REQUIRE(a && b && streams_a && streams_b && na > 0 && nb > 0, "program_run_pair: NULL program / streams");
    size_t ia = 0, ib = 0;
    const size_t ea = a->ops.size(), eb = b->ops.size();
    // EXPERIMENT (round 6, CMS_PAIR_SYNC=k, read once; 0 = off, the default): every k-th group boundary the FIRST streams of the
    // two programs wait for each other -- the two passes then walk their bottlenecks in phase (conv1 || conv1, conv2 || conv2,
    // expansion || expansion: launches of one kind share the machine better than a whole-CU eight-phase launch beside a
    // four-per-CU expansion). Events come from a small static pool (the pair is issued from one host thread).
    static int pair_sync = -1;
    static hipEvent_t sync_ev[64][2];
    static bool sync_ev_made = false;
    if (pair_sync < 0) {
        const char* e = getenv("CMS_PAIR_SYNC");
        pair_sync = e ? atoi(e) : 0;
    }
    if (pair_sync > 0 && !sync_ev_made) {
        for (auto& pr : sync_ev)
            for (auto& ev : pr)
                if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { set_error("program_run_pair: event"); return CMS_ELAUNCH; }
        sync_ev_made = true;
    }
    int groups_done = 0, ev_slot = 0;
    begin_replay(a, streams_a);
    begin_replay(b, streams_b);
    while (ia < ea || ib < eb) {
        // next group = the smaller of the two heads' groups (groups are recorded in non-decreasing order)
        const int ga = ia < ea ? a->ops[ia].group : 0x7fffffff;
        const int gb = ib < eb ? b->ops[ib].group : 0x7fffffff;
        const int g = ga < gb ? ga : gb;
        a->prev_sync_ev = nullptr;           // (the other program's ops were issued in between, possibly on shared streams)
        a->prev_flag_slot = -1;
        while (ia < ea && a->ops[ia].group <= g) {
            const int rc = issue(a, a->ops[ia++], streams_a, na);
            if (rc != CMS_OK) return rc;
        }
        b->prev_sync_ev = nullptr;
        b->prev_flag_slot = -1;
        while (ib < eb && b->ops[ib].group <= g) {
            const int rc = issue(b, b->ops[ib++], streams_b, nb);
            if (rc != CMS_OK) return rc;
        }
        if (pair_sync > 0 && (++groups_done % pair_sync) == 0 && ia < ea && ib < eb) {
            hipStream_t sa = (hipStream_t)streams_a[0], sb = (hipStream_t)streams_b[0];
            hipEvent_t* ev = sync_ev[ev_slot];
            ev_slot = (ev_slot + 1) % 64;
            if (hipEventRecord(ev[0], sa) != hipSuccess || hipEventRecord(ev[1], sb) != hipSuccess ||
                hipStreamWaitEvent(sa, ev[1], 0) != hipSuccess || hipStreamWaitEvent(sb, ev[0], 0) != hipSuccess) {
                set_error("program_run_pair: cross-stream sync failed");
                return CMS_ELAUNCH;
            }
        }
    }
    return CMS_OK;
}

extern "C" int cms_program_set_timing(cms_program* p, int every_k) {
    CMS_REQUIRE(p, "program_set_timing: NULL program");
    p->timing_every = every_k > 0 ? every_k : 0;
    return CMS_OK;
}

// Sums the event-bracketed convolution launches since the last call (waits for them to finish) and resets.
extern "C" int cms_program_read_timing(cms_program* p, double* sum_ms, double* sum_flops, long* launches,
                                       double* head_ms, long* head_launches) {
    CMS_REQUIRE(p, "program_read_timing: NULL program");
    double ms = 0.0, fl = 0.0, hms = 0.0;
    long n = 0, hn = 0;
    for (size_t i = 0; i < p->timed_used; ++i) {
        Timed& t = p->timed[i];
        if (!t.used) continue;
        float dt = 0.0f;
        if (hipEventSynchronize(t.e1) != hipSuccess || hipEventElapsedTime(&dt, t.e0, t.e1) != hipSuccess) {
            set_error("program_read_timing: event query failed");
            return CMS_ELAUNCH;
        }
        if (t.head) {
            hms += dt;
            ++hn;
        } else {
            ms += dt;
            fl += t.flops;
            ++n;
        }
        t.used = false;
    }
    p->timed_used = 0;
    if (sum_ms) *sum_ms = ms;
    if (sum_flops) *sum_flops = fl;
    if (launches) *launches = n;
    if (head_ms) *head_ms = hms;
    if (head_launches) *head_launches = hn;
    return CMS_OK;
}
