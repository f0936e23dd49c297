#!/usr/bin/env python
"""RCCL on a ONE-GPU box: backend "nccl" with world size 1. No scaling is measured here -- the point is that the code paths the
8-GPU run depends on have executed against RCCL itself at least once (every multi-rank test of this repository
runs on gloo, whose collectives on device tensors are host-synchronous, so stream-ordering mistakes cannot show there):

  1. init_process_group('nccl') + the first collective, then the side-stream probe (ops.probe_streams(again=True)) -- which streams
     clash with the queue RCCL's internal stream took;
  2. step.GradBuckets exactly as the fused step drives it: gradient slices written by kernels on a SIDE stream, `on_block` (async
     all_reduce) issued from that stream right behind the writer, `finish()` on the main stream, the optimizer-like reader on the main
     stream -- fp32 and through the bf16 staging arena; with one rank the sum is the identity, so the reader must see exactly what the
     writers wrote (a missing stream edge shows as stale data: the writers are slow kernels, the reader is launched immediately);
  3. the 32-byte loss-statistics all-reduce and an fp64 SyncBN-sized all-reduce issued between two launches on one stream;
  4. three iterations of the fused CutMix step of a small DeepLab v2 with the process group alive (exchange skipped at world 1; timing
     the headline with and without RCCL resident is bench.py's job: CMS_BENCH_FORCE_PG=1);
  5. the same iterations with the step told it has two ranks: the bucketed exchange then runs inside the recorded backward pass.
    python tools/rccl_single_rank.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29531')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch
import torch.distributed as dist

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group('nccl')
    probe = torch.ones(1, device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
except Exception as e:          # noqa: BLE001 -- RCCL cannot come up on this box (environment): exit code 77 = "not run", not "failed"
    print('RCCL did not come up: {}: {}'.format(type(e).__name__, e), flush=True)
    sys.exit(77)
assert int(probe.item()) == 1
print('1. RCCL up: backend {}, world {}'.format(dist.get_backend(), dist.get_world_size()), flush=True)
from cutmix_semisup_seg_amd import ops
from cutmix_semisup_seg_amd.step import GradBuckets
log = ops.probe_streams(dev, again=True)
print('   stream probe after the first collective: alone {} ms, chosen side streams {}, clash matrix {}'.format(log[1], log[3], log[2]), flush=True)

# ---- 2. bucketed exchange driven like the step
n = 40_000_000
grad = torch.zeros(n, dtype=torch.float32, device=dev)
offsets = [0, n // 8, n // 4, n // 2, 3 * n // 4]           # "bottlenecks" 0..4; buckets open at 4, 2, 1
side = ops.pooled_stream(dev, 'wgrad0')
main = torch.cuda.current_stream()
for dtype in ('fp32', 'bf16'):
    gb = GradBuckets(grad, offsets, [4, 2, 1], group=None, dtype=dtype, timing=True)
    for it in range(3):
        grad.zero_()
        torch.cuda.synchronize()
        gb.begin()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for bi in (4, 3, 2, 1, 0):
                lo = offsets[bi]
                hi = offsets[bi + 1] if bi + 1 < len(offsets) else n
                torch.cuda._sleep(2_000_000)                                     # ~1 ms: the writer is late
                grad[lo:hi].fill_(float(it * 10 + bi + 1))                       # this bottleneck's "weight gradients"
                gb.on_block(bi)
        main.wait_stream(side)
        gb.finish()
        got = [float(grad[offsets[bi]].item()) for bi in range(5)] + [float(grad[-1].item())]      # read on the main stream
        want = [float(it * 10 + bi + 1) for bi in range(5)] + [float(it * 10 + 5)]
        assert got == want, (dtype, it, got, want)
        full = torch.cat([torch.full((offsets[bi + 1] - offsets[bi] if bi < 4 else n - offsets[4],), float(it * 10 + bi + 1)) for bi in range(5)])
        assert torch.equal(grad.cpu(), full), (dtype, it)
    print('2. GradBuckets over RCCL ({}): 3 iterations exact; buckets (bytes, issue-to-wait ms): {}'.format(
        dtype, [(r['bytes'], round(r['issue_to_wait_ms'], 3)) for r in gb.read_timing()]), flush=True)

# ---- 3. small collectives between two launches of one stream
st = torch.zeros(4, dtype=torch.float32, device=dev)
s64 = torch.zeros(4 * 2 * 256, dtype=torch.float64, device=dev)
for it in range(5):
    torch.cuda._sleep(1_000_000)
    st.fill_(float(it))
    s64.fill_(float(it) + 0.5)
    dist.all_reduce(st)
    dist.all_reduce(s64)
    a, b = st * 2.0, s64 * 2.0
    assert float(a[0].item()) == 2.0 * it and float(b[-1].item()) == 2.0 * it + 1.0
print('3. 16-byte and fp64 statistics all-reduces between launches: exact', flush=True)

# ---- 4. the fused step with the process group alive; then the SAME iterations with the step told it has two ranks, so that the
# bucketed exchange really runs inside the recorded backward pass (hooks on the weight-gradient stream -> async all_reduce on RCCL ->
# finish() in front of the optimizer). With one rank the sum is the identity: the gradient arena must come out the same.
import numpy as np
from cutmix_semisup_seg_amd import optim as fo
from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
from architectures import deeplab2
import mask_gen
import optim_weight_ema
C, N, H, W = 5, 2, 65, 65


def run(pretend_world, allreduce_dtype='fp32'):
    torch.manual_seed(7)
    mk = lambda: deeplab2.ResNetDeepLab(deeplab2.Bottleneck, [2, 1, 2, 1], C, np.zeros(3), np.ones(3)).to(dev)
    stu, tea = mk(), mk()
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=1e-4), dict(params=list(stu.new_parameters()), lr=1e-3)])
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=0.3, deterministic=True, allreduce_dtype=allreduce_dtype))
    step.world = pretend_world
    step.time_buckets = pretend_world > 1
    g = torch.Generator(device=dev).manual_seed(3)
    grads = []
    for it in range(3):
        im = lambda: torch.randn(N, 3, H, W, generator=g, device=dev).bfloat16()
        y = torch.randint(0, C, (N, 1, H, W), generator=g, device=dev).to(torch.uint8)
        r = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(it))
        res = step(im(), y, [UnsupBatch(im(), ops.ranges_to_device(r, dev), x1_tea=im())])
        grads.append(opt.arena.grad.clone())
    torch.cuda.synchronize()
    assert np.isfinite(float(res['sup_loss']))
    return grads, float(res['sup_loss']), step.bucket_timing()


g1, loss1, _ = run(1)
print('4. fused step with RCCL resident: 3 iterations, sup_loss {:.4f}'.format(loss1), flush=True)
g2, loss2, tim = run(2)
# iteration 0 starts from identical weights: identical gradients (deterministic weight gradients); later iterations differ by the
# optimizer's 1 / world factor, which is the point of pretending -- only iteration 0 is compared
assert torch.equal(g1[0], g2[0]), float((g1[0] - g2[0]).abs().max())
print('5. bucketed exchange inside the recorded backward pass over RCCL: gradients of iteration 0 bit-identical to the run without '
      'exchange; buckets of the last step (bytes, issue-to-wait ms): {}'.format([(t['bytes'], round(t['issue_to_wait_ms'], 3)) for t in tim]),
      flush=True)
g3, _, _ = run(2, 'bf16')
err = float((g3[0] - g1[0]).abs().max() / (g1[0].abs().max() + 1e-30))
assert err <= 1e-2, err
print('   ... through the bf16 staging arena: max relative difference {:.2e}'.format(err), flush=True)
dist.destroy_process_group()
print('OK')
