"""
GPU: launch programs (csrc/program.hip, ops.Program) -- the recorded / replayed passes of the DeepLab v2 executor give
the results of the launch-by-launch path, stay correct across repeated replays with new inputs and new weights, refuse a
backward pass whose activations were overwritten, and the pipelined convolution variants agree with the default kernel.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _net(dtype, layers=(1, 1, 1, 1), C=5, programs=True):
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, list(layers), C, np.zeros(3), np.ones(3))
    net.load_state_dict(odl.closed_form_state(C, list(layers)))
    net = net.to(DEV)
    net.compute_dtype = dtype
    net.engine_kind = 'hip'
    net.train()
    net.freeze_batchnorm()
    net.hip_executor().use_programs = programs
    return net


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32], ids=['bf16', 'fp32'])
def test_program_replay_equals_eager_launches(dtype):
    g = torch.Generator().manual_seed(3)
    a, b = _net(dtype, programs=True), _net(dtype, programs=False)
    for it in range(3):                       # replay 0 records; replays 1, 2 reuse the buffers with new inputs
        x = torch.randn(2, 3, 65, 81, generator=g).to(DEV).to(dtype)
        gr = torch.randn(2, 5, 9, 11, generator=g).to(DEV)
        outs = []
        for net in (a, b):
            net._cms_arena.zero_grad()
            lo = net.forward_lowres(x)
            lo.backward(gr)
            outs.append((lo.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}))
        # same kernels in the same order; the head accumulates its tap groups with fp32 atomics (order-dependent ulps)
        torch.testing.assert_close(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-5)
        for k in outs[0][1]:
            ga, gb = outs[0][1][k], outs[1][1][k]
            assert float((ga - gb).abs().max()) <= 2e-3 * float(gb.abs().max()) + 1e-6, k    # fp32 atomics order
    ex = a.hip_executor()
    assert len(ex.programs()) == 2 and ex.issued['conv_launches'] > 0 and ex.issued['flops'] > 0
    # new weights reach the replayed program (operands are views of the arena, BN tables re-folded on demand)
    with torch.no_grad():
        for net in (a, b):
            net.layer5.conv2d_list[0].weight.mul_(0.5)
            net.bn1.running_var.mul_(1.3)
            for e in (net.hip_executor(),):
                e.invalidate()
    x = torch.randn(2, 3, 65, 81, generator=g).to(DEV).to(dtype)
    with torch.no_grad():
        la, lb = a.forward_lowres(x), b.forward_lowres(x)
    torch.testing.assert_close(la, lb, rtol=1e-5, atol=1e-5)
    assert float((la - outs[0][0]).abs().max()) > 1e-3               # ... and they did change the output


def test_backward_after_overwritten_activations_is_refused():
    net = _net(torch.bfloat16)
    x = torch.randn(1, 3, 33, 33).to(DEV).bfloat16()
    lo1 = net.forward_lowres(x)
    lo2 = net.forward_lowres(x)                       # same shape: the program's activation buffers are reused
    lo2.backward(torch.ones_like(lo2))                # fine: the latest pass
    with pytest.raises(RuntimeError, match='overwritten'):
        lo1.backward(torch.ones_like(lo1))


def test_pair_issue_matches_separate_passes():
    """student || teacher through cms_program_run_pair == the two passes one after the other."""
    from cutmix_semisup_seg_amd.backbone_hip import run_body_pair, run_body
    stu, tea = _net(torch.bfloat16), _net(torch.bfloat16)
    for p in tea.parameters():
        p.requires_grad = False
    g = torch.Generator().manual_seed(4)
    xs = torch.randn(2, 3, 65, 65, generator=g).to(DEV).bfloat16()
    xt = torch.randn(4, 3, 65, 65, generator=g).to(DEV).bfloat16()
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side), torch.no_grad():
        t_in = tea.stem_nhwc(xt)
    s_in = stu.stem_nhwc(xs)
    ls, lt = run_body_pair(stu.hip_executor(), s_in, tea.hip_executor(), t_in, side)
    main.wait_stream(side)
    with torch.no_grad():
        want_t = tea.forward_lowres(xt)
    want_s = run_body(stu.hip_executor(), s_in)
    torch.testing.assert_close(lt, want_t, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ls, want_s, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('tile', [0, 256])
@pytest.mark.parametrize('variant', [10, 11, 12, 13, 14])
def test_pipelined_conv_variants_equal_default_kernel(tile, variant):
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(variant)
    for (N, H, W, Cin, Cout, k, dil) in [(2, 41, 41, 256, 256, 3, 2), (3, 23, 29, 128, 384, 1, 1), (1, 9, 7, 64, 128, 3, 1)]:
        pad = dil * (k - 1) // 2
        x = torch.randn(N, H, W, Cin, generator=g, device=DEV).bfloat16()
        wp = (torch.randn(k * k, Cout, Cin, generator=g, device=DEV) * 0.05).bfloat16()
        scale = torch.rand(Cout, generator=g, device=DEV) + 0.5
        bias = torch.randn(Cout, generator=g, device=DEV) * 0.1
        res = torch.randn(N, H, W, Cout, generator=g, device=DEV).bfloat16()
        taps = ops.conv_taps(k, k, dil, pad)
        ref = ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True)
        for _ in range(3):
            out = ops.conv_igemm(x, wp, taps, scale=scale, bias=bias, res=res, relu=True, tile=tile, variant=variant)
            assert torch.equal(out, ref), (tile, variant, N, H, W, Cin, Cout, k)


def test_flag_syncs_order_the_streams_like_events():
    """(round 6) csrc/program.hip replays a cross-stream sync as a one-wave setter kernel + a one-wave polling kernel when the
    program has its flag words (ops.Program._ensure_sync_flags). A chain that ping-pongs between two streams -- every launch
    reads what the other stream's previous launch wrote, into buffers that hold stale values from the previous replay -- gives
    the single-stream result on every replay, and no waiter ever timed out."""
    from cutmix_semisup_seg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(4, 33, 33, 256, generator=g, device=DEV).bfloat16()
    ws = [(torch.randn(1, 256, 256, generator=g, device=DEV) * 0.06).bfloat16() for _ in range(8)]
    taps = ops.conv_taps(1, 1, 1, 0)
    bufs = [torch.zeros_like(x) for _ in range(9)]
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    prog = ops.Program()
    with ops.recording(prog, [main, side]):
        src = x
        for i, w in enumerate(ws):
            st = main if i % 2 == 0 else side
            other = side if i % 2 == 0 else main
            if i > 0:
                ops.stream_wait(st, other)
            with torch.cuda.stream(st):
                ops.conv_igemm(src, w, taps, relu=True, out=bufs[i])
            src = bufs[i]
        ops.stream_wait(main, side)
    assert int(ops.fn['cms_program_sync_count'](prog.h)) == len(ws)
    for rep in range(4):
        x.copy_(torch.randn(x.shape, generator=g, device=DEV).bfloat16())
        ref = x
        for w in ws:
            ref = ops.conv_igemm(ref, w, taps, relu=True)
        torch.cuda.synchronize()
        if rep == 1:
            prog._ensure_sync_flags(force=True)       # replay 0: events (the default); replays 1..3: flags
        prog.run([main, side])
        torch.cuda.synchronize()
        assert torch.equal(bufs[len(ws) - 1], ref), rep
    assert prog.sync_flags is not None and int(prog.sync_flags[:-1].max()) >= 3      # the setters ran (sequence numbers grow)
    assert prog.sync_timeouts() == 0
