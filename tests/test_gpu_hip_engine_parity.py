"""
GPU parity of the hand-written convolution ENGINE (backbone_hip.py on csrc/conv.hip / csrc/conv_f32.hip) against the
pinned CPU oracle -- the gate VERDICT r1 asked for: the bf16 MFMA executor and the fp32 (f32-input MFMA) executor run
one whole CutMix mean-teacher iteration of the ResNet-101 DeepLab v2 at the geometry of BASELINE configs[1] (321 x 321,
21 classes) and configs[2] (512 x 1024, 19 classes) at small N, on identical bf16-rounded weights and inputs, and are
compared with oracle/step.py (train_seg_semisup_mask_mt.py:296-301, 354-367, 407-461): cross-entropy loss, consistency
loss (threshold at the median teacher confidence, so the rate is ~0.5, not 0), confidence rate, every weight gradient
against the oracle's AUTOGRAD (not the library's kernels), and the mIoU of the updated student.

Two configurations:
  * fp32  -- the PARITY configuration: north-star bar, 1e-4 relative on losses / IoU, asserted here;
  * bf16  -- the THROUGHPUT configuration: bf16 storage of activations cannot reach 1e-4 (one rounding of 2^-9 per
             layer, ~100 layers); its achieved errors are printed and bounded at what bf16 storage gives.
The fp32 kernels themselves are also checked op by op against an fp64 host reference.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    from cutmix_semisup_seg_amd import ops as _ops
    return _ops


def _pack(w):          # (Cout, Cin, kh, kw) -> (taps, Cout, Cin)
    co, ci, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci).contiguous()


def _rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


F32_CASES = [
    # name, N, H, W, Cin, Cout, k, stride, dil
    ('l3_conv3_1x1', 2, 41, 41, 256, 1024, 1, 1, 1),
    ('l3_conv2_3x3_d2', 1, 41, 41, 256, 256, 3, 1, 2),
    ('l4_conv2_3x3_d4', 1, 23, 29, 512, 512, 3, 1, 4),
    ('l2_conv1_1x1_s2', 2, 41, 41, 256, 128, 1, 2, 1),
    ('l1_conv1_c64', 2, 33, 47, 256, 64, 1, 1, 1),
    ('l1_conv2_3x3_c64', 1, 33, 47, 64, 64, 3, 1, 1),
    ('cout32_tail', 1, 7, 9, 64, 32, 3, 1, 1),
]


@pytest.mark.parametrize('case', F32_CASES, ids=[c[0] for c in F32_CASES])
@pytest.mark.parametrize('epi', ['plain', 'bn_res_relu'])
def test_conv_f32_forward_vs_fp64(ops, case, epi):
    name, N, H, W, Cin, Cout, k, stride, dil = case
    g = torch.Generator().manual_seed(abs(hash(name)) % 1000)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    full = epi != 'plain'
    scale = torch.rand(Cout, generator=g) + 0.5 if full else None
    bias = torch.randn(Cout, generator=g) * 0.1 if full else None
    res = torch.randn(N, Ho, Wo, Cout, generator=g) if full else None
    cu = lambda t: None if t is None else t.to(DEV)
    y = ops.conv_igemm(cu(x), cu(_pack(w)), ops.conv_taps(k, k, dil, pad), stride=stride, out_hw=(Ho, Wo),
                       scale=cu(scale), bias=cu(bias), res=cu(res), relu=full)
    assert y.dtype == torch.float32 and tuple(y.shape) == (N, Ho, Wo, Cout)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, stride, pad, dil)
    if full:
        ref = F.relu(ref * scale.double().view(1, -1, 1, 1) + bias.double().view(1, -1, 1, 1) + res.double().permute(0, 3, 1, 2))
    err = _rel(y.permute(0, 3, 1, 2), ref)
    assert err <= 2e-6, err


@pytest.mark.parametrize('case', F32_CASES[:5], ids=[c[0] for c in F32_CASES[:5]])
def test_conv_f32_dgrad_and_wgrad_vs_fp64_autograd(ops, case):
    name, N, H, W, Cin, Cout, k, stride, dil = case
    g = torch.Generator().manual_seed(abs(hash(name)) % 1000 + 7)
    pad = dil * (k - 1) // 2
    x = torch.randn(N, Cin, H, W, generator=g).double().requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).double().requires_grad_(True)
    scale = (torch.rand(Cout, generator=g) + 0.5)
    y = F.conv2d(x, w, None, stride, pad, dil) * scale.double().view(1, -1, 1, 1)
    du = torch.randn(y.shape, generator=g)
    y.backward(du.double())
    Ho, Wo = y.shape[2], y.shape[3]
    cu = lambda t: t.to(DEV)
    taps = ops.conv_taps(k, k, dil, pad)
    x_nhwc = cu(x.detach().float().permute(0, 2, 3, 1).contiguous())
    du_nhwc = cu(du.permute(0, 2, 3, 1).contiguous())
    wp = cu(_pack(w.detach().float()))
    # weight gradient (BN scale folded, accumulated into a pre-filled buffer)
    dw = torch.full((k * k, Cout, Cin), 0.5, dtype=torch.float32, device=DEV)
    ops.conv_wgrad(du_nhwc, x_nhwc, taps, dw, stride=stride, scale=cu(scale))
    want_w = _pack(w.grad.float()) + 0.5
    assert _rel(dw, want_w) <= 5e-6
    # data gradient = forward kernel on the transposed, scale-folded weights with negated taps (+ ReLU mask + add)
    wT = ops.conv_pack_transpose(wp, scale=cu(scale), flip=False, out_dtype=torch.float32)
    mask = torch.randn(N, H, W, Cin, generator=g)
    add = torch.randn(N, H, W, Cin, generator=g)
    neg = [(-dy, -dx) for dy, dx in taps]
    if stride == 1:
        dx_ = ops.conv_igemm(du_nhwc, wT, neg, res=cu(add), mode=1, mask_src=cu(mask))
    else:
        # residual / mask are indexed like the (full-size) output; the strided scatter visits the even positions only
        dx_ = ops.conv_igemm(du_nhwc, wT, neg, res=cu(add), mode=1, mask_src=cu(mask), out_hw=(Ho, Wo),
                             out_stride=stride, out_full_hw=(H, W))
    want = x.grad.float().permute(0, 2, 3, 1)
    if stride == 1:
        want = (want + add) * (mask > 0)
        assert _rel(dx_, want) <= 5e-6
    else:
        sub = (want[:, ::stride, ::stride] + add[:, ::stride, ::stride]) * (mask[:, ::stride, ::stride] > 0)
        assert _rel(dx_[:, ::stride, ::stride], sub) <= 5e-6
        rest = dx_.clone()
        rest[:, ::stride, ::stride] = 0
        assert float(rest.abs().max()) == 0.0          # a 1x1 stride-2 conv never reads the other pixels


def test_aspp_head_f32_ksplit_and_padded_classes(ops):
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 2, 17, 19, 21
    x = torch.randn(N, H, W, 2048, generator=g)
    ws = [torch.randn(C, 2048, 3, 3, generator=g) * 0.01 for _ in range(2)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
    wpad = torch.zeros(18, 32, 2048)
    taps = []
    for i, d in enumerate((6, 12)):
        wpad[9 * i:9 * i + 9, :C] = _pack(ws[i])
        taps += ops.conv_taps(3, 3, d, d)
    bias = torch.zeros(32)
    bias[:C] = bs[0] + bs[1]
    out = torch.zeros(N, C, H, W, device=DEV)
    ops.conv_igemm(x.to(DEV), wpad.to(DEV), taps, bias=bias.to(DEV), out_f32_nchw=out, cout_real=C, ksplit=6)
    xd = x.double().permute(0, 3, 1, 2)
    ref = sum(F.conv2d(xd, ws[i].double(), bs[i].double(), 1, d, d) for i, d in enumerate((6, 12)))
    assert _rel(out, ref) <= 5e-6


# ================================================================================ whole iteration vs oracle/step.py
def _state(layers, C, seed=1234):
    """Seeded weights with a healthy signal and gradient flow through 101 layers: He-initialised convolutions, BN scale
    in [0.6, 1.4] (x 0.2 on the last BatchNorm of every bottleneck, so that the residual stream keeps O(1) activations
    through 33 blocks), small running statistics, a head scaled to logits of std ~2.5 (class probabilities spread over
    (0.2, 1): the confidence threshold has something to cut). Convolution weights are rounded to bf16 so that the bf16
    operand copy, the fp32 master and the oracle hold IDENTICAL values."""
    from oracle import deeplab2 as odl
    g = torch.Generator().manual_seed(seed)
    st = {}
    for k, (shape, dt) in odl.state_spec(C, layers).items():
        if dt == torch.int64:
            st[k] = torch.zeros(shape, dtype=torch.int64)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            amp = (2.0 / fan_in) ** 0.5 * (0.3 if k.startswith('layer5.') else 1.0)
            st[k] = (torch.randn(shape, generator=g) * amp).bfloat16().float()
        elif k.endswith('running_var'):
            st[k] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif k.endswith('running_mean'):
            st[k] = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith('.weight'):
            st[k] = (0.6 + 0.8 * torch.rand(shape, generator=g)) * (0.2 if '.bn3.' in k else 1.0)
        else:
            st[k] = 0.1 * torch.randn(shape, generator=g)
    return st


GEOMETRIES = {
    'cfg2_321x321_c21': dict(N=2, H=321, W=321, C=21),
    'cfg3_512x1024_c19': dict(N=1, H=512, W=1024, C=19),
}
LAYERS = [3, 4, 23, 3]
LR = 3e-5
_ORACLE = {}


def _problem(name):
    """Inputs + the oracle's iteration for one geometry (computed once, shared by the bf16 and fp32 runs)."""
    if name in _ORACLE:
        return _ORACLE[name]
    from oracle import deeplab2 as odl, step as ostep, boxmask as obox, evaluation as oev
    import mask_gen
    geo = GEOMETRIES[name]
    N, H, W, C = geo['N'], geo['H'], geo['W'], geo['C']
    g = torch.Generator().manual_seed(77)
    rnd = lambda: torch.randn(N, 3, H, W, generator=g).bfloat16().float()
    x, ux0, ux1 = rnd(), rnd(), rnd()
    y = torch.randint(0, C, (N, 1, H, W), generator=g)
    y[torch.rand(N, 1, H, W, generator=g) < 0.05] = 255
    ranges = mask_gen.BoxMaskGenerator(0.5, invert=True).generate_ranges(N, (H, W), rng=np.random.RandomState(9))
    masks = torch.tensor(obox.rasterise(ranges, (H, W), True).astype(np.float32))
    ones = torch.ones(N, 1, H, W)
    st = _state(LAYERS, C)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        conf = torch.softmax(odl.forward(ux0, st, LAYERS, frozen=True), dim=1).max(dim=1)[0]
    tau = float(conf.median())
    S = ostep.StepState(st, C, LAYERS, opt='adam', lr=LR, teacher_alpha=0.99)
    grads = {}
    ref = ostep.train_iteration(S, x, y, ux0, ux1, ones, ones, masks, conf_thresh=tau, grads_out=grads)
    # IoU of the UPDATED student on the supervised images; truth = the oracle's own prediction with 30 % random flips
    with torch.no_grad():
        pred = odl.forward(x, S.student, LAYERS, frozen=True).argmax(dim=1)
    truth = pred.clone()
    flip = torch.rand(truth.shape, generator=g) < 0.3
    truth[flip] = torch.randint(0, C, truth.shape, generator=g)[flip]
    truth[y[:, 0] == 255] = 255
    acc = oev.IoUAccumulator(C)
    for i in range(N):
        acc.sample(truth[i].numpy(), pred[i].numpy(), ignore_value=255)
    out = dict(x=x, y=y, ux0=ux0, ux1=ux1, ranges=ranges, st=st, tau=tau, ref=ref, grads=grads, truth=truth,
               miou=float(acc.score().mean()), geo=geo)
    _ORACLE[name] = out
    return out


def _device_iteration(ops, P, dtype, taps=None):
    from architectures import deeplab2
    import optim_weight_ema
    import evaluation
    from cutmix_semisup_seg_amd import optim as fo
    from cutmix_semisup_seg_amd.step import CutMixMeanTeacherStep, StepConfig, UnsupBatch
    geo = P['geo']
    C, H, W = geo['C'], geo['H'], geo['W']
    mk = lambda: deeplab2.ResNetDeepLab(deeplab2.Bottleneck, LAYERS, C, np.zeros(3), np.ones(3))
    stu, tea = mk(), mk()
    stu.load_state_dict(P['st'])
    stu, tea = stu.to(DEV), tea.to(DEV)
    stu.compute_dtype = tea.compute_dtype = dtype
    stu.engine_kind = tea.engine_kind = 'hip'           # fail loudly if the hand-written engine is not the one running
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=LR * 0.1),
                             dict(params=list(stu.new_parameters()), lr=LR)])
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
    step = CutMixMeanTeacherStep(stu, tea, opt, ema, StepConfig(conf_thresh=P['tau'], compute_dtype=dtype))
    if taps is not None:
        stu.hip_executor().debug_capture = {}        # the recorded backward pass keeps the dC buffer of every bottleneck
    cu = lambda t: t.to(DEV).to(dtype)
    ub = UnsupBatch(cu(P['ux0']), ops.ranges_to_device(P['ranges'], DEV), x1_tea=cu(P['ux1']))
    r = step(cu(P['x']), P['y'].to(torch.uint8).to(DEV), [ub])
    got = {k: float(v) for k, v in r.items()}
    ex = stu.hip_executor()
    assert ex.dtype == dtype and stu._hip_executor is ex
    grads = {k: p.grad.detach().float().cpu().clone() for k, p in stu.named_parameters() if p.grad is not None}
    if taps is not None:
        # the recorded student pass over [x_sup; x_mix] keeps (input, a1, a2) of every bottleneck + the layer4 output
        progs = [p_ for k_, p_ in ex._programs.items() if k_[0] == 'fwd' and k_[2]]
        assert len(progs) == 1
        sv = progs[0].saved
        taps['blocks'] = [t[0].float().cpu() for t in sv[:-1]] + [sv[-1].float().cpu()]      # NHWC, batch [sup; mix]
        taps['a1'] = [t[1].float().cpu() for t in sv[:-1]]
        taps['a2'] = [t[2].float().cpu() for t in sv[:-1]]
        taps['logits'] = progs[0].logits.float().cpu()
        bw = list(progs[0].bwd.values())
        assert len(bw) == 1
        taps['dlg'], taps['dx'] = bw[0].dlg.float().cpu(), bw[0].dx.float().cpu()
        taps['dC'] = {bi: t.float().cpu() for bi, t in ex.debug_capture.items()}
        taps['x_stu'] = torch.cat([cu(P['x']), ops.cutmix_paste(cu(P['ux0']), cu(P['ux1']), ranges=ub.ranges, invert=True)]).float().cpu()
    stu.eval()
    with torch.no_grad():
        lo = stu.forward_lowres(cu(P['x']))
    ev = evaluation.EvaluatorIoU(C)
    ev.sample_logits(lo, P['truth'].to(torch.uint8).to(DEV), (H, W), ignore_value=255, align_corners=True)
    got['miou'] = float(ev.score().mean())
    return got, grads


def _errors(P, got, grads):
    ref = P['ref']
    e = dict(sup_loss=abs(got['sup_loss'] - ref['sup_loss']) / abs(ref['sup_loss']),
             consistency_loss=abs(got['consistency_loss'] - ref['consistency_loss']) / abs(ref['consistency_loss']),
             conf_rate=abs(got['conf_rate'] - ref['conf_rate']),
             miou=abs(got['miou'] - P['miou']))
    ge = {}
    for k, want in P['grads'].items():
        if want is None:
            continue
        ge[k] = _rel(grads[k], want)
    e['grad_max'] = max(ge.values())
    e['grad_mean'] = float(np.mean(list(ge.values())))
    e['grad_worst_key'] = max(ge, key=ge.get)
    e['grad_head'] = ge['layer5.conv2d_list.0.weight']
    e['grad_stem'] = ge['conv1.weight']
    return e


@pytest.mark.parametrize('name', sorted(GEOMETRIES))
def test_fp32_hip_engine_iteration_matches_oracle_to_1e4(ops, name):
    """PARITY configuration: north-star bar (BASELINE.json: within 1e-4 relative on losses / IoU)."""
    P = _problem(name)
    got, grads = _device_iteration(ops, P, torch.float32)
    e = _errors(P, got, grads)
    print('\nPARITY fp32 HIP engine vs oracle [{}] tau={:.4f} ref={} got={} errors={}'.format(name, P['tau'], P['ref'], got, e))
    assert 0.2 < P['ref']['conf_rate'] < 0.8                      # the consistency term really is switched on
    assert e['sup_loss'] <= 1e-4 and e['consistency_loss'] <= 1e-4, e
    assert e['conf_rate'] <= 1e-4, e
    assert e['miou'] <= 1e-4 * max(P['miou'], 1e-3) + 2e-5, e
    assert e['grad_max'] <= 2e-3 and e['grad_mean'] <= 2e-4, e    # every trainable tensor vs the oracle's autograd


@pytest.mark.parametrize('name', sorted(GEOMETRIES))
def test_bf16_hip_engine_iteration_vs_oracle_reports_errors(ops, name):
    """THROUGHPUT configuration: bf16 storage of activations. The bounds are what bf16 storage gives (measured values
    are printed and recorded in DESIGN.md section 2); the 1e-4 bar is met by the fp32 configuration above."""
    P = _problem(name)
    got, grads = _device_iteration(ops, P, torch.bfloat16)
    e = _errors(P, got, grads)
    print('\nPARITY bf16 HIP engine vs oracle [{}] tau={:.4f} ref={} got={} errors={}'.format(name, P['tau'], P['ref'], got, e))
    # measured (MI355X, profiles/r02d_*): CE loss 7e-5 / 9e-5, confidence rate 3e-4 / 9e-5, head gradient 5e-4 / 2e-4;
    # the 'var' consistency loss is a mean of SQUARED probability differences (~4e-3), bf16 noise adds a positive bias to
    # it: 2.5e-3 (cfg 2) / 1.2e-2 (cfg 3) relative
    assert e['sup_loss'] <= 1e-3 and e['consistency_loss'] <= 3e-2, e
    assert e['conf_rate'] <= 2e-3, e
    # the truth map of this check is the ORACLE'S OWN argmax (with flips) on a random-initialised network whose classes
    # are nearly tied everywhere: every argmax the bf16 activations flip costs IoU. Measured 0.059 (cfg 2) / 0.021
    # (cfg 3); the trained-network statement (within 0.2 pt) is tests/test_gpu_miou_training.py
    assert e['miou'] <= 1e-1, e
    assert e['grad_head'] <= 2e-2 and e['grad_mean'] <= 3e-2 and e['grad_max'] <= 1.5e-1, e


_ORACLE16 = {}


def _problem_bf16(name):
    """The SAME iteration on the SAME inputs through the bf16-storage restatement (oracle/deeplab2_chain.py): what the
    timed configuration is held to. The threshold is the fp32 problem's (median teacher confidence)."""
    if name in _ORACLE16:
        return _ORACLE16[name]
    from oracle import deeplab2_chain as och, step as ostep, boxmask as obox, evaluation as oev, losses as OL
    P = _problem(name)
    geo = P['geo']
    N, H, W, C = geo['N'], geo['H'], geo['W'], geo['C']
    masks = torch.tensor(obox.rasterise(P['ranges'], (H, W), True).astype(np.float32))
    ones = torch.ones(N, 1, H, W)
    S = ostep.StepState(P['st'], C, LAYERS, opt='adam', lr=LR, teacher_alpha=0.99)
    grads, taps = {}, {}
    ref = ostep.train_iteration(S, P['x'], P['y'], P['ux0'], P['ux1'], ones, ones, masks, conf_thresh=P['tau'],
                                grads_out=grads, storage='bf16', taps_out=taps)
    with torch.no_grad():
        lo = och.Chain(S.student, C, LAYERS, 'bf16').forward(P['x'], save=False)[0]
        pred = OL.upsample(lo, (H, W)).argmax(dim=1)
    g = torch.Generator().manual_seed(78)
    truth = pred.clone()
    flip = torch.rand(truth.shape, generator=g) < 0.3
    truth[flip] = torch.randint(0, C, truth.shape, generator=g)[flip]
    truth[P['y'][:, 0] == 255] = 255
    acc = oev.IoUAccumulator(C)
    for i in range(N):
        acc.sample(truth[i].numpy(), pred[i].numpy(), ignore_value=255)
    out = dict(P)
    out.update(ref=ref, grads=grads, truth=truth, miou=float(acc.score().mean()), taps=taps)
    _ORACLE16[name] = out
    return out


@pytest.mark.parametrize('name', sorted(GEOMETRIES))
def test_bf16_hip_engine_iteration_matches_the_bf16_storage_oracle(ops, name):
    """THROUGHPUT (timed) configuration against the oracle WITH the storage model: same algorithm, bf16 rounding where
    the fused epilogues store -- what is left is fp32 summation order (and the rare bf16 tie it flips). Every loss, the
    rate, the IoU and EVERY gradient tensor are bounded >= 10x tighter than against the fp32 oracle;
    a wrong tap / epilogue / mask in any one of the 104 layers moves its own gradient tensor by O(1)."""
    P = _problem_bf16(name)
    taps = {}
    got, grads = _device_iteration(ops, P, torch.bfloat16, taps=taps)
    e = _errors(P, got, grads)
    ge = {k: _rel(grads[k], w) for k, w in P['grads'].items() if w is not None}
    worst = sorted(ge.items(), key=lambda kv: -kv[1])[:5]
    print('\nPARITY bf16 HIP engine vs bf16-storage oracle [{}] ref={} got={} errors={} worst gradients={}'.format(
        name, P['ref'], got, e, worst))
    # per-block curve: input of every bottleneck (and the layer4 output), device vs bf16-storage oracle
    n = P['geo']['N']
    curve = []
    for i, dev_t in enumerate(taps['blocks']):
        want = torch.cat([P['taps']['sup'][i], P['taps']['mix'][i]], dim=0).permute(0, 2, 3, 1)
        assert tuple(dev_t.shape) == tuple(want.shape)
        curve.append((_rel(dev_t, want), float((dev_t != want).float().mean())))
    print('PER-BLOCK bf16 engine vs bf16-storage oracle [{}] (relative error, fraction of differing elements): {}'.format(
        name, ' '.join('{}:{:.1e}/{:.1e}'.format(i, a, b) for i, (a, b) in enumerate(curve))))
    # The two pipelines agree almost bit for bit after the stem and then DECORRELATE: one bf16 tie flipped by fp32
    # summation order perturbs every later tensor by as much as the storage noise itself (measured: 1e-5 of the elements
    # differ after the stem, 30 % after 8 bottlenecks, 63 % at the end; relative distance 9.7e-3 = the distance of either
    # from the fp32 oracle). Whole-network quantities are therefore two samples of the same noise: the cross-entropy loss
    # (a mean over all pixels) meets the 1e-4 bar, the rest is bounded at the noise level -- the tight per-layer statement
    # is test_bf16_engine_every_layer_teacher_forced_vs_bf16_storage_oracle below.
    assert curve[0][0] <= 2e-4 and curve[0][1] <= 2e-4 and curve[-1][0] <= 2e-2, curve
    assert e['sup_loss'] <= 1e-4 and e['consistency_loss'] <= 3e-2 and e['conf_rate'] <= 2e-3, e
    assert e['miou'] <= 3e-2, e
    assert e['grad_head'] <= 2e-3 and e['grad_mean'] <= 1.5e-2 and e['grad_max'] <= 8e-2, e


@pytest.mark.parametrize('name', sorted(GEOMETRIES))
def test_bf16_engine_every_layer_teacher_forced_vs_bf16_storage_oracle(ops, name):
    """The tight statement about the TIMED configuration. Whole-network comparisons of two bf16 pipelines decorrelate
    (one flipped bf16 tie perturbs every later tensor by as much as the storage noise itself: the PER-BLOCK line of the test
    above), so every unit is checked on its own instead: each convolution of each of the 33 bottlenecks, the stem, the head
    and every stage of the backward chain is recomputed by the bf16-storage oracle FROM THE DEVICE'S OWN STORED INPUTS of
    that unit (forward: the block input, a1, a2 kept for the backward pass; backward: the dC buffer of the block), and
    must match to the few bf16 ties that fp32 summation order flips. A wrong tap, mask, residual, scale or rounding point
    in any one layer is an O(1e-2 .. 1) error here."""
    from oracle import deeplab2_chain as och
    P = _problem_bf16(name)
    taps = {}
    _, grads = _device_iteration(ops, P, torch.bfloat16, taps=taps)
    C = P['geo']['C']
    chain = och.Chain(P['st'], C, LAYERS, 'bf16')
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
    frac = lambda a, b: float((a != b).float().mean())
    nb = len(chain.units)
    fwd, bwd, wg = [], [], {}
    with torch.no_grad():
        s_o, p_o = chain.stem_forward(taps['x_stu'])
        stem_err = (_rel(nchw(taps['blocks'][0]), p_o), frac(nchw(taps['blocks'][0]), p_o))
        for bi in range(nb):
            xin, a1d, a2d = nchw(taps['blocks'][bi]), nchw(taps['a1'][bi]), nchw(taps['a2'][bi])
            a1, a2, out = chain.block_forward(bi, xin, a1_given=a1d, a2_given=a2d)
            outd = nchw(taps['blocks'][bi + 1])
            fwd.append((max(_rel(a1d, a1), _rel(a2d, a2), _rel(outd, out)), max(frac(a1d, a1), frac(a2d, a2), frac(outd, out))))
        x4 = nchw(taps['blocks'][nb])
        head_err = _rel(taps['logits'], chain.head_forward(x4))
        g = {}

        def acc(key, val):
            g[key] = val if key not in g else g[key] + val
        dC = chain.head_backward(x4, taps['dlg'], acc)
        bwd.append((_rel(nchw(taps['dC'][nb - 1]), dC), frac(nchw(taps['dC'][nb - 1]), dC)))
        for bi in range(nb - 1, -1, -1):
            dprev = chain.block_backward(bi, nchw(taps['blocks'][bi]), nchw(taps['a1'][bi]), nchw(taps['a2'][bi]),
                                         nchw(taps['dC'][bi]), acc)
            want = nchw(taps['dC'][bi - 1]) if bi > 0 else nchw(taps['dx'])
            bwd.append((_rel(want, dprev), frac(want, dprev)))
        chain.stem_backward(taps['x_stu'], s_o, nchw(taps['dx']), acc)
    for k, v in g.items():
        wg[k] = _rel(grads[k], v)
    worst = sorted(wg.items(), key=lambda kv: -kv[1])[:4]
    fmax, bmax = max(a for a, _ in fwd), max(a for a, _ in bwd)
    print('\nPARITY teacher-forced bf16 engine vs bf16-storage oracle [{}]: stem {:.1e}/{:.1e} forward max rel {:.2e} '
          '(max differing fraction {:.1e}) head logits {:.1e} backward dC max rel {:.2e} ({:.1e}) weight gradients: max '
          '{:.2e} mean {:.2e} worst {}'.format(name, stem_err[0], stem_err[1], fmax, max(b for _, b in fwd), head_err, bmax,
                                              max(b for _, b in bwd), max(wg.values()), float(np.mean(list(wg.values()))), worst))
    print('PER-BLOCK teacher-forced forward rel: ' + ' '.join('{:.1e}'.format(a) for a, _ in fwd))
    print('PER-BLOCK teacher-forced backward rel (head, then block 32 .. 0): ' + ' '.join('{:.1e}'.format(a) for a, _ in bwd))
    assert len(wg) == 103 + 1 + 4                 # 103 body convolutions + the stem + 2 x (weight, bias) of the live head branches
    # measured on MI355X (profiles/r03b_*): every convolution of the forward pass <= 4.6e-5 (<= 1.3e-4 of the elements
    # differ, by one bf16 ulp), head logits 4e-7, block-level backward <= 7.8e-4, EVERY weight-gradient tensor <= 8.1e-5
    assert stem_err[0] <= 1e-4 and head_err <= 2e-6
    assert fmax <= 1.5e-4 and max(b for _, b in fwd) <= 1e-3, fwd
    assert bmax <= 2e-3, bwd
    assert max(wg.values()) <= 3e-4, worst


def test_aspp_single_pass_formulation_vs_fp64(ops):
    """csrc/aspp.hip: Z = X . Wall^T (1x1 GEMM) + shifted-plane gather == conv_d6(x) + conv_d12(x) + biases; the
    spread / GEMM backward == autograd of the same (architectures/deeplab2.py:124-128)."""
    g = torch.Generator().manual_seed(11)
    N, H, W, C = 2, 15, 17, 21
    ZC = (18 * C + 127) // 128 * 128
    x = torch.randn(N, H, W, 2048, generator=g)
    ws = [torch.randn(C, 2048, 3, 3, generator=g) * 0.01 for _ in range(2)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
    wall = torch.zeros(1, ZC, 2048)
    taps = []
    for i, d in enumerate((6, 12)):
        wall[0, 9 * C * i:9 * C * (i + 1)] = _pack(ws[i]).reshape(9 * C, 2048)
        taps += ops.conv_taps(3, 3, d, d)
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = [w.double().requires_grad_(True) for w in ws]
    ref = sum(F.conv2d(xd, wd[i], bs[i].double(), 1, d, d) for i, d in enumerate((6, 12)))
    dl = torch.randn(ref.shape, generator=g)
    ref.backward(dl.double())
    xg = x.to(DEV)
    z = torch.empty(N, ZC, H, W, device=DEV)
    ops.conv_igemm(xg, wall.to(DEV), [(0, 0)], out_f32_nchw=z, cout_real=ZC)
    logits = ops.aspp_gather_fwd(z, (bs[0] + bs[1]).to(DEV), taps, C)
    assert _rel(logits, ref) <= 5e-6
    d = ops.aspp_spread_bwd(dl.to(DEV), taps, ZC, torch.float32)
    wallT = ops.conv_pack_transpose(wall.to(DEV), flip=False, out_dtype=torch.float32)
    dx = ops.conv_igemm(d, wallT, [(0, 0)], mode=1)
    assert _rel(dx.permute(0, 3, 1, 2), xd.grad) <= 5e-6
    dwall = torch.zeros(1, ZC, 2048, device=DEV)
    ops.conv_wgrad(d, xg, [(0, 0)], dwall)
    for i in range(2):
        got = dwall[0, 9 * C * i:9 * C * (i + 1)].view(9, C, 2048)
        assert _rel(got, _pack(wd[i].grad.float())) <= 5e-6
    assert float(dwall[0, 18 * C:].abs().max()) == 0.0
