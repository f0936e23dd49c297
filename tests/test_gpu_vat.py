"""
GPU: the VAT mean-teacher iteration (vat.py; train_seg_semisup_vat_mt.py:213-301, 346-476)
against oracle/vat.py and -- round 4 -- against outputs of the reference's own VAT closures (tests/golden/vat.npz, made by
tests/golden/make_golden.py::gen_vat: the closures cut out of the trainer with `ast` and run on a reference network).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _state(C, layers):
    from oracle import deeplab2 as odl
    g = torch.Generator().manual_seed(77)
    st = {}
    for k, (shape, dt) in odl.state_spec(C, layers).items():
        if dt == torch.int64:
            st[k] = torch.zeros(shape, dtype=torch.int64)
        elif len(shape) == 4:
            st[k] = torch.randn(shape, generator=g) * (1.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
        elif k.endswith('running_var'):
            st[k] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif k.endswith('running_mean'):
            st[k] = 0.1 * torch.randn(shape, generator=g)
        elif k.endswith('.weight'):
            st[k] = 0.6 + 0.8 * torch.rand(shape, generator=g)
        else:
            st[k] = 0.1 * torch.randn(shape, generator=g)
    return st


def _net(C, layers, st, dtype):
    from architectures import deeplab2
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(st)
    net = net.to(DEV)
    net.compute_dtype = dtype
    net.train()
    net.freeze_batchnorm()
    return net


@pytest.mark.parametrize('loss_fn', ['kld', 'var', 'logits_var', 'bce'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_vat_direction_matches_the_oracle(loss_fn, dtype):
    """Paired layout (x_hat != x: the distance is dominated by the augmentation gap, so the power-iteration step is
    well conditioned). Whatever the compute dtype of the iteration, the direction pass runs in fp32."""
    from oracle import deeplab2 as odl, vat as ov
    from cutmix_semisup_seg_amd import vat
    C, layers = 5, [1, 1, 1, 1]
    st = _state(C, layers)
    net = _net(C, layers, st, dtype)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 33, 41, generator=g)
    x_hat = x + 0.3 * torch.randn(x.shape, generator=g)
    eps0 = ov.normalize_eps(torch.randn(x.shape, generator=g)) * ov.noise_scale(x.shape)
    fnet = lambda t: odl.forward(t, st, layers, frozen=True)
    want, y_want = ov.vat_direction(fnet, x, x_hat, eps0, loss_fn)
    got, y_lo = vat.vat_direction(net, x.to(DEV).to(dtype), x_hat.to(DEV).to(dtype), loss_fn, eps0=eps0.to(DEV))
    assert not net.training and net.compute_dtype == dtype          # eval mode stays (:237), dtype restored
    got = got.cpu()
    xh = x_hat.to(dtype).float() if dtype != torch.float32 else x_hat
    if dtype != torch.float32:                                       # the oracle on the same (bf16-rounded) images
        want, _ = ov.vat_direction(fnet, x.to(dtype).float(), xh, eps0, loss_fn)
    cos = (got.reshape(2, -1) * want.reshape(2, -1)).sum(dim=1)
    assert float(cos.min()) >= 0.995, cos
    torch.testing.assert_close(got.reshape(2, -1).norm(dim=1), torch.ones(2), rtol=1e-4, atol=1e-4)
    # no weight gradient leaked into the network during the direction pass
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in net.parameters())


@pytest.mark.parametrize('loss_fn', ['kld', 'var', 'logits_var', 'bce'])
@pytest.mark.parametrize('adaptive', [False, True], ids=['fixed_radius', 'adaptive_radius'])
def test_vat_perturbation_vs_the_reference_closures(loss_fn, adaptive):
    """The device path (fp32 hand-written engine for the direction pass) against what the REFERENCE's closures computed for
    the same network, images and initial noise (tests/golden/vat.npz)."""
    import json
    import os
    from oracle import deeplab2 as odl
    from cutmix_semisup_seg_amd import vat
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    g = np.load(os.path.join(here, 'vat.npz'))
    meta = json.load(open(os.path.join(here, 'vat_meta.json')))
    C, layers = meta['num_classes'], meta['layers']
    from architectures import deeplab2
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225]))
    net.load_state_dict(odl.closed_form_state(C, layers))
    net = net.to(DEV)
    net.compute_dtype = torch.float32
    net.train()
    net.freeze_batchnorm()
    key = '{}__{}'.format(loss_fn, 'adaptive' if adaptive else 'fixed')
    x, x_hat = torch.from_numpy(g['x']).to(DEV), torch.from_numpy(g['x_hat']).to(DEV)
    eps0 = torch.from_numpy(g[key + '__eps0']).to(DEV)
    pert, _ = vat.vat_perturbation(net, x, x_hat, vat_radius=0.5, adaptive=adaptive, cons_loss_fn=loss_fn, eps0=eps0)
    want = torch.from_numpy(g[key + '__perturbation'])
    got = pert.float().cpu()
    cos = (got.reshape(2, -1) * want.reshape(2, -1)).sum(dim=1) / (got.reshape(2, -1).norm(dim=1) * want.reshape(2, -1).norm(dim=1))
    assert float(cos.min()) >= 0.9995, cos
    torch.testing.assert_close(got.reshape(2, -1).norm(dim=1), want.reshape(2, -1).norm(dim=1), rtol=2e-4, atol=0)     # the radius


def test_vat_iteration_bf16_executor():
    """One VAT iteration of the student on the MFMA executor: perturbation of the requested norm, finite losses, student
    moved, teacher = EMA; and the direction pass through the executor's data-gradient-only backward (forced bf16) leaves
    the gradient arena untouched."""
    from cutmix_semisup_seg_amd import ops, optim as fo, vat
    import optim_weight_ema
    C, layers = 5, [1, 1, 2, 1]
    st = _state(C, layers)
    stu, tea = _net(C, layers, st, torch.bfloat16), _net(C, layers, st, torch.bfloat16)
    opt = fo.FusedAdam(stu, [dict(params=list(stu.pretrained_parameters()), lr=1e-5),
                             dict(params=list(stu.new_parameters()), lr=1e-4)])
    for p in tea.parameters():
        p.requires_grad = False
    ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
    ema.fuse_into(opt)
    g = torch.Generator(device=DEV).manual_seed(2)
    N, H, W = 2, 65, 65
    im = lambda: torch.randn(N, 3, H, W, generator=g, device=DEV).bfloat16()
    y = torch.randint(0, C, (N, 1, H, W), generator=g, device=DEV).to(torch.uint8)
    x_tea, x_stu = im(), im()
    pert, _ = vat.vat_perturbation(tea, x_tea, x_stu, 0.5, False, 'kld', generator=g)
    torch.testing.assert_close(pert.reshape(N, -1).norm(dim=1), torch.full((N,), 0.5 * (3 * H * W) ** 0.5, device=DEV),
                               rtol=1e-3, atol=1e-3)
    pa, _ = vat.vat_perturbation(tea, x_tea, x_stu, 0.5, True, 'kld', generator=g)
    assert pa.shape == x_stu.shape and bool(torch.isfinite(pa).all())
    tea.train(); tea.freeze_batchnorm()

    cfg = vat.VATConfig(vat_radius=0.5, cons_loss_fn='kld', conf_thresh=0.0)
    step = vat.VATMeanTeacherStep(stu, tea, opt, ema, cfg, generator=g)
    w0 = {k: v.clone() for k, v in stu.state_dict().items() if v.dtype == torch.float32}
    t0 = {k: v.clone() for k, v in tea.state_dict().items() if v.dtype == torch.float32}
    r = step(im(), y, [vat.VATUnsupBatch(x_tea, x_stu)])
    assert np.isfinite(float(r['sup_loss'])) and np.isfinite(float(r['consistency_loss'])) and float(r['consistency_loss']) > 0
    sd_s, sd_t = stu.state_dict(), tea.state_dict()
    assert any(not torch.equal(w0[k], sd_s[k]) for k in w0)
    k = 'layer3.1.conv2.weight'
    torch.testing.assert_close(sd_t[k], t0[k] * 0.99 + sd_s[k] * (1.0 - 0.99), rtol=1e-5, atol=1e-7)

    # data-gradient-only backward of the executor (bf16 direction pass, for the mechanism's sake)
    ex = stu.hip_executor()
    opt.zero_grad()
    stu.eval()
    xg = x_stu.float().clone().requires_grad_(True)
    with vat._DataGradOnly(stu):
        lo = stu.forward_lowres(xg.to(torch.bfloat16))
        gx, = torch.autograd.grad(lo, xg, torch.ones_like(lo))
    assert ex is not None and bool(torch.isfinite(gx).all()) and float(gx.abs().max()) > 0
    assert float(opt.arena.grad.abs().max()) == 0.0


def test_vat_trainer_cli_synthetic_end_to_end(tmp_path, monkeypatch):
    import re
    from click.testing import CliRunner
    import train_seg_semisup_vat_mt as trainer
    monkeypatch.chdir(tmp_path)
    args = ['--job_desc', 'vat', '--synthetic', '--arch', 'resnet101_deeplab_imagenet', '--freeze_bn', '--batch_size', '2',
            '--crop_size', '65,65', '--learning_rate', '3e-5', '--vat_radius', '0.5', '--adaptive_vat_radius',
            '--conf_thresh', '0.97', '--num_epochs', '1', '--iters_per_epoch', '2', '--synthetic_val_batches', '1']
    res = CliRunner().invoke(trainer.experiment, args, catch_exceptions=False)
    assert res.exit_code == 0, res.output
    log = open(tmp_path / 'results' / 'train_seg_semisup_vat_mt' / 'log_vat.txt').read()
    lines = [l for l in log.splitlines() if l.startswith('Epoch ')]
    assert len(lines) == 1
    assert re.match(r'Epoch \d+: took [\d.]+s, TRAIN clf loss=[\d.]+, consistency loss=[\d.]+, conf rate=[\d.]+%, '
                    r'VAL mIoU=[\d.]+%', lines[0]), lines
