#!/usr/bin/env python
"""
Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE's own Python modules.

Runs only in the build container (needs /root/reference); its outputs (*.npz / *.json, small) are committed
and are all that travels to the GPU box. Nothing in tests/, bench.py or smoke() reads /root/reference.

    cd /root/repo && python tests/golden/make_golden.py

What comes from where:
  * mask_gen.BoxMaskGenerator, optim_weight_ema.EMAWeightOptimizer, evaluation.EvaluatorIoU,
    lr_schedules.make_lr_schedulers, architectures.network_architectures.{robust_binary_crossentropy,
    sigmoid_rampup, seg}, architectures.deeplab2 (ResNetDeepLab + Bottleneck): imported reference code,
    executed as is.
  * The loss section of the training loop (train_seg_semisup_mask_mt.py:346-459) is inline code in a 570-line
    function hard-wired to cuda:0 and cannot be imported; `_trainer_unsup_losses` below drives the SAME torch
    calls the trainer makes (F.softmax, F.kl_div, F.smooth_l1_loss, the reference's
    robust_binary_crossentropy, nn.CrossEntropyLoss(ignore_index=255)) in the trainer's order. The oracle
    (oracle/losses.py) uses explicit formulas instead, so the two are independent statements.
  * torch itself (F.interpolate, torch.optim.Adam/SGD with the duplicated parameter list) is the same library
    the reference calls.
"""
import os
import sys
import json
import hashlib
import importlib.util
import types
import warnings

sys.dont_write_bytecode = True
os.environ['PYTHONDONTWRITEBYTECODE'] = '1'

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'

# keep the repo root (which holds same-named drop-in modules) OFF the path; the reference goes first
sys.path = [p for p in sys.path if os.path.abspath(p or '.') not in (REPO, HERE)]
sys.path.insert(0, REF)

for name in ('torchvision', 'torchvision.models', 'torchvision.models.resnet'):
    sys.modules[name] = types.ModuleType(name)
sys.modules['torchvision'].models = sys.modules['torchvision.models']
sys.modules['torchvision.models'].resnet = sys.modules['torchvision.models.resnet']

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

warnings.filterwarnings('ignore')
torch.set_num_threads(8)

import mask_gen                                     # reference
import optim_weight_ema                             # reference
import evaluation                                   # reference
import lr_schedules                                 # reference
from architectures import network_architectures     # reference
from architectures import deeplab2 as ref_deeplab2  # reference

assert mask_gen.__file__.startswith(REF), mask_gen.__file__


def _load_oracle(name):
    spec = importlib.util.spec_from_file_location('oracle_' + name, os.path.join(REPO, 'oracle', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote {} ({:.1f} KB)'.format(path, os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------------------------------
# 1. box masks
# ----------------------------------------------------------------------------------------------------------
BOX_FLAGSETS = {
    # name: kwargs of BoxMaskGenerator
    'cli_default': dict(prop_range=0.5, n_boxes=1, random_aspect_ratio=True, prop_by_area=True,
                        within_bounds=True, invert=True),
    'cutout_range': dict(prop_range=(0.0, 1.0), n_boxes=1, random_aspect_ratio=True, prop_by_area=True,
                         within_bounds=True, invert=True),
    'three_boxes': dict(prop_range=(0.25, 0.5), n_boxes=3, random_aspect_ratio=True, prop_by_area=True,
                        within_bounds=True, invert=True),
    'fixed_aspect': dict(prop_range=(0.1, 0.6), n_boxes=1, random_aspect_ratio=False, prop_by_area=True,
                         within_bounds=True, invert=True),
    'fixed_aspect_3': dict(prop_range=(0.2, 0.9), n_boxes=3, random_aspect_ratio=False, prop_by_area=True,
                           within_bounds=True, invert=True),
    'by_size': dict(prop_range=(0.2, 0.8), n_boxes=2, random_aspect_ratio=True, prop_by_area=False,
                    within_bounds=True, invert=True),
    'by_size_fixed': dict(prop_range=(0.2, 0.8), n_boxes=2, random_aspect_ratio=False, prop_by_area=False,
                          within_bounds=True, invert=True),
    'outside_bounds': dict(prop_range=(0.3, 0.9), n_boxes=2, random_aspect_ratio=True, prop_by_area=True,
                           within_bounds=False, invert=True),
    'no_invert': dict(prop_range=0.5, n_boxes=1, random_aspect_ratio=True, prop_by_area=True,
                      within_bounds=True, invert=False),
    'zero_prop': dict(prop_range=(0.0, 0.0), n_boxes=1, random_aspect_ratio=True, prop_by_area=True,
                      within_bounds=True, invert=True),
}
BOX_SEEDS = (12345, 1, 2)
BOX_SHAPES = ((32, 32), (321, 321), (512, 1024), (33, 47))
BOX_N = 4


def gen_boxmask():
    out = {}
    meta = {'flagsets': {k: {kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in v.items()}
                         for k, v in BOX_FLAGSETS.items()},
            'seeds': list(BOX_SEEDS), 'shapes': [list(s) for s in BOX_SHAPES], 'n': BOX_N, 'cases': []}
    for fname, kw in BOX_FLAGSETS.items():
        gen = mask_gen.BoxMaskGenerator(**kw)
        for seed in BOX_SEEDS:
            for shape in BOX_SHAPES:
                rng = np.random.RandomState(seed)
                with np.errstate(all='ignore'):
                    m = gen.generate_params(BOX_N, shape, rng=rng)
                assert m.shape == (BOX_N, 1) + shape and m.dtype == np.float64
                assert set(np.unique(m).tolist()) <= {0.0, 1.0}
                key = '{}__s{}__{}x{}'.format(fname, seed, shape[0], shape[1])
                m8 = m.astype(np.uint8)
                out[key + '__rowsum'] = m8.sum(axis=3).astype(np.int32)[:, 0]
                out[key + '__colsum'] = m8.sum(axis=2).astype(np.int32)[:, 0]
                if shape[0] * shape[1] <= 2048:
                    out[key + '__mask'] = np.packbits(m8.reshape(BOX_N, -1), axis=1)
                meta['cases'].append(dict(key=key, flagset=fname, seed=seed, shape=list(shape),
                                          sha256=hashlib.sha256(m8.tobytes()).hexdigest(),
                                          total=int(m8.sum())))
    save('boxmask', **out)
    with open(os.path.join(HERE, 'boxmask_meta.json'), 'w') as f:
        json.dump(meta, f, indent=0)


# ----------------------------------------------------------------------------------------------------------
# 2. EMA
# ----------------------------------------------------------------------------------------------------------
class _ToyNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 3, bias=True)
        self.bn = nn.BatchNorm2d(4)
        self.fc = nn.Linear(4, 2)


def _fill(module, phase):
    with torch.no_grad():
        for i, (k, t) in enumerate(module.state_dict().items()):
            if t.dtype == torch.float32:
                idx = torch.arange(t.numel(), dtype=torch.float64)
                t.copy_(torch.sin(phase + 0.37 * i + 0.9137 * idx).reshape(t.shape).float() * (1.0 + 0.01 * i))
            else:
                t.fill_(int(phase * 10) + 3)


def gen_ema():
    out = {}
    for alpha in (0.99, 0.5, 0.999):
        stu, tea = _ToyNet(), _ToyNet()
        _fill(stu, 0.1)
        _fill(tea, 2.3)
        keys = list(stu.state_dict().keys())
        opt = optim_weight_ema.EMAWeightOptimizer(tea, stu, alpha)
        tag = 'a{}'.format(alpha)
        for k in keys:
            out['{}__init__{}'.format(tag, k)] = tea.state_dict()[k].numpy().copy()
        for step in range(3):
            _fill(stu, 0.1 + 1.7 * (step + 1))
            for k in keys:
                out['{}__src{}__{}'.format(tag, step, k)] = stu.state_dict()[k].numpy().copy()
            opt.step()
            for k in keys:
                out['{}__tgt{}__{}'.format(tag, step, k)] = tea.state_dict()[k].numpy().copy()
    out['keys'] = np.array(keys)
    save('ema', **out)

    # key mismatch must raise ValueError (optim_weight_ema.py:15-18)
    class _Other(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, 4, 3)
    try:
        optim_weight_ema.EMAWeightOptimizer(_Other(), _Other(), 0.9)
        optim_weight_ema.EMAWeightOptimizer(_ToyNet(), _Other(), 0.9)
        raise AssertionError('expected ValueError')
    except ValueError:
        pass


# ----------------------------------------------------------------------------------------------------------
# 3. evaluation
# ----------------------------------------------------------------------------------------------------------
def gen_evaluation():
    out = {}
    for C, shape in ((2, (37, 41)), (19, (64, 128)), (21, (65, 65))):
        rng = np.random.RandomState(100 + C)
        ev = evaluation.EvaluatorIoU(C)
        ev_noign = evaluation.EvaluatorIoU(C)
        for s in range(3):
            truth = rng.randint(0, C, size=shape).astype(np.int64)
            pred = np.where(rng.uniform(size=shape) < 0.6, truth, rng.randint(0, C, size=shape)).astype(np.int64)
            truth_ign = truth.copy()
            truth_ign[rng.uniform(size=shape) < 0.05] = 255
            out['C{}__truth{}'.format(C, s)] = truth_ign.astype(np.uint8)
            out['C{}__pred{}'.format(C, s)] = pred.astype(np.uint8)
            ev.sample(truth_ign, pred, ignore_value=255)
            ev_noign.sample(truth, pred)
            out['C{}__truth_noign{}'.format(C, s)] = truth.astype(np.uint8)
        out['C{}__intersection'.format(C)] = ev.intersection
        out['C{}__union'.format(C)] = ev.union
        out['C{}__cm'.format(C)] = ev.cm
        out['C{}__score'.format(C)] = ev.score()
        out['C{}__noign_intersection'.format(C)] = ev_noign.intersection
        out['C{}__noign_union'.format(C)] = ev_noign.union
        out['C{}__noign_cm'.format(C)] = ev_noign.cm
        out['C{}__noign_score'.format(C)] = ev_noign.score()
    save('evaluation', **out)
    try:
        evaluation.EvaluatorIoU(3, fill_holes=True)
        raise AssertionError('expected ValueError')
    except ValueError:
        pass


# ----------------------------------------------------------------------------------------------------------
# 4. LR schedules + rampup
# ----------------------------------------------------------------------------------------------------------
def gen_lr():
    out = {}
    base = 3e-5

    def make_opt():
        p0, p1 = nn.Parameter(torch.zeros(2)), nn.Parameter(torch.zeros(2))
        return torch.optim.Adam([dict(params=[p0], lr=base * 0.1), dict(params=[p1], lr=base)])

    total = 40
    for sched in ('poly', 'cosine'):
        opt = make_opt()
        ep, it = lr_schedules.make_lr_schedulers(opt, total, sched, '', 0.1, poly_power=0.9)
        assert ep is None and it is not None
        lrs = []
        for i in range(total):
            it.step(i)                                        # train_seg_semisup_mask_mt.py:288-289
            lrs.append([g['lr'] for g in opt.param_groups])
        out[sched] = np.array(lrs, dtype=np.float64)
    opt = make_opt()
    ep, it = lr_schedules.make_lr_schedulers(opt, total, 'stepped', '[3, 6]', 0.1)
    assert it is None and ep is not None
    lrs = []
    for e in range(10):
        ep.step(e)                                            # train_seg_semisup_mask_mt.py:258-259
        lrs.append([g['lr'] for g in opt.param_groups])
    out['stepped'] = np.array(lrs, dtype=np.float64)
    opt = make_opt()
    assert lr_schedules.make_lr_schedulers(opt, total, 'none', '', 0.1) == (None, None)
    # quirk: 'stepped' with an empty milestone string falls through every elif to the ValueError
    # (lr_schedules.py:44, 61-62)
    for bad in (('stepped', ''), ('bogus', '')):
        try:
            lr_schedules.make_lr_schedulers(opt, total, bad[0], bad[1], 0.1)
            raise AssertionError('expected ValueError')
        except ValueError:
            pass
    out['rampup'] = np.array([[e, R, network_architectures.sigmoid_rampup(e, R)]
                              for R in (0, 5, 40) for e in (0, 1, 3, 5, 7, 40, 100)], dtype=np.float64)
    out['base_lr'] = np.array(base)
    save('lr', **out)


# ----------------------------------------------------------------------------------------------------------
# 5. losses (driving the trainer's own torch calls, train_seg_semisup_mask_mt.py:346-459)
# ----------------------------------------------------------------------------------------------------------
def _trainer_unsup_losses(mode, l_stu, l_tea_pair, masks, ums, cons_loss_fn, conf_thresh, conf_per_pixel,
                          ramp_val, rampup, cons_weight, n_classes):
    root_n_classes = np.sqrt(n_classes) if False else __import__('math').sqrt(n_classes)   # :75
    if mode == 'mix':
        logits_cons_tea = l_tea_pair[0] * (1 - masks) + l_tea_pair[1] * masks
        loss_mask = ums[0] * (1 - masks) + ums[1] * masks
    else:
        logits_cons_tea = l_tea_pair[0]
        loss_mask = masks * ums[0]
    logits_cons_stu = l_stu
    prob_cons_tea = F.softmax(logits_cons_tea, dim=1)
    prob_cons_stu = F.softmax(logits_cons_stu, dim=1)
    conf_rate = float('nan')
    if conf_thresh > 0.0:
        conf_tea = prob_cons_tea.max(dim=1)[0]
        conf_mask = (conf_tea >= conf_thresh).float()[:, None, :, :]
        conf_rate = float(conf_mask.mean())
        if not conf_per_pixel:
            conf_mask = conf_mask.mean()
        loss_mask = loss_mask * conf_mask
    if cons_loss_fn == 'var':
        d = prob_cons_stu - prob_cons_tea
        closs = (d * d).sum(dim=1, keepdim=True)
    elif cons_loss_fn == 'logits_var':
        d = logits_cons_stu - logits_cons_tea
        closs = (d * d).sum(dim=1, keepdim=True) / root_n_classes
    elif cons_loss_fn == 'logits_smoothl1':
        closs = F.smooth_l1_loss(logits_cons_stu, logits_cons_tea, reduction='none')
        closs = closs.sum(dim=1, keepdim=True) / root_n_classes
    elif cons_loss_fn == 'bce':
        closs = network_architectures.robust_binary_crossentropy(prob_cons_stu, prob_cons_tea)
        closs = closs.sum(dim=1, keepdim=True)
    elif cons_loss_fn == 'kld':
        closs = F.kl_div(F.log_softmax(logits_cons_stu, dim=1), prob_cons_tea, reduction='none')
        closs = closs.sum(dim=1, keepdim=True)
    closs = (closs * loss_mask).mean()
    if rampup > 0:
        closs = closs * ramp_val
    unsup = closs * cons_weight
    return closs, unsup, conf_rate


def gen_losses():
    out = {}
    cases = []
    gen = mask_gen.BoxMaskGenerator(0.5, invert=True)
    for C, hw in ((21, (9, 11)), (2, (9, 11))):
        g = torch.Generator().manual_seed(777 + C)
        N = 2
        H, W = hw
        # logits scaled so that teacher confidences straddle the thresholds
        l_stu0 = torch.randn(N, C, H, W, generator=g) * 2.0
        l0 = torch.randn(N, C, H, W, generator=g) * 3.0
        l1 = torch.randn(N, C, H, W, generator=g) * 3.0
        um0 = (torch.rand(N, 1, H, W, generator=g) > 0.2).float()
        um1 = (torch.rand(N, 1, H, W, generator=g) > 0.2).float()
        m = torch.tensor(gen.generate_params(N, (H, W), rng=np.random.RandomState(5 + C)).astype(np.float32))
        pre = 'C{}__'.format(C)
        out[pre + 'l_stu'] = l_stu0.numpy()
        out[pre + 'l0_tea'] = l0.numpy()
        out[pre + 'l1_tea'] = l1.numpy()
        out[pre + 'um0'] = um0.numpy()
        out[pre + 'um1'] = um1.numpy()
        out[pre + 'mask'] = m.numpy()
        for fn in ('var', 'logits_var', 'logits_smoothl1', 'bce', 'kld'):
            for mode in ('mix', 'cut'):
                for (tau, pp) in ((0.0, False), (0.5, False), (0.5, True), (0.97, False), (0.97, True)):
                    for (rampup, ramp_val, w) in ((-1, 1.0, 1.0), (5, network_architectures.sigmoid_rampup(2, 5), 0.3)):
                        if rampup > 0 and not (tau == 0.5 and not pp):
                            continue
                        l_stu = l_stu0.clone().requires_grad_(True)
                        closs, unsup, rate = _trainer_unsup_losses(
                            mode, l_stu, (l0, l1), m, (um0, um1), fn, tau, pp, ramp_val, rampup, w, C)
                        unsup.backward()
                        key = '{}{}__{}__t{}__pp{}__r{}'.format(pre, fn, mode, tau, int(pp), rampup)
                        out[key + '__grad'] = l_stu.grad.numpy().copy()
                        out[key + '__vals'] = np.array([float(closs), float(unsup), rate], dtype=np.float64)
                        cases.append(dict(key=key, C=C, fn=fn, mode=mode, conf_thresh=tau, conf_per_pixel=pp,
                                          rampup=rampup, ramp_val=ramp_val, cons_weight=w))
        # supervised CE
        labels = torch.randint(0, C, (N, H, W), generator=g)
        labels[torch.rand(N, H, W, generator=g) < 0.1] = 255
        l_sup = l_stu0.clone().requires_grad_(True)
        ce = nn.CrossEntropyLoss(ignore_index=255)(l_sup, labels)
        ce.backward()
        out[pre + 'labels'] = labels.numpy().astype(np.uint8)
        out[pre + 'ce__val'] = np.array(float(ce), dtype=np.float64)
        out[pre + 'ce__grad'] = l_sup.grad.numpy().copy()
    # bilinear upsample, both corner conventions
    g = torch.Generator().manual_seed(4242)
    lo = torch.randn(2, 5, 5, 7, generator=g)
    out['up__lo'] = lo.numpy()
    for ac in (True, False):
        lo_g = lo.clone().requires_grad_(True)
        hi = F.interpolate(lo_g, size=(33, 41), mode='bilinear', align_corners=ac)
        wgt = torch.randn(hi.shape, generator=g)
        (hi * wgt).sum().backward()
        out['up__hi_ac{}'.format(int(ac))] = hi.detach().numpy()
        out['up__wgt_ac{}'.format(int(ac))] = wgt.numpy()
        out['up__grad_ac{}'.format(int(ac))] = lo_g.grad.numpy().copy()
    save('losses', **out)
    with open(os.path.join(HERE, 'losses_meta.json'), 'w') as f:
        json.dump(cases, f, indent=0)


# ----------------------------------------------------------------------------------------------------------
# 6. DeepLab v2: reference nn.Module on closed-form weights
# ----------------------------------------------------------------------------------------------------------
def _closed_form_input(n, h, w, phase):
    idx = torch.arange(n * 3 * h * w, dtype=torch.float64)
    return torch.sin(phase + 0.61803398875 * idx).reshape(n, 3, h, w).float() * 1.5


def gen_deeplab2():
    oracle_dl = _load_oracle('deeplab2')
    out = {}
    meta = {}
    mean = np.array([0.485, 0.456, 0.406])
    std = np.array([0.229, 0.224, 0.225])
    for tag, layers, C, inputs in (
            ('tiny', [1, 1, 1, 1], 5, [(2, 33, 33), (1, 40, 57)]),
            ('r101', [3, 4, 23, 3], 21, [(2, 33, 33), (1, 65, 97)])):
        if tag == 'r101':
            net = ref_deeplab2.resnet101_deeplab_imagenet(C, pretrained=False)
        else:
            net = ref_deeplab2.ResNetDeepLab(ref_deeplab2.Bottleneck, layers, C, mean, std)
        st = oracle_dl.closed_form_state(C, layers)
        assert list(st.keys()) == list(net.state_dict().keys()), 'state spec mismatch'
        net.load_state_dict(st)
        meta[tag] = dict(layers=layers, num_classes=C, n_state=len(st),
                         n_float=sum(1 for v in st.values() if v.dtype == torch.float32),
                         float_elems=int(sum(v.numel() for v in st.values() if v.dtype == torch.float32)),
                         n_params=int(sum(p.numel() for p in net.parameters())),
                         n_trainable=int(sum(p.numel() for p in net.parameters() if p.requires_grad)))
        # parameter-group multiplicities (deeplab2.py:208-242)
        id2key = {id(p): k for k, p in net.named_parameters()}
        meta[tag]['pretrained_order'] = [id2key[id(p)] for p in net.pretrained_parameters()]
        meta[tag]['new_order'] = [id2key[id(p)] for p in net.new_parameters()]

        net.train()
        net.freeze_batchnorm()                      # the --freeze_bn configuration
        for ii, (n, h, w) in enumerate(inputs):
            x = _closed_form_input(n, h, w, 0.3 + ii)
            feats = {}
            hook = net.layer5.register_forward_hook(lambda m, i, o: feats.__setitem__('lo', o))
            hook4 = net.layer4.register_forward_hook(lambda m, i, o: feats.__setitem__('l4', o))
            net.zero_grad()
            y = net(x)
            hook.remove()
            hook4.remove()
            wsum = torch.cos(0.11 * torch.arange(y.numel(), dtype=torch.float64)).reshape(y.shape).float()
            (y * wsum).sum().backward()
            key = '{}__in{}'.format(tag, ii)
            out[key + '__lowres'] = feats['lo'].detach().numpy()
            out[key + '__full_sub4'] = y.detach().numpy()[:, :, ::4, ::4].copy()
            out[key + '__l4_stats'] = np.array([float(feats['l4'].double().sum()),
                                                float((feats['l4'].double() ** 2).sum())])
            out[key + '__full_stats'] = np.array([float(y.double().sum()), float((y.double() ** 2).sum())])
            gk = ['conv1.weight', 'layer1.0.conv2.weight', 'layer3.0.downsample.0.weight',
                  'layer4.0.conv3.weight', 'layer5.conv2d_list.0.weight', 'layer5.conv2d_list.1.bias']
            named = dict(net.named_parameters())
            for k in gk:
                gr = named[k].grad
                out['{}__grad__{}'.format(key, k)] = gr.numpy().reshape(-1)[:4096].copy()
                out['{}__gradstats__{}'.format(key, k)] = np.array([float(gr.double().sum()),
                                                                    float((gr.double() ** 2).sum())])
            meta[tag]['none_grads_in{}'.format(ii)] = sorted(k for k, p in named.items()
                                                             if p.requires_grad and p.grad is None)
        # one un-frozen (batch-statistics) forward on the tiny net only
        if tag == 'tiny':
            net.load_state_dict(st)
            net.train()
            x = _closed_form_input(2, 33, 33, 0.3)
            y = net(x)
            out['tiny__bnstat__full_sub4'] = y.detach().numpy()[:, :, ::4, ::4].copy()
            sd = net.state_dict()
            for k in ('bn1.running_mean', 'bn1.running_var', 'layer4.0.bn3.running_var',
                      'layer2.0.downsample.1.running_mean'):
                out['tiny__bnstat__' + k] = sd[k].numpy().copy()
    save('deeplab2', **out)
    with open(os.path.join(HERE, 'deeplab2_meta.json'), 'w') as f:
        json.dump(meta, f, indent=0)
    # registry names (network_architectures.py:41-112)
    with open(os.path.join(HERE, 'registry_names.json'), 'w') as f:
        json.dump(sorted(network_architectures.seg.names()), f)


# ----------------------------------------------------------------------------------------------------------
# 7. Adam / SGD with the duplicated parameter list
# ----------------------------------------------------------------------------------------------------------
def gen_optim():
    out = {}
    g = torch.Generator().manual_seed(99)
    for opt_name in ('adam', 'sgd', 'sgd_nesterov'):
        for k in (1, 3, 4):
            p = nn.Parameter(torch.randn(257, generator=g))
            p0 = p.detach().clone()
            if opt_name == 'adam':
                opt = torch.optim.Adam([dict(params=[p] * k, lr=3e-3)])
            else:
                opt = torch.optim.SGD([dict(params=[p] * k, lr=3e-3)], momentum=0.9,
                                      nesterov=(opt_name == 'sgd_nesterov'), weight_decay=5e-4)
            grads, ps = [], []
            for step in range(3):
                gr = torch.randn(257, generator=g) * (0.1 if step != 1 else 3.0)
                p.grad = gr.clone()
                opt.step()
                grads.append(gr.numpy().copy())
                ps.append(p.detach().numpy().copy())
            key = '{}__k{}'.format(opt_name, k)
            out[key + '__p0'] = p0.numpy()
            out[key + '__grads'] = np.stack(grads)
            out[key + '__ps'] = np.stack(ps)
            if opt_name == 'adam':
                stt = opt.state[p]
                out[key + '__step'] = np.array(float(stt['step']))
                out[key + '__m'] = stt['exp_avg'].numpy().copy()
                out[key + '__v'] = stt['exp_avg_sq'].numpy().copy()
            else:
                out[key + '__buf'] = opt.state[p]['momentum_buffer'].numpy().copy()
    save('optim', **out)


# ----------------------------------------------------------------------------------------------------------
# 8. three iterations of the whole step on the tiny network (train_seg_semisup_mask_mt.py:287-467)
# ----------------------------------------------------------------------------------------------------------
def gen_step():
    oracle_dl = _load_oracle('deeplab2')
    C, layers = 5, [1, 1, 1, 1]
    N, H, W = 2, 33, 33
    mean = np.array([0.485, 0.456, 0.406])
    std = np.array([0.229, 0.224, 0.225])
    out = {}
    for cfg_name, cfg in (('adam_var_mix', dict(opt='adam', fn='var', mode='mix', tau=0.3, pp=False)),
                          ('sgd_kld_cut_pp', dict(opt='sgd', fn='kld', mode='cut', tau=0.3, pp=True))):
        stu = ref_deeplab2.ResNetDeepLab(ref_deeplab2.Bottleneck, layers, C, mean, std)
        tea = ref_deeplab2.ResNetDeepLab(ref_deeplab2.Bottleneck, layers, C, mean, std)
        stu.load_state_dict(oracle_dl.closed_form_state(C, layers))
        lr = 1e-3
        groups = [dict(params=stu.pretrained_parameters(), lr=lr * 0.1), dict(params=stu.new_parameters(), lr=lr)]
        if cfg['opt'] == 'adam':
            opt = torch.optim.Adam(groups)
        else:
            opt = torch.optim.SGD(groups, momentum=0.9, nesterov=False, weight_decay=5e-4)
        for p in tea.parameters():
            p.requires_grad = False
        ema = optim_weight_ema.EMAWeightOptimizer(tea, stu, 0.99)
        gen = mask_gen.BoxMaskGenerator(0.5, invert=True)
        rng = np.random.RandomState(12345)
        ce = nn.CrossEntropyLoss(ignore_index=255)
        stu.train(); tea.train(); stu.freeze_batchnorm(); tea.freeze_batchnorm()
        logs = []
        for it in range(3):
            g = torch.Generator().manual_seed(1000 + it)
            x = torch.randn(N, 3, H, W, generator=g)
            y = torch.randint(0, C, (N, 1, H, W), generator=g)
            y[torch.rand(N, 1, H, W, generator=g) < 0.05] = 255
            ux0 = torch.randn(N, 3, H, W, generator=g)
            ux1 = torch.randn(N, 3, H, W, generator=g)
            um0 = torch.ones(N, 1, H, W)
            um1 = torch.ones(N, 1, H, W)
            m = torch.tensor(gen.generate_params(N, (H, W), rng=rng).astype(np.float32))
            opt.zero_grad()
            sup_loss = ce(stu(x), y[:, 0])
            sup_loss.backward()
            if cfg['mode'] == 'mix':
                x_mixed = ux0 * (1 - m) + ux1 * m
                with torch.no_grad():
                    lt = (tea(ux0).detach(), tea(ux1).detach())
                ls = stu(x_mixed)
            else:
                with torch.no_grad():
                    lt = (tea(ux0).detach(), None)
                ls = stu(ux0 * m)
            closs, unsup, rate = _trainer_unsup_losses(cfg['mode'], ls, lt, m, (um0, um1), cfg['fn'], cfg['tau'],
                                                       cfg['pp'], 1.0, -1, 1.0, C)
            unsup.backward()
            opt.step()
            ema.step()
            ssd, tsd = stu.state_dict(), tea.state_dict()
            logs.append([float(sup_loss), float(closs), rate,
                         float(sum(v.double().sum() for v in ssd.values() if v.dtype == torch.float32)),
                         float(sum((v.double() ** 2).sum() for v in ssd.values() if v.dtype == torch.float32)),
                         float(sum(v.double().sum() for v in tsd.values() if v.dtype == torch.float32)),
                         float(sum((v.double() ** 2).sum() for v in tsd.values() if v.dtype == torch.float32))])
        out[cfg_name + '__log'] = np.array(logs, dtype=np.float64)
        for k in ('conv1.weight', 'layer3.0.conv2.weight', 'layer5.conv2d_list.1.weight',
                  'layer5.conv2d_list.3.weight'):
            out['{}__stu__{}'.format(cfg_name, k)] = stu.state_dict()[k].numpy().reshape(-1)[:2048].copy()
            out['{}__tea__{}'.format(cfg_name, k)] = tea.state_dict()[k].numpy().reshape(-1)[:2048].copy()
        if cfg['opt'] == 'adam':
            named = dict(stu.named_parameters())
            out[cfg_name + '__adam_steps'] = np.array(
                [float(opt.state[named[k]]['step']) for k in ('conv1.weight', 'layer1.0.conv1.weight',
                                                              'layer1.0.downsample.0.weight',
                                                              'layer5.conv2d_list.0.weight')])
    save('step', **out)


# ----------------------------------------------------------------------------------------------------------
# 9. command-line surface of the trainer (train_seg_semisup_mask_mt.py:581-638)
# ----------------------------------------------------------------------------------------------------------
def gen_cli():
    import click
    import train_seg_semisup_mask_mt as ref_trainer      # reference (module body only defines the click command)
    assert ref_trainer.__file__.startswith(REF)
    opts = []
    for prm in ref_trainer.experiment.params:
        kind = type(prm.type).__name__
        choices = list(prm.type.choices) if isinstance(prm.type, click.Choice) else None
        opts.append(dict(name=prm.name, opts=list(prm.opts), is_flag=bool(getattr(prm, 'is_flag', False)),
                         default=prm.default if not callable(prm.default) else None, type=kind, choices=choices))
    with open(os.path.join(HERE, 'cli_options.json'), 'w') as f:
        json.dump(opts, f, indent=0, default=str)
    print('wrote cli_options.json ({} options)'.format(len(opts)))


def gen_cli_vat():
    """command-line surface of the VAT trainer (train_seg_semisup_vat_mt.py:592-644)"""
    import click
    import train_seg_semisup_vat_mt as ref_trainer
    assert ref_trainer.__file__.startswith(REF)
    opts = []
    for prm in ref_trainer.experiment.params:
        kind = type(prm.type).__name__
        choices = list(prm.type.choices) if isinstance(prm.type, click.Choice) else None
        opts.append(dict(name=prm.name, opts=list(prm.opts), is_flag=bool(getattr(prm, 'is_flag', False)),
                         default=prm.default if not callable(prm.default) else None, type=kind, choices=choices))
    with open(os.path.join(HERE, 'cli_options_vat.json'), 'w') as f:
        json.dump(opts, f, indent=0, default=str)
    print('wrote cli_options_vat.json ({} options)'.format(len(opts)))


def gen_checkpoint():
    """A whole-module pickle written by the REFERENCE's classes the way its trainer does (`torch.save(eval_net, path)`,
    train_seg_semisup_mask_mt.py:533-535), plus the outcome of the reference's `_load_state_into_model`
    (architectures/deeplab2.py:310-322) on a state dict with a missing key, a wrong-shaped entry and a foreign key.
    To keep the fixture small every tensor of the pickled module is a stride-0 expansion of ONE element (value = a
    closed form of its key), so the file holds the object graph, class paths, shapes and dtypes -- not 36 MB of weights."""
    import io
    odl = _load_oracle('deeplab2')
    C, layers = 5, [1, 1, 1, 1]
    net = ref_deeplab2.ResNetDeepLab(ref_deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.eval()
    tag = {}
    with torch.no_grad():
        for m in net.modules():
            for name, p in list(m._parameters.items()):
                if p is not None:
                    m._parameters[name] = nn.Parameter(torch.zeros(1).expand(p.shape), requires_grad=p.requires_grad)
            for name, b in list(m._buffers.items()):
                if b is not None and b.dtype == torch.float32:
                    m._buffers[name] = torch.zeros(1).expand(b.shape)
        for k, v in net.state_dict().items():
            if v.dtype == torch.float32:
                val = ((odl._key_seed(k) % 1000) - 500) / 1000.0
                v.untyped_storage().copy_(torch.tensor([val]).untyped_storage())
                tag[k] = val
    path = os.path.join(HERE, 'ref_module_checkpoint.pth')
    torch.save(net, path)
    print('wrote {} ({:.1f} KB)'.format(path, os.path.getsize(path) / 1024))
    # _load_state_into_model semantics
    net2 = ref_deeplab2.ResNetDeepLab(ref_deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    torch.manual_seed(0)
    before = {k: v.clone() for k, v in net2.state_dict().items()}
    sd = odl.closed_form_state(C, layers)
    sd = {k: v for k, v in sd.items()}
    del sd['layer2.0.conv2.weight']                                   # missing -> keeps its initialisation
    sd['layer5.conv2d_list.0.weight'] = torch.ones(C + 1, 2048, 3, 3)  # wrong shape -> keeps its initialisation
    sd['not.a.key'] = torch.ones(3)                                   # foreign key -> ignored
    buf = io.StringIO()
    old = sys.stdout
    sys.stdout = buf
    try:
        ref_deeplab2._load_state_into_model(net2, sd, verbose=True)
    finally:
        sys.stdout = old
    after = net2.state_dict()
    kept = sorted(k for k in after if torch.equal(after[k], before[k]) and k in ('layer2.0.conv2.weight', 'layer5.conv2d_list.0.weight'))
    loaded = sorted(k for k in after if k in sd and after[k].shape == sd[k].shape and torch.equal(after[k], sd[k]))
    with open(os.path.join(HERE, 'checkpoint_meta.json'), 'w') as f:
        json.dump(dict(tag=tag, num_classes=C, layers=layers, kept_init=kept, n_loaded=len(loaded), n_keys=len(after),
                       verbose_lines=buf.getvalue().strip().splitlines()), f, indent=0)
    print('wrote checkpoint_meta.json (loaded {} of {} keys, kept {})'.format(len(loaded), len(after), kept))


# ----------------------------------------------------------------------------------------------------------
# 12. rotate / scale crop: the affine matrices of SegCVTransformRandomCropRotateScale.transform_single
#     (datapipe/seg_transforms_cv.py:331-358). The transform class itself imports cv2 (absent) at module level; the
#     matrix arithmetic lives in datapipe/affine.py, which imports cleanly: this section runs THAT code on seeded
#     parameter draws made in the transform's order, so the fixture pins draws -> centre -> local_xf (float32).
# ----------------------------------------------------------------------------------------------------------
def gen_affine():
    import math
    from datapipe import affine                         # reference
    assert affine.__file__.startswith(REF)
    cases = []
    for seed, crop, img, rot_mag, max_scale, uniform in (
            (1, (321, 321), (375, 500), 30.0, 1.5, True), (2, (256, 512), (1024, 2048), 10.0, 2.0, False),
            (3, (224, 224), (200, 180), 45.0, 1.25, True), (4, (65, 97), (300, 300), 0.0, 1.5, True),
            (5, (129, 129), (140, 400), 90.0, 1.0, False)):
        rng = np.random.RandomState(seed)
        crop_arr = np.array(crop)
        log_max_scale, rot_rad = np.log(max_scale), math.radians(rot_mag)
        for _ in range(3):
            # (the statements of transform_single, :331-358, with the class attributes spelled out)
            if uniform:
                sf = np.exp(rng.uniform(-log_max_scale, log_max_scale, size=(1,)))
                sf = np.repeat(sf, 2, axis=0)
            else:
                sf = np.exp(rng.uniform(-log_max_scale, log_max_scale, size=(2,)))
            theta = rng.uniform(-rot_rad, rot_rad, size=(1,))
            sc_size = crop_arr / sf
            img_size = np.array(img)
            extra = np.maximum(img_size - sc_size, 0.0)
            centre = extra * rng.uniform(0.0, 1.0, size=(2,)) + np.minimum(sc_size, img_size) * 0.5
            local_xf = affine.cat_nx2x3(
                affine.translation_matrices(crop_arr[None, ::-1] * 0.5),
                affine.rotation_matrices(theta),
                affine.scale_matrices(sf[None, ::-1]),
                affine.translation_matrices(-centre[None, ::-1]),
            )
            interp = int(rng.choice([0, 1]))            # rng.choice([cv2.INTER_NEAREST, cv2.INTER_LINEAR]) (:354), values 0 / 1
            cases.append(dict(seed=seed, crop=list(crop), img=list(img), rot_mag=rot_mag, max_scale=max_scale,
                              uniform=uniform, sf=sf.tolist(), theta=float(theta[0]), centre=centre.tolist(),
                              xf=local_xf[0].astype(np.float64).tolist(), xf_dtype=str(local_xf.dtype), interp=interp))
    with open(os.path.join(HERE, 'affine_rotate_scale.json'), 'w') as f:
        json.dump(cases, f)
    print('wrote affine_rotate_scale.json ({} cases)'.format(len(cases)))


# ----------------------------------------------------------------------------------------------------------
# 13. virtual adversarial direction / perturbation (train_seg_semisup_vat_mt.py:213-301)
# ----------------------------------------------------------------------------------------------------------
def gen_vat():
    """The VAT pieces are closures inside the reference's 570-line trainer function and cannot be imported. Here their
    FunctionDef nodes are cut out of the reference's source with `ast` AT GENERATION TIME, compiled, and executed with the
    trainer's free variables (vat_dir_net, cons_loss_fn, adaptive_vat_radius, vat_radius) bound to a tiny reference DeepLab v2
    with closed-form weights: the arrays written below are outputs of the reference's own code. Nothing of its text is
    stored; the initial noise (torch.randn in `normalized_noise_like`) is captured so that the oracle and the device start
    from the same draw."""
    import ast
    import math
    oracle_dl = _load_oracle('deeplab2')
    src = open(os.path.join(REF, 'train_seg_semisup_vat_mt.py')).read()
    tree = ast.parse(src)
    wanted = ['t_dot', 'normalize_eps', 'normalized_noise_like', 'vat_direction', 'vat_perburbation']
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in wanted and node.name not in found:
            found[node.name] = node
    assert sorted(found) == sorted(wanted), sorted(found)
    mod = ast.Module(body=[found[n] for n in wanted], type_ignores=[])
    ast.fix_missing_locations(mod)
    mean = np.array([0.485, 0.456, 0.406])
    std = np.array([0.229, 0.224, 0.225])
    C = 5
    net = ref_deeplab2.ResNetDeepLab(ref_deeplab2.Bottleneck, [1, 1, 1, 1], C, mean, std)
    net.load_state_dict(oracle_dl.closed_form_state(C, [1, 1, 1, 1]))
    net.train()
    net.freeze_batchnorm()
    out = {}
    meta = dict(layers=[1, 1, 1, 1], num_classes=C, cases=[])
    x = _closed_form_input(2, 33, 33, 0.7)
    x_hat = x + 0.05 * _closed_form_input(2, 33, 33, 1.9)          # the colour-augmented view of the same crop
    out['x'], out['x_hat'] = x.numpy(), x_hat.numpy()
    for loss_fn in ('var', 'bce', 'kld', 'logits_var'):
        for adaptive in (False, True):
            ns = dict(torch=torch, F=F, math=math, network_architectures=network_architectures, vat_dir_net=net,
                      cons_loss_fn=loss_fn, adaptive_vat_radius=adaptive, vat_radius=0.5)
            exec(compile(mod, '<reference VAT closures>', 'exec'), ns)
            ref_noise = ns['normalized_noise_like']
            drawn = {}

            def capture(xx, requires_grad=False, scale=1.0, _f=ref_noise, _d=drawn):
                e = _f(xx, requires_grad=requires_grad, scale=scale)
                _d['eps0'] = e.detach().clone()
                return e
            ns['normalized_noise_like'] = capture
            torch.manual_seed(1234)
            pert, y_logits, y_prob = ns['vat_perburbation'](x, x_hat, None)
            torch.manual_seed(1234)
            direction, _, _ = ns['vat_direction'](x, x_hat)
            key = '{}__{}'.format(loss_fn, 'adaptive' if adaptive else 'fixed')
            out[key + '__eps0'] = drawn['eps0'].numpy()
            out[key + '__direction'] = direction.detach().numpy()
            out[key + '__perturbation'] = pert.detach().numpy()
            out[key + '__y_logits_sub4'] = y_logits.detach().numpy()[:, :, ::4, ::4].copy()
            meta['cases'].append(dict(key=key, loss_fn=loss_fn, adaptive=adaptive, vat_radius=0.5,
                                      pert_norm=[float(v) for v in pert.detach().reshape(2, -1).norm(dim=1)]))
            assert not net.training, 'vat_direction leaves the direction network in eval mode (:237)'
            net.train()
            net.freeze_batchnorm()
    save('vat', **out)
    with open(os.path.join(HERE, 'vat_meta.json'), 'w') as f:
        json.dump(meta, f, indent=0)


# ----------------------------------------------------------------------------------------------------------
# 14. mask_gen.gaussian_kernels (mask_gen.py:26-43; a helper of the reference's mask generators, unused by the trainer)
# ----------------------------------------------------------------------------------------------------------
def gen_gauss():
    sig = np.array([0.5, 1.0, 2.5, 4.0])
    save('gaussian_kernels', sigma=sig, auto=mask_gen.gaussian_kernels(sig),
         wide=mask_gen.gaussian_kernels(sig, max_sigma=6.0, truncate=3.0))


if __name__ == '__main__':
    which = sys.argv[1:] or ['boxmask', 'ema', 'evaluation', 'lr', 'losses', 'deeplab2', 'optim', 'step', 'cli', 'cli_vat',
                             'checkpoint', 'affine', 'vat', 'gauss']
    for w in which:
        globals()['gen_' + w]()
