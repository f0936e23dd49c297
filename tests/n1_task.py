"""
Shared by tests/golden/make_n1_oracle_runs.py (CPU, build container) and tests/test_gpu_miou_training.py (GPU box): the
learnable synthetic segmentation task and the seeded initial weights of the N1 training-parity runs (north star: "mIoU
within 0.2 pt of reference"). Test infrastructure.

Task: 65 x 65 images made of a background and two rectangles, every region filled with its class's mean colour + noise;
5 classes, 2 % ignore labels; a DeepLab v2 over a ResNet-[3, 4, 6, 3] body (the depth of ResNet-50), He-initialised from the
run's seed; 300 CutMix mean-teacher iterations (Adam, teacher alpha 0.95, confidence threshold 0.6), then the reference's
evaluation of the TEACHER (train_seg_semisup_mask_mt.py:484-517) on 8 validation batches.
"""
import numpy as np
import torch

C, LAYERS, N, H, W = 5, [3, 4, 6, 3], 4, 65, 65
ITERS, LR, ALPHA, TAU = 300, 3e-4, 0.95, 0.6
SEEDS = (0, 1, 2)
MEANS = torch.tensor([[1.2, -0.8, 0.1], [-1.0, 1.1, 0.3], [0.2, 0.1, -1.3], [-0.4, -1.2, 1.0], [1.0, 1.0, 1.0]])


def batch(g, n, with_labels=True):
    y = torch.zeros(n, H, W, dtype=torch.int64)
    for i in range(n):
        y[i] = int(torch.randint(0, C, (1,), generator=g))
        for _ in range(2):
            y0, x0 = int(torch.randint(0, H - 16, (1,), generator=g)), int(torch.randint(0, W - 16, (1,), generator=g))
            hh, ww = int(torch.randint(12, 40, (1,), generator=g)), int(torch.randint(12, 40, (1,), generator=g))
            y[i, y0:y0 + hh, x0:x0 + ww] = int(torch.randint(0, C, (1,), generator=g))
    x = MEANS[y].permute(0, 3, 1, 2) + 0.35 * torch.randn(n, 3, H, W, generator=g)
    x = x.bfloat16().float()                     # identical (bf16-representable) inputs for every configuration
    if with_labels:
        y = y.clone()
        y[torch.rand(n, H, W, generator=g) < 0.02] = 255
    return x, y.unsqueeze(1)


def data(seed, iters=ITERS):
    import mask_gen
    g = torch.Generator().manual_seed(2024 + seed)
    rng = np.random.RandomState(7 + seed)
    gen = mask_gen.BoxMaskGenerator(0.5, invert=True)
    train = []
    for _ in range(iters):
        x, y = batch(g, N)
        u0, _ = batch(g, N, False)
        u1, _ = batch(g, N, False)
        train.append((x, y, u0, u1, gen.generate_ranges(N, (H, W), rng=rng)))
    val = [batch(g, N) for _ in range(8)]
    return train, val


def init_state(seed):
    """Seeded He initialisation (bf16-representable conv weights), non-trivial frozen BatchNorm statistics."""
    from oracle import deeplab2 as odl
    g = torch.Generator().manual_seed(900 + seed)
    st = {}
    for k, (shape, dt) in odl.state_spec(C, LAYERS).items():
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


def oracle_run(seed, iters=ITERS, log_every=0):
    from oracle import deeplab2 as odl, step as ostep, boxmask as obox, evaluation as oev
    train, val = data(seed, iters)
    S = ostep.StepState(init_state(seed), C, LAYERS, opt='adam', lr=LR, teacher_alpha=ALPHA)
    ones = torch.ones(N, 1, H, W)
    log = []
    for it, (x, y, u0, u1, ranges) in enumerate(train):
        m = torch.tensor(obox.rasterise(ranges, (H, W), True).astype(np.float32))
        r = ostep.train_iteration(S, x, y, u0, u1, ones, ones, m, conf_thresh=TAU)
        log.append(r['sup_loss'])
        if log_every and it % log_every == 0:
            print('seed {} it {} sup {:.4f} cons {:.3e} rate {:.3f}'.format(seed, it, r['sup_loss'], r['consistency_loss'],
                                                                          r['conf_rate']), flush=True)
    acc = oev.IoUAccumulator(C)
    with torch.no_grad():
        for x, y in val:
            pred = odl.forward(x, S.teacher, LAYERS, frozen=True).argmax(dim=1)
            for i in range(N):
                acc.sample(y[i, 0].numpy(), pred[i].numpy(), ignore_value=255)
    return float(acc.score().mean()), log
