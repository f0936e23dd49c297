"""
Virtual adversarial training (VAT) mean-teacher iteration -- the body of the reference's second trainer,
train_seg_semisup_vat_mt.py:213-301 (direction / perturbation) and :346-476 (iteration); SURVEY.md 8(f) rank 2.

    student(x_sup) -> CE -> backward                                                    :356-358
    x_perturb = vat_perburbation(x_tea, x_stu)                                          :398   (see below)
    teacher(x_tea) (no grad), student(x_stu + x_perturb)                                :404-407
    consistency between the two, masked by the validity mask x confidence               :413-459
    optimizer + EMA                                                                     :476-478

What runs where:
  * the two network passes of the ITERATION, the cross entropy, the consistency loss (the fused kernel in "cut" mode with
    an all-ones box mask: loss_mask = um * confidence, exactly :415-429) and Adam/SGD + EMA are the same MI355X kernels
    as in the CutMix step (step.py);
  * the VAT DIRECTION is one step of power iteration: the gradient, wrt a perturbation of norm 1e-6*H*W/1000, of the
    distance between net(x_hat + eps) and net(x) (:244-268). That perturbation is ~2e-7 per pixel -- below bf16
    resolution of the input (and barely above fp32's); the pass therefore runs the network in fp32, whatever
    `compute_dtype` the iteration uses: for the DeepLab networks on the fp32 configuration of the hand-written engine
    (f32-input MFMA convolutions csrc/conv_f32.hip, stem incl. its image gradient csrc/stem.hip -- no library
    convolution in the pass), for the U-Nets on the library's fp32 convolutions with csrc/bn.hip BatchNorm. The
    distance and its gradient wrt the low-resolution logits come
    from the fused consistency kernels (the reference SUMS where the kernel averages; the direction is normalised, so a
    positive factor is immaterial), the gradient wrt the image from the network's backward pass with the weight
    gradients switched off (`torch.autograd.grad` wrt eps only, like the reference);
  * quirk kept: the network used for the direction is put in eval mode and stays there (:237).
"""
import math
import os

import torch

from . import ops


def normalize_eps(x):
    """x / (per-sample L2 norm + 1e-12), train_seg_semisup_vat_mt.py:216-219."""
    flat = x.reshape(len(x), -1)
    mag = torch.sqrt((flat * flat).sum(dim=1))
    return x / (mag[:, None, None, None] + 1e-12)


def normalized_noise_like(x, scale=1.0, generator=None):
    """:221-226 (without the requires_grad switch: the direction pass sets it)."""
    eps = torch.randn(x.shape, dtype=torch.float32, device=x.device, generator=generator)
    return normalize_eps(eps) * scale


def _ones_ranges(n, device):
    # an empty box with invert=False rasterises to an all-ones mask (mask_gen.py:110-116)
    return torch.zeros((n, 1, 4), dtype=torch.int32, device=device)


class _DataGradOnly(object):
    """Backward passes inside this context produce input gradients only (no weight gradients into the arenas)."""

    def __init__(self, net):
        self.net = net

    def __enter__(self):
        # read by whichever executor of the network runs the backward pass (backbone_hip.py)
        self.prev = getattr(self.net, '_data_grad_only', False)
        self.net._data_grad_only = True

    def __exit__(self, *exc):
        self.net._data_grad_only = self.prev


def vat_direction(net, x, x_hat, cons_loss_fn='kld', eps0=None, generator=None):
    """
    -> (normalised adversarial direction fp32 (N,C,H,W), low-resolution logits of net(x) in eval mode).
    `eps0` (optional): the initial noise, already normalised and scaled (tests pass the oracle's draw).
    """
    if cons_loss_fn not in ('var', 'bce', 'kld', 'logits_var'):
        raise ValueError('Unknown consistency loss function {}'.format(cons_loss_fn))
    net.eval()                                                  # and it stays there (:237)
    out_size = x.shape[2:4]
    align = getattr(net, 'upsample_align_corners', True)
    prev_dtype = getattr(net, 'compute_dtype', None)
    try:
        if prev_dtype is not None:
            net.compute_dtype = torch.float32                  # the perturbation is below bf16 resolution
        with torch.no_grad():
            y_lo = net.forward_lowres(x.float())
        if eps0 is None:
            eps0 = normalized_noise_like(x, 1.0e-6 * x.shape[2] * x.shape[3] / 1000, generator)
        eps = eps0.detach().clone().float().requires_grad_(True)
        with torch.enable_grad(), _DataGradOnly(net):
            e_lo = net.forward_lowres(x_hat.detach().float() + eps)
            cfg = ops.ConsistencyConfig(mode='cut', loss_fn=cons_loss_fn, conf_thresh=0.0, conf_per_pixel=False,
                                        align_corners=align, invert=False)
            sc, ctx = ops.consistency_forward(cfg, e_lo.detach(), y_lo, None, out_size,
                                              ranges=_ones_ranges(x.shape[0], x.device), sync_conf_rate=False)
            g_lo = ops.consistency_backward(ctx, sc)
            g, = torch.autograd.grad(outputs=e_lo, inputs=eps, grad_outputs=g_lo.to(e_lo.dtype))
    finally:
        if prev_dtype is not None:
            net.compute_dtype = prev_dtype
    return normalize_eps(g.float()), y_lo


def vat_radius_of(x_hat, vat_radius, adaptive):
    """:277-299 -- a scalar, or a per-sample (N,1,1,1) tensor from the image Jacobian."""
    if adaptive:
        xf = x_hat.float()
        dv = (xf[:, :, 2:, :] - xf[:, :, :-2, :]).reshape(len(xf), -1)
        dh = (xf[:, :, :, 2:] - xf[:, :, :, :-2]).reshape(len(xf), -1)
        return vat_radius * torch.sqrt((dv ** 2).sum(dim=1) + (dh ** 2).sum(dim=1))[:, None, None, None] * 0.5
    return vat_radius * math.sqrt(float(x_hat.shape[1] * x_hat.shape[2] * x_hat.shape[3]))


def vat_perturbation(net, x, x_hat, vat_radius=0.5, adaptive=False, cons_loss_fn='kld', eps0=None, generator=None):
    d, y_lo = vat_direction(net, x, x_hat, cons_loss_fn, eps0, generator)
    return (d * vat_radius_of(x_hat, vat_radius, adaptive)).detach(), y_lo


class VATConfig(object):
    def __init__(self, vat_radius=0.5, adaptive_vat_radius=False, cons_loss_fn='kld', cons_weight=1.0, conf_thresh=0.97,
                 conf_per_pixel=False, rampup=-1, unsup_batch_ratio=1):
        if cons_loss_fn not in ('var', 'bce', 'kld', 'logits_var'):
            raise ValueError('Unknown consistency loss function {}'.format(cons_loss_fn))
        self.vat_radius = float(vat_radius)
        self.adaptive = bool(adaptive_vat_radius)
        self.cons_loss_fn = cons_loss_fn
        self.cons_weight = float(cons_weight)
        self.rampup = rampup
        self.unsup_batch_ratio = int(unsup_batch_ratio)
        self.cons = ops.ConsistencyConfig(mode='cut', loss_fn=cons_loss_fn, conf_thresh=conf_thresh,
                                          conf_per_pixel=conf_per_pixel, invert=False)


class VATUnsupBatch(object):
    """x_tea / x_stu: teacher (weakly) and student (strongly) augmented images; um: validity mask (N,1,H,W) or None."""

    def __init__(self, x_tea, x_stu=None, um=None):
        self.x_tea = x_tea
        self.x_stu = x_tea if x_stu is None else x_stu
        self.um = um


class VATMeanTeacherStep(object):
    def __init__(self, student_net, teacher_net, student_optim, teacher_optim, cfg, vat_dir_from_student=False,
                 group=None, generator=None):
        self.student, self.teacher = student_net, teacher_net
        self.student_optim, self.teacher_optim = student_optim, teacher_optim
        self.cfg = cfg
        self.vat_dir_net = student_net if vat_dir_from_student else teacher_net       # :102-105
        self.group = group
        self.generator = generator
        self.align_corners = getattr(student_net, 'upsample_align_corners', True)
        cfg.cons.align_corners = self.align_corners
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # (round 6) the gradient passes as one hipGraph launch (see _graphed_grads)
        # Default ('auto'): ON for networks that run layer by layer through the Python layer engines (the U-Nets: 51.7 -> 76.4 img/s on
        # the DenseNet-161 U-Net, 185.8 -> 209.4 on the ResNet-50 U-Net, profiles/r06bc_*), OFF for the DeepLab networks, whose passes
        # are recorded programs already (114.3 eager vs 108.2 captured). CMS_VAT_GRAPH=0 / 1 forces it.
        env = os.environ.get('CMS_VAT_GRAPH', 'auto')
        layerwise = not any(hasattr(student_net, a) for a in ('_use_hip_body', '_use_hip_backbone'))
        self.use_graph = env == '1' or (env not in ('0', '1') and layerwise)
        self.graph_warmup = 2
        self._graphs = {}

    # ------------------------------------------------------------------------------------------ the gradient passes
    def _grads(self, sup_x, sup_y, unsup_batches, ramp, eps0, teacher_early=False):
        """Everything of the iteration in front of the gradient exchange / optimizer: gradient clear, supervised pass, VAT
        direction, teacher pass, perturbed student pass. -> (ce scalars, [consistency scalars]) on the device.
        `teacher_early` (the captured form): the teacher's passes over the unperturbed images are issued FIRST, on a side stream,
        beside everything up to the consistency loss -- only when the teacher is in eval() mode (it is from the second iteration on
        when it provides the VAT direction, :237; its pass then changes no state and its place in the order is immaterial)."""
        cfg = self.cfg
        out_size = sup_x.shape[2:4]
        self.student_optim.zero_grad()
        early = adv_early = None
        if (teacher_early and cfg.cons_weight > 0.0 and unsup_batches and self.teacher is not self.student and not self.teacher.training
                and sup_x.is_cuda):
            main = torch.cuda.current_stream()
            side = ops.pooled_stream(sup_x.device, 'teacher')
            if side.cuda_stream != main.cuda_stream:
                side.wait_stream(main)
                with torch.cuda.stream(side), torch.no_grad():
                    early = [self.teacher.forward_lowres(ub.x_tea) for ub in unsup_batches]
                # ... and when the direction comes from the teacher too (the default, :102-105), the whole VAT direction pass goes with
                # it: nothing on that stream touches the student, whose supervised forward / backward runs beside it on the main stream
                # (CMS_VAT_GRAPH_DIR_SIDE=0: the direction stays on the main stream)
                if self.vat_dir_net is self.teacher and os.environ.get('CMS_VAT_GRAPH_DIR_SIDE', '1') != '0':
                    with torch.cuda.stream(side):
                        adv_early = []
                        for ub in unsup_batches:
                            x_perturb, _ = vat_perturbation(self.vat_dir_net, ub.x_tea, ub.x_stu, cfg.vat_radius, cfg.adaptive,
                                                            cfg.cons_loss_fn, eps0=eps0, generator=self.generator)
                            adv_early.append((ub.x_stu.float() + x_perturb).to(ub.x_stu.dtype))
        lo = self.student.forward_lowres(sup_x)
        ce_sc, ce_ctx = ops.ce_forward(lo.detach(), sup_y, out_size, 255, self.align_corners, group=self.group)
        lo.backward(ops.ce_backward(ce_ctx, ce_sc).to(lo.dtype))
        cons_vals = []
        if cfg.cons_weight > 0.0:
            if adv_early is not None:
                torch.cuda.current_stream().wait_stream(side)          # the perturbed images (and the teacher's logits)
            for bi, ub in enumerate(unsup_batches):
                if adv_early is not None:
                    x_adv = adv_early[bi]
                else:
                    x_perturb, _ = vat_perturbation(self.vat_dir_net, ub.x_tea, ub.x_stu, cfg.vat_radius, cfg.adaptive,
                                                    cfg.cons_loss_fn, eps0=eps0, generator=self.generator)
                    x_adv = (ub.x_stu.float() + x_perturb).to(ub.x_stu.dtype)
                if early is not None:
                    l_tea = early[bi]
                else:
                    with torch.no_grad():
                        l_tea = self.teacher.forward_lowres(ub.x_tea)
                l_stu = self.student.forward_lowres(x_adv)
                if early is not None and adv_early is None and bi == 0:
                    torch.cuda.current_stream().wait_stream(side)
                sc, cctx = ops.consistency_forward(cfg.cons, l_stu.detach(), l_tea, None, out_size,
                                                   ranges=_ones_ranges(x_adv.shape[0], x_adv.device), um0=ub.um,
                                                   ramp_val=ramp, cons_weight=cfg.cons_weight, group=self.group)
                l_stu.backward(ops.consistency_backward(cctx, sc).to(l_stu.dtype))
                cons_vals.append(sc)
        ops.join_side_streams()
        return ce_sc, cons_vals

    # ------------------------------------------------------------------------------------------ hipGraph replay (round 6)
    def _graph_key(self, sup_x, sup_y, unsup_batches, ramp, eps0):
        def sig(t):
            return None if t is None else (tuple(t.shape), t.dtype, t.device.index)
        return (sig(sup_x), sig(sup_y), tuple((sig(u.x_tea), sig(u.x_stu), sig(u.um), u.x_stu is u.x_tea) for u in unsup_batches),
                float(ramp), sig(eps0), self.student.training, self.teacher.training, self.vat_dir_net.training,
                getattr(self.student, 'compute_dtype', None), getattr(self.teacher, 'compute_dtype', None))

    def _graphed_grads(self, sup_x, sup_y, unsup_batches, ramp, eps0):
        """The gradient passes as ONE hipGraph launch. The layer engines of the U-Nets issue ~9 000 launches per VAT iteration
        through Python autograd: 169 ms of host work per 213 ms step of the DenseNet-161 U-Net (profiles/r06i_*) -- the GPU waits
        for the host. After `graph_warmup` eager iterations of a (shapes, modes, ramp) signature (lazy initialisation, stream probe,
        BatchNorm modes settled: the direction network goes to eval() in its first iteration and stays there, :237) the passes
        are captured once into a torch.cuda.CUDAGraph over static input buffers and replayed: everything inside -- random
        direction (the generator is registered with the graph), batch-statistics BatchNorm incl. running statistics, weight
        gradients into the arena -- is device work of the same kernels. Outside the graph: gradient exchange, optimizer, EMA.
        Not captured: a ramp that still changes (`rampup > 0` while ramp < 1 would need a graph per value: those iterations run
        eagerly), data parallelism (SyncBN host operations)."""
        key = self._graph_key(sup_x, sup_y, unsup_batches, ramp, eps0)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {'seen': 0}
        if 'graph' not in ent:
            ent['seen'] += 1
            if ent['seen'] <= self.graph_warmup or ent.get('failed'):
                return self._grads(sup_x, sup_y, unsup_batches, ramp, eps0)
            # static inputs
            st = {'sup_x': sup_x.clone(), 'sup_y': sup_y.clone(), 'eps0': None if eps0 is None else eps0.clone(), 'ubs': []}
            for u in unsup_batches:
                xt = u.x_tea.clone()
                xs = xt if u.x_stu is u.x_tea else u.x_stu.clone()
                st['ubs'].append(VATUnsupBatch(xt, xs, None if u.um is None else u.um.clone()))
            # operands derived from the weights (padded / transposed copies) are cached per weight version by the eager path: the
            # capture must contain their refresh, so both arenas are marked stale first
            for net in (self.student, self.teacher):
                a = getattr(net, '_cms_arena', None)
                if a is not None:
                    a.touch()
            # everything lazy that synchronises must have happened BEFORE the capture: the side-stream probe above all (a trainer whose
            # eager iterations never asked for a pooled stream met it inside the capture: `operation not permitted when stream is capturing`)
            ops.pooled_stream(sup_x.device, 'teacher')
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if self.generator is not None and hasattr(g, 'register_generator_state'):
                g.register_generator_state(self.generator)
            # one capture stream by default; CMS_VAT_GRAPH_SIDE=1: the layer engines' side streams fork / join inside the capture
            prev = ops.set_side_streams_enabled(os.environ.get('CMS_VAT_GRAPH_SIDE', '0') == '1')
            try:
                with torch.cuda.graph(g):
                    ce_sc, cons_vals = self._grads(st['sup_x'], st['sup_y'], st['ubs'], ramp, st['eps0'],
                                                   teacher_early=os.environ.get('CMS_VAT_GRAPH_TEACHER_EARLY', '1') != '0')
            except Exception as e:               # noqa: BLE001 -- an operation the capture cannot hold (nothing ran on the device)
                import warnings
                warnings.warn('cutmix-semisup-seg_amd: the VAT gradient passes could not be captured into a hipGraph ({}: {}); this '
                              'signature keeps running launch by launch'.format(type(e).__name__, str(e).splitlines()[0] if str(e) else ''),
                              RuntimeWarning, stacklevel=2)
                ent['failed'] = True
                try:
                    torch.cuda.synchronize()
                except Exception as e2:          # noqa: BLE001 -- a forked stream is still inside the aborted capture: this process cannot launch any more
                    raise RuntimeError('a failed hipGraph capture left the device in capture mode ({}); restart with {}=0 (launch by launch) '
                                       'and report the operation named in the warning above'.format(e2, 'CMS_VAT_GRAPH')) from e
                return self._grads(sup_x, sup_y, unsup_batches, ramp, eps0)
            finally:
                ops.set_side_streams_enabled(prev)
            ent.update(graph=g, static=st, out=(ce_sc, cons_vals))
        st = ent['static']
        st['sup_x'].copy_(sup_x)
        st['sup_y'].copy_(sup_y)
        if eps0 is not None:
            st['eps0'].copy_(eps0)
        for su, u in zip(st['ubs'], unsup_batches):
            su.x_tea.copy_(u.x_tea)
            if su.x_stu is not su.x_tea:
                su.x_stu.copy_(u.x_stu)
            if u.um is not None:
                su.um.copy_(u.um)
        ent['graph'].replay()
        ce_sc, cons_vals = ent['out']
        return ce_sc.clone(), [c.clone() for c in cons_vals]

    def __call__(self, sup_x, sup_y, unsup_batches, ramp_val=1.0, eps0=None):
        cfg = self.cfg
        ramp = ramp_val if cfg.rampup > 0 else 1.0
        use_graph = self.use_graph and self.world == 1 and sup_x.is_cuda and (cfg.rampup <= 0 or float(ramp) >= 1.0)
        if use_graph:
            ce_sc, cons_vals = self._graphed_grads(sup_x, sup_y, unsup_batches, ramp, eps0)
        else:
            ce_sc, cons_vals = self._grads(sup_x, sup_y, unsup_batches, ramp, eps0)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.student_optim.arena.grad, op=dist.ReduceOp.SUM, group=self.group)
            self.student_optim.grad_scale = 1.0 / self.world
        self.student_optim.step()
        if self.teacher_optim is not None:
            self.teacher_optim.step()
        res = dict(sup_loss=ce_sc[0], consistency_loss=None, conf_rate=None)
        if cons_vals:
            stacked = torch.stack(cons_vals)
            res['consistency_loss'] = stacked[:, 0].mean()
            res['conf_rate'] = stacked[:, 1].mean()
        return res
