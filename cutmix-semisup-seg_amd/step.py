"""
One CutMix mean-teacher training iteration on the GPU -- the body of the reference's hot loop,
train_seg_semisup_mask_mt.py:287-476, as a reusable object (the trainer and bench.py both drive it).

Reference order (per iteration):                                        here
  lr scheduler step, zero_grad                         :288-290          caller / arena memset
  student(x_sup) -> CE(ignore 255) -> backward         :296-301          fused CE kernel on low-res logits
  paste images + validity masks with the box mask      :346-351          cms_cutmix_paste (mask rasterised in-kernel)
  teacher(x0), teacher(x1) under no_grad               :354-356          ONE teacher pass over [x0; x1]      (*)
  student(x_mix)                                       :358              ONE student pass over [x_sup; x_mix] (*)
  paste teacher logits, softmax x2, confidence, loss   :363-458          fused consistency kernel (fwd, finalize, bwd)
  unsup_loss.backward()                                :459              one backward for both losses        (*)
  student_optim.step(); teacher_optim.step()           :465-467          one fused Adam/SGD + EMA kernel
  float(sup_loss) / float(consistency_loss) / float(conf rate)           device scalars, no host sync
  NaN bail                                             :469-472          checked one iteration late (async copy)

(*) With frozen BatchNorm (`--freeze_bn`, the configuration of the reference's run_*_experiments.sh) and no dropout
every sample goes through the networks independently, so concatenating batches and summing the two losses before a
single backward is the same computation with half the kernel launches and twice the GEMM M dimension. With
batch-statistics BN (the reference CLI's default) the batches still travel together when both networks run on the executor
(single process): the BatchNorm kernels keep SAMPLE GROUPS apart -- each of the reference's forward passes is one group with
its own statistics, and the running statistics move once per group in the reference's order (`_sample_groups`, csrc/bn.hip).
Otherwise (DeepLab v3+ head, U-Nets, data parallel, Pi model) the passes are kept separate and in the reference's order.

Data parallel: one process per GPU; gradients of the flat arena are summed with ONE all-reduce (RCCL) and scaled by
1/world inside the optimizer kernel; the confidence count is all-reduced (8 bytes) so that the default
"scalar confidence rate" mode uses the global rate.
"""
import os
import torch

from . import ops


class StepConfig(object):
    def __init__(self, mask_mode='mix', cons_loss_fn='var', cons_weight=1.0, conf_thresh=0.97, conf_per_pixel=False,
                 rampup=-1, unsup_batch_ratio=1, invert=True, fuse_batches=True, compute_dtype=torch.bfloat16,
                 overlap_teacher=True, bucketed_allreduce=True, allreduce_dtype='fp32', deterministic=False,
                 early_optimizer=False):
        if mask_mode not in ('mix', 'zero', 'cut'):
            raise ValueError('Unknown mask_mode {}'.format(mask_mode))
        self.mix = mask_mode == 'mix'
        self.cons_weight = float(cons_weight)
        self.rampup = rampup
        self.unsup_batch_ratio = int(unsup_batch_ratio)
        self.fuse_batches = bool(fuse_batches)
        self.overlap_teacher = bool(overlap_teacher)
        self.bucketed_allreduce = bool(bucketed_allreduce)
        # data-parallel gradient exchange: 'fp32' (the arena itself, 177 MB per step for DeepLab v2) or 'bf16' (a bf16
        # staging copy: 86 MB on the xGMI links; the sum over ranks is then rounded -- not the parity
        # configuration)
        if allreduce_dtype not in ('fp32', 'bf16'):
            raise ValueError('allreduce_dtype must be fp32 or bf16')
        self.allreduce_dtype = allreduce_dtype
        # run-to-run deterministic weight gradients (ops.set_deterministic_wgrad): 1.5-2 % slower inside the step
        self.deterministic = bool(deterministic)
        # single process: optimizer + EMA launches per finished gradient slice during the backward pass (_arm_early_optimizer);
        # measured: no gain (507 / 509 vs 507 / 505 img/s, profiles/r03eo_*) -- the 0.28 ms HBM-bound update only trades bandwidth
        # with the convolutions it overlaps -- so it is off by default
        self.early_optimizer = bool(early_optimizer)
        # (round 5) the TAIL version of it: ONE early launch, for everything but the stem's slice, on a third stream as soon as the
        # body's last weight gradient is enqueued -- the 0.3 ms HBM-bound update then runs beside the max-pool / stem backward
        # (MFMA / LDS-bound, 0.2 ms) instead of alone behind them; CMS_TAIL_OPT=0 switches it off (A/B)
        self.tail_optimizer = os.environ.get('CMS_TAIL_OPT', '1') != '0'
        # fused-batch step on the executor: the main stream joins the weight-gradient stream at the optimizer instead of at the
        # end of the body's backward pass (the stem's backward then overlaps layer1's last weight gradients); CMS_DEFER_JOIN=0
        # switches it off (A/B)
        self.defer_wgrad_join = os.environ.get('CMS_DEFER_JOIN', '1') != '0'
        # the consistency branch of the loss on the teacher's stream, concurrently with the cross entropy on the main stream
        self.overlap_losses = os.environ.get('CMS_OVERLAP_LOSSES', '1') != '0'
        # the teacher's stream forks at the head of the step instead of behind the paste / concatenation of the student's inputs (A/B)
        self.early_teacher_fork = os.environ.get('CMS_EARLY_TEACHER_FORK', '1') != '0'
        # (measured: 626.1 vs 629.2 img/s at cfg 2, 134.2 vs 134.9 at cfg 3 WITHOUT it, profiles/r05f_*: the two halves slow each other
        # and the cross entropy down by more than the overlap buys -- off; CMS_SPLIT_CONS_BWD=1 switches it on)
        self.split_cons_bwd = os.environ.get('CMS_SPLIT_CONS_BWD', '0') not in ('0', '')
        # (round 6) each loss as ONE launch (forward + backward, the gradient's scalar factor applied afterwards: ops.consistency_fused
        # / ops.ce_fused) in the fused-batch step; CMS_FUSED_LOSSES=0 = the forward / backward launch pairs of rounds 1-5 (A/B)
        self.fused_losses = os.environ.get('CMS_FUSED_LOSSES', '1') != '0'
        # (round 6) the step's gradient clear on the first weight-gradient stream instead of at the head of the main stream (A/B)
        self.zero_grad_side = os.environ.get('CMS_ZERO_GRAD_SIDE', '1') != '0'
        self.compute_dtype = compute_dtype
        self.cons = ops.ConsistencyConfig(mode='mix' if self.mix else 'cut', loss_fn=cons_loss_fn,
                                          conf_thresh=conf_thresh, conf_per_pixel=conf_per_pixel, invert=invert)


class UnsupBatch(object):
    """
    One unsupervised batch in the reference's layout. `ranges` replaces the full-resolution
    `mask_params`; `*_stu` default to the teacher images (no colour augmentation pair).
    """

    def __init__(self, x0_tea, ranges, um0=None, x1_tea=None, um1=None, x0_stu=None, x1_stu=None):
        self.x0_tea, self.x1_tea = x0_tea, x1_tea
        self.x0_stu = x0_tea if x0_stu is None else x0_stu
        self.x1_stu = x1_tea if x1_stu is None else x1_stu
        self.um0, self.um1 = um0, um1
        self.ranges = ranges


class GradBuckets(object):
    """
    Bucketed gradient all-reduce overlapped with the backward pass (SURVEY.md 8(e), row "gradients").

    The flat gradient arena is laid out in state_dict order (stem, layer1 .. layer4, head) and the backward pass
    finishes it from the END towards the start, one bottleneck at a time. `on_block(bi)` is called by the executor
    right after the weight gradients of bottleneck `bi` were enqueued (on the stream they run on); when `bi` opens a
    bucket, the finished slice [offset(bi), previous bucket start) is all-reduced asynchronously -- RCCL then works
    on its own stream while the data-gradient chain continues. `finish()` reduces what is left (the stem, whose
    gradient comes last) and makes the current stream wait for all of it. Every element is reduced exactly once.
    """

    def __init__(self, grad, block_offsets, bucket_starts, group=None, dtype='fp32', timing=False):
        self.grad = grad
        self.block_offsets = list(block_offsets)
        self.starts = set(int(b) for b in bucket_starts)
        self.group = group
        self.hi = int(grad.numel())
        self.works = []
        # bf16 exchange: each bucket is copied into a bf16 staging arena, reduced there, and copied back after its wait
        self.stage = torch.empty(grad.numel(), dtype=torch.bfloat16, device=grad.device) if dtype == 'bf16' else None
        # per-bucket record for the bench line: (bytes on the wire, event at issue, event after the wait)
        self.timing = bool(timing)
        self.records = []

    def begin(self):
        self.hi = int(self.grad.numel())
        self.works = []
        self.records = []

    def _issue(self, lo, hi):
        import torch.distributed as dist
        ops.join_side_streams()          # weight gradients of the layer engines on side streams (ops.layer_wgrad_stream)
        if self.stage is not None:
            buf = self.stage[lo:hi]
            buf.copy_(self.grad[lo:hi])
        else:
            buf = self.grad[lo:hi]
        ev = None
        if self.timing:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.works.append((work, lo, hi, ev, buf.numel() * buf.element_size()))

    def on_block(self, bi):
        if bi not in self.starts:
            return
        lo = int(self.block_offsets[bi])
        if lo < self.hi:
            self._issue(lo, self.hi)
            self.hi = lo

    def finish(self):
        if self.hi > 0:
            self._issue(0, self.hi)
            self.hi = 0
        for work, lo, hi, ev, nbytes in self.works:
            work.wait()
            if self.stage is not None:
                self.grad[lo:hi].copy_(self.stage[lo:hi])
            if ev is not None:
                done = torch.cuda.Event(enable_timing=True)
                done.record()
                self.records.append((nbytes, ev, done))
        self.works = []

    def read_timing(self):
        """[{'bytes', 'issue_to_wait_ms'}] of the buckets of the last step (waits for their events)."""
        out = []
        for nbytes, e0, e1 in self.records:
            e1.synchronize()
            out.append({'bytes': int(nbytes), 'issue_to_wait_ms': float(e0.elapsed_time(e1))})
        return out


class CutMixMeanTeacherStep(object):
    def __init__(self, student_net, teacher_net, student_optim, teacher_optim, cfg, group=None):
        self.student = student_net
        self.teacher = teacher_net
        self.student_optim = student_optim
        self.teacher_optim = teacher_optim
        self.cfg = cfg
        self.group = group
        if group is not None:
            # the executors' SyncBN exchanges must run over the SAME ranks as the gradient exchange
            student_net.dist_group = teacher_net.dist_group = group
        self.align_corners = getattr(student_net, 'upsample_align_corners', True)
        cfg.cons.align_corners = self.align_corners
        import torch.distributed as dist
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self._nan_probe = None
        self._nan_event = None
        self._side = None
        self._buckets = None
        self._bucket_obj = None
        self._whole = None
        self.time_buckets = False           # bench.py: keep a (bytes, issue-to-wait) record per gradient bucket
        if cfg.deterministic:
            ops.set_deterministic_wgrad(True)

    # ------------------------------------------------------------------------------------------ helpers
    def _allreduce_grads(self):
        if self.world > 1:
            if self._buckets is not None:
                self._buckets.finish()
            else:
                # one exchange of the whole arena after the backward pass(es) (non-fused steps, engines without hooks)
                if self._whole is None:
                    g = self.student_optim.arena.grad
                    self._whole = GradBuckets(g, [0], [0], group=self.group, dtype=self.cfg.allreduce_dtype,
                                              timing=self.time_buckets)
                self._whole.timing = self.time_buckets
                self._whole.begin()
                self._whole.finish()
            self.student_optim.grad_scale = 1.0 / self.world

    def bucket_timing(self):
        """Per-bucket records of the last step's gradient exchange (empty on one GPU / when `time_buckets` is off)."""
        b = self._buckets if self._buckets is not None else self._whole
        return [] if b is None else b.read_timing()

    def _arm_buckets(self):
        """Overlap the gradient all-reduce with the (single) backward pass of the fused-batch step on the executor."""
        self._buckets = None
        if self.world <= 1 or not self.cfg.fuse_batches or not self.cfg.bucketed_allreduce:
            return None
        use_hip = getattr(self.student, '_use_hip_body', None)
        if use_hip is None or not use_hip():
            return None
        ex = self.student.hip_executor()
        if self._bucket_obj is None:
            offs = ex.block_grad_offsets()
            # [layer4 + head], the two halves of layer3, [layer1 - layer2]: the executor orders its weight-gradient
            # streams at exactly these bottlenecks (backbone_hip.bucket_starts)
            self._bucket_obj = GradBuckets(self.student_optim.arena.grad, offs, ex.bucket_starts(), group=self.group,
                                           dtype=self.cfg.allreduce_dtype, timing=self.time_buckets)
        self._bucket_obj.timing = self.time_buckets
        self._buckets = self._bucket_obj
        self._buckets.begin()
        hook = self._buckets.on_block
        if self.__dict__.get('_bucket_hook') is None or self._bucket_hook[0] is not self._buckets:
            buckets = self._buckets

            def hook(bi):
                buckets.on_block(bi)
            hook.blocks = set(buckets.starts)       # the executor cuts its recorded backward pass at the bucket boundaries only
            self._bucket_hook = (buckets, hook)
        ex.grad_hook = self._bucket_hook[1]
        return ex

    def _arm_early_optimizer(self):
        """Single process: the fused optimizer + EMA kernel for a slice of the arena is issued on the weight-gradient stream
        as soon as that slice's gradients are final -- [layer4 + head], the two halves of layer3, [layer1 - layer2], at the
        bottlenecks the data-parallel buckets close at -- instead of one 0.28 ms launch after the whole backward pass; the
        stem's slice follows in `student_optim.step()`. Same arithmetic per element, so the results are bit-identical."""
        every = getattr(self.cfg, 'early_optimizer', False)
        if self.world > 1 or not (every or getattr(self.cfg, 'tail_optimizer', False)) \
                or not hasattr(self.student_optim, 'step_range'):
            return None
        use_hip = getattr(self.student, '_use_hip_body', None)
        if use_hip is None or not use_hip() or not hasattr(self.student.hip_executor(), 'block_grad_offsets'):
            return None
        ex = self.student.hip_executor()
        if not ex.use_programs or not ex._want_w():
            return None
        opt = self.student_optim
        offs = ex.block_grad_offsets()
        if every:
            starts = set(ex.bucket_starts())
        else:
            # where the early launch(es) go. 'l1' (default): one launch behind the LAST weight gradient, beside the stem's backward.
            # 'l2': behind the first bottleneck of layer2 -- 99 % of the parameters (layer2 .. head) are final there and the update
            # runs beside layer1's backward (HBM-heavy itself); layer1's slice follows at the end. CMS_TAIL_OPT_CUT: A/B switch
            cut = os.environ.get('CMS_TAIL_OPT_CUT', 'l1')
            starts = {0}
            if cut in ('l2', 'l3') and hasattr(ex, 'layer_first_blocks'):
                starts.add(ex.layer_first_blocks()[1 if cut == 'l2' else 2])
        state = {'hi': int(opt.arena.flat.numel())}
        opt.begin_ranged()
        if self.__dict__.get('_opt_stream') is None:
            self._opt_stream = ops.pooled_stream(opt.arena.device, 'optimizer')
        side = self._opt_stream

        def on_block(bi):
            # (called on the weight-gradient stream right after the slice's last weight gradient was enqueued there: the
            # update runs on a THIRD stream behind that point, next to the weight gradients of the earlier layers)
            if bi in starts and offs[bi] < state['hi']:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    opt.step_range(offs[bi], state['hi'])
                state['hi'] = int(offs[bi])
        on_block.blocks = starts                  # the executor cuts its recorded backward pass at these bottlenecks only
        ex.grad_hook = on_block
        self._early_armed = True
        return ex

    def _samples_independent(self):
        """Batches may only be concatenated when no layer couples the samples of a batch: every BatchNorm frozen and
        no active dropout, in BOTH networks (DeepLab v3+ keeps batch statistics and dropout in its head even under
        --freeze_bn, deeplab3plus.py:120-121 -> the reference's separate passes are kept for it)."""
        mods = self.__dict__.get('_coupling_modules')
        if mods is None:                  # (the module trees are static: collect the candidates once, test their flags per call)
            mods = self.__dict__['_coupling_modules'] = [
                m for net in (self.student, self.teacher) for m in net.modules()
                if 'BatchNorm' in type(m).__name__ or 'Dropout' in type(m).__name__]
        for m in mods:
            if m.training and ('BatchNorm' in type(m).__name__ or getattr(m, 'p', 0) > 0):
                return False
        return True

    def _sample_groups(self, n_sup, unsup_batches, use_unsup):
        """Batch-statistics BatchNorm couples the samples of a batch -- but only WITHIN a forward pass of the reference. When both
        networks run on kernels that keep sample groups apart (`supports_sample_groups`: DeepLab v2 on the executor; under data
        parallelism every group's statistics are all-reduced on their own: SyncBN), the passes can still travel as one batch: [supervised; mixed_1; ...] through the student and
        [x0_1; x1_1; ...] through the teacher, every group normalised with its own statistics and the running statistics moved
        once per group in the reference's order. -> (student groups, teacher groups) or None. Needs equal group sizes and
        separate student / teacher networks (the Pi model interleaves both kinds of passes through ONE set of running
        statistics). Dropout (DeepLab v3+'s head) draws per element and couples nothing."""
        if use_unsup and self.teacher is self.student:
            return None
        for net in (self.student, self.teacher) if use_unsup else (self.student,):
            ok = getattr(net, 'supports_sample_groups', None)
            if ok is None or not ok():
                return None
        if not use_unsup:
            return (1, 0)
        if any(ub.x0_tea.shape[0] != n_sup or ub.x0_stu.shape[0] != n_sup for ub in unsup_batches):
            return None
        k = len(unsup_batches)
        return (1 + k, (2 if self.cfg.mix else 1) * k)

    def _both_on_executor(self):
        for net in (self.student, self.teacher):
            use = getattr(net, '_use_hip_body', None)
            if use is None or not use() or not hasattr(net, 'stem_nhwc'):
                return False
        return self.teacher is not self.student

    def _teacher_stream(self):
        if self._side is None:
            self._side = ops.pooled_stream(torch.cuda.current_device(), 'teacher')
        return self._side

    def nan_detected(self):
        """True if the supervised loss of an EARLIER iteration was NaN (checked without stalling the stream)."""
        if self._nan_probe is None:
            return False
        if not self._nan_event.query():
            return False
        return bool(torch.isnan(self._nan_probe).any())

    def _post_nan_probe(self, sup_loss):
        if self._nan_probe is None:
            self._nan_probe = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._nan_event = torch.cuda.Event()
        self._nan_probe.copy_(sup_loss.reshape(1), non_blocking=True)
        self._nan_event.record()

    def _zero_grad(self, main):
        """Gradient clear of the step (a 177 MB fill, 36 us): nothing reads or writes the gradient arena before the backward pass,
        8 ms later, so it goes out on the FIRST WEIGHT-GRADIENT STREAM (idle during the forward passes) instead of in front of the
        paste / concatenation / stems of the main stream (round 6). Order: the fill waits for the main stream (last
        step's optimizer has read the gradients), the main stream waits for the fill right before the backward pass -- every
        gradient writer forks from the main stream behind that point. CMS_ZERO_GRAD_SIDE=0: on the main stream, as in rounds 1-5."""
        if not self.cfg.zero_grad_side or not torch.cuda.is_available():
            self.student_optim.zero_grad()
            return
        wg = ops.pooled_stream(torch.cuda.current_device(), 'wgrad0')
        if wg is main:
            self.student_optim.zero_grad()
            return
        wg.wait_stream(main)
        with torch.cuda.stream(wg):
            self.student_optim.zero_grad()
            ev = torch.cuda.Event()
            ev.record(wg)
        self.__dict__['_zero_grad_event'] = ev

    def _student_inputs(self, ub):
        cfg = self.cfg
        if cfg.mix:
            return ops.cutmix_paste(ub.x0_stu, ub.x1_stu, ranges=ub.ranges, invert=cfg.cons.invert)
        return ops.cutmix_paste(None, ub.x0_stu, ranges=ub.ranges, invert=cfg.cons.invert)      # x * m

    # ------------------------------------------------------------------------------------------ separate passes (+ hipGraph replay)
    def _separate_passes(self, sup_x, sup_y, unsup_batches, ramp, out_size, use_unsup, allow_overlap=True):
        """The gradient passes in the reference's order (batch-statistics BN couples the samples of a pass: no concatenation). The
        teacher's passes depend on nothing the student does within the iteration (its weights only move in the EMA at the end),
        so they are issued FIRST, on the side stream, and run concurrently with the student's supervised forward / backward and
        mixed forward; the teacher's own two passes keep their order (they update the same running statistics).
        -> (ce scalars, [consistency scalars])"""
        cfg = self.cfg
        main = torch.cuda.current_stream()
        self.student_optim.zero_grad()
        tea_out = []
        overlap = allow_overlap and use_unsup and cfg.overlap_teacher and self.teacher is not self.student
        if overlap:
            side = self._teacher_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                for ub in unsup_batches:
                    tea_out.append((self.teacher.forward_lowres(ub.x0_tea),
                                    self.teacher.forward_lowres(ub.x1_tea) if cfg.mix else None))
        lo = self.student.forward_lowres(sup_x)
        ce_sc, ce_ctx = ops.ce_forward(lo.detach(), sup_y, out_size, 255, self.align_corners, group=self.group)
        lo.backward(ops.ce_backward(ce_ctx, ce_sc).to(lo.dtype))
        cons_vals = []
        if use_unsup:
            for bi, ub in enumerate(unsup_batches):
                x_stu = self._student_inputs(ub)
                if overlap:
                    l0, l1 = tea_out[bi]
                else:
                    with torch.no_grad():
                        l0 = self.teacher.forward_lowres(ub.x0_tea)
                        l1 = self.teacher.forward_lowres(ub.x1_tea) if cfg.mix else None
                ls = self.student.forward_lowres(x_stu)
                if overlap and bi == 0:
                    main.wait_stream(side)
                sc, cctx = ops.consistency_forward(cfg.cons, ls.detach(), l0, l1, out_size, ranges=ub.ranges,
                                                   um0=ub.um0, um1=ub.um1, ramp_val=ramp,
                                                   cons_weight=cfg.cons_weight, group=self.group)
                ls.backward(ops.consistency_backward(cctx, sc).to(ls.dtype))
                cons_vals.append(sc)
        return ce_sc, cons_vals

    def _graph_wanted(self):
        """hipGraph replay of the separate passes ('auto'): single process, both networks layer-engine networks (the U-Nets: every
        launch goes through Python autograd and the step is HOST-bound -- 51.9 ms of enqueue per 51.9 ms step on the ResNet-50
        U-Net, 158 per 158 on the DenseNet-161 U-Net, tools/unet_cutmix_bench.py). CMS_STEP_GRAPH=0 / 1 forces it."""
        env = os.environ.get('CMS_STEP_GRAPH', 'auto')
        if env in ('0', '1'):
            return env == '1' and self.world == 1
        layerwise = not any(hasattr(n, a) for n in (self.student, self.teacher) for a in ('_use_hip_body', '_use_hip_backbone'))
        return layerwise and self.world == 1

    def _separate_or_graphed(self, sup_x, sup_y, unsup_batches, ramp, out_size, use_unsup):
        """`_separate_passes`, or -- after two eager iterations of a (shapes, modes, ramp) signature -- its capture into ONE
        torch.cuda.CUDAGraph over static input buffers, replayed (see vat.VATMeanTeacherStep._graphed_grads, whose rules apply:
        weight-derived operands marked stale before the capture, one capture stream, no teacher side stream inside, dropout draws
        from the default generator the graph registers; a ramp that still changes runs eagerly; an operation the capture cannot hold
        -> warning + launches). Gradient exchange, optimizer, EMA and the NaN probe stay outside."""
        cfg = self.cfg
        if not (sup_x.is_cuda and self._graph_wanted() and (cfg.rampup <= 0 or float(ramp) >= 1.0)):
            return self._separate_passes(sup_x, sup_y, unsup_batches, ramp, out_size, use_unsup)
        ubs = list(unsup_batches) if use_unsup else []
        flat = [sup_x, sup_y]
        for u in ubs:
            flat += [u.x0_tea, u.x1_tea, u.x0_stu, u.x1_stu, u.ranges, u.um0, u.um1]
        sig = lambda t: None if t is None else (tuple(t.shape), t.dtype, t.device.index)
        alias = tuple(next(j for j, q in enumerate(flat) if q is t) if t is not None else -1 for t in flat)   # which inputs are ONE tensor
        key = (tuple(sig(t) for t in flat), alias, float(ramp), tuple(out_size), bool(use_unsup), self.student.training, self.teacher.training,
               getattr(self.student, 'compute_dtype', None), getattr(self.teacher, 'compute_dtype', None))
        store = self.__dict__.setdefault('_graphs', {})
        ent = store.setdefault(key, {'seen': 0})

        def rebuild(ts):
            out, i = [], 2
            for _ in ubs:
                x0t, x1t, x0s, x1s, rg, m0, m1 = ts[i:i + 7]
                out.append(UnsupBatch(x0t, rg, um0=m0, x1_tea=x1t, um1=m1, x0_stu=x0s, x1_stu=x1s))
                i += 7
            return out

        if 'graph' not in ent:
            ent['seen'] += 1
            if ent['seen'] <= 2 or ent.get('failed'):
                return self._separate_passes(sup_x, sup_y, unsup_batches, ramp, out_size, use_unsup)
            static = []
            for j, t in enumerate(flat):
                static.append(None if t is None else (static[alias[j]] if alias[j] != j else t.clone()))
            for net in (self.student, self.teacher):
                a = getattr(net, '_cms_arena', None)
                if a is not None:
                    a.touch()
            # everything lazy that synchronises must have happened BEFORE the capture: the side-stream probe above all (a trainer whose
            # eager iterations never asked for a pooled stream met it inside the capture: `operation not permitted when stream is capturing`)
            ops.pooled_stream(sup_x.device, 'teacher')
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            prev = ops.set_side_streams_enabled(False)
            try:
                with torch.cuda.graph(g):
                    # the teacher's passes fork to their side stream inside the capture too (one fork, one join: 353 -> 413 img/s on the
                    # ResNet-50 U-Net, 119 -> 140 on the DenseNet-161 U-Net; CMS_STEP_GRAPH_OVERLAP=0: one stream)
                    outs = self._separate_passes(static[0], static[1], rebuild(static), ramp, out_size, use_unsup,
                                                 allow_overlap=os.environ.get('CMS_STEP_GRAPH_OVERLAP', '1') != '0')
            except Exception as e:               # noqa: BLE001 -- nothing ran on the device
                import warnings
                warnings.warn('cutmix-semisup-seg_amd: the gradient passes of the step could not be captured into a hipGraph ({}: {}); '
                              'this signature keeps running launch by launch'.format(type(e).__name__, str(e).splitlines()[0] if str(e) else ''),
                              RuntimeWarning, stacklevel=2)
                ent['failed'] = True
                try:
                    torch.cuda.synchronize()
                except Exception as e2:          # noqa: BLE001 -- a forked stream is still inside the aborted capture: this process cannot launch any more
                    raise RuntimeError('a failed hipGraph capture left the device in capture mode ({}); restart with {}=0 (launch by launch) '
                                       'and report the operation named in the warning above'.format(e2, 'CMS_STEP_GRAPH')) from e
                return self._separate_passes(sup_x, sup_y, unsup_batches, ramp, out_size, use_unsup)
            finally:
                ops.set_side_streams_enabled(prev)
            ent.update(graph=g, static=static, out=outs)
        for j, (st, t) in enumerate(zip(ent['static'], flat)):
            if t is not None and alias[j] == j:
                st.copy_(t)
        ent['graph'].replay()
        ce_sc, cons_vals = ent['out']
        return ce_sc.clone(), [c.clone() for c in cons_vals]

    # ------------------------------------------------------------------------------------------ the iteration
    def __call__(self, sup_x, sup_y, unsup_batches, ramp_val=1.0):
        """
        sup_x (N,3,H,W); sup_y (N,1,H,W) or (N,H,W) uint8/int64 with 255 = ignore; unsup_batches: list of UnsupBatch
        (len == unsup_batch_ratio; may be empty when cons_weight == 0).
        Returns device scalars: dict(sup_loss, consistency_loss, conf_rate) (the latter two averaged over the
        unsupervised batches, None when there are none).
        """
        cfg = self.cfg
        out_size = sup_x.shape[2:4]
        use_unsup = cfg.cons_weight > 0.0 and len(unsup_batches) > 0
        n_sup = sup_x.shape[0]
        ramp = ramp_val if cfg.rampup > 0 else 1.0

        independent = self._samples_independent()
        groups = None if (independent or not cfg.fuse_batches) else self._sample_groups(n_sup, unsup_batches, use_unsup)
        if cfg.fuse_batches and (independent or groups is not None):
            if groups is not None:
                self.student.set_sample_groups(groups[0])
                if use_unsup:
                    self.teacher.set_sample_groups(groups[1])
            self.__dict__['_groups_armed'] = groups is not None      # restored by _fused_iteration_done, whatever raises
        try:
            return self._train_step_body(sup_x, sup_y, unsup_batches, ramp_val, groups, independent, use_unsup, n_sup, ramp,
                                         out_size)
        finally:
            if self.__dict__.pop('_groups_armed', False):
                self.student.set_sample_groups(1)
                self.teacher.set_sample_groups(1)

    def _train_step_body(self, sup_x, sup_y, unsup_batches, ramp_val, groups, independent, use_unsup, n_sup, ramp, out_size):
        cfg = self.cfg
        if cfg.fuse_batches and (independent or groups is not None):
            stu_in = [sup_x]
            tea_in = []
            if use_unsup:
                # the teacher pass only meets the student at the loss: it runs on its own HIP stream, concurrently
                # with the student's forward pass (the two fill each other's launch tails and memory stalls)
                main = torch.cuda.current_stream()
                # Pi model (teacher IS the student): one network, one executor -- its lazily refreshed operand tables
                # (BN affine, ASPP weights) must not be rewritten on one stream while the other reads them
                side = self._teacher_stream() if (cfg.overlap_teacher and self.teacher is not self.student) else main
                early = side is not main and cfg.early_teacher_fork
                if early:
                    # (round 5) the teacher's inputs are the caller's tensors: its stream forks HERE -- behind last step's optimizer /
                    # EMA writes, in front of this step's gradient clear, paste and concatenation on the main stream, which it does not need
                    side.wait_stream(main)
                self._zero_grad(main)
                for ub in unsup_batches:
                    tea_in.append(ub.x0_tea)
                    if cfg.mix:
                        tea_in.append(ub.x1_tea)
                if early:
                    with torch.cuda.stream(side):
                        x_tea = torch.cat(tea_in, dim=0) if len(tea_in) > 1 else tea_in[0]
                for ub in unsup_batches:
                    stu_in.append(self._student_inputs(ub))
                if not early:
                    x_tea = torch.cat(tea_in, dim=0) if len(tea_in) > 1 else tea_in[0]
                    if side is not main:
                        side.wait_stream(main)              # inputs (and last step's optimizer / EMA writes) are ready
            else:
                self.student_optim.zero_grad()
            x_stu = torch.cat(stu_in, dim=0) if len(stu_in) > 1 else stu_in[0]
            if use_unsup and side is not main and self._both_on_executor():
                # both bodies on the MFMA executor: issue them interleaved, bottleneck by bottleneck
                from .backbone_hip import run_body_pair
                with torch.cuda.stream(side), torch.no_grad():
                    xt = self.teacher.stem_nhwc(x_tea)
                xs = self.student.stem_nhwc(x_stu)
                stu_lo, tea_lo = run_body_pair(self.student.hip_executor(), xs, self.teacher.hip_executor(), xt, side)
            else:
                if use_unsup:
                    with torch.cuda.stream(side), torch.no_grad():
                        tea_lo = self.teacher.forward_lowres(x_tea)
                stu_lo = self.student.forward_lowres(x_stu)
            # dense NCHW scratch for the loss kernels (stu_lo itself may be channels-last strided)
            grad_lo = torch.zeros(stu_lo.shape, dtype=torch.float32, device=stu_lo.device)
            lo_det = stu_lo.detach()
            # the supervised loss needs nothing from the teacher: its kernels are issued BEFORE the join and overlap the tail
            # of the teacher's pass on the other stream
            # Two independent branches between the forward passes and the backward pass: the supervised loss (student logits
            # only) and the consistency loss (student + teacher logits). Each is a forward + backward kernel pair that is
            # latency-bound on a few MB -- they write disjoint slices of grad_lo and run CONCURRENTLY: the
            # consistency branch on the teacher's stream (behind the teacher's pass, where its second operand comes from),
            # the cross entropy on the main stream.
            split = use_unsup and side is not main and cfg.overlap_losses
            if split:
                side.wait_stream(main)                  # the student's logits (and grad_lo's zero fill)
            else:
                if use_unsup and side is not main:
                    main.wait_stream(side)
            cons_vals = []
            # (round 5) the consistency branch is the longer of the two (forward 0.12 + backward 0.17 ms against 0.05 + 0.10 for the
            # cross entropy): when the branches run on two streams, the SECOND half of the samples of its backward is issued on the main
            # stream behind the cross entropy (per-pixel work, independent between samples; it only needs the finalised scalars).
            # Measured slower (StepConfig.split_cons_bwd): off by default
            deferred = []                   # (context, scalars, grad rows, (s0, s1), event): backward halves the main stream takes

            def consistency_branch():
                s_off, t_off = n_sup, 0
                for ub in unsup_batches:
                    n = ub.x0_tea.shape[0]
                    l0 = tea_lo[t_off:t_off + n]
                    l1 = tea_lo[t_off + n:t_off + 2 * n] if cfg.mix else None
                    t_off += 2 * n if cfg.mix else n
                    if self.cfg.fused_losses and not (split and self.cfg.split_cons_bwd and n >= 2):
                        # (round 6) loss + gradient in ONE launch; grad_lo's rows are zero here (filled above, disjoint per branch)
                        sc = ops.consistency_fused(cfg.cons, lo_det[s_off:s_off + n], l0, l1, out_size, grad_lo[s_off:s_off + n],
                                                   ranges=ub.ranges, um0=ub.um0, um1=ub.um1, ramp_val=ramp,
                                                   cons_weight=cfg.cons_weight, group=self.group)
                        s_off += n
                        if split and isinstance(sc, torch.Tensor):
                            sc.record_stream(main)
                        cons_vals.append(sc)
                        continue
                    sc, cctx = ops.consistency_forward(cfg.cons, lo_det[s_off:s_off + n], l0, l1, out_size,
                                                       ranges=ub.ranges, um0=ub.um0, um1=ub.um1, ramp_val=ramp,
                                                       cons_weight=cfg.cons_weight, group=self.group)
                    if split and self.cfg.split_cons_bwd and n >= 2:
                        ev = torch.cuda.Event()
                        ev.record()                                       # scalars final on the teacher's stream
                        ops.consistency_backward(cctx, sc, grad_lo[s_off:s_off + n], samples=(0, n // 2))
                        deferred.append((cctx, sc, grad_lo[s_off:s_off + n], (n // 2, n), ev))
                    else:
                        ops.consistency_backward(cctx, sc, grad_lo[s_off:s_off + n])
                    s_off += n
                    if split and isinstance(sc, torch.Tensor):
                        sc.record_stream(main)          # allocated on the teacher's stream, read (and freed) on the main one
                    cons_vals.append(sc)
            if split:
                with torch.cuda.stream(side):
                    consistency_branch()
            if self.cfg.fused_losses:
                ce_sc = ops.ce_fused(lo_det[:n_sup], sup_y, grad_lo[:n_sup], out_size, 255, self.align_corners, group=self.group)
            else:
                ce_sc, ce_ctx = ops.ce_forward(lo_det[:n_sup], sup_y, out_size, 255, self.align_corners, group=self.group)
                ops.ce_backward(ce_ctx, ce_sc, grad_lo[:n_sup])
            for cctx, sc, rows, rng_, ev in deferred:
                main.wait_event(ev)
                ops.consistency_backward(cctx, sc, rows, samples=rng_)
            if split:
                # the re-pack of the data-gradient operands (one HBM-bound launch, ~0.12 ms, needs only the weights) is due in
                # front of the backward pass: issued HERE it runs beside the consistency kernels of the other stream instead
                # of alone behind the join
                pre = getattr(self.student, 'hip_executor', None)
                if independent and groups is None and pre is not None and getattr(self.student, '_use_hip_body', None) \
                        and self.student._use_hip_body() and hasattr(pre(), '_refresh_for_backward'):
                    pre()._refresh_for_backward()        # (frozen-statistics passes: the chain that uses these operands)
                main.wait_stream(side)
            elif use_unsup:
                consistency_branch()
            ex = self._arm_buckets() or self._arm_early_optimizer()
            # the join of the weight-gradient stream moves from the end of the body's backward to where the gradients are next
            # touched (exchange / optimizer, below): the stem's backward overlaps the last weight gradients
            dex = self.student.hip_executor() if (getattr(self.student, '_use_hip_body', None) and self.student._use_hip_body()
                                                  and hasattr(self.student.hip_executor(), 'join_wgrad')) else None
            if dex is not None:
                dex.defer_wgrad_join = self.cfg.defer_wgrad_join
            zev = self.__dict__.pop('_zero_grad_event', None)
            if zev is not None:
                main.wait_event(zev)                    # the gradient clear issued on the weight-gradient stream at the head of the step
            try:
                stu_lo.backward(grad_lo.to(stu_lo.dtype))
            finally:
                if ex is not None:
                    ex.grad_hook = None
                if dex is not None:
                    dex.defer_wgrad_join = False
                    dex.join_wgrad()
        else:
            ce_sc, cons_vals = self._separate_or_graphed(sup_x, sup_y, unsup_batches, ramp, out_size, use_unsup)

        ops.join_side_streams()           # weight gradients the layer engines issued on side streams (ops.layer_wgrad_stream)
        self._allreduce_grads()
        if self.__dict__.pop('_early_armed', False):
            torch.cuda.current_stream().wait_stream(self._opt_stream)       # the early slices of this step's update
        self.student_optim.step()
        if self.teacher_optim is not None:
            self.teacher_optim.step()

        sup_loss = ce_sc[0]
        self._post_nan_probe(sup_loss)
        res = dict(sup_loss=sup_loss, consistency_loss=None, conf_rate=None)
        if cons_vals:
            stacked = torch.stack(cons_vals)
            res['consistency_loss'] = stacked[:, 0].mean()
            res['conf_rate'] = stacked[:, 1].mean()
        return res
