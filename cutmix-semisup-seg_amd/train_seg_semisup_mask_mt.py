"""
Mirror of the reference's train_seg_semisup_mask_mt.py: CutMix / Cutout mean-teacher (or Pi-model) trainer with the
same 56 command-line options (names and defaults, train_seg_semisup_mask_mt.py:581-638), the same job/log layout
(job_helper) and the same per-epoch log lines (:521-530, :576-577), driving the MI355X step (step.py).

Differences, all additive:
  * `--synthetic` (plus `--synthetic_n_classes`, `--synthetic_val_batches`) trains on synthetic tensors of the crop
    shape: there are no datasets, pretrained weights or network in this environment, and the
    reference's CPU data pipeline (datapipe/, cv2 / skimage) is outside the hot path. Without `--synthetic` the
    trainer stops with a clear message. Synthetic runs use `pretrained=False`.
  * `--compute_dtype {bf16,fp32}` and `--no_fuse_batches`.
  * one process per GPU under torchrun (RANK / LOCAL_RANK / WORLD_SIZE); the reference is single-GPU (`cuda:0`, :58).
  * losses are accumulated on the device and read back once per epoch instead of three host syncs per iteration;
    the NaN bail (:469-472) fires one iteration late.
"""
import click

from . import job_helper


@job_helper.job('train_seg_semisup_mask_mt', enumerate_job_names=False)
def train_seg_semisup_mask_mt(submit_config, dataset, model, arch, freeze_bn,
                              opt_type, sgd_momentum, sgd_nesterov, sgd_weight_decay,
                              learning_rate, lr_sched, lr_step_epochs, lr_step_gamma, lr_poly_power,
                              teacher_alpha, bin_fill_holes,
                              crop_size, aug_hflip, aug_vflip, aug_hvflip, aug_scale_hung, aug_max_scale,
                              aug_scale_non_uniform, aug_rot_mag,
                              aug_strong_colour, aug_colour_brightness, aug_colour_contrast, aug_colour_saturation,
                              aug_colour_hue, aug_colour_prob, aug_colour_greyscale_prob,
                              mask_mode, mask_prop_range,
                              boxmask_n_boxes, boxmask_fixed_aspect_ratio, boxmask_by_size, boxmask_outside_bounds,
                              boxmask_no_invert,
                              cons_loss_fn, cons_weight, conf_thresh, conf_per_pixel, rampup, unsup_batch_ratio,
                              num_epochs, iters_per_epoch, batch_size,
                              n_sup, n_unsup, n_val, split_seed, split_path, val_seed, save_preds, save_model,
                              num_workers,
                              synthetic=False, synthetic_n_classes=21, synthetic_val_batches=2, compute_dtype='bf16',
                              no_fuse_batches=False, synthetic_source_size='', deterministic=False,
                              allreduce_dtype='fp32'):
    settings = locals().copy()
    del settings['submit_config']

    if ':' in mask_prop_range:
        lo, hi = mask_prop_range.split(':')
        mask_prop_range = (float(lo.strip()), float(hi.strip()))
    else:
        mask_prop_range = float(mask_prop_range)

    if mask_mode not in ('zero', 'mix'):
        raise ValueError('Unknown mask_mode {}'.format(mask_mode))

    import os
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    from .architectures import network_architectures
    from . import evaluation, optim_weight_ema, mask_gen, lr_schedules, optim as fused_optim, ops
    from .step import CutMixMeanTeacherStep, StepConfig, UnsupBatch

    crop = None if crop_size == '' else [int(x.strip()) for x in crop_size.split(',')]

    if not synthetic:
        raise job_helper.JobNotRun('This build covers the training step, not the dataset pipeline (datapipe/, cv2, dataset ZIPs are out of '
              'scope and absent); run with --synthetic.')
    if crop is None:
        raise ValueError('--synthetic needs a --crop_size')

    # one process per GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise RuntimeError('train_seg_semisup_mask_mt needs a GPU; there is no CPU fallback')
    torch.cuda.set_device(local_rank)
    torch_device = torch.device('cuda', local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group('nccl')
    if world > 1:
        # RCCL creates its internal stream with the first collective; it occupies one of the four hardware queues. Probe the side
        # streams AFTER that, so the step's roles avoid the queue RCCL sits on (ops.probe_streams)
        from cutmix_semisup_seg_amd import ops as _ops
        _t = torch.ones(1, device=torch_device)
        dist.all_reduce(_t)
        torch.cuda.synchronize(torch_device)
        _ops.probe_streams(torch_device, again=True)

    n_classes = int(synthetic_n_classes)
    if bin_fill_holes and n_classes != 2:
        print('Binary hole filling can only be used with binary (2-class) segmentation datasets')
        return
    print('Loaded data')

    NetClass = network_architectures.seg.get(arch)
    student_net = NetClass(n_classes, pretrained=False).to(torch_device)
    dtype = torch.bfloat16 if compute_dtype == 'bf16' else torch.float32
    student_net.compute_dtype = dtype
    if world > 1:
        for t in student_net.state_dict().values():       # identical replicas
            dist.broadcast(t, src=0)

    groups = [dict(params=list(student_net.pretrained_parameters()), lr=learning_rate * 0.1),
              dict(params=list(student_net.new_parameters()), lr=learning_rate)]
    if opt_type == 'adam':
        student_optim = fused_optim.FusedAdam(student_net, groups)
    elif opt_type == 'sgd':
        student_optim = fused_optim.FusedSGD(student_net, groups, momentum=sgd_momentum, nesterov=sgd_nesterov,
                                             weight_decay=sgd_weight_decay)
    else:
        raise ValueError('Unknown opt_type {}'.format(opt_type))

    if model == 'mean_teacher':
        teacher_net = NetClass(n_classes, pretrained=False).to(torch_device)
        teacher_net.compute_dtype = dtype
        for p in teacher_net.parameters():
            p.requires_grad = False
        teacher_optim = optim_weight_ema.EMAWeightOptimizer(teacher_net, student_net, teacher_alpha)
        teacher_optim.fuse_into(student_optim)
        eval_net = teacher_net
    elif model == 'pi':
        teacher_net = student_net
        teacher_optim = None
        eval_net = student_net
    else:
        print('Unknown model type {}'.format(model))
        return

    if freeze_bn and not hasattr(student_net, 'freeze_batchnorm'):
        raise ValueError('Network {} does not support batchnorm freezing'.format(arch))
    print('Built network')

    mask_generator = mask_gen.BoxMaskGenerator(prop_range=mask_prop_range, n_boxes=boxmask_n_boxes,
                                               random_aspect_ratio=not boxmask_fixed_aspect_ratio,
                                               prop_by_area=not boxmask_by_size,
                                               within_bounds=not boxmask_outside_bounds, invert=not boxmask_no_invert)

    if iters_per_epoch == -1:
        iters_per_epoch = 1000
    total_iters = iters_per_epoch * num_epochs
    lr_epoch_scheduler, lr_iter_scheduler = lr_schedules.make_lr_schedulers(
        optimizer=student_optim, total_iters=total_iters, schedule_type=lr_sched, step_epochs=lr_step_epochs,
        step_gamma=lr_step_gamma, poly_power=lr_poly_power)

    step_cfg = StepConfig(mask_mode=mask_mode, cons_loss_fn=cons_loss_fn, cons_weight=cons_weight,
                          conf_thresh=conf_thresh, conf_per_pixel=conf_per_pixel, rampup=rampup,
                          unsup_batch_ratio=unsup_batch_ratio, invert=not boxmask_no_invert,
                          fuse_batches=not no_fuse_batches, compute_dtype=dtype,
                          deterministic=deterministic, allreduce_dtype=allreduce_dtype)
    step = CutMixMeanTeacherStep(student_net, teacher_net, student_optim, teacher_optim, step_cfg)

    # synthetic data: N(0,1) images, uniform labels with 5 % ignore, all-ones validity masks
    H, W = crop
    gen = torch.Generator(device=torch_device).manual_seed(12345 + rank)
    mask_rng = np.random.RandomState(12345 + rank)

    def synth_images():
        return torch.randn(batch_size, 3, H, W, generator=gen, device=torch_device).to(dtype)

    def synth_labels():
        y = torch.randint(0, n_classes, (batch_size, 1, H, W), generator=gen, device=torch_device)
        y[torch.rand(batch_size, 1, H, W, generator=gen, device=torch_device) < 0.05] = 255
        return y.to(torch.uint8)

    # `--synthetic_source_size h,w`: the synthetic samples are uint8 SOURCE images of that size resident in HBM and every
    # batch goes through the device-side input staging (device_pipeline.py: crop / Hung scale / flips / colour
    # augmentation / standardisation -- the reference's loader-worker transforms, :150-183, with the --aug_* options)
    augment = None
    if synthetic_source_size:
        from .device_pipeline import DeviceAugmenter
        hs, ws = [int(v.strip()) for v in synthetic_source_size.split(',')]
        augment = DeviceAugmenter((H, W), student_net.MEAN, student_net.STD, scale_hung=aug_scale_hung,
                                  scale_non_uniform=aug_scale_non_uniform, hflip=aug_hflip, vflip=aug_vflip,
                                  hvflip=aug_hvflip, strong_colour=aug_strong_colour, brightness=aug_colour_brightness,
                                  contrast=aug_colour_contrast, saturation=aug_colour_saturation, hue=aug_colour_hue,
                                  colour_prob=aug_colour_prob, greyscale_prob=aug_colour_greyscale_prob, out_dtype=dtype,
                                  rng=np.random.RandomState(54321 + rank), colour_rng=np.random.RandomState(99 + rank),
                                  rot_mag=aug_rot_mag, max_scale=aug_max_scale)
        src_pool = torch.randint(0, 256, (4 * batch_size, hs, ws, 3), generator=gen, device=torch_device, dtype=torch.uint8)
        lab_pool = torch.randint(0, n_classes, (4 * batch_size, hs, ws), generator=gen, device=torch_device).to(torch.uint8)
        pool_pos = [0]

        def staged(with_labels):
            i = pool_pos[0] % 4
            pool_pos[0] += 1
            sl = slice(i * batch_size, (i + 1) * batch_size)
            return augment(src_pool[sl], lab_pool[sl] if with_labels else None)

    print('Settings:')
    print(', '.join(['{}={}'.format(key, settings[key]) for key in sorted(list(settings.keys()))]))
    print('Dataset:')
    print('synthetic: crop={}x{}, classes={}, world_size={}'.format(H, W, n_classes, world))

    iter_i = 0
    print('Training...')
    for epoch_i in range(num_epochs):
        if lr_epoch_scheduler is not None:
            lr_epoch_scheduler.step(epoch_i)
        t1 = time.time()
        ramp_val = network_architectures.sigmoid_rampup(epoch_i, rampup) if rampup > 0 else 1.0

        student_net.train()
        if teacher_net is not student_net:
            teacher_net.train()
        if freeze_bn:
            student_net.freeze_batchnorm()
            if teacher_net is not student_net:
                teacher_net.freeze_batchnorm()

        acc = torch.zeros(3, dtype=torch.float64, device=torch_device)    # sup, consistency, conf-rate sums
        n_sup_batches = 0
        n_unsup_batches = 0
        for _ in range(iters_per_epoch):
            if lr_iter_scheduler is not None:
                lr_iter_scheduler.step(iter_i)
            if step.nan_detected():
                print('NaN detected; network dead, bailing.')
                return
            if augment is not None:
                sb = staged(True)
                batch_x, batch_y = sb['image'], sb['labels']
            else:
                batch_x, batch_y = synth_images(), synth_labels()
            unsup = []
            if cons_weight > 0.0:
                for _r in range(unsup_batch_ratio):
                    rng_np = mask_generator.generate_ranges(batch_size, (H, W), rng=mask_rng)
                    ranges = ops.ranges_to_device(rng_np, torch_device)
                    if augment is not None:
                        u0 = staged(False)
                        u1 = staged(False) if step_cfg.mix else None
                        unsup.append(UnsupBatch(u0['image'], ranges, um0=u0['mask'],
                                                x1_tea=None if u1 is None else u1['image'],
                                                um1=None if u1 is None else u1['mask'], x0_stu=u0.get('image_stu'),
                                                x1_stu=None if u1 is None else u1.get('image_stu')))
                        continue
                    x0 = synth_images()
                    x1 = synth_images() if step_cfg.mix else None
                    x0s = synth_images() if aug_strong_colour else None
                    x1s = synth_images() if (aug_strong_colour and step_cfg.mix) else None
                    unsup.append(UnsupBatch(x0, ranges, x1_tea=x1, x0_stu=x0s, x1_stu=x1s))
            res = step(batch_x, batch_y, unsup, ramp_val=ramp_val)
            acc[0] += res['sup_loss']
            n_sup_batches += 1
            if res['consistency_loss'] is not None:
                acc[1] += res['consistency_loss']
                if conf_thresh > 0.0:
                    acc[2] += res['conf_rate']
                elif rampup > 0:
                    acc[2] += ramp_val          # reference quirk (:419-420)
                n_unsup_batches += len(unsup)
            iter_i += 1

        sums = acc.cpu().numpy()                                           # the one host sync of the epoch
        sup_loss_acc = sums[0] / max(n_sup_batches, 1)
        consistency_loss_acc = sums[1] / max(n_sup_batches, 1) if n_unsup_batches > 0 else 0.0
        conf_rate_acc = sums[2] / max(n_sup_batches, 1) if n_unsup_batches > 0 else 0.0
        if np.isnan(sup_loss_acc):
            print('NaN detected; network dead, bailing.')
            return

        eval_net.eval()
        tgt_iou_eval = evaluation.EvaluatorIoU(n_classes, bin_fill_holes)
        with torch.no_grad():
            for _b in range(synthetic_val_batches):
                vx, vy = synth_images(), synth_labels()
                tgt_iou_eval.sample_logits(eval_net.forward_lowres(vx), vy, (H, W), ignore_value=255,
                                           align_corners=step.align_corners)
        tgt_iou_eval.all_reduce()
        tgt_iou = tgt_iou_eval.score()
        tgt_miou = tgt_iou.mean()
        t2 = time.time()
        if rank == 0:
            print('Epoch {}: took {:.3f}s, TRAIN clf loss={:.6f}, consistency loss={:.6f}, conf rate={:.3%}, '
                  'VAL mIoU={:.3%}'.format(epoch_i + 1, t2 - t1, sup_loss_acc, consistency_loss_acc, conf_rate_acc,
                                           tgt_miou))
            print('-- {}'.format(', '.join(['{:.3%}'.format(x) for x in tgt_iou])))
            print('-- {:.2f} img/s ({} GPU{})'.format(iters_per_epoch * batch_size * world / max(t2 - t1, 1e-9), world,
                                                      's' if world > 1 else ''))

    if save_model and rank == 0 and submit_config.run_dir is not None:
        # the reference pickles the whole module (:533-535): a clean replica without this build's runtime state, under
        # the reference's class paths (checkpoint.py)
        from . import checkpoint
        model_path = os.path.join(submit_config.run_dir, 'model.pth')
        checkpoint.save_model(eval_net, model_path)


_OPTIONS = [
    click.option('--job_desc', type=str, default=''),
    click.option('--dataset', type=click.Choice(['camvid', 'cityscapes', 'pascal', 'pascal_aug', 'isic2017']),
                 default='pascal_aug'),
    click.option('--model', type=click.Choice(['mean_teacher', 'pi']), default='mean_teacher'),
    click.option('--arch', type=str, default='resnet101_deeplab_imagenet'),
    click.option('--freeze_bn', is_flag=True, default=False),
    click.option('--opt_type', type=click.Choice(['adam', 'sgd']), default='adam'),
    click.option('--sgd_momentum', type=float, default=0.9),
    click.option('--sgd_nesterov', is_flag=True, default=False),
    click.option('--sgd_weight_decay', type=float, default=5e-4),
    click.option('--learning_rate', type=float, default=1e-4),
    click.option('--lr_sched', type=click.Choice(['none', 'stepped', 'cosine', 'poly']), default='none'),
    click.option('--lr_step_epochs', type=str, default=''),
    click.option('--lr_step_gamma', type=float, default=0.1),
    click.option('--lr_poly_power', type=float, default=0.9),
    click.option('--teacher_alpha', type=float, default=0.99),
    click.option('--bin_fill_holes', is_flag=True, default=False),
    click.option('--crop_size', type=str, default='321,321'),
    click.option('--aug_hflip', is_flag=True, default=False),
    click.option('--aug_vflip', is_flag=True, default=False),
    click.option('--aug_hvflip', is_flag=True, default=False),
    click.option('--aug_scale_hung', is_flag=True, default=False),
    click.option('--aug_max_scale', type=float, default=1.0),
    click.option('--aug_scale_non_uniform', is_flag=True, default=False),
    click.option('--aug_rot_mag', type=float, default=0.0),
    click.option('--aug_strong_colour', is_flag=True, default=False),
    click.option('--aug_colour_brightness', type=float, default=0.4),
    click.option('--aug_colour_contrast', type=float, default=0.4),
    click.option('--aug_colour_saturation', type=float, default=0.4),
    click.option('--aug_colour_hue', type=float, default=0.1),
    click.option('--aug_colour_prob', type=float, default=0.8),
    click.option('--aug_colour_greyscale_prob', type=float, default=0.2),
    click.option('--mask_mode', type=click.Choice(['zero', 'mix']), default='mix'),
    click.option('--mask_prop_range', type=str, default='0.5'),
    click.option('--boxmask_n_boxes', type=int, default=1),
    click.option('--boxmask_fixed_aspect_ratio', is_flag=True, default=False),
    click.option('--boxmask_by_size', is_flag=True, default=False),
    click.option('--boxmask_outside_bounds', is_flag=True, default=False),
    click.option('--boxmask_no_invert', is_flag=True, default=False),
    click.option('--cons_loss_fn', type=click.Choice(['var', 'bce', 'kld', 'logits_var', 'logits_smoothl1']),
                 default='var'),
    click.option('--cons_weight', type=float, default=1.0),
    click.option('--conf_thresh', type=float, default=0.97),
    click.option('--conf_per_pixel', is_flag=True, default=False),
    click.option('--rampup', type=int, default=-1),
    click.option('--unsup_batch_ratio', type=int, default=1),
    click.option('--num_epochs', type=int, default=300),
    click.option('--iters_per_epoch', type=int, default=-1),
    click.option('--batch_size', type=int, default=10),
    click.option('--n_sup', type=int, default=100),
    click.option('--n_unsup', type=int, default=-1),
    click.option('--n_val', type=int, default=-1),
    click.option('--split_seed', type=int, default=12345),
    click.option('--split_path', type=click.Path(readable=True, exists=True)),
    click.option('--val_seed', type=int, default=131),
    click.option('--synthetic_source_size', type=str, default=''),
    click.option('--save_preds', is_flag=True, default=False),
    click.option('--save_model', is_flag=True, default=False),
    click.option('--num_workers', type=int, default=4),
    # additions of this build
    click.option('--synthetic', is_flag=True, default=False),
    click.option('--synthetic_n_classes', type=int, default=21),
    click.option('--synthetic_val_batches', type=int, default=2),
    click.option('--compute_dtype', type=click.Choice(['bf16', 'fp32']), default='bf16'),
    click.option('--no_fuse_batches', is_flag=True, default=False),
    # run-to-run deterministic weight gradients (slab + ordered reduce instead of fp32 atomics; 1.5-2 % slower)
    click.option('--deterministic', is_flag=True, default=False),
    # data-parallel gradient exchange: the fp32 arena (default) or a bf16 staging copy (half the bytes on xGMI)
    click.option('--allreduce_dtype', type=click.Choice(['fp32', 'bf16']), default='fp32'),
]


def _with_options(f):
    for opt in reversed(_OPTIONS):
        f = opt(f)
    return f


@click.command()
@_with_options
def experiment(**params):
    train_seg_semisup_mask_mt.submit(**params)


if __name__ == '__main__':
    experiment()
