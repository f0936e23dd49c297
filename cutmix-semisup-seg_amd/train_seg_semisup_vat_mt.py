"""
Mirror of the reference's train_seg_semisup_vat_mt.py: the VAT mean-teacher (or Pi-model) trainer with the same 52
command-line options (names and defaults, train_seg_semisup_vat_mt.py:592-644), job/log layout and per-epoch log
lines, driving the MI355X VAT iteration (vat.py). SURVEY.md 8(f) rank 2.

As in train_seg_semisup_mask_mt.py of this build: `--synthetic` data only (plus `--synthetic_n_classes`,
`--synthetic_val_batches`, `--compute_dtype`), one process per GPU under torchrun, losses accumulated on the device.
The reference runs this trainer with its DenseNet-161 U-Net (BASELINE configs[4]); that backbone's arithmetic lives in
torchvision and is not part of this build -- the trainer takes any registered architecture, as the reference's does.
"""
import click

from . import job_helper


@job_helper.job('train_seg_semisup_vat_mt', enumerate_job_names=False)
def train_seg_semisup_vat_mt(submit_config, dataset, model, arch, freeze_bn,
                             opt_type, sgd_momentum, sgd_nesterov, sgd_weight_decay,
                             learning_rate, lr_sched, lr_step_epochs, lr_step_gamma, lr_poly_power,
                             teacher_alpha, bin_fill_holes,
                             crop_size, aug_hflip, aug_vflip, aug_hvflip, aug_scale_hung, aug_max_scale,
                             aug_scale_non_uniform, aug_rot_mag,
                             aug_strong_colour, aug_colour_brightness, aug_colour_contrast, aug_colour_saturation,
                             aug_colour_hue, aug_colour_prob, aug_colour_greyscale_prob,
                             vat_radius, adaptive_vat_radius, vat_dir_from_student,
                             cons_loss_fn, cons_weight, conf_thresh, conf_per_pixel, rampup, unsup_batch_ratio,
                             num_epochs, iters_per_epoch, batch_size,
                             n_sup, n_unsup, n_val, split_seed, split_path, val_seed, save_preds, save_model,
                             num_workers,
                             synthetic=False, synthetic_n_classes=21, synthetic_val_batches=2, compute_dtype='bf16'):
    settings = locals().copy()
    del settings['submit_config']

    import os
    import time
    import numpy as np
    import torch
    import torch.distributed as dist
    from .architectures import network_architectures
    from . import evaluation, optim_weight_ema, lr_schedules, optim as fused_optim
    from .vat import VATMeanTeacherStep, VATConfig, VATUnsupBatch

    crop = None if crop_size == '' else [int(x.strip()) for x in crop_size.split(',')]
    if not synthetic:
        raise job_helper.JobNotRun('This build covers the training step, not the dataset pipeline (datapipe/, cv2, dataset ZIPs are out of '
              'scope and absent); run with --synthetic.')
    if crop is None:
        raise ValueError('--synthetic needs a --crop_size')

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise RuntimeError('train_seg_semisup_vat_mt needs a GPU; there is no CPU fallback')
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
        for t in student_net.state_dict().values():
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

    if iters_per_epoch == -1:
        iters_per_epoch = 1000
    total_iters = iters_per_epoch * num_epochs
    lr_epoch_scheduler, lr_iter_scheduler = lr_schedules.make_lr_schedulers(
        optimizer=student_optim, total_iters=total_iters, schedule_type=lr_sched, step_epochs=lr_step_epochs,
        step_gamma=lr_step_gamma, poly_power=lr_poly_power)

    cfg = VATConfig(vat_radius=vat_radius, adaptive_vat_radius=adaptive_vat_radius, cons_loss_fn=cons_loss_fn,
                    cons_weight=cons_weight, conf_thresh=conf_thresh, conf_per_pixel=conf_per_pixel, rampup=rampup,
                    unsup_batch_ratio=unsup_batch_ratio)
    H, W = crop
    gen = torch.Generator(device=torch_device).manual_seed(12345 + rank)
    step = VATMeanTeacherStep(student_net, teacher_net, student_optim, teacher_optim, cfg,
                              vat_dir_from_student=vat_dir_from_student, generator=gen)

    def synth_images():
        return torch.randn(batch_size, 3, H, W, generator=gen, device=torch_device).to(dtype)

    def synth_labels():
        y = torch.randint(0, n_classes, (batch_size, 1, H, W), generator=gen, device=torch_device)
        y[torch.rand(batch_size, 1, H, W, generator=gen, device=torch_device) < 0.05] = 255
        return y.to(torch.uint8)

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

        acc = torch.zeros(3, dtype=torch.float64, device=torch_device)
        n_sup_batches = 0
        n_unsup_batches = 0
        for _ in range(iters_per_epoch):
            if lr_iter_scheduler is not None:
                lr_iter_scheduler.step(iter_i)
            batch_x, batch_y = synth_images(), synth_labels()
            unsup = []
            if cons_weight > 0.0:
                for _r in range(unsup_batch_ratio):
                    x_tea = synth_images()
                    unsup.append(VATUnsupBatch(x_tea, synth_images() if aug_strong_colour else None))
            res = step(batch_x, batch_y, unsup, ramp_val=ramp_val)
            acc[0] += res['sup_loss']
            n_sup_batches += 1
            if res['consistency_loss'] is not None:
                acc[1] += res['consistency_loss']
                if conf_thresh > 0.0:
                    acc[2] += res['conf_rate']
                elif rampup > 0:
                    acc[2] += ramp_val
                n_unsup_batches += len(unsup)
            iter_i += 1

        sums = acc.cpu().numpy()
        sup_loss_acc = sums[0] / max(n_sup_batches, 1)
        consistency_loss_acc = sums[1] / max(n_sup_batches, 1) if n_unsup_batches > 0 else 0.0
        conf_rate_acc = sums[2] / max(n_sup_batches, 1) if n_unsup_batches > 0 else 0.0
        if np.isnan(sup_loss_acc) or np.isnan(consistency_loss_acc):
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

    if save_model and rank == 0 and submit_config.run_dir is not None:
        torch.save(eval_net.state_dict(), os.path.join(submit_config.run_dir, 'model.pth'))


_OPTIONS = [
    click.option('--job_desc', type=str, default=''),
    click.option('--dataset', type=click.Choice(['camvid', 'cityscapes', 'pascal', 'pascal_aug', 'isic2017']),
                 default='pascal_aug'),
    click.option('--model', type=click.Choice(['mean_teacher', 'pi']), default='mean_teacher'),
    click.option('--arch', type=str, default='resnet101_deeplab_imagenet'),
    click.option('--freeze_bn', is_flag=True, default=False),
    click.option('--opt_type', type=click.Choice(['adam', 'sgd']), default='adam'),
    click.option('--sgd_momentum', type=float, default=0.9),
    click.option('--sgd_nesterov', is_flag=True, default=True),
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
    click.option('--vat_radius', type=float, default=0.5),
    click.option('--adaptive_vat_radius', is_flag=True, default=False),
    click.option('--vat_dir_from_student', is_flag=True, default=False),
    click.option('--cons_loss_fn', type=click.Choice(['var', 'bce', 'kld', 'logits_var']), default='kld'),
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
    click.option('--save_preds', is_flag=True, default=False),
    click.option('--save_model', is_flag=True, default=False),
    click.option('--num_workers', type=int, default=4),
    # additions of this build
    click.option('--synthetic', is_flag=True, default=False),
    click.option('--synthetic_n_classes', type=int, default=21),
    click.option('--synthetic_val_batches', type=int, default=2),
    click.option('--compute_dtype', type=click.Choice(['bf16', 'fp32']), default='bf16'),
]


def _with_options(f):
    for opt in reversed(_OPTIONS):
        f = opt(f)
    return f


@click.command()
@_with_options
def experiment(**params):
    train_seg_semisup_vat_mt.submit(**params)


if __name__ == '__main__':
    experiment()
