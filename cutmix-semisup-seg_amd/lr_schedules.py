"""
Mirror of the reference's lr_schedules.py: PolynomialLR and make_lr_schedulers.

Host-side logic only. The trainer steps schedulers with an explicit index before the optimizer step
(train_seg_semisup_mask_mt.py:258-259, 288-289), so each schedule is evaluated in closed form of that index and
written into `optimizer.param_groups[i]['lr']`; works with any optimizer object exposing `param_groups`
(torch.optim.* or this package's fused optimizers).
"""
import ast
import bisect
import math


class _IndexedLR(object):
    def __init__(self, optimizer, last_epoch=-1):
        self.optimizer = optimizer
        for group in optimizer.param_groups:
            group.setdefault('initial_lr', group['lr'])
        self.base_lrs = [group['initial_lr'] for group in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def _lr_at(self, base_lr, index):
        raise NotImplementedError

    def get_lr(self):
        return [self._lr_at(b, self.last_epoch) for b in self.base_lrs]

    def get_last_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def step(self, epoch=None):
        self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr


class PolynomialLR(_IndexedLR):
    """lr = base * (1 - it/T_max)^power (lr_schedules.py:4-35); it == 0 returns the base rates."""

    def __init__(self, optimizer, T_max, power=0.9, eta_min=0.0, last_epoch=-1):
        self.T_max = T_max
        self.power = power
        self.eta_min = eta_min
        super(PolynomialLR, self).__init__(optimizer, last_epoch)

    def _lr_at(self, base_lr, index):
        if index == 0:
            return base_lr
        progress = min(max(float(index) / float(self.T_max), 0), 1)
        return base_lr * max((1.0 - progress) ** self.power, self.eta_min)


class CosineAnnealingLR(_IndexedLR):
    """Closed form torch.optim.lr_scheduler.CosineAnnealingLR applies for an explicit index."""

    def __init__(self, optimizer, T_max, eta_min=0.0, last_epoch=-1):
        self.T_max = T_max
        self.eta_min = eta_min
        super(CosineAnnealingLR, self).__init__(optimizer, last_epoch)

    def _lr_at(self, base_lr, index):
        return self.eta_min + (base_lr - self.eta_min) * (1 + math.cos(math.pi * index / self.T_max)) / 2


class MultiStepLR(_IndexedLR):
    def __init__(self, optimizer, milestones, gamma=0.1, last_epoch=-1):
        self.milestones = sorted(milestones)
        self.gamma = gamma
        super(MultiStepLR, self).__init__(optimizer, last_epoch)

    def _lr_at(self, base_lr, index):
        return base_lr * self.gamma ** bisect.bisect_right(self.milestones, index)


def make_lr_schedulers(optimizer, total_iters, schedule_type, step_epochs, step_gamma, poly_power=0.9):
    """-> (lr_epoch_scheduler or None, lr_iter_scheduler or None); lr_schedules.py:39-64, including its quirk that
    'stepped' with an empty `step_epochs` string ends in the "Unknown schedule_type" ValueError."""
    lr_epoch_scheduler = None
    lr_iter_scheduler = None
    has_steps = step_epochs is not None and (not isinstance(step_epochs, str) or step_epochs.strip() != '')
    if schedule_type == 'none':
        pass
    elif schedule_type == 'stepped' and has_steps:
        if isinstance(step_epochs, str):
            step_epochs = ast.literal_eval(step_epochs)
        if isinstance(step_epochs, (list, tuple)) and len(step_epochs) > 0:
            lr_epoch_scheduler = MultiStepLR(optimizer, milestones=step_epochs, gamma=step_gamma)
    elif schedule_type == 'cosine':
        lr_iter_scheduler = CosineAnnealingLR(optimizer, T_max=total_iters, eta_min=0.0)
    elif schedule_type == 'poly':
        lr_iter_scheduler = PolynomialLR(optimizer, T_max=total_iters, power=poly_power, eta_min=0.0)
    else:
        raise ValueError('Unknown schedule_type {}'.format(schedule_type))
    return lr_epoch_scheduler, lr_iter_scheduler
