"""
Mirror of the reference's optim_weight_ema.py (EMAWeightOptimizer).

Reference: a Python loop over the float tensors of both state dicts, three ATen kernels and a temporary per tensor
per step (528 x 3 launches for DeepLab v2), optim_weight_ema.py:21-25. Here both networks' float state is re-homed
into flat fp32 arenas (arena.py) and one kernel applies t = t*alpha + s*(1-alpha) to all 44 M elements with the same
three fp32 roundings (bit exact). When the student is driven by this package's fused optimizer the EMA rides along
in the optimizer kernel (`fuse_into`), saving one more pass over memory; `step()` then does nothing extra.
"""
import torch

from . import ops
from .arena import ensure_arena


class EMAWeightOptimizer(object):
    def __init__(self, target_net, source_net, ema_alpha):
        self.target_net = target_net
        self.source_net = source_net
        self.ema_alpha = ema_alpha
        self.target_arena = ensure_arena(target_net, with_grad=False)
        self.source_arena = ensure_arena(source_net, with_grad=any(p.requires_grad for p in source_net.parameters()))
        self.target_params = [p for p in target_net.state_dict().values() if p.dtype == torch.float]
        self.source_params = [p for p in source_net.state_dict().values() if p.dtype == torch.float]
        self._flat = self.target_arena.same_layout(self.source_arena)

        # initial copy of every float tensor, positional like the reference (optim_weight_ema.py:12-13)
        with torch.no_grad():
            if self._flat:
                self.target_arena.flat.copy_(self.source_arena.flat)
            else:
                for tgt_p, src_p in zip(self.target_params, self.source_params):
                    tgt_p[...] = src_p[...]
        self.target_arena.refresh_bf16()
        self._touch_target()

        target_keys = set(target_net.state_dict().keys())
        source_keys = set(source_net.state_dict().keys())
        if target_keys != source_keys:
            raise ValueError('Source and target networks do not have the same state dict keys; do they have '
                             'different architectures?')
        self._fused_into = None
        self._fused_pending = False

    def fuse_into(self, student_optimizer):
        """Let `student_optimizer.step()` (FusedAdam/FusedSGD of this package) apply the EMA in the same kernel."""
        student_optimizer.attach_ema(self)
        self._fused_into = student_optimizer

    def _touch_target(self):
        # the teacher's weights (and possibly its BN statistics) moved: packed operands of its executor are stale
        from .backbone_hip import executors_of
        if getattr(self, 'target_arena', None) is not None:
            self.target_arena.touch()
        for ex in executors_of(self.target_net):
            ex.weights_changed(bn_too=True)

    def _mark_fused_step_done(self):
        self._fused_pending = True

    def step(self):
        if self._fused_pending:
            # the EMA of this iteration already happened inside the student optimizer's kernel
            self._fused_pending = False
            return
        if not self._flat:
            raise RuntimeError('EMAWeightOptimizer: source and target layouts differ')
        ops.ema_flat(self.target_arena.flat, self.source_arena.flat, self.ema_alpha, self.target_arena.bf16)
        self._touch_target()
