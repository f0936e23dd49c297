"""
cutmix-semisup-seg_amd -- MI355X-native CutMix mean-teacher training step behind the reference's Python API.

Layout
  csrc/           hand-written HIP kernels (gfx950) + the C ABI declared in include/cutmixseg.h
  _lib.py         ctypes binding of libcutmixseg_hip.so (fails loudly when the library is missing)
  ops.py          tensor-level wrappers / autograd functions over the C ABI
  arena.py        flat fp32 parameter arenas (student, teacher, gradients, optimizer slots, bf16 copies)
  mask_gen.py, optim_weight_ema.py, evaluation.py, lr_schedules.py, job_helper.py,
  architectures/, train_seg_semisup_mask_mt.py
                  host-side mirror of the reference modules of the same names (same classes, signatures,
                  error behaviour), routed to the kernels above
  step.py         the fused student+teacher training iteration used by the trainer and bench.py

Import as `cutmix_semisup_seg_amd` (alias package next to this directory).
"""
import os as _os

# The few convolutions still run by the library (stem; the DeepLab v3+ head) go through MIOpen's solver search on
# their first call. On gfx950 that search also times MIOpen's naive reference kernels -- up to 0.8 s PER CALL, ~45 s of
# GPU time per process for the DeepLab v3+ head (rocprofv3: naive_conv_ab_nonpacked_*). They can never win; take them
# out of the search unless the user has said otherwise.
for _k in ('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', 'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD',
           'MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW'):
    _os.environ.setdefault(_k, '0')

__version__ = '0.1.0'
