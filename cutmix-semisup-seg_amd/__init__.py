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
__version__ = '0.1.0'
