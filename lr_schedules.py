"""Drop-in module name of the reference (`lr_schedules.py`); the implementation lives in cutmix-semisup-seg_amd/lr_schedules.py."""
from cutmix_semisup_seg_amd import lr_schedules as _impl

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith('__')})
