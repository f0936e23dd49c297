"""Drop-in module name of the reference (`evaluation.py`); the implementation lives in cutmix-semisup-seg_amd/evaluation.py."""
from cutmix_semisup_seg_amd import evaluation as _impl

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith('__')})
