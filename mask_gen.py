"""Drop-in module name of the reference (`mask_gen.py`); the implementation lives in cutmix-semisup-seg_amd/mask_gen.py."""
from cutmix_semisup_seg_amd import mask_gen as _impl

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith('__')})
