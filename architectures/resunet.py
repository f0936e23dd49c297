"""Drop-in module name of the reference (`architectures/resunet.py`); implementation in
cutmix-semisup-seg_amd/architectures/resunet.py."""
from cutmix_semisup_seg_amd.architectures import resunet as _impl

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith('__')})
