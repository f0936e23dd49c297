"""Drop-in module name of the reference (`architectures/network_architectures.py`); implementation in
cutmix-semisup-seg_amd/architectures/network_architectures.py."""
from cutmix_semisup_seg_amd.architectures import network_architectures as _impl

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith('__')})
