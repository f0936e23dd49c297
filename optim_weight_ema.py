"""Drop-in module name of the reference (`optim_weight_ema.py`); the implementation lives in cutmix-semisup-seg_amd/optim_weight_ema.py."""
from cutmix_semisup_seg_amd import optim_weight_ema as _impl

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith('__')})
