"""Drop-in module name of the reference (`job_helper.py`); the implementation lives in cutmix-semisup-seg_amd/job_helper.py."""
from cutmix_semisup_seg_amd import job_helper as _impl

globals().update({_k: _v for _k, _v in vars(_impl).items() if not _k.startswith('__')})
