"""
Import alias: the product package lives in the directory `cutmix-semisup-seg_amd/` (a name Python cannot import
directly because of the hyphens). This shim makes it importable as `cutmix_semisup_seg_amd` by pointing the
package search path at that directory and executing its __init__.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'cutmix-semisup-seg_amd')
__path__ = [_real]
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f
