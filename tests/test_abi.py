"""
CPU-only: the C-ABI shared library loads and exports every symbol include/cutmixseg.h declares; the ctypes mirrors
of its structs have the C compiler's sizes. No compute calls (no GPU here).
"""
import ctypes
import os
import re
import subprocess
import tempfile

import pytest

from conftest import REPO

HEADER = os.path.join(REPO, 'include', 'cutmixseg.h')


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cms_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from cutmix_semisup_seg_amd import _lib
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_lib.lib, n), 'libcutmixseg_hip.so does not export {}'.format(n)
    assert sorted(_lib.PROTOTYPES.keys()) == names, 'ctypes prototype table and header disagree'
    assert _lib.version() == 100


def test_struct_layouts_match_the_c_compiler():
    from cutmix_semisup_seg_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "cutmixseg.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(cms_consistency_desc), sizeof(cms_ce_desc), sizeof(cms_param_segment),
         sizeof(cms_optim_desc));
  printf("%zu %zu %zu\n", offsetof(cms_consistency_desc, n), offsetof(cms_consistency_desc, conf_thresh),
         offsetof(cms_optim_desc, grad_scale));
  printf("%zu %zu %zu %zu %zu\n", sizeof(cms_conv_desc), sizeof(cms_wgrad_desc), sizeof(cms_pack_item),
         offsetof(cms_conv_desc, zeros), offsetof(cms_wgrad_desc, stride));
  printf("%zu\n", offsetof(cms_wgrad_desc, dbeta));
  printf("%zu %zu %zu %zu\n", sizeof(cms_bn_op), offsetof(cms_bn_op, count), offsetof(cms_bn_op, eps), sizeof(cms_augment_desc));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 't.c')
        open(src, 'w').write(prog)
        exe = os.path.join(d, 't')
        subprocess.check_call(['gcc', '-I', os.path.join(REPO, 'include'), src, '-o', exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = [int(x) for x in out]
    assert sizes[0] == ctypes.sizeof(_lib.ConsistencyDesc)
    assert sizes[1] == ctypes.sizeof(_lib.CeDesc)
    assert sizes[2] == ctypes.sizeof(_lib.ParamSegment)
    assert sizes[3] == ctypes.sizeof(_lib.OptimDesc)
    assert sizes[4] == _lib.ConsistencyDesc.n.offset
    assert sizes[5] == _lib.ConsistencyDesc.conf_thresh.offset
    assert sizes[6] == _lib.OptimDesc.grad_scale.offset
    assert sizes[7] == ctypes.sizeof(_lib.ConvDesc)
    assert sizes[8] == ctypes.sizeof(_lib.WgradDesc)
    assert sizes[9] == ctypes.sizeof(_lib.PackItem)
    assert sizes[10] == _lib.ConvDesc.zeros.offset
    assert sizes[11] == _lib.WgradDesc.stride.offset
    assert sizes[12] == _lib.WgradDesc.dbeta.offset
    assert sizes[13] == ctypes.sizeof(_lib.BnOp) and sizes[14] == _lib.BnOp.count.offset and sizes[15] == _lib.BnOp.eps.offset
    assert sizes[16] == ctypes.sizeof(_lib.AugmentDesc)


def test_bad_arguments_come_back_as_error_codes_not_crashes():
    """Argument validation happens before any HIP call, so it is checkable without a GPU."""
    from cutmix_semisup_seg_amd import _lib
    rc = _lib.fn['cms_boxmask_rasterize'](None, 1, 1, 4, 4, 1, None, None)
    assert rc == -1
    assert b'NULL' in _lib.fn['cms_last_error']()
    with pytest.raises(ValueError):
        _lib.check(rc, 'cms_boxmask_rasterize')
    d = _lib.ConsistencyDesc()
    assert _lib.fn['cms_consistency_fwd'](ctypes.byref(d), None, None, None) == -1


def test_ops_refuse_cpu_tensors():
    import torch
    from cutmix_semisup_seg_amd import ops
    with pytest.raises(RuntimeError, match='GPU only'):
        ops.cutmix_paste(torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4), ranges=torch.zeros(1, 1, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match='GPU only'):
        ops.ema_flat(torch.zeros(8), torch.zeros(8), 0.99)
