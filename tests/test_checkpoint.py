"""
CPU: reference-compatible checkpoint I/O (checkpoint.py).

  * a whole-module pickle WRITTEN BY THE REFERENCE'S CLASSES (tests/golden/ref_module_checkpoint.pth, made by
    tests/golden/make_golden.py the way train_seg_semisup_mask_mt.py:533-535 does; every tensor a stride-0 expansion of
    one tagged element, so the fixture is 48 KB) loads as this build's network, values intact, runtime state added;
  * `_load_state_into_model` has the reference's semantics on a state dict with a missing key, a wrong-shaped entry and a
    foreign key (architectures/deeplab2.py:310-322) -- outcome and verbose output recorded from the reference;
  * `save_model` writes a pickle that names only the reference's class paths and round-trips;
  * (build container only, skipped elsewhere) the REFERENCE's own code loads a checkpoint written here and computes the
    oracle's logits with it.
"""
import io
import json
import os
import pickletools
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, GOLDEN

META = json.load(open(os.path.join(GOLDEN, 'checkpoint_meta.json')))


def test_reference_made_module_pickle_loads_as_this_builds_network():
    from architectures import deeplab2
    from cutmix_semisup_seg_amd import checkpoint
    from oracle import deeplab2 as odl
    net = checkpoint.load_model(os.path.join(GOLDEN, 'ref_module_checkpoint.pth'))
    assert type(net) is deeplab2.ResNetDeepLab
    assert net.num_classes == META['num_classes'] and net.engine_kind == 'auto' and net._hip_executors == {}
    assert net.compute_dtype == torch.bfloat16 and net.BLOCK_SIZE == (1, 1)
    sd = net.state_dict()
    assert list(sd.keys()) == list(odl.state_spec(META['num_classes'], META['layers']).keys())
    for k, v in sd.items():
        if v.dtype == torch.float32:
            assert float(v.min()) == float(v.max()) == pytest.approx(META['tag'][k], abs=1e-7), k
    # parameter groups of the loaded object behave like a constructed one (A3b)
    assert len(list(net.pretrained_parameters())) == len(list(deeplab2.ResNetDeepLab(
        deeplab2.Bottleneck, META['layers'], META['num_classes'], np.zeros(3), np.ones(3)).pretrained_parameters()))


def test_load_state_into_model_keeps_the_references_semantics(capsys):
    from architectures import deeplab2
    from oracle import deeplab2 as odl
    C, layers = META['num_classes'], META['layers']
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    sd = dict(odl.closed_form_state(C, layers))
    del sd['layer2.0.conv2.weight']
    sd['layer5.conv2d_list.0.weight'] = torch.ones(C + 1, 2048, 3, 3)
    sd['not.a.key'] = torch.ones(3)
    deeplab2._load_state_into_model(net, sd, verbose=True)
    out = capsys.readouterr().out.strip().splitlines()
    assert sorted(out) == sorted(META['verbose_lines'])
    after = net.state_dict()
    for k in META['kept_init']:
        assert torch.equal(after[k], before[k]), k
    loaded = [k for k in after if k in sd and after[k].shape == sd[k].shape and torch.equal(after[k], sd[k])]
    assert len(loaded) == META['n_loaded'] and len(after) == META['n_keys']


def _export_bytes():
    from architectures import deeplab2
    from cutmix_semisup_seg_amd import checkpoint
    from oracle import deeplab2 as odl
    C, layers = 5, [1, 1, 1, 1]
    net = deeplab2.ResNetDeepLab(deeplab2.Bottleneck, layers, C, np.zeros(3), np.ones(3))
    net.load_state_dict(odl.closed_form_state(C, layers))
    net._hip_executors = {'fake': object()}          # runtime state that must not travel
    net._cms_arena = object()
    buf = io.BytesIO()
    torch.save(checkpoint.export_module(net), buf)
    assert net._hip_executors and net._cms_arena is not None      # the live network keeps its runtime state
    return net, buf.getvalue()


def test_save_model_writes_a_clean_pickle_under_the_references_class_paths(tmp_path):
    import zipfile
    net, raw = _export_bytes()
    with zipfile.ZipFile(io.BytesIO(raw)) as z:
        pkl = [n for n in z.namelist() if n.endswith('data.pkl')][0]
        names = set()
        for op, arg, _ in pickletools.genops(z.read(pkl)):
            if op.name in ('GLOBAL', 'STACK_GLOBAL') and arg:
                names.add(str(arg))
            if op.name in ('BINUNICODE', 'SHORT_BINUNICODE') and isinstance(arg, str) and '.' in arg:
                names.add(arg)
    mods = {n.split(' ')[0] for n in names}
    assert 'architectures.deeplab2' in mods
    assert not any('cutmix' in m for m in mods), sorted(mods)
    back = torch.load(io.BytesIO(raw), weights_only=False)
    assert back._hip_executors == {} and '_cms_arena' not in back.__dict__
    for (k, a), (_, b) in zip(net.state_dict().items(), back.state_dict().items()):
        assert torch.equal(a, b) and b.is_contiguous(), k


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='needs the reference checkout (build container only)')
def test_reference_code_loads_a_checkpoint_written_here(tmp_path):
    from oracle import deeplab2 as odl
    _, raw = _export_bytes()
    path = tmp_path / 'model.pth'
    path.write_bytes(raw)
    prog = r'''
import sys, types
sys.dont_write_bytecode = True
sys.path = [p for p in sys.path if p not in ('', %r)]
sys.path.insert(0, '/root/reference')
for name in ('torchvision', 'torchvision.models', 'torchvision.models.resnet'):
    sys.modules[name] = types.ModuleType(name)
import torch
from architectures import deeplab2
assert deeplab2.__file__.startswith('/root/reference')
net = torch.load(%r, weights_only=False)
assert type(net) is deeplab2.ResNetDeepLab, type(net)
net.eval()
idx = torch.arange(2 * 3 * 33 * 33, dtype=torch.float64)
x = torch.sin(0.3 + 0.61803398875 * idx).reshape(2, 3, 33, 33).float() * 1.5
with torch.no_grad():
    y = net(x)
torch.save(y, %r)
''' % (REPO, str(path), str(tmp_path / 'y.pt'))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env.pop('PYTHONPATH', None)
    r = subprocess.run([sys.executable, '-c', prog], cwd=str(tmp_path), env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    y = torch.load(tmp_path / 'y.pt')
    idx = torch.arange(2 * 3 * 33 * 33, dtype=torch.float64)
    x = torch.sin(0.3 + 0.61803398875 * idx).reshape(2, 3, 33, 33).float() * 1.5
    want = odl.forward(x, odl.closed_form_state(5, [1, 1, 1, 1]), [1, 1, 1, 1], frozen=True)
    torch.testing.assert_close(y, want, rtol=1e-4, atol=1e-5)


def test_deeplab_v3plus_wrapper_round_trips_with_its_runtime_state_restored():
    """`export_module` strips the runtime attributes; DeepLabv3Wrapper restores them in `_init_runtime`
    / `__setstate__` like the other networks, and the reference's OWN classes (architectures/deeplab3plus.py:26-158)
    pickle under the reference's module path."""
    import zipfile
    from architectures import deeplab3plus as d3
    from cutmix_semisup_seg_amd import checkpoint
    torch.manual_seed(0)
    net = d3.DeepLabv3Wrapper(d3._deeplabv3plus(5, 8, (1, 1, 1, 1)))
    net._hip_executors = {'fake': object()}
    net._hip_engine = object()
    net._cms_arena = object()
    buf = io.BytesIO()
    torch.save(checkpoint.export_module(net), buf)
    assert net._hip_executors and net._cms_arena is not None            # the live network keeps its runtime state
    zf = zipfile.ZipFile(io.BytesIO(buf.getvalue()))
    pkl = zf.read([n for n in zf.namelist() if n.endswith('data.pkl')][0])
    names = {arg for op, arg, _ in pickletools.genops(pkl) if isinstance(arg, str) and 'deeplab3plus' in arg}
    assert {'architectures.deeplab3plus DeepLabv3Wrapper', 'architectures.deeplab3plus DeepLabV3Plus',
            'architectures.deeplab3plus DeepLabHeadV3Plus'} <= names, names
    buf.seek(0)
    back = checkpoint.load_model(buf)
    assert type(back) is d3.DeepLabv3Wrapper
    assert back.engine_kind == 'auto' and back.engine is None and back.compute_dtype == torch.bfloat16
    assert back._hip_executor is None and back._hip_executors == {} and back._hip_engine is None and back._hip_engines == {}
    assert not hasattr(back, '_cms_arena')
    for (k, a), (_, b) in zip(net.state_dict().items(), back.state_dict().items()):
        assert torch.equal(a, b), k
    assert back.pretrained_parameters() == [] and len(back.new_parameters()) == len(list(back.parameters()))
    # the methods the first forward pass calls find their attributes (they raised AttributeError before)
    assert back._use_hip_backbone() in (True, False)
