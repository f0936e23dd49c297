import os
import sys
import json

# (round 6) the library (MIOpen) comparison engine is tests/_library_engine.py, plugged in by the tests that A/B against it
# (`net.engine = LibraryEngine(dtype)`); nothing in the product package calls a library convolution, and no environment switch
# enables one

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without a GPU skips the gpu-marked tests instead of failing in them."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason='needs a GPU (MI355X); run with -m gpu on the GPU box')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def load_golden_json(name):
    with open(os.path.join(GOLDEN, name + '.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden():
    return load_golden


@pytest.fixture(scope='session')
def golden_json():
    return load_golden_json


class _NoLibraryConvolutions(object):
    """Context manager: inside it, any call that would reach the library's convolution / BatchNorm kernels (MIOpen through
    torch.nn.functional.conv2d / conv_transpose2d / batch_norm, torch.conv2d, nn.Conv2d.forward, nn.BatchNorm2d.forward)
    raises -- what passes ran on the hand-written kernels only. (A context, not a whole-test patch: the CPU oracle of the
    same test is made of exactly those calls.) `.refused` counts the calls that were turned away."""
    TARGETS = (('torch.nn.functional', 'conv2d'), ('torch.nn.functional', 'conv_transpose2d'),
               ('torch.nn.functional', 'batch_norm'), ('torch', 'conv2d'), ('torch', 'batch_norm'), ('torch', 'convolution'),
               ('torch', 'conv_transpose2d'))

    def __init__(self):
        self.refused = 0
        self._saved = []

    def _refuse(self, name):
        def f(*a, **k):
            self.refused += 1
            raise AssertionError('library kernel reached: ' + name)
        return f

    def __enter__(self):
        import importlib
        import torch
        for modname, name in self.TARGETS:
            mod = importlib.import_module(modname)
            self._saved.append((mod, name, getattr(mod, name)))
            setattr(mod, name, self._refuse(modname + '.' + name))
        for cls in (torch.nn.Conv2d, torch.nn.BatchNorm2d):
            self._saved.append((cls, 'forward', cls.forward))
            cls.forward = self._refuse(cls.__name__ + '.forward')
        return self

    def __exit__(self, *exc):
        for obj, name, val in reversed(self._saved):
            setattr(obj, name, val)
        self._saved = []
        return False


@pytest.fixture
def no_library_convolutions():
    return _NoLibraryConvolutions()
