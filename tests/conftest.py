import os
import sys
import json

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
