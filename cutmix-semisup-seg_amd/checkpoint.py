"""
Checkpoint I/O compatible with the reference.

The reference saves the WHOLE evaluation module, `torch.save(eval_net, model_path)`
(train_seg_semisup_mask_mt.py:533-535), and initialises networks from plain state dicts through
`_load_state_into_model` (architectures/deeplab2.py:310-322). Both directions work here:

  * `export_module(net)` -> a clean CPU replica of a live network: same class, same attribute names and state_dict keys,
    ordinary contiguous tensors in the reference's (Cout, Cin, kh, kw) layout, and none of this build's runtime state
    (parameter arenas, MFMA executors, launch programs, ctypes handles, hooks). `save_model(net, path)` pickles that
    replica exactly like the reference's line does. The network classes pickle under the reference's module path
    (`architectures.deeplab2.ResNetDeepLab`, ...), so the file loads with the REFERENCE's code as well.
  * `load_model(path)` / plain `torch.load(path, weights_only=False)`: whole-module pickles made by the reference (or by
    `save_model`) come back as this build's classes; `__setstate__` adds the runtime attributes the pickle lacks.
  * `architectures.deeplab2._load_state_into_model` keeps the reference's semantics (copy what matches by name AND
    shape, leave the rest at its initialisation, optional verbose report).
"""
import copy
from collections import OrderedDict

import torch

# attributes of this build's runtime that must not travel in a pickle
RUNTIME_ATTRS = ('_hip_executor', '_hip_executors', '_cms_arena', '_hip_engine', '_hip_engines', '_hip_engine_hooked', 'engine',
                 '_sentinel', '_data_grad_only', '_bn_groups')


def export_module(net):
    stripped = []
    for m in net.modules():
        for k in RUNTIME_ATTRS:
            if k in m.__dict__:
                stripped.append((m, k, m.__dict__.pop(k)))
    hooks = []
    for m in net.modules():
        h = m.__dict__.get('_load_state_dict_post_hooks')
        if h:
            hooks.append((m, h))
            m.__dict__['_load_state_dict_post_hooks'] = OrderedDict()
    try:
        replica = copy.deepcopy(net)
    finally:
        for m, k, v in stripped:
            m.__dict__[k] = v
        for m, h in hooks:
            m.__dict__['_load_state_dict_post_hooks'] = h
    replica = replica.cpu()
    with torch.no_grad():
        for m in replica.modules():
            for name, p in list(m._parameters.items()):
                if p is not None:
                    rg = p.requires_grad
                    m._parameters[name] = torch.nn.Parameter(p.detach().contiguous().clone(), requires_grad=rg)
            for name, b in list(m._buffers.items()):
                if b is not None:
                    m._buffers[name] = b.detach().contiguous().clone()
    for m in replica.modules():
        init = getattr(m, '_init_runtime', None)
        if init is not None:
            init()
    return replica


def save_model(net, path):
    """The reference's `torch.save(eval_net, model_path)` for a live network of this build."""
    torch.save(export_module(net), path)


def load_model(path, map_location='cpu'):
    """Whole-module pickle (reference-made or `save_model`-made) -> network object of this build."""
    return torch.load(path, map_location=map_location, weights_only=False)
