"""Mirror of the reference's architectures/util.py."""


def freeze_bn_module(m):
    """Put `m` into eval mode if it is a batch-norm layer (use with `net.apply`), architectures/util.py:2-10."""
    if 'BatchNorm' in type(m).__name__:
        m.eval()
