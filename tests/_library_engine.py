"""
TEST INFRASTRUCTURE -- the library (MIOpen via torch) convolution engine the GPU tests A/B the hand-written kernels against.

Until round 5 this class lived inside the product package (`TorchEngine`, behind an environment guard). It is a comparison engine,
not an implementation, so it lives here now: the product package contains no `F.conv2d` / `F.batch_norm` call
(`tests/test_host_api.py::test_product_package_has_no_library_convolution_or_batchnorm_call`). It plugs into a network through the
hook every network of the package has:

    net.engine = LibraryEngine(torch.float32)        # every convolution / classifier / odd BatchNorm through torch's own kernels

`LibraryEngine` derives from the product's `LayerEngine`, whose batch-statistics BatchNorm runs on csrc/bn.hip for device tensors --
the part that is NOT a library call stays the product's; on CPU tensors (tests/test_deeplab3plus_cpu.py) everything is torch.
"""
import torch
import torch.nn.functional as F

from cutmix_semisup_seg_amd.architectures.deeplab2 import LayerEngine


class LibraryEngine(LayerEngine):
    def conv2d(self, x, conv):
        return F.conv2d(x, self._weight(conv), None, conv.stride, conv.padding, conv.dilation, conv.groups)

    def bn_act(self, y, bn, relu, residual=None):
        if bn is not None and bn.training and not self._bn_on_hip(y, bn):
            # CPU tensors / odd channel counts: torch's BatchNorm in fp32
            if int(getattr(self, 'bn_groups', 1)) != 1:
                raise RuntimeError('grouped batch statistics need the csrc/bn.hip path (channels-last, channels % 8 == 0)')
            y = F.batch_norm(y.float(), bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps).to(y.dtype)
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            if residual is not None:
                y = y + residual
            return F.relu(y, inplace=True) if relu else y
        return super(LibraryEngine, self).bn_act(y, bn, relu, residual)

    def aspp_head(self, x, convs):
        out = None
        for conv in convs:
            y = F.conv2d(x, self._weight(conv), None, conv.stride, conv.padding, conv.dilation)
            out = y if out is None else out + y
        bias = sum(c.bias for c in convs)
        return out.float() + bias.view(1, -1, 1, 1)

    def classifier(self, x, conv):
        """1 x 1 convolution with bias -> fp32 logits (what the product's engines compute in a convolution epilogue)."""
        y = F.conv2d(x, conv.weight.to(x.dtype), None)
        return y.float() + conv.bias.view(1, -1, 1, 1)


def use_library_engine(net, dtype=None):
    """Plug the comparison engine into `net` (and return it)."""
    net.engine = LibraryEngine(dtype if dtype is not None else net.compute_dtype)
    return net
