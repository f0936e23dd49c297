"""How many convolutions of a network the default ('auto') bf16 engine sends to the MFMA kernels and how many to the library
(missing 5: DenseNet-161's 48-multiple channel counts). One training-mode forward pass each."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from architectures import network_architectures
from cutmix_semisup_seg_amd import backbone_hip

calls = {'hip': 0}
orig = backbone_hip.hip_conv2d


def counted(*a, **k):
    calls['hip'] += 1
    return orig(*a, **k)


backbone_hip.hip_conv2d = counted
for arch, shape in (('densenet161unet', (2, 3, 224, 224)), ('resnet50unet_imagenet', (2, 3, 256, 256)),
                    ('resnet101_deeplabv3plus_imagenet', (2, 3, 257, 257))):
    torch.manual_seed(0)
    Net = network_architectures.seg.get(arch)
    net = (Net(2) if arch == 'densenet161unet' else Net(2, pretrained=False)).cuda()
    net.train()
    if hasattr(net, 'freeze_batchnorm') and 'deeplab' in arch:
        net.freeze_batchnorm()
    calls['hip'] = 0
    with torch.no_grad():
        net.forward_lowres(torch.randn(*shape, device='cuda').bfloat16())
    eng = net._hip_engine
    n_conv = sum(1 for m in net.modules() if isinstance(m, torch.nn.Conv2d))
    print('{}: {} Conv2d modules; engine {} strict={}: {} launches on the MFMA kernels, {} on the library{}'.format(
        arch, n_conv, type(eng).__name__, getattr(eng, 'strict', None), calls['hip'], getattr(eng, 'library_convs', None),
        ' (+ the backbone on the static executor)' if getattr(net, '_hip_executor', None) is not None else ''))
