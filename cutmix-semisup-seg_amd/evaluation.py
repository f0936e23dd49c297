"""
Mirror of the reference's evaluation.py (fast_cm, per_class_i_and_u_cm, EvaluatorIoU) backed by the confusion-matrix
kernels of csrc/eval.hip.

The reference loops over classes in numpy on the host (evaluation.py:24-33; 0.28 s per 321x321 image at 21 classes).
Here truth / prediction maps are histogrammed on the GPU into one C x C int64 matrix; intersection and union follow
from it exactly: I = diag(cm), U = rowsum + colsum - diag (integer identity).
`EvaluatorIoU.sample_logits` is the fused path used by the trainer: bilinear upsample + argmax + histogram straight
from the network's low-resolution logits, nothing full-resolution ever leaves the device.
"""
import numpy as np
import torch

from . import ops


def _to_u8_cuda(a, device, what):
    if torch.is_tensor(a):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    if t.dtype != torch.uint8:
        if t.numel() and (int(t.min()) < 0 or int(t.max()) > 255):
            raise ValueError('{} values must lie in [0, 255]'.format(what))
        t = t.to(torch.uint8)
    return t.to(device, non_blocking=True).contiguous()


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('evaluation runs on the GPU only; no CUDA/HIP device available')
    return torch.device('cuda', torch.cuda.current_device())


def fast_cm(tru, pred, num_classes):
    """Confusion matrix (row = true class, column = predicted class), evaluation.py:6-16."""
    dev = _device()
    cm = ops.confusion(_to_u8_cuda(tru, dev, 'tru'), _to_u8_cuda(pred, dev, 'pred'), num_classes)
    return cm.cpu().numpy()


def per_class_i_and_u_cm(pred, tru, num_classes, ignore_value=None):
    """-> (intersection (C,), union (C,), cm (C,C)) as integer numpy arrays, evaluation.py:18-37."""
    dev = _device()
    cm = ops.confusion(_to_u8_cuda(tru, dev, 'tru'), _to_u8_cuda(pred, dev, 'pred'), num_classes,
                       ignore_index=ignore_value).cpu().numpy()
    diag = np.diag(cm)
    return diag.copy(), cm.sum(axis=0) + cm.sum(axis=1) - diag, cm


class EvaluatorIoU(object):
    def __init__(self, num_classes, fill_holes=False):
        if fill_holes:
            if num_classes != 2:
                raise ValueError('num_classes must be 2 if fill_holes is True')
        self.num_classes = num_classes
        self.fill_holes = fill_holes
        self._cm_dev = None

    def _cm(self):
        if self._cm_dev is None:
            self._cm_dev = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=_device())
        return self._cm_dev

    def sample(self, truth, prediction, ignore_value=None):
        """Accumulate one (H,W) [or batched] pair of integer maps (numpy or torch)."""
        if self.fill_holes:
            # binary post-processing stays a host-side scipy call, as in the reference (evaluation.py:53-55)
            from scipy.ndimage import binary_fill_holes
            p = prediction.cpu().numpy() if torch.is_tensor(prediction) else np.asarray(prediction)
            prediction = binary_fill_holes(p != 0).astype(np.uint8)
        dev = _device()
        ops.confusion(_to_u8_cuda(truth, dev, 'truth'), _to_u8_cuda(prediction, dev, 'prediction'), self.num_classes,
                      ignore_index=ignore_value, cm=self._cm())

    def sample_logits(self, logits, truth, out_size=None, ignore_value=255, align_corners=True):
        """Fused: logits (N,C,h,w) CUDA [low-res or full-res], truth (N,H,W)/(N,1,H,W) uint8/int64 CUDA."""
        if self.fill_holes:
            _, pred = ops.argmax_confusion(logits, None, self.num_classes, out_size or truth.shape[-2:],
                                           align_corners=align_corners, want_pred=True)
            for i in range(pred.shape[0]):
                t = truth[i, 0] if truth.dim() == 4 else truth[i]
                self.sample(t, pred[i], ignore_value=ignore_value)
            return
        ops.argmax_confusion(logits, truth, self.num_classes, out_size, ignore_index=ignore_value,
                             align_corners=align_corners, cm=self._cm())

    # reference attributes (float arrays, evaluation.py:49-51)
    @property
    def cm(self):
        return self._cm().cpu().numpy().astype(np.float64)

    @property
    def intersection(self):
        return np.diag(self.cm).copy()

    @property
    def union(self):
        cm = self.cm
        return cm.sum(axis=0) + cm.sum(axis=1) - np.diag(cm)

    def all_reduce(self, group=None):
        """Sum the confusion matrix over data-parallel ranks (validation set sharded by sample)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self._cm(), op=dist.ReduceOp.SUM, group=group)

    def score(self):
        return self.intersection.astype(float) / np.maximum(self.union.astype(float), 1.0)
