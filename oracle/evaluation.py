"""
Oracle (test infrastructure): IoU evaluation restated from the reference's evaluation.py.

  confusion()      -- fast_cm, evaluation.py:6-16
  per_class_iu()   -- per_class_i_and_u_cm, evaluation.py:18-37
  IoUAccumulator   -- EvaluatorIoU, evaluation.py:41-62 (without the optional scipy hole filling, :53-55,
                      which is a host-side scipy call the product keeps on the host as well)

Pinned by tests/golden/evaluation_*.npz.
"""
import numpy as np


def confusion(truth, pred, num_classes):
    """Row = truth class, column = predicted class; inputs already restricted to valid pixels."""
    flat = truth.astype(np.int64) * num_classes + pred.astype(np.int64)
    return np.bincount(flat.ravel(), minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def per_class_iu(pred, truth, num_classes, ignore_value=None):
    if ignore_value is None:
        valid = np.ones(truth.shape, dtype=bool)
    else:
        valid = truth != ignore_value
    inter = np.zeros(num_classes, dtype=np.int64)
    union = np.zeros(num_classes, dtype=np.int64)
    for c in range(num_classes):
        p = (pred == c) & valid
        t = (truth == c) & valid
        inter[c] = np.count_nonzero(p & t)
        union[c] = np.count_nonzero(p | t)
    cm = confusion(truth[valid], pred[valid], num_classes)
    return inter, union, cm


def iu_from_confusion(cm):
    """SURVEY.md 8(a) A12 identity: I = diag(cm), U = rowsum + colsum - diag (valid when pred in [0, C))."""
    d = np.diag(cm)
    return d, cm.sum(axis=0) + cm.sum(axis=1) - d


class IoUAccumulator(object):
    def __init__(self, num_classes):
        self.num_classes = num_classes
        self.intersection = np.zeros((num_classes,))
        self.union = np.zeros((num_classes,))
        self.cm = np.zeros((num_classes, num_classes))

    def sample(self, truth, prediction, ignore_value=None):
        i, u, cm = per_class_iu(prediction, truth, self.num_classes, ignore_value)
        self.intersection += i
        self.union += u
        self.cm += cm

    def score(self):
        return self.intersection.astype(float) / np.maximum(self.union.astype(float), 1.0)
