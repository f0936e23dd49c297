"""
Device-side input staging (second half): the per-sample training transforms of the reference's
loader workers -- random crop, Hung-style random-scale crop, flips, torchvision ColorJitter / RandomGrayscale on the
student view, standardisation, NCHW (datapipe/seg_transforms_cv.py:29-133, 169-231, 452-497, 541-623; assembled at
train_seg_semisup_mask_mt.py:150-183) -- applied on the GPU to uint8 source images that already sit in HBM
(csrc/augment.hip: one gather kernel per batch, both views of the paired layout in one pass). Only the random parameters
are drawn on the host, per sample and in the reference's order:

    crop    f_scale = 0.5 + rng.randint(0, 11, size=(1 | 2,)) / 10          (:193, Hung scale; skipped without it)
            sc_size = round(crop_size / f_scale)                            (:196)
            pad to sc_size if the image is smaller: h0 = pad // 2 on top    (:36-43)
            pos = round((padded_size - sc_size) * rng.uniform(0, 1, 2))     (:203-204 / :122-123)
    warp    (rot_mag / max_scale given: SegCVTransformRandomCropRotateScale.transform_single, :331-362, selected at
            train_seg_semisup_mask_mt.py:153-155)
            scale = exp(rng.uniform(-log max_scale, log max_scale, (1 | 2,)));  theta = rng.uniform(-rot, rot, (1,))
            centre = max(img - crop / scale, 0) * rng.uniform(0, 1, 2) + min(crop / scale, img) / 2
            local_xf = T(crop / 2) . R(theta) . S(scale) . T(-centre)       (datapipe/affine.py, float32)
            interpolation = NEAREST with labels, else rng.choice([NEAREST, LINEAR])
            -> the kernel samples the source at local_xf^-1 (x, y): image REFLECT_101, labels nearest / 255, mask 0 outside
    flips   rng.binomial(1, 0.5, size=(3,)) & [hflip, vflip, hvflip]        (:479-480)
    colour  (student view of a pair only, :575-583) RandomApply(ColorJitter, p), RandomGrayscale(p)   [torchvision]

What is NOT reproduced bit for bit, and cannot be pinned here (cv2 / PIL / torchvision are absent): cv2.resize's 11-bit
fixed-point interpolation and its rounding to uint8 before the colour operations, PIL's rounding after every jitter
operation, and torchvision's use of Python's global `random` for the colour draws (a numpy RandomState here). The
arithmetic is the published definition of each operation in floating point (oracle/augment.py is the numpy restatement
the tests compare against); PARITY UNPINNED for this stage.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import fn, check


class DeviceAugmenter(object):
    def __init__(self, crop_size, mean, std, scale_hung=False, scale_non_uniform=False, hflip=False, vflip=False,
                 hvflip=False, strong_colour=False, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.1, colour_prob=0.8,
                 greyscale_prob=0.2, out_dtype=torch.bfloat16, rng=None, colour_rng=None, rot_mag=0.0, max_scale=1.0):
        self.crop_size = (int(crop_size[0]), int(crop_size[1]))
        self.mean = np.zeros(3) if mean is None else np.asarray(mean, dtype=np.float64)
        self.std = np.ones(3) if std is None else np.asarray(std, dtype=np.float64)
        self.scale_hung, self.uniform_scale = bool(scale_hung), not scale_non_uniform
        self.flips = np.array([hflip, vflip, hvflip], dtype=bool)
        if hvflip and self.crop_size[0] != self.crop_size[1]:
            raise ValueError('aug_hvflip (transpose) needs a square crop')
        self.strong_colour = bool(strong_colour)
        self.jitter = (float(brightness), float(contrast), float(saturation), float(hue))
        self.colour_prob, self.greyscale_prob = float(colour_prob), float(greyscale_prob)
        self.out_dtype = out_dtype
        self._rng, self._crng = rng, colour_rng
        # random rotate + scale crop (--aug_rot_mag / --aug_max_scale); the reference picks Hung's scale crop first when
        # both are given (train_seg_semisup_mask_mt.py:150-155)
        self.rot_mag_rad = float(np.radians(rot_mag))
        self.log_max_scale = float(np.log(max_scale))
        self.warp = (not self.scale_hung) and (max_scale != 1.0 or rot_mag != 0.0)

    @property
    def rng(self):
        if self._rng is None:
            self._rng = np.random.RandomState()
        return self._rng

    @property
    def colour_rng(self):
        if self._crng is None:
            self._crng = np.random.RandomState()
        return self._crng

    @staticmethod
    def local_xf(crop_hw, theta, scale_yx, centre_yx):
        """The reference's float32 2x3 matrix  T(crop / 2) . R(theta) . S(scale) . T(-centre)  (seg_transforms_cv.py:347-352
        with datapipe/affine.py:60-118: every factor a float32 matrix, products in float32, x before y) -- pinned bit for bit
        by tests/golden/affine_rotate_scale.json."""
        f32 = np.float32

        def cat(a, b):
            a2, b2, ax, bx = a[:, :2], b[:, :2], a[:, 2:3], b[:, 2:3]
            return np.append(np.matmul(a2, b2), ax + np.matmul(a2, bx), axis=1)

        def T(xy):
            m = np.zeros((2, 3), dtype=f32)
            m[0, 0] = m[1, 1] = 1.0
            m[:, 2] = xy
            return m
        R = np.zeros((2, 3), dtype=f32)
        R[0, 0] = R[1, 1] = np.cos(theta)
        R[1, 0], R[0, 1] = -np.sin(theta), np.sin(theta)
        S = np.zeros((2, 3), dtype=f32)
        S[0, 0], S[1, 1] = scale_yx[1], scale_yx[0]
        crop = np.asarray(crop_hw, dtype=np.float64)
        centre = np.asarray(centre_yx, dtype=np.float64)
        return cat(cat(cat(T(crop[::-1] * 0.5), R), S), T(-centre[::-1]))

    def draw_params(self, n, src_hw, with_labels=False):
        """-> float32 (n, CMS_AUG_PARAMS) parameter table of cms_augment_desc (slot 14, the contrast pivot, is filled on the
        device). `with_labels`: the samples carry label maps (the supervised stream) -- decides the interpolation draw of the
        rotate / scale crop exactly as the reference does (:353-356)."""
        hs, ws = int(src_hw[0]), int(src_hw[1])
        crop = np.array(self.crop_size)
        out = np.zeros((n, _lib.AUG_PARAMS), dtype=np.float32)
        for i in range(n):
            if self.warp:
                if self.uniform_scale:
                    sf = np.exp(self.rng.uniform(-self.log_max_scale, self.log_max_scale, size=(1,)))
                    sf = np.repeat(sf, 2, axis=0)
                else:
                    sf = np.exp(self.rng.uniform(-self.log_max_scale, self.log_max_scale, size=(2,)))
                theta = self.rng.uniform(-self.rot_mag_rad, self.rot_mag_rad, size=(1,))
                sc_size = crop / sf
                img = np.array([hs, ws])
                extra = np.maximum(img - sc_size, 0.0)
                centre = extra * self.rng.uniform(0.0, 1.0, size=(2,)) + np.minimum(sc_size, img) * 0.5
                xf = self.local_xf(crop, theta[0], sf, centre)
                interp = 0 if with_labels else int(self.rng.choice([0, 1]))        # cv2.INTER_NEAREST = 0, INTER_LINEAR = 1
                m = xf.astype(np.float64)                                           # (cv2 inverts in double precision)
                det = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
                inv2 = np.array([[m[1, 1], -m[0, 1]], [-m[1, 0], m[0, 0]]]) / det
                invt = -inv2 @ m[:, 2]
                out[i, 2:4] = crop
                out[i, 15] = 1.0
                out[i, 16:19] = (inv2[0, 0], inv2[0, 1], invt[0])
                out[i, 19:22] = (inv2[1, 0], inv2[1, 1], invt[1])
                out[i, 22] = interp
                self._finish_row(out, i)
                continue
            if self.scale_hung:
                f_scale = 0.5 + self.rng.randint(0, 11, size=(1 if self.uniform_scale else 2,)) / 10.0
                sc = np.round(crop / f_scale).astype(int)
            else:
                sc = crop.copy()
            img = np.array([hs, ws])
            pad = np.maximum(sc - img, 0)
            lead = pad // 2
            extra = img + pad - sc
            pos = np.round(extra * self.rng.uniform(0.0, 1.0, size=(2,))).astype(int)
            out[i, 0:2] = pos - lead
            out[i, 2:4] = sc
            self._finish_row(out, i)
        return out

    def _finish_row(self, out, i):
        """Flip and colour draws of sample i (after its geometry), in the reference's order."""
        if self.flips.any():
            f = (self.rng.binomial(1, 0.5, size=(3,)) != 0) & self.flips
            out[i, 4:7] = f
        out[i, 7:10] = 1.0
        if self.strong_colour:
            cr = self.colour_rng
            b, c, s, h = self.jitter
            apply = cr.uniform(0.0, 1.0) < self.colour_prob
            fb = cr.uniform(max(0.0, 1.0 - b), 1.0 + b)
            fc = cr.uniform(max(0.0, 1.0 - c), 1.0 + c)
            fs = cr.uniform(max(0.0, 1.0 - s), 1.0 + s)
            fh = cr.uniform(-h, h)
            order = cr.permutation(4)
            grey = cr.uniform(0.0, 1.0) < self.greyscale_prob
            out[i, 7:11] = (fb, fc, fs, fh)
            out[i, 11], out[i, 12] = grey, apply
            out[i, 13] = (int(order[0]) << 6) | (int(order[1]) << 4) | (int(order[2]) << 2) | int(order[3])

    def __call__(self, src_u8, labels_u8=None, params=None):
        """src_u8: CUDA uint8 (N, Hs, Ws, 3); labels_u8: CUDA uint8 (N, Hs, Ws) or None.
        -> dict(image [teacher / only view], image_stu (with strong colour), labels (N,1,h,w) uint8, mask (N,1,h,w) fp32)."""
        if not src_u8.is_cuda or src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3:
            raise RuntimeError('DeviceAugmenter: CUDA uint8 (N, Hs, Ws, 3) source images required (no CPU path)')
        src_u8 = src_u8.contiguous()
        n, hs, ws, _ = (int(v) for v in src_u8.shape)
        h, w = self.crop_size
        if params is None:
            params = self.draw_params(n, (hs, ws), with_labels=labels_u8 is not None)
        dev = src_u8.device
        p_dev = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float32)).to(dev, non_blocking=True)
        out0 = torch.empty((n, 3, h, w), dtype=self.out_dtype, device=dev)
        out1 = torch.empty_like(out0) if self.strong_colour else None
        mask = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
        labs = None
        if labels_u8 is not None:
            labels_u8 = labels_u8.contiguous()
            if labels_u8.dtype != torch.uint8 or tuple(labels_u8.shape) != (n, hs, ws):
                raise ValueError('DeviceAugmenter: labels must be uint8 (N, Hs, Ws)')
            labs = torch.empty((n, 1, h, w), dtype=torch.uint8, device=dev)
        d = _lib.AugmentDesc()
        d.src, d.src_labels = src_u8.data_ptr(), (labels_u8.data_ptr() if labels_u8 is not None else None)
        d.out0, d.out1 = out0.data_ptr(), (out1.data_ptr() if out1 is not None else None)
        d.out_labels = labs.data_ptr() if labs is not None else None
        d.out_mask, d.params = mask.data_ptr(), p_dev.data_ptr()
        for i in range(3):
            d.mean[i], d.std_[i] = float(self.mean[i]), float(self.std[i])
        d.n, d.hs, d.ws, d.h, d.w = n, hs, ws, h, w
        d.out_dtype = _lib.F32 if self.out_dtype == torch.float32 else _lib.BF16
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if self.strong_colour:
            # contrast pivot: mean luminance of the transformed image, times the brightness factor where brightness is
            # applied before contrast (ColorJitter applies its four operations in the drawn order)
            luma = torch.empty(n, dtype=torch.float32, device=dev)
            check(fn['cms_augment_luma'](C.byref(d), C.c_void_p(luma.data_ptr()), stream), 'cms_augment_luma')
            order = params[:, 13].astype(np.int64)
            pos_b = np.array([[(o >> s) & 3 for s in (6, 4, 2, 0)].index(0) for o in order])
            pos_c = np.array([[(o >> s) & 3 for s in (6, 4, 2, 0)].index(1) for o in order])
            scale = np.where(pos_b < pos_c, params[:, 7], 1.0).astype(np.float32)
            p_dev[:, 14] = luma * torch.from_numpy(scale).to(dev)
        check(fn['cms_augment_batch'](C.byref(d), stream), 'cms_augment_batch')
        res = dict(image=out0, mask=mask)
        if out1 is not None:
            res['image_stu'] = out1
        if labs is not None:
            res['labels'] = labs
        return res
