"""
Oracle (test infrastructure): the reference's per-sample training transforms restated in numpy float64 for a GIVEN
parameter row (crop window, flips, colour factors): datapipe/seg_transforms_cv.py:29-133 (pad + crop), :169-231
(random-scale crop + cv2.resize), :452-497 (flips), :541-585 (torchvision ColorJitter / RandomGrayscale through PIL),
:587-623 (standardise, NCHW).

PARITY UNPINNED: cv2, PIL and torchvision are third-party dependencies absent from /root/reference and from this image.
Restated from their published definitions: cv2.resize(INTER_LINEAR) = bilinear with half-pixel centres and replicated
border, (INTER_NEAREST) = floor(dst * scale); ColorJitter: brightness x*f, contrast (x - mean_grey)*f + mean_grey,
saturation blend with the ITU-R 601 grey (0.299, 0.587, 0.114), hue shift in HSV; all clamped to [0, 1]. The uint8
re-quantisation cv2 / PIL perform between the steps is left out on both sides (device and oracle).
"""
import colorsys

import numpy as np

GREY = np.array([0.299, 0.587, 0.114])


def _bilinear_window(img, y0, x0, sh, sw, H, W):
    """img (Hs, Ws, K) float with zeros outside; window origin (y0, x0) size (sh, sw) -> (H, W, K), + inside weight."""
    Hs, Ws = img.shape[:2]
    oy, ox = np.arange(H), np.arange(W)
    fy = np.clip((oy + 0.5) * (sh / H) - 0.5, 0.0, sh - 1.0)
    fx = np.clip((ox + 0.5) * (sw / W) - 0.5, 0.0, sw - 1.0)
    iy0, ix0 = np.floor(fy).astype(int), np.floor(fx).astype(int)
    wy, wx = fy - iy0, fx - ix0
    iy1, ix1 = np.minimum(iy0 + 1, sh - 1), np.minimum(ix0 + 1, sw - 1)

    def take(iy, ix):
        Y, X = iy[:, None] + y0, ix[None, :] + x0
        ok = (Y >= 0) & (Y < Hs) & (X >= 0) & (X < Ws)
        v = img[np.clip(Y, 0, Hs - 1), np.clip(X, 0, Ws - 1)] * ok[..., None]
        return v, ok.astype(np.float64)
    out = np.zeros((H, W, img.shape[2]))
    alpha = np.zeros((H, W))
    for iy, wy_ in ((iy0, 1.0 - wy), (iy1, wy)):
        for ix, wx_ in ((ix0, 1.0 - wx), (ix1, wx)):
            v, ok = take(iy, ix)
            wgt = wy_[:, None] * wx_[None, :]
            out += v * wgt[..., None]
            alpha += ok * wgt
    return out, alpha


def _flip(a, fx, fy, fd):
    if fx:
        a = a[:, ::-1]
    if fy:
        a = a[::-1]
    if fd:
        a = np.swapaxes(a, 0, 1)
    return a


def _hue(rgb, dh):
    out = np.empty_like(rgb)
    flat, o = rgb.reshape(-1, 3), out.reshape(-1, 3)
    for i, (r, g, b) in enumerate(flat):
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        o[i] = colorsys.hsv_to_rgb((h + dh) % 1.0, s, v)
    return out


def _reflect101(i, n):
    """cv2.BORDER_REFLECT_101 index map (period 2n - 2)."""
    if n == 1:
        return np.zeros_like(i)
    period = 2 * n - 2
    i = np.mod(i, period)
    return np.where(i < n, i, period - i)


def _warp(src, labels_u8, p, H, W):
    """cv2.warpAffine of datapipe/seg_transforms_cv.py:358-366 for ONE sample from the INVERSE matrix in p[16:22]:
    output pixel (x, y) samples the source at (a00 x + a01 y + a02, a10 x + a11 y + a12). Image: nearest (floor(s + 0.5))
    or bilinear (p[22]), BORDER_REFLECT_101; labels: nearest, constant 255; mask: in-bounds weight (constant 0).
    Unpinned (cv2 absent): OpenCV evaluates the coordinates in 1/1024 fixed point and the bilinear weights in 1/32 steps;
    this is the exact-arithmetic definition."""
    Hs, Ws = src.shape[:2]
    a = np.asarray(p[16:22], dtype=np.float64)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    sx = a[0] * xx + a[1] * yy + a[2]
    sy = a[3] * xx + a[4] * yy + a[5]
    nx, ny = np.floor(sx + 0.5).astype(np.int64), np.floor(sy + 0.5).astype(np.int64)
    inside = lambda Y, X: ((Y >= 0) & (Y < Hs) & (X >= 0) & (X < Ws))
    if int(p[22]) == 0:
        rgb = src[_reflect101(ny, Hs), _reflect101(nx, Ws)]
        alpha = inside(ny, nx).astype(np.float64)
    else:
        ix0, iy0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
        wx, wy = sx - ix0, sy - iy0
        rgb = np.zeros((H, W, src.shape[2]))
        alpha = np.zeros((H, W))
        for Y, wy_ in ((iy0, 1.0 - wy), (iy0 + 1, wy)):
            for X, wx_ in ((ix0, 1.0 - wx), (ix0 + 1, wx)):
                wgt = wy_ * wx_
                rgb += src[_reflect101(Y, Hs), _reflect101(X, Ws)] * wgt[..., None]
                alpha += inside(Y, X) * wgt
    lab = None
    if labels_u8 is not None:
        ok = inside(ny, nx)
        lab = np.full((H, W), 255, dtype=np.uint8)
        lab[ok] = labels_u8[ny[ok], nx[ok]]
    return rgb, alpha, lab


def augment_sample(src_u8, labels_u8, p, crop_hw, mean, std, pivot=None):
    """One sample. src_u8 (Hs, Ws, 3) uint8, labels_u8 (Hs, Ws) uint8 or None, p = one row of the parameter table.
    -> (image (3,H,W), image_colour (3,H,W), labels (H,W) uint8 | None, mask (H,W))."""
    H, W = crop_hw
    y0, x0, sh, sw = int(p[0]), int(p[1]), int(p[2]), int(p[3])
    fx, fy, fd = bool(p[4]), bool(p[5]), bool(p[6])
    warp = len(p) > 15 and p[15] != 0
    warp_lab = None
    if warp:
        rgb, alpha, warp_lab = _warp(src_u8.astype(np.float64) / 255.0, labels_u8, p, H, W)
    else:
        rgb, alpha = _bilinear_window(src_u8.astype(np.float64) / 255.0, y0, x0, sh, sw, H, W)
    rgb, alpha = _flip(rgb, fx, fy, fd), _flip(alpha, fx, fy, fd)
    mean, std = np.asarray(mean, dtype=np.float64), np.asarray(std, dtype=np.float64)
    # (window crops zero-pad through an alpha channel, :46-52, 600-608; the warp reflects the image and only the MASK knows
    # what lies outside)
    img_alpha = np.ones_like(alpha) if warp else alpha
    img0 = (rgb - mean * img_alpha[..., None]) / std
    col = rgb.copy()
    if p[12]:
        order = int(p[13])
        for op in [(order >> s) & 3 for s in (6, 4, 2, 0)]:
            if op == 0:
                col = np.clip(col * p[7], 0.0, 1.0)
            elif op == 1:
                m = float(pivot) if pivot is not None else float((col @ GREY).mean())
                col = np.clip((col - m) * p[8] + m, 0.0, 1.0)
            elif op == 2:
                g = (col @ GREY)[..., None]
                col = np.clip((col - g) * p[9] + g, 0.0, 1.0)
            elif p[10] != 0.0:
                col = _hue(col, float(p[10]))
    if p[11]:
        col = np.repeat((col @ GREY)[..., None], 3, axis=2)
    img1 = (col - mean * img_alpha[..., None]) / std
    lab = None
    if warp_lab is not None:
        lab = np.ascontiguousarray(_flip(warp_lab, fx, fy, fd))
    elif labels_u8 is not None:
        Hs, Ws = labels_u8.shape
        # crop + nearest resize first (cv2.INTER_NEAREST: floor(dst * scale)), then the flips, as the reference does
        cy, cx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
        ny = np.minimum((cy.astype(np.float32) * (np.float32(sh) / np.float32(H))).astype(int), sh - 1) + y0
        nx = np.minimum((cx.astype(np.float32) * (np.float32(sw) / np.float32(W))).astype(int), sw - 1) + x0
        ok = (ny >= 0) & (ny < Hs) & (nx >= 0) & (nx < Ws)
        crop = np.full((H, W), 255, dtype=np.uint8)
        crop[ok] = labels_u8[ny[ok], nx[ok]]
        lab = np.ascontiguousarray(_flip(crop, fx, fy, fd))
    return img0.transpose(2, 0, 1), img1.transpose(2, 0, 1), lab, alpha
