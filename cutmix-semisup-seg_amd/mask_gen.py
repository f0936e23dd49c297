"""
Mirror of the reference's mask_gen.py (BoxMaskGenerator + AddMaskParamsToBatch), MI355X-first.

The reference draws box parameters with numpy on the CPU (mask_gen.py:70-108), rasterises full-resolution float64
masks there (:110-116), ships N*H*W fp32 to the GPU every iteration (train_seg_semisup_mask_mt.py:331) and declares
`torch_masks_from_params` the identity (:119-120).

Here the random draws stay on the host with the SAME numpy call sequence (so a seeded RandomState gives the same
boxes, bit for bit), but what travels to the device is the (N, n_boxes, 4) int32 table of half-open ranges
[y0, y1, x0, x1]; kernels rasterise it on the fly (csrc/boxmask.hip, csrc/losses.hip). `generate_params` still
returns the reference's float64 (N,1,H,W) array for drop-in use; `generate_ranges` + `torch_masks_from_params`
is the device path.
"""
import numpy as np
import torch

from . import ops


class MaskGenerator(object):
    """Mask Generator (abstract; mask_gen.py:10-23)"""

    def generate_params(self, n_masks, mask_shape, rng=None):
        raise NotImplementedError('Abstract')

    def append_to_batch(self, *batch):
        x = batch[0]
        params = self.generate_params(len(x), x.shape[2:4])
        return batch + (params,)

    def torch_masks_from_params(self, t_params, mask_shape, torch_device):
        raise NotImplementedError('Abstract')


def gaussian_kernels(sigma, max_sigma=None, truncate=4.0):
    """Rows of normalised 1-D Gaussian kernels, one per entry of `sigma` ((N,) array), all of the width the LARGEST sigma asks
    for (radius = int(truncate * max_sigma + 0.5)) -- mask_gen.py:26-43 of the reference, a host-side helper of its mask
    generators that the CutMix trainer itself never calls; kept so that `import mask_gen` offers the reference's names."""
    sigma = np.asarray(sigma, dtype=np.float64)
    if max_sigma is None:
        max_sigma = sigma.max()
    radius = int(truncate * max_sigma + 0.5)
    offsets = np.arange(-radius, radius + 1, dtype=np.float64)[None, :]
    weights = np.exp(offsets * offsets * (-0.5 / (sigma[:, None] * sigma[:, None])))
    return weights / weights.sum(axis=1, keepdims=True)


class BoxMaskGenerator(MaskGenerator):
    def __init__(self, prop_range, n_boxes=1, random_aspect_ratio=True, prop_by_area=True, within_bounds=True,
                 invert=False):
        if isinstance(prop_range, float):
            prop_range = (prop_range, prop_range)
        self.prop_range = prop_range
        self.n_boxes = n_boxes
        self.random_aspect_ratio = random_aspect_ratio
        self.prop_by_area = prop_by_area
        self.within_bounds = within_bounds
        self.invert = invert

    # ------------------------------------------------------------------ host: RNG draws (numpy call order kept)
    def generate_rectangles(self, n_masks, mask_shape, rng=None):
        """float64 (N, n_boxes, 4) rectangles [y0, x0, y1, x1] exactly as mask_gen.py:73-108 computes them."""
        if rng is None:
            rng = np.random
        lo, hi = self.prop_range
        nb = self.n_boxes
        dims = np.array(mask_shape)
        scale = np.sqrt(1.0 / nb)
        if self.prop_by_area:
            props = rng.uniform(lo, hi, size=(n_masks, nb))
            dead = props == 0.0
            if self.random_aspect_ratio:
                with np.errstate(divide='ignore', invalid='ignore'):
                    hy = np.exp(rng.uniform(low=0.0, high=1.0, size=(n_masks, nb)) * np.log(props))
                    wx = props / hy
                hy = hy * scale
                wx = wx * scale
            else:
                # the reference scales one shared array twice (mask_gen.py:84-87)
                hy = wx = np.sqrt(props) * scale * scale
            hy = np.where(dead, 0.0, hy)
            wx = np.where(dead, 0.0, wx)
        else:
            if self.random_aspect_ratio:
                hy = rng.uniform(lo, hi, size=(n_masks, nb)) * scale
                wx = rng.uniform(lo, hi, size=(n_masks, nb)) * scale
            else:
                hy = wx = rng.uniform(lo, hi, size=(n_masks, nb)) * scale * scale   # shared array, scaled twice
        sizes = np.round(np.stack([hy, wx], axis=2) * dims[None, None, :])
        if self.within_bounds:
            origin = np.round((dims - sizes) * rng.uniform(low=0.0, high=1.0, size=sizes.shape))
            return np.concatenate([origin, origin + sizes], axis=2)
        centre = np.round(dims * rng.uniform(low=0.0, high=1.0, size=sizes.shape))
        return np.concatenate([centre - sizes * 0.5, centre + sizes * 0.5], axis=2)

    @staticmethod
    def rectangles_to_ranges(rectangles, mask_shape):
        """numpy basic-slice rules of `m[int(y0):int(y1), int(x0):int(x1)]` -> int32 (N, nb, 4) [y0, y1, x0, x1]."""
        H, W = int(mask_shape[0]), int(mask_shape[1])
        flat = rectangles.reshape(-1, 4)
        out = np.empty((flat.shape[0], 4), dtype=np.int32)
        for i, (y0, x0, y1, x1) in enumerate(flat):
            ys, ye, _ = slice(int(y0), int(y1)).indices(H)
            xs, xe, _ = slice(int(x0), int(x1)).indices(W)
            out[i] = (ys, max(ys, ye), xs, max(xs, xe))
        return out.reshape(rectangles.shape[0], rectangles.shape[1], 4)

    def generate_ranges(self, n_masks, mask_shape, rng=None):
        """Device-path parameters: int32 (N, n_boxes, 4) numpy array."""
        return self.rectangles_to_ranges(self.generate_rectangles(n_masks, mask_shape, rng), mask_shape)

    # ------------------------------------------------------------------ reference-compatible surface
    def generate_params(self, n_masks, mask_shape, rng=None):
        """
        Box masks as a float64 `(N, 1, H, W)` numpy array, same values as the reference for the same `rng`
        (host-side, like the reference, which also builds them in DataLoader workers).
        """
        mask_shape = tuple(int(s) for s in mask_shape)
        ranges = self.generate_ranges(n_masks, mask_shape, rng)
        canvas = np.zeros((n_masks, 1) + mask_shape, dtype=bool)
        for i in range(n_masks):
            for ys, ye, xs, xe in ranges[i]:
                canvas[i, 0, ys:ye, xs:xe] ^= True
        if not self.invert:
            canvas = ~canvas
        return canvas.astype(np.float64)

    def torch_masks_from_params(self, t_params, mask_shape, torch_device):
        """
        Reference behaviour: identity on full masks (mask_gen.py:119-120). Device path: an int32 (N, nb, 4) range
        table is rasterised on the GPU.
        """
        if torch.is_tensor(t_params) and t_params.dtype == torch.int32 and t_params.dim() == 3 \
                and t_params.shape[-1] == 4:
            return ops.boxmask_rasterize(t_params.to(torch_device), mask_shape, self.invert)
        return t_params


class AddMaskParamsToBatch(object):
    """
    Collate hook (mask_gen.py:123-142): adds `mask_params` to every sample of a batch. With `as_ranges=True` the
    per-sample entry is the (n_boxes, 4) int32 range table instead of a full-resolution fp32 mask, which removes the
    N*H*W*4-byte host->device copy per iteration.
    """

    def __init__(self, mask_gen, as_ranges=False):
        self.mask_gen = mask_gen
        self.as_ranges = as_ranges

    def __call__(self, batch):
        sample = batch[0]
        sample0 = sample['sample0'] if 'sample0' in sample else sample
        mask_size = sample0['image'].shape[1:3]
        if self.as_ranges:
            params = self.mask_gen.generate_ranges(len(batch), mask_size)
            for sample, p in zip(batch, params):
                sample['mask_params'] = p
        else:
            params = self.mask_gen.generate_params(len(batch), mask_size)
            for sample, p in zip(batch, params):
                sample['mask_params'] = p.astype(np.float32)
        return batch
