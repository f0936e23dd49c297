"""
Oracle (test infrastructure): box-mask parameter draw + rasterisation.

Restates `BoxMaskGenerator.generate_params` (reference mask_gen.py:57-117) in two separable stages so that the
device path can be checked stage by stage:

  draw_rects()      -- the RNG draws and float64 arithmetic that yield (N, n_boxes, 4) rectangles [y0, x0, y1, x1]
                       (mask_gen.py:70-108)
  rects_to_ranges() -- the `int(y0):int(y1)` numpy slice semantics (truncation toward zero, negative-index
                       wrap, clamping, empty slices) turned into explicit half-open int ranges (mask_gen.py:114-116)
  rasterise()       -- XOR-accumulate the boxes into a zeros (invert) / ones canvas (mask_gen.py:110-116)

Pinned by tests/golden/boxmask_*.npz (generated from the reference module itself).
"""
import numpy as np


def _as_range(prop_range):
    # mask_gen.py:48-49 -- a float means a degenerate range
    if isinstance(prop_range, float):
        return (prop_range, prop_range)
    return tuple(prop_range)


def draw_rects(n_masks, mask_shape, prop_range, n_boxes=1, random_aspect_ratio=True, prop_by_area=True,
               within_bounds=True, rng=None):
    """Rectangles as float64 (N, n_boxes, 4) = [y0, x0, y1, x1]; RNG draw order as mask_gen.py:73-108."""
    if rng is None:
        rng = np.random
    lo, hi = _as_range(prop_range)
    shp = (n_masks, n_boxes)
    hw = np.array(mask_shape)
    fac = np.sqrt(1.0 / n_boxes)

    if prop_by_area:
        area = rng.uniform(lo, hi, size=shp)                       # mask_gen.py:75
        was_zero = area == 0.0                                     # :78
        if random_aspect_ratio:
            u = rng.uniform(low=0.0, high=1.0, size=shp)           # :81
            py = np.exp(u * np.log(area))
            px = area / py
            py = py * fac                                          # :86-87 (two distinct arrays)
            px = px * fac
        else:
            # :84 binds ONE array to both names, so the two in-place `*= fac` at :86-87 hit it twice.
            root = np.sqrt(area)
            py = px = root * fac * fac
        py = np.where(was_zero, 0.0, py)                           # :89-90
        px = np.where(was_zero, 0.0, px)
    else:
        if random_aspect_ratio:
            py = rng.uniform(lo, hi, size=shp) * fac               # :93-94, :98-99
            px = rng.uniform(lo, hi, size=shp) * fac
        else:
            # :96 aliases x_props and y_props -> scaled by fac twice (:98-99)
            py = px = rng.uniform(lo, hi, size=shp) * fac * fac

    sizes = np.round(np.stack([py, px], axis=2) * hw[None, None, :])   # :101 (half-to-even)

    if within_bounds:
        pos = np.round((hw - sizes) * rng.uniform(low=0.0, high=1.0, size=sizes.shape))   # :104
        rects = np.concatenate([pos, pos + sizes], axis=2)                                  # :105
    else:
        ctr = np.round(hw * rng.uniform(low=0.0, high=1.0, size=sizes.shape))             # :107
        rects = np.concatenate([ctr - sizes * 0.5, ctr + sizes * 0.5], axis=2)             # :108
    return rects


def rects_to_ranges(rects, mask_shape):
    """
    (N, nb, 4) float rects -> (N, nb, 4) int32 [y0, y1, x0, x1] half-open ranges with numpy basic-slice
    semantics for `a[int(y0):int(y1), int(x0):int(x1)]` (mask_gen.py:116). Empty slices come out as y0 == y1.
    """
    H, W = int(mask_shape[0]), int(mask_shape[1])
    out = np.zeros(rects.shape[:2] + (4,), dtype=np.int32)
    for i in range(rects.shape[0]):
        for b in range(rects.shape[1]):
            y0, x0, y1, x1 = rects[i, b]
            ys, ye, _ = slice(int(y0), int(y1)).indices(H)
            xs, xe, _ = slice(int(x0), int(x1)).indices(W)
            if ye < ys:
                ye = ys
            if xe < xs:
                xe = xs
            out[i, b] = (ys, ye, xs, xe)
    return out


def rasterise(ranges, mask_shape, invert):
    """(N, nb, 4) int ranges -> float64 (N, 1, H, W) masks; boxes XOR into the canvas (mask_gen.py:110-116)."""
    H, W = int(mask_shape[0]), int(mask_shape[1])
    n = ranges.shape[0]
    parity = np.zeros((n, 1, H, W), dtype=bool)
    for i in range(n):
        for ys, ye, xs, xe in ranges[i]:
            parity[i, 0, ys:ye, xs:xe] ^= True
    # invert=True starts from zeros so box pixels become 1; otherwise starts from ones.
    return parity.astype(np.float64) if invert else 1.0 - parity.astype(np.float64)


def generate_params(n_masks, mask_shape, prop_range, n_boxes=1, random_aspect_ratio=True, prop_by_area=True,
                    within_bounds=True, invert=False, rng=None):
    """Whole of mask_gen.py:57-117 -> float64 (N,1,H,W)."""
    rects = draw_rects(n_masks, mask_shape, prop_range, n_boxes, random_aspect_ratio, prop_by_area,
                       within_bounds, rng)
    return rasterise(rects_to_ranges(rects, mask_shape), mask_shape, invert)
