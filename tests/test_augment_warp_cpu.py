"""
CPU: the random rotate + scale crop of the device-side staging (datapipe/seg_transforms_cv.py:306-372, selected by
--aug_rot_mag / --aug_max_scale at train_seg_semisup_mask_mt.py:153-155) -- its HOST half against vectors made by the
reference's own datapipe/affine.py (tests/golden/affine_rotate_scale.json, generator tests/golden/make_golden.py::gen_affine):
the float32 matrix composition bit for bit, and the parameter draws in the reference's order. The warp arithmetic itself
(cv2.warpAffine; cv2 is absent: unpinned) is restated in oracle/augment.py and checked here on transforms whose result is
known in closed form.
"""
import json
import os

import numpy as np

from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, 'affine_rotate_scale.json')))


def test_local_xf_equals_the_references_affine_composition_bit_for_bit():
    from cutmix_semisup_seg_amd.device_pipeline import DeviceAugmenter
    for c in CASES:
        xf = DeviceAugmenter.local_xf(c['crop'], c['theta'], c['sf'], c['centre'])
        assert xf.dtype == np.float32 and c['xf_dtype'] == 'float32'
        assert np.array_equal(xf, np.array(c['xf'], dtype=np.float32)), c


def test_parameter_draws_follow_the_references_order():
    from cutmix_semisup_seg_amd.device_pipeline import DeviceAugmenter
    by_seed = {}
    for c in CASES:
        by_seed.setdefault(c['seed'], []).append(c)
    for seed, cs in by_seed.items():
        c0 = cs[0]
        aug = DeviceAugmenter(c0['crop'], None, None, scale_non_uniform=not c0['uniform'], rot_mag=c0['rot_mag'],
                              max_scale=c0['max_scale'], rng=np.random.RandomState(seed))
        assert aug.warp
        p = aug.draw_params(len(cs), c0['img'], with_labels=False)        # unlabelled: the interpolation is drawn (:356)
        for row, c in zip(p, cs):
            m = np.array(c['xf'], dtype=np.float64)
            inv2 = np.linalg.inv(m[:, :2])
            invt = -inv2 @ m[:, 2]
            want = np.array([inv2[0, 0], inv2[0, 1], invt[0], inv2[1, 0], inv2[1, 1], invt[1]], dtype=np.float32)
            np.testing.assert_allclose(row[16:22], want, rtol=1e-6, atol=1e-6)
            assert row[15] == 1 and int(row[22]) == c['interp'] and tuple(row[2:4]) == tuple(c['crop'])
    # labelled samples: nearest, and NO draw -- the stream after it is shifted accordingly
    c0 = by_seed[1][0]
    a = DeviceAugmenter(c0['crop'], None, None, rot_mag=c0['rot_mag'], max_scale=c0['max_scale'], rng=np.random.RandomState(1))
    b = DeviceAugmenter(c0['crop'], None, None, rot_mag=c0['rot_mag'], max_scale=c0['max_scale'], rng=np.random.RandomState(1))
    pa, pb = a.draw_params(2, c0['img'], with_labels=True), b.draw_params(2, c0['img'], with_labels=False)
    assert (pa[:, 22] == 0).all() and np.allclose(pa[0, 16:22], pb[0, 16:22]) and not np.allclose(pa[1, 16:22], pb[1, 16:22])
    # Hung's scale crop wins when both are asked for (train_seg_semisup_mask_mt.py:150-155)
    assert not DeviceAugmenter((33, 33), None, None, scale_hung=True, rot_mag=10.0, max_scale=2.0).warp


def test_oracle_warp_on_transforms_with_known_results():
    from oracle import augment as oaug
    rng = np.random.RandomState(0)
    src = rng.randint(0, 256, size=(20, 30, 3)).astype(np.uint8)
    lab = rng.randint(0, 7, size=(20, 30)).astype(np.uint8)
    H, W = 20, 30

    def row(a, interp):
        p = np.zeros(24)
        p[2:4], p[7:10], p[15], p[16:22], p[22] = (H, W), 1.0, 1.0, a, interp
        return p
    ident = [1, 0, 0, 0, 1, 0]
    for interp in (0, 1):
        i0, _, lb, al = oaug.augment_sample(src, lab, row(ident, interp), (H, W), np.zeros(3), np.ones(3))
        np.testing.assert_allclose(i0.transpose(1, 2, 0), src / 255.0, atol=1e-12)
        assert np.array_equal(lb, lab) and (al == 1).all()
    # integer translation by (+3, -2): image reflected (101) outside, labels 255, mask 0
    i0, _, lb, al = oaug.augment_sample(src, lab, row([1, 0, 3, 0, 1, -2], 1), (H, W), np.zeros(3), np.ones(3))
    img = i0.transpose(1, 2, 0) * 255.0
    np.testing.assert_allclose(img[2:, :W - 3], src[:H - 2, 3:], atol=1e-9)
    np.testing.assert_allclose(img[0], np.concatenate([src[2, 3:], src[2, W - 2:W - 5:-1]]), atol=1e-9)    # rows -2 -> 2
    assert (lb[:2] == 255).all() and (lb[:, W - 3:] == 255).all() and np.array_equal(lb[2:, :W - 3], lab[:H - 2, 3:])
    assert (al[:2] == 0).all() and (al[2:, :W - 3] == 1).all()
    # half-pixel shift, bilinear: the average of two neighbours; mask = in-bounds weight
    i0, _, _, al = oaug.augment_sample(src, None, row([1, 0, 0.5, 0, 1, 0], 1), (H, W), np.zeros(3), np.ones(3))
    np.testing.assert_allclose(i0.transpose(1, 2, 0)[:, :W - 1] * 255.0, 0.5 * (src[:, :-1].astype(float) + src[:, 1:]), atol=1e-9)
    assert np.allclose(al[:, W - 1], 0.5) and np.allclose(al[:, :W - 1], 1.0)
    assert list(oaug._reflect101(np.array([-3, -1, 0, 4, 5, 6, 9]), 5)) == [3, 1, 0, 4, 3, 2, 1]
