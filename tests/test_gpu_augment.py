"""
GPU: device-side input staging (csrc/augment.hip, device_pipeline.py) against the numpy restatement of the reference's
per-sample transforms (oracle/augment.py; PARITY UNPINNED -- cv2 / PIL / torchvision absent, see its header): identity
parameters are an exact standardisation + NCHW transpose; crops with random scale, padding, flips and colour
augmentation agree with the oracle for the same parameter rows; both views of the paired layout share their geometry.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
MEAN, STD = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])


def test_identity_parameters_are_standardisation_and_transpose():
    from cutmix_semisup_seg_amd.device_pipeline import DeviceAugmenter
    g = torch.Generator().manual_seed(0)
    src = torch.randint(0, 256, (3, 40, 56, 3), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, 21, (3, 40, 56), generator=g).to(torch.uint8)
    aug = DeviceAugmenter((40, 56), MEAN, STD, out_dtype=torch.float32, rng=np.random.RandomState(0))
    out = aug(src.to(DEV), lab.to(DEV))
    want = (src.double() / 255.0 - torch.tensor(MEAN)) / torch.tensor(STD)
    torch.testing.assert_close(out['image'].cpu().double(), want.permute(0, 3, 1, 2), rtol=1e-5, atol=1e-5)
    assert torch.equal(out['labels'].cpu()[:, 0], lab) and float(out['mask'].min()) == 1.0
    assert 'image_stu' not in out


@pytest.mark.parametrize('cfg', [dict(scale_hung=True, hflip=True, vflip=True), dict(scale_hung=True, scale_non_uniform=True),
                                 dict(hflip=True, vflip=True, hvflip=True, square=True), dict(strong_colour=True, scale_hung=True, hflip=True)],
                         ids=['hung_flips', 'hung_nonuniform', 'all_flips_square', 'colour'])
def test_device_staging_vs_numpy_oracle(cfg):
    from cutmix_semisup_seg_amd.device_pipeline import DeviceAugmenter
    from oracle import augment as oaug
    cfg = dict(cfg)
    crop = (48, 48) if cfg.pop('square', False) else (48, 64)
    g = torch.Generator().manual_seed(3)
    N, Hs, Ws = 6, 60, 70                        # Hung scale 0.5 asks for a 96 x 128 window: padding on both axes
    src = torch.randint(0, 256, (N, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, 5, (N, Hs, Ws), generator=g).to(torch.uint8)
    aug = DeviceAugmenter(crop, MEAN, STD, out_dtype=torch.float32, rng=np.random.RandomState(11),
                          colour_rng=np.random.RandomState(12), **cfg)
    params = aug.draw_params(N, (Hs, Ws))
    out = aug(src.to(DEV), lab.to(DEV), params=params)
    if cfg.get('strong_colour'):
        assert 'image_stu' in out
    for i in range(N):
        pivot = None
        if cfg.get('strong_colour') and params[i, 12]:
            geo_only = np.ones(params.shape[1])
            geo_only[10:15] = 0                                   # no hue / greyscale / jitter: the geometric transform alone
            img0, _, _, _ = oaug.augment_sample(src[i].numpy(), None, params[i] * geo_only, crop, np.zeros(3), np.ones(3))
            luma = float((img0.transpose(1, 2, 0) @ oaug.GREY).mean())
            order = int(params[i, 13])
            ops_ = [(order >> s) & 3 for s in (6, 4, 2, 0)]
            pivot = luma * (params[i, 7] if ops_.index(0) < ops_.index(1) else 1.0)
        i0, i1, lb, al = oaug.augment_sample(src[i].numpy(), lab[i].numpy(), params[i], crop, MEAN, STD, pivot=pivot)
        torch.testing.assert_close(out['image'][i].cpu().double(), torch.from_numpy(i0), rtol=2e-4, atol=2e-4)
        torch.testing.assert_close(out['mask'][i, 0].cpu().double(), torch.from_numpy(np.ascontiguousarray(al)), rtol=1e-5, atol=1e-5)
        assert np.array_equal(out['labels'][i, 0].cpu().numpy(), lb)
        if cfg.get('strong_colour'):
            # (round 4: the luminance pre-pass uses the image kernel's own taps and weights, so the contrast pivot is the mean
            # of the very pixels that are jittered: fp32 reduction noise only; it used a nearest tap and a 2e-2 bound)
            torch.testing.assert_close(out['image_stu'][i].cpu().double(), torch.from_numpy(i1), rtol=2e-3, atol=2e-3)
    if cfg.get('strong_colour'):
        same = [i for i in range(N) if not params[i, 12] and not params[i, 11]]
        for i in same:                                           # jitter not drawn: the student view IS the teacher view
            assert torch.equal(out['image'][i], out['image_stu'][i])
        assert any(params[:, 12] != 0)


@pytest.mark.parametrize('cfg', [dict(rot_mag=30.0, max_scale=1.5), dict(rot_mag=10.0, max_scale=2.0, scale_non_uniform=True, hflip=True),
                                 dict(rot_mag=45.0, max_scale=1.25, strong_colour=True, vflip=True), dict(rot_mag=0.0, max_scale=1.5)],
                         ids=['rot30_scale1.5', 'nonuniform_hflip', 'colour_vflip', 'scale_only'])
@pytest.mark.parametrize('with_labels', [True, False], ids=['sup', 'unsup'])
def test_rotate_scale_crop_vs_numpy_oracle(cfg, with_labels):
    """SegCVTransformRandomCropRotateScale.transform_single (datapipe/seg_transforms_cv.py:331-362) on the device: image
    warped with the reflected border (nearest for labelled samples, the drawn interpolation otherwise), labels nearest with
    255 outside, validity mask zero outside -- against the exact-arithmetic numpy restatement for the same parameter rows.
    The kernel evaluates the source coordinates in fp32: a pixel whose coordinate lands within 1e-4 of a rounding boundary
    may pick the neighbour (cv2's own 1/1024 fixed point moves far more of them) -- bounded, not ignored."""
    from cutmix_semisup_seg_amd.device_pipeline import DeviceAugmenter
    from oracle import augment as oaug
    crop = (48, 64)
    g = torch.Generator().manual_seed(5)
    N, Hs, Ws = 6, 60, 90                          # smaller than some scaled crops: the reflected border is exercised
    src = torch.randint(0, 256, (N, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    lab = torch.randint(0, 5, (N, Hs, Ws), generator=g).to(torch.uint8) if with_labels else None
    aug = DeviceAugmenter(crop, MEAN, STD, out_dtype=torch.float32, rng=np.random.RandomState(21),
                          colour_rng=np.random.RandomState(22), **cfg)
    assert aug.warp
    params = aug.draw_params(N, (Hs, Ws), with_labels=with_labels)
    assert (params[:, 15] == 1).all()
    if with_labels:
        assert (params[:, 22] == 0).all()                          # labelled samples: nearest, no draw (:353-354)
    out = aug(src.to(DEV), None if lab is None else lab.to(DEV), params=params)
    bad_px = tot_px = 0
    for i in range(N):
        i0, _, lb, al = oaug.augment_sample(src[i].numpy(), None if lab is None else lab[i].numpy(), params[i], crop, MEAN, STD)
        got = out['image'][i].cpu().double().numpy()
        wrong = (np.abs(got - i0) > 2e-3).any(axis=0)               # a whole-pixel disagreement = a flipped rounding
        bad_px += int(wrong.sum())
        tot_px += wrong.size
        gm = out['mask'][i, 0].cpu().double().numpy()
        assert (np.abs(gm - al) > 2e-3).mean() <= 2e-3
        if with_labels:
            assert (out['labels'][i, 0].cpu().numpy() != lb).mean() <= 2e-3
    assert bad_px <= 2e-3 * tot_px, (bad_px, tot_px)
    assert float(out['mask'].min()) == 0.0                          # some crop reaches past its source image: reflected
                                                                    # image, zero mask (and label 255) out there
    assert float(out['mask'].min()) >= 0.0 and float(out['mask'].max()) <= 1.0 + 1e-6
    if cfg.get('strong_colour'):
        assert 'image_stu' in out and torch.isfinite(out['image_stu']).all()


def test_trainer_cli_with_device_side_staging(tmp_path, monkeypatch):
    """The reference's augmentation options on the command line, served by the device-side staging."""
    from click.testing import CliRunner
    import train_seg_semisup_mask_mt as trainer
    monkeypatch.chdir(tmp_path)
    args = ['--job_desc', 'aug', '--synthetic', '--synthetic_source_size', '90,120', '--arch', 'resnet101_deeplab_imagenet',
            '--freeze_bn', '--batch_size', '2', '--crop_size', '65,65', '--aug_scale_hung', '--aug_hflip', '--aug_strong_colour',
            '--num_epochs', '1', '--iters_per_epoch', '2', '--synthetic_val_batches', '1']
    res = CliRunner().invoke(trainer.experiment, args, catch_exceptions=False)
    assert res.exit_code == 0, res.output
    log = open(tmp_path / 'results' / 'train_seg_semisup_mask_mt' / 'log_aug.txt').read()
    assert 'Epoch 1' in log
    # --aug_rot_mag / --aug_max_scale select the rotate + scale crop (train_seg_semisup_mask_mt.py:153-155)
    args2 = ['--job_desc', 'rot', '--synthetic', '--synthetic_source_size', '90,120', '--arch', 'resnet101_deeplab_imagenet',
             '--freeze_bn', '--batch_size', '2', '--crop_size', '65,65', '--aug_rot_mag', '20', '--aug_max_scale', '1.5',
             '--aug_hflip', '--num_epochs', '1', '--iters_per_epoch', '2', '--synthetic_val_batches', '1']
    res = CliRunner().invoke(trainer.experiment, args2, catch_exceptions=False)
    assert res.exit_code == 0, res.output
