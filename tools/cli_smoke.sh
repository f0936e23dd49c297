#!/bin/bash
# End-to-end runs of the two trainers' command lines on synthetic data (GPU box): the reference's DEFAULT flags (no
# --freeze_bn: BatchNorm on batch statistics, on the executor), the experiment scripts' flags (--freeze_bn), deterministic
# mode, the rotate + scale crop, the VAT trainer. Each prints its last log lines.
set -e
cd "$(dirname "$0")/.."
W=$(mktemp -d /tmp/cli_smoke.XXXX)
run() { name=$1; shift; echo "== $name: $*"; ( cd $W && python $OLDPWD/"$@" > $name.out 2>&1 ) || { tail -20 $W/$name.out; exit 1; }; tail -3 $W/$name.out | cut -c1-200; }
COMMON="--synthetic --arch resnet101_deeplab_imagenet --batch_size 4 --crop_size 129,129 --learning_rate 3e-5 --num_epochs 1 --iters_per_epoch 4 --synthetic_val_batches 1"
run default_cli train_seg_semisup_mask_mt.py --job_desc d $COMMON
run freeze_bn train_seg_semisup_mask_mt.py --job_desc f $COMMON --freeze_bn
run deterministic train_seg_semisup_mask_mt.py --job_desc det $COMMON --freeze_bn --deterministic
run rot_scale train_seg_semisup_mask_mt.py --job_desc r $COMMON --freeze_bn --synthetic_source_size 160,200 --aug_rot_mag 20 --aug_max_scale 1.5 --aug_hflip --aug_strong_colour
V3="--synthetic --arch resnet101_deeplabv3plus_imagenet --batch_size 4 --crop_size 129,129 --learning_rate 3e-5 --num_epochs 1 --iters_per_epoch 3 --synthetic_val_batches 1"
run v3plus_freeze_bn train_seg_semisup_mask_mt.py --job_desc v3f $V3 --freeze_bn
run v3plus_default_cli train_seg_semisup_mask_mt.py --job_desc v3d $V3
run cut_mode_default_cli train_seg_semisup_mask_mt.py --job_desc cut $COMMON --mask_mode zero
run vat train_seg_semisup_vat_mt.py --job_desc v --synthetic --arch resnet101_deeplab_imagenet --freeze_bn --batch_size 2 --crop_size 65,65 --num_epochs 1 --iters_per_epoch 2 --synthetic_val_batches 1
# round 5: the U-Nets through the default ('auto' = all hand-written) engine -- no library convolution may be reached (it would raise)
run resunet_cutmix train_seg_semisup_mask_mt.py --job_desc ru --synthetic --arch resnet50unet_imagenet --batch_size 2 --crop_size 64,64 --learning_rate 3e-5 --num_epochs 1 --iters_per_epoch 2 --synthetic_val_batches 1
run denseunet_vat train_seg_semisup_vat_mt.py --job_desc dv --synthetic --arch densenet161unet_imagenet --batch_size 2 --crop_size 64,64 --num_epochs 1 --iters_per_epoch 2 --synthetic_val_batches 1
# round 6: the VAT trainer of a U-Net long enough for the hipGraph replay of the gradient passes (two eager iterations, the capture,
# replays) across an evaluation in between (train -> eval -> train: same signature, the graph is replayed again in epoch 2)
run denseunet_vat_graph train_seg_semisup_vat_mt.py --job_desc dvg --synthetic --arch densenet161unet_imagenet --batch_size 2 --crop_size 64,64 --num_epochs 2 --iters_per_epoch 5 --synthetic_val_batches 1
run resunet_cutmix_graph train_seg_semisup_mask_mt.py --job_desc rug --synthetic --arch resnet50unet_imagenet --batch_size 2 --crop_size 64,64 --learning_rate 3e-5 --num_epochs 2 --iters_per_epoch 5 --synthetic_val_batches 1
echo "cli_smoke OK"
