#!/bin/bash
# ImageNet TFRecord delivery while training (ResNet-50 bs256), region decoder on / off.
set -e
mkdir -p gpurun_out/s7
[ -d /tmp/fake_imagenet ] || python scripts/make_fake_imagenet.py /tmp/fake_imagenet 24 > gpurun_out/s7/make.log 2>&1
for region in 1 0; do
  DTF_JPEG_REGION=$region timeout -k 10 400 python resnet_imagenet_main.py --data_dir /tmp/fake_imagenet --dtype bf16 --batch_size 256 --distribution_strategy off --train_steps 200 --skip_eval --log_steps 20 > gpurun_out/s7/imagenet_region$region.log 2>&1
  echo "== DTF_JPEG_REGION=$region"; grep -h "examples_per_second" gpurun_out/s7/imagenet_region$region.log | sed "s/.*'global step'/step/"
done
