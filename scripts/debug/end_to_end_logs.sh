set -e
mkdir -p gpurun_out/s7
python scripts/make_fake_imagenet.py /tmp/fake_imagenet 16 > gpurun_out/s7/imagenet_tfrecords.log 2>&1
timeout -k 10 400 python resnet_imagenet_main.py --data_dir /tmp/fake_imagenet --dtype bf16 --batch_size 256 --distribution_strategy off --train_steps 120 --skip_eval --log_steps 20 >> gpurun_out/s7/imagenet_tfrecords.log 2>&1
python scripts/make_fake_cifar.py /tmp/fake_cifar > gpurun_out/s7/cifar_records.log 2>&1
timeout -k 10 300 python resnet_cifar_main.py --data_dir /tmp/fake_cifar --dtype bf16 --batch_size 128 --distribution_strategy off --train_epochs 3 --skip_eval --log_steps 100 >> gpurun_out/s7/cifar_records.log 2>&1
timeout -k 10 300 python resnet_imagenet_main.py --use_synthetic_data --dtype bf16 --batch_size 256 --distribution_strategy off --train_steps 300 --skip_eval --log_steps 100 > gpurun_out/s7/imagenet_synth.log 2>&1
timeout -k 10 300 python resnet_cifar_main.py --use_synthetic_data --dtype bf16 --batch_size 128 --distribution_strategy off --train_steps 600 --skip_eval --log_steps 200 > gpurun_out/s7/cifar_synth.log 2>&1
DTF_BACKEND=gloo timeout -k 10 300 python resnet_imagenet_main.py --use_synthetic_data --dtype bf16 --batch_size 256 --distribution_strategy parameter_server --num_gpus 2 --train_steps 100 --skip_eval --log_steps 20 > gpurun_out/s7/ps.log 2>&1
grep -h "examples_per_second" gpurun_out/s7/*.log | tail -30
