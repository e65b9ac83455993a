cd /root/repo
export TMPDIR=/tmp
DTF_WGRAD_STREAM=0 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof56 -- python bench.py --model resnet56 --batch 128 --steps 20 --warmup 2 --no_graph > gpurun_out/prof56.log 2>&1 || { tail -n 30 gpurun_out/prof56.log; exit 1; }
python scripts/summarize_profile.py stats gpurun_out/prof56 30
