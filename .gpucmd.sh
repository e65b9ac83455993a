cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_hip_executor.py tests/test_a_multirank_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/t_exec.log 2>&1 || { tail -n 40 gpurun_out/t_exec.log; exit 1; }
tail -n 2 gpurun_out/t_exec.log
timeout -k 10 200 python bench.py --model resnet56 --batch 128 --steps 300 --warmup 20 2>/dev/null
timeout -k 10 200 python bench.py --model resnet56 --batch 128 --steps 300 --warmup 20 --rehearse_collectives 2>/dev/null
