cd /root/repo
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests2.log 2>&1 || { tail -n 40 gpurun_out/gpu_tests2.log; exit 1; }
tail -n 2 gpurun_out/gpu_tests2.log
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --rehearse_collectives 2>/dev/null
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --rehearse_collectives --num_packs 8 2>/dev/null | cut -c1-200
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --rehearse_collectives --num_packs 2 2>/dev/null | cut -c1-200
