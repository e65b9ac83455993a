cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_hip_kernels.py tests/test_hip_executor.py -x -q --timeout 180 --timeout-method thread > gpurun_out/t_lazy.log 2>&1 || { tail -n 40 gpurun_out/t_lazy.log; exit 1; }
tail -n 2 gpurun_out/t_lazy.log
for f in 1 0 1 0; do
DTF_BN_LAZY_SHORTCUT=$f timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r50 lazy=$f', d['ms_per_step'], d['value'])"
done
