mkdir -p gpurun_out/s5
timeout -k 10 300 python -u -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "bn" --timeout 120 --timeout-method thread > gpurun_out/s5/bn_tests.txt 2>&1 || { tail -20 gpurun_out/s5/bn_tests.txt; exit 1; }
DTF_BN_REVERSE=7 timeout -k 10 300 python -u -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "bn" --timeout 120 --timeout-method thread > gpurun_out/s5/bn_tests7.txt 2>&1 || { tail -20 gpurun_out/s5/bn_tests7.txt; exit 1; }
for m in 0 1 2 4 5 6 7 3; do
  echo "== DTF_BN_REVERSE=$m" >> gpurun_out/s5/rev.txt
  DTF_BN_REVERSE=$m timeout -k 10 120 python bench.py --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/s5/rev.txt || exit 1
done
cat gpurun_out/s5/rev.txt
