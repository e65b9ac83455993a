set -e
cd /root/repo
timeout -k 10 300 python -u -m pytest tests/test_hip_executor.py -x -q --timeout 120 --timeout-method thread -k "folded or matches_reference" > gpurun_out/fold_test.log 2>&1 || { tail -40 gpurun_out/fold_test.log; exit 1; }
tail -5 gpurun_out/fold_test.log
DTF_EVAL_FOLD_BN=0 timeout -k 10 120 python scripts/bench_eval.py resnet50 256 50 > gpurun_out/eval_nofold.log 2>&1
timeout -k 10 120 python scripts/bench_eval.py resnet50 256 50 > gpurun_out/eval_fold.log 2>&1
tail -1 gpurun_out/eval_nofold.log gpurun_out/eval_fold.log
