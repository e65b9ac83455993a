cd /root/repo
export TMPDIR=/tmp
OUT=gpurun_out/profile_step timeout -k 10 500 sh scripts/profile_step.sh > gpurun_out/profile_step.txt 2>&1 || { tail -n 30 gpurun_out/profile_step.txt; exit 1; }
timeout -k 10 200 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_final1.log 2>/dev/null && cat gpurun_out/bench_final1.log
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --rehearse_collectives > gpurun_out/bench_final_reh.log 2>/dev/null && cat gpurun_out/bench_final_reh.log
timeout -k 10 200 python bench.py --model resnet56 --batch 128 --steps 200 --warmup 20 > gpurun_out/bench_final56.log 2>/dev/null && cat gpurun_out/bench_final56.log
timeout -k 10 200 python scripts/bench_layers.py resnet50 256 > gpurun_out/layers_final.txt 2>&1 && tail -n 3 gpurun_out/layers_final.txt
