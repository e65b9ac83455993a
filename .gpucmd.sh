cd /root/repo
export TMPDIR=/tmp
timeout -k 10 400 python scripts/debug/check_resnet50_bs256.py > gpurun_out/numerics_now.txt 2>&1 || { tail -n 30 gpurun_out/numerics_now.txt; exit 1; }
tail -n 16 gpurun_out/numerics_now.txt
OUT=gpurun_out/profile_step2 timeout -k 10 500 sh scripts/profile_step.sh > gpurun_out/profile_step2.txt 2>&1 || { tail -n 30 gpurun_out/profile_step2.txt; exit 1; }
timeout -k 10 200 python bench.py --steps 300 --warmup 10 2>/dev/null
