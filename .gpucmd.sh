cd /root/repo
export TMPDIR=/tmp
timeout -k 10 200 python -u -m pytest tests/test_hip_kernels.py -x -q --timeout 120 --timeout-method thread -k "fused" > gpurun_out/pool_test.log 2>&1 || { tail -n 60 gpurun_out/pool_test.log; exit 1; }
tail -n 2 gpurun_out/pool_test.log
for mode in 1 0; do
DTF_FUSE_BN_POOL=$mode DTF_WGRAD_STREAM=0 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pool$mode -- python bench.py --steps 10 --warmup 2 --no_graph > gpurun_out/prof_pool.log 2>&1 || { tail -n 30 gpurun_out/prof_pool.log; exit 1; }
f=$(ls gpurun_out/prof_pool$mode/*/*kernel_stats.csv | tail -n 1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms/step", tot/20/1e6)
for r in rows:
  if any(k in r['Name'] for k in ('pool','bn_apply','bn_bwd_reduce','bn_bwd_apply')):
    print("%-90s %6s %10.1f %8.1f %5s"%(r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e3/20, float(r['AverageNs'])/1e3, r['Percentage']))
PY
done
timeout -k 10 200 python bench.py --steps 50 --warmup 10
