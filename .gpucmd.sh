cd /root/repo
for f in 384 512 768 1024 1536; do
DTF_WGRAD_TARGET_BLOCKS=$f DTF_AUTOTUNE_IGNORE_CACHE=0 timeout -k 10 200 python bench.py --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('r50 target=$f', d['ms_per_step'], d['value'])"
done
