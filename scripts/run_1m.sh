set -e
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  timeout -k 10 300 python bench.py --workload chain1m --steps 2000 --warmup 500 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chain1m', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
