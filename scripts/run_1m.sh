set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for wl in chain32k chain100k chain1m; do
  timeout -k 10 300 python bench.py --workload $wl --steps 3000 --warmup 500 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
