set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload chain8m --steps 300 --warmup 100 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chain8m', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout -k 10 300 python bench.py --workload chains10x100k --steps 1000 --warmup 300 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chains10x100k', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout -k 10 700 python tests/parity_1m.py 1000000 1004 > gpurun_out/parity_1m.log 2>&1
cat gpurun_out/parity_1m.log
