set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
for wl in chain32k chain100k; do for l in 1 4; do
  LAMMPS_LE_LPB=$l timeout -k 10 200 python bench.py --workload $wl --steps 4000 --warmup 500 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl LPB=$l', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
timeout -k 10 300 python bench.py --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chain1m', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
