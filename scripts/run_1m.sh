set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_md.py tests/test_gpu_le.py tests/test_gpu_misc.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for nb in 1 0; do
  if [ $nb = 1 ]; then export LAMMPS_LE_NO_FUSED_BIN=1; else unset LAMMPS_LE_NO_FUSED_BIN; fi
  for wl in chain1m chain100k chain32k; do
    timeout -k 10 300 python bench.py --workload $wl --steps 2000 --warmup 500 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl no_fused_bin=$nb', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done
