# A/B of environment settings on one workload, interleaved and repeated (run-to-run spread on this box is a few %):
#   AB_WORKLOAD=chain8m AB_REPS=3 AB_ENVS="X=0|X=1|X=2" bash scripts/run_ab.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rep=0
while [ $rep -lt ${AB_REPS:-3} ]; do
  rep=$((rep+1))
  IFS='|'
  for e in $AB_ENVS; do
    IFS=' '
    env $e timeout -k 10 400 python bench.py --workload ${AB_WORKLOAD:-chain1m} --steps ${AB_STEPS:-500} --warmup 100 --cpu-steps 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
    python3 -c "
import json
j=json.loads(open('gpurun_out/ab.json').read().strip().split('\n')[-1]); print('rep $rep [$e]', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'])"
    IFS='|'
  done
  IFS=' '
done
