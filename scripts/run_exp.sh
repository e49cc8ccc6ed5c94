# experiment harness: bench lines for a set of workloads, for the library as built and for rebuilds with EXTRA flags
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout -k 10 500 python bench.py --workload $2 --steps $3 --warmup 200 --cpu-steps 0 > gpurun_out/exp_$1_$2.json 2> gpurun_out/exp_$1_$2.err && python3 -c "
import json,sys
j=json.loads(open('gpurun_out/exp_$1_$2.json').read().strip().split('\n')[-1]); print('$1 $2', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'])"; }
suite() { run $1 chain1m 2000 && run $1 walk1m 2000 && run $1 chain32k 4000 && run $1 chain8m 500; }
suite asbuilt || exit 1
IFS='|'
for extra in $EXP_EXTRAS; do
  tagname=$(echo "$extra" | tr -c 'A-Za-z0-9=\n' '_')
  (cd lammps_le_amd/csrc && rm -f kernels_md.o && make EXTRA="$extra" > /dev/null 2>&1) && suite "$tagname" || exit 1
done
