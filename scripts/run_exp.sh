# experiment harness: bench lines (+ per-kernel averages from a kernel trace) for a set of workloads, for the library as
# built and for rebuilds with EXTRA flags:  EXP_EXTRAS="-DA=1|-DB=2" EXP_WORKLOADS="chain1m chain8m" bash scripts/run_exp.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { steps=2000; [ $2 = chain8m ] && steps=500; [ $2 = chain32k ] && steps=4000
  timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/exp_$1_$2 -o t --output-format csv -- python3 bench.py --workload $2 --steps $steps --warmup 200 --cpu-steps 0 > gpurun_out/exp_$1_$2.json 2> gpurun_out/exp_$1_$2.err && python3 -c "
import json,csv
j=json.loads(open('gpurun_out/exp_$1_$2.json').read().strip().split('\n')[-1]); print('$1 $2', j['value'], 'k_step(events)', j['roofline']['kernel_ms'], j['roofline']['frac'])
for r in csv.DictReader(open('gpurun_out/exp_$1_$2/t_kernel_stats.csv')):
    if float(r['Percentage'])>0.5: print('    %-60s %6s calls %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
"; rm -rf gpurun_out/exp_$1_$2; }
suite() { for w in ${EXP_WORKLOADS:-chain1m}; do run $1 $w || return 1; done; }
suite asbuilt || exit 1
IFS='|'
for extra in $EXP_EXTRAS; do
  tagname=$(echo "$extra" | tr -c 'A-Za-z0-9=\n' '_')
  (cd lammps_le_amd/csrc && rm -f *.o && make -j16 EXTRA="$extra" > /dev/null 2>&1) && IFS=' ' suite "$tagname" || exit 1
  IFS='|'
done
