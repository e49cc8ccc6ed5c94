# list build: staging capacity / occupancy variants on one workload (libraries prebuilt under build/ab/), kernel times by rocprofv3
# usage: AB_WORKLOAD=walk1m bash scripts/r03_abcap.sh lib_a.so lib_b.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_abcap; mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  export LAMMPS_LE_LIBRARY=build/ab/$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p -o t --output-format csv -- python3 bench.py --workload ${AB_WORKLOAD:-walk1m} --steps ${AB_STEPS:-1000} --cpu-steps 0 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python3 -c "
import json,csv
j=json.loads(open('$O/b.json').read().strip().split('\n')[-1]); print('rep $rep $lib', j['value'], 'events', j['roofline']['kernel_ms'], end='  ')
for r in csv.DictReader(open('$O/p/t_kernel_stats.csv')):
    if 'k_build_neigh' in r['Name'] or 'k_step' in r['Name']: print(r['Name'][12:26], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us', end='  ')
print()
"
  rm -rf $O/p
done; done
