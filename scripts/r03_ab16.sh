cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab16; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_md.py tests/test_gpu_le.py tests/test_gpu_misc.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not 8m" > $O/quick.log 2>&1; echo "quick rc=$?"; tail -3 $O/quick.log
for mode in list16 list32 list16 list32; do
  if [ $mode = list32 ]; then export LAMMPS_LE_NO_LIST16=1; else unset LAMMPS_LE_NO_LIST16; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p -o t --output-format csv -- python3 bench.py --cpu-steps 0 > $O/b_$mode.json 2> $O/b.err
  python3 -c "
import json,csv
j=json.loads(open('$O/b_$mode.json').read().strip().split('\n')[-1]); print('$mode', j['value'], 'events', j['roofline']['kernel_ms'])
for r in csv.DictReader(open('$O/p/t_kernel_stats.csv')):
    if 'k_build_neigh' in r['Name'] or 'k_step' in r['Name']: print('    ', r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
"
  rm -rf $O/p
done
unset LAMMPS_LE_NO_LIST16
timeout -k 10 400 python3 bench.py --workload walk8m --steps 300 --cpu-steps 0 > $O/b8_16.json 2>$O/b.err; LAMMPS_LE_NO_LIST16=1 timeout -k 10 400 python3 bench.py --workload walk8m --steps 300 --cpu-steps 0 > $O/b8_32.json 2>>$O/b.err
python3 -c "
import json
for m in ('16','32'):
    j=json.loads(open('$O/b8_'+m+'.json').read().strip().split('\n')[-1]); print('walk8m list'+m, j['value'], 'k_step events', j['roofline']['kernel_ms'])"
