# list-build experiment: variants by macro, bench line + kernel stats + counters for each
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_buildexp; mkdir -p $O
for v in 0 1 2 3; do
  (cd lammps_le_amd/csrc && rm -f kernels_neigh.o && make -j16 EXTRA="-DBUILD_MASKWALK=$v" > /dev/null 2>&1) || { echo "build $v failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p$v -o t --output-format csv -- python3 bench.py --steps 600 --warmup 100 --pre-roll 1010 --cpu-steps 0 > $O/b$v.json 2> $O/b$v.err
  python3 -c "
import json,csv
j=json.loads(open('$O/b$v.json').read().strip().split('\n')[-1]); print('variant $v', j['value'])
for r in csv.DictReader(open('$O/p$v/t_kernel_stats.csv')):
    if 'k_build_neigh' in r['Name']: print('    build', r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
"
  rm -rf $O/p$v
  if [ $v -le 1 ]; then
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
      timeout -k 10 300 rocprofv3 --pmc $set -d $O/c$v -o p --output-format csv -- python3 bench.py --steps 100 --warmup 20 --pre-roll 1010 --cpu-steps 0 > /dev/null 2> $O/c$v.err || { tail -3 $O/c$v.err; continue; }
      python3 - <<PY
import csv,collections,glob
f=glob.glob('$O/c$v/**/p_counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(f)):
    if 'k_build_neigh' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print('    counters v$v', {c: round(v/n[c]) for c,v in acc.items()})
PY
      rm -rf $O/c$v
    done
  fi
done
(cd lammps_le_amd/csrc && rm -f kernels_neigh.o && make -j16 > /dev/null 2>&1)
