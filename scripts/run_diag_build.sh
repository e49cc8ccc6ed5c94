cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for bits in ${DIAG_BITS:-1 3 9 17 5 65 129}; do
  LAMMPS_LE_DIAG_BUILD=$bits timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/dg$bits -o d --output-format csv -- python3 bench.py --workload ${DIAG_WORKLOAD:-chain1m} --steps 300 --warmup 100 --cpu-steps 0 > gpurun_out/dg$bits.json 2> gpurun_out/dg$bits.err
  python3 - <<PY
import csv
for r in csv.DictReader(open('gpurun_out/dg$bits/d_kernel_stats.csv')):
    if 'k_build_neigh' in r['Name']: print('bits=$bits', r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3)
PY
  rm -rf gpurun_out/dg$bits
done
