set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for bits in ${DIAG_BITS:-8 2 128}; do
  LAMMPS_LE_DIAG_STEP=$bits timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/diag$bits -o d --output-format csv -- python3 bench.py --steps 300 --warmup 100 --pre-roll 200 --cpu-steps 0 > gpurun_out/diag$bits.json 2> gpurun_out/diag$bits.err
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/diag$bits/d_kernel_stats.csv')))
for r in rows:
    if 'k_step' in r['Name']: print('bits=$bits', r['Name'][:70], r['Calls'], r['AverageNs'][:8])
PY
done
