set -e
cd $GRAFT_REPO_ROOT
for wl in chain100k chain250k chain500k chain1m; do for a in 0 100000000; do
  LAMMPS_LE_LPB=1 LAMMPS_LE_AHEAD_MAX_N=$a timeout -k 10 300 python bench.py --workload $wl --steps 2000 --warmup 500 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl AHEAD_MAX=$a', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
