# in-process probe: decomposed step cost with and without Langevin segment skipping (all ranks share ONE GPU)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ddprobe2}; mkdir -p $O
for w in 4 8; do
  timeout -k 10 400 python3 tests/perf_dd_local.py $w 1000000 1000 walk 2>$O/err_$w.txt | tail -1 > $O/skip_w$w.json; echo "w=$w rc=$?"
  LAMMPS_LE_RNG_NO_SKIP=1 timeout -k 10 400 python3 tests/perf_dd_local.py $w 1000000 1000 walk 2>>$O/err_$w.txt | tail -1 > $O/noskip_w$w.json
done
timeout -k 10 500 python3 tests/perf_dd_local.py 8 8000000 300 walk 2>$O/err_8m.txt | tail -1 > $O/skip_8x1m.json; echo "8x1M rc=$?"
LAMMPS_LE_RNG_NO_SKIP=1 LAMMPS_LE_DD_FULL_GATHER=1 timeout -k 10 500 python3 tests/perf_dd_local.py 8 8000000 300 walk 2>>$O/err_8m.txt | tail -1 > $O/r02path_8x1m.json; echo "8x1M old rc=$?"
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read())
        print(f.split('/')[-1], j['us_per_step'], 'us/step  firing', j['firing_ms'], 'ms  gather', j['firing_bytes_allgather_per_rank'], 'B reduce', j['firing_bytes_allreduce_per_rank'], 'B')
    except Exception as e: print(f, 'unreadable', e)
PY
