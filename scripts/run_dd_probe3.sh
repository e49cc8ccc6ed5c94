# in-process probe at the weak-scaling size: 8 ranks x 1M beads on ONE GPU (pools capped: eight ranks share 288 GB here)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ddprobe3}; mkdir -p $O
export LAMMPS_LE_RNG_W=24
timeout -k 10 560 python3 tests/perf_dd_local.py 8 8000000 300 walk 2>$O/err_8m.txt | tail -1 > $O/skip_8x1m.json; echo "8x1M rc=$?"
LAMMPS_LE_RNG_NO_SKIP=1 LAMMPS_LE_DD_FULL_GATHER=1 timeout -k 10 560 python3 tests/perf_dd_local.py 8 8000000 300 walk 2>>$O/err_8m.txt | tail -1 > $O/r02path_8x1m.json; echo "8x1M old rc=$?"
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read())
        print(f.split('/')[-1], j['us_per_step'], 'us/step (/8 =', round(j['us_per_step']/8,1), ') firing', j['firing_ms'], 'ms  gather', j['firing_bytes_allgather_per_rank'], 'B reduce', j['firing_bytes_allreduce_per_rank'], 'B')
    except Exception as e: print(f, 'unreadable', e)
PY
