# in-process decomposition probe (one GPU shared by all ranks): per-step cost and what a firing of the LE fixes costs
# and moves, new path against the round-2 whole-system gather.  usage: bash scripts/run_dd_probe.sh [OUTDIR]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ddprobe}; mkdir -p $O
for w in 1 2 4 8; do
  timeout -k 10 400 python3 tests/perf_dd_local.py $w 1000000 1000 walk 2>$O/err_$w.txt | tail -1 > $O/dd_1m_w$w.json; echo "w=$w rc=$?"
  [ $w -gt 1 ] && LAMMPS_LE_DD_FULL_GATHER=1 timeout -k 10 400 python3 tests/perf_dd_local.py $w 1000000 1000 walk 2>>$O/err_$w.txt | tail -1 > $O/dd_1m_w${w}_fullgather.json
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/dd_1m_w*.json")):
    try:
        j=json.loads(open(f).read())
        print(f.split('/')[-1], j['us_per_step'], 'us/step  firing', j['firing_ms'], 'ms  gather', j['firing_bytes_allgather_per_rank'], 'B reduce', j['firing_bytes_allreduce_per_rank'], 'B')
    except Exception as e: print(f, 'unreadable', e)
PY
