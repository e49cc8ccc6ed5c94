# in-process probe (all ranks threads of one process on ONE GPU): strong 1M at 4 / 8 ranks, weak 8 x 1M
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03_ddprobe4}; mkdir -p $O
for w in 4 8; do
  timeout -k 10 400 python3 tests/perf_dd_local.py $w 1000000 1000 walk 2>$O/err_$w.txt | tail -1 > $O/dd_1m_w$w.json; echo "w=$w rc=$?"
done
LAMMPS_LE_RNG_W=24 timeout -k 10 560 python3 tests/perf_dd_local.py 8 8000000 300 walk 2>$O/err_8m.txt | tail -1 > $O/dd_8x1m.json; echo "8x1M rc=$?"
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read())
        print(f.split('/')[-1], j['us_per_step'], 'us/step (per rank', round(j['us_per_step']/j['world'],1), ') firing', j['firing_ms'], 'ms')
    except Exception as e: print(f, 'unreadable', e)
PY
