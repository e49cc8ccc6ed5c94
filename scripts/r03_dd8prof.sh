cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_dd8prof; mkdir -p $O
export LAMMPS_LE_RNG_W=24
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d $O/p -o d --output-format csv -- python3 tests/perf_dd_local.py 8 8000000 200 walk > $O/dd8.json 2> $O/err.txt; echo rc=$?
python3 profiles/summarize_stats.py $O/p/d_kernel_stats.csv 45 | cut -c1-150 | tee $O/dd8_kernel_stats.txt
rm -rf $O/p
python3 -c "
import json; j=json.loads(open('$O/dd8.json').read().strip().split('\n')[-1]); print(j['us_per_step'], j['firing_ms'])"
