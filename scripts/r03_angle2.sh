cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_angle2; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_angle.py tests/test_gpu_md.py -x -q -m gpu > $O/angle.log 2>&1; echo "angle rc=$?"; tail -5 $O/angle.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o f --output-format csv -- python3 bench.py --workload walk1m_angles --cpu-steps 0 > $O/bench_walk1m_angles.json 2> $O/err.txt; echo rc=$?; tail -3 $O/err.txt
python3 profiles/summarize_stats.py $O/prof/f_kernel_stats.csv 10 | cut -c1-140
rm -rf $O/prof
python3 -c "
import json; j=json.loads(open('$O/bench_walk1m_angles.json').read().strip().split('\n')[-1]); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['extruders'], j['le_firing'], j['fene_warnings'])"
