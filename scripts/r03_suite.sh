# GPU suite + bench line + kernel stats (round 3 working call)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03b}; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_fullsize.py > $O/suite.log 2>&1; echo "suite rc=$?"
tail -8 $O/suite.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o f --output-format csv -- python3 bench.py --cpu-steps 0 > $O/bench_under_rocprof.json 2> $O/err.txt; echo rc=$?
python3 profiles/summarize_stats.py $O/prof/f_kernel_stats.csv 14 | tee $O/kernel_stats_top.txt
rm -f $O/prof/f_kernel_trace.csv
python3 -c "
import json; j=json.loads(open('$O/bench_under_rocprof.json').read().strip().split('\n')[-1]); print(j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['le_firing'])"
