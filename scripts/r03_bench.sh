# bench line + kernel stats only.  usage: bash scripts/r03_bench.sh OUTDIR [workload] [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03x}; mkdir -p $O; W=${2:-walk1m}; shift; shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o f --output-format csv -- python3 bench.py --workload $W --cpu-steps 0 "$@" > $O/bench_$W.json 2> $O/err.txt; echo rc=$?
python3 profiles/summarize_stats.py $O/prof/f_kernel_stats.csv 12 | tee $O/kernel_stats_$W.txt
rm -f $O/prof/f_kernel_trace.csv
python3 -c "
import json; j=json.loads(open('$O/bench_$W.json').read().strip().split('\n')[-1]); print(j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['le_firing'])"
