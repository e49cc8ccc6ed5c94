cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 1 2 4 8; do timeout -k 10 300 python tests/perf_dd_local.py $w 1000000 1000 2>&1 | tail -1 | cut -c1-200; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/dd8 -o d --output-format csv -- python3 tests/perf_dd_local.py 8 1000000 1000 > gpurun_out/dd8.log 2>&1
python3 profiles/summarize_stats.py gpurun_out/dd8/d_kernel_stats.csv 40 | cut -c1-150
rm -rf gpurun_out/dd8
