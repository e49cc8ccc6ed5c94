set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof -o f --output-format csv -- python3 bench.py > gpurun_out/final_bench_under_rocprof.json 2>> gpurun_out/final_bench.err
python3 profiles/summarize_stats.py gpurun_out/final_prof/f_kernel_stats.csv 22
cat gpurun_out/final_bench.json
