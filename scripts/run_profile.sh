# round-end evidence: default bench line, rocprofv3 kernel stats of the same command, and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, each in a run of its own) behind profiles/traffic.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/final_prof -o f --output-format csv -- python3 bench.py > gpurun_out/final_bench_under_rocprof.json 2>> gpurun_out/final_bench.err
python3 profiles/summarize_stats.py gpurun_out/final_prof/f_kernel_stats.csv 12
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 100 --warmup 300 --cpu-steps 0 > gpurun_out/pmc_fetch.json 2>> gpurun_out/final_bench.err
python3 profiles/summarize_pmc.py gpurun_out/pmc_fetch/p_counter_collection.csv k_step > gpurun_out/pmc_fetch_summary.txt
cat gpurun_out/pmc_fetch_summary.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o p --output-format csv -- python3 bench.py --steps 100 --warmup 300 --cpu-steps 0 > gpurun_out/pmc_write.json 2>> gpurun_out/final_bench.err
python3 profiles/summarize_pmc.py gpurun_out/pmc_write/p_counter_collection.csv k_step > gpurun_out/pmc_write_summary.txt
cat gpurun_out/pmc_write_summary.txt
cat gpurun_out/final_bench.json
