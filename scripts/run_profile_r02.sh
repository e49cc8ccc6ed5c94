# round-2 evidence: the driver's 20-step line, the default 2000-step line, the rocprofv3 kernel stats of the default
# command, the two PMC passes (FETCH_SIZE / WRITE_SIZE, each in a run of its own) behind profiles/traffic.json, and the
# 8M-bead line with its kernel stats
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_final
mkdir -p $O
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/err.txt
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2>> $O/err.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o f --output-format csv -- python3 bench.py --cpu-steps 0 > $O/bench_under_rocprof.json 2>> $O/err.txt
python3 profiles/summarize_stats.py $O/prof/f_kernel_stats.csv 14 | tee $O/kernel_stats_top.txt
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 100 --warmup 100 --cpu-steps 0 > $O/pmc_fetch.json 2>> $O/err.txt
python3 profiles/summarize_pmc.py $O/pmc_fetch/p_counter_collection.csv k_step | tee $O/pmc_fetch_summary.txt
python3 profiles/summarize_pmc.py $O/pmc_fetch/p_counter_collection.csv k_build | tee -a $O/pmc_fetch_summary.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --steps 100 --warmup 100 --cpu-steps 0 > $O/pmc_write.json 2>> $O/err.txt
python3 profiles/summarize_pmc.py $O/pmc_write/p_counter_collection.csv k_step | tee $O/pmc_write_summary.txt
python3 profiles/summarize_pmc.py $O/pmc_write/p_counter_collection.csv k_build | tee -a $O/pmc_write_summary.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof8m -o f --output-format csv -- python3 bench.py --workload chain8m --cpu-steps 0 > $O/bench_chain8m.json 2>> $O/err.txt
python3 profiles/summarize_stats.py $O/prof8m/f_kernel_stats.csv 8 | tee $O/kernel_stats_8m_top.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch8m -o p --output-format csv -- python3 bench.py --workload chain8m --steps 50 --warmup 50 --cpu-steps 0 > $O/pmc_fetch8m.json 2>> $O/err.txt
python3 profiles/summarize_pmc.py $O/pmc_fetch8m/p_counter_collection.csv k_step | tee $O/pmc_fetch8m_summary.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write8m -o p --output-format csv -- python3 bench.py --workload chain8m --steps 50 --warmup 50 --cpu-steps 0 > $O/pmc_write8m.json 2>> $O/err.txt
python3 profiles/summarize_pmc.py $O/pmc_write8m/p_counter_collection.csv k_step | tee $O/pmc_write8m_summary.txt
rm -rf $O/pmc_fetch8m $O/pmc_write8m $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv
rm -f $O/prof/f_kernel_trace.csv $O/prof8m/f_kernel_trace.csv $O/pmc_fetch/p_counter_collection.csv.bak
ls -la $O $O/prof $O/pmc_fetch | head -40
cat $O/bench_driver_args.json $O/bench_default.json $O/bench_chain8m.json
