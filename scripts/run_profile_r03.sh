# round-3 evidence: the driver's 20-step line and the default 2000-step line of the default workload (walk1m), rocprofv3
# kernel stats of the default command, the two PMC passes (FETCH_SIZE / WRITE_SIZE, each in a run of its own) behind
# profiles/traffic.json, the lattice-start and 8M lines with their kernel stats
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final
mkdir -p $O
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/err.txt
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2>> $O/err.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof -o f --output-format csv -- python3 bench.py --cpu-steps 0 > $O/bench_under_rocprof.json 2>> $O/err.txt
python3 profiles/summarize_stats.py $O/prof/f_kernel_stats.csv 14 | tee $O/kernel_stats_top.txt
cp $O/prof/f_kernel_stats.csv $O/walk1m_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 bench.py --steps 100 --warmup 100 --cpu-steps 0 > $O/pmc_fetch.json 2>> $O/err.txt
python3 profiles/summarize_pmc.py $O/pmc_fetch/p_counter_collection.csv k_step | tee $O/pmc_fetch_summary.txt
python3 profiles/summarize_pmc.py $O/pmc_fetch/p_counter_collection.csv k_build | tee -a $O/pmc_fetch_summary.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 bench.py --steps 100 --warmup 100 --cpu-steps 0 > $O/pmc_write.json 2>> $O/err.txt
python3 profiles/summarize_pmc.py $O/pmc_write/p_counter_collection.csv k_step | tee $O/pmc_write_summary.txt
python3 profiles/summarize_pmc.py $O/pmc_write/p_counter_collection.csv k_build | tee -a $O/pmc_write_summary.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof1m -o f --output-format csv -- python3 bench.py --workload chain1m --cpu-steps 0 > $O/bench_chain1m.json 2>> $O/err.txt
python3 profiles/summarize_stats.py $O/prof1m/f_kernel_stats.csv 8 | tee $O/kernel_stats_chain1m_top.txt
timeout -k 10 400 python3 bench.py --workload chain1m_dense --cpu-steps 0 > $O/bench_chain1m_dense.json 2>> $O/err.txt || echo "chain1m_dense failed"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof8m -o f --output-format csv -- python3 bench.py --workload walk8m --steps 500 --cpu-steps 0 > $O/bench_walk8m.json 2>> $O/err.txt
python3 profiles/summarize_stats.py $O/prof8m/f_kernel_stats.csv 8 | tee $O/kernel_stats_8m_top.txt
cp $O/prof8m/f_kernel_stats.csv $O/walk8m_kernel_stats.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/prof $O/prof1m $O/prof8m
ls -la $O | head -40
cat $O/bench_driver_args.json $O/bench_default.json
