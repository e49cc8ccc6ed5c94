# round 3, first call: the self-launch test, the new default line (walk1m) at the driver's arguments and at the defaults
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_dd.py -x -q -m gpu -k "bench_launches" > $O/test_launch.log 2>&1; echo "launch test rc=$?"
tail -3 $O/test_launch.log
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/err1.txt; echo rc=$?
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/err2.txt; echo rc=$?
cat $O/bench_driver_args.json $O/bench_default.json
