# quick parity (md + le) then bench lines with kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03q}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_md.py tests/test_gpu_le.py tests/test_gpu_misc.py -x -q -m gpu > $O/quick.log 2>&1; echo "quick rc=$?"; tail -3 $O/quick.log
bash scripts/r03_bench.sh $1 walk1m && bash scripts/r03_bench.sh $1 chain1m
