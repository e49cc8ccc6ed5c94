cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03d}; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests/test_gpu_dd.py tests/test_gpu_fuzz.py -x -q -m gpu -k "${2:-dd or decomposed}" > $O/dd.log 2>&1; echo "dd rc=$?"
tail -15 $O/dd.log
