cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03h}; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_angle.py -x -q -m gpu > $O/angle.log 2>&1; echo "angle rc=$?"
tail -40 $O/angle.log
