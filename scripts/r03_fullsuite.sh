cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03s}; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/suite_full.log 2>&1; echo "suite rc=$?"
tail -8 $O/suite_full.log
