# fix nve / fix langevin on groups in decomposed runs: the new dd test + the decomposed mixed sweep over a seed list
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ddgroups; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_dd.py -q -m gpu -x -k "fixes_on_groups_across_slabs or respa_is_refused" > $O/dd.log 2>&1; echo "dd rc=$?"; tail -25 $O/dd.log
LE_FUZZ3_SEEDS=${1:-0:0} LE_FUZZ3_MD_SEEDS=0:0 timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz3.py -q -m gpu -k "mixed_decomposed" > $O/fuzz3dd.log 2>&1; echo "fuzz3dd rc=$?"; tail -8 $O/fuzz3dd.log
