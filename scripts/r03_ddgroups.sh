# fix nve / fix langevin on groups - decomposed and under Atom::sort: the targeted tests + the two fuzz3 sweeps over seed ranges
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ddgroups; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_dd.py tests/test_gpu_md.py -q -m gpu -x -k "fixes_on_groups or segments_are_skipped" > $O/dd.log 2>&1; echo "targeted rc=$?"; tail -25 $O/dd.log
LE_FUZZ3_SEEDS=${1:-0:0} LE_FUZZ3_MD_SEEDS=${2:-0:0} timeout -k 10 1000 python3 -m pytest tests/test_gpu_fuzz3.py -q -m gpu -k "mixed_decomposed or md_settings" > $O/fuzz3.log 2>&1; echo "fuzz3 rc=$?"; tail -8 $O/fuzz3.log
