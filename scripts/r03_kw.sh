# one-off check of host-side additions (thermo keywords, fix langevin keywords, dumps on groups) + the MD settings sweep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_kw; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_md.py tests/test_gpu_misc.py tests/test_gpu_angle.py -q -m gpu -x -k "${2:-langevin_keywords or dumps_on_a_group or dump_ or thermo}" > $O/kw.log 2>&1; echo "kw rc=$?"; tail -25 $O/kw.log
LE_FUZZ3_SEEDS=0:0 LE_FUZZ3_MD_SEEDS=${1:-0:250} timeout -k 10 900 python3 -m pytest tests/test_gpu_fuzz3.py -q -m gpu -k "md_settings" > $O/fuzz3.log 2>&1; echo "fuzz3 rc=$?"; tail -8 $O/fuzz3.log
