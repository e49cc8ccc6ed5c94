# one-off wide randomised LE sweep (seeds beyond the suite's): 1-rank seeds 16..215, decomposed seeds 12..71
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03_fuzz_wide}; mkdir -p $O
LE_FUZZ_SEEDS=${2:-16:216} LE_FUZZ_SEEDS_DD=0:0 timeout -k 10 1100 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "not decomposed" > $O/fuzz_1rank.log 2>&1; echo "1-rank rc=$?"; tail -4 $O/fuzz_1rank.log
LE_FUZZ_SEEDS=0:0 LE_FUZZ_SEEDS_DD=${3:-12:72} timeout -k 10 1100 python3 -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "decomposed" > $O/fuzz_dd.log 2>&1; echo "decomposed rc=$?"; tail -4 $O/fuzz_dd.log
LE_FUZZ2_SEEDS=${4:-24:424} timeout -k 10 1100 python3 -m pytest tests/test_gpu_fuzz2.py -q -m gpu > $O/fuzz2.log 2>&1; echo "mixed rc=$?"; tail -12 $O/fuzz2.log
LE_FUZZ3_SEEDS=${5:-10:70} LE_FUZZ3_MD_SEEDS=${6:-16:316} timeout -k 10 1100 python3 -m pytest tests/test_gpu_fuzz3.py -q -m gpu > $O/fuzz3.log 2>&1; echo "fuzz3 rc=$?"; tail -12 $O/fuzz3.log
