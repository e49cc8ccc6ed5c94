# kernels of ONE rank of a decomposed run, per step and per rebuild: every rank a process of its own under its own rocprofv3
# (file-mailbox transport, all on the one GPU).  usage: bash scripts/r03_ddrank_prof.sh [WORLD] [NBEADS] [STEPS] [OUTDIR]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-4}; NB=${2:-1000000}; ST=${3:-1000}; O=gpurun_out/${4:-r03_ddrank}; mkdir -p $O
S=ddr$$
pids=""
for r in $(seq 0 $((W-1))); do
  if [ $r -eq 1 ] || [ $W -eq 1 ]; then
    timeout -k 10 800 rocprofv3 --kernel-trace --stats -d $O/p$r -o d --output-format csv -- python3 tests/perf_dd_rank.py $r $W $S $NB $ST walk > $O/rank$r.json 2> $O/err$r.txt &
  else
    timeout -k 10 800 python3 tests/perf_dd_rank.py $r $W $S $NB $ST walk > $O/rank$r.json 2> $O/err$r.txt &
  fi
  pids="$pids $!"
done
rc=0; for p in $pids; do wait $p || rc=$?; done; echo "ranks rc=$rc"
R=1; [ $W -eq 1 ] && R=0
python3 profiles/summarize_stats.py $O/p$R/d_kernel_stats.csv 60 | cut -c1-160 | tee $O/rank${R}_kernel_stats_w${W}.txt
cat $O/rank*.json
rm -rf $O/p*
