# anatomy of the list build at the scrambled start (diagnostic launches + counters), and a 300k-step soak of the default workload
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_anatomy; mkdir -p $O
DIAG_WORKLOAD=walk1m bash scripts/run_diag_build.sh > $O/diag_build_walk1m.txt 2>&1; cat $O/diag_build_walk1m.txt
PMC_WORKLOAD=walk1m PMC_PREROLL=1010 PMC_KERNELS=k_build_neigh bash scripts/run_pmc_build.sh > $O/pmc_build_walk1m.txt 2>&1; cat $O/pmc_build_walk1m.txt
SOAK_GEN=walk timeout -k 10 300 python3 tests/soak_1m.py 0.002 0.05 15 > $O/soak_walk1m_300k_steps.log 2>&1; tail -16 $O/soak_walk1m_300k_steps.log
