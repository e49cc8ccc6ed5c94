# PMC passes on the list build (k_build_neigh): where do its wave cycles go.  PMC_SETS="A B C|D E" overrides the counter sets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
IFS='|'
for set in ${PMC_SETS:-SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES|SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH|SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL}; do
  i=$((i+1))
  IFS=' '
  timeout -k 10 400 rocprofv3 --pmc $set -d gpurun_out/pmcb$i -o p --output-format csv -- python3 bench.py --workload ${PMC_WORKLOAD:-chain1m} --steps 100 --warmup 20 --pre-roll ${PMC_PREROLL:-300} --cpu-steps 0 > gpurun_out/pmcb$i.json 2> gpurun_out/pmcb$i.err || { tail -5 gpurun_out/pmcb$i.err; exit 1; }
  IFS='|'
  python3 - <<PY
import csv,collections,glob
f=glob.glob('gpurun_out/pmcb$i/**/p_counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'][:36]
    if any(t in k for t in '${PMC_KERNELS:-k_build_neigh k_step}'.split()):
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
for k in acc:
    print(k, {c: round(v/n[(k,c)]) for c,v in acc[k].items()})
PY
  rm -rf gpurun_out/pmcb$i
done
