set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1m -o f --output-format csv -- python3 bench.py --cpu-steps 0 > gpurun_out/prof1m.json 2> gpurun_out/prof1m.err
python3 - <<'PY'
import csv, numpy as np, json
rows=list(csv.DictReader(open('gpurun_out/prof1m/f_kernel_trace.csv')))
ks=sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows])
rng=[(s,e) for s,e,n in ks if 'k_rng_calls' in n]
def overl(s,e): return any(not (e<=a or s>=b) for a,b in rng)
pos=-1; byp={}
for s,e,n in ks:
    if 'k_build_neigh' in n: pos=0; continue
    if 'k_step<' in n and pos>=0:
        if not overl(s,e): byp.setdefault(pos,[]).append((e-s)/1e3)
        pos+=1
print(json.load(open('gpurun_out/prof1m.json'))['value'], ' k_step by position:', ' '.join('%d:%.1f' % (p, np.mean(byp[p])) for p in sorted(byp)[:10]))
PY
