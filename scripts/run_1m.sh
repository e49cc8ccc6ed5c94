set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --workload chain1m --steps 2000 --warmup 500 --cpu-steps 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chain1m', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1m -o f --output-format csv -- python3 bench.py --cpu-steps 0 > gpurun_out/prof1m.json 2> gpurun_out/prof1m.err
python3 - <<'PY'
import csv, numpy as np
rows=list(csv.DictReader(open('gpurun_out/prof1m/f_kernel_trace.csv')))
ks=[(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
rng=[(s,e) for n,s,e in ks if 'k_rng_calls' in n]
def overl(s,e): return any(not (e<=a or s>=b) for a,b in rng)
for pat in ('k_step<','k_build_neigh'):
    d_in=[(e-s)/1e3 for n,s,e in ks if pat in n and overl(s,e)]
    d_out=[(e-s)/1e3 for n,s,e in ks if pat in n and not overl(s,e)]
    print(pat, 'overlapping rng: n=%d mean %.1f | clear: n=%d mean %.1f' % (len(d_in), np.mean(d_in), len(d_out), np.mean(d_out)))
print('rng calls ms', [(round((e-s)/1e6,2)) for s,e in rng])
PY
