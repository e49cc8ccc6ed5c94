"""Repeat the 4-rank in-process decomposition test to flush out intermittent failures.  usage: stress_dd_local.py [REPS]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from systems import CHAIN_SCRIPT, lattice_chain, run_oracle
from test_gpu_dd import run_ranks_local

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s = lattice_chain(40000, nchains=2, seed=23)
script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + \
    "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 30\nrun 60\n"
o = run_oracle(script, s)
bad = 0
for k in range(reps):
    try:
        r = run_ranks_local(4, s, script, tempfile.mkdtemp())
        err = np.abs(r["x"] - o.x()).max()
        print("rep", k, "max|dx| %.2e" % err, flush=True)
        bad += err > 1e-9
    except AssertionError as e:
        print("rep", k, "FAILED", str(e)[:400], flush=True)
        bad += 1
print("failures:", bad)
