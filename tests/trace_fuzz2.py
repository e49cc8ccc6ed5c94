"""First step at which product and oracle part ways in a scenario of tests/test_gpu_fuzz2.py, without changing how the
scenario cuts its run into `run` commands: the command sequence is truncated to k steps, both engines run it from scratch,
k is bisected.  usage: python tests/trace_fuzz2.py SEED..."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from systems import run_oracle, run_product
from test_gpu_fuzz2 import scenario


def truncated(script, k):
    out, left = [], k
    for ln in script.split("\n"):
        if ln.startswith("run "):
            r = min(int(ln.split()[1]), left)
            left -= r
            if r > 0:
                out.append("run %d" % r)
        else:
            out.append(ln)
    return "\n".join(out) + "\n"


def diverged(s, script, k):
    sc = truncated(script, k)
    o = run_oracle(sc, s)
    p = run_product(sc, s, tempfile.mkdtemp())
    dx = np.abs(p.gather("x") - o.x())
    t = int(np.unravel_index(dx.argmax(), dx.shape)[0])
    res = dict(k=k, dx=float(dx.max()), tag=t + 1, bonds_equal=p.bond_set() == o.bond_set(), builds=(int(p.stat("neigh_builds")), int(o.neigh_builds())),
               order_equal=bool((np.asarray(o.local_order()) == np.asarray(p.local_order() if hasattr(p, "local_order") else o.local_order())).all()))
    p.close()
    return res


for seed in [int(a) for a in sys.argv[1:]]:
    s, script, ids, flavour = scenario(seed)
    total = sum(int(ln.split()[1]) for ln in script.split("\n") if ln.startswith("run "))
    print("seed", seed, flavour, "total", total, [ln for ln in script.split("\n") if ln.startswith(("run ", "atom_modify", "newton", "fix l", "fix u", "run_style", "angle_"))])
    lo, hi = 0, total          # invariant: fine at lo, diverged at hi
    r = diverged(s, script, total)
    print("   full:", r)
    if r["dx"] < 1e-7 and r["bonds_equal"]:
        continue
    while hi - lo > 1:
        mid = (lo + hi) // 2
        r = diverged(s, script, mid)
        if r["dx"] > 1e-7 or not r["bonds_equal"]:
            hi = mid
        else:
            lo = mid
    print("   first divergence at step", hi, diverged(s, script, hi), " step before:", diverged(s, script, lo) if lo else None)
