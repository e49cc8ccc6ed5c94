import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from systems import *
import test_gpu_le as T

def first_divergence(script, s, nsteps, chunk=1):
    tmp = tempfile.mkdtemp()
    o = run_oracle(script, s)
    p = run_product(script, s, tmp)
    for step in range(0, nsteps, chunk):
        o.run(chunk); p.command("run %d" % chunk)
        a, b = p.bond_set(), o.bond_set()
        if a != b:
            print("DIVERGED after step", step + chunk)
            print("  product-only:", sorted(x for x in a - b)[:20])
            print("  oracle-only :", sorted(x for x in b - a)[:20])
            print("  counters p:", [p.extract_fix(f,0,1,0) for f in ("loop","loading","unloading")])
            print("  counters o:", [o.fix_vector(f)[0] for f in ("loop","loading","unloading")])
            return
    print("no divergence in", nsteps, "steps; ext bonds:", len([x for x in o.bond_set() if x[0]==2]))

n = 3000
s = T.melted(n, types=T.barrier_types(n, 5))
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "load"):
    print("== ex_load only, no prob")
    first_divergence(T.le_script(n1=1000, nl=10, nu=1000, lprob="", uprob=""), s, 25)
    print("== ex_load only, prob")
    first_divergence(T.le_script(n1=1000, nl=10, nu=1000, uprob=""), s, 25)
if which in ("all", "unload"):
    print("== load + unload")
    first_divergence(T.le_script(n1=1000, nl=10, nu=10), s, 35)
if which in ("all", "ext"):
    print("== load + extrusion tp=1")
    first_divergence(T.le_script(n1=10, nl=10, nu=1000, uprob=""), s, 45)
    print("== load + extrusion tp=0.5")
    first_divergence(T.le_script(n1=10, nl=10, nu=1000, uprob="", tp=0.5), s, 45)
