"""History of a few beads' bonds and special lists in product and oracle, step by step, for one fuzz scenario.
usage: python tests/trace_special.py SEED FIRST_STEP TAG..."""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from systems import *
from trace_fuzz_lib import scenario

seed, first = int(sys.argv[1]), int(sys.argv[2])
tags = [int(a) for a in sys.argv[3:]]
s, script = scenario(seed)
o = run_oracle(script, s); p = run_product(script, s, tempfile.mkdtemp())
def show(step):
    nbp, bap, nsp, spp = p.gather("num_bond"), p.gather("bond_atom"), p.gather("nspecial"), p.gather("special")
    nbo, bto, bao = o.bond_table(); nso, spo = o.special_table()
    for t in tags:
        i = t - 1
        print("step %3d tag %5d  P bonds %-18s nsp %-12s sp %-28s | O bonds %-18s nsp %-12s sp %s" % (
            step, t, list(bap[i, :nbp[i]]), list(nsp[i]), list(spp[i, :max(nsp[i, 2], 0)]), list(bao[i, :nbo[i]]), list(nso[i]),
            list(spo[i, :max(nso[i, 2], 0)])))
if first > 1:
    o.run(first - 1); p.command("run %d" % (first - 1))
show(first - 1)
for step in range(first, 46):
    o.run(1); p.command("run 1")
    show(step)
