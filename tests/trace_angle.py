"""step-by-step tracer for the LE + angles scenario (debug helper): first step at which product and oracle differ"""
import sys, os, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_angle import semiflexible, ANGLE_SCRIPT
from systems import OracleScript, run_product
s = semiflexible(3000, 3, seed=6)
head = ANGLE_SCRIPT + """angle_style harmonic
angle_coeff 1 3.0 170.0
angle_coeff 2 1.0 100.0
fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion 7 1 1 1 1.0 2
fix loading all ex_load 5 1 1 1.12 2 prob 0.3 684474 iparam 1 1 jparam 1 1 atype 2
fix unloading all ex_unload 6 2 0.5 prob 0.4 456456
thermo 0
"""
osc = OracleScript(s); osc.run(head); o = osc.o
p = run_product(head, s, tempfile.mkdtemp())
for step in range(1, 65):
    o.run(1); p.command("run 1")
    bo, bp = o.bond_set(), p.bond_set()
    dx = np.abs(p.gather("x") - o.x()).max()
    ao, ap = o.angle_set(), p.angle_set()
    print(step, "dx %.2e" % dx, "bonds2", len([b for b in bo if b[0] == 2]), len([b for b in bp if b[0] == 2]), "angles", o.nangles(), p.extract_setting("nangles"),
          [o.fix_vector(f)[0] for f in ("loop", "loading", "unloading")], [p.extract_fix(f, 0, 1, 0) for f in ("loop", "loading", "unloading")], flush=True)
    if dx > 1e-9:
        na, at, a1, a2, a3 = o.angle_table()
        pn = p.gather("num_angle"); pt = p.gather("angle_type"); p1 = p.gather("angle_atom1"); p2 = p.gather("angle_atom2"); p3 = p.gather("angle_atom3")
        bad = 0
        for i in range(len(na)):
            ro = [(at[i, m], a1[i, m], a2[i, m], a3[i, m]) for m in range(na[i])]
            rp = [(pt[i, m], p1[i, m], p2[i, m], p3[i, m]) for m in range(pn[i])]
            if ro != rp:
                bad += 1
                if bad < 6: print("atom", i + 1, "oracle", ro, "product", rp)
        print("atoms with different angle tables:", bad)
        # asymmetric copies in the oracle: angles whose lowest atom holds no copy
        held = set()
        for i in range(len(na)):
            for m in range(na[i]):
                held.add((i + 1, min(a1[i, m], a3[i, m]), a2[i, m], max(a1[i, m], a3[i, m])))
        asym = 0
        for (holder, x, c, y) in held:
            for other in (x, c, y):
                if (other, x, c, y) not in held: asym += 1
        print("asymmetric copies (some atom of an angle holds no copy of it):", asym)
        break
    if bo != bp or ao != ap:
        print("BOND DIFF only oracle", sorted(bo - bp)[:10], "only product", sorted(bp - bo)[:10])
        print("ANGLE DIFF only oracle", sorted(ao - ap)[:10], "only product", sorted(ap - ao)[:10])
        break
