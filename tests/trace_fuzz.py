"""Step-by-step trace of one fuzz scenario (tests/test_gpu_fuzz.py): runs product and oracle one step at a time and
reports the first step where forces, special lists or bonds differ.  Usage: python tests/trace_fuzz.py SEED..."""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from systems import *
from test_gpu_le import LE, barrier_types, melted

from trace_fuzz_lib import scenario

for seed in [int(a) for a in sys.argv[1:]]:
    s, script = scenario(seed)
    o = run_oracle(script, s); p = run_product(script, s, tempfile.mkdtemp())
    def snap(e, is_p):
        if is_p:
            return e.gather("num_bond"), e.gather("bond_atom"), e.gather("nspecial"), e.gather("special")
        nb, bt, ba = e.bond_table(); ns, sp = e.special_table()
        return nb, ba, ns, sp
    prev = None
    for step in range(1, 46):
        prev_p, prev_o = snap(p, True), snap(o, False)
        o.run(1); p.command("run 1")
        dx = np.abs(p.gather("x") - o.x()); df = np.abs(p.gather("f") - o.f())
        if df.max() > 1e-6:
            t = np.unravel_index(df.argmax(), df.shape)[0]
            nb, bt, ba = o.bond_table(); ns, sp = o.special_table()
            print("   FORCE differs after step", step, "tag", t + 1, "df", df[t], "f_o", o.f()[t], "bonds", ba[t,:nb[t]], "types", bt[t,:nb[t]], "special", ns[t], sp[t,:ns[t,2]])
            xo = o.x(); L = s["box"][0][1]
            for u in list(ba[t,:nb[t]]) + list(sp[t,:ns[t,2]]):
                d = xo[t] - xo[u-1]; dm = (d + L/2) % L - L/2
                print("      partner", u, "raw |d| %.4f  minimg |d| %.4f" % (np.linalg.norm(d), np.linalg.norm(dm)), "nb", nb[u-1], "bonds", ba[u-1,:nb[u-1]], "nspecial", ns[u-1], sp[u-1,:ns[u-1,2]])
            break

        if step % 1 == 0: print("   step", step, "max|dx| %.3e at tag %d" % (dx.max(), np.unravel_index(dx.argmax(), dx.shape)[0] + 1), "fene warn", o.fene_warnings(), p.stat("fene_warnings"))
        nsp, spp = p.gather("nspecial"), p.gather("special"); nso, spo = o.special_table()
        bad = [t for t in range(len(nsp)) if tuple(nsp[t]) != tuple(nso[t]) or list(spp[t,:nsp[t,2]]) != list(spo[t,:nso[t,2]])]
        if bad:
            print("   SPECIAL lists differ after step", step, "at tags", [t+1 for t in bad[:10]])
            for t in bad[:4]:
                print("      tag", t+1, "P", nsp[t], spp[t,:nsp[t,2]], " O", nso[t], spo[t,:nso[t,2]], "bonds O", o.bond_table()[2][t,:o.bond_table()[0][t]])
            break
        a, b = p.bond_set(), o.bond_set()
        if a != b:
            import ctypes
            n = len(s["x"])
            pi = np.zeros(n + 2, dtype=np.int32); pd = np.zeros(n + 2)
            p.lib.lammps_le_debug_le_array.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            p.lib.lammps_le_debug_le_array(p.lmp, 1, n + 2, pi.ctypes.data_as(ctypes.c_void_p), pd.ctypes.data_as(ctypes.c_void_p))
            oi = np.zeros(n, dtype=np.int32); od = np.zeros(n)
            o.L.leo_debug_scratch(o.h, oi.ctypes.data_as(ctypes.c_void_p), None, od.ctypes.data_as(ctypes.c_void_p))
            diff = np.nonzero(pi[1:n+1] != oi)[0] + 1
            print("    partner arrays differ at tags", diff[:20], "P:", pi[diff[:20]], "O:", oi[diff[:20] - 1])
            for t in diff[:6]:
                print("      tag", t, "P rsq(pair t)", pd[t], "rsq(pair t-2)", pd[t-2], " O dist/prob", od[t-1])
            for (t_, lo, hi) in sorted((a - b) | (b - a))[:2]:
                for t in range(lo - 1, hi + 2):
                    for name, sn in (("P", prev_p), ("O", prev_o)):
                        nb, ba, ns, sp = sn
                        print("    before: %s bead %d nb %d bonds %s nspecial %s special %s" % (name, t, nb[t-1], ba[t-1,:nb[t-1]], ns[t-1], sp[t-1,:ns[t-1,2]]))
            n1 = int(script.split("extrusion")[1].split()[0]); 
            print("  DIVERGED at step", step, "product-only", sorted(a - b)[:8], "oracle-only", sorted(b - a)[:8])
            print("  counters p", [p.extract_fix(f,0,1,0) for f in ("loop","loading","unloading")], "o", [o.fix_vector(f)[0] for f in ("loop","loading","unloading")])
            nb, bt, ba = o.bond_table()
            for (t_, lo, hi) in sorted((a - b) | (b - a))[:6]:
                for t in (lo - 1, lo, hi, hi + 1):
                    print("    bead", t, "type", s["type"][t-1], "oracle bonds", ba[t-1,:nb[t-1]], "x", o.x()[t-1])
            break
    else:
        print("  no divergence (step-by-step)")
