"""Full-size parity probe (BASELINE config: 1M-bead chain): product vs oracle through the first firings of all three LE
fixes (steps 1001-1003 and 2001-2003).  Prints bond-set equality and the position difference.  ~5 min of CPU."""
import os, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data
from systems import OracleScript
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2004
sysd = lattice_chains(n, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.01, punload=0.01)
lmp = lammps(cmdargs=["-screen", "none"])
for ln in script.split("\n"):
    lmp.command(ln)
t0 = time.time()
lmp.command("run %d" % steps)
print("product: %d steps in %.1f s, bonds %d" % (steps, time.time() - t0, lmp.get_thermo("bonds")), flush=True)
osc = OracleScript(dict(sysd))
for ln in script.split("\n"):
    if not ln.startswith("thermo_style"):
        osc.line(ln)
t0 = time.time()
osc.o.run(steps)
print("oracle: %d steps in %.1f s, bonds %d" % (steps, time.time() - t0, osc.o.nbonds()), flush=True)
pb, ob = lmp.bond_set(), osc.o.bond_set()
ext = [b for b in ob if b[0] == 2]
L = sysd["box"][0][1] - sysd["box"][0][0]
d = lmp.gather("x") - osc.o.x()
dx = np.abs((d + L / 2) % L - L / 2).max()          # a bead may be wrapped at a different rebuild: compare modulo the box
print("only in product:", sorted(pb - ob)[:10], "only in oracle:", sorted(ob - pb)[:10], flush=True)
print("images equal after unwrapping:", np.abs((lmp.gather("x") + lmp.gather("image") * L) - (osc.o.x() + osc.o.image() * L)).max())
print("extruders oracle %d product %d; bond sets equal: %s; max|dx| %.3e; fix counters equal: %s" % (
    len(ext), len([b for b in pb if b[0] == 2]), pb == ob, dx,
    all(lmp.extract_fix(f, 0, 1, k) == osc.o.fix_vector(f)[k] for f in ("loop", "loading", "unloading") for k in (0, 1))))
