"""Where do product and oracle part ways at full size?  Unwrapped position difference after S steps (no LE firing
before step 1001).  usage: parity_1m_md.py NBEADS S1 S2 ..."""
import os, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data
from systems import OracleScript
n = int(sys.argv[1]); marks = [int(a) for a in sys.argv[2:]]
sysd = lattice_chains(n, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.01, punload=0.01)
L = sysd["box"][0][1] - sysd["box"][0][0]
lmp = lammps(cmdargs=["-screen", "none"])
osc = OracleScript(dict(sysd))
for ln in script.split("\n"):
    lmp.command(ln)
    if not ln.startswith("thermo_style"):
        osc.line(ln)
done = 0
for m in marks:
    lmp.command("run %d" % (m - done)); osc.o.run(m - done); done = m
    xu_p = lmp.gather("x") + lmp.gather("image") * L
    xu_o = osc.o.x() + osc.o.image() * L
    d = np.abs(xu_p - xu_o)
    dv = np.abs(lmp.gather("v") - osc.o.v()).max()
    print("step %d: max|dx| %.3e (bead %d)  max|dv| %.3e  builds product %d oracle %d  pairs %d / %d" % (
        m, d.max(), np.unravel_index(d.argmax(), d.shape)[0] + 1, dv, lmp.stat("neigh_builds"), osc.o.neigh_builds(),
        lmp.stat("neigh_pairs"), 2 * osc.o.neigh_pairs()), flush=True)
