"""Long-run behaviour of the dense LE parameter set on a mid-size system: product (GPU) or oracle (CPU), printing the
extruder count, FENE warnings and temperature every 10 000 steps.  Trajectories diverge chaotically, so this is a
statistical comparison (do both jam / abort the same way?), not a parity test.
usage: soak_compare.py product|oracle [NBEADS] [BLOCKS] [PLOAD] [PUNLOAD] [lattice|walk]   (defaults: the dense set 0.01 / 0.01,
serpentine-lattice start; `walk` = the scrambled start of bench.py's default workload)"""
import os, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, scrambled_chains, write_data
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
pload = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
punload = float(sys.argv[5]) if len(sys.argv) > 5 else pload
gen = sys.argv[6] if len(sys.argv) > 6 else "lattice"
sysd = (scrambled_chains if gen == "walk" else lattice_chains)(n, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=pload, punload=punload)
t0 = time.time()
if which == "product":
    from lammps_le_amd import lammps
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in script.split("\n"):
        lmp.command(ln)
    lmp.command("thermo 10000")
    for k in range(blocks):
        try:
            lmp.command("run 10000")
        except Exception as e:
            print("product aborted in block", k + 1, ":", e); break
        print("product step %d T %.4f extruders %d fene_warn %d (%.0f s)" % ((k + 1) * 10000, lmp.get_thermo("temp"),
              lmp.get_thermo("bonds") - (n - 1), lmp.stat("fene_warnings"), time.time() - t0), flush=True)
else:
    from systems import OracleScript
    osc = OracleScript(dict(sysd))
    for ln in script.split("\n"):
        if not ln.startswith("thermo_style"):
            osc.line(ln)
    for k in range(blocks):
        try:
            osc.o.run(10000)
        except Exception as e:
            print("oracle aborted in block", k + 1, ":", e); break
        print("oracle step %d T %.4f extruders %d fene_warn %d (%.0f s)" % ((k + 1) * 10000, osc.o.thermo()[0],
              osc.o.nbonds() - (n - 1), osc.o.fene_warnings(), time.time() - t0), flush=True)
