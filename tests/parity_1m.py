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
period = int(sys.argv[3]) if len(sys.argv) > 3 else 1000        # firing period of the three LE fixes (shorter: more firings per CPU minute)
if len(sys.argv) > 4 and sys.argv[4] == "walk":
    from lammps_le_amd.synth import scrambled_chains
    sysd = scrambled_chains(n, nchains=1, seed=1, barrier_every=200)
else:
    sysd = lattice_chains(n, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=period, left=2, right=3, tp=0.5, lr="4", nload=period, pload=0.01, punload=0.01)
if len(sys.argv) > 5 and sys.argv[5] == "anchors":       # the barrier beads (every 200th) neither move nor are thermostatted:
    script = script.replace("fix 1 all nve", "group mobile type 1\nfix 1 mobile nve").replace("fix 2 all langevin", "fix 2 mobile langevin")   # k_step<.., GRP>
lmp = lammps(cmdargs=["-screen", "none"])
for ln in script.split("\n"):
    lmp.command(ln)
CHUNK = 25 if n > 2000000 else steps       # (both engines issue the same `run` commands: every `run` has a setup that draws)
t0 = time.time()
done = 0
while done < steps:
    lmp.command("run %d" % min(CHUNK, steps - done))
    done += min(CHUNK, steps - done)
print("product: %d steps in %.1f s, bonds %d" % (steps, time.time() - t0, lmp.get_thermo("bonds")), flush=True)
osc = OracleScript(dict(sysd))
for ln in script.split("\n"):
    if not ln.startswith("thermo_style"):
        osc.line(ln)
t0 = time.time()
done = 0
while done < steps:            # in chunks, with a line per chunk: a silent CPU phase of many minutes looks like a hung run
    chunk = min(CHUNK, steps - done)
    osc.o.run(chunk)
    done += chunk
    print("oracle: step %d after %.1f s" % (done, time.time() - t0), flush=True)
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
