"""Full-size probe of the decomposed path: the 8M-bead scrambled-start system of BASELINE configs[4] (dense LE set, firing
period 100) on ONE rank and on 8 z-slabs (in-process transport, all on the one GPU) - same commands, the bond topology must be
identical and the positions equal to summation-order rounding.  The one-rank run is the one profiles/r03/parity_8m_304_steps.log
compares with the oracle.  usage: python tests/parity_dd_8m.py [NBEADS] [STEPS] [WORLD]"""
import os, sys, tempfile, threading, time, uuid
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
os.environ.setdefault("LAMMPS_LE_RNG_W", "24")        # eight ranks' draw pools share one GPU's memory here
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, scrambled_chains, write_data
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 304
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sysd = scrambled_chains(n, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=100, left=2, right=3, tp=0.5, lr="4", nload=100, pload=0.01, punload=0.01)
print("system written", flush=True)


def bond_rows(lmp):
    nb, bt, ba = lmp.gather("num_bond"), lmp.gather("bond_type"), lmp.gather("bond_atom")
    return nb, bt, ba


t0 = time.time()
one = lammps(cmdargs=["-screen", "none"])
for ln in script.split("\n"):
    one.command(ln)
one.command("run %d" % steps)
x1 = one.gather("x"); nb1, bt1, ba1 = bond_rows(one)
c1 = [one.extract_fix(f, 0, 1, k) for f in ("loop", "loading", "unloading") for k in (0, 1)]
print("1 rank: %d steps in %.1f s, bonds %d" % (steps, time.time() - t0, one.get_thermo("bonds")), flush=True)
one.close()
session = uuid.uuid4().hex[:10]
out = [None] * world


def work(rank):
    lmp = lammps(cmdargs=["-screen", "none"])
    lmp.comm_init("local", rank, world, session=session)
    for ln in script.split("\n"):
        lmp.command(ln)
    lmp.command("run %d" % steps)
    x = lmp.gather("x"); nb, bt, ba = bond_rows(lmp)         # collective
    if rank == 0:
        out[0] = (x, nb, bt, ba, [lmp.extract_fix(f, 0, 1, k) for f in ("loop", "loading", "unloading") for k in (0, 1)], lmp.get_thermo("bonds"))
    lmp.close()


t0 = time.time()
th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
[t.start() for t in th]
[t.join() for t in th]
x8, nb8, bt8, ba8, c8, bonds8 = out[0]
print("%d ranks: %d steps in %.1f s, bonds %d" % (world, steps, time.time() - t0, bonds8), flush=True)
L = sysd["box"][0][1] - sysd["box"][0][0]
d = x8 - x1
print("bond tables equal (count, types, partners per slot): %s; extruder bonds %d; fix counters equal: %s; max|dx| %.3e" % (
    bool((nb1 == nb8).all() and (bt1 == bt8).all() and (ba1 == ba8).all()), int(((bt1 == 2) * (np.arange(bt1.shape[1])[None, :] < nb1[:, None])).sum() // 2),
    c1 == c8, np.abs((d + L / 2) % L - L / 2).max()))
