"""Development probe: the decomposed step loop with all ranks as threads of one process on ONE GPU (in-process
transport).  The GPU is shared, so the rate is not a multi-GPU number; what it shows is the per-step and
per-rebuild overhead of the decomposition (launches, synchronisations, idle gaps) next to the 1-rank loop.
usage: python tests/perf_dd_local.py WORLD [NBEADS] [STEPS]"""
import os
import sys
import tempfile
import threading
import time
import uuid

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data

world = int(sys.argv[1])
nbeads = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
sysd = lattice_chains(nbeads, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(prefix="le_ddl_"), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.01, punload=0.01)
session = uuid.uuid4().hex[:10]
times, stats = [0.0] * world, [None] * world
bar = threading.Barrier(world)


def work(rank):
    lmp = lammps(cmdargs=["-screen", "none"])
    if world > 1:
        lmp.comm_init("local", rank, world, session=session)
    for ln in script.split("\n"):
        lmp.command(ln)
    lmp.command("run 300")
    bar.wait()
    t0 = time.perf_counter()
    lmp.command("run %d" % steps)
    times[rank] = time.perf_counter() - t0
    stats[rank] = (lmp.stat("nlocal"), lmp.stat("neigh_builds"), lmp.stat("loop_time"))
    bar.wait()
    lmp.close()


th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
[t.start() for t in th]
[t.join() for t in th]
print("world %d beads %d: %.1f steps/s (%.1f us/step), per-rank (nlocal, builds, loop s): %s"
      % (world, nbeads, steps / max(times), 1e6 * max(times) / steps, stats))
