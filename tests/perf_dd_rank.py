"""Development probe: ONE rank of a decomposed run as a process of its own (file-mailbox transport, every rank on the one
GPU of the box), so that each rank can sit under its own `rocprofv3 --kernel-trace --stats` and a rank's kernels per step can
be priced without the in-process transport (whose rank threads crash under the profiler, profiles/r03/crash_*.txt):

  python tests/perf_dd_rank.py RANK WORLD SESSION [NBEADS] [STEPS] [lattice|walk]

The GPU is shared and the mailbox stages through the host, so the RATE means nothing; the kernel list of one rank is what a
rank of a real multi-GPU run launches per step and per rebuild.  Prints one JSON line."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, scrambled_chains, write_data

rank, world, session = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nbeads = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
gen = sys.argv[6] if len(sys.argv) > 6 else "walk"
sysd = (scrambled_chains if gen == "walk" else lattice_chains)(nbeads, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(prefix="le_ddr_"), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.002, punload=0.05)
lmp = lammps(cmdargs=["-screen", "none"])
if world > 1:
    lmp.comm_init("shm", rank, world, session=session)
for ln in script.split("\n"):
    lmp.command(ln)
lmp.command("run 1010")
b0 = int(lmp.stat("neigh_builds"))
lmp.command("run %d" % steps)
print(json.dumps(dict(rank=rank, world=world, beads=nbeads, steps=steps, nlocal=int(lmp.stat("nlocal")), nghost=int(lmp.stat("nghost")),
                      builds=int(lmp.stat("neigh_builds")), us_per_step=round(1e6 * lmp.stat("loop_time") / steps, 2),
                      window_exchanges=int(lmp.stat("halo_window_exchanges")))))
lmp.close()
