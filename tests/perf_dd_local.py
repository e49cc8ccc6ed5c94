"""Development probe: the decomposed step loop with all ranks as threads of one process on ONE GPU (in-process
transport: a message is a stream-ordered device-to-device copy, i.e. RCCL's semantics without a second device).  The GPU
is shared, so the rate is not a multi-GPU number; what it shows is the per-step and per-rebuild overhead of the
decomposition (launches, synchronisations, idle gaps) next to the 1-rank loop, and what a firing of the three LE fixes
costs and moves between ranks:

  python tests/perf_dd_local.py WORLD [NBEADS] [STEPS] [lattice|walk]

LAMMPS_LE_DD_FULL_GATHER=1 selects the round-2 firing path (every bead's (tag, x, xhold) all-gathered) for comparison.
Prints one JSON line."""
import json
import os
import sys
import tempfile
import threading
import time
import uuid

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, scrambled_chains, write_data

world = int(sys.argv[1])
nbeads = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
gen = sys.argv[4] if len(sys.argv) > 4 else "lattice"
sysd = (scrambled_chains if gen == "walk" else lattice_chains)(nbeads, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(prefix="le_ddl_"), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.01, punload=0.01)
session = uuid.uuid4().hex[:10]
out = [None] * world
bar = threading.Barrier(world)


def work(rank):
    lmp = lammps(cmdargs=["-screen", "none"])
    if world > 1:
        lmp.comm_init("local", rank, world, session=session)
    for ln in script.split("\n"):
        lmp.command(ln)
    lmp.command("run 1010")                 # past the first firing of every LE fix: extruders on the chain
    bar.wait()
    t0 = time.perf_counter()
    lmp.command("run %d" % steps)           # (holds the firings of the periods it crosses)
    wall = time.perf_counter() - t0
    res = dict(rank=rank, nlocal=int(lmp.stat("nlocal")), nghost=int(lmp.stat("nghost")), builds=int(lmp.stat("neigh_builds")),
               loop_s=lmp.stat("loop_time"), wall_s=wall)
    # one firing period on its own: steps ...001 - ...010 against ten ordinary steps
    now = int(lmp.get_thermo("step"))
    to_boundary = (1000 - now % 1000) % 1000
    if to_boundary:
        lmp.command("run %d" % to_boundary)
    g0, r0 = lmp.stat("comm_bytes_allgather"), lmp.stat("comm_bytes_allreduce")
    bar.wait()
    lmp.command("run 10")
    fire = lmp.stat("loop_time")
    g1, r1 = lmp.stat("comm_bytes_allgather"), lmp.stat("comm_bytes_allreduce")
    lmp.command("run 30")
    bar.wait()
    lmp.command("run 10")
    plain = lmp.stat("loop_time")
    res.update(firing_ms=1e3 * (fire - plain), ten_firing_steps_ms=1e3 * fire, ten_plain_steps_ms=1e3 * plain,
               firing_bytes_allgather=g1 - g0, firing_bytes_allreduce=r1 - r0,
               extruders=int(lmp.get_thermo("bonds")) - (nbeads - 1))
    out[rank] = res
    bar.wait()
    lmp.close()


th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
[t.start() for t in th]
[t.join() for t in th]
loop = max(o["loop_s"] for o in out)
print(json.dumps(dict(world=world, beads=nbeads, start=gen, steps=steps, us_per_step=round(1e6 * loop / steps, 2),
                      firing_path="whole-system gather" if os.environ.get("LAMMPS_LE_DD_FULL_GATHER") else "owner bits + extruder rows",
                      firing_ms=round(max(o["firing_ms"] for o in out), 3),
                      firing_bytes_allgather_per_rank=out[0]["firing_bytes_allgather"],
                      firing_bytes_allreduce_per_rank=out[0]["firing_bytes_allreduce"], per_rank=out)))
