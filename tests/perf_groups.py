"""Development probe: the default 1M-bead workload (scrambled start + the three LE fixes) with fix nve / fix langevin on a group
- every 200th bead (the barrier beads, types 2-4) an anchor that neither moves nor is thermostatted:

  python tests/perf_groups.py [NBEADS] [STEPS]

Prints one JSON line with the rate; LAMMPS_LE_NO_FUSED_GROUPS=1 gives the unfused kernels for comparison."""
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, scrambled_chains, write_data

nbeads = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
sysd = scrambled_chains(nbeads, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(prefix="le_grp_"), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.002, punload=0.05)
script = script.replace("fix 1 all nve", "group mobile type 1\nfix 1 mobile nve").replace("fix 2 all langevin", "fix 2 mobile langevin")
os.environ["LAMMPS_LE_KERNEL_TIMING"] = "1"
lmp = lammps(cmdargs=["-screen", "none"])
for ln in script.split("\n"):
    lmp.command(ln)
if len(sys.argv) > 3 and sys.argv[3] == "nole":          # anchors only: no loop extrusion pulling on pinned beads
    for fid in ("loop", "loading", "unloading"):
        lmp.command("unfix " + fid)
    lmp.command("thermo_style one")
lmp.command("run 3010")
lmp.command("run 500")
try:
    lmp.command("run %d" % steps)
except Exception as e:                                   # (an extruder bond stretched between pinned beads ends the run, as in the reference)
    print(json.dumps(dict(error=str(e), step=int(lmp.get_thermo("step")), bonds=int(lmp.get_thermo("bonds")))))
    sys.exit(1)
loop = lmp.stat("loop_time")
print(json.dumps(dict(beads=nbeads, steps=steps, fused=os.environ.get("LAMMPS_LE_NO_FUSED_GROUPS") is None,
                      timesteps_per_s=round(steps / loop, 1), us_per_step=round(1e6 * loop / steps, 2),
                      k_step_us=round(1e3 * lmp.stat("pair_kernel_ms"), 2), builds=int(lmp.stat("neigh_builds")),
                      temp_all_atoms=round(lmp.get_thermo("temp"), 4), mobile_fraction=round(float((sysd["type"] == 1).mean()), 4),
                      bonds=int(lmp.get_thermo("bonds")), fene_warnings=int(lmp.stat("fene_warnings")))))
lmp.close()
