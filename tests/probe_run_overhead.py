import os, sys, tempfile, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lammps_le_amd import lammps
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data
n = 1000000
sysd = lattice_chains(n, nchains=1, seed=1, barrier_every=200)
data = os.path.join(tempfile.mkdtemp(), "data")
write_data(data, sysd)
script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.01, punload=0.01)
lmp = lammps(cmdargs=["-screen", "none"])
for ln in script.split("\n"):
    lmp.command(ln)
lmp.command("run 100")
for k in (0, 0, 1, 10, 100, 1000):
    t0 = time.perf_counter(); lmp.command("run %d" % k); t1 = time.perf_counter()
    print("run %4d: wall %.2f ms, engine loop %.2f ms -> setup+teardown %.2f ms" % (k, 1e3 * (t1 - t0), 1e3 * lmp.stat("loop_time"), 1e3 * (t1 - t0 - lmp.stat("loop_time"))))
x = lmp.gather("x")
t0 = time.perf_counter(); lmp.command("run 100"); t1 = time.perf_counter()
print("after gather: run 100: wall %.2f ms, engine loop %.2f ms -> outside the loop %.2f ms" % (1e3 * (t1 - t0), 1e3 * lmp.stat("loop_time"), 1e3 * (t1 - t0 - lmp.stat("loop_time"))))
import torch
t0 = time.perf_counter(); torch.cuda.synchronize(); lmp.command("run 100"); torch.cuda.synchronize(); t1 = time.perf_counter()
print("with torch syncs: run 100: wall %.2f ms, engine loop %.2f ms" % (1e3 * (t1 - t0), 1e3 * lmp.stat("loop_time")))
