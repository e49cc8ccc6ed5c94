"""One rank of a multi-process run of the decomposed engine (test helper, launched by test_gpu_dd.py).
usage: dd_worker.py RANK WORLD SESSION SYSTEM.npz SCRIPT.txt OUT.npz"""
import os
import pickle
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lammps_le_amd import lammps
from systems import write_data

rank, world, session, sysfile, scriptfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
system = pickle.load(open(sysfile, "rb"))
script = open(scriptfile).read()
lmp = lammps(cmdargs=["-screen", "none"])
if os.environ.get("LE_BACKEND") == "rccl":
    # experiment only: RCCL normally refuses two ranks on one device
    import time
    from lammps_le_amd import comm_unique_id
    idfile = "/dev/shm/le_id_" + session
    if rank == 0:
        open(idfile + ".tmp", "wb").write(comm_unique_id())
        os.rename(idfile + ".tmp", idfile)
    while not os.path.exists(idfile):
        time.sleep(0.01)
    lmp.comm_init("rccl", rank, world, open(idfile, "rb").read())
else:
    lmp.comm_init("shm", rank, world, session=session)
tmp = os.path.dirname(out)
for ln in script.split("\n"):
    w = ln.split("#")[0].split()
    if w and w[0] == "read_data":
        path = os.path.join(tmp, "data.r%d" % rank)
        write_data(path, system)
        ln = "read_data " + path
    lmp.command(ln)
# gathers are collective: every rank calls them
res = dict(x=lmp.gather("x"), v=lmp.gather("v"), image=lmp.gather("image"), type=lmp.gather("type"),
           num_bond=lmp.gather("num_bond"), bond_type=lmp.gather("bond_type"), bond_atom=lmp.gather("bond_atom"),
           nspecial=lmp.gather("nspecial"), special=lmp.gather("special"),
           thermo=np.array([lmp.get_thermo(k) for k in ("temp", "epair", "emol", "etotal", "press", "bonds")]),
           neigh_pairs=np.array([lmp.stat("neigh_pairs")]), builds=np.array([lmp.stat("neigh_builds")]),
           window_exchanges=np.array([lmp.stat("halo_window_exchanges")]),
           window_mismatches=np.array([lmp.stat("halo_window_mismatches")]))
for fid in ("loop", "loading", "unloading"):
    try:
        res["f_" + fid] = np.array([lmp.extract_fix(fid, 0, 1, 0), lmp.extract_fix(fid, 0, 1, 1)])
    except Exception:
        pass
if rank == 0:
    np.savez(out, **res)
lmp.close()
