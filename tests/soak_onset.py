"""Where do the FENE warnings of a long dense-LE run come from, and does the oracle produce the same ones?

GPU part (`soak_onset.py gpu [NBEADS] [MAXSTEPS] [PLOAD] [PUNLOAD]`): runs the bench system in chunks of CHUNK steps,
keeping the complete state (x, v, image, type, bond tables, step) of the START of the current chunk.  At the first chunk
in which `FENE bond too long` warnings appear it writes that state plus the warning counts of this and the next chunks
to gpurun_out/soak_onset.npz and stops.

`soak_onset.py both NBEADS MAXSTEPS PLOAD PUNLOAD [QUIET] [CHUNKS]` does both in one process without writing the state (an
8M-bead state does not fit the 64 MiB that travel back from the GPU box): GPU run to the first warning chunk, then the
oracle replay of CHUNKS chunks on the same box's host cores.

CPU part (`soak_onset.py replay [file]`): rebuilds the oracle from the saved state (same step number, so the LE fixes
fire at the same steps) and runs the same chunks.  Over a few hundred steps the two trajectories agree to ~1e-10, so
the oracle must report the SAME number of warnings in the same chunks if the warnings are reference behaviour and not
a product bug; it then prints the over-stretched bonds (type, tags, length, both beads' bond tables) at the end of the
first warning chunk, which shows the mechanism.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data  # noqa: E402

CHUNK = 200


def make(n, pload, punload):
    sysd = lattice_chains(n, nchains=1, seed=1, barrier_every=200)
    script = CHAIN_INPUT.format(data="{data}", n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=pload, punload=punload)
    return sysd, script


def bonds_rows(nb, bt, ba):
    n, w = bt.shape
    own = np.repeat(np.arange(1, n + 1, dtype=np.int64)[:, None], w, axis=1)
    keep = (np.arange(w)[None, :] < nb[:, None]) & (own < ba)
    return np.stack([bt[keep], own[keep], ba[keep]], axis=1).astype(np.int32)


def gpu(n, maxsteps, pload, punload, quiet_steps=None, save=True):
    from lammps_le_amd import lammps
    sysd, script = make(n, pload, punload)
    data = os.path.join(tempfile.mkdtemp(), "data")
    write_data(data, sysd)
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in script.format(data=data).split("\n"):
        lmp.command(ln)

    def state():
        return dict(x=lmp.gather("x"), v=lmp.gather("v"), image=lmp.gather("image"), type=lmp.gather("type"),
                    num_bond=lmp.gather("num_bond"), bond_type=lmp.gather("bond_type"), bond_atom=lmp.gather("bond_atom"),
                    step=int(lmp.get_thermo("step")))
    # the first 40 000 steps never warned in any run: skip the per-chunk state copies there
    quiet = min(40000, maxsteps // 2) if quiet_steps is None else quiet_steps
    lmp.command("run %d" % quiet)
    prev, warn_prev, found = state(), int(lmp.stat("fene_warnings")), None
    counts = []
    while prev["step"] < maxsteps:
        lmp.command("run %d" % CHUNK)
        wnow = int(lmp.stat("fene_warnings"))       # cumulative over the engine's life
        w, warn_prev = wnow - warn_prev, wnow
        if found is None and w > 0:
            found = prev
            print("first warnings in steps %d..%d: %d" % (prev["step"] + 1, prev["step"] + CHUNK, w), flush=True)
        if found is not None:
            counts.append(w)
            if len(counts) == 3:
                break
        else:
            prev = state()
        if prev["step"] % 10000 == 0:
            print("step %d extruders %d, no warnings yet" % (prev["step"], lmp.get_thermo("bonds") - (n - 1)), flush=True)
    if found is None:
        print("no FENE warnings up to step %d" % prev["step"])
        return None
    lmp.close()
    if not save:
        return dict(n=n, pload=pload, punload=punload, counts=np.array(counts), **found)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out = os.path.join(ROOT, "gpurun_out", "soak_onset.npz")
    np.savez_compressed(out, n=n, pload=pload, punload=punload, counts=np.array(counts), **found)
    print("saved", out, "counts per chunk", counts)


def replay(path, nchunks=None):
    from systems import OracleScript
    d = path if isinstance(path, dict) else np.load(path)
    n, pload, punload = int(d["n"]), float(d["pload"]), float(d["punload"])
    sysd, script = make(n, pload, punload)
    s2 = dict(sysd)
    s2["x"], s2["v"], s2["image"], s2["type"] = d["x"], d["v"], d["image"], d["type"].astype(np.int32)
    s2["bonds"] = bonds_rows(d["num_bond"], d["bond_type"], d["bond_atom"])
    osc = OracleScript(s2)
    for ln in script.format(data="x").split("\n"):
        if not ln.startswith("thermo_style"):
            osc.line(ln)
    step = int(d["step"])
    osc.line("reset_timestep %d" % step)
    o = osc.o
    want = list(d["counts"])[:nchunks]
    print("state of step %d, %d beads, %d extruders; product warnings per %d-step chunk: %s" %
          (step, n, len(s2["bonds"]) - (n - 1), CHUNK, want))
    got = []
    for k in range(len(want)):
        before = o.fene_warnings()
        try:
            o.run(CHUNK)
        except RuntimeError as e:
            print("oracle aborted in chunk %d: %s" % (k, e))
            break
        got.append(int(o.fene_warnings() - before))
        if k == 0:
            x, L = o.x(), sysd["box"][0][1]
            nb, bt, ba = o.bond_table()
            rows = bonds_rows(nb, bt, ba)
            dd = x[rows[:, 1] - 1] - x[rows[:, 2] - 1]
            dd -= L * np.round(dd / L)
            r = np.sqrt((dd ** 2).sum(1))
            r0 = np.where(rows[:, 0] == 1, 1.5, 4.0)
            bad = np.nonzero(1.0 - (r / r0) ** 2 < 0.15)[0]
            print("bonds near the FENE limit at step %d (rlogarg < 0.15):" % (step + CHUNK))
            for b in bad[:20]:
                t, a, c = rows[b]
                print("  type %d  %d-%d  r = %.3f (R0 %.1f)   bonds of %d: %s   bonds of %d: %s" %
                      (t, a, c, r[b], r0[b], a, list(zip(bt[a - 1, :nb[a - 1]], ba[a - 1, :nb[a - 1]])), c,
                       list(zip(bt[c - 1, :nb[c - 1]], ba[c - 1, :nb[c - 1]]))))
    print("oracle warnings per chunk:", got, "->", "SAME as the product" if got == want[:len(got)] else "DIFFERENT")


if __name__ == "__main__":
    if sys.argv[1] == "both":
        st = gpu(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]),
                 quiet_steps=int(sys.argv[6]) if len(sys.argv) > 6 else None, save=False)
        if st is not None:
            replay(st, nchunks=int(sys.argv[7]) if len(sys.argv) > 7 else 1)
    elif sys.argv[1] == "gpu":
        gpu(int(sys.argv[2]) if len(sys.argv) > 2 else 250000, int(sys.argv[3]) if len(sys.argv) > 3 else 120000,
            float(sys.argv[4]) if len(sys.argv) > 4 else 0.01, float(sys.argv[5]) if len(sys.argv) > 5 else 0.01)
    else:
        replay(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "soak_onset.npz"))
