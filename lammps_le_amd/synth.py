"""Synthetic bead-spring inputs for benchmarks and examples: serpentine-lattice chains at melt density
(the reference's own generator, tools/chain.f, is a Fortran random-walk tool; this one is deterministic,
overlap-free and O(N) in numpy) and a fast LAMMPS data-file writer."""
import numpy as np


def lattice_chains(nbeads, nchains=1, density=0.85, seed=1, jitter=0.03, temp=1.0, barrier_every=0):
    """nchains chains laid along one serpentine path through a simple-cubic lattice (bond = lattice
    spacing = density**(-1/3)), small seeded jitter, Maxwell velocities with zero net momentum.
    barrier_every > 0 types every barrier_every-th bead 2 (left), 3 (right), 4 (roadblock) in turn."""
    rng = np.random.RandomState(seed)
    n = int(nbeads)
    a = (1.0 / density) ** (1.0 / 3.0)
    L = int(np.ceil(n ** (1.0 / 3.0)))
    k = np.arange(n)
    iz = k // (L * L)
    yy = (k % (L * L)) // L
    col = k % L
    iy = np.where(iz % 2 == 0, yy, L - 1 - yy)
    ix = np.where((k // L) % 2 == 0, col, L - 1 - col)
    x = (np.stack([ix, iy, iz], axis=1).astype(np.float64) + 0.5) * a
    x += rng.uniform(-jitter, jitter, size=(n, 3))
    v = rng.normal(0.0, np.sqrt(temp), size=(n, 3))
    v -= v.mean(axis=0)
    per = n // nchains
    first = np.arange(n - 1)
    keep = np.ones(n - 1, dtype=bool)
    for c in range(1, nchains):
        keep[c * per - 1] = False          # no bond across a chain boundary
    a1 = first[keep] + 1
    bonds = np.stack([np.ones_like(a1), a1, a1 + 1], axis=1).astype(np.int32)
    typ = np.ones(n, dtype=np.int32)
    ntypes = 1
    if barrier_every > 0:
        idx = np.arange(barrier_every, n - barrier_every, barrier_every)
        typ[idx] = 2 + (np.arange(len(idx)) % 3)
        ntypes = 4
        ends = np.concatenate([[0], np.arange(1, nchains) * per - 1, np.arange(1, nchains) * per, [n - 1]])
        typ[ends] = 1
    mol = (np.minimum(k // per, nchains - 1) + 1).astype(np.int32)
    return dict(box=np.array([[0.0, L * a]] * 3), x=x, v=v, type=typ, mol=mol, image=np.zeros((n, 3), dtype=np.int32),
                bonds=bonds, ntypes=ntypes, nbondtypes=2, mass=[1.0] * ntypes, extra_bond=1, extra_special=20,
                atom_style="bond")


def _finish(x, n, nchains, barrier_every, box_edge, rng, temp):
    v = rng.normal(0.0, np.sqrt(temp), size=(n, 3))
    v -= v.mean(axis=0)
    per = n // nchains
    first = np.arange(n - 1)
    keep = np.ones(n - 1, dtype=bool)
    for c in range(1, nchains):
        keep[c * per - 1] = False          # no bond across a chain boundary
    a1 = first[keep] + 1
    bonds = np.stack([np.ones_like(a1), a1, a1 + 1], axis=1).astype(np.int32)
    typ = np.ones(n, dtype=np.int32)
    ntypes = 1
    k = np.arange(n)
    if barrier_every > 0:
        idx = np.arange(barrier_every, n - barrier_every, barrier_every)
        typ[idx] = 2 + (np.arange(len(idx)) % 3)
        ntypes = 4
        ends = np.concatenate([[0], np.arange(1, nchains) * per - 1, np.arange(1, nchains) * per, [n - 1]])
        typ[ends] = 1
    mol = (np.minimum(k // per, nchains - 1) + 1).astype(np.int32)
    return dict(box=np.array([[0.0, box_edge]] * 3), x=x, v=v, type=typ, mol=mol, image=np.zeros((n, 3), dtype=np.int32),
                bonds=bonds, ntypes=ntypes, nbondtypes=2, mass=[1.0] * ntypes, extra_bond=1, extra_special=20,
                atom_style="bond")


def scrambled_chains(nbeads, nchains=1, density=0.85, seed=1, jitter=0.03, temp=1.0, barrier_every=0, block=10, sweeps=6):
    """Overlap-free start whose chain order is NOT its memory order: a space-filling walk over the simple-cubic lattice
    that is a random Hamiltonian path (backbite moves) inside every block of `block`^3 sites, blocks visited in
    serpentine order (csrc/tools.cpp).  Same density, bond length and barrier typing as lattice_chains; locally the
    walk turns at 77 % of its sites (the serpentine start: 1 %), so a cell's beads carry unrelated tags the way a melt's do.
    The reference's own generator (tools/chain.f: phantom random walks) needs a soft push-off this engine has no style for."""
    import ctypes as C
    import os
    rng = np.random.RandomState(seed)
    n = int(nbeads)
    a = (1.0 / density) ** (1.0 / 3.0)
    L0 = int(np.ceil(n ** (1.0 / 3.0)))
    # smallest lattice edge that is a multiple of an even block edge near the requested one
    best = None
    for b in (block, block - 2, block + 2, block - 4):
        if b >= 4 and b % 2 == 0:
            L = ((L0 + b - 1) // b) * b
            if best is None or L < best[0]:
                best = (L, b)
    L, b = best
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblammps_le.so"))
    xyz = np.zeros((L ** 3, 3), dtype=np.int32)
    rc = lib.lammps_le_tool_scrambled_path(L, b, int(seed), int(sweeps), xyz.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError("scrambled path generator failed (%d)" % rc)
    x = (xyz[:n].astype(np.float64) + 0.5) * a
    x += rng.uniform(-jitter, jitter, size=(n, 3))
    return _finish(x, n, nchains, barrier_every, L * a, rng, temp)


def add_backbone_angles(s, nchains=1, atype=1, nangletypes=2, extra_angle=24):
    """Semiflexible chains: one angle (i, i+1, i+2) of type `atype` per backbone triple (not across a chain end); the data
    file then needs `atom_style angle | molecular`.  extra_angle: room for the angles `fix ex_load ... atype N` creates."""
    n = len(s["x"])
    per = n // nchains
    i = np.arange(1, n - 1)
    keep = (np.minimum((i - 1) // per, nchains - 1) == np.minimum((i + 1) // per, nchains - 1))
    i = i[keep]
    s["angles"] = np.stack([np.full_like(i, atype), i, i + 1, i + 2], axis=1).astype(np.int32)
    s["nangletypes"], s["extra_angle"], s["atom_style"] = nangletypes, extra_angle, "molecular"
    return s


def write_data(path, s):
    """LAMMPS data file (atom_style bond) with %.17g coordinates; pandas C writer for speed."""
    import pandas as pd
    n = len(s["x"])
    with open(path, "w") as fh:
        fh.write("synthetic bead-spring chains (lammps_le_amd.synth)\n\n")
        fh.write("%d atoms\n%d atom types\n%d bonds\n%d bond types\n" % (n, s["ntypes"], len(s["bonds"]), s["nbondtypes"]))
        if s.get("nangletypes"):
            fh.write("%d angles\n%d angle types\n" % (len(s.get("angles", [])), s["nangletypes"]))
            if s.get("extra_angle"):
                fh.write("%d extra angle per atom\n" % s["extra_angle"])
        if s.get("extra_bond"):
            fh.write("%d extra bond per atom\n" % s["extra_bond"])
        if s.get("extra_special"):
            fh.write("%d extra special per atom\n" % s["extra_special"])
        fh.write("\n")
        for d, nm in enumerate("xyz"):
            fh.write("%.17g %.17g %slo %shi\n" % (s["box"][d][0], s["box"][d][1], nm, nm))
        fh.write("\nMasses\n\n")
        for t, m in enumerate(s["mass"]):
            fh.write("%d %.17g\n" % (t + 1, m))
        fh.write("\nAtoms\n\n")
        ids = np.arange(1, n + 1)
        df = pd.DataFrame({"id": ids, "mol": s["mol"], "type": s["type"], "x": s["x"][:, 0], "y": s["x"][:, 1],
                           "z": s["x"][:, 2], "ix": s["image"][:, 0], "iy": s["image"][:, 1], "iz": s["image"][:, 2]})
        df.to_csv(fh, sep=" ", header=False, index=False, float_format="%.17g")
        fh.write("\nVelocities\n\n")
        dv = pd.DataFrame({"id": ids, "x": s["v"][:, 0], "y": s["v"][:, 1], "z": s["v"][:, 2]})
        dv.to_csv(fh, sep=" ", header=False, index=False, float_format="%.17g")
        if len(s["bonds"]):
            fh.write("\nBonds\n\n")
            b = s["bonds"]
            db = pd.DataFrame({"k": np.arange(1, len(b) + 1), "t": b[:, 0], "a": b[:, 1], "b": b[:, 2]})
            db.to_csv(fh, sep=" ", header=False, index=False)
        if s.get("nangletypes") and len(s.get("angles", [])):
            fh.write("\nAngles\n\n")
            a = s["angles"]
            da = pd.DataFrame({"k": np.arange(1, len(a) + 1), "t": a[:, 0], "a": a[:, 1], "b": a[:, 2], "c": a[:, 3]})
            da.to_csv(fh, sep=" ", header=False, index=False)


CHAIN_INPUT = """units lj
atom_style bond
newton off
atom_modify sort 0 0
special_bonds fene
read_data {data}
neighbor 0.4 bin
neigh_modify every 1 delay 10 check yes
comm_modify cutoff 5.0
bond_style fene
bond_coeff 1 30.0 1.5 1.0 1.0
bond_coeff 2 30.0 4.0 1.0 1.0
pair_style lj/cut 1.12
pair_modify shift yes
pair_coeff * * 1.0 1.0 1.12
fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion {n1} 1 {left} {right} {tp} 2 {lr}
fix loading all ex_load {nload} 1 1 1.12 2 prob {pload} 684474 iparam 1 1 jparam 1 1
fix unloading all ex_unload {nload} 2 0.5 prob {punload} 456456
timestep 0.005
thermo_style custom step temp epair emol press bonds f_loop[1] f_loading[2] f_unloading[2]
"""
