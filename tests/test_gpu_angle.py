"""Semiflexible chains: angle_style harmonic | cosine in the HIP path and the angle bookkeeping of the LE
fixes - `fix ex_load ... atype N` creates angles around every new bond (fix_ex_load.cpp:855-954), fix ex_unload takes the
angles of a broken bond along (fix_ex_unload.cpp:551-582), fix extrusion leaves angles alone (fix_extrusion.cpp:924-1002).
Against the reference's own known answers (angle-harmonic.yaml, angle-cosine.yaml on data.fourmol) and against the oracle."""
import json
import os

import numpy as np
import pytest

from systems import CHAIN_SCRIPT, lattice_chain, run_oracle, run_product, write_data
from test_gpu_le import melted

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(a, b, floor=1.0):
    a, b = np.asarray(a), np.asarray(b)
    return (np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)).max()


@pytest.mark.parametrize("style,fixture", [("harmonic", "angle_harmonic.json"), ("cosine", "angle_cosine.json")])
def test_fourmol_angle_known_answers(tmp_path, style, fixture):
    """The HIP path against unittest/force-styles/tests/angle-harmonic.yaml / angle-cosine.yaml: forces and angle energy of
    data.fourmol's 30 angles, initial state and after 4 NVE steps, through the C-ABI (pair zero 2.0 instead of 8.0: no pair
    forces either way, the engine's cell lists want a box of three neighbor cutoffs)."""
    from lammps_le_amd import lammps
    d = json.load(open(os.path.join(G, "fourmol.json")))
    g = json.load(open(os.path.join(G, fixture)))
    tag = np.array(d["tag"])
    order = np.argsort(tag)
    sysd = dict(box=np.array(d["box"]), x=np.array(d["x"])[order], type=np.array(d["type"])[order], mol=np.array(d["mol"])[order],
                image=np.array(d["image"])[order], v=np.array([d["vel"][str(t)] for t in tag[order]]),
                bonds=np.array(d["bonds"], dtype=np.int32), ntypes=d["ntypes"], nbondtypes=d["nbondtypes"],
                mass=[d["mass"][str(t + 1)] for t in range(d["ntypes"])], nangletypes=d["nangletypes"],
                angles=np.array(d["angles"], dtype=np.int32))
    data = os.path.join(str(tmp_path), "data.fourmol")
    write_data(data, sysd)
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in ("units real", "atom_style molecular", "atom_modify map array", "neigh_modify delay 2 every 2 check no", "timestep 0.1",
               "special_bonds lj %g %g %g" % tuple(d["special_lj"]), "pair_style zero 2.0", "bond_style zero", "angle_style " + style,
               "read_data " + data, "pair_coeff * *", "bond_coeff *"):
        lmp.command(ln)
    for row in g["angle_coeff"]:
        lmp.command("angle_coeff %d %s" % (int(row[0]), " ".join(repr(v) for v in row[1:])))
    lmp.command("thermo_modify norm no")
    lmp.command("run 0")
    assert abs(lmp.get_thermo("emol") - g["init_energy"]) / abs(g["init_energy"]) < 5e-12
    assert relerr(lmp.gather("f"), g["init_forces"]) < 1e-11
    lmp.command("fix 1 all nve")
    lmp.command("run 4")
    assert abs(lmp.get_thermo("emol") - g["run_energy"]) / abs(g["run_energy"]) < 5e-11
    assert relerr(lmp.gather("f"), g["run_forces"]) < 1e-10
    lmp.close()


def semiflexible(n, nchains, seed, steps=400):
    s = melted(n, nchains=nchains, seed=seed, steps=steps)
    per = n // nchains
    ang = [(1, i, i + 1, i + 2) for i in range(1, n - 1) if (i - 1) // per == (i + 1) // per]
    s["nangletypes"], s["angles"], s["extra_angle"] = 2, np.array(ang, dtype=np.int32), 24
    s["atom_style"] = "molecular"
    return s


ANGLE_SCRIPT = CHAIN_SCRIPT.replace("atom_style bond", "atom_style molecular").replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0")


@pytest.mark.parametrize("style,coeffs", [("harmonic", ("angle_coeff 1 3.0 170.0", "angle_coeff 2 1.0 120.0")),
                                          ("cosine", ("angle_coeff 1 2.5", "angle_coeff 2 0.5"))])
def test_semiflexible_chain_trajectory(tmp_path, style, coeffs):
    """3k-bead semiflexible chains: forces, thermo (emol = ebond + eangle, pressure with the angle virial) of the initial
    state, then 100 steps of NVE + Langevin against the oracle."""
    s = semiflexible(3000, 3, seed=4)
    script = ANGLE_SCRIPT + "angle_style %s\n%s\n%s\nrun 0\n" % ((style,) + coeffs)
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("f"), o.f()) < 1e-11
    to = o.thermo()
    for key, idx in (("epair", 1), ("emol", 2), ("etotal", 3), ("press", 4)):
        assert abs(p.get_thermo(key) - to[idx]) <= 1e-11 * max(1.0, abs(to[idx])), key
    assert abs(o.angle_energy()) > 1.0
    more = "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 50\nrun 100\n"
    for ln in more.split("\n"):
        p.command(ln)
    o2 = run_oracle(script.replace("run 0\n", "") + more, s)
    assert relerr(p.gather("x"), o2.x()) < 1e-8
    t2 = o2.thermo()
    assert abs(p.get_thermo("emol") - t2[2]) < 1e-8 and abs(p.get_thermo("press") - t2[4]) < 1e-7
    assert p.stat("neigh_builds") == o2.neigh_builds()


def test_le_cycle_with_angles(tmp_path):
    """ex_load with `atype 2` on semiflexible chains: every new extruder bond brings angles of type 2 around it (on all three
    atoms of each), ex_unload removes the angles of the bonds it breaks, extrusion moves bonds and leaves angles alone.  Bond
    topology, the angle tables (count, order and content per atom), the angle count and the trajectory against the oracle."""
    s = semiflexible(3000, 3, seed=6)
    script = ANGLE_SCRIPT + """angle_style harmonic
angle_coeff 1 3.0 170.0
angle_coeff 2 1.0 100.0
fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion 7 1 1 1 1.0 2
fix loading all ex_load 5 1 1 1.12 2 prob 0.3 684474 iparam 1 1 jparam 1 1 atype 2
fix unloading all ex_unload 6 2 0.5 prob 0.4 456456
thermo 20
run 64
"""
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert p.bond_set() == o.bond_set()
    na, at, a1, a2, a3 = o.angle_table()
    assert (p.gather("num_angle") == na).all()
    for name, ref in (("angle_type", at), ("angle_atom1", a1), ("angle_atom2", a2), ("angle_atom3", a3)):
        got = p.gather(name)
        for i in np.nonzero(na)[0]:
            assert list(got[i, :na[i]]) == list(ref[i, :na[i]]), (name, i + 1)
    assert p.extract_setting("nangles") == o.nangles()
    assert any(a[0] == 2 for a in o.angle_set())                         # ex_load created angles ...
    assert o.fix_vector("unloading")[1] > 0 and o.fix_vector("loop")[0] >= 0   # ... and unloading broke bonds
    for fid in ("loop", "loading", "unloading"):
        assert p.extract_fix(fid, 0, 1, 0) == o.fix_vector(fid)[0] and p.extract_fix(fid, 0, 1, 1) == o.fix_vector(fid)[1]
    assert relerr(p.gather("x"), o.x()) < 1e-7
    assert p.angle_set() == o.angle_set()


def test_write_data_round_trip_with_angles(tmp_path):
    s = semiflexible(1200, 2, seed=3, steps=100)
    script = ANGLE_SCRIPT + "angle_style cosine\nangle_coeff * 1.5\nrun 0\n"
    p = run_product(script, s, tmp_path)
    out = os.path.join(str(tmp_path), "out.data")
    p.command("write_data " + out)
    from lammps_le_amd import lammps
    q = lammps(cmdargs=["-screen", "none"])
    for ln in script.split("\n"):
        w = ln.split()
        q.command("read_data " + out if w and w[0] == "read_data" else ln)
    assert q.angle_set() == p.angle_set() and q.extract_setting("nangles") == p.extract_setting("nangles") == len(s["angles"])
    assert relerr(q.gather("f"), p.gather("f")) < 1e-12
