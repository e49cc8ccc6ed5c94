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


def test_thermo_keywords_and_multi_style(tmp_path):
    """`thermo_style custom` with the keyword set of src/thermo.cpp:716-880 that has a meaning for this model (ebond / eangle /
    emol apart, enthalpy, density, box, time across a `timestep` change, elapsed, nbuild, cpu ..) and `thermo_style multi`
    (`---------------- Step ... CPU = ...` + `%-8s = %14.4f`, three values per line: src/thermo.cpp:171-180, 239-266),
    read back from the log the engine writes; energies against the oracle."""
    import re
    from lammps_le_amd import LammpsError
    n = 3000
    s = semiflexible(n, 3, seed=8)
    s["mass"] = [1.5] * s["ntypes"]
    body = "angle_style harmonic\nangle_coeff 1 3.0 170.0\nangle_coeff 2 1.0 100.0\nfix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n"
    log = str(tmp_path / "log.thermo")
    kws = ("step elapsed dt time cpu atoms temp press pe ke etotal enthalpy evdwl ecoul epair ebond eangle edihed eimp emol elong "
           "etail vol density lx ly lz xlo xhi ylo yhi zlo zhi bonds angles nbuild ndanger pxx pyy pzz pxy pxz pyz")
    script = (ANGLE_SCRIPT + body + "log %s\nthermo 10\nthermo_style custom %s\nrun 20\ntimestep 0.002\nrun 10\n"
              "thermo_style multi\nrun 10\n" % (log, kws))
    o = run_oracle(ANGLE_SCRIPT + body + "thermo 10\nrun 20\ntimestep 0.002\nrun 10\nrun 10\n", s)
    p = run_product(script, s, tmp_path)
    p.command("log none")
    text = open(log).read()
    head = ("Step Elapsed Dt Time CPU Atoms Temp Press PotEng KinEng TotEng Enthalpy E_vdwl E_coul E_pair E_bond E_angle E_dihed "
            "E_impro E_mol E_long E_tail Volume Density Lx Ly Lz Xlo Xhi Ylo Yhi Zlo Zhi Bonds Angles Nbuild Ndanger Pxx Pyy Pzz Pxy Pxz Pyz")
    assert text.count(head) == 2
    names = kws.split()
    rows = [dict(zip(names, [float(v) for v in ln.split()])) for ln in text.split("\n")
            if len(ln.split()) == len(names) and re.match(r"^\s*\d+ ", ln)]
    assert [r["step"] for r in rows] == [0, 10, 20, 20, 30]
    assert [r["elapsed"] for r in rows] == [0, 10, 20, 0, 10]
    L = float(s["box"][0][1] - s["box"][0][0])
    ho = o.thermo_history()            # rows (step, temp, epair, emol, etotal, press): three runs -> 0 10 20 | 20 30 | 30 40
    for r, ref in zip(rows, ho[:5]):
        assert r["step"] == ref[0]
        for key, col in (("temp", 1), ("epair", 2), ("emol", 3), ("etotal", 4), ("press", 5)):
            assert abs(r[key] - ref[col]) <= 2e-7 * max(1.0, abs(ref[col])), (key, r["step"])      # (%12.8g)
        assert abs(r["ebond"] + r["eangle"] - r["emol"]) < 5e-6 and r["eangle"] > 0.0 and r["evdwl"] == r["epair"]
        assert abs(r["pe"] - (r["epair"] + r["emol"])) < 5e-6 and abs(r["etotal"] - (r["pe"] + r["ke"])) < 5e-6    # (8 digits each)
        assert abs(r["enthalpy"] - (r["etotal"] + r["press"] * L ** 3 / n)) < 5e-5
        assert abs((r["pxx"] + r["pyy"] + r["pzz"]) / 3.0 - r["press"]) < 5e-6 * max(1.0, abs(r["press"]))
        assert r["ecoul"] == r["elong"] == r["etail"] == r["edihed"] == r["eimp"] == 0.0
        assert r["atoms"] == n and r["bonds"] == len(s["bonds"]) and r["angles"] == len(s["angles"])
        assert abs(r["vol"] - L ** 3) < 1e-3 and abs(r["density"] - 1.5 * n / L ** 3) < 1e-7 and abs(r["lx"] - L) < 1e-6
        assert abs(r["xhi"] - r["xlo"] - L) < 1e-6 and r["dt"] == (0.005 if r is not rows[3] and r is not rows[4] else 0.002)
    assert [round(r["time"], 9) for r in rows] == [0.0, 0.05, 0.1, 0.1, 0.12]
    assert rows[0]["cpu"] == 0.0 and rows[3]["cpu"] == 0.0 and rows[2]["cpu"] > rows[1]["cpu"] > 0.0
    assert rows[2]["nbuild"] >= rows[1]["nbuild"] >= rows[0]["nbuild"] == 0
    assert abs(p.get_thermo("eangle") * 1.0 - o.angle_energy() / n) <= 1e-9 * max(1.0, abs(o.angle_energy() / n))
    # the pressure tensor of the last step: (sum m v_i v_j + W_ij) / V with the oracle's velocities and virials
    v = o.v()
    ke6 = 1.5 * np.array([(v[:, a] * v[:, b]).sum() for a, b in ((0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2))])
    w6 = np.asarray(o.pair_virial()) + np.asarray(o.bond_virial()) + np.asarray(o.angle_virial())
    for k, key in enumerate(("pxx", "pyy", "pzz", "pxy", "pxz", "pyz")):
        ref = (ke6[k] + w6[k]) / L ** 3
        assert abs(p.get_thermo(key) - ref) <= 1e-8 * max(1.0, abs(ref)), key
    # the multi style of the third run
    blocks = re.findall(r"-{16} Step +(\d+) -{5} CPU = +([0-9.]+) \(sec\) -{16}\n((?:.+\n){4})", text)
    assert [int(b[0]) for b in blocks] == [30, 40] and float(blocks[0][1]) == 0.0
    first = blocks[1][2].split("\n")
    assert re.match(r"^TotEng   = +-?\d+\.\d{4} KinEng   = +-?\d+\.\d{4} Temp     = +-?\d+\.\d{4} $", first[0])
    assert first[1].startswith("PotEng   = ") and "E_bond   = " in first[1] and "E_angle  = " in first[1]
    assert first[3].startswith("E_coul   = ") and "E_long   = " in first[3] and "Press    = " in first[3]
    vals = dict(re.findall(r"(\S+) += +(-?\d+\.\d+)", blocks[1][2]))
    last = ho[-1]
    assert abs(float(vals["Temp"]) - last[1]) < 1e-4 and abs(float(vals["TotEng"]) - last[4]) < 1e-4
    with pytest.raises(LammpsError, match="Unknown keyword in thermo_style custom command"):
        p.command("thermo_style custom step colour")
    p.command("thermo_style custom step temp")
    p.command("thermo_modify line multi")


@pytest.mark.parametrize("style", ["run_style respa 2 3", "run_style respa 3 2 2 bond 1 angle 2 pair 3",
                                   "run_style respa 2 2 bond 1 angle 2 pair 2"])
def test_semiflexible_chains_under_respa(tmp_path, style):
    """Angles in r-RESPA runs (src/respa.cpp:85-88, 172, 707-710): evaluated at their level (default: the bonds') behind the
    level's bonds; trajectory, thermo lines and - with ex_load `atype` - the angle bookkeeping against the oracle."""
    s = semiflexible(3000, 3, seed=9)
    script = ANGLE_SCRIPT + """angle_style harmonic
angle_coeff 1 3.0 170.0
angle_coeff 2 1.0 100.0
fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion 7 1 1 1 1.0 2
fix loading all ex_load 5 1 1 1.12 2 prob 0.3 684474 iparam 1 1 jparam 1 1 atype 2
fix unloading all ex_unload 6 2 0.5 prob 0.4 456456
thermo 10
""" + style + "\nrun 40\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert p.bond_set() == o.bond_set() and p.angle_set() == o.angle_set()
    assert relerr(p.gather("x"), o.x()) < 1e-8 and relerr(p.gather("v"), o.v()) < 1e-7
    hp, ho = p.thermo_history(), o.thermo_history()
    assert len(hp) == len(ho) == 5
    for rp, ro in zip(hp, ho):
        assert rp[0] == ro[0]
        for k in range(1, 6):
            assert abs(rp[k] - ro[k]) <= 1e-7 * max(1.0, abs(ro[k])), (int(rp[0]), k)
    from lammps_le_amd import LammpsError
    with pytest.raises(LammpsError, match="Invalid order of forces within respa levels"):
        p.command("run_style respa 2 2 bond 2 angle 1")


def test_semiflexible_chains_with_anchors(tmp_path):
    """Semiflexible chains (angle cosine) with anchored beads - fix nve and fix langevin on `group mobile` - at a size that takes
    the throughput shape of the step kernel: its group + angle instantiation (bead's group bits by tag, listed angles evaluated
    in the kernel) against the oracle, thermo steps through the unfused kernels."""
    n = 70000
    s = semiflexible(n, 2, seed=12, steps=0)
    types = 1 + (np.arange(n) % 50 == 0).astype(np.int32)
    s["type"], s["ntypes"], s["mass"] = types, 2, [1.0, 1.0]
    script = ANGLE_SCRIPT + ("angle_style cosine\nangle_coeff * 2.0\ngroup mobile type 1\nfix 1 mobile nve\n"
                             "fix 2 mobile langevin 1.0 1.0 1.0 4711\nthermo 25\nrun 50\n")
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9 and relerr(p.gather("v"), o.v()) < 1e-8
    frozen = types == 2
    assert np.array_equal(p.gather("x").reshape(n, 3)[frozen], o.x()[frozen])
    hp, ho = p.thermo_history(), o.thermo_history()
    assert len(hp) == len(ho) == 3
    for rp, ro in zip(hp, ho):
        for k in range(1, 6):
            assert abs(rp[k] - ro[k]) <= 1e-8 * max(1.0, abs(ro[k])), (int(rp[0]), k)
