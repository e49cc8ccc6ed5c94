"""GPU parity tests of the MD core (pair lj/cut, bond fene/harmonic, fix nve, fix langevin, neighbor
rebuild schedule) — product engine through the C-ABI vs the CPU oracle on the same script + data.

Tolerances: all device arithmetic is IEEE double in the reference's operation order; only the order
in which one bead's pair/bond terms are summed differs, so single evaluations agree to ~1e-13
relative and short trajectories to ~1e-9 (chaotic growth of rounding differences)."""
import json
import os

import numpy as np
import pytest

from systems import CHAIN_SCRIPT, lattice_chain, run_oracle, run_product, write_data

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def relerr(a, b, floor=1.0):
    a, b = np.asarray(a), np.asarray(b)
    return (np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)).max()


@pytest.mark.parametrize("n,nchains", [(2000, 1), (4096, 4), (30000, 3)])
def test_single_force_evaluation(tmp_path, n, nchains):
    s = lattice_chain(n, nchains=nchains, seed=3, jitter=0.08)
    script = CHAIN_SCRIPT + "run 0\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("f"), o.f()) < 1e-12
    to = o.thermo()
    for key, idx in (("temp", 0), ("epair", 1), ("emol", 2), ("etotal", 3), ("press", 4)):
        assert abs(p.get_thermo(key) - to[idx]) <= 1e-12 * max(1.0, abs(to[idx])), key
    assert p.stat("neigh_pairs") == 2 * o.neigh_pairs()   # full list = 2 x half list


def test_hybrid_bonds_and_special_weights(tmp_path):
    """bond_style hybrid fene harmonic + special_bonds lj with fractional weights (bits in the list)."""
    s = lattice_chain(3000, seed=5, jitter=0.08)
    n = len(s["x"])
    extra = np.array([(2, i, i + 2) for i in range(10, n - 10, 37)], dtype=np.int32)
    s["bonds"] = np.concatenate([s["bonds"], extra])
    script = CHAIN_SCRIPT.replace("special_bonds fene", "special_bonds lj 0.0 0.3 0.7") \
        .replace("bond_style fene", "bond_style hybrid fene harmonic") \
        .replace("bond_coeff 1 30.0 1.5 1.0 1.0", "bond_coeff 1 fene 30.0 1.5 1.0 1.0") \
        .replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 harmonic 10.0 1.2") + "run 0\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("f"), o.f()) < 1e-12
    to = o.thermo()
    assert abs(p.get_thermo("epair") - to[1]) < 1e-12 and abs(p.get_thermo("emol") - to[2]) < 1e-12
    assert abs(p.get_thermo("press") - to[4]) < 1e-11


def test_nve_trajectory(tmp_path):
    s = lattice_chain(4000, seed=7)
    script = CHAIN_SCRIPT + "fix 1 all nve\nthermo 50\nrun 100\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9
    assert relerr(p.gather("v"), o.v()) < 1e-8
    assert (p.gather("image") == o.image()).all()
    assert p.stat("neigh_builds") == o.neigh_builds()
    assert abs(p.get_thermo("etotal") - o.thermo()[3]) < 1e-9


@pytest.mark.parametrize("segments", [0, 3, 7])
def test_langevin_stream_parity(tmp_path, segments, monkeypatch):
    """fix langevin: 3 RanMars draws per bead per call in canonical order, bit-exact stream; two runs
    (each run calls setup() again and consumes another 3N draws, src/verlet.cpp:153).  `segments`: the batch generator cuts
    every call into that many independently generated pieces (one wavefront and one jumped window each, the last one
    shorter; 0 = its own choice, one piece at this size) - the stream must not notice."""
    if segments:
        monkeypatch.setenv("LAMMPS_LE_RNG_SEGMENTS", str(segments))
    s = lattice_chain(5000, seed=11)
    script = CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 20\nrun 40\nrun 30\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9
    assert relerr(p.gather("v"), o.v()) < 1e-8
    assert abs(p.get_thermo("temp") - o.thermo()[0]) < 1e-9
    assert p.stat("neigh_builds") == o.neigh_builds()


@pytest.mark.parametrize("style,sort", [("run_style respa 2 4", 0), ("run_style respa 3 2 3 bond 1 pair 2", 0),
                                        ("run_style respa 2 2", 5)])
def test_respa_trajectory_and_thermo(tmp_path, style, sort):
    """r-RESPA (src/respa.cpp:600-741) with nve + langevin over two runs incl. rebuilds (decided at the outermost level on
    the positions the previous step left) and, in the third case, Atom::sort: trajectory, every thermo line and the
    rebuild count against the oracle."""
    s = lattice_chain(5000, seed=11)
    script = CHAIN_SCRIPT.replace("atom_modify sort 0 0", "atom_modify sort %d 0" % sort) + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 10\n" + style + "\nrun 40\nrun 25\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9
    assert relerr(p.gather("v"), o.v()) < 1e-8
    to = o.thermo()
    for k, key in enumerate(("temp", "epair", "emol", "etotal", "press")):
        assert abs(p.get_thermo(key) - to[k]) < 1e-9 * max(1.0, abs(to[k])), key
    assert p.stat("neigh_builds") == o.neigh_builds()


@pytest.mark.parametrize("n", [3000, 70000])
def test_velocity_create_between_runs(tmp_path, n):
    """`velocity all create` before the first run and again between two runs (the second call has to fetch the
    newest state from the device, replace v and upload it again); oracle driven from the same script."""
    s = lattice_chain(n, seed=13)       # (70000: the one-lane-per-bead shape of the step kernel; the second `velocity`
    s["v"] = np.zeros_like(s["v"])      #  replaces the device arrays while bins of the old ones are still around)
    script = CHAIN_SCRIPT + ("velocity all create 1.0 4928459 dist gaussian\nfix 1 all nve\nthermo 20\nrun 40\n"
                             "velocity all create 0.6 8723 loop local\nrun 30\n")
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9
    assert relerr(p.gather("v"), o.v()) < 1e-8
    assert abs(p.get_thermo("temp") - o.thermo()[0]) < 1e-9
    assert p.stat("neigh_builds") == o.neigh_builds()


@pytest.mark.parametrize("env", [
    {"LAMMPS_LE_LPB": "1", "LAMMPS_LE_AHEAD_MAX_N": "0"},           # throughput variant (what >64k-bead systems run)
    {"LAMMPS_LE_LPB": "1", "LAMMPS_LE_AHEAD_MAX_N": "1000000000"},  # loads issued ahead
    {"LAMMPS_LE_LPB": "4"},                                          # four lanes per bead
    {"LAMMPS_LE_LPB": "1", "LAMMPS_LE_AHEAD_MAX_N": "0", "LAMMPS_LE_NO_FUSED_THERMO": "1"},   # thermo steps through k_force + k_langevin
], ids=["plain", "ahead", "lpb4", "plain-unfused-thermo"])
def test_step_kernel_variants(tmp_path, env):
    """Every variant of the fused step kernel (chosen by system size in production) on the same system: hybrid bonds,
    fractional special weights, two atom types, Langevin; against the oracle.  "plain" takes its thermo steps (every 25th)
    through the energy variant of the step kernel, as systems above 64k beads do."""
    import pickle
    import subprocess
    import sys
    n = 6000
    s = lattice_chain(n, seed=17, jitter=0.05, types=1 + (np.arange(n) % 5 == 0))
    s["mass"] = [1.0, 1.7]
    extra = np.array([(2, i, i + 2) for i in range(10, n - 10, 41)], dtype=np.int32)
    s["bonds"] = np.concatenate([s["bonds"], extra])
    script = CHAIN_SCRIPT.replace("special_bonds fene", "special_bonds lj 0.0 0.4 0.8") \
        .replace("bond_style fene", "bond_style hybrid fene harmonic") \
        .replace("bond_coeff 1 30.0 1.5 1.0 1.0", "bond_coeff 1 fene 30.0 1.5 1.0 1.0") \
        .replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 harmonic 10.0 1.2") \
        .replace("pair_coeff * * 1.0 1.0 1.12", "pair_coeff * * 1.0 1.0 1.12\npair_coeff 2 2 1.3 0.9 1.05") \
        + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 5544\nthermo 25\nrun 60\n"
    o = run_oracle(script, s)
    sysfile, scriptfile, out = str(tmp_path / "sys.pkl"), str(tmp_path / "in.txt"), str(tmp_path / "out.npz")
    pickle.dump(s, open(sysfile, "wb"))
    open(scriptfile, "w").write(script)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "variant_worker.py"),
                        sysfile, scriptfile, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = np.load(out)
    assert relerr(res["x"].reshape(-1, 3), o.x()) < 1e-9
    assert relerr(res["v"].reshape(-1, 3), o.v()) < 1e-8
    assert (res["image"].reshape(-1, 3) == o.image()).all()
    assert int(res["builds"][0]) == o.neigh_builds()
    to = o.thermo()
    for k in range(5):
        assert abs(res["thermo"][k] - to[k]) <= 1e-9 * max(1.0, abs(to[k]))


def test_chain_benchmark_golden(tmp_path):
    """BASELINE configs[0]: bench/in.chain settings on bench/data.chain; the published 1-rank log's
    step-0 / step-100 thermo (reference default atom_modify sort 1000 -> Atom::sort order emulated)."""
    z = np.load(os.path.join(G, "chain32k.npz"))
    t = json.load(open(os.path.join(G, "chain32k_thermo.json")))
    n = len(z["tag"])
    s = dict(box=z["box"], x=z["x"], v=z["v"], type=z["type"], mol=z["mol"], image=z["image"], bonds=z["bonds"],
             ntypes=1, nbondtypes=1, mass=[float(z["mass"][0])], atom_style="bond")
    script = """units lj
atom_style bond
special_bonds fene
read_data data.chain
neighbor 0.4 bin
neigh_modify every 1 delay 1
bond_style fene
bond_coeff 1 30.0 1.5 1.0 1.0
pair_style lj/cut 1.12
pair_modify shift yes
pair_coeff 1 1 1.0 1.0 1.12
fix 1 all nve
fix 2 all langevin 1.0 1.0 10.0 904297
thermo 100
timestep 0.012
run 100
"""
    log = os.path.join(str(tmp_path), "log.chain")
    p = run_product(script, s, tmp_path, cmdargs=("-screen", "none", "-log", log))
    gold = t["thermo"][1]
    for key, g in zip(("temp", "epair", "emol", "etotal", "press"), gold[1:]):
        assert float("%.8g" % p.get_thermo(key)) == g, (key, p.get_thermo(key), g)        # every printed digit
    # the printed thermo block, byte for byte as in bench/log.6Oct16.chain.fixed.icc.1:47-49 (src/thermo.cpp formats)
    p.close()
    text = open(log).read()
    assert "Step Temp E_pair E_mol TotEng Press \n" in text
    assert "       0   0.97029772   0.44484087    20.494523    22.394765    4.6721833 \n" in text
    assert "     100    0.9729966    0.4361122    20.507698     22.40326    4.6548819 \n" in text
    assert "Loop time of " in text and " on 1 procs for 100 steps with 32000 atoms" in text
    p = run_product(script, s, tmp_path)
    assert p.stat("neigh_builds") == t["builds"]
    assert p.stat("neigh_pairs") == 2 * t["neighbors"]
    assert n == 32000


def test_device_ranmars_serial_stream():
    """The single-wavefront RanMars generator used by the LE fixes (33 values per dependent step)."""
    import ctypes
    from lammps_le_amd import library_path
    from oracle import ranmars_stream
    lib = ctypes.CDLL(library_path())
    fn = lib.lammps_le_test_device_ranmars
    fn.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    for seed, skip, count, ncalls in ((684474, 0, 5000, 1), (12345, 0, 100, 7), (456456, 777, 4001, 3), (12345, 0, 20, 20)):
        out = np.zeros(count)
        assert fn(seed, skip, count, ncalls, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0
        ref = ranmars_stream(seed, skip + count)[skip:]
        assert np.array_equal(out, ref), (seed, skip, count, ncalls, np.nonzero(out != ref)[0][:5])


@pytest.mark.parametrize("style,fixture", [("fene", "bond_fene.json"), ("harmonic", "bond_harmonic.json"),
                                           ("hybrid harmonic morse", "bond_hybrid.json")])
def test_fourmol_bond_known_answers(tmp_path, style, fixture):
    """The HIP path against the reference's OWN known answers (unittest/force-styles/tests/bond-fene.yaml, bond-harmonic.yaml,
    bond-hybrid.yaml on data.fourmol; fixtures made by tests/golden/make_golden.py): forces and bond energy of the initial
    state and after 4 NVE steps, through the C-ABI.  The harness' `pair_style zero 8.0` becomes `zero 2.0` (no pair forces
    either way; the engine's cell lists want a box of three neighbor cutoffs, data.fourmol's is 15 A)."""
    from lammps_le_amd import lammps
    d = json.load(open(os.path.join(G, "fourmol.json")))
    g = json.load(open(os.path.join(G, fixture)))
    tag = np.array(d["tag"])
    order = np.argsort(tag)
    n = d["natoms"]
    sysd = dict(box=np.array(d["box"]), x=np.array(d["x"])[order], type=np.array(d["type"])[order], mol=np.array(d["mol"])[order],
                image=np.array(d["image"])[order], v=np.array([d["vel"][str(t)] for t in tag[order]]),
                bonds=np.array(d["bonds"], dtype=np.int32), ntypes=d["ntypes"], nbondtypes=d["nbondtypes"],
                mass=[d["mass"][str(t + 1)] for t in range(d["ntypes"])])
    data = os.path.join(str(tmp_path), "data.fourmol")
    write_data(data, sysd)
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in ("units real", "atom_style bond", "atom_modify map array", "neigh_modify delay 2 every 2 check no", "timestep 0.1",
               "special_bonds lj %g %g %g" % tuple(d["special_lj"]), "pair_style zero 2.0", "bond_style " + style,
               "read_data " + data, "pair_coeff * *"):
        lmp.command(ln)
    for row in g["bond_coeff"]:
        lmp.command("bond_coeff %d %s" % (int(row[0]), " ".join(str(v) for v in row[1:])))
    lmp.command("thermo_modify norm no")
    lmp.command("run 0")
    assert abs(lmp.get_thermo("ebond") - g["init_energy"]) / abs(g["init_energy"]) < 5e-12
    assert relerr(lmp.gather("f"), g["init_forces"]) < 1e-11
    lmp.command("fix 1 all nve")
    lmp.command("run 4")
    assert abs(lmp.get_thermo("ebond") - g["run_energy"]) / abs(g["run_energy"]) < 5e-11
    assert relerr(lmp.gather("f"), g["run_forces"]) < 1e-10
    lmp.close()


@pytest.mark.parametrize("case", ["frozen-type", "id-stride", "langevin-all", "two-nve", "molecule+LE", "id-stride+sort", "frozen-type+respa", "frozen-type+unfused", "langevin-all+unfused"])
def test_fixes_on_groups(tmp_path, case, monkeypatch):
    """`group` (type / id ranges with stride / molecule / union / subtract) and fix nve / fix langevin on a group other than all
    (src/fix_nve.cpp:82, src/fix_langevin.cpp:661): atoms outside fix nve's group stay where they are, only the members of fix
    langevin's group draw - three draws per member and call, handed out in local order.  Against the oracle."""
    n = 6000
    types = 1 + (np.arange(n) % 7 == 0).astype(np.int32)
    s = lattice_chain(n, nchains=3, seed=23, jitter=0.04, types=types)
    s["mass"] = [1.0, 1.0]
    head = CHAIN_SCRIPT
    if case.endswith("+unfused"):        # (one fix nve + a pair style: the group variant of the fused step kernel unless switched off)
        monkeypatch.setenv("LAMMPS_LE_NO_FUSED_GROUPS", "1")
        case = case[:-len("+unfused")]
    if case.startswith("frozen-type"):   # every seventh bead is an anchor: neither integrated nor thermostatted
        body = "group mobile type 1\nfix 1 mobile nve\nfix 2 mobile langevin 1.0 1.0 1.0 5544\n"
        if case.endswith("respa"):       # (the respa variants of fix nve use the same group mask)
            body = "run_style respa 2 3 bond 1 pair 2\n" + body
    elif case.startswith("id-stride"):      # (+sort: Atom::sort every 5 steps - the members' ranks, which address their draws, follow it)
        if case.endswith("sort"):
            head = head.replace("atom_modify sort 0 0", "atom_modify sort 5 0")
        body = "group a id 1:3000:2 4000 4500:5000\ngroup b id 3001:3999\ngroup ab union a b\nfix 1 ab nve\nfix 2 ab langevin 1.0 1.2 2.0 91\n"
    elif case == "langevin-all":     # thermostat on everything, integration on a subset (forces on the others are simply unused)
        body = "group anchors type 2\ngroup mobile subtract all anchors\nfix 1 mobile nve\nfix 2 all langevin 1.0 1.0 1.0 77\n"
    elif case == "two-nve":          # two integrators on disjoint groups, one thermostat on one of them
        body = "group lo id 1:2999\ngroup hi id 3000:6000\nfix 1 lo nve\nfix 3 hi nve\nfix 2 hi langevin 0.8 0.8 1.0 313\n"
    else:                            # the first chain frozen, loop extrusion on everything
        body = ("group free molecule 2 3\nfix 1 free nve\nfix 2 free langevin 1.0 1.0 1.0 904297\n"
                "fix loop all extrusion 7 1 1 1 1.0 2\nfix loading all ex_load 5 1 1 1.12 2 prob 0.5 684474 iparam 1 1 jparam 1 1\n"
                "fix unloading all ex_unload 6 2 0.5 prob 0.3 456456\n")
        head = head.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0")
    script = head + body + "thermo 20\nrun 35\nrun 25\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9
    assert relerr(p.gather("v"), o.v()) < 1e-8
    to = o.thermo()
    for k, key in enumerate(("temp", "epair", "emol", "etotal", "press")):
        assert abs(p.get_thermo(key) - to[k]) <= 1e-9 * max(1.0, abs(to[k])), key
    assert p.stat("neigh_builds") == o.neigh_builds()
    if case.startswith("frozen-type"):
        frozen = types == 2
        assert np.array_equal(p.gather("x")[frozen], s["x"][frozen])
    if case == "molecule+LE":
        assert p.bond_set() == o.bond_set() and len([b for b in o.bond_set() if b[0] == 2]) > 3


@pytest.mark.parametrize("case", ["scale", "zero", "zero-group", "respa-zero"])
def test_langevin_keywords(tmp_path, case):
    """fix langevin `scale itype ratio` (per-type damping time, src/fix_langevin.cpp:135-141, 307-308) and `zero yes` (the mean
    random force of the group's members comes off every member, :725-729, 752-772), against the oracle; with `zero yes` the
    thermostat leaves the group's total momentum to the conservative forces."""
    n = 5000
    types = 1 + (np.arange(n) % 5 == 0).astype(np.int32)
    s = lattice_chain(n, nchains=2, seed=31, jitter=0.03, types=types)
    s["mass"] = [1.0, 2.0]
    head = CHAIN_SCRIPT
    if case == "scale":
        body = "fix 1 all nve\nfix 2 all langevin 1.0 1.2 1.0 4711 scale 2 3.5 scale 1 0.8\n"
    elif case == "zero":
        body = "fix 1 all nve\nfix 2 all langevin 1.0 1.0 2.0 4711 zero yes scale 2 2.0\n"
    elif case == "zero-group":
        body = "group heavy type 2\ngroup light subtract all heavy\nfix 1 all nve\nfix 2 light langevin 1.0 1.0 1.0 99 zero yes\n"
    else:
        body = "run_style respa 2 2\nfix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 99 zero yes tally no\n"
    script = head + body + "thermo 10\nrun 30\nrun 20\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9
    assert relerr(p.gather("v"), o.v()) < 1e-8
    to = o.thermo()
    for k, key in enumerate(("temp", "epair", "emol", "etotal", "press")):
        assert abs(p.get_thermo(key) - to[k]) <= 1e-9 * max(1.0, abs(to[k])), key
    hp, ho = p.thermo_history(), o.thermo_history()
    assert len(hp) == len(ho)
    if case == "zero":
        # pair and bond forces sum to zero, so does the zeroed random force; the drag changes the momentum only through -p/damp
        m = np.asarray(s["mass"])[types - 1]
        f = p.gather("f").reshape(n, 3)
        vhalf = p.gather("v").reshape(n, 3) - (0.5 * 0.005 / m)[:, None] * f      # the velocities post_force saw
        drag = -(m[:, None] / np.where(types == 2, 2.0, 1.0)[:, None] / 2.0 * vhalf).sum(axis=0)
        assert np.abs(f.sum(axis=0) - drag).max() < 1e-7
    assert relerr(p.gather("f"), o.f()) < 1e-8
    from lammps_le_amd import LammpsError
    for bad, msg in (("fix 9 all langevin 1.0 1.0 1.0 5 scale 3 1.0", "Illegal fix langevin command"),
                     ("fix 9 all langevin 1.0 1.0 1.0 5 zero maybe", "Illegal fix langevin command"),
                     ("fix 9 all langevin 1.0 1.0 1.0 5 gjf vhalf", "not supported"),
                     ("fix 9 all langevin 1.0 1.0 1.0 5 colour red", "Illegal fix langevin command")):
        with pytest.raises(LammpsError, match=msg):
            p.command(bad)


def test_group_command_errors(tmp_path):
    from lammps_le_amd import LammpsError
    s = lattice_chain(3000, seed=3)
    for bad, msg in (("group all type 1\n", "Cannot change the group all"), ("fix 1 nosuch nve\n", "Could not find fix group ID"),
                     ("group g region box\n", "Group region ID does not exist"), ("group g dynamic all every 10\n", "not supported"), ("group g union nosuch\n", "Group ID does not exist"),
                     ("fix loop nosuch extrusion 7 1 1 1 1.0 2\n", "Could not find fix group ID")):
        with pytest.raises(LammpsError, match=msg):
            run_product(CHAIN_SCRIPT + bad + "run 1\n", s, tmp_path)


def test_set_selects_by_group(tmp_path):
    """`set group ID type N` acts on the members of a group defined by `group` (src/set.cpp:60-65)."""
    n = 3000
    s = lattice_chain(n, nchains=3, seed=5)
    s["ntypes"], s["mass"] = 2, [1.0, 1.0]
    p = run_product(CHAIN_SCRIPT + "group mid molecule 2\ngroup tail id 2500:3000:100\ngroup both union mid tail\nset group both type 2\nfix 1 all nve\nrun 1\n", s, tmp_path)
    want = np.ones(n, dtype=np.int32)
    want[1000:2000] = 2
    want[np.arange(2499, 3000, 100)] = 2
    assert (p.gather("type") == want).all()
