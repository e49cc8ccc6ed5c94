"""Third randomised parity sweep.
(a) the mixed LE scenarios of test_gpu_fuzz2.py that a decomposed run accepts (local order by `atom_modify sort N`, `newton on
    off`, second fix instances, type conversion, cut runs) on 2 or 3 z-slabs (in-process transport) against the ONE-rank oracle:
    the replicated topology must come out bit for bit;
(b) plain MD with random force-field and neighbor settings (types with their own epsilon / sigma / cutoff, mixing, shift,
    fractional special weights, fene / harmonic / hybrid bonds, masses, skin, every / delay / check, timestep, Langevin or NVE,
    thermo cadence, cut runs): trajectory, thermo and reneighbor count against the oracle.
LE_FUZZ3_SEEDS="start:stop" / LE_FUZZ3_MD_SEEDS runs a one-off wider sweep (scripts/r03_fuzz_wide.sh)."""
import os

import numpy as np
import pytest

from systems import CHAIN_SCRIPT, lattice_chain, run_oracle, run_product
from test_gpu_le import barrier_types, melted

pytestmark = pytest.mark.gpu


def _seeds(env, n):
    v = os.environ.get(env)
    if not v:
        return list(range(n))
    a, b = v.split(":")
    return range(int(a), int(b))


def dd_scenario(seed):
    rng = np.random.RandomState(9000 + seed)
    world = int(rng.choice([2, 2, 3]))
    n = int(rng.choice([9000, 11000])) if world == 2 else 27000     # slabs must be two 5.0 ghost shells thick
    if os.environ.get("LE_FUZZ3_WORLD"):                            # one-off sweeps on more slabs
        world = int(os.environ["LE_FUZZ3_WORLD"])
        n = {2: 9000, 3: 27000, 4: 58000, 5: 110000, 6: 190000}[world]
    nchains = int(rng.choice([1, 3]))
    frac = float(rng.choice([0.0, 0.15, 0.4]))
    types = barrier_types(n, 190 + seed, frac=frac) if frac > 0 else np.ones(n, dtype=np.int32)
    s = melted(n, nchains=nchains, seed=1 + seed % 2, steps=400, types=types)
    s["ntypes"], s["mass"] = 4, [1.0] * 4
    flavour = str(rng.choice(["plain", "sort", "newton", "sort+newton", "convert"]))
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 8.0 5.0 1.0 1.0")
    rng3 = np.random.RandomState(31000 + seed)      # (drawn apart: the scenarios of the first decomposed sweep keep their seeds)
    angle_lines, atype = "", ""
    if rng3.rand() < 0.3:                            # semiflexible chains: angles in a decomposed run
        per = n // nchains
        ang = [(1, i, i + 1, i + 2) for i in range(1, n - 1) if (i - 1) // per == (i + 1) // per]
        s["nangletypes"], s["angles"], s["extra_angle"] = 2, np.array(ang, dtype=np.int32), 24
        s["atom_style"] = "molecular"
        base = base.replace("atom_style bond", "atom_style molecular")
        if rng3.rand() < 0.5:
            angle_lines = "angle_style harmonic\nangle_coeff 1 %.2f %.1f\nangle_coeff 2 %.2f %.1f\n" % (
                rng3.uniform(0.5, 4.0), rng3.uniform(120.0, 180.0), rng3.uniform(0.5, 2.0), rng3.uniform(90.0, 150.0))
        else:
            angle_lines = "angle_style cosine\nangle_coeff 1 %.2f\nangle_coeff 2 %.2f\n" % (rng3.uniform(0.5, 3.0), rng3.uniform(0.2, 1.5))
        atype = str(rng3.choice(["", " atype 1", " atype 2"]))
        flavour += "+angles"
    if "sort" in flavour:
        base = base.replace("atom_modify sort 0 0", "atom_modify sort %d 0" % int(rng.randint(3, 16)))
    if "newton" in flavour:
        base = base.replace("newton off", "newton on off")
    n1, nl, nu = int(rng.randint(3, 10)), int(rng.randint(3, 10)), int(rng.randint(3, 10))
    tp = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
    lp, up = float(rng.choice([0.2, 0.6, 1.0])), float(rng.choice([0.1, 0.5, 1.0]))
    lprob = "" if lp >= 1.0 else "prob %g %d" % (lp, 100 + seed)
    uprob = "" if up >= 1.0 else "prob %g %d" % (up, 200 + seed)
    rmax = float(rng.choice([0.5, 1.3, 2.0]))
    lr = "4" if rng.rand() < 0.7 else ""
    nt = int(rng.choice([1, 2, 3, 4])) if flavour == "convert" else 1
    fixes = ["fix 1 all nve", "fix 2 all langevin 1.0 1.0 1.0 %d" % int(rng.randint(1, 900000)),
             "fix loop all extrusion %d 1 2 3 %g 2 %s" % (n1, tp, lr),
             "fix loading all ex_load %d 1 1 1.12 2 %s iparam 1 %d jparam 1 %d%s" % (nl, lprob, nt, nt, atype),
             "fix unloading all ex_unload %d 2 %g %s" % (nu, rmax, uprob)]
    if rng3.rand() < 0.25:                           # the LE fixes on a group (masks replicated by tag, like the topology)
        lo = int(rng3.randint(1, n // 2))
        fixes = ["group g id %d:%d" % (lo, int(rng3.randint(lo + n // 4, n + 1)))] + [f.replace(" all ", " g ", 1) if k >= 2 else f for k, f in enumerate(fixes)]
        flavour += "+group"
        if rng3.rand() < 0.5:                                # the integrator and the thermostat on the group too (beads outside
            fixes = [f.replace(" all ", " g ", 1) for f in fixes]   # it stay put)
            flavour += "+mdgroup"
    total = int(rng.randint(25, 60))
    cuts = sorted(set(int(c) for c in rng.randint(1, total, size=int(rng.randint(0, 3)))))
    runs, last = [], 0
    for c in cuts + [total]:
        runs.append(c - last)
        last = c
    respa = ""
    if rng3.rand() < 0.15:                           # r-RESPA across slabs (with angles: at the bonds' level or their own)
        respa = str(rng3.choice(["run_style respa 2 3", "run_style respa 3 2 2 bond 1 pair 3", "run_style respa 2 2 bond 1 pair 2"])) + "\n"
        if "angles" in flavour and rng3.rand() < 0.5:
            respa = "run_style respa 3 2 2 bond 1 angle 2 pair 3\n"
        flavour += "+respa"
    script = base + angle_lines + "\n".join(fixes) + "\nthermo 10\n" + respa + "".join("run %d\n" % r for r in runs if r > 0)
    return s, script, world, flavour


@pytest.mark.parametrize("seed", _seeds("LE_FUZZ3_SEEDS", 10))
def test_random_le_scenarios_mixed_decomposed(tmp_path, seed):
    from test_gpu_dd import bond_set, run_ranks_local
    s, script, world, flavour = dd_scenario(seed)
    try:
        o = run_oracle(script, s)
    except RuntimeError:
        with pytest.raises(Exception):
            run_ranks_local(world, s, script, tmp_path)
        return
    r = run_ranks_local(world, s, script, tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set(), flavour
    nso, spo = o.special_table()
    assert (r["nspecial"] == nso).all(), flavour
    for t in np.nonzero(nso[:, 2])[0]:
        assert list(r["special"][t, :nso[t, 2]]) == list(spo[t, :nso[t, 2]]), t + 1
    for fid in ("loop", "loading", "unloading"):
        assert r["f_" + fid][0] == o.fix_vector(fid)[0] and r["f_" + fid][1] == o.fix_vector(fid)[1], (fid, flavour)
    assert np.abs(r["x"] - o.x()).max() < 1e-6, flavour
    assert r["builds"][0] == o.neigh_builds()
    if "angles" in flavour:
        na, at, a1, a2, a3 = o.angle_table()
        assert (r["num_angle"] == na).all()
        for name, ref in (("angle_type", at), ("angle_atom1", a1), ("angle_atom2", a2), ("angle_atom3", a3)):
            for i in np.nonzero(na)[0]:
                assert list(r[name][i, :na[i]]) == list(ref[i, :na[i]]), (name, i + 1)
        assert int(r["nangles"][0]) == o.nangles()


def md_scenario(seed):
    rng = np.random.RandomState(11000 + seed)
    n = int(rng.choice([1000, 2200, 4096, 6000]))
    ntypes = int(rng.choice([1, 2, 3]))
    types = rng.randint(1, ntypes + 1, size=n).astype(np.int32)
    nchains = int(rng.choice([1, 2, 5]))
    s = lattice_chain(n, nchains=nchains, seed=30 + seed, jitter=float(rng.uniform(0.0, 0.08)), temp=float(rng.uniform(0.5, 1.5)), types=types)
    s["ntypes"] = ntypes
    s["mass"] = [float(rng.choice([1.0, 1.0, 0.7, 2.5])) for _ in range(ntypes)]
    extra = np.array([(2, i, i + 2) for i in range(5, n - 5, int(rng.randint(17, 60)))], dtype=np.int32)   # a few type-2 bonds
    if rng.rand() < 0.7:
        s["bonds"] = np.concatenate([s["bonds"], extra])
    rc = float(rng.choice([1.12, 1.12, 1.5, 2.0]))
    lines = ["units lj", "atom_style bond", "newton off" if rng.rand() < 0.7 else "newton on off",
             "atom_modify sort %d 0" % int(rng.choice([0, 0, 5, 1000])),
             str(rng.choice(["special_bonds fene", "special_bonds lj 0.0 1.0 1.0", "special_bonds lj 1 1 1",
                             "special_bonds lj %g %g %g" % tuple(rng.choice([0.0, 0.3, 0.5, 1.0], size=3))])),
             "read_data data.chain", "neighbor %g bin" % float(rng.choice([0.2, 0.3, 0.4, 0.6])),
             "neigh_modify every %d delay %d check %s" % (int(rng.choice([1, 1, 2, 3])), int(rng.choice([0, 1, 2, 5, 10])),
                                                          str(rng.choice(["yes", "yes", "no"])))]
    bstyle = str(rng.choice(["fene", "harmonic", "hybrid"]))
    if bstyle == "fene":
        lines += ["bond_style fene", "bond_coeff 1 30.0 1.5 1.0 1.0", "bond_coeff 2 %g %g 1.0 1.0" % (rng.uniform(5.0, 30.0), rng.uniform(3.0, 5.0))]
    elif bstyle == "harmonic":
        lines += ["bond_style harmonic", "bond_coeff 1 %g %g" % (rng.uniform(50.0, 300.0), rng.uniform(0.9, 1.1)),
                  "bond_coeff 2 %g %g" % (rng.uniform(5.0, 30.0), rng.uniform(1.2, 2.2))]
    else:
        lines += ["bond_style hybrid fene harmonic", "bond_coeff 1 fene 30.0 1.5 1.0 1.0", "bond_coeff 2 harmonic %g %g" % (rng.uniform(5.0, 30.0), rng.uniform(1.2, 2.2))]
    lines += ["pair_style lj/cut %g" % rc]
    if rng.rand() < 0.7:
        lines.append("pair_modify shift yes")
    if rng.rand() < 0.3:
        lines.append("pair_modify mix %s" % str(rng.choice(["geometric", "arithmetic"])))
    lines.append("pair_coeff * * 1.0 1.0")
    for t in range(1, ntypes + 1):
        if rng.rand() < 0.6:
            lines.append("pair_coeff %d %d %g %g %g" % (t, t, rng.uniform(0.5, 1.5), rng.uniform(0.85, 1.05), rng.uniform(1.0, rc)))
    if ntypes > 1 and rng.rand() < 0.5:
        lines.append("pair_coeff 1 2 %g %g" % (rng.uniform(0.5, 1.5), rng.uniform(0.9, 1.0)))
    lines.append("timestep %g" % float(rng.choice([0.003, 0.005, 0.008])))
    rng4 = np.random.RandomState(41000 + seed)          # (drawn apart: the scenarios of the first MD sweep keep their seeds)
    gn = gl = "all"
    if rng4.rand() < 0.3:                               # fixes on groups (under any atom_modify sort: the members' ranks follow it)
        kind = str(rng4.choice(["type", "id", "molecule"]))
        if kind == "type":
            lines.append("group g type %d" % int(rng4.randint(1, ntypes + 1)))
        elif kind == "id":
            lo = int(rng4.randint(1, n // 2))
            lines.append("group g id %d:%d:%d" % (lo, int(rng4.randint(lo, n + 1)), int(rng4.choice([1, 2, 3]))))
        else:
            lines.append("group g molecule %d" % int(rng4.randint(1, nchains + 1)))
        if rng4.rand() < 0.5:
            lines.append("group h subtract all g")
            gn = "h"
        else:
            gn = "g"
        gl = gn if rng4.rand() < 0.7 else "all"
    lines.append("fix 1 %s nve" % gn)
    if rng.rand() < 0.7:
        lines.append("fix 2 %s langevin %g %g %g %d" % (gl, rng.uniform(0.5, 1.5), rng.uniform(0.5, 1.5), float(rng.choice([0.5, 1.0, 10.0])), int(rng.randint(1, 900000))))
        if rng4.rand() < 0.25:                      # optional keywords: per-type damping, zeroed total random force
            lines[-1] += " scale %d %g" % (int(rng4.randint(1, ntypes + 1)), rng4.uniform(0.5, 4.0))
        if rng4.rand() < 0.2:
            lines[-1] += " zero yes"
    if rng.rand() < 0.3:
        lines.append("thermo_modify norm %s" % str(rng.choice(["yes", "no"])))
    lines.append("thermo %d" % int(rng.choice([5, 10, 25, 1000])))
    total = int(rng.randint(20, 90))
    cuts = sorted(set(int(c) for c in rng.randint(1, total, size=int(rng.randint(0, 3)))))
    last = 0
    for c in cuts + [total]:
        if c > last:
            lines.append("run %d" % (c - last))
            if c < total and rng4.rand() < 0.4:       # new velocities between two runs, for everyone or for the members of a group
                lines.append("velocity %s create %g %d %s" % (gn if rng4.rand() < 0.6 else "all", rng4.uniform(0.5, 1.5),
                                                              int(rng4.randint(1, 900000)),
                                                              str(rng4.choice(["", "dist gaussian", "loop local", "mom no", "rot yes", "sum yes"]))))
                # (not `loop geom` here: it seeds each atom from its coordinates' decimal digits, so after a run the 1e-12
                #  differences between two correct integrations give unrelated velocities; tests/test_host_cpu.py covers it)
        last = c
    return s, "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", _seeds("LE_FUZZ3_MD_SEEDS", 16))
def test_random_md_settings(tmp_path, seed):
    s, script = md_scenario(seed)
    try:
        o = run_oracle(script, s)
    except RuntimeError:
        from lammps_le_amd import LammpsError
        with pytest.raises(LammpsError):
            run_product(script, s, tmp_path)
        return
    p = run_product(script, s, tmp_path)
    xo = o.x()
    assert np.abs(p.gather("x") - xo).max() < 1e-8 * max(1.0, np.abs(xo).max())
    assert np.abs(p.gather("v") - o.v()).max() < 1e-7
    assert (p.gather("image") == o.image()).all()
    to = o.thermo()            # (per-atom energies; the oracle's script layer does not read `thermo_modify norm`)
    if "thermo_modify norm no" in script:
        to = to.copy()
        to[1:4] *= len(xo)
    for k, key in enumerate(("temp", "epair", "emol", "etotal", "press")):
        assert abs(p.get_thermo(key) - to[k]) <= 1e-8 * max(1.0, abs(to[k])), key
    assert p.stat("neigh_builds") == o.neigh_builds()
    assert p.stat("neigh_pairs") == 2 * o.neigh_pairs()
    # every thermo line of every run, not only the last one (thermo steps take other kernels than ordinary steps)
    hp, ho = p.thermo_history(), o.thermo_history()
    assert len(hp) == len(ho) and len(hp) >= 2
    scale = len(xo) if "thermo_modify norm no" in script else 1.0
    for rp, ro in zip(hp, ho):
        assert rp[0] == ro[0]
        for k, f in ((1, 1.0), (2, scale), (3, scale), (4, scale), (5, 1.0)):
            assert abs(rp[k] - f * ro[k]) <= 1e-8 * max(1.0, abs(f * ro[k])), (int(rp[0]), k)


@pytest.mark.parametrize("seed", _seeds("LE_FUZZ3_RESTART_SEEDS", 8))
def test_random_restart_continuity(tmp_path, seed):
    """A mixed LE scenario (test_gpu_fuzz2.py; not the r-RESPA ones: run_style is not part of the file) cut at a random step by write_restart + read_restart into a NEW instance: bit-identical to the uninterrupted run of the
    same two `run` commands - positions, velocities, images, types, bonds, special lists, fix counters - whatever the local
    order (`atom_modify sort N`: the sorted order and the next sort step are part of the state), and equal to the oracle."""
    from lammps_le_amd import lammps
    from test_gpu_fuzz2 import scenario
    s, script, ids, flavour = scenario(seed)
    if flavour == "respa":
        pytest.skip("run_style is not part of a restart file")
    lines = [ln for ln in script.split("\n") if not ln.startswith("run ")]
    total = sum(int(ln.split()[1]) for ln in script.split("\n") if ln.startswith("run "))
    rng = np.random.RandomState(23000 + seed)
    a_steps = int(rng.randint(1, total))
    head = "\n".join(lines) + "\n"
    later = "\n".join(ln for ln in lines if ln.startswith(("fix ", "thermo "))) + "\n"      # (groups, angles and their coefficients travel in the file)
    full = head + "run %d\nrun %d\n" % (a_steps, total - a_steps)
    try:
        o = run_oracle(full, s)
    except RuntimeError:
        pytest.skip("the oracle stops on this parameter set")
    a = run_product(full, s, tmp_path)
    rfile = str(tmp_path / "state.restart")
    b1 = run_product(head + "run %d\nwrite_restart %s\n" % (a_steps, rfile), s, tmp_path)
    b1.close()
    b = lammps(cmdargs=["-screen", "none"])
    b.command("read_restart " + rfile)
    for ln in (later + "run %d\n" % (total - a_steps)).split("\n"):
        b.command(ln)
    for name in ("x", "v", "image", "type", "num_bond", "bond_type", "bond_atom", "nspecial", "special") + \
            (("num_angle", "angle_type", "angle_atom1", "angle_atom2", "angle_atom3") if flavour == "angles" else ()):
        assert np.array_equal(a.gather(name), b.gather(name)), (name, flavour)
    for fid in ids:
        assert a.extract_fix(fid, 0, 1, 0) == b.extract_fix(fid, 0, 1, 0) and a.extract_fix(fid, 0, 1, 1) == b.extract_fix(fid, 0, 1, 1), fid
    assert b.bond_set() == o.bond_set() and np.abs(b.gather("x") - o.x()).max() < 1e-6
    b.close()


@pytest.mark.parametrize("seed", _seeds("LE_FUZZ3_PROC_SEEDS", 4))
def test_random_le_scenarios_mixed_decomposed_processes(tmp_path, seed, monkeypatch):
    """The decomposed mixed scenarios once more with the ranks as PROCESSES (file-mailbox transport) and the per-step halo
    through the peer windows in the one-launch form a multi-GPU run takes (forced here: the ranks share the GPU), every window
    halo also checked against the transport.  Oracle-stopped parameter sets are skipped (a stopped rank ends its peers by
    time-out: covered once, in test_gpu_dd.py)."""
    from test_gpu_dd import bond_set, run_ranks
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO", "1")
    monkeypatch.setenv("LAMMPS_LE_HALO_FUSED", str(seed % 2))
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO_VERIFY", "1")
    s, script, world, flavour = dd_scenario(seed)
    try:
        o = run_oracle(script, s)
    except RuntimeError:
        pytest.skip("the oracle stops on this parameter set")
    r = run_ranks(world, s, script, tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set(), flavour
    nso, spo = o.special_table()
    assert (r["nspecial"] == nso).all(), flavour
    for fid in ("loop", "loading", "unloading"):
        assert r["f_" + fid][0] == o.fix_vector(fid)[0] and r["f_" + fid][1] == o.fix_vector(fid)[1], (fid, flavour)
    assert np.abs(r["x"] - o.x()).max() < 1e-6, flavour
    assert r["builds"][0] == o.neigh_builds()
    assert int(r["window_mismatches"][0]) == 0
    assert int(r["window_exchanges"][0]) > 0 or "respa" in flavour      # (r-RESPA: every halo goes through the transport)
