"""Second randomised parity sweep: the LE fixes under everything else that changes what they see - local order (`atom_modify
sort N`, `newton on off`), r-RESPA, angles (`ex_load ... atype`, angle breaking in ex_unload), second instances of the
loader / unloader, type conversion at the bond limit, and runs cut into several `run` commands.  Product (HIP) against the
oracle: bond tables, special lists, fix counters, angle tables, reneighbor count bit for bit, positions to 1e-6.
The suite runs a fixed set of seeds; LE_FUZZ2_SEEDS="start:stop" runs a one-off wider sweep (scripts/r03_fuzz_wide.sh)."""
import os

import numpy as np
import pytest

from systems import CHAIN_SCRIPT, run_oracle, run_product
from test_gpu_le import barrier_types, compare, melted

pytestmark = pytest.mark.gpu

RESPA = ["run_style respa 2 4", "run_style respa 3 2 2 bond 1 pair 3", "run_style respa 2 3 bond 2 pair 2"]


def _seeds():
    v = os.environ.get("LE_FUZZ2_SEEDS")
    if not v:
        return list(range(24))
    a, b = v.split(":")
    return range(int(a), int(b))


def scenario(seed):
    rng = np.random.RandomState(7000 + seed)
    n = int(rng.choice([1500, 2400, 3600]))
    if os.environ.get("LE_FUZZ2_N"):       # one-off sweeps at sizes that take the throughput shapes of the kernels (> 64k beads:
        n = int(os.environ["LE_FUZZ2_N"])  # one lane per bead, energy variant on thermo steps, bond table by the permute pass)
    nchains = int(rng.choice([1, 3]))
    frac = float(rng.choice([0.0, 0.15, 0.4]))
    types = barrier_types(n, 90 + seed, frac=frac) if frac > 0 else np.ones(n, dtype=np.int32)
    s = melted(n, nchains=nchains, seed=1 + seed % 2, steps=600, types=types)
    s["ntypes"], s["mass"] = 4, [1.0] * 4
    flavour = str(rng.choice(["plain", "sort", "newton", "sort+newton", "respa", "angles", "angles", "convert"]))
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 6.0 8.0 1.0 1.0")
    if "sort" in flavour:
        base = base.replace("atom_modify sort 0 0", "atom_modify sort %d 0" % int(rng.randint(3, 16)))
    if "newton" in flavour:
        base = base.replace("newton off", "newton on off")
    angle_lines, atype = "", ""
    if flavour == "angles":
        per = n // nchains
        ang = [(1, i, i + 1, i + 2) for i in range(1, n - 1) if (i - 1) // per == (i + 1) // per]
        s["nangletypes"], s["angles"], s["extra_angle"] = 2, np.array(ang, dtype=np.int32), 24
        s["atom_style"] = "molecular"
        base = base.replace("atom_style bond", "atom_style molecular")
        if rng.rand() < 0.5:
            angle_lines = "angle_style harmonic\nangle_coeff 1 %.2f %.1f\nangle_coeff 2 %.2f %.1f\n" % (
                rng.uniform(0.5, 4.0), rng.uniform(120.0, 180.0), rng.uniform(0.5, 2.0), rng.uniform(90.0, 150.0))
        else:
            angle_lines = "angle_style cosine\nangle_coeff 1 %.2f\nangle_coeff 2 %.2f\n" % (rng.uniform(0.5, 3.0), rng.uniform(0.2, 1.5))
        atype = str(rng.choice(["", " atype 1", " atype 2"]))
    # (drawn behind everything above, so that the seeds of the first 600-seed sweep keep their scenarios when the three
    #  below stay at their defaults)
    rng2 = np.random.RandomState(17000 + seed)
    if rng2.rand() < 0.5:         # the reneighbor cadence the LE fixes' forced rebuilds cut into
        base = base.replace("neigh_modify every 1 delay 1 check yes", "neigh_modify every %d delay %d check %s" % (
            int(rng2.choice([1, 1, 2])), int(rng2.choice([0, 2, 5, 10])), str(rng2.choice(["yes", "yes", "no"]))))
    if rng2.rand() < 0.3:         # extruder bonds as harmonic springs (bond hybrid): their partner image is frozen at the rebuild
        base = base.replace("bond_style fene", "bond_style hybrid fene harmonic").replace("bond_coeff 1 30.0 1.5 1.0 1.0", "bond_coeff 1 fene 30.0 1.5 1.0 1.0") \
            .replace("bond_coeff 2 6.0 8.0 1.0 1.0", "bond_coeff 2 harmonic %g %g" % (rng2.uniform(2.0, 8.0), rng2.uniform(1.0, 2.0)))
    if rng2.rand() < 0.3:         # special weights other than the FENE set (fractional: list entries carry the level)
        base = base.replace("special_bonds fene", "special_bonds lj %s" % str(rng2.choice(["0 1 1", "1 1 1", "0 0.5 1", "0.5 0.5 0.5", "0 0 1"])))
    n1, nl, nu = int(rng.randint(3, 10)), int(rng.randint(3, 10)), int(rng.randint(3, 10))
    tp = float(rng.choice([0.0, 0.3, 0.7, 1.0]))
    lp, up = float(rng.choice([0.2, 0.6, 1.0])), float(rng.choice([0.1, 0.5, 1.0]))
    lprob = "" if lp >= 1.0 else "prob %g %d" % (lp, 100 + seed)
    uprob = "" if up >= 1.0 else "prob %g %d" % (up, 200 + seed)
    rmax = float(rng.choice([0.5, 1.3, 2.0]))
    lr = "4" if rng.rand() < 0.7 else ""
    iparam = "iparam 1 1 jparam 1 1"
    if flavour == "convert":          # beads at their bond limit change type (and stop being loadable / become barriers)
        nt = int(rng.choice([1, 2, 3, 4]))     # (itype == jtype: the reference insists on equal limits and new types for both ends)
        iparam = "iparam 1 %d jparam 1 %d" % (nt, nt)
    fixes = ["fix 1 all nve", "fix 2 all langevin 1.0 1.0 1.0 %d" % int(rng.randint(1, 900000)),
             "fix loop all extrusion %d 1 2 3 %g 2 %s" % (n1, tp, lr),
             "fix loading all ex_load %d 1 1 1.12 2 %s %s%s" % (nl, lprob, iparam, atype),
             "fix unloading all ex_unload %d 2 %g %s" % (nu, rmax, uprob)]
    ids = ["loop", "loading", "unloading"]
    if rng.rand() < 0.3:
        fixes.append("fix loading2 all ex_load %d 1 1 1.1 2 prob 0.4 %d iparam 1 1 jparam 1 1%s" % (int(rng.randint(4, 12)), 300 + seed, atype))
        ids.append("loading2")
    if rng.rand() < 0.3:
        fixes.append("fix unloading2 all ex_unload %d 2 0.9 prob 0.3 %d" % (int(rng.randint(5, 14)), 400 + seed))
        ids.append("unloading2")
    rng5 = np.random.RandomState(51000 + seed)      # (drawn apart, as above)
    group_line = ""
    if rng5.rand() < 0.2:                            # the LE fixes on a group: both atoms of a bond / candidate pair must be members
        lo = int(rng5.randint(1, n // 2))
        group_line = "group g id %d:%d\n" % (lo, int(rng5.randint(lo + n // 4, n + 1)))
        which = rng5.rand(len(fixes)) < 0.7
        fixes = [f.replace(" all ", " g ", 1) if (k >= 2 and which[k]) else f for k, f in enumerate(fixes)]
    fixes = [group_line.strip()] + fixes if group_line else fixes
    tail = "thermo 10\n"
    if flavour == "respa":
        tail += str(rng.choice(RESPA)) + "\n"
    rng6 = np.random.RandomState(61000 + seed)      # (drawn apart, as above)
    if lr and rng6.rand() < 0.15:                    # roadblock type = a barrier type: chained barrier draws (fix_extrusion.cpp:413-429)
        fixes = [f.replace(" 2 4", " 2 %d" % int(rng6.choice([2, 3])), 1) if f.startswith("fix loop ") else f for f in fixes]
    if flavour == "angles" and rng6.rand() < 0.3:    # semiflexible chains under r-RESPA: the angles at the bonds' level or their own
        tail += str(rng6.choice(["run_style respa 2 3", "run_style respa 3 2 2 bond 1 angle 2 pair 3",
                                 "run_style respa 2 2 bond 1 angle 2 pair 2"])) + "\n"
        flavour = "angles+respa"
    total = int(rng.randint(40, 110))
    cuts = sorted(set(int(c) for c in rng.randint(1, total, size=int(rng.randint(0, 3)))))
    runs, last = [], 0
    for c in cuts + [total]:
        runs.append(c - last)
        last = c
    # commands between two `run`s (every `run` re-runs setup: new lists, forces, fix setup)
    runs = [r for r in runs if r > 0]
    body = ""
    for k, r in enumerate(runs):
        body += "run %d\n" % r
        if k + 1 < len(runs) and rng2.rand() < 0.5:
            what = str(rng2.choice(["reset", "velocity", "timestep", "neigh"]))
            if what == "reset":          # shifts the phase of every firing period (and of the next Atom::sort)
                body += "reset_timestep %d\n" % int(rng2.choice([0, 7, 100, 1000]))
            elif what == "velocity":
                who = "g" if group_line and rng5.rand() < 0.6 else "all"     # (`velocity <group>`: members only)
                body += "velocity %s create %g %d %s\n" % (who, rng2.uniform(0.5, 1.5), int(rng2.randint(1, 900000)),
                                                            str(rng2.choice(["dist gaussian", "loop local", ""])))
            elif what == "timestep":
                body += "timestep %g\n" % float(rng2.choice([0.003, 0.004, 0.006]))
            else:
                body += "neigh_modify every 1 delay %d check yes\n" % int(rng2.choice([0, 1, 3]))
    script = base + angle_lines + "\n".join(fixes) + "\n" + tail + body
    return s, script, ids, flavour


@pytest.mark.parametrize("seed", _seeds())
def test_random_le_scenarios_mixed(tmp_path, seed):
    s, script, ids, flavour = scenario(seed)
    try:
        o = run_oracle(script, s)
    except RuntimeError:           # the oracle stops on this parameter set (Bad FENE bond, a full special list ..): so must the product
        from lammps_le_amd import LammpsError
        with pytest.raises(LammpsError):
            run_product(script, s, tmp_path)
        return
    p = run_product(script, s, tmp_path)
    compare(p, o, ids)
    assert p.stat("neigh_builds") == o.neigh_builds(), flavour
    hp, ho = p.thermo_history(), o.thermo_history()        # every thermo line: step, temp, epair, emol, etotal, press, bonds
    assert len(hp) == len(ho)
    for rp, ro in zip(hp, ho):
        assert rp[0] == ro[0] and rp[6] == ro[15], (flavour, int(rp[0]))
        for k in range(1, 6):
            assert abs(rp[k] - ro[k]) <= 1e-6 * max(1.0, abs(ro[k])), (flavour, int(rp[0]), k)
    if flavour.startswith("angles"):
        na, at, a1, a2, a3 = o.angle_table()
        assert (p.gather("num_angle") == na).all()
        for name, ref in (("angle_type", at), ("angle_atom1", a1), ("angle_atom2", a2), ("angle_atom3", a3)):
            got = p.gather(name)
            for i in np.nonzero(na)[0]:
                assert list(got[i, :na[i]]) == list(ref[i, :na[i]]), (name, i + 1)
        assert p.extract_setting("nangles") == o.nangles()
