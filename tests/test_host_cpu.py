"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/lammps_le.h
declares, the script layer parses / rejects like the reference, and the run command fails loudly without
a HIP device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from systems import CHAIN_SCRIPT, lattice_chain, write_data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from lammps_le_amd import library_path
    lib = ctypes.CDLL(library_path())
    hdr = open(os.path.join(ROOT, "include", "lammps_le.h")).read()
    names = set(re.findall(r"\b(lammps_[a-z_]+)\s*\(", hdr))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n


def _open(tmp_path, n=500):
    from lammps_le_amd import lammps
    s = lattice_chain(n)
    path = os.path.join(str(tmp_path), "data.chain")
    write_data(path, s)
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in CHAIN_SCRIPT.split("\n"):
        lmp.command(ln.replace("data.chain", path))
    return lmp, s


def test_script_layer_and_queries(tmp_path):
    lmp, s = _open(tmp_path)
    assert lmp.get_natoms() == 500
    assert lmp.extract_setting("bond_per_atom") == 3        # 2 backbone + 1 extra bond per atom
    assert lmp.extract_setting("maxspecial") == 22          # special_bonds fene: 2 + 20 extra
    x = lmp.gather("x")
    assert np.allclose(x, s["x"])
    nb = lmp.gather("num_bond")
    assert nb[0] == 1 and nb[-1] == 1 and (nb[1:-1] == 2).all()   # both atoms store every bond
    assert (1, 1, 2) in lmp.bond_set() and len(lmp.bond_set()) == 499
    lmp.command("fix 1 all nve")
    lmp.command("fix loop all extrusion 1000 1 1 1 1.0 2")
    lmp.command("fix loading all ex_load 1000 1 1 1.12 2 prob 0.5 684474 iparam 1 1 jparam 1 1")
    lmp.command("fix unloading all ex_unload 1000 2 0.5 prob 0.5 456456")
    lmp.command("fix breaking all bond/break 1000 2 0.5 prob 0.5 456456")          # src/MC/fix_bond_break.cpp grammar
    lmp.command("fix creating all bond/create 1000 1 1 1.1 2 iparam 2 1 jparam 2 1 prob 0.5 8847")   # src/MC/fix_bond_create.cpp grammar
    for style in ("nve", "langevin", "extrusion", "ex_load", "ex_unload", "bond/break", "bond/create"):
        assert lmp.has_style("fix", style)


def test_errors_match_reference_messages(tmp_path):
    from lammps_le_amd import LammpsError
    lmp, _ = _open(tmp_path)
    with pytest.raises(LammpsError, match="Unknown command: frobnicate"):      # src/input.cpp:352-353
        lmp.command("frobnicate 1 2")
    with pytest.raises(LammpsError, match="Illegal fix extrusion command"):   # fix_extrusion.cpp:43-44 (+ arg[8] read)
        lmp.command("fix bad all extrusion 1000 1 1 1 1.0")
    with pytest.raises(LammpsError, match="Illegal fix ex_load command"):
        lmp.command("fix bad2 all ex_load 1000 1 1 1.12 2 prob 1.5 1")
    with pytest.raises(LammpsError, match="Illegal fix bond/break command"):
        lmp.command("fix bad2b all bond/break 1000 2 0.5 prob 1.5 1")
    lmp.command("run_style verlet")
    lmp.command("run_style respa 2 2")
    with pytest.raises(LammpsError, match="Respa levels must be >= 1"):           # src/respa.cpp:57
        lmp.command("run_style respa 0")
    with pytest.raises(LammpsError, match="Illegal run_style respa command"):     # :60-64
        lmp.command("run_style respa 3 2")
    with pytest.raises(LammpsError, match="Invalid order of forces within respa levels"):   # :218-224
        lmp.command("run_style respa 2 2 bond 2 pair 1")
    with pytest.raises(LammpsError, match="run_style respa inner"):
        lmp.command("run_style respa 2 2 inner 1 0.8 1.0 outer 2")
    with pytest.raises(LammpsError, match="run_style verlet/split is not supported"):
        lmp.command("run_style verlet/split")
    lmp.command("run_style verlet")
    with pytest.raises(LammpsError, match="Unknown fix style"):
        lmp.command("fix bad3 all nvt temp 1 1 1")
    with pytest.raises(LammpsError, match="Fix langevin period must be > 0.0"):
        lmp.command("fix bad4 all langevin 1.0 1.0 0.0 1234")


def test_dump_and_compute_commands_are_parsed(tmp_path):
    """Argument grammar and error strings of dump / dump_modify / undump / compute property/local (host side only)."""
    from lammps_le_amd import LammpsError
    lmp, _ = _open(tmp_path)
    lmp.command("compute pl all property/local btype batom1 batom2")
    lmp.command("dump 1 all custom 100 %s id type x y z" % (tmp_path / "a.dump"))
    lmp.command("dump 2 all local 100 %s index c_pl[1] c_pl[3]" % (tmp_path / "b.dump"))
    lmp.command("dump_modify 1 sort id")
    assert lmp.has_style("dump", "custom") and lmp.has_style("dump", "local") and lmp.has_style("compute", "property/local")
    with pytest.raises(LammpsError, match="Reuse of dump ID"):                 # src/output.cpp:560
        lmp.command("dump 1 all atom 10 x.dump")
    with pytest.raises(LammpsError, match="Invalid dump frequency"):           # src/output.cpp:575
        lmp.command("dump 3 all atom 0 x.dump")
    with pytest.raises(LammpsError, match="Unknown dump style"):
        lmp.command("dump 3 all xyzzy 10 x.dump")
    with pytest.raises(LammpsError, match="out-of-range"):                      # src/dump_local.cpp:443-446
        lmp.command("dump 3 all local 10 x.dump c_pl[4]")
    with pytest.raises(LammpsError, match="Could not find dump local compute ID"):
        lmp.command("dump 3 all local 10 x.dump c_nope[1]")
    with pytest.raises(LammpsError, match="not supported"):
        lmp.command("dump 3 all custom 10 x.dump id q")
    with pytest.raises(LammpsError, match="Could not find undump ID"):        # src/output.cpp:690
        lmp.command("undump 7")
    lmp.command("undump 2")
    lmp.command("uncompute pl")


def test_restart_round_trip_on_the_host(tmp_path):
    """write_restart / read_restart of a system that has not run yet: everything the script defined comes back."""
    from lammps_le_amd import LammpsError, lammps
    lmp, s = _open(tmp_path)
    lmp.command("fix 2 all langevin 1.0 1.0 1.0 904297")
    rf = str(tmp_path / "r.bin")
    lmp.command("write_restart " + rf)
    new = lammps(cmdargs=["-screen", "none"])
    with pytest.raises(LammpsError, match="Cannot open restart file"):
        new.command("read_restart " + rf + ".missing")
    new.command("read_restart " + rf)
    assert new.get_natoms() == 500 and np.array_equal(new.gather("x"), lmp.gather("x"))
    assert new.bond_set() == lmp.bond_set() and np.array_equal(new.gather("special"), lmp.gather("special"))
    assert new.extract_setting("bond_per_atom") == 3
    open(rf + ".bad", "wb").write(b"not a restart file at all")
    other = lammps(cmdargs=["-screen", "none"])
    with pytest.raises(LammpsError, match="truncated|not a lammps_le_amd restart"):
        other.command("read_restart " + rf + ".bad")
    with pytest.raises(LammpsError, match="before simulation box is defined"):
        other.command("write_restart x")


def test_run_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from lammps_le_amd import LammpsError
    lmp, _ = _open(tmp_path)
    lmp.command("fix 1 all nve")
    with pytest.raises(LammpsError, match="No HIP device"):
        lmp.command("run 1")


def test_ranmars_host_matches_oracle_and_jump():
    """The product's integer RanMars (host side of the device generator) against the oracle's restatement
    of src/random_mars.cpp, including polynomial jump-ahead."""
    from oracle import ranmars_stream
    lib = ctypes.CDLL(os.path.join(ROOT, "lammps_le_amd", "liblammps_le.so"))
    fn = lib.lammps_le_test_ranmars
    fn.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    for seed in (12345, 904297, 1, 900000000):
        ref = ranmars_stream(seed, 30000)
        out = np.zeros(1000)
        fn(seed, 0, 1000, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        assert np.array_equal(out, ref[:1000])
        for skip in (5, 4096, 12345, 28999):
            fn(seed, skip, 1000, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
            assert np.array_equal(out, ref[skip:skip + 1000]), (seed, skip)


# ------------------------------------------------------------------------------------------------------------
# velocity command (SURVEY 8f widening: the reference's scripts create their start velocities this way)
def _velocity_case(tmp_path, n, cmd, shuffle=False, v0=False):
    from lammps_le_amd import lammps
    types = 1 + (np.arange(n) % 3 == 0).astype(np.int32)
    s = lattice_chain(n, types=types)
    s["mass"] = [1.0, 2.5]
    s["image"] = np.random.RandomState(5).randint(-1, 2, size=(n, 3)).astype(np.int32)
    if not v0:
        s["v"] = np.zeros_like(s["v"])
    path = os.path.join(str(tmp_path), "data.chain")
    write_data(path, s)
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in CHAIN_SCRIPT.split("\n"):
        lmp.command(ln.replace("data.chain", path))
    lmp.command(cmd)
    return lmp, s, np.asarray(s["mass"])[types - 1]


def test_ranpark_published_check_value():
    """Park & Miller 1988: the minimal standard generator started at 1 holds 1043618065 after 10000 draws."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from velocity_oracle import RanPark
    r = RanPark(1)
    for _ in range(10000):
        r.uniform()
    assert r.seed == 1043618065


@pytest.mark.parametrize("opts", ["", "dist gaussian", "loop local", "loop geom dist gaussian", "mom no rot yes",
                                  "rot yes dist gaussian loop local"])
def test_velocity_create_matches_the_restatement(tmp_path, opts):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from velocity_oracle import velocity_create
    n = 700
    lmp, s, m = _velocity_case(tmp_path, n, "velocity all create 1.3 4928459 " + opts)
    v = lmp.gather("v").reshape(n, 3)
    kw = dict(dist="gaussian" if "gaussian" in opts else "uniform", mom="mom no" not in opts, rot="rot yes" in opts,
              loop="local" if "local" in opts else "geom" if "geom" in opts else "all")
    prd = s["box"][:, 1] - s["box"][:, 0]
    # the engine holds the coordinates it parsed from the data file: hash those, not the pre-print values
    x = lmp.gather("x").reshape(n, 3)
    ref = velocity_create(x, s["image"], prd, m, 1.3, 4928459, **kw)
    assert np.abs(v - ref).max() < 1e-13
    # the properties the command promises
    T = (m[:, None] * v * v).sum() / (3 * n - 3)
    assert abs(T - 1.3) < 1e-12
    if kw["mom"]:
        assert np.abs((m[:, None] * v).sum(axis=0)).max() < 1e-10
    if kw["rot"]:
        xu = x + s["image"] * prd
        xcm = (m[:, None] * xu).sum(axis=0) / m.sum()
        assert np.abs((m[:, None] * np.cross(xu - xcm, v)).sum(axis=0)).max() < 1e-8


def test_velocity_set_scale_zero_sum(tmp_path):
    from lammps_le_amd import LammpsError
    n = 300
    lmp, s, m = _velocity_case(tmp_path, n, "velocity all set 0.5 NULL -0.25", v0=True)
    v = lmp.gather("v").reshape(n, 3)
    assert (v[:, 0] == 0.5).all() and (v[:, 2] == -0.25).all() and np.allclose(v[:, 1], s["v"][:, 1], rtol=0, atol=1e-15)
    lmp.command("velocity all set 0.5 0.5 0.5 sum yes")
    v2 = lmp.gather("v").reshape(n, 3)
    assert np.abs(v2 - (v + 0.5)).max() == 0.0
    lmp.command("velocity all zero linear")
    v3 = lmp.gather("v").reshape(n, 3)
    assert np.abs((m[:, None] * v3).sum(axis=0)).max() < 1e-10
    lmp.command("velocity all scale 0.7")
    v4 = lmp.gather("v").reshape(n, 3)
    assert abs((m[:, None] * v4 * v4).sum() / (3 * n - 3) - 0.7) < 1e-12
    lmp.command("velocity all create 2.0 77 sum yes")
    v5 = lmp.gather("v").reshape(n, 3)
    w = v5 - v4
    assert abs((m[:, None] * w * w).sum() / (3 * n - 3) - 2.0) < 1e-9
    with pytest.raises(LammpsError, match="Illegal velocity create command"):     # src/velocity.cpp:167
        lmp.command("velocity all create 1.0 0")
    with pytest.raises(LammpsError, match="Illegal velocity command"):
        lmp.command("velocity all create 1.0 5 dist cauchy")
    with pytest.raises(LammpsError, match="Illegal velocity command"):
        lmp.command("velocity all spin 1.0")
    lmp.command("velocity all set 0 0 0")
    with pytest.raises(LammpsError, match="Attempting to rescale a 0.0 temperature"):   # src/velocity.cpp:735
        lmp.command("velocity all scale 1.0")


@pytest.mark.parametrize("opts", ["", "dist gaussian loop local", "loop geom rot yes", "sum yes"])
def test_velocity_on_a_group(tmp_path, opts):
    """`velocity <group> ...` (src/velocity.cpp: `mask[i] & groupbit` in every loop): loop all draws a triple for every ID and
    assigns the members', loop local / geom draw for members only; the temperature, the momentum and the angular momentum are the
    group's; everyone else keeps the velocity it had."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from velocity_oracle import velocity_create
    from lammps_le_amd import LammpsError
    n = 600
    lmp, s, m = _velocity_case(tmp_path, n, "group heavy type 2", v0=True)
    lmp.command("group odd id 1:%d:2" % n)
    lmp.command("group sel union heavy odd")
    member = (np.arange(n) % 3 == 0) | (np.arange(n) % 2 == 0)
    v0 = lmp.gather("v").reshape(n, 3).copy()
    x = lmp.gather("x").reshape(n, 3)
    lmp.command("velocity sel create 0.9 31337 " + opts)
    v = lmp.gather("v").reshape(n, 3)
    kw = dict(dist="gaussian" if "gaussian" in opts else "uniform", rot="rot yes" in opts,
              loop="local" if "local" in opts else "geom" if "geom" in opts else "all")
    prd = s["box"][:, 1] - s["box"][:, 0]
    ref = velocity_create(x, s["image"], prd, m, 0.9, 31337, member=member, vcur=v0,
                          vold=v0 if "sum yes" in opts else None, **kw)
    assert np.abs(v - ref).max() < 1e-13
    assert np.array_equal(v[~member], v0[~member])
    w = v[member] - (v0[member] if "sum yes" in opts else 0.0)
    mm = m[member]
    assert abs((mm[:, None] * w * w).sum() / (3 * member.sum() - 3) - 0.9) < 1e-12
    assert np.abs((mm[:, None] * w).sum(axis=0)).max() < 1e-10
    # set / scale / zero act on the members only
    lmp.command("velocity heavy set NULL 0.25 NULL")
    v2 = lmp.gather("v").reshape(n, 3)
    heavy = np.arange(n) % 3 == 0
    assert (v2[heavy, 1] == 0.25).all() and np.array_equal(v2[~heavy], v[~heavy]) and np.array_equal(v2[:, 0], v[:, 0])
    lmp.command("velocity odd scale 0.4")
    v3 = lmp.gather("v").reshape(n, 3)
    odd = np.arange(n) % 2 == 0
    assert abs((m[odd][:, None] * v3[odd] ** 2).sum() / (3 * odd.sum() - 3) - 0.4) < 1e-12
    assert np.array_equal(v3[~odd], v2[~odd])
    lmp.command("velocity heavy zero linear")
    v4 = lmp.gather("v").reshape(n, 3)
    assert np.abs((m[heavy][:, None] * v4[heavy]).sum(axis=0)).max() < 1e-10 and np.array_equal(v4[~heavy], v3[~heavy])
    # the group bits as `mask` (bit 0 = all, then the groups in definition order: heavy, odd, sel)
    mask = lmp.gather("mask")
    assert (mask & 1).all() and np.array_equal((mask & 2) != 0, heavy) and np.array_equal((mask & 4) != 0, odd)
    assert np.array_equal((mask & 8) != 0, member)
    with pytest.raises(LammpsError, match="Could not find velocity group ID"):     # src/velocity.cpp:65
        lmp.command("velocity nobody create 1.0 5")
    lmp.command("group none empty")
    with pytest.raises(LammpsError, match="Cannot zero momentum of no atoms"):     # src/velocity.cpp:760
        lmp.command("velocity none zero linear")


def test_round3_commands_are_parsed(tmp_path):
    """Parse-time behaviour of what round 3 added to the script layer (no device needed): thermo keywords are checked when the
    style is set (src/thermo.cpp:884-1040), fix langevin keywords (src/fix_langevin.cpp:105-155), the `angle` level of
    run_style respa (src/respa.cpp:85-88, 217-224), dumps / computes / velocity on groups, chained barrier types."""
    from lammps_le_amd import LammpsError
    lmp, _ = _open(tmp_path)
    lmp.command("thermo_style custom step elapsed time cpu temp press pe ke etotal enthalpy evdwl ecoul epair ebond eangle emol vol "
                "density lx ylo zhi bonds angles nbuild ndanger pxx pyz")
    lmp.command("thermo_style multi")
    lmp.command("thermo_modify line one norm no")
    with pytest.raises(LammpsError, match="Unknown keyword in thermo_style custom command: colour"):
        lmp.command("thermo_style custom step colour")
    with pytest.raises(LammpsError, match="Illegal thermo_modify command"):
        lmp.command("thermo_modify line diagonal")
    assert lmp.get_thermo("vol") > 0.0 and lmp.get_thermo("atoms") == lmp.get_natoms() and lmp.get_thermo("dt") == 0.005
    lmp.command("group odd id 1:1000:2")
    lmp.command("fix t1 odd langevin 1.0 1.0 1.0 77 scale 1 2.5 zero yes tally no gjf no omega no angmom no")
    for bad, msg in (("fix t2 all langevin 1.0 1.0 1.0 77 scale 9 1.0", "Illegal fix langevin command"),
                     ("fix t2 all langevin 1.0 1.0 1.0 77 zero", "Illegal fix langevin command"),
                     ("fix t2 all langevin 1.0 1.0 1.0 77 gjf vfull", "not supported"),
                     ("fix t2 all langevin 1.0 1.0 1.0 77 tally yes", "not supported"),
                     ("fix t2 nobody langevin 1.0 1.0 1.0 77", "Could not find fix group ID")):
        with pytest.raises(LammpsError, match=msg):
            lmp.command(bad)
    lmp.command("run_style respa 3 2 2 bond 1 angle 2 pair 3")
    with pytest.raises(LammpsError, match="Invalid order of forces within respa levels"):
        lmp.command("run_style respa 3 2 2 bond 2 angle 1 pair 3")
    lmp.command("run_style verlet")
    lmp.command("compute pg odd property/local btype batom1")
    lmp.command("dump d1 odd custom 10 %s id type x" % (tmp_path / "g.dump"))
    lmp.command("dump d2 odd dcd 10 %s" % (tmp_path / "g.dcd"))
    lmp.command("velocity odd create 1.0 4711 loop local")
    lmp.command("velocity odd zero linear")
    for bad, msg in (("dump d3 nobody atom 10 x.dump", "Could not find dump group ID"),
                     ("compute c3 nobody property/local btype", "Could not find compute group ID"),
                     ("velocity nobody set 0 0 0", "Could not find velocity group ID")):
        with pytest.raises(LammpsError, match=msg):
            lmp.command(bad)
    # a roadblock type equal to a barrier type is a legal fix extrusion (chained barrier draws)
    lmp.command("fix loop all extrusion 1000 1 1 1 0.5 2 1")


def test_group_operators_clear_and_delete(tmp_path):
    """`group ID type|id|molecule <op> value` (`<>` = between two bounds), `group ID clear`, `group ID delete` (a deleted
    group's bit is reused by the next new group; a group a fix / dump / compute uses cannot be deleted): src/group.cpp:103-280."""
    from lammps_le_amd import LammpsError
    n = 600
    lmp, s, m = _velocity_case(tmp_path, n, "group a id <= 100")
    ids = np.arange(1, n + 1)
    types = lmp.gather("type")
    lmp.command("group b id <> 250 300")
    lmp.command("group c type != 1")
    lmp.command("group d id > 590")
    mask = lmp.gather("mask")
    assert np.array_equal((mask & 2) != 0, ids <= 100) and np.array_equal((mask & 4) != 0, (ids >= 250) & (ids <= 300))
    assert np.array_equal((mask & 8) != 0, types != 1) and np.array_equal((mask & 16) != 0, ids > 590)
    lmp.command("group b clear")
    assert not (lmp.gather("mask") & 4).any()
    lmp.command("group b id 7 8 9")                       # the cleared group is still there
    assert np.array_equal((lmp.gather("mask") & 4) != 0, np.isin(ids, [7, 8, 9]))
    lmp.command("fix 1 c nve")
    with pytest.raises(LammpsError, match="Cannot delete group currently used by a fix"):
        lmp.command("group c delete")
    lmp.command("dump 1 d atom 10 %s" % (tmp_path / "d.dump"))
    with pytest.raises(LammpsError, match="Cannot delete group currently used by a dump"):
        lmp.command("group d delete")
    lmp.command("group a delete")
    assert not (lmp.gather("mask") & 2).any()
    with pytest.raises(LammpsError, match="Could not find fix group ID"):
        lmp.command("fix 2 a nve")
    lmp.command("group e id == 42")                       # takes the freed bit
    assert np.array_equal((lmp.gather("mask") & 2) != 0, ids == 42)
    for bad, msg in (("group all delete", "Cannot change the group all"), ("group nosuch clear", "Could not find group clear group ID"),
                     ("group nosuch delete", "Could not find group delete group ID"), ("group f id <> 5", "Illegal group command"),
                     ("group f id < 5 6", "Illegal group command")):
        with pytest.raises(LammpsError, match=msg):
            lmp.command(bad)


def test_regions_select_atoms(tmp_path):
    """`region ID block | sphere | cylinder | union | intersect ... [side in|out]` (src/region*.cpp: closed boundaries, match =
    !(inside ^ interior)), `group ID region R` (the atoms inside NOW, src/group.cpp:174-186) and `set region R ...`
    (src/set.cpp:671-678) against numpy."""
    from lammps_le_amd import LammpsError
    n = 900
    lmp, s, m = _velocity_case(tmp_path, n, "region b block 2.0 6.0 INF INF EDGE 5.5")
    x = lmp.gather("x").reshape(n, 3)
    lo, hi = s["box"][:, 0], s["box"][:, 1]
    c = 0.5 * (lo + hi)
    lmp.command("region s sphere %g %g %g 3.5 units box" % tuple(c))
    lmp.command("region so sphere %g %g %g 3.5 side out" % tuple(c))
    lmp.command("region cy cylinder y %g %g 2.5 %g EDGE" % (c[0], c[2], c[1]))
    lmp.command("region u union 2 b s")
    lmp.command("region it intersect 3 b s cy")
    in_b = (x[:, 0] >= 2.0) & (x[:, 0] <= 6.0) & (x[:, 2] >= lo[2]) & (x[:, 2] <= 5.5)
    in_s = np.sqrt(((x - c) ** 2).sum(axis=1)) <= 3.5
    in_cy = (np.sqrt((x[:, 0] - c[0]) ** 2 + (x[:, 2] - c[2]) ** 2) <= 2.5) & (x[:, 1] >= c[1]) & (x[:, 1] <= hi[1])
    want = dict(b=in_b, s=in_s, so=~in_s, cy=in_cy, u=in_b | in_s, it=in_b & in_s & in_cy)
    for k, (name, sel) in enumerate(want.items()):
        assert 0 < sel.sum() < n, name
        lmp.command("group g_%s region %s" % (name, name))
        mask = lmp.gather("mask")
        assert np.array_equal((mask & (2 << k)) != 0, sel), name
    types0 = lmp.gather("type").copy()
    lmp.command("set region it type 2")
    t = lmp.gather("type")
    assert (t[want["it"]] == 2).all() and np.array_equal(t[~want["it"]], types0[~want["it"]])
    lmp.command("region b delete")
    for bad, msg in (("region s sphere 0 0 0 1", "Reuse of region ID"), ("region q block 1 0 0 1 0 1", "Illegal region block command"),
                     ("region q cone z 0 0 1 2 0 1", "Unknown region style"), ("group h region b", "Group region ID does not exist"),
                     ("set region nowhere type 1", "Set region ID does not exist"), ("region q sphere 0 0 0 1 move v_a NULL NULL", "not supported"),
                     ("region q union 2 s nowhere", "region ID does not exist"), ("region b delete", "Delete region ID does not exist"),
                     ("group h region u", "region ID does not exist")):
        with pytest.raises(LammpsError, match=msg):
            lmp.command(bad)


# ------------------------------------------------------------------------------------------------------------
# script control flow: variable index / loop / equal, next, label, jump, if, include, $(...) (src/input.cpp, variable.cpp)
def test_script_control_flow(tmp_path):
    from lammps_le_amd import lammps
    s = lattice_chain(200)
    data = os.path.join(str(tmp_path), "data.chain")
    write_data(data, s)
    log = tmp_path / "log.flow"
    inc = tmp_path / "in.inc"
    inc.write_text('print "included ${tag}"\n')
    script = tmp_path / "in.flow"
    script.write_text(CHAIN_SCRIPT.replace("data.chain", data) + f'''
log {log}
variable tag string alpha
variable i loop 3
variable f index {tmp_path}/a.data {tmp_path}/b.data
variable twice equal 2*v_i+atoms/100
label top
print "pass $i of 3: twice=${{twice}} next=$(v_i+1) fmt=$(v_i/4:%.3f)"
if "$i == 2" then "print 'second pass'" "variable tag string beta" elif "$i > 2" "print 'late'" else "print 'first pass'"
include {inc}
next i
jump SELF top
print "after loop"
variable k loop 2 4 pad
label again
write_data ${{f}}
next f
next k
jump SELF again
if "${{tag}} == beta" then "print 'string compare ok'"
print "done $(step)"
''')
    lmp = lammps(cmdargs=["-screen", "none"])
    lmp.file(str(script))
    lmp.close()
    text = log.read_text().split("\n")
    assert "pass 1 of 3: twice=4 next=2 fmt=0.250" in text
    assert "pass 2 of 3: twice=6 next=3 fmt=0.500" in text
    assert "pass 3 of 3: twice=8 next=4 fmt=0.750" in text
    assert [t for t in text if t in ("first pass", "second pass", "late")] == ["first pass", "second pass", "late"]
    assert [t for t in text if t.startswith("included")] == ["included alpha", "included beta", "included beta"]
    assert "after loop" in text and "string compare ok" in text and "done 0" in text
    assert os.path.exists(tmp_path / "a.data") and os.path.exists(tmp_path / "b.data")   # f ran out first: loop left after 2


def test_script_control_flow_errors(tmp_path):
    from lammps_le_amd import LammpsError
    lmp, _ = _open(tmp_path, 50)
    with pytest.raises(LammpsError, match="Substitution for illegal variable"):
        lmp.command("print ${nope}")
    with pytest.raises(LammpsError, match="Illegal if command"):
        lmp.command('if "1 > 0" print')
    with pytest.raises(LammpsError, match="Divide by 0"):
        lmp.command("variable z equal 1/0")
        lmp.command("print ${z}")
    with pytest.raises(LammpsError, match="jump is only available inside an input script"):
        lmp.command("jump SELF x")
    p = tmp_path / "in.nolabel"
    p.write_text("jump SELF missing\n")
    with pytest.raises(LammpsError, match="Label wasn't found"):
        lmp.file(str(p))
    lmp.command("variable a equal (3+4)*2^2-sqrt(16)")
    lmp.command("variable b equal v_a>=24&&!(v_a>24)")
    lmp.command('if "${b}" then "variable ok string yes"')
    lmp.command("print ${ok}")


def test_set_command(tmp_path):
    """`set atom|type|mol|group ... type | type/fraction | mol | x.. | vx.. | image` (src/set.cpp); the fraction variant
    re-seeds RanPark from each bead's coordinates, checked against the numpy restatement."""
    import sys
    from lammps_le_amd import LammpsError
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from velocity_oracle import RanPark
    n = 400
    lmp, s = _open(tmp_path, n)
    lmp.command("set atom 100*120 type 1")
    with pytest.raises(LammpsError, match="Invalid value in set command"):
        lmp.command("set atom 5 type 9")
    # the chain script defines one type: re-open with three
    from lammps_le_amd import lammps
    sys3 = lattice_chain(n, types=np.ones(n, dtype=np.int32))
    sys3["ntypes"], sys3["mass"] = 3, [1.0, 1.0, 1.0]
    path = os.path.join(str(tmp_path), "data3.chain")
    write_data(path, sys3)
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in CHAIN_SCRIPT.split("\n"):
        lmp.command(ln.replace("data.chain", path))
    lmp.command("set atom 100*120 type 2")
    lmp.command("set atom 7 type 3 vx 0.5 image 1 NULL -2")
    t = lmp.gather("type")
    assert (t[99:120] == 2).all() and t[6] == 3 and (np.delete(t, list(range(99, 120)) + [6]) == 1).all()
    assert lmp.gather("v").reshape(n, 3)[6, 0] == 0.5 and list(lmp.gather("image").reshape(n, 3)[6]) == [1, 0, -2]
    lmp.command("set type 2 type/fraction 3 0.4 12345")
    t2 = lmp.gather("type")
    x = lmp.gather("x").reshape(n, 3)
    rp = RanPark(1)
    expect = t.copy()
    for i in range(n):
        if t[i] == 2:
            rp.reset(12345, x[i])
            if rp.uniform() <= 0.4:
                expect[i] = 3
    assert (t2 == expect).all() and 0 < (t2[99:120] == 3).sum() < 21
    lmp.command("set group all mol 5")
    lmp.command("variable sh equal 0.25")
    lmp.command("set mol 5 vz v_sh")
    assert (lmp.gather("v").reshape(n, 3)[:, 2] == 0.25).all()
    with pytest.raises(LammpsError, match="set keyword charge is not supported"):
        lmp.command("set atom 1 charge 1.0")


def test_oracle_ex_load_candidates_are_pair_list_entries():
    """Oracle side of tests/test_gpu_le.py::test_ex_load_sees_only_pairs_of_the_pair_list (runs without a GPU)."""
    from systems import CHAIN_SCRIPT, lattice_chain, run_oracle
    s = lattice_chain(1500, seed=4)
    o = run_oracle(CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 99\nrun 1200\n", s)
    s["x"], s["v"], s["image"] = o.x(), o.v(), o.image()
    tail = ("fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n"
            "fix loading all ex_load 10 1 1 1.12 2 iparam 1 1 jparam 1 1\nrun 24\n")
    loads = {}
    for sp in ("fene", "lj 0 0 1", "lj 0 0 1 coul 0 1 1", "lj 1 1 1"):
        o = run_oracle(CHAIN_SCRIPT.replace("special_bonds fene", "special_bonds " + sp) + tail, s)
        loads[sp] = o.fix_vector("loading")[1]
    assert loads["fene"] > 0 and loads["lj 1 1 1"] > 0 and loads["lj 0 0 1 coul 0 1 1"] > 0
    assert loads["lj 0 0 1"] == 0      # 1-3 pairs are not in the list: nothing to scan


@pytest.mark.parametrize("order", ["newton off\natom_modify sort 0 0", "newton off\natom_modify sort 7 0", "newton on off\natom_modify sort 0 0"])
def test_oracle_le_cycle_invariants(order):
    """The oracle alone through extrusion / ex_load / ex_unload cycles in the three visit orders it restates (ID order,
    Atom::sort order, half/bin/newton storing order): structural invariants of the topology it leaves behind.  Also the
    workload for the sanitizer build (`make -C oracle asan`, see oracle/Makefile)."""
    from systems import CHAIN_SCRIPT, lattice_chain, run_oracle
    n = 2000
    s = lattice_chain(n, seed=6)
    o = run_oracle(CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 77\nrun 1000\n", s)
    s["x"], s["v"], s["image"] = o.x(), o.v(), o.image()
    script = CHAIN_SCRIPT.replace("newton off\natom_modify sort 0 0", order) \
        .replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0") + """fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion 4 1 1 1 1.0 2
fix loading all ex_load 5 1 1 1.12 2 prob 0.5 684474 iparam 1 1 jparam 1 1
fix unloading all ex_unload 6 2 0.5 prob 0.3 456456
run 120
"""
    assert order.split("\n")[0] in script and order.split("\n")[1] in script
    o = run_oracle(script, s)
    nb, bt, ba = o.bond_table()
    ns, sp = o.special_table()
    half = {(int(bt[i, m]), i + 1, int(ba[i, m])) for i in range(n) for m in range(nb[i])}
    assert all((t, j, i) in half for (t, i, j) in half)                     # stored by both ends
    assert all((1, i, i + 1) in half for i in range(1, n))                  # backbone intact
    ext = [(i, j) for (t, i, j) in half if t == 2 and i < j]
    ends = [e for p in ext for e in p]
    assert len(ext) > 3 and len(ends) == len(set(ends))                     # one anchor per bead at most
    assert o.nbonds() == n - 1 + len(ext)
    assert o.fix_vector("loading")[1] - o.fix_vector("unloading")[1] == len(ext)
    for i in range(n):                                                      # 1-2 block = bond partners
        assert sorted(sp[i, :ns[i, 0]]) == sorted(int(ba[i, m]) for m in range(nb[i])), i + 1


def test_oracle_one_level_respa_is_verlet():
    """Respa::recurse with a single level performs Verlet::run's operations in Verlet::run's order (src/respa.cpp:600-741 vs
    src/verlet.cpp:240-353): the oracle's two integrators must agree to the last bit; more levels must not."""
    from systems import CHAIN_SCRIPT, lattice_chain, run_oracle
    s = lattice_chain(600, nchains=1, seed=3)
    base = CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 10\n"
    a = run_oracle(base + "run 40\n", s)
    b = run_oracle(base + "run_style respa 1\nrun 40\n", s)
    assert np.array_equal(a.x(), b.x()) and np.array_equal(a.v(), b.v())
    assert np.array_equal(np.array(a.thermo()), np.array(b.thermo()))
    c = run_oracle(base + "run_style respa 2 4\nrun 40\n", s)
    d = run_oracle(base + "run_style respa 3 2 2 bond 1 pair 3\nrun 40\n", s)      # an empty middle level changes nothing
    assert 1e-6 < np.abs(a.x() - c.x()).max() < 0.1
    assert np.abs(c.x() - d.x()).max() < 1e-9
