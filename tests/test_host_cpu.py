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
    for style in ("nve", "langevin", "extrusion", "ex_load", "ex_unload"):
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
