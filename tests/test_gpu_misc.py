"""GPU edge cases of the script layer + kernels: non-uniform pair coefficients (table path), special_bonds
lj 1 1 1 (no exclusions), ex_load type conversion, error propagation (Bad FENE bond), several runs with
thermo fix keywords, write_data round trip, the lmp_le command-line driver."""
import os
import subprocess

import numpy as np
import pytest

from systems import CHAIN_SCRIPT, lattice_chain, run_oracle, run_product, write_data
from test_gpu_le import LE, barrier_types, compare, melted

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b, floor=1.0):
    a, b = np.asarray(a), np.asarray(b)
    return (np.abs(a - b) / np.maximum(np.maximum(np.abs(a), np.abs(b)), floor)).max()


def test_nonuniform_pair_coefficients_and_mixing(tmp_path):
    n = 5000
    s = lattice_chain(n, seed=31, jitter=0.08, types=1 + (np.arange(n) % 3))
    script = CHAIN_SCRIPT.replace("pair_coeff * * 1.0 1.0 1.12",
                                  "pair_coeff 1 1 1.0 1.0 1.12\npair_coeff 2 2 0.8 0.9 1.2\npair_coeff 3 3 1.5 1.05 1.3\n"
                                  "pair_coeff 1 3 0.5 1.0 1.0") + "fix 1 all nve\nrun 20\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-10
    assert abs(p.get_thermo("epair") - o.thermo()[1]) < 1e-10
    assert abs(p.get_thermo("press") - o.thermo()[4]) < 1e-9


def test_special_bonds_all_ones(tmp_path):
    s = lattice_chain(3000, seed=33, jitter=0.08)
    script = CHAIN_SCRIPT.replace("special_bonds fene", "special_bonds lj 1.0 1.0 1.0") + "run 0\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("f"), o.f()) < 1e-12
    assert p.stat("neigh_pairs") == 2 * o.neigh_pairs()


def test_ex_load_type_conversion(tmp_path):
    """iparam 1 2 / jparam 1 2: a bead that reaches its bond limit becomes type 2 (fix_ex_load.cpp:593-598)."""
    n = 3000
    s = melted(n, seed=5)
    s["ntypes"], s["mass"] = 2, [1.0, 1.0]
    script = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0") + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n" \
        "fix loading all ex_load 5 1 1 1.12 2 prob 0.7 684474 iparam 1 2 jparam 1 2\nrun 23\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert (p.gather("type") == o.types()).all() and (o.types() == 2).sum() > 10
    assert p.bond_set() == o.bond_set()
    assert relerr(p.gather("x"), o.x()) < 1e-8


def test_bad_fene_bond_is_reported(tmp_path):
    from lammps_le_amd import LammpsError
    s = lattice_chain(2000, seed=35)
    s["x"][1000] += np.array([3.2, 0.0, 0.0])          # stretch two backbone bonds far beyond 2*R0
    script = CHAIN_SCRIPT + "fix 1 all nve\n"
    p = run_product(script, s, tmp_path)
    with pytest.raises(LammpsError, match="Bad FENE bond"):   # src/MOLECULE/bond_fene.cpp:90
        p.command("run 5")
    with pytest.raises(RuntimeError, match="Bad FENE bond"):
        run_oracle(script + "run 5\n", s)


def test_multiple_runs_thermo_keywords_and_write_data(tmp_path):
    n = 4000
    s = melted(n, nchains=2, seed=7, types=barrier_types(n, 3))
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0")
    script = base + LE.format(n1=10, nl=10, nu=10, neutral=1, left=2, right=3, tp=0.5, lr="4",
                              lprob="prob 0.5 684474", uprob="prob 0.3 456456", rmax=0.5) + \
        "thermo_style custom step temp epair emol press bonds f_loop[1] f_loading[1] f_loading[2] f_unloading[2]\n" \
        "run 17\nrun 8\nrun 21\n"
    o = run_oracle(script.replace("thermo_style", "#thermo_style"), s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    # write_data -> a fresh instance reads it back: same beads, same bonds (each once), same images
    out = os.path.join(str(tmp_path), "out.data")
    p.command("write_data " + out)
    from lammps_le_amd import lammps
    q = lammps(cmdargs=["-screen", "none"])
    for ln in ("units lj", "atom_style bond", "special_bonds fene", "read_data " + out):
        q.command(ln)
    assert q.bond_set() == p.bond_set()
    assert np.array_equal(q.gather("x"), p.gather("x")) and np.array_equal(q.gather("image"), p.gather("image"))
    assert np.array_equal(q.gather("v"), p.gather("v")) and np.array_equal(q.gather("type"), p.gather("type"))


def test_command_line_driver(tmp_path):
    s = lattice_chain(1000, seed=41)
    data = os.path.join(str(tmp_path), "data.chain")
    write_data(data, s)
    script = os.path.join(str(tmp_path), "in.chain")
    open(script, "w").write(CHAIN_SCRIPT.replace("data.chain", data) +
                            "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 25\nrun 50\n")
    exe = os.path.join(ROOT, "lammps_le_amd", "lmp_le")
    r = subprocess.run([exe, "-in", script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Loop time of" in r.stdout and "timesteps/s" in r.stdout
    rows = [ln.split() for ln in r.stdout.split("\n") if ln.split()[:1] in (["0"], ["25"], ["50"])]
    assert len(rows) == 3
    o = run_oracle(CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 25\nrun 50\n", s)
    h = o.thermo_history()
    for row, ref in zip(rows, h):
        assert abs(float(row[1]) - ref[1]) < 1e-6 and abs(float(row[5]) - ref[5]) < 1e-5
    bad = subprocess.run([exe, "-in", "/nonexistent/in.x"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "Cannot open input script" in bad.stdout


@pytest.mark.parametrize("env", [{"LAMMPS_LE_RNG_MODE": "block"}, {"LAMMPS_LE_RNG_W": "3"}, {"LAMMPS_LE_RNG_W": "64"}])
def test_langevin_generators_agree(tmp_path, env, monkeypatch):
    """The Langevin stream is one serial RanMars (fix_langevin.cpp:670-674); the batch generator (whole calls per
    wavefront, W calls ahead), the same with tiny batches (many pool switches, several runs) and the block-parallel
    per-call generator must all hand the step kernel the same draws as the oracle's serial generator."""
    s = lattice_chain(2500, seed=41, jitter=0.08)
    script = CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n"
    o = run_oracle(script + "run 37\nrun 5\nrun 1\n", s)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    p = run_product(script + "run 37\nrun 5\nrun 1\n", s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9
    assert relerr(p.gather("v"), o.v()) < 1e-8


def _read_dump(path):
    """-> list of (timestep, columns, rows) for a text dump with one or several snapshots"""
    out, lines, k = [], open(path).read().split("\n"), 0
    while k < len(lines) and lines[k].startswith("ITEM: TIMESTEP"):
        step = int(lines[k + 1])
        assert lines[k + 2].startswith("ITEM: NUMBER OF")
        n = int(lines[k + 3])
        assert lines[k + 4] == "ITEM: BOX BOUNDS pp pp pp"
        head = lines[k + 8].split()
        cols = head[2:]
        rows = [ln.split() for ln in lines[k + 9:k + 9 + n]]
        assert all(len(r) == len(cols) for r in rows) and not any(ln.endswith(" ") for ln in lines[k + 9:k + 9 + n])
        out.append((step, cols, rows))
        k += 9 + n
    return out


def test_dump_custom_atom_and_local(tmp_path):
    """dump custom / atom / local (+ compute property/local) in the reference's text formats (dump_custom.cpp:510-527,
    dump_local.cpp:255-279): snapshots at step 0 and every N steps, values of THAT step (x, v and f), bonds listed once
    from the lower ID - the contact-map input of a loop-extrusion run."""
    n = 3000
    s = melted(n, seed=5)
    s["ntypes"], s["mass"] = 4, [1.0] * 4
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0")
    le = LE.format(n1=7, nl=5, nu=5, neutral=1, left=2, right=3, tp=0.5, lr="4", lprob="prob 0.6 101", uprob="prob 0.3 202",
                   rmax=1.3)
    d1, d2, d3 = (str(tmp_path / f) for f in ("atoms.dump", "bonds.*.dump", "scaled.dump"))
    dumps = ("compute pl all property/local btype batom1 batom2\n"
             "dump 1 all custom 10 %s id type x y z ix iy iz vx vy vz fx fy fz xu\n"
             "dump 2 all local 10 %s index c_pl[1] c_pl[2] c_pl[3]\n"
             "dump 3 all atom 20 %s\ndump_modify 1 sort id\n" % (d1, d2, d3))
    o = run_oracle(base + le + "run 20\n", s)
    p = run_product(base + le + dumps + "run 20\n", s, tmp_path)
    snaps = _read_dump(d1)
    assert [st for st, _, _ in snaps] == [0, 10, 20]
    step, cols, rows = snaps[-1]
    assert cols == "id type x y z ix iy iz vx vy vz fx fy fz xu".split()
    a = np.array(rows, dtype=float)
    assert (a[:, 0] == np.arange(1, n + 1)).all()
    L = s["box"][0][1] - s["box"][0][0]
    g6 = 6e-6      # %g = 6 significant digits
    assert relerr(a[:, 2:5], o.x()) < g6 and relerr(a[:, 8:11], o.v()) < g6 and relerr(a[:, 11:14], o.f()) < g6
    assert (a[:, 5:8] == o.image()).all() and relerr(a[:, 14], o.x()[:, 0] + o.image()[:, 0] * L) < g6
    # step-0 snapshot = the input state
    a0 = np.array(snaps[0][2], dtype=float)
    d0 = a0[:, 2:5] - s["x"]                     # the input may lie outside the box; the snapshot is wrapped (setup)
    assert np.abs((d0 + L / 2) % L - L / 2).max() < 1e-4
    # bonds: one file per snapshot, every bond once from its lower ID, in atom order
    bsn = _read_dump(d2.replace("*", "20"))
    assert len(bsn) == 1 and bsn[0][1] == ["index", "c_pl[1]", "c_pl[2]", "c_pl[3]"]
    b = np.array(bsn[0][2], dtype=float).astype(int)
    assert (b[:, 0] == np.arange(1, len(b) + 1)).all() and (b[:, 2] < b[:, 3]).all() and (np.diff(b[:, 2]) >= 0).all()
    assert {(t, i, j) for _, t, i, j in b} == o.bond_set() and len(b) == o.nbonds()
    assert os.path.exists(d2.replace("*", "0")) and os.path.exists(d2.replace("*", "10"))
    sc = _read_dump(d3)
    assert [st for st, _, _ in sc] == [0, 20] and sc[0][1] == ["id", "type", "xs", "ys", "zs"]
    xs = np.array(sc[-1][2], dtype=float)[:, 2:5]
    assert relerr(xs, (o.x() - s["box"][0][0]) / L, floor=0.1) < 1e-5
    # dumping does not perturb the trajectory
    assert np.abs(p.gather("x") - o.x()).max() < 1e-9 and p.bond_set() == o.bond_set()


def test_dumps_on_a_group(tmp_path):
    """`dump ID <group> ...`: rows for the members only (`mask[i] & groupbit`, src/dump_custom.cpp:607-612, dump_atom.cpp:356,
    dump_dcd.cpp:69,200); `compute ID <group> property/local` lists a bond when both its atoms are members
    (src/compute_property_local.cpp:477-480)."""
    import struct
    n = 1200
    s = lattice_chain(n, seed=4)
    d1, d2, d3, d4 = (str(tmp_path / f) for f in ("g.dump", "g.atom", "g.dcd", "gb.dump"))
    script = CHAIN_SCRIPT + ("group g id 100:700:3\ngroup h id 1:400\nfix 1 all nve\n"
                             "compute pl h property/local btype batom1 batom2\n"
                             "dump 1 g custom 10 %s id type x y z\ndump 2 g atom 10 %s\ndump 3 g dcd 10 %s\n"
                             "dump 4 all local 10 %s index c_pl[2] c_pl[3]\nrun 10\n" % (d1, d2, d3, d4))
    p = run_product(script, s, tmp_path)
    ids = np.arange(100, 701, 3)
    x = p.gather("x").reshape(n, 3)
    for path in (d1, d2):
        snaps = _read_dump(path)
        assert [st for st, _, _ in snaps] == [0, 10]
        a = np.array(snaps[-1][2], dtype=float)
        assert (a[:, 0] == ids).all()
    assert relerr(np.array(_read_dump(d1)[-1][2], dtype=float)[:, 2:5], x[ids - 1]) < 6e-6
    p.command("undump 3")
    b = open(d3, "rb").read()
    off = 100 + 160
    assert struct.unpack("<4i", b[off:off + 16]) == (164, 4, len(ids), 4)
    last = b[-3 * (8 + 4 * len(ids)):]
    xs = np.frombuffer(last[4:4 + 4 * len(ids)], dtype="<f4")
    assert np.abs(xs - x[ids - 1, 0].astype(np.float32)).max() == 0.0
    rows = np.array(_read_dump(d4)[-1][2], dtype=float).astype(int)
    assert len(rows) == 399 and (rows[:, 1] == np.arange(1, 400)).all() and (rows[:, 2] == np.arange(2, 401)).all()
    from lammps_le_amd import LammpsError
    with pytest.raises(LammpsError, match="Could not find dump group ID"):
        p.command("dump 9 nobody atom 10 %s" % d2)
    with pytest.raises(LammpsError, match="Could not find compute group ID"):
        p.command("compute c9 nobody property/local btype")


def test_dump_dcd(tmp_path):
    """`dump ID all dcd N file` (+ `dump_modify unwrap yes`): CHARMM/NAMD DCD as src/dump_dcd.cpp writes it - header with
    the snapshot count patched after every frame, unit-cell record, float32 x / y / z records in atom-ID order."""
    import struct
    n = 1500
    s = lattice_chain(n, seed=9)
    wrapped, unwrapped = str(tmp_path / "w.dcd"), str(tmp_path / "u.dcd")
    script = CHAIN_SCRIPT + ("fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 77\n"
                             "dump 1 all dcd 20 %s\ndump 2 all dcd 20 %s\ndump_modify 2 unwrap yes\nrun 40\n" % (wrapped, unwrapped))
    p = run_product(script, s, tmp_path)
    x, img = p.gather("x").reshape(n, 3), p.gather("image").reshape(n, 3)
    prd = s["box"][:, 1] - s["box"][:, 0]
    p.command("undump 1")
    p.command("undump 2")

    def read(path):
        b = open(path, "rb").read()
        assert struct.unpack("<i4s", b[:8]) == (84, b"CORD")
        nfile, start, skip, nstep = struct.unpack("<4i", b[8:24])
        assert struct.unpack("<f", b[44:48])[0] == np.float32(0.005) and struct.unpack("<i", b[48:52])[0] == 1
        assert struct.unpack("<2i", b[84:92]) == (24, 84)
        assert struct.unpack("<2i", b[92:100]) == (164, 2) and b[100:117] == b"Written by LAMMPS"
        off = 100 + 160
        assert struct.unpack("<4i", b[off:off + 16]) == (164, 4, n, 4)
        off += 16
        frames = []
        for _ in range(nfile):
            assert struct.unpack("<i", b[off:off + 4])[0] == 48
            cell = struct.unpack("<6d", b[off + 4:off + 52])
            off += 56
            xyz = []
            for d in range(3):
                assert struct.unpack("<i", b[off:off + 4])[0] == 4 * n
                xyz.append(np.frombuffer(b[off + 4:off + 4 + 4 * n], dtype="<f4"))
                off += 8 + 4 * n
            frames.append((cell, np.stack(xyz, axis=1)))
        assert off == len(b)
        return (nfile, start, skip, nstep), frames

    hw, fw = read(wrapped)
    hu, fu = read(unwrapped)
    assert hw == (3, 0, 20, 40) and hu == (3, 0, 20, 40)
    assert fw[0][0] == (prd[0], 0.0, prd[1], 0.0, 0.0, prd[2])
    assert np.abs(fw[0][1] - (s["x"] % prd)).max() < 2e-5 or np.abs(fw[0][1] - s["x"]).max() < 2e-5
    assert np.abs(fw[2][1] - x.astype(np.float32)).max() == 0.0
    assert np.abs(fu[2][1] - (x + img * prd).astype(np.float32)).max() == 0.0


def test_restart_is_bit_continuous(tmp_path):
    """write_restart / read_restart keep the RanMars streams of fix langevin and of the three LE fixes (the reference
    does not): `run 25; write_restart` + a NEW instance `read_restart; fix ...; run 25` is bit-identical to
    `run 25; run 25` in one instance - positions, velocities, images, types, bonds, special lists, fix counters."""
    from lammps_le_amd import lammps
    n = 3000
    s = melted(n, seed=5, types=barrier_types(n, 7))
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0")
    le = LE.format(n1=6, nl=5, nu=7, neutral=1, left=2, right=3, tp=0.5, lr="4", lprob="prob 0.6 101", uprob="prob 0.3 202",
                   rmax=1.3)
    a = run_product(base + le + "run 25\nrun 25\n", s, tmp_path)
    o = run_oracle(base + le + "run 25\nrun 25\n", s)
    rfile = str(tmp_path / "state.restart")
    b1 = run_product(base + le + "run 25\nwrite_restart %s\n" % rfile, s, tmp_path)
    b1.close()
    b = lammps(cmdargs=["-screen", "none"])
    b.command("read_restart " + rfile)
    for ln in (le + "run 25\n").split("\n"):
        b.command(ln)
    for name in ("x", "v", "image", "type", "num_bond", "bond_type", "bond_atom", "nspecial", "special"):
        assert np.array_equal(a.gather(name), b.gather(name)), name
    for fid in ("loop", "loading", "unloading"):
        assert a.extract_fix(fid, 0, 1, 0) == b.extract_fix(fid, 0, 1, 0) and a.extract_fix(fid, 0, 1, 1) == b.extract_fix(fid, 0, 1, 1)
    assert b.bond_set() == o.bond_set() and np.abs(b.gather("x") - o.x()).max() < 1e-9
    assert len([t for t in o.bond_set() if t[0] == 2]) > 5
    # a restart file cannot be read into a system that already has a box
    from lammps_le_amd import LammpsError
    with pytest.raises(LammpsError, match="Cannot read_restart after simulation box is defined"):
        b.command("read_restart " + rfile)


def test_periodic_restart_files(tmp_path):
    """`restart N root` writes root.N, root.2N, ... DURING a run (state of that step, incl. the generator positions):
    reading root.25, written inside `run 50`, and running 25 more steps equals `run 25; run 25` bit for bit; the run
    that wrote the files is the same run to FP tolerance; `restart N a b` alternates two files."""
    from lammps_le_amd import lammps
    n = 3000
    s = melted(n, seed=5, types=barrier_types(n, 7))
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0")
    le = LE.format(n1=6, nl=5, nu=7, neutral=1, left=2, right=3, tp=0.5, lr="4", lprob="prob 0.6 101", uprob="prob 0.3 202",
                   rmax=1.3)
    a = run_product(base + le + "run 25\nrun 25\n", s, tmp_path)
    root = str(tmp_path / "poly.restart")
    w = run_product(base + le + "restart 25 %s\nrun 50\n" % root, s, tmp_path)
    plain = run_product(base + le + "run 50\n", s, tmp_path)
    # (a step that writes a file runs the unfused kernels, as a thermo step does: same arithmetic, other rounding)
    assert np.abs(w.gather("x") - plain.gather("x")).max() < 1e-9 and w.bond_set() == plain.bond_set()
    assert os.path.exists(root + ".25") and os.path.exists(root + ".50")
    b = lammps(cmdargs=["-screen", "none"])
    b.command("read_restart " + root + ".25")
    for ln in (le + "run 25\n").split("\n"):
        b.command(ln)
    for name in ("x", "v", "image", "type", "num_bond", "bond_type", "bond_atom", "nspecial", "special"):
        assert np.array_equal(a.gather(name), b.gather(name)), name
    for fid in ("loop", "loading", "unloading"):
        assert a.extract_fix(fid, 0, 1, 1) == b.extract_fix(fid, 0, 1, 1)
    fa, fb = str(tmp_path / "t.a"), str(tmp_path / "t.b")
    w.command("restart 10 %s %s" % (fa, fb))
    w.command("run 30")
    assert os.path.exists(fa) and os.path.exists(fb)
    c = lammps(cmdargs=["-screen", "none"])
    c.command("read_restart " + fa)          # third write (step 80) went to file a again
    assert c.get_thermo("step") == 80
    w.command("restart 0")


@pytest.mark.parametrize("n", [4000, 70000])
def test_neighbor_table_overflow_is_recovered(tmp_path, monkeypatch, n):
    """A list that does not fit the ELL table: at setup the build is repeated at once; inside the loop the step kernel
    has already been enqueued behind the build (deferred check) - it must leave the state untouched, and the host must
    grow the table, rebuild and launch the step again.  Test hook: the table is shrunk to 4 entries before build 0.
    70000 beads: the step kernel shape that also bins the new positions (the relaunch has to bin them again)."""
    monkeypatch.setenv("LAMMPS_LE_TEST_OVERFLOW_AT", "0")
    s = lattice_chain(n, seed=11, jitter=0.08)
    script = CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nrun 40\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    assert relerr(p.gather("x"), o.x()) < 1e-9 and relerr(p.gather("v"), o.v()) < 1e-8
    assert p.stat("neigh_builds") == o.neigh_builds() and p.stat("neigh_pairs") == 2 * o.neigh_pairs()
