"""Spatial decomposition (z slabs, one process per rank) against the CPU oracle.  The ranks share the single
GPU of the test box and talk through the /dev/shm test transport (RCCL refuses duplicate devices); the device
kernels, migration, ghost lists, halo updates and replicated LE fixes are exactly those of a multi-GPU run."""
import os
import pickle
import subprocess
import sys
import uuid

import numpy as np
import pytest

from systems import CHAIN_SCRIPT, lattice_chain, run_oracle
from test_gpu_le import LE, barrier_types, melted, special_sets

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(world, system, script, tmp_path):
    session = uuid.uuid4().hex[:12]
    sysfile, scriptfile, out = (os.path.join(str(tmp_path), n) for n in ("system.pkl", "script.txt", "out.npz"))
    pickle.dump(system, open(sysfile, "wb"))
    open(scriptfile, "w").write(script)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dd_worker.py"), str(r), str(world), session, sysfile,
                               scriptfile, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


def run_ranks_local(world, system, script, tmp_path):
    """Same as run_ranks, but the ranks are threads of this process talking through the in-process transport
    (stream-ordered device-to-device copies): no files, no host staging, any number of ranks on the one GPU."""
    import threading
    from lammps_le_amd import lammps
    from systems import write_data
    session = uuid.uuid4().hex[:12]
    path = os.path.join(str(tmp_path), "data.local")
    write_data(path, system)
    out, errs = [None] * world, []

    def work(rank):
        try:
            lmp = lammps(cmdargs=["-screen", "none"])
            lmp.comm_init("local", rank, world, session=session)
            for ln in script.split("\n"):
                w = ln.split("#")[0].split()
                lmp.command("read_data " + path if w and w[0] == "read_data" else ln)
            res = dict(x=lmp.gather("x"), v=lmp.gather("v"), image=lmp.gather("image"),
                       num_bond=lmp.gather("num_bond"), bond_type=lmp.gather("bond_type"), bond_atom=lmp.gather("bond_atom"),
                       nspecial=lmp.gather("nspecial"), special=lmp.gather("special"),
                       thermo=np.array([lmp.get_thermo(k) for k in ("temp", "epair", "emol", "etotal", "press", "bonds")]),
                       neigh_pairs=np.array([lmp.stat("neigh_pairs")]), builds=np.array([lmp.stat("neigh_builds")]))
            if lmp.extract_setting("angle_per_atom") > 0:
                res.update(num_angle=lmp.gather("num_angle"), angle_type=lmp.gather("angle_type"), angle_atom1=lmp.gather("angle_atom1"),
                           angle_atom2=lmp.gather("angle_atom2"), angle_atom3=lmp.gather("angle_atom3"),
                           nangles=np.array([lmp.extract_setting("nangles")]))
            for fid in ("loop", "loading", "unloading"):
                try:
                    res["f_" + fid] = np.array([lmp.extract_fix(fid, 0, 1, 0), lmp.extract_fix(fid, 0, 1, 1)])
                except Exception:
                    pass
            out[rank] = res
            lmp.close()
        except Exception as e:       # a failing rank leaves the others waiting for the transport's timeout
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for r in range(1, world):        # every rank holds the same gathered state
        assert np.array_equal(out[r]["x"], out[0]["x"]) and np.array_equal(out[r]["bond_atom"], out[0]["bond_atom"])
    return out[0]


def bond_set(nb, bt, ba):
    out = set()
    for i in np.nonzero(nb)[0]:
        for m in range(nb[i]):
            a, b = int(i) + 1, int(ba[i, m])
            out.add((int(bt[i, m]), min(a, b), max(a, b)))
    return out


@pytest.mark.parametrize("world,n,overlap,windows", [(2, 6000, 0, 1), (3, 20000, 0, 1), (2, 6000, 1, 1), (3, 20000, 0, 0),
                                                     (5, 60000, 0, 1), (2, 6000, 0, 2), (3, 20000, 0, 2)])
def test_md_across_slabs(tmp_path, world, n, overlap, windows, monkeypatch):
    """NVE + Langevin for 60 steps incl. reneighbors with migration across slab faces (one process per rank).  `windows`:
    the per-step halo goes through the peer windows (the step kernel stores into the neighbour's IPC-mapped buffer) or
    through the transport; 2 = windows with the one-launch exchange kernel a multi-GPU run takes (here the ranks share a
    GPU, where the engine would choose the two-launch form by itself)."""
    monkeypatch.setenv("LAMMPS_LE_OVERLAP", str(overlap))
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO", str(min(windows, 1)))
    monkeypatch.setenv("LAMMPS_LE_HALO_FUSED", "1" if windows == 2 else "0")
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO_VERIFY", "1" if world in (3, 5) else "0")
    s = lattice_chain(n, nchains=2, seed=21)
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 30\nrun 60\n"
    o = run_oracle(script, s)
    r = run_ranks(world, s, script, tmp_path)
    assert np.abs(r["x"] - o.x()).max() < 1e-9
    assert np.abs(r["v"] - o.v()).max() < 1e-8
    assert (r["image"] == o.image()).all()
    to = o.thermo()
    assert np.abs(r["thermo"][:5] - to[:5]).max() < 1e-9
    assert r["neigh_pairs"][0] == 2 * o.neigh_pairs()
    assert r["builds"][0] == o.neigh_builds()
    # every step that did not rebuild exchanged its halo through the windows (not in overlap mode, which keeps the transport)
    # (all of them but the steps that rebuild, the first step and the steps behind a thermo evaluation, whose positions do not
    # come out of the fused step kernel)
    nwin = int(r["window_exchanges"][0])
    if windows and not overlap:
        assert 60 - int(o.neigh_builds()) - 5 <= nwin <= 60, nwin
        assert int(r["window_mismatches"][0]) == 0          # every window halo was also sent through the transport and compared
    else:
        assert nwin == 0


def test_window_halo_verify_mode_repairs_a_corrupted_window(tmp_path, monkeypatch):
    """LAMMPS_LE_FAST_HALO_VERIFY: a window halo that differs from the transport's copy is counted and replaced (the test hook
    shifts one ghost per exchange by 0.25 sigma on every rank); the trajectory stays the oracle's."""
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO", "1")
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO_VERIFY", "1")
    monkeypatch.setenv("LAMMPS_LE_TEST_HALO_CORRUPT", "1")
    s = lattice_chain(6000, nchains=2, seed=21)
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 30\nrun 40\n"
    o = run_oracle(script, s)
    r = run_ranks(2, s, script, tmp_path)
    assert np.abs(r["x"] - o.x()).max() < 1e-9
    assert int(r["window_exchanges"][0]) > 20 and int(r["window_mismatches"][0]) == int(r["window_exchanges"][0])


@pytest.mark.parametrize("fused", [0, 1])
def test_windows_that_do_not_deliver_are_rejected_not_fatal(tmp_path, monkeypatch, fused):
    """Verify mode on a node whose peer windows do not work (test hook: no rank's "stores complete" counter reaches its
    neighbours): the first exchange gives up after 5 s, counts a mismatch and later exchanges do not wait again; the transport's
    copy of every halo wins, the trajectory stays the oracle's, and bench.py - which looks at the mismatch count after its
    pre-roll - then keeps the transport.  Outside verify mode the same condition is ERR_HALO_TIMEOUT."""
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO", "1")
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO_VERIFY", "1")
    monkeypatch.setenv("LAMMPS_LE_HALO_FUSED", str(fused))
    monkeypatch.setenv("LAMMPS_LE_TEST_HALO_MUTE", "1")
    s = lattice_chain(6000, nchains=2, seed=21)
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 30\nrun 40\n"
    o = run_oracle(script, s)
    r = run_ranks(2, s, script, tmp_path)
    assert np.abs(r["x"] - o.x()).max() < 1e-9
    assert int(r["window_exchanges"][0]) > 20 and int(r["window_mismatches"][0]) > 0


def test_le_fixes_across_slabs(tmp_path):
    """Replicated extruder table: every rank runs the same deterministic LE kernels on all-gathered positions;
    topology must be bit-exact against the 1-rank oracle."""
    n = 14000
    s = melted(n, nchains=2, seed=8, types=barrier_types(n, 13))
    # ghost shell 6.2 > longest extruder bond of this (slow-stepping) scenario: partners stay reachable
    base = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 6.2") \
        .replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 10.0 6.0 1.0 1.0")
    script = base + LE.format(n1=20, nl=10, nu=10, neutral=1, left=2, right=3, tp=0.5, lr="4",
                              lprob="prob 0.5 684474", uprob="prob 0.3 456456", rmax=0.5) + "run 50\n"
    o = run_oracle(script, s)
    r = run_ranks(2, s, script, tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set()
    ns_o, sp_o = o.special_table()
    assert special_sets(r["nspecial"], r["special"]) == special_sets(ns_o, sp_o)
    for fid in ("loop", "loading", "unloading"):
        assert r["f_" + fid][0] == o.fix_vector(fid)[0] and r["f_" + fid][1] == o.fix_vector(fid)[1]
    assert r["thermo"][5] == o.nbonds()
    assert np.abs(r["x"] - o.x()).max() < 1e-7
    assert len([b for b in o.bond_set() if b[0] == 2]) > 20


@pytest.mark.parametrize("world,n,overlap", [(2, 6000, 0), (4, 40000, 0), (5, 40000, 1), (2, 6000, 1), (3, 40000, 1),
                                              (6, 70000, 0), (7, 100000, 1)])
def test_md_across_slabs_in_process(tmp_path, world, n, overlap, monkeypatch):
    """4 and 5 slabs (interior ranks with two different neighbours) over the in-process transport; overlap=1 also
    splits every step into the beads that touch ghosts and the interior, with the halo exchange of the next step
    travelling on a second stream behind the first part (LAMMPS_LE_OVERLAP)."""
    monkeypatch.setenv("LAMMPS_LE_OVERLAP", str(overlap))
    s = lattice_chain(n, nchains=2, seed=23)
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 30\nrun 60\n"
    o = run_oracle(script, s)
    r = run_ranks_local(world, s, script, tmp_path)
    assert np.abs(r["x"] - o.x()).max() < 1e-9
    assert np.abs(r["v"] - o.v()).max() < 1e-8
    assert (r["image"] == o.image()).all()
    assert np.abs(r["thermo"][:5] - o.thermo()[:5]).max() < 1e-9
    assert r["neigh_pairs"][0] == 2 * o.neigh_pairs()
    assert r["builds"][0] == o.neigh_builds()


@pytest.mark.parametrize("overlap", [0, 1])
def test_le_fixes_across_three_slabs_in_process(tmp_path, overlap, monkeypatch):
    monkeypatch.setenv("LAMMPS_LE_OVERLAP", str(overlap))
    n = 60000      # slab width 13.8 >= two ghost shells of 6.2
    s = melted(n, nchains=3, seed=9, types=barrier_types(n, 17))
    base = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 6.2") \
        .replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 10.0 6.0 1.0 1.0")
    script = base + LE.format(n1=20, nl=10, nu=10, neutral=1, left=2, right=3, tp=0.5, lr="4",
                              lprob="prob 0.5 684474", uprob="prob 0.3 456456", rmax=0.5) + "run 50\n"
    o = run_oracle(script, s)
    r = run_ranks_local(3, s, script, tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set()
    ns_o, sp_o = o.special_table()
    assert special_sets(r["nspecial"], r["special"]) == special_sets(ns_o, sp_o)
    for fid in ("loop", "loading", "unloading"):
        assert r["f_" + fid][0] == o.fix_vector(fid)[0] and r["f_" + fid][1] == o.fix_vector(fid)[1]
    assert np.abs(r["x"] - o.x()).max() < 1e-7


def test_le_fixes_across_slabs_under_atom_sort(tmp_path):
    """`atom_modify sort 5 0` on three slabs: every rank derives the one-rank Atom::sort order from the all-gathered positions
    (the reference's own decomposed order depends on the decomposition; the engine's result is the 1-rank one), so Langevin
    draws and the LE fixes' visit order follow the oracle's single-rank run: topology and counters bit-exact."""
    n = 60000
    s = melted(n, nchains=3, seed=9, types=barrier_types(n, 17))
    base = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 6.2") \
        .replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 10.0 6.0 1.0 1.0").replace("atom_modify sort 0 0", "atom_modify sort 5 0")
    script = base + LE.format(n1=20, nl=10, nu=10, neutral=1, left=2, right=3, tp=0.5, lr="4",
                              lprob="prob 0.5 684474", uprob="prob 0.3 456456", rmax=0.5) + "run 50\n"
    o = run_oracle(script, s)
    r = run_ranks_local(3, s, script, tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set()
    for fid in ("loop", "loading", "unloading"):
        assert r["f_" + fid][0] == o.fix_vector(fid)[0] and r["f_" + fid][1] == o.fix_vector(fid)[1]
    assert np.abs(r["x"] - o.x()).max() < 1e-7
    # the order matters in this scenario: the run with the ID order gives another topology
    o0 = run_oracle(script.replace("atom_modify sort 5 0", "atom_modify sort 0 0"), s)
    assert o0.bond_set() != o.bond_set()


@pytest.mark.parametrize("world,case", [(2, "frozen-type"), (3, "langevin-subset"), (4, "two-nve"), (3, "zero"),
                                        (3, "langevin-subset+sort"), (3, "frozen-type+overlap"), (2, "frozen-type+unfused")])
def test_md_fixes_on_groups_across_slabs(tmp_path, world, case, monkeypatch):
    """fix nve / fix langevin on a group in a decomposed run (unfused kernels; the thermostat's draws go by the bead's rank among
    the members in the reference's local order, the stream segments a rank generates follow that rank table): anchors that
    never move, a thermostat on a subset, two integrators - against the oracle on 2-4 slabs."""
    n = 20000 if world <= 3 else 40000
    types = 1 + (np.arange(n) % 7 == 0).astype(np.int32)
    s = lattice_chain(n, nchains=2, seed=29, jitter=0.03, types=types)
    s["mass"] = [1.0, 1.0]
    head = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0")
    if case.endswith("+overlap"):        # (the group variant of the step kernel in its two-phase launches)
        monkeypatch.setenv("LAMMPS_LE_OVERLAP", "1")
        case = case[:-len("+overlap")]
    if case.endswith("+unfused"):
        monkeypatch.setenv("LAMMPS_LE_NO_FUSED_GROUPS", "1")
        case = case[:-len("+unfused")]
    if case == "frozen-type":
        body = "group mobile type 1\nfix 1 mobile nve\nfix 2 mobile langevin 1.0 1.0 1.0 5544\n"
    elif case.startswith("langevin-subset"):
        if case.endswith("sort"):
            head = head.replace("atom_modify sort 0 0", "atom_modify sort 7 0")
        body = "group hot id 1:%d:3 %d:%d\nfix 1 all nve\nfix 2 hot langevin 1.2 0.8 2.0 91 scale 2 2.5\n" % (n // 2, n // 2 + 100, n)
    elif case == "two-nve":
        body = "group lo id 1:%d\ngroup hi subtract all lo\nfix 1 lo nve\nfix 3 hi nve\nfix 2 hi langevin 0.8 0.8 1.0 313\n" % (n // 3)
    else:                # `zero yes`: the ranks' summed random forces are added across the slabs every step
        body = "group hot id 1:%d:2\nfix 1 all nve\nfix 2 hot langevin 1.0 1.0 1.0 99 zero yes\n" % n
    script = head + body + "thermo 20\nrun 45\nrun 25\n"
    o = run_oracle(script, s)
    r = run_ranks_local(world, s, script, tmp_path)
    assert np.abs(r["x"] - o.x()).max() < 1e-9
    assert np.abs(r["v"] - o.v()).max() < 1e-8
    assert (r["image"] == o.image()).all()
    assert np.abs(r["thermo"][:5] - o.thermo()[:5]).max() < 1e-9
    assert r["builds"][0] == o.neigh_builds()
    if case == "frozen-type":
        frozen = types == 2
        assert np.array_equal(r["x"].reshape(n, 3)[frozen], o.x()[frozen])


@pytest.mark.parametrize("world,style,le", [(2, "run_style respa 2 4", False), (3, "run_style respa 3 2 3 bond 1 pair 2", False),
                                            (3, "run_style respa 2 3", True)])
def test_respa_across_slabs(tmp_path, world, style, le):
    """run_style respa in a decomposed run (the slow path of unfused kernels): ghosts follow every move of the innermost level,
    the per-level force tables (by tag) are completed on every rank before a rebuild so that a migrating bead finds its rows
    on the new owner; trajectory, thermo, rebuild count - and with the three LE fixes the topology - against the oracle."""
    n = 20000 if world == 2 else 27000
    if le:
        n = 60000      # slab width 13.8 >= two ghost shells of 6.2
        s = melted(n, nchains=3, seed=9, types=barrier_types(n, 17))
        base = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 6.2") \
            .replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 10.0 6.0 1.0 1.0")
        body = LE.format(n1=20, nl=10, nu=10, neutral=1, left=2, right=3, tp=0.5, lr="4", lprob="prob 0.5 684474",
                         uprob="prob 0.3 456456", rmax=0.5)
    else:
        s = lattice_chain(n, nchains=2, seed=21, jitter=0.03)
        base = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0")
        body = "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n"
    script = base + body + style + "\nthermo 20\nrun 40\nrun 20\n"
    o = run_oracle(script, s)
    r = run_ranks_local(world, s, script, tmp_path)
    assert np.abs(r["x"] - o.x()).max() < 1e-8
    assert np.abs(r["v"] - o.v()).max() < 1e-7
    assert (r["image"] == o.image()).all()
    assert np.abs(r["thermo"][:5] - o.thermo()[:5]).max() < 1e-8
    assert r["builds"][0] == o.neigh_builds()
    if le:
        assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set()
        assert len([b for b in o.bond_set() if b[0] == 2]) > 5


def test_script_commands_between_runs_when_decomposed(tmp_path):
    """Host-side commands between two runs of a decomposed system (three slabs, in process): `velocity create` replaces
    the velocities on every rank's replicated host copy (the download before it is collective), periodic restart files
    are written by rank 0 during the run, `write_data` after it; trajectories as on one rank."""
    s = lattice_chain(20000, nchains=2, seed=23)
    rfile = str(tmp_path / "dd.restart")
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + (
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 20\nrun 30\n"
        "velocity all create 0.8 77123 dist gaussian loop local\nrestart 20 %s\nrun 40\n" % rfile)
    o = run_oracle(script.replace("restart 20 %s\n" % rfile, ""), s)
    r = run_ranks_local(3, s, script, tmp_path)
    assert np.abs(r["x"] - o.x()).max() < 1e-9
    assert np.abs(r["v"] - o.v()).max() < 1e-8
    assert r["builds"][0] == o.neigh_builds()
    assert os.path.exists(rfile + ".40") and os.path.exists(rfile + ".60")


def test_stock_bond_create_across_slabs(tmp_path):
    """`fix bond/create` on three slabs: each rank picks partners for the beads it owns from its own lists (ghost
    partners included, current positions from the all-gather), the partner table is completed by a max-reduction and
    the rest runs replicated like ex_load; topology as on one rank."""
    from test_gpu_le import melted
    n = 20000
    rng = np.random.RandomState(4)
    types = np.where(rng.rand(n) < 0.3, 2, 1).astype(np.int32)
    s = melted(n, nchains=2, seed=2, steps=400, types=types)
    s["ntypes"], s["mass"] = 3, [1.0, 1.0, 1.0]
    script = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0") \
        .replace("comm_modify cutoff 5.0", "comm_modify cutoff 3.0") + (
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n"
        "fix creating all bond/create 7 2 2 1.1 2 iparam 1 3 jparam 1 3 prob 0.6 8847\n"
        "fix breaking all bond/break 9 2 1.3 prob 0.5 2211\nthermo 10\nrun 45\n")
    o = run_oracle(script, s)
    r = run_ranks_local(3, s, script, tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set()
    ns_o, sp_o = o.special_table()
    assert special_sets(r["nspecial"], r["special"]) == special_sets(ns_o, sp_o)
    assert o.fix_vector("creating")[1] > 50 and o.fix_vector("breaking")[1] > 0
    assert np.abs(r["x"] - o.x()).max() < 1e-7


def test_rccl_bindings_on_one_rank():
    """The engine declares RCCL's entry points by hand (dlopen, no header): a size-1 communicator on the test GPU
    checks argument layouts and enum values through all-reduce(max), all-gather and a grouped send/recv to self."""
    import ctypes
    from lammps_le_amd import library_path
    lib = ctypes.CDLL(library_path())
    assert lib.lammps_le_rccl_selftest() == 0


@pytest.mark.parametrize("mode", ["shared", "own"])
def test_rccl_next_to_a_torch_process_group(mode):
    """bench.py --gpus N runs the engine's communicator in a process where torch.distributed (NCCL = RCCL) is alive.
    `shared`: the engine binds the RCCL copy torch loaded (one library instance, what bench.py does);
    `own`: it loads the system copy next to torch's.  Both must pass the binding self-test and leave torch working."""
    port = 29000 + os.getpid() % 2000
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_with_torch_worker.py"), str(port), mode],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    files = [ln for ln in r.stdout.split("\n") if ln.startswith("RCCL_FILES")][0].split()
    if mode == "shared":
        assert int(files[1]) == 1, files


def test_eight_slabs_with_the_bench_script(tmp_path):
    """The shape of the 8-GPU scaling run: eight z-slabs, `comm_modify cutoff 5.0`, the bench input with the three LE
    fixes firing - here on 500k beads (slabs 10.5 thick, just above two ghost shells) with all ranks on the test GPU."""
    from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains
    from systems import OracleScript
    n = 500000
    sysd = lattice_chains(n, nchains=1, seed=3, barrier_every=200)
    script = CHAIN_INPUT.format(data="data.chain", n1=10, left=2, right=3, tp=0.5, lr="4", nload=10, pload=0.2, punload=0.2) + "run 34\n"
    osc = OracleScript(dict(sysd))
    for ln in script.split("\n"):
        if not ln.startswith("thermo_style"):
            osc.line(ln)
    o = osc.o
    r = run_ranks_local(8, sysd, script.replace("thermo_style", "#thermo_style"), tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set()
    assert len([b for b in o.bond_set() if b[0] == 2]) > 10
    for fid in ("loop", "loading", "unloading"):
        assert r["f_" + fid][0] == o.fix_vector(fid)[0] and r["f_" + fid][1] == o.fix_vector(fid)[1]
    assert np.abs(r["x"] - o.x()).max() < 1e-7 and (r["image"] == o.image()).all()
    assert r["builds"][0] == o.neigh_builds() and r["neigh_pairs"][0] == 2 * o.neigh_pairs()


def test_overlap_two_ranks_one_million_beads(tmp_path, monkeypatch):
    """Regression at the shape of the round-1 crash record (gpurun_out/prof_ov2.log: SIGSEGV inside hipMemcpyAsync <-
    local_exchange <- dd_halo on comm_stream, 2 ranks, 1M beads, halo/compute overlap on, while that mode was being
    written).  Two in-process ranks, overlap on, 400 steps incl. ~40 collective rebuilds and the first LE firings would
    not: the run must finish, both ranks must hold the same gathered state, and it must be the 1-rank run's topology."""
    import threading
    from lammps_le_amd import lammps
    from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data
    monkeypatch.setenv("LAMMPS_LE_OVERLAP", "1")
    n, world = 1000000, 2
    sysd = lattice_chains(n, nchains=1, seed=1, barrier_every=200)
    data = os.path.join(str(tmp_path), "data.1m")
    write_data(data, sysd)
    script = CHAIN_INPUT.format(data=data, n1=100, left=2, right=3, tp=0.5, lr="4", nload=100, pload=0.01, punload=0.05)
    session = uuid.uuid4().hex[:12]
    out, errs = [None] * (world + 1), []

    def work(rank, nranks):
        try:
            lmp = lammps(cmdargs=["-screen", "none"])
            if nranks > 1:
                lmp.comm_init("local", rank, nranks, session=session)
            for ln in script.split("\n"):
                lmp.command(ln)
            lmp.command("run 400")
            out[rank if nranks > 1 else world] = (lmp.gather("x"), lmp.gather("num_bond"), lmp.gather("bond_atom"),
                                                  lmp.get_thermo("bonds"))
            lmp.close()
        except Exception as e:
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r, world)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    work(0, 1)                                   # the same script on one rank
    assert not errs, errs
    a, b, one = out[0], out[1], out[world]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert a[3] == one[3] > n - 1                # extruders were loaded, and as many as on one rank
    assert np.array_equal(a[1], one[1]) and np.array_equal(a[2], one[2])      # bit-exact topology
    assert np.abs(a[0] - one[0]).max() < 1e-6


def test_bench_launches_its_own_ranks():
    """`python3 bench.py --gpus 4` WITHOUT a launcher (how a driver may start the scaling run): the script
    starts its four ranks itself, rank 0 prints the one JSON line, exit code 0.  Rehearsal transport (file mailbox, all
    ranks on the test GPU), a 100k-bead system and a short pre-roll so that the test takes seconds, not minutes; the
    full-size form of the same command is recorded under profiles/r03."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["LAMMPS_LE_BENCH_SHM"] = "1"
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "20", "--warmup", "5",
                        "--workload", "walk100k", "--pre-roll", "1010"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["steps"] == 20 and out["value"] > 0
    assert len(out["per_rank"]) == 4 and sorted(r["rank"] for r in out["per_rank"]) == [0, 1, 2, 3]
    assert sum(r["owned_beads"] for r in out["per_rank"]) == 100000
    assert all(r["ghost_beads"] > 0 and r["us_per_step"] > 0 for r in out["per_rank"])
    assert out["extruders"] > 0


def test_a_failing_rank_ends_every_rank(tmp_path, monkeypatch):
    """the failure paths.  Rank 1 of a 2-process run (peer windows on) stops with an error at step 31 (test
    hook).  Rank 0 must not hang in its halo wait or in the next collective: it ends with the communicator / halo error
    within the short time-out set here, and both processes exit non-zero."""
    import time
    monkeypatch.setenv("LAMMPS_LE_FAST_HALO", "1")
    monkeypatch.setenv("LAMMPS_LE_TEST_FAIL_AT", "1:31")
    monkeypatch.setenv("LAMMPS_LE_COMM_TIMEOUT", "6")
    s = lattice_chain(6000, nchains=2, seed=21)
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 0\nrun 60\n"
    session = uuid.uuid4().hex[:12]
    sysfile, scriptfile, out = (os.path.join(str(tmp_path), n) for n in ("system.pkl", "script.txt", "out.npz"))
    pickle.dump(s, open(sysfile, "wb"))
    open(scriptfile, "w").write(script)
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dd_worker.py"), str(r), "2", session, sysfile,
                               scriptfile, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    took = time.time() - t0
    assert all(p.returncode != 0 for p in procs), "\n".join(logs)
    assert "test hook: rank 1 fails at step 31" in logs[1]
    assert ("communicator aborted" in logs[0]) or ("did not deliver its halo" in logs[0]) or ("timeout" in logs[0]), logs[0]
    assert took < 120, took


def test_aborted_communicator_refuses_further_collectives(tmp_path, monkeypatch):
    """After a run that ended in an error the handle's communicator is in the ABORTED state: the peer wakes up at once
    (no time-out), and a later run or gather on either handle raises `communicator aborted` instead of quietly
    degrading to a one-rank version of the collective."""
    import threading
    import time
    from lammps_le_amd import lammps
    from systems import write_data
    monkeypatch.setenv("LAMMPS_LE_TEST_FAIL_AT", "0:12")
    monkeypatch.setenv("LAMMPS_LE_COMM_TIMEOUT", "60")
    s = lattice_chain(6000, nchains=2, seed=21)
    path = os.path.join(str(tmp_path), "data.local")
    write_data(path, s)
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + "fix 1 all nve\nthermo 0\n"
    session = uuid.uuid4().hex[:12]
    first, later = [None, None], [None, None]

    def work(rank):
        lmp = lammps(cmdargs=["-screen", "none"])
        lmp.comm_init("local", rank, 2, session=session)
        for ln in script.split("\n"):
            w = ln.split("#")[0].split()
            lmp.command("read_data " + path if w and w[0] == "read_data" else ln)
        try:
            lmp.command("run 40")
        except Exception as e:
            first[rank] = str(e)
        for cmd in ("run 5", None):
            try:
                lmp.command(cmd) if cmd else lmp.gather("x")
            except Exception as e:
                later[rank] = (later[rank] or "") + "|" + str(e)

    t0 = time.time()
    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert time.time() - t0 < 45                       # the peer did not sit out the 60 s time-out
    assert "test hook" in first[0] and "communicator aborted" in first[1], first
    for r in range(2):
        assert later[r] is not None and later[r].count("communicator aborted") == 2, later


@pytest.mark.parametrize("ghost_margin,grouped", [(True, False), (False, False), (False, True)])
def test_langevin_stream_segments_are_skipped_not_lost(tmp_path, monkeypatch, ghost_margin, grouped):
    """Decomposed runs generate only the segments of the Langevin stream that hold draws of owned or ghost beads and jump
    over the rest (VERDICT r02 #2a).  Three slabs of a lattice-start chain (tags follow z, so each rank really skips most
    segments), 64 segments per call, 400 steps with ~40 rebuilds and migration: the trajectory must stay the oracle's -
    a bead that draws from a skipped segment would get wrong noise on its first step - and every rank must hold fewer
    segments than a call has."""
    import threading
    from lammps_le_amd import lammps
    from systems import write_data
    monkeypatch.setenv("LAMMPS_LE_RNG_SEGMENTS", "64" if ghost_margin else "100" if grouped else "200")   # (>= 256 draws each)
    monkeypatch.setenv("LAMMPS_LE_RNG_W", "32")          # short batches: many pool switches, markings and validations
    if not ghost_margin:
        # test hook: a batch marks the segments of OWNED beads only, so a bead that migrates in finds its draws missing and the
        # validation at the rebuild has to generate them late from the kept windows - the path a bead from beyond the ghost
        # shell takes in a real run
        monkeypatch.setenv("LAMMPS_LE_TEST_RNG_NO_GHOST_MARK", "1")
    s = lattice_chain(20000, nchains=2, seed=21)
    script = CHAIN_SCRIPT.replace("comm_modify cutoff 5.0", "comm_modify cutoff 2.0") + \
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\nthermo 100\nrun 400\n"
    if grouped:      # the thermostat on every other bead: segments are addressed by the rank among the group's members
        script = script.replace("fix 2 all langevin", "group hot id 1:20000:2\nfix 2 hot langevin")
    o = run_oracle(script, s)
    world = 3
    path = os.path.join(str(tmp_path), "data.local")
    write_data(path, s)
    session = uuid.uuid4().hex[:12]
    out, errs = [None] * world, []

    def work(rank):
        try:
            lmp = lammps(cmdargs=["-screen", "none"])
            lmp.comm_init("local", rank, world, session=session)
            for ln in script.split("\n"):
                w = ln.split("#")[0].split()
                lmp.command("read_data " + path if w and w[0] == "read_data" else ln)
            held, nseg, late = lmp.stat("rng_segments_held"), lmp.stat("rng_segments"), lmp.stat("rng_late_generations")
            out[rank] = (lmp.gather("x"), lmp.gather("v"), held, nseg, late)
            lmp.close()
        except Exception as e:
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert np.abs(out[0][0] - o.x()).max() < 1e-7 and np.abs(out[0][1] - o.v()).max() < 1e-6
    nseg = 64 if ghost_margin else 100 if grouped else 200
    for r in range(world):
        assert out[r][3] == nseg and 0 < out[r][2] < nseg, out[r][2:]
    if not ghost_margin:
        assert sum(out[r][4] for r in range(world)) > 0, [out[r][4] for r in range(world)]     # late generations happened


@pytest.mark.parametrize("world,style", [(2, "harmonic"), (3, "cosine")])
def test_semiflexible_chains_across_slabs(tmp_path, world, style):
    """Angles in a decomposed run (in-process transport): the ghost shell of a run with an angle style holds every bead
    within the ghost cutoff of a slab face, each rank lists the angles that move a bead it owns, the LE fixes keep the
    replicated angle tables (`ex_load ... atype 2`, angle breaking in ex_unload).  Bond topology, angle tables, angle count,
    thermo (emol = bond + angle energy, pressure with the angle virial) and trajectory against the one-rank oracle."""
    from test_gpu_angle import ANGLE_SCRIPT, semiflexible
    n = 9000 if world == 2 else 27000
    s = semiflexible(n, 3, seed=4, steps=300)
    coeffs = ("angle_coeff 1 1.5 160.0", "angle_coeff 2 1.0 100.0") if style == "harmonic" else ("angle_coeff 1 1.5", "angle_coeff 2 0.5")
    script = ANGLE_SCRIPT.replace("bond_coeff 2 5.0 10.0 1.0 1.0", "bond_coeff 2 8.0 5.0 1.0 1.0") + """angle_style %s
%s
%s
fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion 19 1 1 1 1.0 2
fix loading all ex_load 5 1 1 1.12 2 prob 0.3 684474 iparam 1 1 jparam 1 1 atype 2
fix unloading all ex_unload 6 2 0.5 prob 0.4 456456
thermo 20
run 50
""" % ((style,) + coeffs)     # (few extrusion steps: an extruder bond of a stiff chain must stay inside the 5.0 ghost shell)
    o = run_oracle(script, s)
    r = run_ranks_local(world, s, script, tmp_path)
    assert bond_set(r["num_bond"], r["bond_type"], r["bond_atom"]) == o.bond_set()
    na, at, a1, a2, a3 = o.angle_table()
    assert (r["num_angle"] == na).all()
    for name, ref in (("angle_type", at), ("angle_atom1", a1), ("angle_atom2", a2), ("angle_atom3", a3)):
        for i in np.nonzero(na)[0]:
            assert list(r[name][i, :na[i]]) == list(ref[i, :na[i]]), (name, i + 1)
    assert int(r["nangles"][0]) == o.nangles()
    assert any(a[0] == 2 for a in o.angle_set())
    assert np.abs(r["x"] - o.x()).max() < 1e-8
    to = o.thermo()
    assert np.abs(r["thermo"][:5] - to[:5]).max() < 1e-8
    assert r["builds"][0] == o.neigh_builds()
