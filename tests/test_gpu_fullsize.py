"""BASELINE.json's full size (1M-bead chain, the default bench system) through properties that do not need the oracle
(which takes two minutes per thousand steps there; tests/parity_1m.py does that comparison by hand): conservation
laws of the NVE path, bit-reproducibility of a whole run, and the structural invariants of the bond topology after
the three LE fixes have fired twice."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1000000


@pytest.fixture(scope="module")
def system(tmp_path_factory):
    from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, write_data
    sysd = lattice_chains(N, nchains=1, seed=1, barrier_every=200)
    data = str(tmp_path_factory.mktemp("full") / "data.chain1m")
    write_data(data, sysd)
    script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.01, punload=0.01)
    return sysd, script


def _open(script, drop=()):
    from lammps_le_amd import lammps
    lmp = lammps(cmdargs=["-screen", "none"])
    for ln in script.split("\n"):
        if not any(ln.startswith(d) for d in drop):
            lmp.command(ln)
    return lmp


def test_nve_conserves_momentum_and_energy(system):
    """pair lj/cut + bond fene + fix nve only (no thermostat, no LE fixes): sum of m*v is conserved to rounding and the
    total energy to the integrator's accuracy over 300 steps incl. ~30 rebuilds."""
    sysd, script = system
    lmp = _open(script, drop=("fix 2 ", "fix loop", "fix loading", "fix unloading", "thermo_style"))
    lmp.command("run 200")                       # let the lattice start relax a little
    p0 = lmp.gather("v").sum(axis=0)
    e0 = lmp.get_thermo("etotal")
    lmp.command("run 300")
    p1 = lmp.gather("v").sum(axis=0)
    e1 = lmp.get_thermo("etotal")
    assert np.abs(p1 - p0).max() < 1e-7, (p0, p1)          # sums over 1e6 beads of O(1) velocities
    assert abs(e1 - e0) < 2e-4 * abs(e0), (e0, e1)         # per-bead total energy, dt = 0.005
    assert lmp.stat("neigh_builds") >= 10


def test_run_is_bit_reproducible_and_topology_is_consistent(system):
    """Two instances, same script, 2004 steps (both firings of every LE fix): identical to the last bit (no atomics in any
    floating-point sum, deterministic list order).  Then the invariants of the bond topology."""
    sysd, script = system
    a, b = _open(script), _open(script)
    a.command("run 2004")
    b.command("run 2004")
    for name in ("x", "v", "image", "num_bond", "bond_type", "bond_atom", "nspecial"):
        assert np.array_equal(a.gather(name), b.gather(name)), name
    nb, bt, ba = a.gather("num_bond"), a.gather("bond_type"), a.gather("bond_atom")
    ns, sp = a.gather("nspecial"), a.gather("special")
    n = len(nb)
    # every stored bond is stored by both ends with the same type (newton_bond off storage)
    half = set()
    for i in np.nonzero(nb)[0]:
        for m in range(nb[i]):
            half.add((int(bt[i, m]), int(i) + 1, int(ba[i, m])))
    assert all((t, j, i) in half for (t, i, j) in half)
    ext = {(i, j) for (t, i, j) in half if t == 2 and i < j}
    assert len(ext) > 20 and int(a.get_thermo("bonds")) == (n - 1) + len(ext) == len(half) // 2
    # backbone intact, at most one extruder anchor per bead, anchors are backbone-interior beads
    assert all((1, i, i + 1) in half for i in range(1, n))
    ends = [i for e in ext for i in e]
    assert len(ends) == len(set(ends)) and min(ends) > 1 and max(ends) < n
    # the 1-2 special block of a bead = its bond partners
    for i in list(range(0, n, 997)) + [e - 1 for e in ends]:
        assert sorted(sp[i, :ns[i, 0]]) == sorted(int(ba[i, m]) for m in range(nb[i])), i + 1
    # fix counters: extrusion moves preserve the bond count; loads minus unloads = extruders present
    assert a.extract_fix("loading", 0, 1, 1) - a.extract_fix("unloading", 0, 1, 1) == len(ext)


# ------------------------------------------------------------------------------------------------------------------
# the other two full-size configurations of BASELINE.json: 10 chains x 100k with barrier types (configs[2]) and the
# 8M-bead system = the per-GPU size of the 8 x 1M weak-scaling run (configs[4]: dense load, prob 0.01, N1 = 1000)
def _topology_invariants(lmp, n, nchains):
    """Vectorised: storage symmetry, backbone intact inside every chain and absent across chain ends, one extruder
    anchor per bead at most, 1-2 special block = bond partners, bond counter = backbone + extruders."""
    nb, bt, ba = lmp.gather("num_bond"), lmp.gather("bond_type"), lmp.gather("bond_atom")
    ns, sp = lmp.gather("nspecial"), lmp.gather("special")
    w = bt.shape[1]
    own = np.repeat(np.arange(1, n + 1, dtype=np.int64)[:, None], w, axis=1)
    live = np.arange(w)[None, :] < nb[:, None]
    t, i, j = bt[live].astype(np.int64), own[live], ba[live].astype(np.int64)
    key = (t * (n + 1) + i) * (n + 1) + j
    rkey = (t * (n + 1) + j) * (n + 1) + i
    assert np.array_equal(np.sort(key), np.sort(rkey))                 # every bond stored by both ends, same type
    per = n // nchains
    a = np.arange(1, n, dtype=np.int64)                                # bond (a, a+1) exists iff a is not a chain's last bead
    inside = (a % per != 0) | (a // per >= nchains)
    have = np.isin((1 * (n + 1) + a) * (n + 1) + a + 1, key)
    assert np.array_equal(have, inside)
    e = t == 2
    ends = np.concatenate([i[e & (i < j)], j[e & (i < j)]])
    assert len(ends) == len(np.unique(ends))
    next_ = int(e.sum()) // 2
    assert int(lmp.get_thermo("bonds")) == int(inside.sum()) + next_
    # 1-2 block of a sample of beads (every 997th and every anchor) = its bond partners
    sample = np.unique(np.concatenate([np.arange(0, n, 997), ends - 1]))
    for k in sample[:20000]:
        assert sorted(sp[k, :ns[k, 0]]) == sorted(int(ba[k, m]) for m in range(nb[k])), k + 1
    return next_


def _full_size_case(tmp_path_factory, name, n, nchains, steps, ttol=0.05, start="lattice"):
    from lammps_le_amd.synth import CHAIN_INPUT, lattice_chains, scrambled_chains, write_data
    sysd = (scrambled_chains if start == "walk" else lattice_chains)(n, nchains=nchains, seed=1, barrier_every=200)
    data = str(tmp_path_factory.mktemp(name) / ("data." + name))
    write_data(data, sysd)
    script = CHAIN_INPUT.format(data=data, n1=1000, left=2, right=3, tp=0.5, lr="4", nload=1000, pload=0.01, punload=0.01)
    a, b = _open(script), _open(script)
    a.command("run %d" % steps)
    b.command("run %d" % steps)
    for nm in ("x", "v", "image", "num_bond", "bond_type", "bond_atom", "nspecial"):
        assert np.array_equal(a.gather(nm), b.gather(nm)), nm           # bit-reproducible
    b.close()
    x = a.gather("x")
    assert np.isfinite(x).all()
    # the thermostat holds T* = 1 once the start lattice's straight runs (as long as the box edge: 100 beads at 1M, 200 at
    # 8M) have contracted to the FENE bond length; measured T(502 / 1004 / 1506 steps) = 1.30 / 1.012 / 1.002 at 1M
    assert abs(a.get_thermo("temp") - 1.0) < ttol
    next_ = _topology_invariants(a, n, nchains)
    assert a.extract_fix("loading", 0, 1, 1) - a.extract_fix("unloading", 0, 1, 1) == next_
    assert a.stat("fene_warnings") == 0
    return a, next_


def test_chains10x100k_full_size(tmp_path_factory):
    """BASELINE configs[2]: 1M beads in 10 chains with left / right / roadblock barrier beads, through the first firing
    of every LE fix twice over (2004 steps)."""
    a, next_ = _full_size_case(tmp_path_factory, "chains10x100k", 1000000, 10, 2004)
    assert next_ > 20


def test_chain8m_full_size(tmp_path_factory):
    """BASELINE configs[4] at its per-GPU size: 8M beads, dense load (prob 0.01, N1 = 1000), two firings of every fix, from
    the scrambled (melt-like) start - the start that survives long runs (the serpentine lattice's 200-bead straight runs do
    not: profiles/r02/soak_8m_lattice_start.log) - with the thermostat bound of the other full-size cases."""
    a, next_ = _full_size_case(tmp_path_factory, "walk8m", 8000000, 1, 2004, ttol=0.05, start="walk")
    assert next_ > 1000        # two load firings on a melt-like chain: thousands of (i, i+2) pairs within reach
    assert a.stat("neigh_builds") > 100
