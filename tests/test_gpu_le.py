"""GPU parity tests of the USER-LE fixes: bond topology must be BIT-EXACT against the CPU oracle's
literal serial restatement of fix_extrusion / fix_ex_load / fix_ex_unload on identical RNG seeds
(positions within FP tolerance).  Covers: adjacent-candidate runs in ex_load, extruder collisions and
stalling, left/right/roadblock barriers with through_prob in {0, 0.5, 1}, chain ends (multi-chain),
absent roadblock type, extruder bonds that straddle a periodic face (double bond-list entries)."""
import numpy as np
import pytest

from systems import CHAIN_SCRIPT, lattice_chain, run_oracle, run_product

pytestmark = pytest.mark.gpu

_cache = {}


def melted(n, nchains=1, seed=1, steps=1500, types=None):
    """A relaxed configuration (oracle MD from the lattice start), cached per parameter set."""
    key = (n, nchains, seed, steps)
    if key not in _cache:
        s = lattice_chain(n, nchains=nchains, seed=seed)
        o = run_oracle(CHAIN_SCRIPT + "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 %d\nrun %d\n" % (777 + seed, steps), s)
        s["x"], s["v"], s["image"] = o.x(), o.v(), o.image()
        _cache[key] = s
    s = dict(_cache[key])
    if types is not None:
        s["type"] = np.asarray(types, dtype=np.int32)
        s["ntypes"] = int(s["type"].max())
        s["mass"] = [1.0] * s["ntypes"]
    return s


def special_sets(ns, sp):
    out = []
    for i in range(len(ns)):
        a, b, c = ns[i]
        out.append((frozenset(sp[i, :a]), frozenset(sp[i, a:b]), frozenset(sp[i, b:c])))
    return out


def compare(p, o, fix_ids):
    assert p.bond_set() == o.bond_set()
    nb_p, nb_o = p.gather("num_bond"), o.bond_table()[0]
    assert (nb_p == nb_o).all()
    # stored order of each bead's bond slots is part of the state the fixes read
    assert (p.gather("bond_atom") * (np.arange(p.gather("bond_atom").shape[1])[None, :] < nb_p[:, None]) ==
            o.bond_table()[2] * (np.arange(o.bond_table()[2].shape[1])[None, :] < nb_o[:, None])).all()
    assert (p.gather("type") == o.types()).all()
    ns_o, sp_o = o.special_table()
    assert (p.gather("nspecial") == ns_o).all()
    assert special_sets(p.gather("nspecial"), p.gather("special")) == special_sets(ns_o, sp_o)
    assert p.get_thermo("bonds") == o.nbonds()
    for fid in fix_ids:
        assert p.extract_fix(fid, 0, 1, 0) == o.fix_vector(fid)[0], fid
        assert p.extract_fix(fid, 0, 1, 1) == o.fix_vector(fid)[1], fid
    err = np.abs(p.gather("x") - o.x()).max()
    assert err < 1e-7, err


LE = """fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion {n1} {neutral} {left} {right} {tp} 2 {lr}
fix loading all ex_load {nl} 1 1 1.12 2 {lprob} iparam 1 1 jparam 1 1
fix unloading all ex_unload {nu} 2 {rmax} {uprob}
thermo 10
"""


def le_script(n1=10, nl=10, nu=10, neutral=1, left=2, right=3, tp=1.0, lr="4", lprob="prob 0.5 684474",
              uprob="prob 0.3 456456", rmax=0.5):
    # soft, long FENE for the extruder bond: stepping every few steps (far faster than any physical N1)
    # must not run into the reference's own "Bad FENE bond" abort
    base = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0")
    return base + LE.format(n1=n1, nl=nl, nu=nu, neutral=neutral, left=left, right=right, tp=tp, lr=lr,
                                    lprob=lprob, uprob=uprob, rmax=rmax)


def barrier_types(n, seed, frac=0.15):
    rng = np.random.RandomState(seed)
    t = np.ones(n, dtype=np.int32)
    pick = rng.rand(n) < frac
    t[pick] = rng.randint(2, 5, size=pick.sum())
    t[0] = t[-1] = 1
    return t


@pytest.mark.parametrize("tp", [1.0, 0.5, 0.0])
def test_le_cycle_with_barriers(tmp_path, tp):
    """extrusion + ex_load + ex_unload over several firings each; 15% barrier beads of the three kinds."""
    n = 3000
    s = melted(n, types=barrier_types(n, 5))
    script = le_script(tp=tp).replace("pair_coeff * * 1.0 1.0 1.12", "pair_coeff * * 1.0 1.0 1.12") + "run 64\n"
    # ex_load only initiates between type-1 beads; barrier types still get LJ/FENE parameters via wildcards
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    assert len([b for b in o.bond_set() if b[0] == 2]) > 5


@pytest.mark.parametrize("lr,tp,frac", [("2", 0.5, 0.3), ("3", 0.7, 0.3), ("2", 0.3, 0.6), ("3", 1.0, 0.15)])
def test_roadblock_type_equal_to_a_barrier_type(tmp_path, lr, tp, frac):
    """`ctcf_left_right` = the left (or right) barrier type: `can()` tests such a bead twice on that side - `type != blk ||
    p > U()` and `type != ctcf_left_right || p > U()`, the second draw only when the first let the extruder through
    (fix_extrusion.cpp:413-429) - so the number of draws a listing consumes depends on its own first draw.  Topology,
    counters and - through the later firings of the same stream - the stream position against the oracle."""
    n = 3000
    s = melted(n, types=barrier_types(n, 11, frac=frac))
    script = le_script(n1=4, nl=5, nu=9, tp=tp, lr=lr) + "run 90\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    assert len([b for b in o.bond_set() if b[0] == 2]) > 5


def test_stock_bond_break_in_place_of_ex_unload(tmp_path):
    """`fix bond/break` (src/MC) = the ex_unload text firing at step % N == 0: it then acts BEFORE the extrusion step
    (N+1) and the loading (N+3) of the same cycle, on the bond list of the last reneighboring."""
    n = 3000
    s = melted(n, types=barrier_types(n, 7))
    script = le_script().replace("fix unloading all ex_unload 10 2 0.5 prob 0.3 456456",
                                 "fix unloading all bond/break 10 2 0.5 prob 0.3 456456") + "run 64\n"
    assert "bond/break" in script
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    assert o.fix_vector("unloading")[1] > 0          # something was broken
    # and it is not the ex_unload schedule in disguise
    o2 = run_oracle(le_script() + "run 64\n", s)
    assert o2.bond_set() != o.bond_set()


@pytest.mark.parametrize("prob", ["", "prob 0.5 8847"])
def test_stock_bond_create_sticky_beads(tmp_path, prob):
    """`fix bond/create` (src/MC): any two type-2 beads closer than 1.1 may bond (closest mutual partner wins), one bond
    per bead, bonded beads become type 3; `fix bond/break` opens stretched ones again (it does not lower the creator's
    bond count: a bead that was bonded once stays type 3, as in the reference)."""
    n = 4000
    rng = np.random.RandomState(3)
    types = np.where(rng.rand(n) < 0.3, 2, 1).astype(np.int32)
    s = melted(n, types=types)
    s["ntypes"], s["mass"] = 3, [1.0, 1.0, 1.0]
    script = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0") + (
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n"
        "fix creating all bond/create 10 2 2 1.1 2 iparam 1 3 jparam 1 3 %s\n"
        "fix breaking all bond/break 15 2 1.3 prob 0.5 2211\nthermo 10\nrun 65\n" % prob)
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("creating", "breaking"))
    assert o.fix_vector("creating")[1] > 20 and o.fix_vector("breaking")[1] > 0
    assert (o.types() == 3).sum() >= 2 * o.fix_vector("creating")[1] - 2 * o.fix_vector("breaking")[1]


def _stock_seeds():
    import os
    v = os.environ.get("LE_STOCK_SEEDS")          # "start:stop": one-off wider sweep
    return range(*[int(a) for a in v.split(":")]) if v else range(6)


@pytest.mark.parametrize("seed", _stock_seeds())
def test_stock_bond_create_randomised(tmp_path, seed):
    """bond/create between two DIFFERENT types with unequal bond limits and new types, random cadence and probability;
    natural reneighbor schedule (the fix scans the pair list of the last build)."""
    rng = np.random.RandomState(100 + seed)
    n = 3000
    u = rng.rand(n)
    types = np.where(u < 0.2, 2, np.where(u < 0.45, 3, 1)).astype(np.int32)
    s = melted(n, seed=1 + seed % 2, types=types)
    s["ntypes"], s["mass"] = 5, [1.0] * 5
    imax, jmax = 1, 1      # (two extra bonds per bead would overflow the 32-entry special lists the device handles)
    prob = "" if seed % 3 == 0 else "prob %.2f %d" % (rng.uniform(0.2, 0.9), rng.randint(1, 900000))
    script = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0") \
        .replace("1 extra bond per atom", "2 extra bond per atom") + (
        "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 %d\n"
        "fix creating all bond/create %d 2 3 %.3f 2 iparam %d 4 jparam %d 5 %s\n"
        "fix breaking all bond/break %d 2 %.2f prob 0.4 %d\nthermo 10\nrun %d\n"
        % (rng.randint(1, 900000), rng.randint(3, 12), rng.uniform(1.0, 1.12), imax, jmax, prob, rng.randint(4, 15),
           rng.uniform(1.2, 1.5), rng.randint(1, 900000), 70))
    s["extra_bond"], s["extra_special"] = 2, 26
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("creating", "breaking"))
    assert o.fix_vector("creating")[1] > 5


def test_dense_extruders_collide_and_stall(tmp_path):
    """No prob keyword -> every eligible (i,i+2) pair loads (first-order recurrence over runs of candidates);
    extruders then collide head-on and stall; nothing unloads."""
    n = 4000
    s = melted(n, nchains=4, seed=2)
    # one atom type: neutral = left = right = 1, so every move draws a barrier RNG value (through_prob 1.0 passes)
    script = le_script(n1=5, nl=20, nu=1000, left=1, right=1, lprob="", uprob="", lr="") + "run 58\n"
    # (stalled, stretched extruder bonds may grow past half the small test box: product and reference both keep the
    # partner image that was the closest one at the last reneighbor, src/ntopo_bond_all.cpp:52-73)
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    ext = [b for b in o.bond_set() if b[0] == 2]
    assert len(ext) > 100 and max(b[2] - b[1] for b in ext) >= 6


def test_small_periodic_box_straddling_bonds(tmp_path):
    """~12 sigma box: many extruder bonds cross a periodic face -> listed from both ends
    (ntopo_bond_all.cpp:66-67), barrier RNG drawn twice, raw-coordinate distances."""
    n = 1500
    s = melted(n, seed=3, types=barrier_types(n, 9, frac=0.3))
    script = le_script(n1=4, nl=8, nu=8, tp=0.5, lprob="prob 0.8 684474", uprob="prob 0.2 456456", rmax=1.5) + "run 70\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))


def test_step_by_step_runs(tmp_path):
    """`run 1` repeated: every run re-runs setup() (Langevin draws, reneighbor) in both engines."""
    n = 2000
    s = melted(n, nchains=2, seed=4, types=barrier_types(n, 11))
    script = le_script(n1=3, nl=6, nu=6, tp=0.5)
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    for step in range(30):
        o.run(1)
        p.command("run 1")
        assert p.bond_set() == o.bond_set(), step + 1
    compare(p, o, ("loop", "loading", "unloading"))


def test_readme_parameters_short(tmp_path):
    """README.md:17,33-34 parameter set (17500 / 7000 / prob 0.001): firings at steps 1, 2, 3."""
    n = 20000
    s = melted(n, seed=6, steps=600)
    script = CHAIN_SCRIPT + """fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loop all extrusion 17500 1 2 3 1.0 2 4
fix loading all ex_load 7000 1 1 1.12 2 prob 0.001 684474 iparam 1 1 jparam 1 1
fix unloading all ex_unload 7000 2 0.5 prob 0.001 456456
run 12
""".replace("pair_coeff * * 1.0 1.0 1.12", "pair_coeff * * 1.0 1.0 1.12")
    s["ntypes"] = 4
    s["mass"] = [1.0] * 4
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))


@pytest.mark.parametrize("special,expect_loads", [("lj 0 0 1", False), ("lj 0 1 1", True), ("lj 0 0 1 coul 0 1 1", True)])
def test_ex_load_sees_only_pairs_of_the_pair_list(tmp_path, special, expect_loads):
    """ex_load's candidates are ENTRIES of the pair list (fix_ex_load.cpp:427-451 walks list->firstneigh, an NPairCopy
    of the pair list).  `special_bonds lj 0 0 1`: the 1-3 level has lj = coul = 0 -> every (i, i+2) pair is dropped at the
    build (npair_half_bin_newtoff.cpp:103-112) -> nothing is ever loaded.  With a non-zero coul weight on that level the
    entries stay in the list (special_flag 2, factor_lj 0) and ex_load loads again, although the LJ forces are the same."""
    n = 3000
    s = melted(n)
    script = CHAIN_SCRIPT.replace("special_bonds fene", "special_bonds " + special) + """fix 1 all nve
fix 2 all langevin 1.0 1.0 1.0 904297
fix loading all ex_load 10 1 1 1.12 2 prob 0.9 684474 iparam 1 1 jparam 1 1
thermo 10
run 24
"""
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loading",))
    loaded = o.fix_vector("loading")[1]
    assert (loaded > 0) == expect_loads, loaded


@pytest.mark.parametrize("sortfreq,tp", [(1000, 0.5), (7, 0.5), (3, 0.0), (5, 1.0)])
def test_le_cycle_under_atom_sort(tmp_path, sortfreq, tp):
    """with the reference's default `atom_modify sort N` the local index is no longer ID - 1, and everything
    order-dependent in the LE fixes follows the SORTED order: which end lists an extruder bond and in which order the
    listings are visited (barrier draws, closest-wins ties), the greedy `partner[mid]` rule of the ex_load scan, which
    bead gets which draw.  The oracle permutes its arrays like Atom::sort does and loops over local indices; the engine
    keeps the arrays and carries the local index (crank) instead.  Topology must still be bit-exact."""
    n = 3000
    s = melted(n, types=barrier_types(n, 11))
    script = le_script(tp=tp, n1=4, nl=5, nu=6).replace("atom_modify sort 0 0", "atom_modify sort %d 0" % sortfreq) + "run 64\n"
    o = run_oracle(script, s)
    assert (o.local_order() != np.arange(1, n + 1)).any()          # the oracle really is in a sorted order
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    assert len([b for b in o.bond_set() if b[0] == 2]) > 5


def test_data_file_order_is_the_local_order(tmp_path):
    """Without Atom::sort the reference's local order is the order of the data file's Atoms section (1 rank), not the ID
    order: a shuffled file must give every bead the draws of ITS file position."""
    n = 2000
    s = melted(n)
    s["file_order"] = np.random.RandomState(5).permutation(n)
    script = le_script(left=1, right=1, lr="") + "run 40\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))


@pytest.mark.parametrize("sort", ["0 0", "9 0"])
def test_ex_load_visit_order_with_newton_pair_on(tmp_path, sort):
    """`newton on off` (pair on, bond off): the reference's half/bin/newton list stores an owned-owned pair under the atom
    whose neighbor bin comes first in (z, y, x) order (the lower local index inside a bin), so the ex_load scan meets the
    (i, i+2) pairs in an order set by bin geometry - and its `partner[mid]` rule depends on that order."""
    n = 3000
    s = melted(n, types=barrier_types(n, 17))
    script = le_script(n1=4, nl=5, nu=6, lprob="prob 0.8 684474").replace("newton off", "newton on off") \
        .replace("atom_modify sort 0 0", "atom_modify sort " + sort) + "run 48\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    # and the order really matters: the same run with the newton-off order differs (without Atom::sort; right after a
    # sort the local order IS the bin order - the sort bins are the neighbor bins - and the two rules coincide)
    if sort == "0 0":
        o2 = run_oracle(script.replace("newton on off", "newton off"), s)
        assert o2.bond_set() != o.bond_set()


def test_le_fixes_refuse_newton_bond_on(tmp_path):
    from lammps_le_amd import LammpsError
    s = melted(3000)
    with pytest.raises(LammpsError, match="newton_bond off"):
        run_product(le_script(left=1, right=1, lr="").replace("newton off", "newton on") + "run 1\n", s, tmp_path)


def test_ex_load_atype_without_an_angle_style_has_no_effect(tmp_path):
    """fix_ex_load.cpp:236-243: `atype` creates angles only when the script defined an angle style; without one (this
    engine has none) the keyword is accepted and the run is the one without it."""
    n = 3000
    s = melted(n, types=barrier_types(n, 5))
    base = le_script(tp=0.5) + "run 44\n"
    with_kw = base.replace("iparam 1 1 jparam 1 1", "iparam 1 1 jparam 1 1 atype 1 dtype 0 itype 2")
    assert with_kw != base
    o = run_oracle(with_kw, s)
    p = run_product(with_kw, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    q = run_product(base, s, tmp_path)
    assert np.array_equal(p.gather("x"), q.gather("x")) and np.array_equal(p.gather("bond_atom"), q.gather("bond_atom"))
    from lammps_le_amd import LammpsError
    with pytest.raises(LammpsError, match="Illegal fix ex_load"):
        run_product(base.replace("iparam 1 1 jparam 1 1", "iparam 1 1 jparam 1 1 atype -1"), s, tmp_path)


# ------------------------------------------------------------------------------------------------------------------
# run_style respa: the LE fixes act through post_integrate_respa at the outermost level
# (fix_extrusion.cpp:1139-1143, fix_ex_load.cpp:1283-1286, fix_ex_unload.cpp:694-697)
RESPA_STYLES = ["run_style respa 1", "run_style respa 2 4", "run_style respa 3 2 2 bond 1 pair 3", "run_style respa 2 3 bond 2 pair 2"]


@pytest.mark.parametrize("style", RESPA_STYLES)
def test_le_cycle_under_respa(tmp_path, style):
    """extrusion + ex_load + ex_unload with the r-RESPA integrator (bond forces on the inner level by default): topology,
    special lists and fix counters bit-exact against the oracle's restatement of Respa::recurse, positions to 1e-9."""
    n = 3000
    s = melted(n, types=barrier_types(n, 5))
    script = le_script(tp=0.5) + style + "\nrun 44\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    assert len([b for b in o.bond_set() if b[0] == 2]) > 5
    assert p.stat("neigh_builds") == o.neigh_builds()


def test_one_level_respa_is_verlet(tmp_path):
    """`run_style respa 1` performs exactly the operations of run_style verlet in the same order (the single level both
    moves x and carries every force): the two runs agree to the last bit, in the oracle and in the product."""
    n = 3000
    s = melted(n, types=barrier_types(n, 5))
    base = le_script(tp=0.5)
    ov, orr = run_oracle(base + "run 30\n", s), run_oracle(base + "run_style respa 1\nrun 30\n", s)
    assert np.array_equal(ov.x(), orr.x()) and ov.bond_set() == orr.bond_set()
    pv, pr = run_product(base + "run 30\n", s, tmp_path), run_product(base + "run_style respa 1\nrun 30\n", s, tmp_path)
    # (the product's verlet path is the fused kernel, its respa path the unfused kernels: same arithmetic, same order)
    assert np.abs(pv.gather("x") - pr.gather("x")).max() < 1e-10
    assert np.array_equal(pv.gather("bond_atom"), pr.gather("bond_atom"))


def test_respa_is_refused_where_it_is_not_supported(tmp_path):
    from lammps_le_amd import LammpsError
    s = melted(3000)
    with pytest.raises(LammpsError, match="split pair forces"):
        run_product(le_script(left=1, right=1, lr="") + "run_style respa 2 2 inner 1 0.8 1.0 outer 2\n", s, tmp_path)
    with pytest.raises(LammpsError, match="Invalid order of forces"):
        run_product(le_script(left=1, right=1, lr="") + "run_style respa 2 2 bond 2 pair 1\n", s, tmp_path)


def test_five_le_fixes_at_once(tmp_path):
    """More than one instance per LE style (two loaders with different seeds and periods, two unloaders, one extrusion): every
    instance keeps its own RanMars stream and counters; bit-exact against the oracle."""
    n = 3000
    s = melted(n, types=barrier_types(n, 5))
    script = le_script(tp=0.5) + \
        "fix loading2 all ex_load 7 1 1 1.12 2 prob 0.4 91823 iparam 1 1 jparam 1 1\n" \
        "fix unloading2 all ex_unload 13 2 0.8 prob 0.2 55113\nrun 64\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading", "loading2", "unloading2"))
    assert o.fix_vector("loading2")[1] > 0 and o.fix_vector("unloading2")[1] > 0


# ------------------------------------------------------------------------------------------------------------------
# degenerate inputs of the LE fixes: nothing to do, everything to do, the shortest chains
@pytest.mark.parametrize("case", ["no_loadable_type", "prob_zero", "unload_everything", "chains_of_three", "chains_of_four", "chains_of_five"])
def test_le_fixes_degenerate_inputs(tmp_path, case):
    """Empty and extreme cases against the oracle: no bead of the loadable type (the three fixes find nothing and the run is
    plain MD), `prob 0.0` on the loader, an unloader that removes every extruder one step after it was loaded (`prob 1.0`,
    `Rmax 0`), and systems of the shortest chains: with 3 or 4 beads every (i, i+2) pair contains a chain end, which cannot be
    loaded, so nothing happens; 5 beads is the shortest chain that loads (its one admissible pair is next to both ends)."""
    if case == "chains_of_three":
        s = melted(3 * 700, nchains=700, seed=4)
    elif case == "chains_of_four":
        s = melted(4 * 600, nchains=600, seed=4)
    elif case == "chains_of_five":
        s = melted(5 * 500, nchains=500, seed=4)
    else:
        s = melted(3000)
    kw = dict(left=1, right=1, lr="")
    if case == "no_loadable_type":
        s = dict(s)
        s["type"] = np.full_like(s["type"], 2)
        s["ntypes"] = max(int(s["ntypes"]), 2)
        s["mass"] = list(s["mass"]) + [1.0] * (s["ntypes"] - len(s["mass"]))
        script = le_script(neutral=1, **kw)
    elif case == "prob_zero":
        script = le_script(lprob="prob 0.0 684474", **kw)
    elif case == "unload_everything":
        script = le_script(uprob="prob 1.0 456456", rmax=0.0, nu=5, **kw)
    else:
        script = le_script(**kw)
    script += "run 44\n"
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ("loop", "loading", "unloading"))
    loads = o.fix_vector("loading")[1]
    if case in ("no_loadable_type", "prob_zero", "chains_of_three", "chains_of_four"):
        assert loads == 0 and not [b for b in o.bond_set() if b[0] == 2]
    elif case == "unload_everything":
        assert loads > 0 and o.fix_vector("unloading")[1] > 0
    else:
        assert loads > 0


@pytest.mark.parametrize("which", ["all-three", "loader-only", "stock"])
def test_le_fixes_on_a_group(tmp_path, which):
    """The LE fixes (and their src/MC parents) on a group other than all: a bond / a candidate pair counts only if BOTH of its
    atoms are members (fix_extrusion.cpp:373-376, fix_ex_load.cpp:435,450, fix_ex_unload.cpp:228-229).  Three chains, the
    fixes on two of them; against the oracle, and no extruder ever touches the third chain."""
    n = 3600
    s = melted(n, nchains=3, seed=1, types=barrier_types(n, 5))
    grp = "group two molecule 1 2\n"
    if which == "all-three":
        script = le_script(tp=0.5, n1=5, nl=4, nu=6).replace("fix loop all", "fix loop two").replace("fix loading all", "fix loading two") \
            .replace("fix unloading all", "fix unloading two").replace("fix 1 all nve", grp + "fix 1 all nve") + "run 60\n"
        ids = ("loop", "loading", "unloading")
    elif which == "loader-only":
        script = le_script(tp=0.5, n1=5, nl=4, nu=6).replace("fix loading all", "fix loading two").replace("fix 1 all nve", grp + "fix 1 all nve") + "run 60\n"
        ids = ("loop", "loading", "unloading")
    else:
        s["type"] = np.where(np.random.RandomState(3).rand(n) < 0.3, 2, 1).astype(np.int32)
        s["ntypes"], s["mass"] = 3, [1.0, 1.0, 1.0]
        script = CHAIN_SCRIPT.replace("bond_coeff 2 30.0 4.0 1.0 1.0", "bond_coeff 2 5.0 10.0 1.0 1.0") + grp + (
            "fix 1 all nve\nfix 2 all langevin 1.0 1.0 1.0 904297\n"
            "fix creating two bond/create 10 2 2 1.1 2 iparam 1 3 jparam 1 3 prob 0.5 8847\n"
            "fix breaking two bond/break 15 2 1.3 prob 0.5 2211\nthermo 10\nrun 65\n")
        ids = ("creating", "breaking")
    o = run_oracle(script, s)
    p = run_product(script, s, tmp_path)
    compare(p, o, ids)
    ext = [b for b in o.bond_set() if b[0] == 2]
    assert len(ext) > 3
    per = n // 3
    assert all(b[1] <= 2 * per and b[2] <= 2 * per for b in ext)          # nothing on the third chain
