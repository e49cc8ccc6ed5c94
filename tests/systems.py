"""Test infrastructure: synthetic bead-spring systems, data-file writer, and a small interpreter
that drives the CPU oracle from the SAME in.* script text the product engine runs.
"""
import os

import numpy as np

from oracle import Oracle


# ------------------------------------------------------------------------------------------
def lattice_chain(nbeads, nchains=1, density=0.85, seed=1, jitter=0.03, temp=1.0, types=None):
    """nchains chains of nbeads/nchains beads laid along a serpentine path through a simple-cubic
    lattice (no overlaps, every bond = lattice spacing); small seeded jitter and Maxwell velocities."""
    rng = np.random.RandomState(seed)
    n = nbeads
    a = (1.0 / density) ** (1.0 / 3.0)
    L = int(np.ceil(n ** (1.0 / 3.0)))
    k = np.arange(n)
    iz = k // (L * L)
    yy = (k % (L * L)) // L
    col = k % L
    iy = np.where(iz % 2 == 0, yy, L - 1 - yy)
    ix = np.where((k // L) % 2 == 0, col, L - 1 - col)
    pts = np.stack([ix, iy, iz], axis=1).astype(np.float64) * a
    box = np.array([[0.0, L * a]] * 3)
    x = pts + 0.5 * a + rng.uniform(-jitter, jitter, size=(n, 3))
    v = rng.normal(0.0, np.sqrt(temp), size=(n, 3))
    v -= v.mean(axis=0)
    per = n // nchains
    bonds = []
    for c in range(nchains):
        lo = c * per
        hi = n if c == nchains - 1 else (c + 1) * per
        for i in range(lo, hi - 1):
            bonds.append((1, i + 1, i + 2))
    typ = np.ones(n, dtype=np.int32) if types is None else np.asarray(types, dtype=np.int32)
    mol = np.minimum(np.arange(n) // per, nchains - 1).astype(np.int32) + 1
    return dict(box=box, x=x, v=v, type=typ, mol=mol, image=np.zeros((n, 3), dtype=np.int32),
                bonds=np.array(bonds, dtype=np.int32).reshape(-1, 3), ntypes=int(typ.max()), nbondtypes=2,
                mass=[1.0] * int(typ.max()), extra_bond=1, extra_special=20, atom_style="bond")


def write_data(path, s):
    n = len(s["x"])
    with open(path, "w") as fh:
        fh.write("synthetic bead-spring system\n\n")
        fh.write("%d atoms\n%d atom types\n%d bonds\n%d bond types\n" % (n, s["ntypes"], len(s["bonds"]), s["nbondtypes"]))
        if s.get("nangletypes"):
            fh.write("%d angles\n%d angle types\n" % (len(s.get("angles", [])), s["nangletypes"]))
            if s.get("extra_angle"):
                fh.write("%d extra angle per atom\n" % s["extra_angle"])
        if s.get("extra_bond"):
            fh.write("%d extra bond per atom\n" % s["extra_bond"])
        if s.get("extra_special"):
            fh.write("%d extra special per atom\n" % s["extra_special"])
        fh.write("\n")
        for d, nm in enumerate("xyz"):
            fh.write("%.17g %.17g %slo %shi\n" % (s["box"][d][0], s["box"][d][1], nm, nm))
        fh.write("\nMasses\n\n")
        for t, m in enumerate(s["mass"]):
            fh.write("%d %.17g\n" % (t + 1, m))
        fh.write("\nAtoms\n\n")
        x, typ, mol, img = s["x"], s["type"], s["mol"], s["image"]
        order = s.get("file_order", np.arange(n))
        for i in order:
            fh.write("%d %d %d %.17g %.17g %.17g %d %d %d\n" % (i + 1, mol[i], typ[i], x[i, 0], x[i, 1], x[i, 2],
                                                            img[i, 0], img[i, 1], img[i, 2]))
        fh.write("\nVelocities\n\n")
        v = s["v"]
        for i in range(n):
            fh.write("%d %.17g %.17g %.17g\n" % (i + 1, v[i, 0], v[i, 1], v[i, 2]))
        if len(s["bonds"]):
            fh.write("\nBonds\n\n")
            for k, (bt, a, b) in enumerate(s["bonds"]):
                fh.write("%d %d %d %d\n" % (k + 1, bt, a, b))
        if s.get("nangletypes") and len(s.get("angles", [])):
            fh.write("\nAngles\n\n")
            for k, (at, a, b, c) in enumerate(s["angles"]):
                fh.write("%d %d %d %d %d\n" % (k + 1, at, a, b, c))


def wrap_into_box(s):
    """What read_data does to the coordinates it reads (domain->remap): wrap + image flags."""
    x = s["x"].copy()
    img = s["image"].copy()
    for d in range(3):
        lo, hi = s["box"][d]
        prd = hi - lo
        while True:
            m = x[:, d] < lo
            if not m.any():
                break
            x[m, d] += prd
            img[m, d] -= 1
        while True:
            m = x[:, d] >= hi
            if not m.any():
                break
            x[m, d] -= prd
            img[m, d] += 1
    return x, img


# ------------------------------------------------------------------------------------------
class OracleScript:
    """Feed LAMMPS script lines to the oracle (the subset of commands of SURVEY §8b(1))."""

    def __init__(self, system):
        self.sys = system
        self.o = None
        self.units = "lj"
        self.special = (0.0, 0.0, 0.0)
        self.special_coul = (0.0, 0.0, 0.0)
        self.pending = []
        self.bond_style = None
        self.hybrid = []
        self.shift = False
        self.mix = "geometric"

    def _make(self):
        s = self.sys
        n = len(s["x"])
        o = Oracle(n, s["ntypes"], s["nbondtypes"], s.get("extra_bond", 0), s.get("extra_special", 0))
        o.units(self.units)
        box = np.asarray(s["box"])
        o.box(box[:, 0], box[:, 1])
        for t, m in enumerate(s["mass"]):
            o.mass(t + 1, m)
        x, img = wrap_into_box(s)
        order = s.get("file_order", np.arange(n))
        tag = (np.asarray(order) + 1).astype(np.int32)
        o.atoms(tag, s["type"][order], x[order], s["v"][order], img[order])
        o.bonds(s["bonds"])
        if s.get("nangletypes"):
            o.angles(s["nangletypes"], s.get("angles", np.zeros((0, 4), dtype=np.int32)), s.get("extra_angle", 0))
        o.special_bonds(*self.special, coul=self.special_coul)
        self.o = o

    def group_define(self, a):
        n = len(self.sys["x"])
        groups = self.__dict__.setdefault("groups", {"all": np.ones(n, dtype=bool)})
        name, style = a[0], a[1]
        cur = groups.get(name, np.zeros(n, dtype=bool))
        if style in ("type", "id", "molecule"):
            vals = (np.asarray(self.o.types()) if self.o else np.asarray(self.sys["type"])) if style == "type" else \
                np.arange(1, n + 1) if style == "id" else np.asarray(self.sys["mol"])
            for t in a[2:]:
                q = [int(v) for v in t.split(":")]
                lo, hi, st = (q[0], q[0], 1) if len(q) == 1 else (q[0], q[1], q[2] if len(q) > 2 else 1)
                cur = cur | ((vals >= lo) & (vals <= hi) & ((vals - lo) % st == 0))
        elif style == "union":
            for g in a[2:]:
                cur = cur | groups[g]
        elif style == "intersect":
            sel = np.ones(n, dtype=bool)
            for g in a[2:]:
                sel &= groups[g]
            cur = cur | sel
        elif style == "subtract":
            sel = groups[a[2]].copy()
            for g in a[3:]:
                sel &= ~groups[g]
            cur = cur | sel
        elif style != "empty":
            raise ValueError("oracle script: group style " + style)
        groups[name] = cur

    def group_flags(self, name):
        return self.groups[name].astype(np.int32)

    def line(self, text):
        text = text.split("#")[0].strip()
        if not text:
            return
        w = text.split()
        c, a = w[0], w[1:]
        o = self.o
        if c == "units":
            self.units = a[0]
        elif c == "newton":
            self.newton_pair = a[0] == "on"          # `newton pair [bond]`; bonds are always stored on both atoms here
            if o:
                o.newton_pair(self.newton_pair)
        elif c in ("atom_style", "comm_modify", "boundary", "thermo_style", "thermo_modify", "echo", "log"):
            pass
        elif c == "group":               # group ID type | id | molecule values / a:b[:stride] | union | subtract | intersect
            self.group_define(a)
        elif c == "run_style":
            if o is None:
                raise ValueError("oracle script: run_style before read_data")
            if a[0] == "verlet":
                o.run_style_respa(None)
            elif a[0] == "respa":
                n = int(a[1])
                loops = [int(v) for v in a[2:1 + n]]
                kw, k = {}, 1 + n
                while k < len(a):
                    if a[k] not in ("bond", "pair", "angle"):
                        raise ValueError("oracle script: run_style respa keyword " + a[k])
                    kw["level_" + a[k]] = int(a[k + 1])
                    k += 2
                o.run_style_respa(loops, **kw)
            else:
                raise ValueError("oracle script: run_style " + a[0])
        elif c == "atom_modify":
            if a[0] == "sort":
                self.sort = int(a[1])
                if o:
                    o.atom_sort(self.sort)
        elif c == "special_bonds":
            # src/force.cpp:748-826: every command starts from lj 0 0 0 / coul 0 0 0
            lj, coul, k = (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0
            while k < len(a):
                if a[k] == "fene":
                    lj, coul, k = (0.0, 1.0, 1.0), (0.0, 1.0, 1.0), k + 1
                elif a[k] in ("lj", "coul", "lj/coul"):
                    w = tuple(float(v) for v in a[k + 1:k + 4])
                    if a[k] != "coul":
                        lj = w
                    if a[k] != "lj":
                        coul = w
                    k += 4
                else:
                    raise ValueError("oracle script: special_bonds " + a[k])
            self.special, self.special_coul = lj, coul
            if o:
                o.special_bonds(*self.special, coul=self.special_coul)
        elif c == "read_data":
            self._make()
            if hasattr(self, "sort"):
                self.o.atom_sort(self.sort)
            if hasattr(self, "newton_pair"):
                self.o.newton_pair(self.newton_pair)
        elif c == "neighbor":
            o.neighbor(skin=float(a[0]))
        elif c == "neigh_modify":
            kw = dict(zip(a[::2], a[1::2]))
            o.neighbor(every=int(kw.get("every", 0)), delay=int(kw.get("delay", -1)),
                       check={"yes": 1, "no": 0, None: -1}[kw.get("check")])
        elif c == "angle_style":
            self.angle_style = a[0]
        elif c == "angle_coeff":
            types = range(1, self.sys["nangletypes"] + 1) if a[0] == "*" else [int(a[0])]
            for at in types:
                o.angle_coeff(at, self.angle_style, *[float(v) for v in a[1:]])
        elif c == "bond_style":
            self.bond_style = a[0]
            self.hybrid = a[1:]
        elif c == "bond_coeff":
            st = self.bond_style
            vals = a[1:]
            if st == "hybrid":
                st, vals = a[1], a[2:]
            types = range(1, self.sys["nbondtypes"] + 1) if a[0] == "*" else [int(a[0])]
            for bt in types:
                o.bond_coeff(bt, st, *[float(v) for v in vals])
        elif c == "pair_style":
            self.pair_cut = float(a[1])
            self._pair = True
            o.pair_lj_cut(self.pair_cut, self.shift, self.mix)
        elif c == "pair_modify":
            kw = dict(zip(a[::2], a[1::2]))
            if "shift" in kw:
                self.shift = kw["shift"] == "yes"
            if "mix" in kw:
                self.mix = kw["mix"]
            o.pair_lj_cut(self.pair_cut, self.shift, self.mix)
        elif c == "pair_coeff":
            def rng(tok):
                nt = self.sys["ntypes"]
                if "*" not in tok:
                    return [int(tok)]
                lo, hi = tok.split("*")
                return range(int(lo) if lo else 1, (int(hi) if hi else nt) + 1)
            for i in rng(a[0]):
                for j in rng(a[1]):
                    if j >= i:
                        o.pair_coeff(i, j, float(a[2]), float(a[3]), float(a[4]) if len(a) > 4 else -1.0)
        elif c == "fix":
            fid, style, p = a[0], a[2], a[3:]
            if style == "nve":
                o.fix_nve(fid)
                if a[1] != "all":
                    o.nve_group(self.group_flags(a[1]))
            elif style == "langevin":
                o.fix_langevin(float(p[0]), float(p[1]), float(p[2]), int(p[3]), fid)
                k = 4
                while k < len(p):                   # optional keywords (src/fix_langevin.cpp:105-155)
                    if p[k] == "scale":
                        o.langevin_scale(int(p[k + 1]), float(p[k + 2]))
                        k += 3
                    elif p[k] == "zero":
                        o.langevin_zero(p[k + 1] == "yes")
                        k += 2
                    elif p[k] in ("tally", "gjf", "omega", "angmom") and p[k + 1] == "no":      # (the defaults)
                        k += 2
                    else:
                        raise ValueError("oracle script: fix langevin keyword " + p[k])
                if a[1] != "all":
                    o.langevin_group(self.group_flags(a[1]))
            elif style == "extrusion":
                o.fix_extrusion(int(p[0]), int(p[1]), int(p[2]), int(p[3]), float(p[4]), int(p[5]),
                                int(p[6]) if len(p) > 6 else -1, fid)
            elif style in ("ex_load", "bond/create"):
                kw = dict(imax=0, inew=None, jmax=0, jnew=None, fraction=1.0, seed=12345)
                atype = 0
                k = 5
                while k < len(p):
                    if p[k] == "iparam":
                        kw["imax"], kw["inew"] = int(p[k + 1]), int(p[k + 2])
                    elif p[k] == "jparam":
                        kw["jmax"], kw["jnew"] = int(p[k + 1]), int(p[k + 2])
                    elif p[k] == "prob":
                        kw["fraction"], kw["seed"] = float(p[k + 1]), int(p[k + 2])
                    elif p[k] in ("atype", "dtype", "itype"):
                        # fix_ex_load.cpp:236-254: effective only `if (atype && force->angle)`: angles when the script
                        # defined an angle style; no dihedral / improper styles here, so dtype / itype have no effect
                        if p[k] == "atype" and getattr(self, "angle_style", None) in ("harmonic", "cosine"):
                            atype = int(p[k + 1])
                        k += 2
                        continue
                    k += 3
                # fix_ex_load.cpp:130-133 / fix_bond_create.cpp:153-156: one atom type at both ends must come with one set of limits
                inew = kw["inew"] if kw["inew"] is not None else int(p[1])
                jnew = kw["jnew"] if kw["jnew"] is not None else int(p[2])
                if int(p[1]) == int(p[2]) and (kw["imax"] != kw["jmax"] or inew != jnew):
                    raise RuntimeError("Inconsistent iparam/jparam values in fix %s command" % style)
                (o.fix_ex_load if style == "ex_load" else o.fix_bond_create)(int(p[0]), int(p[1]), int(p[2]), float(p[3]), int(p[4]), fid=fid, **kw)
                if atype:
                    o.ex_load_atype(fid, atype)
            elif style in ("ex_unload", "bond/break"):
                kw = dict(fraction=1.0, seed=12345)
                if len(p) > 3 and p[3] == "prob":
                    kw["fraction"], kw["seed"] = float(p[4]), int(p[5])
                (o.fix_ex_unload if style == "ex_unload" else o.fix_bond_break)(int(p[0]), int(p[1]), float(p[2]), fid=fid, **kw)
            else:
                raise ValueError("oracle script: unknown fix " + style)
            if a[1] != "all" and style not in ("nve", "langevin"):
                o.fix_group(fid, self.group_flags(a[1]))
        elif c == "velocity":
            self._velocity(a)
        elif c == "timestep":
            o.timestep(float(a[0]))
        elif c == "thermo":
            o.thermo_every(int(a[0]))
        elif c == "reset_timestep":
            o.reset_timestep(int(a[0]))
        elif c == "run":
            o.run(int(a[0]))
        else:
            raise ValueError("oracle script: unknown command " + c)

    def _velocity(self, a):
        """velocity all create T seed [dist ..] [mom ..] [rot ..] [loop ..] [sum ..]  (oracle/velocity_oracle.py)"""
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        from velocity_oracle import velocity_create
        assert a[1] == "create", "oracle script: velocity create only"
        member = None if a[0] == "all" else self.groups[a[0]]
        kw = dict(zip(a[4::2], a[5::2]))
        o = self.o
        box = np.asarray(self.sys["box"])
        m = np.asarray(self.sys["mass"])[o.types() - 1]
        v = velocity_create(o.x(), o.image(), box[:, 1] - box[:, 0], m, float(a[2]), int(a[3]),
                            dist=kw.get("dist", "uniform"), mom=kw.get("mom", "yes") == "yes",
                            rot=kw.get("rot", "no") == "yes", loop=kw.get("loop", "all"),
                            vold=o.v() if kw.get("sum", "no") == "yes" else None, order=o.local_order() - 1,
                            member=member, vcur=o.v())
        o.set_v(v)

    def run(self, script):
        for ln in script.split("\n"):
            self.line(ln)
        return self.o


def run_oracle(script, system):
    return OracleScript(system).run(script)


def run_product(script, system, tmpdir, cmdargs=("-screen", "none")):
    """Run the same script on the product engine (through the C-ABI); the data file named in the
    script's read_data line is written into tmpdir."""
    from lammps_le_amd import lammps
    lmp = lammps(cmdargs=list(cmdargs))
    for ln in script.split("\n"):
        w = ln.split("#")[0].split()
        if w and w[0] == "read_data":
            path = os.path.join(str(tmpdir), os.path.basename(w[1]))
            write_data(path, system)
            ln = "read_data " + path
        lmp.command(ln)
    return lmp


CHAIN_SCRIPT = """
units lj
atom_style bond
newton off
atom_modify sort 0 0
special_bonds fene
read_data data.chain
neighbor 0.4 bin
neigh_modify every 1 delay 1 check yes
comm_modify cutoff 5.0
bond_style fene
bond_coeff 1 30.0 1.5 1.0 1.0
bond_coeff 2 30.0 4.0 1.0 1.0
pair_style lj/cut 1.12
pair_modify shift yes
pair_coeff * * 1.0 1.0 1.12
timestep 0.005
"""
