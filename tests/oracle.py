"""ctypes binding of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Nothing in lammps_le_amd/ imports this.  The oracle is the checker the HIP path is
compared against (tests/, smoke()) and the reported CPU baseline (bench.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("ORACLE_SO") or os.path.join(_ROOT, "oracle", "liboracle.so")   # ORACLE_SO: e.g. the ASan/UBSan build (oracle/Makefile `asan`)


def build():
    src = os.path.join(_ROOT, "oracle", "le_oracle.c")
    if os.environ.get("ORACLE_SO"):
        return _SO
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.leo_new.restype = C.c_void_p
        L.leo_new.argtypes = [C.c_int] * 5
        L.leo_error.restype = C.c_char_p
        for name in ("leo_ntimestep", "leo_nbonds", "leo_neigh_builds", "leo_neigh_pairs", "leo_fene_warnings", "leo_nangles"):
            getattr(L, name).restype = C.c_long
        L.leo_angle_energy.restype = C.c_double
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def ranmars_stream(seed, n):
    out = np.zeros(n)
    lib().leo_ranmars_stream(C.c_int(seed), C.c_int(n), _dp(out))
    return out


class Oracle:
    """One oracle system.  Methods mirror the script commands they restate."""

    def __init__(self, natoms, ntypes, nbondtypes, extra_bond=0, extra_special=0):
        self.L = lib()
        self.h = C.c_void_p(self.L.leo_new(natoms, ntypes, nbondtypes, extra_bond, extra_special))
        self.n = natoms
        self.ntypes = ntypes
        self.nfix = 0
        self.fix_ids = {}

    def __del__(self):
        try:
            self.L.leo_free(self.h)
        except Exception:
            pass

    # -- construction --
    def units(self, name):
        self.L.leo_units(self.h, 0 if name == "lj" else 1)

    def box(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        self.L.leo_set_box(self.h, _dp(lo), _dp(hi))

    def mass(self, t, m):
        self.L.leo_set_mass(self.h, C.c_int(t), C.c_double(m))

    def atoms(self, tag, typ, x, v=None, image=None):
        tag = np.ascontiguousarray(tag, dtype=np.int32)
        typ = np.ascontiguousarray(typ, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.zeros_like(x) if v is None else np.ascontiguousarray(v, dtype=np.float64)
        image = np.zeros((self.n, 3), dtype=np.int32) if image is None else np.ascontiguousarray(image, dtype=np.int32)
        self.L.leo_set_atoms(self.h, _ip(tag), _ip(typ), _dp(x), _dp(v), _ip(image))

    def bonds(self, bonds):
        b = np.ascontiguousarray(bonds, dtype=np.int32).reshape(-1, 3)
        bt, a1, a2 = (np.ascontiguousarray(b[:, k]) for k in range(3))
        self.L.leo_set_bonds(self.h, C.c_int(len(b)), _ip(bt), _ip(a1), _ip(a2))

    def angles(self, nangletypes, angles, extra_angle=0):
        """Angles section rows (type, atom1, atom2, atom3); `extra angle per atom` as extra_angle."""
        a = np.ascontiguousarray(angles, dtype=np.int32).reshape(-1, 4)
        cols = [np.ascontiguousarray(a[:, k]) for k in range(4)]
        self.L.leo_set_angles(self.h, C.c_int(nangletypes), C.c_int(len(a)), _ip(cols[0]), _ip(cols[1]), _ip(cols[2]), _ip(cols[3]),
                              C.c_int(extra_angle))

    def angle_coeff(self, t, style, k, theta0=0.0):
        self.L.leo_angle_coeff(self.h, C.c_int(t), C.c_int({"harmonic": 1, "cosine": 2}[style]), C.c_double(k), C.c_double(theta0))

    def nve_group(self, flag_by_tag):
        f = np.ascontiguousarray(flag_by_tag, dtype=np.int32)
        self.L.leo_nve_group(self.h, _ip(f))

    def fix_group(self, fix_id, flag_by_tag):
        f = np.ascontiguousarray(flag_by_tag, dtype=np.int32)
        self.L.leo_fix_group(self.h, C.c_int(self.fix_ids[fix_id]), _ip(f))

    def langevin_scale(self, itype, ratio):
        self.L.leo_langevin_scale(self.h, C.c_int(itype), C.c_double(ratio))

    def langevin_zero(self, flag):
        self.L.leo_langevin_zero(self.h, C.c_int(1 if flag else 0))

    def langevin_group(self, flag_by_tag):
        f = np.ascontiguousarray(flag_by_tag, dtype=np.int32)
        self.L.leo_langevin_group(self.h, _ip(f))

    def ex_load_atype(self, fix_id, atype):
        self.L.leo_ex_load_atype(self.h, C.c_int(self.fix_ids[fix_id]), C.c_int(atype))

    def nangles(self):
        return int(self.L.leo_nangles(self.h))

    def angle_energy(self):
        return float(self.L.leo_angle_energy(self.h))

    def angle_virial(self):
        o = np.zeros(6)
        self.L.leo_angle_virial(self.h, _dp(o))
        return o

    def angle_table(self):
        apa = int(self.L.leo_angle_per_atom(self.h))
        na = np.zeros(self.n, dtype=np.int32)
        arr = [np.zeros((self.n, max(apa, 1)), dtype=np.int32) for _ in range(4)]
        self.L.leo_get_angles(self.h, _ip(na), *[_ip(a) for a in arr])
        return na, arr[0], arr[1], arr[2], arr[3]

    def angle_set(self):
        """{(type, a1, a2, a3)} of the copies stored on the CENTRAL atom (one per angle), ends ordered."""
        na, at, a1, a2, a3 = self.angle_table()
        out = set()
        for i in np.nonzero(na)[0]:
            for m in range(na[i]):
                if a2[i, m] == i + 1:
                    out.add((int(at[i, m]), int(min(a1[i, m], a3[i, m])), int(a2[i, m]), int(max(a1[i, m], a3[i, m]))))
        return out

    def special_bonds(self, w1, w2, w3, coul=(0.0, 0.0, 0.0)):
        """special_bonds lj w1 w2 w3 [coul c1 c2 c3]; `fene` = lj 0 1 1 coul 0 1 1 (src/force.cpp:748-826)."""
        self.L.leo_special_coul(self.h, *[C.c_double(c) for c in coul])
        self.L.leo_special_build(self.h, C.c_double(w1), C.c_double(w2), C.c_double(w3))

    def pair_lj_cut(self, cut, shift=False, mix="geometric"):
        self.L.leo_pair_lj_cut(self.h, C.c_double(cut), C.c_int(int(shift)), C.c_int(1 if mix == "arithmetic" else 0))

    def pair_coeff(self, i, j, eps, sigma, cut=-1.0):
        self.L.leo_pair_coeff(self.h, C.c_int(i), C.c_int(j), C.c_double(eps), C.c_double(sigma), C.c_double(cut))

    def bond_coeff(self, bt, style, *p):
        p = list(p) + [0.0] * (4 - len(p))
        self.L.leo_bond_coeff(self.h, C.c_int(bt), C.c_int({"fene": 1, "harmonic": 2, "morse": 3}[style]), *[C.c_double(v) for v in p])

    def timestep(self, dt):
        self.L.leo_timestep(self.h, C.c_double(dt))

    def neighbor(self, skin=-1.0, every=0, delay=-1, check=-1):
        self.L.leo_neighbor(self.h, C.c_double(skin), C.c_int(every), C.c_int(delay), C.c_int(check))

    def newton_pair(self, on):
        self.L.leo_newton_pair(self.h, C.c_int(1 if on else 0))

    def run_style_respa(self, loops, level_bond=0, level_pair=0, level_angle=0):
        """run_style respa N loop_1 .. loop_{N-1} [bond L] [pair L]; loops = [] selects run_style verlet again"""
        n = len(loops) + 1 if loops is not None else 0
        arr = (C.c_int * max(n, 1))(*(list(loops) + [1])) if n else (C.c_int * 1)(1)
        self.L.leo_run_style_respa.restype = C.c_int
        if self.L.leo_run_style_respa(self.h, C.c_int(n), arr, C.c_int(level_bond), C.c_int(level_pair)):
            raise RuntimeError(self.L.leo_error(self.h).decode())
        self.L.leo_run_style_respa_angle.restype = C.c_int
        if n and self.L.leo_run_style_respa_angle(self.h, C.c_int(level_angle)):
            raise RuntimeError(self.L.leo_error(self.h).decode())

    def atom_sort(self, freq):
        self.L.leo_atom_sort(self.h, C.c_int(freq))

    def reset_timestep(self, step):
        self.L.leo_reset_timestep(self.h, C.c_long(int(step)))

    def thermo_every(self, n):
        self.L.leo_thermo_every(self.h, C.c_int(n))

    # -- fixes --
    def _reg(self, fid):
        self.fix_ids[fid] = self.nfix
        self.nfix += 1

    def fix_nve(self, fid="nve"):
        self.L.leo_fix_nve(self.h)
        self._reg(fid)

    def fix_langevin(self, t0, t1, damp, seed, fid="langevin"):
        self.L.leo_fix_langevin(self.h, C.c_double(t0), C.c_double(t1), C.c_double(damp), C.c_int(seed))
        self._reg(fid)

    def fix_extrusion(self, nevery, neutral, left, right, tp, btype, lr=-1, fid="loop"):
        self.L.leo_fix_extrusion(self.h, nevery, neutral, left, right, C.c_double(tp), btype, lr)
        self._reg(fid)

    def fix_ex_load(self, nevery, it, jt, cutoff, btype, imax=0, inew=None, jmax=0, jnew=None, fraction=1.0,
                    seed=12345, fid="loading"):
        inew = it if inew is None else inew
        jnew = jt if jnew is None else jnew
        self.L.leo_fix_ex_load(self.h, nevery, it, jt, C.c_double(cutoff), btype, imax, inew, jmax, jnew,
                               C.c_double(fraction), seed)
        self._reg(fid)

    def fix_ex_unload(self, nevery, btype, cutoff, fraction=1.0, seed=12345, fid="unloading"):
        self.L.leo_fix_ex_unload(self.h, nevery, btype, C.c_double(cutoff), C.c_double(fraction), seed)
        self._reg(fid)

    def fix_bond_create(self, nevery, it, jt, cutoff, btype, imax=0, inew=None, jmax=0, jnew=None, fraction=1.0,
                        seed=12345, fid="creating"):
        inew = it if inew is None else inew
        jnew = jt if jnew is None else jnew
        self.L.leo_fix_bond_create(self.h, nevery, it, jt, C.c_double(cutoff), btype, imax, inew, jmax, jnew,
                                   C.c_double(fraction), seed)
        self._reg(fid)

    def fix_bond_break(self, nevery, btype, cutoff, fraction=1.0, seed=12345, fid="breaking"):
        self.L.leo_fix_bond_break(self.h, nevery, btype, C.c_double(cutoff), C.c_double(fraction), seed)
        self._reg(fid)

    # -- running --
    def run(self, n):
        rc = self.L.leo_run(self.h, C.c_int(n))
        if rc:
            raise RuntimeError(self.L.leo_error(self.h).decode())

    def setup_forces(self):
        rc = self.L.leo_setup_forces(self.h)
        if rc:
            raise RuntimeError(self.L.leo_error(self.h).decode())

    def fire_fix(self, fid):
        rc = self.L.leo_fire_fix(self.h, C.c_int(self.fix_ids[fid]))
        if rc:
            raise RuntimeError(self.L.leo_error(self.h).decode())

    # -- queries --
    def thermo(self):
        out = np.zeros(14)
        self.L.leo_thermo(self.h, _dp(out))
        return out

    def thermo_history(self):
        n = self.L.leo_thermo_count(self.h)
        out = np.zeros((n, 16))
        for i in range(n):
            self.L.leo_thermo_get(self.h, C.c_int(i), _dp(out[i]))
        return out

    def _vec(self, fn, w=3, dtype=np.float64):
        out = np.zeros((self.n, w), dtype=dtype)
        getattr(self.L, fn)(self.h, _dp(out) if dtype == np.float64 else _ip(out))
        return out

    def x(self):
        return self._vec("leo_get_x")

    def v(self):
        return self._vec("leo_get_v")

    def f(self):
        return self._vec("leo_get_f")

    def types(self):
        return self._vec("leo_get_type", 1, np.int32)[:, 0]

    def image(self):
        return self._vec("leo_get_image", 3, np.int32)

    def local_order(self):
        return self._vec("leo_get_local_order", 1, np.int32)[:, 0]

    def set_x(self, x):
        self.L.leo_set_x(self.h, _dp(np.ascontiguousarray(x, dtype=np.float64)))

    def set_v(self, v):
        self.L.leo_set_v(self.h, _dp(np.ascontiguousarray(v, dtype=np.float64)))

    def bond_table(self):
        bpa = self.L.leo_bond_per_atom(self.h)
        nb = np.zeros(self.n, dtype=np.int32)
        bt = np.zeros((self.n, bpa), dtype=np.int32)
        ba = np.zeros((self.n, bpa), dtype=np.int32)
        self.L.leo_get_bonds(self.h, _ip(nb), _ip(bt), _ip(ba))
        return nb, bt, ba

    def bond_set(self):
        """Set of (type, lo_tag, hi_tag) over all stored bonds (each stored on both atoms)."""
        nb, bt, ba = self.bond_table()
        out = set()
        for i in np.nonzero(nb)[0]:
            for m in range(nb[i]):
                a, b = i + 1, int(ba[i, m])
                out.add((int(bt[i, m]), min(a, b), max(a, b)))
        return out

    def special_table(self):
        ms = self.L.leo_maxspecial(self.h)
        ns = np.zeros((self.n, 3), dtype=np.int32)
        sp = np.zeros((self.n, ms), dtype=np.int32)
        self.L.leo_get_special(self.h, _ip(ns), _ip(sp))
        return ns, sp

    def fix_vector(self, fid):
        out = np.zeros(2)
        self.L.leo_fix_vector(self.h, C.c_int(self.fix_ids[fid]), _dp(out))
        return out

    def nbonds(self):
        return self.L.leo_nbonds(self.h)

    def ntimestep(self):
        return self.L.leo_ntimestep(self.h)

    def neigh_builds(self):
        return self.L.leo_neigh_builds(self.h)

    def neigh_pairs(self):
        return self.L.leo_neigh_pairs(self.h)

    def fene_warnings(self):
        return self.L.leo_fene_warnings(self.h)

    def pair_virial(self):
        out = np.zeros(6)
        self.L.leo_pair_virial(self.h, _dp(out))
        return out

    def bond_virial(self):
        out = np.zeros(6)
        self.L.leo_bond_virial(self.h, _dp(out))
        return out

    def timers(self):
        out = np.zeros(6)
        self.L.leo_timers(self.h, _dp(out))
        return dict(zip(("pair", "bond", "neigh", "modify", "other", "total"), out))
