"""TEST INFRASTRUCTURE ONLY (never imported by the product): numpy restatement of the reference's
`velocity <group> create|scale|zero` for group all on one rank.

Follows, as algorithm (no code shared with the engine's C++ restatement):
  src/random_park.cpp:41-77    RanPark::uniform / gaussian  (Park-Miller "minimal standard", polar Box-Muller)
  src/random_park.cpp:95-127   RanPark::reset(seed, coord)  (Jenkins one-at-a-time hash, 5 warm-up draws)
  src/velocity.cpp:162-405     Velocity::create  (loop all | local | geom, dist uniform | gaussian, mom, rot, sum)
  src/velocity.cpp:733-830     rescale, zero_momentum, zero_rotation
  src/compute_temp.cpp:60-101  temperature with dof = 3N - 3
  src/group.cpp:1018-1062, 1125-1163, 1426-1459, 1583-1625, 1682-1728   xcm, vcm, angmom, inertia, omega

Parity unpinned: the reference holds no golden vector for this command (its logs only show the requested temperature
back); the generator itself is pinned by Park & Miller's published check value (seed 1 -> 1043618065 after 10000 draws).
"""
import math

import numpy as np

IM = 2147483647


class RanPark:
    def __init__(self, seed):
        assert seed > 0
        self.seed = int(seed)
        self.save = False
        self.second = 0.0

    def uniform(self):
        # Schrage's factorisation in the reference computes exactly 16807*seed mod (2^31-1)
        self.seed = (16807 * self.seed) % IM
        return (1.0 / IM) * self.seed

    def gaussian(self):
        if self.save:
            self.save = False
            return self.second
        while True:
            v1 = 2.0 * self.uniform() - 1.0
            v2 = 2.0 * self.uniform() - 1.0
            rsq = v1 * v1 + v2 * v2
            if rsq < 1.0 and rsq != 0.0:
                break
        fac = math.sqrt(-2.0 * math.log(rsq) / rsq)
        self.second = v1 * fac
        self.save = True
        return v2 * fac

    def reset(self, ibase, coord):
        data = np.array([ibase], dtype=np.int32).tobytes() + np.asarray(coord, dtype=np.float64).tobytes()
        h = 0
        for b in np.frombuffer(data, dtype=np.int8):          # the reference walks plain (signed) chars
            h = (h + int(b)) & 0xFFFFFFFF
            h = (h + (h << 10)) & 0xFFFFFFFF
            h ^= h >> 6
        h = (h + (h << 3)) & 0xFFFFFFFF
        h ^= h >> 11
        h = (h + (h << 15)) & 0xFFFFFFFF
        self.seed = h & 0x7FFFFFF
        if self.seed == 0:
            self.seed = 1
        for _ in range(5):
            self.uniform()
        self.save = False


def temperature(v, m, order, boltz=1.0, mvv2e=1.0):
    """(`order` = the members of the command's group in local order: compute temp on that group, dof = 3 * members - 3)"""
    t = 0.0
    for i in order:
        t += (v[i, 0] * v[i, 0] + v[i, 1] * v[i, 1] + v[i, 2] * v[i, 2]) * m[i]
    dof = 3.0 * len(order) - 3.0
    return t * (mvv2e / (dof * boltz)) if dof > 0.0 else 0.0          # (src/compute_temp.cpp dof_compute: tfactor = 0)


def zero_momentum(v, m, order):
    if len(order) == 0:
        raise RuntimeError("Cannot zero momentum of no atoms")       # src/velocity.cpp:759-760
    mt = 0.0
    for i in order:
        mt += m[i]
    p = [0.0, 0.0, 0.0]
    for i in order:
        for k in range(3):
            p[k] += v[i, k] * m[i]
    for k in range(3):
        v[order, k] -= p[k] / mt


def zero_rotation(v, m, xu, order):
    if len(order) == 0:
        raise RuntimeError("Cannot zero momentum of no atoms")       # src/velocity.cpp:792-793
    mt = 0.0
    for i in order:
        mt += m[i]
    xcm = np.zeros(3)
    for i in order:
        xcm += xu[i] * m[i]
    xcm /= mt
    L = np.zeros(3)
    I = np.zeros((3, 3))
    for i in order:
        dx, dy, dz = xu[i] - xcm
        L[0] += m[i] * (dy * v[i, 2] - dz * v[i, 1])
        L[1] += m[i] * (dz * v[i, 0] - dx * v[i, 2])
        L[2] += m[i] * (dx * v[i, 1] - dy * v[i, 0])
        I[0, 0] += m[i] * (dy * dy + dz * dz)
        I[1, 1] += m[i] * (dx * dx + dz * dz)
        I[2, 2] += m[i] * (dx * dx + dy * dy)
        I[0, 1] -= m[i] * dx * dy
        I[1, 2] -= m[i] * dy * dz
        I[0, 2] -= m[i] * dx * dz
    I[1, 0], I[2, 1], I[2, 0] = I[0, 1], I[1, 2], I[0, 2]
    w = np.linalg.solve(I, L)        # the reference inverts by cofactors; same w up to rounding
    d = xu - xcm
    o = np.asarray(order)
    v[o, 0] -= w[1] * d[o, 2] - w[2] * d[o, 1]
    v[o, 1] -= w[2] * d[o, 0] - w[0] * d[o, 2]
    v[o, 2] -= w[0] * d[o, 1] - w[1] * d[o, 0]


def velocity_create(x, image, prd, mass_of_atom, t_desired, seed, dist="uniform", mom=True, rot=False, loop="all",
                    vold=None, order=None, member=None, vcur=None):
    """x [n,3] wrapped, image [n,3], mass_of_atom [n] — all in ID order; `order` = local order (ID-1 per local index).
    Returns v [n,3] in ID order.  `vold` given = `sum yes`.  `member` [n] bool: the command's group (`mask[i] & groupbit`,
    src/velocity.cpp:279-355) - the others keep `vcur`; loop all draws a triple for EVERY ID, loop local / geom only for members."""
    n = len(x)
    order = list(range(n)) if order is None else list(order)
    m = np.asarray(mass_of_atom, dtype=np.float64)
    member = np.ones(n, dtype=bool) if member is None else np.asarray(member, dtype=bool)
    v = np.zeros((n, 3)) if vcur is None else np.array(vcur, dtype=np.float64)
    order = [i for i in order if member[i]]

    def draw3(rn):
        if dist == "uniform":
            return [rn.uniform() - 0.5 for _ in range(3)]
        return [rn.gaussian() for _ in range(3)]

    if loop == "all":
        rn = RanPark(seed)
        for i in range(n):
            d3 = draw3(rn)
            if member[i]:
                v[i] = np.array(d3) * (1.0 / math.sqrt(m[i]))
    elif loop == "local":
        rn = RanPark(seed)
        for _ in range(100):
            rn.uniform()
        for i in order:
            v[i] = np.array(draw3(rn)) * (1.0 / math.sqrt(m[i]))
    else:
        rn = RanPark(1)
        for i in range(n):
            if not member[i]:
                continue
            rn.reset(seed, x[i])
            v[i] = np.array(draw3(rn)) * (1.0 / math.sqrt(m[i]))
    if mom:
        zero_momentum(v, m, order)
    if rot:
        zero_rotation(v, m, np.asarray(x) + np.asarray(image) * np.asarray(prd), order)
    t_old = temperature(v, m, order)
    if t_old == 0.0:
        raise RuntimeError("Attempting to rescale a 0.0 temperature")   # src/velocity.cpp:735 (an empty or one-atom group)
    v[order] *= math.sqrt(t_desired / t_old)
    if vold is not None:
        v[order] += np.asarray(vold)[order]
    return v
