#!/usr/bin/env python3
"""Generate the committed golden fixtures from the reference's OWN test data and logs.

Run once in the development container (needs /root/reference, read-only):

    python tests/golden/make_golden.py

Only DATA is extracted (atom coordinates, bond tables, coefficients, expected
energies/forces/stresses, thermo numbers).  No reference source or script text is copied.

Sources (relative to /root/reference):
  unittest/force-styles/tests/data.fourmol                  -> fourmol.json (system)
  unittest/force-styles/tests/mol-pair-lj_cut.yaml          -> lj_cut.json  (known answers)
  unittest/force-styles/tests/bond-fene.yaml                -> bond_fene.json
  unittest/force-styles/tests/bond-harmonic.yaml            -> bond_harmonic.json
  unittest/force-styles/tests/bond-hybrid.yaml              -> bond_hybrid.json (hybrid harmonic morse: per-type dispatch)
  unittest/force-styles/tests/angle-harmonic.yaml           -> angle_harmonic.json
  unittest/force-styles/tests/angle-cosine.yaml             -> angle_cosine.json
  unittest/force-styles/tests/fix-timestep-nve.yaml         -> fix_nve.json (positions / velocities after 8 steps of
                                                               `fix nve` on the group of molecules 1-2, with the harness'
                                                               lj/cut + harmonic bonds + harmonic angles; values of that
                                                               force field as set by test_fix_timestep.cpp:115-135)
  bench/data.chain                                          -> chain32k.npz (system)
  bench/log.6Oct16.chain.fixed.icc.1:48-49,68-76            -> chain32k_thermo.json
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def parse_data(path):
    """Minimal LAMMPS data-file parser (header + Masses/Atoms/Velocities/Bonds)."""
    with open(path) as fh:
        lines = fh.read().split("\n")
    hdr = {}
    sections = {}
    i = 1
    sec_names = ("Masses", "Atoms", "Velocities", "Bonds", "Angles", "Dihedrals", "Impropers",
                 "Pair Coeffs", "Bond Coeffs", "Angle Coeffs", "Dihedral Coeffs", "Improper Coeffs")
    cur = None
    while i < len(lines):
        ln = lines[i].split("#")[0].strip()
        raw = lines[i].strip()
        i += 1
        if not ln:
            continue
        name = next((s for s in sec_names if ln == s or raw.startswith(s + " #") or raw == s), None)
        if name:
            cur = name
            sections[cur] = []
            continue
        if cur is None:
            t = ln.split()
            if ln.endswith("xlo xhi"):
                hdr["x"] = (float(t[0]), float(t[1]))
            elif ln.endswith("ylo yhi"):
                hdr["y"] = (float(t[0]), float(t[1]))
            elif ln.endswith("zlo zhi"):
                hdr["z"] = (float(t[0]), float(t[1]))
            else:
                hdr[" ".join(t[1:])] = int(t[0])
        else:
            sections[cur].append(ln.split())
    return hdr, sections


def yaml_block(text, key):
    m = re.search(r"^%s: ! \|[-0-9]*\n((?:[ \t]+.*\n)+)" % re.escape(key), text, re.M)
    rows = [[float(v) for v in r.split()] for r in m.group(1).strip().split("\n")]
    return rows


def yaml_scalar(text, key):
    return float(re.search(r"^%s: (.*)$" % re.escape(key), text, re.M).group(1))


def known_answers(path, ekey):
    text = open(path).read()
    out = {"epsilon": yaml_scalar(text, "epsilon")}
    coeff_key = "pair_coeff" if "pair_coeff" in text else "bond_coeff" if "bond_coeff" in text else "angle_coeff"
    m = re.search(r"^%s: ! \|[-0-9]*\n((?:[ \t]+.*\n)+)" % re.escape(coeff_key), text, re.M)
    rows = [r.split() for r in m.group(1).strip().split("\n")]
    # hybrid styles name the sub-style in front of the numbers: keep it as a string
    out[coeff_key] = [[float(v) if re.match(r"^[-+0-9.eE]+$", v) else v for v in r] for r in rows]
    for phase in ("init", "run"):
        out[phase + "_energy"] = yaml_scalar(text, "%s_%s" % (phase, ekey))
        out[phase + "_stress"] = yaml_block(text, phase + "_stress")[0]
        f = yaml_block(text, phase + "_forces")
        out[phase + "_forces"] = [r[1:] for r in sorted(f)]
    return out


def main():
    ft = REF + "/unittest/force-styles/tests/"
    hdr, sec = parse_data(ft + "data.fourmol")
    atoms = sec["Atoms"]  # id mol type q x y z ix iy iz   (atom_style full)
    four = {
        "natoms": hdr["atoms"], "ntypes": hdr["atom types"], "nbondtypes": hdr["bond types"],
        "box": [list(hdr["x"]), list(hdr["y"]), list(hdr["z"])],
        "mass": {r[0]: float(r[1]) for r in sec["Masses"]},
        "tag": [int(r[0]) for r in atoms], "mol": [int(r[1]) for r in atoms],
        "type": [int(r[2]) for r in atoms], "q": [float(r[3]) for r in atoms],
        "x": [[float(r[4]), float(r[5]), float(r[6])] for r in atoms],
        "image": [[int(r[7]), int(r[8]), int(r[9])] for r in atoms],
        "vel": {r[0]: [float(r[1]), float(r[2]), float(r[3])] for r in sec["Velocities"]},
        "bonds": [[int(r[1]), int(r[2]), int(r[3])] for r in sec["Bonds"]],
        "nangletypes": hdr["angle types"],
        "angles": [[int(r[1]), int(r[2]), int(r[3]), int(r[4])] for r in sec["Angles"]],
        # settings of the harness the known answers were generated with (values only):
        "units": "real", "timestep": 0.1, "special_lj": [0.10, 0.25, 0.50],
        "neigh_modify": {"delay": 2, "every": 2, "check": 0},
    }
    json.dump(four, open(OUT + "/fourmol.json", "w"))
    lj = known_answers(ft + "mol-pair-lj_cut.yaml", "vdwl")
    lj.update({"pair_style": "lj/cut", "cut_global": 8.0, "mix": "arithmetic"})
    json.dump(lj, open(OUT + "/lj_cut.json", "w"))
    json.dump(known_answers(ft + "bond-fene.yaml", "energy"), open(OUT + "/bond_fene.json", "w"))
    json.dump(known_answers(ft + "bond-harmonic.yaml", "energy"), open(OUT + "/bond_harmonic.json", "w"))
    json.dump(known_answers(ft + "bond-hybrid.yaml", "energy"), open(OUT + "/bond_hybrid.json", "w"))
    json.dump(known_answers(ft + "angle-harmonic.yaml", "energy"), open(OUT + "/angle_harmonic.json", "w"))
    json.dump(known_answers(ft + "angle-cosine.yaml", "energy"), open(OUT + "/angle_cosine.json", "w"))
    text = open(ft + "fix-timestep-nve.yaml").read()
    nve = {
        "epsilon": yaml_scalar(text, "epsilon"),
        "run_pos": [r[1:] for r in sorted(yaml_block(text, "run_pos"))],
        "run_vel": [r[1:] for r in sorted(yaml_block(text, "run_vel"))],
        # the force field and run protocol of the harness (values only)
        "pair_style": "lj/cut", "cut_global": 8.0, "mix": "geometric",
        "pair_coeff": [[1, 1, 0.02, 2.5], [2, 2, 0.005, 1.0], [2, 4, 0.005, 0.5], [3, 3, 0.02, 3.2], [4, 4, 0.015, 3.1], [5, 5, 0.015, 3.1]],
        "bond_coeff": [[1, 250.0, 1.5], [2, 300.0, 1.1], [3, 350.0, 1.3], [4, 650.0, 1.2], [5, 450.0, 1.0]],
        "angle_coeff": [[1, 75.0, 110.1], [2, 45.0, 111.0], [3, 50.0, 120.0], [4, 100.0, 108.5]],
        "group_molecules": [1, 2], "timestep": 0.25, "nsteps": 8,
    }
    json.dump(nve, open(OUT + "/fix_nve.json", "w"))

    hdr, sec = parse_data(REF + "/bench/data.chain")
    a = np.array(sec["Atoms"], dtype=object)  # id mol type x y z ix iy iz  (atom_style bond)
    tag = a[:, 0].astype(np.int32)
    order = np.argsort(tag)
    assert (order == np.arange(len(tag))).all(), "data.chain atoms expected in id order"
    v = np.array(sec["Velocities"], dtype=object)
    vtag = v[:, 0].astype(np.int32)
    vel = np.zeros((len(tag), 3))
    vel[vtag - 1] = v[:, 1:4].astype(np.float64)
    b = np.array(sec["Bonds"], dtype=np.int32)
    np.savez_compressed(
        OUT + "/chain32k.npz",
        box=np.array([hdr["x"], hdr["y"], hdr["z"]], dtype=np.float64),
        tag=tag, mol=a[:, 1].astype(np.int32), type=a[:, 2].astype(np.int32),
        x=a[:, 3:6].astype(np.float64), image=a[:, 6:9].astype(np.int32), v=vel,
        bonds=b[:, 1:4], mass=np.array([float(sec["Masses"][0][1])]),
    )
    log = open(REF + "/bench/log.6Oct16.chain.fixed.icc.1").read().split("\n")
    rows = [ln.split() for ln in log if re.match(r"^\s+(0|100)\s", ln)]
    thermo = {
        # settings of the benchmark the log was produced with (values only)
        "units": "lj", "special_lj": [0.0, 1.0, 1.0], "skin": 0.4, "every": 1, "delay": 1,
        "bond_fene": [30.0, 1.5, 1.0, 1.0], "lj_cut": 1.12, "shift": True, "pair_coeff": [1.0, 1.0, 1.12],
        "langevin": [1.0, 1.0, 10.0, 904297], "timestep": 0.012, "nsteps": 100,
        "columns": ["Step", "Temp", "E_pair", "E_mol", "TotEng", "Press"],
        "thermo": [[float(v) for v in r] for r in rows],
        "neighbors": int(re.search(r"Total # of neighbors = (\d+)", "\n".join(log)).group(1)),
        "builds": int(re.search(r"Neighbor list builds = (\d+)", "\n".join(log)).group(1)),
    }
    json.dump(thermo, open(OUT + "/chain32k_thermo.json", "w"), indent=1)
    print("wrote fixtures to", OUT)


if __name__ == "__main__":
    sys.exit(main())
