// fixes.cpp — fix styles of the hot path, registered under the reference's style names
// (src/fix_nve.h, src/fix_langevin.h, src/USER-LE/fix_extrusion.h:14-18, fix_ex_load.h, fix_ex_unload.h).
// Argument grammars and error strings follow the reference constructors.
#include <cstring>

#include "comm.h"
#include "device.h"

namespace lmp_le {

void dd_gather_positions(DeviceState &d, Comm &comm);
void dd_gather_needed(DeviceState &d, Comm &comm, int btype, bool with_nbrs);

static int fix_group(Engine *e, const std::vector<std::string> &arg) {      // src/fix.cpp:67-69
  const int bit = e->group_bit(arg[1]);
  if (!bit) throw LammpsError("Could not find fix group ID");
  return bit;
}

// The LE fixes count on every bond being stored with BOTH of its atoms (`num_bond == 2` for a free backbone bead,
// fix_ex_load.cpp:481-483; bondcount per owned end, fix_extrusion.cpp:281-295).  With newton_bond on the reference stores a
// bond once, those tests never pass and nothing is ever loaded; this engine always stores both copies, so it would NOT
// reproduce that run: refuse instead of silently differing.
static void need_newton_bond_off(Engine *e, const std::string &style) {
  if (e->newton_bond)
    throw LammpsError("fix " + style + " needs newton_bond off (`newton off` or `newton on off`): with newton_bond on the "
                      "reference stores each bond on one atom only and the fix's bond counts never match");
}
// fix ID group nve  (src/fix_nve.cpp:33-45)
FixNVE::FixNVE(Engine *e, const std::vector<std::string> &arg) {
  eng = e; id = arg[0]; group = arg[1]; style = arg[2];
  if (arg.size() < 3) throw LammpsError("Illegal fix nve command");
  groupbit = fix_group(e, arg);
  has_initial_integrate = has_final_integrate = true;
}

// fix ID group langevin Tstart Tstop damp seed  (src/fix_langevin.cpp:54-196)
FixLangevin::FixLangevin(Engine *e, const std::vector<std::string> &arg) {
  eng = e; id = arg[0]; group = arg[1]; style = arg[2];
  if (arg.size() < 7) throw LammpsError("Illegal fix langevin command");
  groupbit = fix_group(e, arg);
  t_start = numeric(arg[3]);
  t_stop = numeric(arg[4]);
  t_period = numeric(arg[5]);
  seed = inumeric(arg[6]);
  if (t_period <= 0.0) throw LammpsError("Fix langevin period must be > 0.0");
  if (seed <= 0) throw LammpsError("Illegal fix langevin command");
  // optional keywords (src/fix_langevin.cpp:105-155): `scale` and `zero` are implemented, the others only at their defaults
  ratio.assign(e->ntypes + 1, 1.0);
  for (size_t k = 7; k < arg.size();) {
    const std::string &kw = arg[k];
    if (kw == "scale") {
      if (k + 3 > arg.size()) throw LammpsError("Illegal fix langevin command");
      int itype = inumeric(arg[k + 1]);
      double sc = numeric(arg[k + 2]);
      if (itype <= 0 || itype > e->ntypes) throw LammpsError("Illegal fix langevin command");
      ratio[itype] = sc;
      k += 3;
    } else if (kw == "zero" || kw == "tally" || kw == "omega" || kw == "angmom" || kw == "gjf") {
      if (k + 2 > arg.size()) throw LammpsError("Illegal fix langevin command");
      const std::string &val = arg[k + 1];
      if (kw == "zero") {
        if (val != "yes" && val != "no") throw LammpsError("Illegal fix langevin command");
        zeroflag = val == "yes";
      } else if (val != "no")
        throw LammpsError("MI355X engine: fix langevin " + kw + " " + val + " is not supported");
      k += 2;
    } else throw LammpsError("Illegal fix langevin command");
  }
  rng.seed(seed);   // seed + comm->me, me = 0
  has_post_force = true;
}

// fix ID group extrusion N1 neutral ctcf_left ctcf_right through_prob btype [ctcf_left_right]
// (src/USER-LE/fix_extrusion.cpp:38-142; narg < 8 is rejected but arg[8] is read, so 9 are needed)
FixExtrusion::FixExtrusion(Engine *e, const std::vector<std::string> &arg) {
  eng = e; id = arg[0]; group = arg[1]; style = arg[2];
  if (arg.size() < 9) throw LammpsError("Illegal fix extrusion command");
  groupbit = fix_group(e, arg);
  nevery = inumeric(arg[3]);
  if (nevery <= 0) throw LammpsError("Illegal fix extrusion command, n_steps <= 0");
  neutral = inumeric(arg[4]); ctcf_left = inumeric(arg[5]); ctcf_right = inumeric(arg[6]);
  if (neutral < 1 || neutral > e->ntypes || ctcf_left < 1 || ctcf_left > e->ntypes || ctcf_right < 1 ||
      ctcf_right > e->ntypes)
    throw LammpsError("Invalid atom type (CTCF) in fix extrusion command");
  through_prob = numeric(arg[7]);
  if (through_prob < 0 || through_prob > 1)
    throw LammpsError("Invalid probability to pass through CTCF in fix extrusion command");
  btype = inumeric(arg[8]);
  if (btype < 1 || btype > e->nbondtypes) throw LammpsError("Invalid atom type in fix extrusion command");
  ctcf_lr = (arg.size() == 10) ? inumeric(arg[9]) : -1;
  e->say("Attention! Type of bidirectional CTCF is " + std::to_string(ctcf_lr) + "\n");
  e->say("Amount of args in loop extrusion is " + std::to_string(arg.size()) + "\n");
  if (ctcf_lr > e->ntypes) throw LammpsError("Invalid atom type in fix extrusion command");
  if (e->atom_style == "atomic") throw LammpsError("Cannot use fix extrusion with non-molecular systems");
  need_newton_bond_off(e, style);
  rng.seed(12345);   // hard-coded 12345 + me (:98-99)
  e->say("Attention! maxspecial = " + std::to_string(e->maxspecial) + "\n");
  force_reneighbor = true;
  has_post_integrate = true;
}
double FixExtrusion::compute_vector(int n) { return n == 0 ? (double)last_break : 0.0; }   // :1496-1501

// fix ID group ex_load Nevery itype jtype Rmin bondtype [iparam ..] [jparam ..] [prob f seed] (:39-176)
FixExLoad::FixExLoad(Engine *e, const std::vector<std::string> &arg) {
  eng = e; id = arg[0]; group = arg[1]; style = arg[2];
  stock = (style == "bond/create");
  phase = stock ? 0 : 3;                                  // src/MC/fix_bond_create.cpp:356 vs src/USER-LE/fix_ex_load.cpp:338
  const std::string ill = "Illegal fix " + style + " command";
  if (arg.size() < 8) throw LammpsError(ill);
  groupbit = fix_group(e, arg);
  nevery = inumeric(arg[3]);
  if (nevery <= 0) throw LammpsError(ill);
  iatomtype = inumeric(arg[4]); jatomtype = inumeric(arg[5]);
  double cutoff = numeric(arg[6]);
  btype = inumeric(arg[7]);
  if (iatomtype < 1 || iatomtype > e->ntypes || jatomtype < 1 || jatomtype > e->ntypes)
    throw LammpsError("Invalid atom type in fix " + style + " command");
  if (cutoff < 0.0) throw LammpsError(ill);
  if (btype < 1 || btype > e->nbondtypes) throw LammpsError("Invalid bond type in fix " + style + " command");
  cutsq = cutoff * cutoff;
  inewtype = iatomtype; jnewtype = jatomtype;
  size_t iarg = 8;
  while (iarg < arg.size()) {
    if (arg[iarg] == "iparam") {
      if (iarg + 3 > arg.size()) throw LammpsError(ill);
      imaxbond = inumeric(arg[iarg + 1]); inewtype = inumeric(arg[iarg + 2]);
      if (imaxbond < 0) throw LammpsError(ill);
      if (inewtype < 1 || inewtype > e->ntypes) throw LammpsError("Invalid atom type in fix " + style + " command");
      iarg += 3;
    } else if (arg[iarg] == "jparam") {
      if (iarg + 3 > arg.size()) throw LammpsError(ill);
      jmaxbond = inumeric(arg[iarg + 1]); jnewtype = inumeric(arg[iarg + 2]);
      if (jmaxbond < 0) throw LammpsError(ill);
      if (jnewtype < 1 || jnewtype > e->ntypes) throw LammpsError("Invalid atom type in fix " + style + " command");
      iarg += 3;
    } else if (arg[iarg] == "prob") {
      if (iarg + 3 > arg.size()) throw LammpsError(ill);
      fraction = numeric(arg[iarg + 1]); seed = inumeric(arg[iarg + 2]);
      if (fraction < 0.0 || fraction > 1.0) throw LammpsError(ill);
      if (seed <= 0) throw LammpsError(ill);
      iarg += 3;
    } else if (arg[iarg] == "atype" || arg[iarg] == "dtype" || arg[iarg] == "itype") {
      if (iarg + 2 > arg.size()) throw LammpsError(ill);
      // fix_ex_load.cpp:108-122 parse the type; :236-254 (and MC/fix_bond_create.cpp:256-270) switch the creation of angles /
      // dihedrals / impropers on only `if (atype && force->angle)`, i.e. when the script defined such a style.  Angles are
      // created (kernels_le.hip dev_create_angles); this engine has no dihedral / improper styles, so dtype / itype stay
      // without effect exactly as in the reference for a script without such styles.
      if (inumeric(arg[iarg + 1]) < 0) throw LammpsError(ill);
      if (arg[iarg] == "atype") atype = inumeric(arg[iarg + 1]);
      iarg += 2;
    } else throw LammpsError(ill);
  }
  if (e->atom_style == "atomic") throw LammpsError("Cannot use fix " + style + " with non-molecular systems");
  need_newton_bond_off(e, style);
  if (iatomtype == jatomtype && (imaxbond != jmaxbond || inewtype != jnewtype))
    throw LammpsError("Inconsistent iparam/jparam values in fix " + style + " command");
  rng.seed(seed);
  force_reneighbor = true;
  has_post_integrate = true;
}
void FixExLoad::init() {
  // src/USER-LE/fix_ex_load.cpp:217-218
  if (!eng->pair_lj) throw LammpsError("Fix " + style + " cutoff is longer than pairwise cutoff");
  int nt = eng->ntypes + 1;
  if (cutsq > eng->cutsq[iatomtype * nt + jatomtype])
    throw LammpsError("Fix " + style + " cutoff is longer than pairwise cutoff");
}
// FixBondCreate::setup (src/MC/fix_bond_create.cpp:302-345): count the bonds of btype each bead stores, once
void FixExLoad::setup() {
  if (!stock || counted) return;
  counted = true;
  const bool was_current = eng->host_current;
  eng->download();                   // setup runs with the device already holding the newest state
  eng->host_current = was_current;   // ... and the run that follows will change it again
  bondcount.assign((size_t)eng->natoms + 2, 0);
  for (int i = 0; i < eng->natoms; i++)
    for (int m = 0; m < eng->num_bond[i]; m++)
      if (eng->bond_type[(size_t)i * eng->bpa + m] == btype) bondcount[i + 1]++;
}
double FixExLoad::compute_vector(int n) { return n == 0 ? (double)last_create : (double)total_create; }

// fix ID group ex_unload Nevery bondtype Rmax [prob fraction seed]  (src/USER-LE/fix_ex_unload.cpp:34-115)
FixExUnload::FixExUnload(Engine *e, const std::vector<std::string> &arg) {
  eng = e; id = arg[0]; group = arg[1]; style = arg[2];
  // the reference's two files are the same text but for the firing step (src/MC/fix_bond_break.cpp:178 `% nevery`,
  // src/USER-LE/fix_ex_unload.cpp:178 `% nevery - 2`)
  phase = (style == "bond/break") ? 0 : 2;
  const std::string ill = "Illegal fix " + style + " command";
  if (arg.size() < 6) throw LammpsError(ill);
  groupbit = fix_group(e, arg);
  nevery = inumeric(arg[3]);
  if (nevery <= 0) throw LammpsError(ill);
  btype = inumeric(arg[4]);
  double cutoff = numeric(arg[5]);
  if (btype < 1 || btype > e->nbondtypes) throw LammpsError("Invalid bond type in fix " + style + " command");
  if (cutoff < 0.0) throw LammpsError(ill);
  cutsq = cutoff * cutoff;
  size_t iarg = 6;
  while (iarg < arg.size()) {
    if (arg[iarg] == "prob") {
      if (iarg + 3 > arg.size()) throw LammpsError(ill);
      fraction = numeric(arg[iarg + 1]); seed = inumeric(arg[iarg + 2]);
      if (fraction < 0.0 || fraction > 1.0) throw LammpsError(ill);
      if (seed <= 0) throw LammpsError(ill);
      iarg += 3;
    } else throw LammpsError(ill);
  }
  need_newton_bond_off(e, style);
  rng.seed(seed);
  force_reneighbor = true;
  has_post_integrate = true;
}
void FixExUnload::init() { angleflag = (eng->apa > 0 && eng->nangles > 0) ? 1 : 0; }
double FixExUnload::compute_vector(int n) { return n == 0 ? (double)last_break : (double)total_break; }

// ---------------------------------------------------------------------------------------------
// post_integrate: firing rule + device launch + counter read-back (firing steps only)
// ---------------------------------------------------------------------------------------------
static int fix_index(Engine *e, Fix *f) {
  for (size_t k = 0; k < e->fixes.size(); k++) if (e->fixes[k].get() == f) return (int)k;
  return -1;
}
static int le_slot(Engine *e, Fix *f) {
  int s = 0;
  for (auto &g : e->fixes) {
    if (g.get() == f) return s;
    if (g->force_reneighbor) s++;
  }
  return s;
}
static void check_le_error(DeviceState &d, const char *style) {
  int code = d.flags_h[FLAG_ERROR];
  if (!code) return;
  std::string st = style;
  switch (code) {
    case ERR_EXT_MULTI: throw LammpsError("Fix extrusion, more than one bond type 2");
    case ERR_BPA: throw LammpsError("New bond exceeded bonds per atom in fix " + st);
    case ERR_SPECIAL: throw LammpsError("New bond exceeded special list size in fix " + st);
    case ERR_COUNT_MISMATCH: throw LammpsError("Numbers of created and broken bonds are not equal");
    case ERR_SPECIAL_SCRATCH: throw LammpsError("Special list size exceeded in fix bond/create");
    case ERR_ANGLES: throw LammpsError("Fix " + st + " induced too many angles/dihedrals/impropers per atom");
    default: throw LammpsError("device error in fix " + st);
  }
}

void FixExtrusion::post_integrate() {
  if (eng->ntimestep % nevery - 1) return;               // src/USER-LE/fix_extrusion.cpp:265
  DeviceState &d = *eng->dev;
  int slot = le_slot(eng, this);
  if (slot >= LE_MAX_FIXES) throw LammpsError("MI355X engine supports at most " + std::to_string(LE_MAX_FIXES) + " extrusion/ex_load/ex_unload fixes");
  if (!rng_on_device) { le_rng_upload(d, slot, rng); rng_on_device = true; }
  if (eng->world > 1) {
    // stored coordinates of the extruder ends and the beads they can step to: O(extruders) rows (dd_gather_needed); every
    // bead's (tag, x, xhold) only when the visit order is not the canonical one
    if (dd_le_fast(d)) dd_gather_needed(d, *eng->comm, btype, true); else dd_gather_positions(d, *eng->comm);
  }
  ExtrusionParams p{neutral, ctcf_left, ctcf_right, ctcf_lr, btype, through_prob, groupbit};
  launch_extrusion(d, p, slot);
  d.topo_dirty = true;
  d.bond_pack_dirty = true;
  d.angle_pack_dirty = true;
  sync_flags(d);
  check_le_error(d, "extrusion");
  last_break = d.flags_h[FLAG_COUNT_A];
  int created = d.flags_h[FLAG_COUNT_B];
  if (last_break != created) throw LammpsError("Numbers of created and broken bonds are not equal");
  if (last_break) eng->le_reneigh_step[fix_index(eng, this)] = eng->ntimestep;
}

void FixExLoad::post_integrate() {
  if (eng->ntimestep % nevery - phase) return;           // src/USER-LE/fix_ex_load.cpp:338, src/MC/fix_bond_create.cpp:356
  DeviceState &d = *eng->dev;
  int slot = le_slot(eng, this);
  if (slot >= LE_MAX_FIXES) throw LammpsError("MI355X engine supports at most " + std::to_string(LE_MAX_FIXES) + " extrusion/ex_load/ex_unload fixes");
  if (!rng_on_device) { le_rng_upload(d, slot, rng); rng_on_device = true; }
  // angles around new bonds only `if (atype && force->angle)` (fix_ex_load.cpp:236-240)
  const int angle_type_new = (atype > 0 && eng->angles_active()) ? atype : 0;
  if (angle_type_new > eng->nangletypes) throw LammpsError("Fix " + style + " angle type is invalid");
  ExLoadParams p{iatomtype, jatomtype, imaxbond, inewtype, jmaxbond, jnewtype, btype, cutsq, fraction, angle_type_new, groupbit};
  if (stock) {
    if (eng->world > 1) dd_gather_positions(d, *eng->comm);   // current positions by tag (ghost slots lag one step here)
    launch_bond_create(d, p, slot, bondcount.data(), (int)bondcount.size(), eng->comm);
  } else {
    if (eng->world > 1) {
      // canonical order: every rank decides the pairs whose storing bead it owns from its own list and (current) ghosts
      if (dd_le_fast(d)) eng->halo_exchange_once(); else dd_gather_positions(d, *eng->comm);
    }
    launch_ex_load(d, p, slot, eng->comm);
  }
  d.topo_dirty = true;
  d.bond_pack_dirty = true;
  d.angle_pack_dirty = true;
  sync_flags(d);
  check_le_error(d, style.c_str());
  last_create = d.flags_h[FLAG_COUNT_A];
  total_create += last_create;
  eng->nbonds += last_create;
  if (angle_type_new) eng->nangles += d.flags_h[FLAG_COUNT_B] / 3;     // copies on three atoms each (fix_ex_load.cpp:760-765)
  if (last_create) eng->le_reneigh_step[fix_index(eng, this)] = eng->ntimestep;
  if (stock && last_create) bond_create_counts(d, bondcount.data(), (int)bondcount.size());
}

void FixExUnload::post_integrate() {
  if (eng->ntimestep % nevery - phase) return;           // src/USER-LE/fix_ex_unload.cpp:178, src/MC/fix_bond_break.cpp:178
  DeviceState &d = *eng->dev;
  int slot = le_slot(eng, this);
  if (slot >= LE_MAX_FIXES) throw LammpsError("MI355X engine supports at most " + std::to_string(LE_MAX_FIXES) + " extrusion/ex_load/ex_unload fixes");
  if (!rng_on_device) { le_rng_upload(d, slot, rng); rng_on_device = true; }
  if (eng->world > 1) {
    if (dd_le_fast(d)) dd_gather_needed(d, *eng->comm, btype, false); else dd_gather_positions(d, *eng->comm);
  }
  ExUnloadParams p{btype, cutsq, fraction, angleflag, groupbit};
  launch_ex_unload(d, p, slot);
  d.topo_dirty = true;
  d.bond_pack_dirty = true;
  d.angle_pack_dirty = true;
  sync_flags(d);
  check_le_error(d, style.c_str());
  last_break = d.flags_h[FLAG_COUNT_A];
  total_break += last_break;
  eng->nbonds -= last_break;
  if (angleflag) eng->nangles -= d.flags_h[FLAG_COUNT_B] / 3;
  if (last_break) eng->le_reneigh_step[fix_index(eng, this)] = eng->ntimestep;
}

}  // namespace lmp_le
