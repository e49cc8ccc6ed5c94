// capi.cpp — extern "C" boundary (include/lammps_le.h): the reference's library.h subset for this path.
// Behaviour follows src/library.cpp; every entry point catches LammpsError and records it
// (the reference does the same when built with LAMMPS_EXCEPTIONS, src/library.cpp BEGIN_CAPTURE/END_CAPTURE).
#include <cstdlib>
#include <cstring>

#include "../../include/lammps_le.h"
#include "comm.h"
#include "device.h"

namespace lmp_le { unsigned dd_halo_mismatches(DeviceState &d); }
using namespace lmp_le;

#define BEGIN_CAPTURE Engine *e = (Engine *)handle; try {
#define END_CAPTURE } catch (const std::exception &ex) { e->last_error = ex.what(); e->has_error = true; \
    if (e->screen) { fprintf(e->screen, "ERROR: %s\n", ex.what()); fflush(e->screen); } \
    if (e->logfile) { fprintf(e->logfile, "ERROR: %s\n", ex.what()); fflush(e->logfile); } }

extern "C" {

void *lammps_open_no_mpi(int argc, char **argv, void **ptr) {
  Engine *e = nullptr;
  try { e = new Engine(argc, argv); } catch (const std::exception &ex) { fprintf(stderr, "LAMMPS Exception: %s\n", ex.what()); }
  if (ptr) *ptr = (void *)e;
  return (void *)e;
}
void lammps_close(void *handle) { delete (Engine *)handle; }

void lammps_file(void *handle, const char *file) { BEGIN_CAPTURE e->file(file); END_CAPTURE }
char *lammps_command(void *handle, const char *cmd) {
  char *result = nullptr;
  BEGIN_CAPTURE result = (char *)e->one(cmd); END_CAPTURE
  return result;
}
void lammps_commands_list(void *handle, int ncmd, const char **cmds) {
  for (int i = 0; i < ncmd; i++) {
    lammps_command(handle, cmds[i]);
    if (((Engine *)handle)->has_error) return;
  }
}
void lammps_commands_string(void *handle, const char *str) {
  std::string s(str), line;
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t nl = s.find('\n', pos);
    line = s.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
    lammps_command(handle, line.c_str());
    if (((Engine *)handle)->has_error || nl == std::string::npos) return;
    pos = nl + 1;
  }
}

double lammps_get_natoms(void *handle) { return (double)((Engine *)handle)->natoms; }

double lammps_get_thermo(void *handle, const char *keyword) {
  double val = 0.0;
  BEGIN_CAPTURE
    const ThermoRow &r = e->last_thermo;
    std::string k = keyword;
    bool isint;
    if (k == "step") val = (double)e->ntimestep;
    else if (k.rfind("f_", 0) == 0 || !e->thermo_keyword(r, k, val, isint))
      throw LammpsError("Unknown keyword in thermo_style custom command: " + k);
  END_CAPTURE
  return val;
}

void lammps_extract_box(void *handle, double *boxlo, double *boxhi, double *xy, double *yz, double *xz, int *pflags,
                        int *boxflag) {
  Engine *e = (Engine *)handle;
  for (int d = 0; d < 3; d++) {
    if (boxlo) boxlo[d] = e->box.lo[d];
    if (boxhi) boxhi[d] = e->box.hi[d];
    if (pflags) pflags[d] = 1;
  }
  if (xy) *xy = 0.0;
  if (yz) *yz = 0.0;
  if (xz) *xz = 0.0;
  if (boxflag) *boxflag = 0;
}

int lammps_extract_setting(void *handle, const char *keyword) {
  Engine *e = (Engine *)handle;
  std::string k = keyword;
  if (k == "bigint") return 8;
  if (k == "tagint" || k == "imageint") return 4;
  if (k == "nlocal" || k == "nall") return e->natoms;
  if (k == "ntypes") return e->ntypes;
  if (k == "nbondtypes") return e->nbondtypes;
  if (k == "bond_per_atom") return e->bpa;
  if (k == "angle_per_atom") return e->apa;
  if (k == "nangles") return (int)e->nangles;
  if (k == "nangletypes") return e->nangletypes;
  if (k == "maxspecial") return e->maxspecial;
  if (k == "newton_bond") return 0;
  if (k == "molecule_flag") return e->atom_style != "atomic";
  if (k == "dimension") return 3;
  if (k == "box_exist") return e->box_exist;
  return -1;
}

void *lammps_extract_global(void *handle, const char *name) {
  Engine *e = (Engine *)handle;
  std::string k = name;
  if (k == "dt") return &e->dt;
  if (k == "ntimestep") return &e->ntimestep;
  if (k == "atime") return &e->atime;              // src/library.cpp:1332-1333
  if (k == "atimestep") return &e->atimestep;
  if (k == "boxlo") return e->box.lo;
  if (k == "boxhi") return e->box.hi;
  if (k == "natoms") { e->scratch_scalar = e->natoms; return &e->scratch_scalar; }
  if (k == "nbonds") return &e->nbonds;
  if (k == "ntypes") return &e->ntypes;
  if (k == "boltz") return &e->boltz;
  if (k == "units") return (void *)e->units.c_str();
  return nullptr;
}

void *lammps_extract_atom(void *handle, const char *name) {
  void *result = nullptr;
  BEGIN_CAPTURE
    e->download();
    std::string k = name;
    auto rows = [&](std::vector<double> &a) {
      e->scratch_rows.resize(e->natoms);
      for (int i = 0; i < e->natoms; i++) e->scratch_rows[i] = &a[3 * (size_t)i];
      return (void *)e->scratch_rows.data();
    };
    if (k == "x") result = rows(e->x);
    else if (k == "v") result = rows(e->v);
    else if (k == "f") result = rows(e->f);
    else if (k == "type") result = e->type.data();
    else if (k == "mass") result = e->mass.data();
    else if (k == "id") {
      auto &a = e->scratch_i["id"]; a.resize(e->natoms);
      for (int i = 0; i < e->natoms; i++) a[i] = i + 1;
      result = a.data();
    } else if (k == "mask") {
      auto &a = e->scratch_i["mask"]; a.assign(e->natoms, 1);       // bit 0 = all; further bits = the groups in definition order
      if ((int)e->gmask.size() == e->natoms) for (int i = 0; i < e->natoms; i++) a[i] = e->gmask[i] | 1;
      result = a.data();
    } else if (k == "image") {
      auto &a = e->scratch_i["image"]; a.resize(e->natoms);
      for (int i = 0; i < e->natoms; i++) a[i] = lammps_encode_image_flags(e->image[3 * i], e->image[3 * i + 1], e->image[3 * i + 2]);
      result = a.data();
    } else if (k == "molecule") result = e->molecule.data();
  END_CAPTURE
  return result;
}

void *lammps_extract_fix(void *handle, char *id, int style, int type, int nrow, int /*ncol*/) {
  void *result = nullptr;
  BEGIN_CAPTURE
    Fix *f = e->find_fix(id);
    if (!f) throw LammpsError(std::string("Could not find fix ID ") + id);
    if (style != 0 || type != 1) throw LammpsError("MI355X engine: only global fix vectors can be extracted");
    double *d = (double *)malloc(sizeof(double));
    *d = f->compute_vector(nrow);
    result = d;
  END_CAPTURE
  return result;
}

static int topo_width(Engine *e, const std::string &k) {
  if (k == "num_bond") return 1;
  if (k == "bond_type" || k == "bond_atom") return e->bpa;
  if (k == "nspecial") return 3;
  if (k == "special") return e->maxspecial;
  if (k == "num_angle") return e->apa > 0 ? 1 : 0;
  if (k == "angle_type" || k == "angle_atom1" || k == "angle_atom2" || k == "angle_atom3") return e->apa;
  return 0;
}

void lammps_gather_atoms(void *handle, char *name, int type, int count, void *data) {
  BEGIN_CAPTURE
    e->download();
    std::string k = name;
    int n = e->natoms;
    if (type == 1) {
      std::vector<double> *src = (k == "x") ? &e->x : (k == "v") ? &e->v : (k == "f") ? &e->f : nullptr;
      if (!src || count != 3) throw LammpsError("lammps_gather_atoms: unknown property name " + k);
      memcpy(data, src->data(), 3 * (size_t)n * sizeof(double));
    } else {
      int *out = (int *)data;
      if (k == "type" && count == 1) memcpy(out, e->type.data(), n * sizeof(int));
      else if (k == "id" && count == 1) for (int i = 0; i < n; i++) out[i] = i + 1;
      else if (k == "mask" && count == 1) for (int i = 0; i < n; i++) out[i] = (int)e->gmask.size() == n ? (e->gmask[i] | 1) : 1;
      else if (k == "molecule" && count == 1) memcpy(out, e->molecule.data(), n * sizeof(int));
      else if (k == "image" && count == 3) memcpy(out, e->image.data(), 3 * (size_t)n * sizeof(int));
      else if (k == "image" && count == 1)
        for (int i = 0; i < n; i++) out[i] = lammps_encode_image_flags(e->image[3 * i], e->image[3 * i + 1], e->image[3 * i + 2]);
      else if (topo_width(e, k) == count && count > 0) {
        const std::vector<int> &src = (k == "num_bond") ? e->num_bond : (k == "bond_type") ? e->bond_type :
                                      (k == "bond_atom") ? e->bond_atom : (k == "nspecial") ? e->nspecial :
                                      (k == "num_angle") ? e->num_angle : (k == "angle_type") ? e->angle_type :
                                      (k == "angle_atom1") ? e->angle_a1 : (k == "angle_atom2") ? e->angle_a2 :
                                      (k == "angle_atom3") ? e->angle_a3 : e->special;
        memcpy(out, src.data(), (size_t)n * count * sizeof(int));
      } else throw LammpsError("lammps_gather_atoms: unknown property name " + k);
    }
  END_CAPTURE
}

void lammps_scatter_atoms(void *handle, char *name, int type, int count, void *data) {
  BEGIN_CAPTURE
    e->download();
    std::string k = name;
    int n = e->natoms;
    if (type == 1 && count == 3 && (k == "x" || k == "v" || k == "f")) {
      std::vector<double> &dst = (k == "x") ? e->x : (k == "v") ? e->v : e->f;
      memcpy(dst.data(), data, 3 * (size_t)n * sizeof(double));
    } else if (type == 0 && count == 1 && k == "type") memcpy(e->type.data(), data, n * sizeof(int));
    else if (type == 0 && count == 3 && k == "image") memcpy(e->image.data(), data, 3 * (size_t)n * sizeof(int));
    else throw LammpsError("lammps_scatter_atoms: unknown property name " + k);
    e->dev_current = false;   // next run re-uploads
  END_CAPTURE
}

int lammps_version(void *) { return 20201029; }
/* src/library.cpp lammps_encode_image_flags: 10 bits per dimension, offset 512 (LAMMPS_SMALLBIG) */
int lammps_encode_image_flags(int ix, int iy, int iz) {
  return ((ix + 512) & 1023) | (((iy + 512) & 1023) << 10) | (((iz + 512) & 1023) << 20);
}
void lammps_decode_image_flags(int image, int *flags) {
  flags[0] = (image & 1023) - 512;
  flags[1] = ((image >> 10) & 1023) - 512;
  flags[2] = (image >> 20) - 512;
}
void lammps_free(void *ptr) { free(ptr); }
int lammps_is_running(void *) { return 0; }
int lammps_has_error(void *handle) { return ((Engine *)handle)->has_error ? 1 : 0; }
int lammps_get_last_error_message(void *handle, char *buffer, int buf_size) {
  Engine *e = (Engine *)handle;
  if (!e->has_error) { if (buf_size > 0) buffer[0] = '\0'; return 0; }
  snprintf(buffer, buf_size, "%s", e->last_error.c_str());
  e->has_error = false;
  e->last_error.clear();
  return 1;   // 1 = normal (recoverable) error, as ERROR_NORMAL in src/library.cpp
}
int lammps_config_has_exceptions(void) { return 1; }
int lammps_has_style(void *, const char *category, const char *name) {
  std::string c = category, s = name;
  if (c == "fix") return s == "nve" || s == "langevin" || s == "extrusion" || s == "ex_load" || s == "ex_unload" || s == "bond/break" || s == "bond/create";
  if (c == "pair") return s == "lj/cut" || s == "zero" || s == "none";
  if (c == "dump") return s == "atom" || s == "custom" || s == "local" || s == "dcd";
  if (c == "compute") return s == "property/local";
  if (c == "bond") return s == "fene" || s == "harmonic" || s == "hybrid" || s == "zero" || s == "none";
  if (c == "atom") return s == "bond" || s == "molecular" || s == "atomic" || s == "full" || s == "angle";
  return 0;
}

// every thermo line the engine has printed since it was opened, as numbers: rows of 7 doubles (step, temp, epair, emol,
// etotal, press, bonds).  Returns the number of rows there are; writes at most max_rows of them.
int lammps_le_thermo_log(void *handle, double *out, int max_rows) {
  Engine *e = (Engine *)handle;
  const int n = (int)e->thermo_log.size();
  for (int k = 0; k < n && k < max_rows; k++) {
    const ThermoRow &r = e->thermo_log[(size_t)k];
    double *o = out + 7 * (size_t)k;
    o[0] = (double)r.step; o[1] = r.temp; o[2] = r.epair; o[3] = r.emol; o[4] = r.etotal; o[5] = r.press; o[6] = (double)r.nbonds;
  }
  return n;
}
double lammps_le_stat(void *handle, const char *name) {
  Engine *e = (Engine *)handle;
  std::string k = name;
  if (k == "loop_time") return e->loop_time;
  if (k == "neigh_builds") return (double)e->neigh_builds;
  if (k == "neigh_time" || k == "time_neigh") return e->timers[Engine::T_NEIGH];
  if (k == "time_pair") return e->timers[Engine::T_PAIR];
  if (k == "time_bond") return e->timers[Engine::T_BOND];
  if (k == "time_comm") return e->timers[Engine::T_COMM];
  if (k == "time_output") return e->timers[Engine::T_OUTPUT];
  if (k == "time_modify") return e->timers[Engine::T_MODIFY];
  if (k == "time_other") {
    double all = 0.0;
    for (int s = 0; s < Engine::T_NSECT; s++) all += e->timers[s];
    return e->loop_time - all;
  }
  if (k == "comm_nranks") return e->comm ? (double)e->comm->nranks() : 1.0;
  if (k == "rng_late_generations") return (double)e->rng_late_count;
  if (k == "rng_segments_held") return e->dev ? (double)rng_segments_held(*e->dev) : 0.0;
  if (k == "rng_segments") return e->dev ? (double)e->dev->rng_nseg : 0.0;
  if (k == "comm_bytes_allgather") return e->comm ? e->comm->bytes_allgather : 0.0;
  if (k == "comm_bytes_allreduce") return e->comm ? e->comm->bytes_allreduce : 0.0;
  if (k == "halo_window_mismatches") return e->dev ? (double)dd_halo_mismatches(*e->dev) : 0.0;   // LAMMPS_LE_FAST_HALO_VERIFY
  if (k == "angle_records" || k == "angle_records_max") {     // listed-angle records of the owned beads (sum / longest), as of the last reneighbor
    if (!e->dev || !e->dev->eff_n || e->dev->n <= 0) return 0.0;
    std::vector<int> c((size_t)e->dev->n);
    HIP_CHECK(hipMemcpy(c.data(), e->dev->eff_n, c.size() * sizeof(int), hipMemcpyDeviceToHost));
    double sum = 0.0, mx = 0.0;
    for (int v : c) { sum += v; mx = std::max(mx, (double)v); }
    return k == "angle_records" ? sum : mx;
  }
  if (k == "special_asym") return e->dev ? (double)e->dev->flags_h[FLAG_SPECIAL_ASYM] : 0.0;   // some 1-2 list lost an entry its partner still has (sticky)
  if (k == "halo_fused") return e->dev && e->dev->fast_halo && e->dev->halo_fused ? 1.0 : 0.0;   // counters + window copy in one launch
  if (k == "halo_window_exchanges") return e->dev ? (double)e->dev->halo_seq : 0.0;   // per-step halos that went through the peer windows
  if (k == "pair_kernel_ms") return e->kstat_ms;
  if (k == "pair_kernel_launches") return (double)e->kstat_n;
  if (k == "neigh_pairs") return e->stat_neigh_pairs();
  if (k == "maxneigh") return e->dev ? (double)e->dev->maxneigh : 0.0;
  if (k == "nlocal") return e->dev ? (double)e->dev->n : 0.0;
  if (k == "nghost") return e->dev ? (double)e->dev->nghost : 0.0;
  if (k == "fene_warnings") return e->dev && e->dev->flags_h ? (double)e->dev->flags_h[FLAG_FENE_WARN] : 0.0;
  return -1.0;
}

}  // extern "C"

// test hook (not part of the reference surface): RanMarsInt stream after a jump, for the CPU unit tests
extern "C" void lammps_le_test_ranmars(int seed, long long skip, int n, double *out) {
  lmp_le::RanMarsInt r;
  r.seed(seed);
  r.jump((uint64_t)skip);
  for (int i = 0; i < n; i++) out[i] = r.uniform();
}

// ---- ranks: one process per GPU (bench.py passes the ncclUniqueId it broadcast with torch.distributed) ----
extern "C" int lammps_le_comm_unique_id(char *out128) {
  try { lmp_le::comm_unique_id(out128); return 0; } catch (const std::exception &ex) { fprintf(stderr, "%s\n", ex.what()); return 1; }
}
extern "C" void lammps_le_comm_init(void *handle, const char *backend, int rank, int world, const char *unique_id,
                                    const char *session) {
  BEGIN_CAPTURE e->comm_init(backend, rank, world, unique_id, session ? session : "default"); END_CAPTURE
}
// RCCL binding self-test on one GPU (size-1 communicator): 0 = pass
extern "C" int lammps_le_rccl_selftest() {
  try { return lmp_le::comm_rccl_selftest(); } catch (const std::exception &ex) { fprintf(stderr, "%s\n", ex.what()); return 1; }
}
// transport self-test without a GPU ("shm" backend): ring exchange + all-gather + max-reduce; returns 0 on success
extern "C" int lammps_le_comm_selftest(const char *session, int rank, int world) {
  using namespace lmp_le;
  try {
    Comm c;
    c.init("shm", rank, world, nullptr, session);
    int up = (rank + 1) % world, dn = (rank + world - 1) % world;
    double sendv[4] = {rank + 0.25, rank + 0.5, 0, 0}, recvv[4] = {0, 0, 0, 0};
    c.exchange_host({{&sendv[0], sizeof(double), dn}, {&sendv[1], sizeof(double), up}},
                    {{&recvv[1], sizeof(double), up}, {&recvv[0], sizeof(double), dn}});
    if (recvv[1] != up + 0.25 || recvv[0] != dn + 0.5) return 2;
    std::vector<int> all(world);
    int mine = 100 + rank;
    c.allgather_host(&mine, all.data(), sizeof(int));
    for (int r = 0; r < world; r++) if (all[r] != 100 + r) return 3;
    if (c.allreduce_host_max(rank * 7) != (world - 1) * 7) return 4;
    double s = c.allreduce_host_sum((double)rank);
    if (s != world * (world - 1) / 2.0) return 5;
    c.barrier();
    return 0;
  } catch (const std::exception &ex) { fprintf(stderr, "%s\n", ex.what()); return 1; }
}

// debug hook (tests only): copy one of the tag-indexed LE scratch arrays (int) or the pair-distance array (double)
extern "C" void lammps_le_debug_le_array(void *handle, int slot, int n, int *out_i, double *out_d) {
  lmp_le::Engine *e = (lmp_le::Engine *)handle;
  if (!e->dev) return;
  if (out_i) (void)hipMemcpy(out_i, e->dev->le_i[slot], (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
  if (out_d) (void)hipMemcpy(out_d, e->dev->le_d[0], (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
}
