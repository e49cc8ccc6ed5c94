// engine.cpp — run control of the MI355X engine: init / setup / Verlet loop / reneighbor decision /
// thermo.  Mirrors src/run.cpp:38-188, src/verlet.cpp:87-156 (setup) and :223-354 (run),
// src/neighbor.cpp:1933-1948 (decide), src/thermo.cpp + compute_temp/pe/pressure for the printed line.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "comm.h"
#include "device.h"

namespace lmp_le {

void dd_reneighbor(DeviceState &d, Comm &comm, double cutneighsq, const double sl[4], bool has_pair, bool build_lists = true);
void dd_halo(DeviceState &d, Comm &comm);
void dd_halo(DeviceState &d, Comm &comm, hipStream_t st, const double4 *src, double4 *dst);
void dd_halo_wait(DeviceState &d);
void dd_fast_halo_setup(DeviceState &d, Comm &comm);
void dd_fast_halo_switch(DeviceState &d);
unsigned dd_halo_mismatches(DeviceState &d);
void dd_gather_positions(DeviceState &d, Comm &comm);
void dd_gather_rows3(DeviceState &d, Comm &comm, double *table_by_tag);   // respa, decomposed: complete a by-tag [T][3] table
void dd_gather_all(DeviceState &d, Comm &comm, std::vector<double> &rows, int &stride);

static double wall() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

Engine::Engine(int argc, char **argv) {
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if ((a == "-screen" || a == "-sc") && i + 1 < argc) {
      std::string v = argv[++i];
      if (v == "none") screen = nullptr;
    } else if ((a == "-log" || a == "-l") && i + 1 < argc) {
      std::string v = argv[++i];
      if (v != "none") logfile = fopen(v.c_str(), "w");
    } else if ((a == "-echo" || a == "-e") && i + 1 < argc) {
      echo_screen = std::string(argv[++i]) != "none";
    } else if ((a == "-var" || a == "-v") && i + 2 < argc) {
      variables[argv[i + 1]] = argv[i + 2];
      i += 2;
    }
  }
  thermo_keywords = {"step", "temp", "epair", "emol", "etotal", "press"};
  const char *kt = getenv("LAMMPS_LE_KERNEL_TIMING");
  kernel_timing = kt && atoi(kt) != 0;
}

Engine::~Engine() {
  for (auto *&fl : respa_flevel) if (fl) { (void)hipFree(fl); fl = nullptr; }
  if (dev) {
    try { dev_free(*dev); } catch (...) {}
    delete dev;
  }
  if (comm) { try { comm->finalize(); } catch (...) {} delete comm; }
  if (logfile) fclose(logfile);
  for (auto &dp : dumps) if (dp.fp) fclose(dp.fp);
}

void Engine::comm_init(const std::string &backend, int rank_, int world_, const void *unique_id, const std::string &session) {
  if (dev && dev->pos) throw LammpsError("comm must be initialised before the first run");
  if (!comm) comm = new Comm();
  if (world_ > 1) device_init();
  comm->init(backend, rank_, world_, unique_id, session);
  rank = rank_; world = world_;
  if (rank != 0) screen = nullptr;
}
void Engine::halo_exchange() {
  if (world <= 1) return;
  dd_halo_wait(*dev);
  if (dev->halo_ahead) { dev->halo_ahead = false; return; }   // already exchanged behind phase 1 of the previous step
  dd_halo(*dev, *comm);
}

void Engine::halo_exchange_once() {
  if (world <= 1 || halo_step == ntimestep) return;
  halo_exchange();
  halo_step = ntimestep;
}

void Engine::say(const std::string &s) {
  if (screen) { fputs(s.c_str(), screen); fflush(screen); }
  if (logfile) { fputs(s.c_str(), logfile); fflush(logfile); }
}
void Engine::warning(const std::string &s) { say("WARNING: " + s + "\n"); }

Fix *Engine::find_fix(const std::string &id) {
  for (auto &f : fixes) if (f->id == id) return f.get();
  return nullptr;
}

// ---------------------------------------------------------------------------------------------
// init: pair coefficients (src/pair_lj_cut.cpp:512-569), neighbor cutoffs (src/neighbor.cpp:240-),
// special lists (src/special.cpp:55-), fix init
// ---------------------------------------------------------------------------------------------
void Engine::init() {
  if (!box_exist) throw LammpsError("Run command before simulation box is defined");
  for (int t = 1; t <= ntypes; t++)
    if (!mass_set[t]) throw LammpsError("Not all per-type masses are set");
  if (ntypes > MAXTYPES || nbondtypes > MAXTYPES)
    throw LammpsError("MI355X engine handles at most " + std::to_string(MAXTYPES) + " atom/bond types");
  int nt = ntypes + 1;
  lj1.assign(nt * nt, 0.0); lj2 = lj3 = lj4 = offset = cutsq = lj1;
  cutforcemax = 0.0;
  if (pair_lj) {
    for (int i = 1; i <= ntypes; i++)
      if (!pc_set[i * nt + i]) throw LammpsError("All pair coeffs are not set");
    for (int i = 1; i <= ntypes; i++)
      for (int j = i; j <= ntypes; j++) {
        int ij = i * nt + j, ji = j * nt + i, ii = i * nt + i, jj = j * nt + j;
        double eps, sig, cut;
        if (pc_set[ij]) { eps = pc_eps[ij]; sig = pc_sig[ij]; cut = pc_cut[ij]; }
        else {   // src/pair.cpp mix_energy / mix_distance
          eps = std::sqrt(pc_eps[ii] * pc_eps[jj]);
          sig = pair_mix ? 0.5 * (pc_sig[ii] + pc_sig[jj]) : std::sqrt(pc_sig[ii] * pc_sig[jj]);
          cut = pair_mix ? 0.5 * (pc_cut[ii] + pc_cut[jj]) : std::sqrt(pc_cut[ii] * pc_cut[jj]);
        }
        lj1[ij] = 48.0 * eps * std::pow(sig, 12.0);
        lj2[ij] = 24.0 * eps * std::pow(sig, 6.0);
        lj3[ij] = 4.0 * eps * std::pow(sig, 12.0);
        lj4[ij] = 4.0 * eps * std::pow(sig, 6.0);
        if (pair_shift && cut > 0.0) {
          double ratio = sig / cut;
          offset[ij] = 4.0 * eps * (std::pow(ratio, 12.0) - std::pow(ratio, 6.0));
        } else offset[ij] = 0.0;
        cutsq[ij] = cut * cut;
        lj1[ji] = lj1[ij]; lj2[ji] = lj2[ij]; lj3[ji] = lj3[ij]; lj4[ji] = lj4[ij];
        offset[ji] = offset[ij]; cutsq[ji] = cutsq[ij];
        cutforcemax = std::max(cutforcemax, cut);
      }
  } else if (pair_zero) cutforcemax = pair_cut_global;
  cutneighmax = (pair_lj || pair_zero) ? cutforcemax + skin : 0.0;
  if (atom_style != "atomic" && !bond_style_name.empty() && bond_style_name != "zero" && bond_style_name != "none")
    for (int b = 1; b <= nbondtypes; b++)
      if (bondtab.style[b] == 0 && nbonds > 0) throw LammpsError("All bond coeffs are not set");
  if (!special_built) build_special();
  for (auto &f : fixes) f->init();
}

// ---------------------------------------------------------------------------------------------
void Engine::device_init() {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    throw LammpsError("No HIP device: the MI355X engine has no CPU fallback (hipGetDeviceCount: " +
                      std::string(hipGetErrorString(e)) + ")");
  const char *lr = getenv("LOCAL_RANK");
  int devid = lr ? atoi(lr) % ndev : 0;
  HIP_CHECK(hipSetDevice(devid));
  if (!dev) dev = new DeviceState();
}

void Engine::upload() {
  device_init();
  if (natoms <= 0) throw LammpsError("No atoms to run");
  for (int d = 0; d < 3; d++) {
    if (cutneighmax > 0.0 && box.prd[d] < 3.0 * cutneighmax)
      throw LammpsError("Box is smaller than 3 neighbor cutoffs in a periodic dimension: "
                        "not supported by the MI355X engine (minimum-image cell lists)");
  }
  DeviceState &d = *dev;
  d.bins_ready = false;      // the device arrays are about to be replaced (tag order): bins of the old arrays are void
  d.bond_pack_p_valid = false;
  bool realloc = (d.ntotal != natoms || d.bpa != bpa || d.maxspecial != maxspecial || d.ntypes != ntypes || !d.pos || d.apa != apa);
  double cellcut = cutneighmax > 0.0 ? cutneighmax : std::max({box.prd[0], box.prd[1], box.prd[2]}) / 3.0;
  if (world > 1) {
    // z-slab decomposition: rank r owns z in [lo + r*w, lo + (r+1)*w); ghost shell = max(neighbor cutoff, comm cutoff)
    d.dd = 1;
    d.ghost_whole_shell = angles_active();
    double w = box.prd[2] / world;
    d.slab_lo = box.lo[2] + rank * w;
    d.slab_hi = d.slab_lo + w;
    d.cutghost = std::max(cutneighmax, comm_cutoff);
    d.zlo_ext = d.slab_lo - d.cutghost;
    if (w < 2.0 * d.cutghost)
      throw LammpsError("slab thinner than two ghost cutoffs: use fewer GPUs or a smaller comm_modify cutoff");
    if (w + 2.0 * d.cutghost > box.prd[2]) throw LammpsError("ghost shells of a slab overlap: box too small for this many GPUs");
  }
  if (!realloc)
    for (int k = 0; k < 3; k++) {
      double extent = (k == 2 && d.dd) ? (d.slab_hi - d.slab_lo) + 2.0 * d.cutghost : box.prd[k];
      if ((int)(extent / (k == 0 ? cellcut / CELL_XSPLIT : cellcut)) != d.ncell[k]) realloc = true;
    }
  if (realloc) {
    if (d.pos) dev_free(d);
    d.apa = apa;
    dev_alloc(d, natoms, natoms, ntypes, bpa, maxspecial, box, cellcut);
    if (comm) comm->main_stream = d.stream;
    if (d.dd) { dd_alloc(d, world); dd_fast_halo_setup(d, *comm); }
    for (auto &f : fixes)
      if (auto *l = dynamic_cast<FixLangevin *>(f.get())) l->dev_ready = false;
  }
  d.box = box;
  for (int k = 1; k <= 3; k++) d.sflag[k] = special_flag(k);
  if (comm) comm->main_stream = d.stream;
  d.comm_watch = (comm && world > 1) ? comm : nullptr;
  int n = natoms, np = d.npad;
  size_t nt = (size_t)n + 2;
  std::vector<double4> pos(np);
  std::vector<double> vv(3 * (size_t)np, 0.0), ff(3 * (size_t)np, 0.0);
  std::vector<int> tg(np, 0), im(3 * (size_t)np, 0), mp(nt, -1), ty(nt, 0), cr(nt, 0);
  int nloc = 0;
  for (int i = 0; i < n; i++) {
    ty[i + 1] = type[i]; cr[i + 1] = crank[i];
    if (d.dd) {   // same owner expression as k_dd_classify
      double zz = x[3 * i + 2];
      if (zz < box.lo[2]) zz += box.prd[2];
      if (zz >= box.hi[2]) zz -= box.prd[2];
      int owner = (int)((zz - box.lo[2]) / (box.prd[2] / world));
      owner = std::min(std::max(owner, 0), world - 1);
      if (owner != rank) continue;
    }
    int q = nloc++;
    pos[q] = make_double4(x[3 * i], x[3 * i + 1], x[3 * i + 2], (double)type[i]);
    for (int k = 0; k < 3; k++) {
      vv[(size_t)k * np + q] = v[3 * i + k];
      ff[(size_t)k * np + q] = f[3 * i + k];
      im[(size_t)k * np + q] = image[3 * i + k];
    }
    tg[q] = i + 1; mp[i + 1] = q;
  }
  d.n = nloc;
  d.nghost = 0;
  auto up = [&](void *dst, const void *src, size_t bytes) {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d.stream));
  };
  up(d.pos, pos.data(), np * sizeof(double4));
  up(d.xhold, pos.data(), np * sizeof(double4));
  for (int k = 0; k < 3; k++) {
    up(d.v[k], vv.data() + (size_t)k * np, np * sizeof(double));
    up(d.f[k], ff.data() + (size_t)k * np, np * sizeof(double));
  }
  up(d.tag, tg.data(), np * sizeof(int));
  up(d.img, im.data(), 3 * (size_t)np * sizeof(int));
  up(d.map, mp.data(), nt * sizeof(int));
  up(d.type_t, ty.data(), nt * sizeof(int));
  up(d.crank, cr.data(), nt * sizeof(int));
  // fixes on groups: masks by tag, and the rank of every bead among the members of fix langevin's group in LOCAL order (the
  // order its draws are handed out in)
  const int members_before = langevin_members;
  langevin_members = natoms;
  std::vector<int> gm, lr;
  {
    int lgbit = 1;
    bool grouped = false;
    for (auto &f : fixes) {
      if (f->groupbit != 1) grouped = true;
      if (dynamic_cast<FixLangevin *>(f.get())) lgbit = f->groupbit;
    }
    if (grouped) {
      if ((int)gmask.size() != natoms) throw LammpsError("internal: group masks do not match the atom count");
      gm.assign(nt, 0); lr.assign(nt, 0);
      for (int i = 0; i < n; i++) gm[i + 1] = gmask[i];
      if (lgbit != 1) {
        std::vector<int> order(n);
        for (int i = 0; i < n; i++) order[crank[i]] = i;          // local index -> atom
        int m = 0;
        for (int k = 0; k < n; k++) { const int i = order[k]; if (gmask[i] & lgbit) lr[i + 1] = m++; }
        langevin_members = m;
      }
      if (!d.gmask) { HIP_CHECK(hipMalloc((void **)&d.gmask, nt * sizeof(int))); HIP_CHECK(hipMalloc((void **)&d.lgrank, nt * sizeof(int))); }
      up(d.gmask, gm.data(), nt * sizeof(int));
      up(d.lgrank, lr.data(), nt * sizeof(int));
    }
    d.lg_grouped = grouped && lgbit != 1;
    d.lg_bit = lgbit;
  }
  if (langevin_members != members_before)       // the stream is cut into calls of 3 * members draws
    for (auto &f : fixes)
      if (auto *l = dynamic_cast<FixLangevin *>(f.get())) l->dev_ready = false;
  // topology: host index t-1 -> device index t
  std::vector<int> nb(nt, 0), bt(nt * bpa, 0), ba(nt * bpa, 0), ns(nt * 3, 0), sp(nt * (size_t)maxspecial, 0);
  std::copy(num_bond.begin(), num_bond.end(), nb.begin() + 1);
  std::copy(bond_type.begin(), bond_type.end(), bt.begin() + bpa);
  std::copy(bond_atom.begin(), bond_atom.end(), ba.begin() + bpa);
  std::copy(nspecial.begin(), nspecial.end(), ns.begin() + 3);
  std::copy(special.begin(), special.end(), sp.begin() + maxspecial);
  up(d.num_bond, nb.data(), nt * sizeof(int));
  up(d.bond_type, bt.data(), nt * bpa * sizeof(int));
  up(d.bond_atom, ba.data(), nt * bpa * sizeof(int));
  up(d.nspecial, ns.data(), nt * 3 * sizeof(int));
  up(d.special, sp.data(), nt * (size_t)maxspecial * sizeof(int));
  std::vector<int> an, ag[4];
  if (apa > 0) {
    an.assign(nt, 0);
    std::copy(num_angle.begin(), num_angle.end(), an.begin() + 1);
    up(d.num_angle, an.data(), nt * sizeof(int));
    const std::vector<int> *src[4] = {&angle_type, &angle_a1, &angle_a2, &angle_a3};
    int *dst[4] = {d.angle_type, d.angle_a1, d.angle_a2, d.angle_a3};
    for (int q = 0; q < 4; q++) {
      ag[q].assign(nt * apa, 0);
      std::copy(src[q]->begin(), src[q]->end(), ag[q].begin() + apa);
      up(dst[q], ag[q].data(), nt * apa * sizeof(int));
    }
  }
  int ntp = ntypes + 1;
  std::vector<double> tab(6 * (size_t)ntp * ntp, 0.0);
  if (pair_lj)
    for (int k = 0; k < ntp * ntp; k++) {
      tab[k] = cutsq[k]; tab[ntp * ntp + k] = lj1[k]; tab[2 * ntp * ntp + k] = lj2[k];
      tab[3 * ntp * ntp + k] = lj3[k]; tab[4 * ntp * ntp + k] = lj4[k]; tab[5 * ntp * ntp + k] = offset[k];
    }
  up(d.pairtab, tab.data(), tab.size() * sizeof(double));
  d.pair_uniform = pair_lj ? 1 : 0;
  d.cutneigh = cutneighmax + skin;   // interior test margin: listed at build (within cutneigh) + drift since then
  if (pair_lj) {
    int ref = 1 * ntp + 1;
    for (int i = 1; i <= ntypes; i++)
      for (int j = 1; j <= ntypes; j++)
        for (int q = 0; q < 6; q++)
          if (tab[(size_t)q * ntp * ntp + i * ntp + j] != tab[(size_t)q * ntp * ntp + ref]) d.pair_uniform = 0;
    for (int q = 0; q < 6; q++) d.pair_u[q] = tab[(size_t)q * ntp * ntp + ref];
  }
  HIP_CHECK(hipMemsetAsync(d.flags, 0, NFLAGS * sizeof(int), d.stream));
  {
    // special lists that disagree between the two ends of a pair (possible after an ex_unload acted on a stale
    // bond-list entry, kernels_le.hip dev_special_remove12) switch the list build to half-list semantics
    auto level = [&](int i, int t) {
      const int *sl = &special[(size_t)i * maxspecial];
      int n1 = nspecial[3 * (size_t)i], n2 = nspecial[3 * (size_t)i + 1], n3 = nspecial[3 * (size_t)i + 2];
      for (int k = 0; k < n3; k++)
        if (sl[k] == t) { int l = k < n1 ? 1 : k < n2 ? 2 : 3; return special_flag(l) == 1 ? 0 : l; }
      return 0;     // a level with weight 1 is the same as not special
    };
    int asym = 0;
    for (int i = 0; i < natoms && !asym && maxspecial > 0; i++) {
      int n1 = nspecial[3 * (size_t)i], n2 = nspecial[3 * (size_t)i + 1], n3 = nspecial[3 * (size_t)i + 2];
      for (int k = 0; k < n3; k++) {
        int j = special[(size_t)i * maxspecial + k] - 1;
        int l = k < n1 ? 1 : k < n2 ? 2 : 3;
        if (special_flag(l) == 1) continue;
        if (j < 0 || j >= natoms || level(j, i + 1) != l) { asym = 1; break; }
      }
    }
    if (asym) {
      HIP_CHECK(hipMemcpyAsync(d.flags + FLAG_SPECIAL_ASYM, &asym, sizeof(int), hipMemcpyHostToDevice, d.stream));
      d.flags_h[FLAG_SPECIAL_ASYM] = 1;
    }
  }
  HIP_CHECK(hipStreamSynchronize(d.stream));
  dev_current = true;
  host_current = true;
}

void Engine::download() {
  if (!dev || !dev_current || host_current) return;
  DeviceState &d = *dev;
  if (d.dd) {
    // collective: every rank ends up with the whole system (rows: tag x y z type vx vy vz fx fy fz ix iy iz)
    std::vector<double> rows;
    int stride = 0;
    dd_gather_all(d, *comm, rows, stride);
    for (size_t r = 0; r < rows.size() / 14; r++) {
      const double *b = &rows[r * 14];
      int t = (int)b[0];
      if (t <= 0) continue;
      int i = t - 1;
      for (int k = 0; k < 3; k++) { x[3 * i + k] = b[1 + k]; v[3 * i + k] = b[5 + k]; f[3 * i + k] = b[8 + k]; image[3 * i + k] = (int)b[11 + k]; }
    }
    size_t nt = (size_t)natoms + 2;
    std::vector<int> ty(nt), nb(nt), bt(nt * bpa), ba(nt * bpa), ns(nt * 3), sp(nt * (size_t)maxspecial);
    HIP_CHECK(hipMemcpy(ty.data(), d.type_t, nt * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(nb.data(), d.num_bond, nt * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(bt.data(), d.bond_type, nt * bpa * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(ba.data(), d.bond_atom, nt * bpa * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(ns.data(), d.nspecial, nt * 3 * sizeof(int), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(sp.data(), d.special, nt * (size_t)maxspecial * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < natoms; i++) type[i] = ty[i + 1];
    std::copy(nb.begin() + 1, nb.begin() + 1 + natoms, num_bond.begin());
    std::copy(bt.begin() + bpa, bt.begin() + bpa + (size_t)natoms * bpa, bond_type.begin());
    std::copy(ba.begin() + bpa, ba.begin() + bpa + (size_t)natoms * bpa, bond_atom.begin());
    std::copy(ns.begin() + 3, ns.begin() + 3 + (size_t)natoms * 3, nspecial.begin());
    std::copy(sp.begin() + maxspecial, sp.begin() + maxspecial + (size_t)natoms * maxspecial, special.begin());
    if (apa > 0) {       // the angle tables are replicated by tag, like the bond tables
      std::vector<int> an(nt), ag(nt * apa);
      HIP_CHECK(hipMemcpy(an.data(), d.num_angle, nt * sizeof(int), hipMemcpyDeviceToHost));
      std::copy(an.begin() + 1, an.begin() + 1 + natoms, num_angle.begin());
      const int *src[4] = {d.angle_type, d.angle_a1, d.angle_a2, d.angle_a3};
      std::vector<int> *dst[4] = {&angle_type, &angle_a1, &angle_a2, &angle_a3};
      for (int q = 0; q < 4; q++) {
        HIP_CHECK(hipMemcpy(ag.data(), src[q], nt * apa * sizeof(int), hipMemcpyDeviceToHost));
        std::copy(ag.begin() + apa, ag.begin() + apa + (size_t)natoms * apa, dst[q]->begin());
      }
    }
    host_current = true;
    return;
  }
  int n = natoms, np = d.npad;
  size_t nt = (size_t)n + 2;
  std::vector<double4> pos(np);
  std::vector<double> vv(3 * (size_t)np), ff(3 * (size_t)np);
  std::vector<int> tg(np), im(3 * (size_t)np), ty(nt);
  auto down = [&](void *dst, const void *src, size_t bytes) {
    HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, d.stream));
  };
  down(pos.data(), d.pos, np * sizeof(double4));
  for (int k = 0; k < 3; k++) {
    down(vv.data() + (size_t)k * np, d.v[k], np * sizeof(double));
    down(ff.data() + (size_t)k * np, d.f[k], np * sizeof(double));
  }
  down(tg.data(), d.tag, np * sizeof(int));
  down(im.data(), d.img, 3 * (size_t)np * sizeof(int));
  down(ty.data(), d.type_t, nt * sizeof(int));
  std::vector<int> nb(nt), bt(nt * bpa), ba(nt * bpa), ns(nt * 3), sp(nt * (size_t)maxspecial);
  down(nb.data(), d.num_bond, nt * sizeof(int));
  down(bt.data(), d.bond_type, nt * bpa * sizeof(int));
  down(ba.data(), d.bond_atom, nt * bpa * sizeof(int));
  down(ns.data(), d.nspecial, nt * 3 * sizeof(int));
  down(sp.data(), d.special, nt * (size_t)maxspecial * sizeof(int));
  std::vector<int> an, ag[4];
  if (apa > 0) {
    an.resize(nt);
    down(an.data(), d.num_angle, nt * sizeof(int));
    const int *src[4] = {d.angle_type, d.angle_a1, d.angle_a2, d.angle_a3};
    for (int q = 0; q < 4; q++) { ag[q].resize(nt * apa); down(ag[q].data(), src[q], nt * apa * sizeof(int)); }
  }
  HIP_CHECK(hipStreamSynchronize(d.stream));
  if (apa > 0) {
    std::copy(an.begin() + 1, an.begin() + 1 + n, num_angle.begin());
    std::vector<int> *dst[4] = {&angle_type, &angle_a1, &angle_a2, &angle_a3};
    for (int q = 0; q < 4; q++) std::copy(ag[q].begin() + apa, ag[q].begin() + apa + (size_t)n * apa, dst[q]->begin());
  }
  for (int p = 0; p < n; p++) {
    int i = tg[p] - 1;
    x[3 * i] = pos[p].x; x[3 * i + 1] = pos[p].y; x[3 * i + 2] = pos[p].z;
    for (int k = 0; k < 3; k++) {
      v[3 * i + k] = vv[(size_t)k * np + p];
      f[3 * i + k] = ff[(size_t)k * np + p];
      image[3 * i + k] = im[(size_t)k * np + p];
    }
  }
  for (int i = 0; i < n; i++) type[i] = ty[i + 1];
  if (crank_on_device) {
    std::vector<int> cr(nt);
    HIP_CHECK(hipMemcpy(cr.data(), d.crank, nt * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) crank[i] = cr[i + 1];
    crank_on_device = false;
  }
  std::copy(nb.begin() + 1, nb.begin() + 1 + n, num_bond.begin());
  std::copy(bt.begin() + bpa, bt.begin() + bpa + (size_t)n * bpa, bond_type.begin());
  std::copy(ba.begin() + bpa, ba.begin() + bpa + (size_t)n * bpa, bond_atom.begin());
  std::copy(ns.begin() + 3, ns.begin() + 3 + (size_t)n * 3, nspecial.begin());
  std::copy(sp.begin() + maxspecial, sp.begin() + maxspecial + (size_t)n * maxspecial, special.begin());
  host_current = true;
}

// ---------------------------------------------------------------------------------------------
static void check_device_error(Engine *e, DeviceState &d) {
  int code = d.flags_h[FLAG_ERROR];
  if (!code) return;
  const char *msg = "device error";
  switch (code) {
    case ERR_BAD_FENE: msg = "Bad FENE bond"; break;
    case ERR_BOND_MISSING: msg = "Bond atoms missing"; break;
    case ERR_EXT_MULTI: msg = "Fix extrusion, more than one bond type 2"; break;
    case ERR_BPA: msg = "New bond exceeded bonds per atom in fix ex_load"; break;
    case ERR_SPECIAL: msg = "New bond exceeded special list size in fix ex_load"; break;
    case ERR_COUNT_MISMATCH: msg = "Numbers of created and broken bonds are not equal"; break;
    case ERR_NONFINITE: msg = "Non-numeric atom coords - simulation unstable"; break;
    case ERR_HALO_TIMEOUT: msg = "a neighbouring rank did not deliver its halo in time (peer window exchange)"; break;
    case ERR_SPECIAL_SCRATCH: msg = "Special list size exceeded in fix bond/create"; break;
    case ERR_ANGLES: msg = "Fix ex_load induced too many angles/dihedrals/impropers per atom"; break;
    case ERR_GHOST_ORDER: msg = "internal: ghost blocks of a slab interleave (slab thinner than two ghost shells?)"; break;
  }
  (void)e;
  throw LammpsError(msg);
}

// defer_check: the flags of the build are published but not waited for - the caller enqueues the step kernel first
// (it leaves the state untouched if a list overflowed) and then calls finish_reneighbor(); the host round trip of
// the check (~25 us) is hidden behind that kernel instead of idling the GPU once per rebuild.
void Engine::reneighbor(bool defer_check, bool sort_now) {
  DeviceState &d = *dev;
  // (read at every rebuild, not cached: the test that sets it shares its process with tests that must not see it)
  const char *ovf = getenv("LAMMPS_LE_TEST_OVERFLOW_AT");
  const long test_overflow_at = ovf ? atol(ovf) : -1;
  if (test_overflow_at >= 0 && neigh_builds == test_overflow_at) dev_alloc_neigh(d, 4);   // test hook: force an overflow
  // FLAG_MOVED / NEIGH_OVERFLOW / MAXNEIGH are zero here: they are reset by the publish kernel that reports them
  // pbc + ownership + cell order, then - on a sort step - the reference's Atom::sort (src/verlet.cpp:270-286: after pbc,
  // BEFORE neighbor->build: the pair list of this very build is stored in the new local order, which decides whose special
  // list a pair's status comes from when the lists are asymmetric), then the lists
  if (d.dd) dd_reneighbor(d, *comm, cutneighmax * cutneighmax, special_lj, pair_lj, false);
  else launch_sort_owned(d);
  if (sort_now) emulate_atom_sort();
  launch_lists(d, cutneighmax * cutneighmax, special_lj, pair_lj);
  // (decomposed: the border pass of the rebuild has also checked that this rank's Langevin pools hold the draws of the beads
  //  it owns now - FLAG_RNG_MISS, kernels_dd.hip k_dd_borders)
  if (defer_check && !d.dd) {
    publish_flags(d, 1u << FLAG_MOVED);          // NEIGH_OVERFLOW stays set on the device: the step kernel reads it
    reneigh_pending = true;
  } else {
    sync_flags(d, 1u << FLAG_MOVED);
    check_device_error(this, d);
    if (d.flags_h[FLAG_RNG_MISS]) { rng_late_generate(d); rng_late_count++; }
    if (d.flags_h[FLAG_NEIGH_OVERFLOW]) regrow_lists();
  }
  if (angles_active()) launch_angle_list(d);     // NTopoAngle::build
  if (d.le_snapshot && d.topo_dirty) {   // NTopoBond::build: the bond list the LE fixes will see until the next reneighbor
    d.topo_dirty = false;
    launch_topo_snapshot(d);
  }
  ago = 0;
  neigh_builds++;
}

// Timer::stamp (src/timer.cpp:100-135): wall clock between stamps goes to a section; `timer sync` drains the device first
void Engine::stamp() {
  if (timer_level < 2) return;
  if (timer_sync && dev && dev->stream) stream_sync(*dev);     // (decomposed runs: a bounded wait, Comm::wait_stream)
  timer_prev = wall();
}
void Engine::stamp(int which) {
  if (timer_level < 2) return;
  if (timer_sync && dev && dev->stream) stream_sync(*dev);
  double now = wall();
  timers[which] += now - timer_prev;
  timer_prev = now;
}

bool Engine::finish_reneighbor() {
  if (!reneigh_pending) return true;
  reneigh_pending = false;
  wait_flags(*dev);
  check_device_error(this, *dev);
  return !dev->flags_h[FLAG_NEIGH_OVERFLOW];
}
// grow the ELL table and rebuild (atoms are already wrapped and sorted: the rebuild is idempotent)
void Engine::regrow_lists() {
  DeviceState &d = *dev;
  for (int attempt = 0; attempt < 6 && d.flags_h[FLAG_NEIGH_OVERFLOW]; attempt++) {
    dev_alloc_neigh(d, d.flags_h[FLAG_MAXNEIGH] + 16);
    HIP_CHECK(hipMemsetAsync(d.flags + FLAG_NEIGH_OVERFLOW, 0, sizeof(int), d.stream));
    HIP_CHECK(hipMemsetAsync(d.flags + FLAG_MAXNEIGH, 0, sizeof(int), d.stream));
    launch_lists(d, cutneighmax * cutneighmax, special_lj, pair_lj);
    sync_flags(d);
    check_device_error(this, d);
  }
  if (d.flags_h[FLAG_NEIGH_OVERFLOW]) throw LammpsError("Neighbor list overflow, boost neigh_modify one");   // src/npair_*.cpp
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_MAXNEIGH, 0, sizeof(int), d.stream));
}

// Neighbor::decide (src/neighbor.cpp:1933-1948); `moved` was produced by the initial_integrate kernel
bool Engine::decide() {
  for (size_t k = 0; k < fixes.size(); k++)
    if (fixes[k]->force_reneighbor && le_reneigh_step[k] == ntimestep) return true;
  ago++;
  if (ago >= neigh_delay && ago % neigh_every == 0) {
    if (!neigh_check) return true;
    if (world > 1) { dd_halo_wait(*dev); comm->allreduce_int_max(dev->stream, dev->flags, 2); }   // FLAG_MOVED, FLAG_ERROR (neighbor.cpp:2011)
    sync_flags(*dev);
    check_device_error(this, *dev);
    bool moved = dev->flags_h[FLAG_MOVED] != 0;
    if (moved && ago == std::max(neigh_every, neigh_delay)) neigh_dangerous++;
    return moved;
  }
  return false;
}

// keep `crank` equal to the reference's local atom order: Atom::sort (src/atom.cpp:2003-2094) with
// bins of 1/2 cutneighmax over the box (setup_sort_bins :2100-2208), stable within a bin
bool Engine::local_order_is_tag_order() const {
  if (sortfreq > 0 || crank_on_device) return false;
  for (int i = 0; i < natoms; i++) if (crank[i] != i) return false;
  return true;
}
void Engine::emulate_atom_sort() {
  nextsort = (ntimestep / sortfreq) * sortfreq + sortfreq;
  double binsize = 0.5 * cutneighmax;
  if (binsize == 0.0) return;
  double bininv = 1.0 / binsize;
  int nb[3]; double binv[3];
  for (int d = 0; d < 3; d++) {
    nb[d] = (int)((box.hi[d] - box.lo[d]) * bininv);
    if (nb[d] == 0) nb[d] = 1;
    binv[d] = nb[d] / (box.hi[d] - box.lo[d]);
  }
  if ((long)nb[0] * nb[1] * nb[2] == 1) return;
  // on the device (kernels_sort.hip): keys (bin, previous rank), radix sort, crank[tag] = new rank.  The host copy of
  // `crank` is refreshed by download().
  if (world > 1) dd_gather_positions(*dev, *comm);      // decomposed: the one-rank order from everybody's wrapped positions
  launch_atom_sort(*dev, nb, binv, world > 1);
  crank_on_device = true;
  if (world > 1) {        // the canonical ranks changed: so did the stream segments this rank draws from
    rng_validate_owned(*dev);
    sync_flags(*dev);
    if (dev->flags_h[FLAG_RNG_MISS]) { rng_late_generate(*dev); rng_late_count++; }
  }
}

// ---------------------------------------------------------------------------------------------
static TypeTables make_tables(Engine *e, FixLangevin *lg) {
  TypeTables tt{};
  double dtf = 0.5 * e->dt * e->ftm2v;
  for (int t = 1; t <= e->ntypes; t++) {
    tt.mass[t] = e->mass[t];
    tt.dtfm[t] = dtf / e->mass[t];
    if (lg) {
      // src/fix_langevin.cpp:296-310 (init) and :784-797 (compute_target), :662-663
      double g1 = -e->mass[t] / lg->t_period / e->ftm2v;
      double g2 = std::sqrt(e->mass[t]) * std::sqrt(24.0 * e->boltz / lg->t_period / e->dt / e->mvv2e) / e->ftm2v;
      g1 *= 1.0 / lg->ratio[t];
      g2 *= 1.0 / std::sqrt(lg->ratio[t]);
      double delta = (double)(e->ntimestep - e->beginstep);
      if (delta != 0.0) delta /= (double)(e->endstep - e->beginstep);
      double t_target = lg->t_start + delta * (lg->t_stop - lg->t_start);
      double tsqrt = std::sqrt(t_target);
      tt.g1[t] = g1;
      tt.g2[t] = g2 * tsqrt;
    }
  }
  return tt;
}

static FixLangevin *the_langevin(Engine *e) {
  FixLangevin *lg = nullptr;
  for (auto &f : e->fixes)
    if (auto *l = dynamic_cast<FixLangevin *>(f.get())) {
      if (lg) throw LammpsError("MI355X engine supports one fix langevin");
      lg = l;
    }
  return lg;
}
static int count_nve(Engine *e) {
  int c = 0;
  for (auto &f : e->fixes) if (dynamic_cast<FixNVE *>(f.get())) c++;
  return c;
}
static std::vector<int> nve_bits(Engine *e) {      // the group of every fix nve, in fix order
  std::vector<int> b;
  for (auto &f : e->fixes) if (dynamic_cast<FixNVE *>(f.get())) b.push_back(f->groupbit);
  return b;
}
// what the uploaded masks depend on beyond the atoms themselves: which fixes act on which group
std::string Engine::group_signature() const {
  std::string sig;
  for (auto &f : fixes) if (f->groupbit != 1) sig += f->id + ":" + f->style + ":" + std::to_string(f->groupbit) + ";";
  return sig;
}
static bool fixes_on_groups(Engine *e) {
  for (auto &f : e->fixes) if (f->groupbit != 1) return true;
  return false;
}
static bool md_fixes_on_groups(Engine *e) {      // fix nve / fix langevin: the ones the fused step kernel stands for
  for (auto &f : e->fixes)
    if (f->groupbit != 1 && (dynamic_cast<FixNVE *>(f.get()) || dynamic_cast<FixLangevin *>(f.get()))) return true;
  return false;
}

static bool timed_begin(Engine *e) {
  DeviceState &d = *e->dev;
  // sample every ktime_every-th launch of this run (uniform over the run; every launch of a short run)
  long c = e->ktime_counter++;
  if (!e->kernel_timing || d.ev_used >= 4096 || (c % e->ktime_every) != 0) return false;
  if (d.ev0.size() <= d.ev_used) {
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b));
    d.ev0.push_back(a); d.ev1.push_back(b);
  }
  return true;
}

bool Engine::angles_active() const {
  if (apa <= 0 || angle_style_name.empty() || angle_style_name == "none" || angle_style_name == "zero") return false;
  for (int a = 1; a <= nangletypes; a++) if (angtab.style[a]) return true;
  return false;
}
void Engine::compute_forces(bool eflag) {
  launch_force(*dev, bondtab, special_lj, eflag, pair_lj);
  if (angles_active()) launch_angle(*dev, angtab, eflag);     // Angle::compute follows Bond::compute (src/verlet.cpp:298-302)
}

double Engine::stat_neigh_pairs() {
  if (!dev || !dev->numneigh) return 0.0;
  std::vector<int> nn(dev->n + 1);
  HIP_CHECK(hipMemcpy(nn.data(), dev->numneigh, (size_t)dev->n * sizeof(int), hipMemcpyDeviceToHost));
  nn[dev->n] = 0;
  double s = 0.0;
  for (int v : nn) s += (v & NN_COUNT_MASK) - ((v >> NN_BOND_SHIFT) & NN_NBOND_MASK);     // pair entries (a bead's bonds open its list)
  if (world > 1) s = comm->allreduce_host_sum(s);
  return s;
}

// advance the Langevin stream by one post_force call: 3N draws into rng_out (canonical order)
static void langevin_draws(Engine *e, FixLangevin *lg) {
  DeviceState &d = *e->dev;
  if (e->langevin_members == 0) return;      // (a group without members: nobody draws)
  if (!lg->dev_ready) {
    // position a host generator at the first draw of this call and cut the stream into blocks
    RanMarsInt r;
    r.seed(lg->seed);
    r.jump(lg->draws);
    rng_langevin_setup(d, r, e->langevin_members);
    lg->dev_ready = true;
  }
  launch_rng_langevin(d, lg->draws + 1);     // raw index of draw k is k+1 (constructor warm-up)
  lg->draws += 3ull * e->langevin_members;   // (three draws per MEMBER of the fix's group and call, src/fix_langevin.cpp:660-674)
}
static void langevin_post_force(Engine *e, FixLangevin *lg, bool fuse_final) {
  langevin_draws(e, lg);
  TypeTables tt = make_tables(e, lg);
  if (e->langevin_members == 0) {
    if (lg->zeroflag) throw LammpsError("Cannot zero Langevin force of 0 atoms");      // src/fix_langevin.cpp:633-634
    return;
  }
  launch_langevin(*e->dev, tt, e->dev->ident_order, fuse_final && !lg->zeroflag, lg->groupbit);
  // `zero yes`: the members' mean random force comes off every member (:752-772); block sums in block order, one total
  if (lg->zeroflag && e->world > 1) {        // decomposed: the ranks' sums are added on the host (MPI_Allreduce of fsum, :755)
    double s3[3];
    launch_langevin_zero(*e->dev, tt, e->dev->ident_order, lg->groupbit, e->langevin_members, s3);
    e->comm->allreduce_host_sum(s3, 3);
    for (int k = 0; k < 3; k++) s3[k] /= (double)e->langevin_members;
    launch_langevin_zero_apply(*e->dev, lg->groupbit, s3);
  } else if (lg->zeroflag) launch_langevin_zero(*e->dev, tt, e->dev->ident_order, lg->groupbit, e->langevin_members);
  rng_langevin_consumed(*e->dev);
}

ThermoRow Engine::eval_thermo(bool ke_summed) {
  DeviceState &d = *dev;
  TypeTables tt = make_tables(this, nullptr);
  if (!ke_summed) launch_ke(d, tt);       // (the energy variant of the step kernel has summed the kinetic energy as well)
  double s[16];
  reduce_partials(d, s);
  if (world > 1) comm->allreduce_host_sum(s, 16);
  ThermoRow r{};
  r.step = ntimestep;
  double dof = 3.0 * natoms - 3.0;                       // src/compute_temp.cpp:60-68
  double tfactor = dof > 0 ? mvv2e / (dof * boltz) : 0.0;
  r.temp = s[14] * tfactor;
  double norm = thermo_norm ? (double)natoms : 1.0;
  double ke = r.temp * 0.5 * dof * boltz;
  r.evdwl = s[0]; r.ebond = s[1];
  for (int k = 0; k < 6; k++) r.virial[k] = s[2 + k] + s[8 + k];
  if (angles_active()) {       // thermo's emol = ebond + eangle (src/thermo.cpp), the pressure's virial includes the angles'
    double a8[8];
    reduce_angle_partials(d, a8);
    if (world > 1) comm->allreduce_host_sum(a8, 8);
    r.eangle = a8[0];
    for (int k = 0; k < 6; k++) r.virial[k] += a8[1 + k];
  }
  double vol = box.prd[0] * box.prd[1] * box.prd[2];
  r.press = (dof * boltz * r.temp + r.virial[0] + r.virial[1] + r.virial[2]) / 3.0 / vol * nktv2p;
  const double emol = r.ebond + r.eangle;
  r.epair = r.evdwl / norm; r.emol = emol / norm;
  r.pe = (r.evdwl + emol) / norm;
  r.ke = ke / norm;
  r.etotal = (ke + r.evdwl + emol) / norm;
  r.nbonds = nbonds;
  if (want_ptensor) {          // (ke_tensor[i] + virial[i]) / volume * nktv2p, src/compute_pressure.cpp:244-290
    double k6[6];
    ke_tensor(d, tt, k6);
    if (world > 1) comm->allreduce_host_sum(k6, 6);
    for (int k = 0; k < 6; k++) r.ptensor[k] = (k6[k] * mvv2e + r.virial[k]) / vol * nktv2p;
    r.has_ptensor = true;
  }
  sync_flags(d);
  check_device_error(this, d);
  return r;
}

bool Engine::thermo_keyword(const ThermoRow &r, const std::string &k, double &val, bool &isint) {
  const double norm = thermo_norm ? (double)natoms : 1.0;
  const double vol = box.prd[0] * box.prd[1] * box.prd[2];
  isint = false;
  if (k == "step") { val = (double)r.step; isint = true; }
  else if (k == "elapsed" || k == "elaplong") { val = (double)(r.step - beginstep); isint = true; }   // firststep = beginstep: no `run start`
  else if (k == "dt") val = dt;
  else if (k == "time") val = atime + (double)(r.step - atimestep) * dt;            // src/thermo.cpp:1600-1603
  else if (k == "cpu") val = thermo_first_line ? 0.0 : wall() - run_wall0;          // :1607-1611
  else if (k == "atoms") { val = (double)natoms; isint = true; }
  else if (k == "temp") val = r.temp;
  else if (k == "press") val = r.press;
  else if (k == "pe") val = r.pe;
  else if (k == "ke") val = r.ke;
  else if (k == "etotal") val = r.etotal;
  else if (k == "enthalpy") val = r.etotal + r.press * (vol / norm) / nktv2p;       // :1732-1745
  else if (k == "evdwl" || k == "epair") val = r.epair;                             // (no Coulomb, no tail correction)
  else if (k == "ecoul" || k == "elong" || k == "etail" || k == "edihed" || k == "eimp") val = 0.0;
  else if (k == "ebond") val = r.ebond / norm;
  else if (k == "eangle") val = r.eangle / norm;
  else if (k == "emol") val = r.emol;
  else if (k == "vol") val = vol;
  else if (k == "density") {                                                         // :1882-1887 (mv2d = 1 in lj units)
    double m = 0.0;
    for (int i = 0; i < natoms; i++) m += mass[type[i]];
    val = mv2d * m / vol;
  }
  else if (k == "lx" || k == "ly" || k == "lz") val = box.prd[k[1] - 'x'];
  else if (k == "xlo" || k == "ylo" || k == "zlo") val = box.lo[k[0] - 'x'];
  else if (k == "xhi" || k == "yhi" || k == "zhi") val = box.hi[k[0] - 'x'];
  else if (k == "xy" || k == "xz" || k == "yz") val = 0.0;
  else if (k == "pxx" || k == "pyy" || k == "pzz" || k == "pxy" || k == "pxz" || k == "pyz") {   // :2024-2063
    static const char *names[6] = {"pxx", "pyy", "pzz", "pxy", "pxz", "pyz"};
    want_ptensor = true;       // (from the next thermo evaluation on; `thermo_style` names its keywords before the run)
    val = 0.0;
    for (int c = 0; c < 6; c++) if (k == names[c]) val = r.ptensor[c];
  }
  else if (k == "bonds") { val = (double)nbonds; isint = true; }     // (atom->nbonds at the time of the call, src/thermo.cpp:1996-1999)
  else if (k == "angles") { val = (double)nangles; isint = true; }
  else if (k == "dihedrals" || k == "impropers") { val = 0.0; isint = true; }
  else if (k == "nbuild") { val = (double)neigh_builds; isint = true; }             // :2099-2110
  else if (k == "ndanger") { val = (double)neigh_dangerous; isint = true; }
  else if (k.rfind("f_", 0) == 0) {
    size_t b = k.find('[');
    std::string id = k.substr(2, b == std::string::npos ? std::string::npos : b - 2);
    Fix *f = find_fix(id);
    if (!f) throw LammpsError("Could not find thermo fix ID " + id);
    int idx = (b == std::string::npos) ? 1 : atoi(k.c_str() + b + 1);
    val = f->compute_vector(idx - 1);
  }
  else return false;
  return true;
}

static const char *thermo_title(const std::string &k) {       // column titles: src/thermo.cpp:716-880
  static const char *names[][2] = {
      {"step", "Step"}, {"elapsed", "Elapsed"}, {"elaplong", "Elaplong"}, {"dt", "Dt"}, {"time", "Time"}, {"cpu", "CPU"},
      {"atoms", "Atoms"}, {"temp", "Temp"}, {"press", "Press"}, {"pe", "PotEng"}, {"ke", "KinEng"}, {"etotal", "TotEng"},
      {"enthalpy", "Enthalpy"}, {"evdwl", "E_vdwl"}, {"ecoul", "E_coul"}, {"epair", "E_pair"}, {"ebond", "E_bond"},
      {"eangle", "E_angle"}, {"edihed", "E_dihed"}, {"eimp", "E_impro"}, {"emol", "E_mol"}, {"elong", "E_long"},
      {"etail", "E_tail"}, {"vol", "Volume"}, {"density", "Density"}, {"lx", "Lx"}, {"ly", "Ly"}, {"lz", "Lz"}, {"xlo", "Xlo"},
      {"xhi", "Xhi"}, {"ylo", "Ylo"}, {"yhi", "Yhi"}, {"zlo", "Zlo"}, {"zhi", "Zhi"}, {"xy", "Xy"}, {"xz", "Xz"}, {"yz", "Yz"},
      {"pxx", "Pxx"}, {"pyy", "Pyy"}, {"pzz", "Pzz"}, {"pxy", "Pxy"}, {"pxz", "Pxz"}, {"pyz", "Pyz"}, {"bonds", "Bonds"}, {"angles", "Angles"}, {"dihedrals", "Diheds"}, {"impropers", "Impros"}, {"nbuild", "Nbuild"},
      {"ndanger", "Ndanger"}};
  for (auto &n : names) if (k == n[0]) return n[1];
  return nullptr;
}

void Engine::print_thermo_header() {
  thermo_first_line = true;
  if (thermo_multi) return;                  // (src/thermo.cpp:314: no header line)
  std::string h;
  for (auto &k : thermo_keywords) {
    const char *t = thermo_title(k);
    h += (t ? std::string(t) : k) + " ";
  }
  say(h + "\n");
}
void Engine::print_thermo(const ThermoRow &r) {
  std::string line;
  char buf[96];
  if (thermo_multi) {
    snprintf(buf, sizeof buf, "---------------- Step %8ld ----- CPU = %11.4f (sec) ----------------", r.step,
             thermo_first_line ? 0.0 : wall() - run_wall0);
    line = buf;
  }
  size_t col = 0;
  for (auto &k : thermo_keywords) {
    double val;
    bool isint;
    if (!thermo_keyword(r, k, val, isint)) throw LammpsError("Unknown keyword in thermo_style custom command: " + k);
    if (thermo_multi) {                      // "\n" before every third value, `%-8s = %14.4f ` (src/thermo.cpp:239-266)
      if (col % 3 == 0) line += "\n";
      const char *t = thermo_title(k);
      if (isint) snprintf(buf, sizeof buf, "%-8s = %14ld ", t ? t : k.c_str(), (long)val);
      else snprintf(buf, sizeof buf, "%-8s = %14.4f ", t ? t : k.c_str(), val);
    } else if (isint) snprintf(buf, sizeof buf, "%8ld ", (long)val);
    else snprintf(buf, sizeof buf, "%12.8g ", val);
    line += buf;
    col++;
  }
  say(line + "\n");
  thermo_first_line = false;
}

// ---------------------------------------------------------------------------------------------
// Verlet::setup (src/verlet.cpp:87-156)
// ---------------------------------------------------------------------------------------------
void Engine::setup() {
  const bool trace = getenv("LAMMPS_LE_TRACE_RUN") != nullptr;
  double s0 = wall();
  reneighbor(false, sortfreq > 0);    // pbc + (spatial sort) + Atom::sort + lists; ncalls reset below
  double s1 = wall();
  neigh_builds = 0;
  compute_forces(true);
  FixLangevin *lg = the_langevin(this);
  for (auto &f : fixes) f->setup();
  if (lg) langevin_post_force(this, lg, false);   // FixLangevin::setup -> post_force (:372-373)
  double s2 = wall();
  last_thermo = eval_thermo();
  double s3 = wall();
  if (trace) fprintf(stderr, "[setup] reneighbor %.2f ms, forces+fixes (enqueue) %.2f ms, thermo %.2f ms\n", 1e3 * (s1 - s0), 1e3 * (s2 - s1), 1e3 * (s3 - s2));
  thermo_log.push_back(last_thermo);
  print_thermo_header();
  print_thermo(last_thermo);
  write_dumps(ntimestep);            // Output::setup (src/output.cpp:150-200): snapshot of the initial state
}

// Verlet::run (src/verlet.cpp:223-354).  With the standard fix set (one fix nve, at most one fix langevin) a
// non-thermo step is ONE fused kernel: forces + post_force + final_integrate + the next step's
// initial_integrate (k_step); LE fixes and the reneighbor decision run between two such kernels exactly
// where the reference runs them (after initial_integrate, before the force computation).
void Engine::iterate(long nsteps) {
  DeviceState &d = *dev;
  FixLangevin *lg = the_langevin(this);
  int nnve = count_nve(this);
  double triggersq = 0.25 * skin * skin;
  bool fusable = (nnve == 1) && !getenv("LAMMPS_LE_NO_FUSE");
  // fix nve / fix langevin on a group other than all: the group variant of the fused step kernel (one fix nve, a pair style;
  // thermo steps and everything else take the unfused kernels, which test the bead's group bits)
  const std::vector<int> nbits = nve_bits(this);
  const bool grouped = md_fixes_on_groups(this);
  int gnve = 1, glg = 1;
  if (grouped) {
    if (nnve == 1 && step_fuses_groups(d, pair_lj, angles_active()) && !(lg && langevin_members == 0)) {
      gnve = nbits[0];
      glg = lg ? lg->groupbit : 1;
    }
    else fusable = false;
  }
  if (lg && lg->zeroflag) fusable = false;         // (`zero yes` needs the group's summed random force before final_integrate)
  // bond morse (the reference's unit-test partner of bond hybrid, not a style of the chromatin model) lives in the
  // unfused force kernel only: its exp() would cost the fused step kernel registers every run pays for
  for (int b = 1; b <= nbondtypes; b++) if (bondtab.style[b] == 3) fusable = false;
  // angles (semiflexible chains): a kernel of their own writes the angle forces right before the fused step
  // kernel, which adds them to its sums (decomposed runs too: every rank for the beads it owns)
  const bool ang = angles_active();
  if (ang && !step_fuses_angles(d, pair_lj)) fusable = false;      // (no pair style: force kernel -> angle kernel -> integrate kernels)
  if (ang && fusable) upload_angle_table(d, angtab);
  bool ident = d.ident_order;
  bool pre_integrated = false;
  // halo/compute overlap issues the per-step halo on a second stream.  With RCCL that means two streams feeding ONE
  // communicator (ordered by events, but never run on multi-GPU hardware): kept to the test transports unless
  // LAMMPS_LE_OVERLAP_RCCL=1 says otherwise
  bool overlap = getenv("LAMMPS_LE_OVERLAP") != nullptr && atoi(getenv("LAMMPS_LE_OVERLAP")) != 0;
  if (overlap && comm && comm->backend == Comm::RCCL && !getenv("LAMMPS_LE_OVERLAP_RCCL")) overlap = false;
  if (angles_active()) overlap = false;       // (the two-phase launches of the overlap mode do not carry the angle forces)
  // test hook (LAMMPS_LE_TEST_FAIL_AT="rank:step"): this rank stops with an error in the middle of a run, the way a
  // rank-local failure (capacity check, LE fix throw) would; its peers must end with an error too, not hang
  long fail_at = -1;
  if (const char *fa = getenv("LAMMPS_LE_TEST_FAIL_AT")) {
    int fr = -1; long fs = -1;
    if (sscanf(fa, "%d:%ld", &fr, &fs) == 2 && fr == rank) fail_at = fs;
  }
  for (long it = 0; it < nsteps; it++) {
    ntimestep++;
    if (ntimestep == fail_at) throw LammpsError("test hook: rank " + std::to_string(rank) + " fails at step " + std::to_string(fail_at));
    bool eflag = (ntimestep == endstep) || (thermo_every > 0 && ntimestep % thermo_every == 0);
    const bool restart_now = restart_every > 0 && ntimestep % restart_every == 0;
    const bool dump_now = (!dumps.empty() && dump_due(ntimestep)) || restart_now;   // needs the complete state of this step: unfused path
    TypeTables tt = make_tables(this, lg);
    stamp();
    if (!pre_integrated) {
      bool will_check = neigh_check && (ago + 1 >= neigh_delay) && ((ago + 1) % neigh_every == 0);
      for (int k = 0; k < nnve; k++) launch_initial_integrate(d, tt, dt, triggersq, will_check && k == nnve - 1, nbits[k]);
    }
    for (auto &f : fixes) if (f->has_post_integrate) f->post_integrate();
    stamp(T_MODIFY);
    if (decide()) {
      const bool sort_due = sortfreq > 0 && ntimestep >= nextsort;
      stamp();
      reneighbor(fusable && !eflag && !dump_now && !sort_due, sort_due);
      stamp(T_NEIGH);
    } else {
      stamp();
      halo_exchange_once();     // ghosts follow their owners (CommBrick::forward_comm)
      stamp(T_COMM);
    }
    stamp();
    if (fusable && !eflag && !dump_now) {
      bool next = (it + 1 < nsteps);
      bool check_next = neigh_check && (ago + 1 >= neigh_delay) && ((ago + 1) % neigh_every == 0);
      if (lg) langevin_draws(this, lg);
      bool timed = timed_begin(this);
      // opt-in (LAMMPS_LE_OVERLAP=1): parity-tested on one GPU with the in-process and mailbox transports, but the
      // RCCL path on a second stream has not run on multi-GPU hardware yet, so the default keeps every
      // collective on the one engine stream
      if (d.dd) dd_halo_wait(d);               // this step's forces read the ghost slots
      if (d.dd && next && overlap) {
        // phase 1 = beads that are sent to a neighbour or read a ghost; their new positions start travelling on
        // comm_stream (ghost slots of the NEXT step) while phase 0 - the interior - is still being computed
        launch_step(d, bondtab, special_lj, tt, lg != nullptr, next, ident, pair_lj, dt, triggersq, check_next, nullptr,
                    nullptr, 1, false, false, false, gnve, glg);
        HIP_CHECK(hipEventRecord(d.ev_phase1, d.stream));
        launch_step(d, bondtab, special_lj, tt, lg != nullptr, next, ident, pair_lj, dt, triggersq, check_next,
                    timed ? d.ev0[d.ev_used] : nullptr, timed ? d.ev1[d.ev_used] : nullptr, 0, false, false, false, gnve, glg);
        HIP_CHECK(hipStreamWaitEvent(d.comm_stream, d.ev_phase1, 0));
        dd_halo(d, *comm, d.comm_stream, d.pos_tmp, d.pos_tmp);
        HIP_CHECK(hipEventRecord(d.ev_halo, d.comm_stream));
        d.halo_inflight = true;
        d.halo_ahead = true;
        std::swap(d.pos, d.pos_tmp);
      } else {
        // (a whole step evaluates the listed angles inside the step kernel; the last step of a run, which stores forces,
        //  takes them from the angle kernel)
        if (ang && !next) launch_angle(d, angtab, false, true);
        launch_step(d, bondtab, special_lj, tt, lg != nullptr, next, ident, pair_lj, dt, triggersq, check_next,
                    timed ? d.ev0[d.ev_used] : nullptr, timed ? d.ev1[d.ev_used] : nullptr, -1, true, ang, false, gnve, glg);
        if (!finish_reneighbor()) {
          // a list of the build that preceded this launch overflowed: the kernel saw the flag and stored nothing.
          // Undo the launch on the host side, grow the table, rebuild, launch again.
          if (next) std::swap(d.pos, d.pos_tmp);
          regrow_lists();
          if (ang && !next) launch_angle(d, angtab, false, true);
          launch_step(d, bondtab, special_lj, tt, lg != nullptr, next, ident, pair_lj, dt, triggersq, check_next,
                      timed ? d.ev0[d.ev_used] : nullptr, timed ? d.ev1[d.ev_used] : nullptr, -1, true, ang, false, gnve, glg);
        }
      }
      if (timed) d.ev_used++;
      if (lg) rng_langevin_consumed(d);
      pre_integrated = next;
      stamp(T_PAIR);          // the fused kernel: pair + bond + post_force + final_integrate (+ next initial_integrate)
    } else if (fusable && eflag && !dump_now && !ang && !grouped && step_fuses_energy(d, pair_lj)) {
      // a thermo step without dumps: the step kernel's energy variant - forces, energies, virial, post_force and
      // final_integrate in one pass; the next step starts with its own initial_integrate (thermo reads the velocities
      // of the END of this step)
      if (!finish_reneighbor()) regrow_lists();
      if (lg) langevin_draws(this, lg);
      launch_step(d, bondtab, special_lj, tt, lg != nullptr, false, ident, pair_lj, dt, triggersq, false, nullptr, nullptr, -1,
                  true, false, true);
      if (lg) rng_langevin_consumed(d);
      pre_integrated = false;
      stamp(T_PAIR);
      last_thermo = eval_thermo(true);
      thermo_log.push_back(last_thermo);
      print_thermo(last_thermo);
      stamp(T_OUTPUT);
    } else {
      if (!finish_reneighbor()) regrow_lists();
      if (d.dd) dd_halo_wait(d);
      compute_forces(eflag);
      stamp(T_PAIR);          // k_force: pair + bond in one pass
      // (Langevin and the final half-kick in one kernel only when both fixes act on the same atoms)
      const bool lg_fused_final = lg && nnve == 1 && nbits[0] == lg->groupbit && langevin_members > 0 && !lg->zeroflag;
      if (lg) langevin_post_force(this, lg, lg_fused_final);
      if (!lg_fused_final)
        for (int k = 0; k < nnve; k++) launch_final_integrate(d, tt, nbits[k]);
      pre_integrated = false;
      stamp(T_MODIFY);
      if (eflag) {
        last_thermo = eval_thermo();
        thermo_log.push_back(last_thermo);
        print_thermo(last_thermo);
      }
      if (dump_now) write_dumps(ntimestep);
      if (restart_now) write_periodic_restart(ntimestep);
      if (eflag || dump_now || restart_now) stamp(T_OUTPUT);
    }
  }
  stamp();
  stream_sync(d);
  stamp(T_PAIR);              // work still in flight when the host leaves the loop is the last step kernel
}

// ---------------------------------------------------------------------------------------------
// run_style respa (src/respa.cpp).  Unfused kernels, one GPU; the per-level force arrays live by tag.
// ---------------------------------------------------------------------------------------------
static TypeTables level_tables(Engine *e, double step) {      // FixNVE::*_integrate_respa: dtf = 0.5 * step_respa[ilevel] * ftm2v
  TypeTables tt{};
  const double dtf = 0.5 * step * e->ftm2v;
  for (int t = 1; t <= e->ntypes; t++) { tt.mass[t] = e->mass[t]; tt.dtfm[t] = dtf / e->mass[t]; }
  return tt;
}
void Engine::respa_level_forces(int l) {                       // Respa::recurse, force part (:664-713): pair, then bond
  const int parts = (respa_level_pair == l ? 1 : 0) | (respa_level_bond == l ? 2 : 0);
  launch_force(*dev, bondtab, special_lj, false, pair_lj, parts);
  if (respa_level_angle == l && angles_active()) launch_angle(*dev, angtab, false);      // (:707-710: behind the level's bonds)
}
void Engine::respa_setup() {                                   // Respa::setup (:369-468)
  DeviceState &d = *dev;
  const int top = respa_levels - 1;
  respa_step[top] = dt;                                        // Respa::init :340-344
  for (int l = top - 1; l >= 0; l--) respa_step[l] = respa_step[l + 1] / respa_loop[l];
  const size_t need = 3 * (size_t)(d.maxtag + 2);
  for (int l = 0; l <= top; l++) {
    if (respa_flevel[l] && respa_flevel_n != need) { HIP_CHECK(hipFree(respa_flevel[l])); respa_flevel[l] = nullptr; }
    if (!respa_flevel[l]) HIP_CHECK(hipMalloc((void **)&respa_flevel[l], need * sizeof(double)));
    HIP_CHECK(hipMemsetAsync(respa_flevel[l], 0, need * sizeof(double), d.stream));
  }
  respa_flevel_n = need;
  reneighbor(false, sortfreq > 0);
  neigh_builds = 0;
  for (int l = 0; l <= top; l++) {
    respa_level_forces(l);
    launch_flevel_copy(d, respa_flevel[l], true, false);
  }
  FixLangevin *lg = the_langevin(this);
  for (auto &f : fixes) f->setup();
  if (lg) {   // FixLangevin::setup, respa branch (src/fix_langevin.cpp:372-378): into the outermost level's array
    launch_flevel_copy(d, respa_flevel[top], false, false);
    langevin_post_force(this, lg, false);
    launch_flevel_copy(d, respa_flevel[top], true, false);
  }
  compute_forces(true);              // energies and virial of the initial state for the thermo line (forces are reloaded per level)
  last_thermo = eval_thermo();
  thermo_log.push_back(last_thermo);
  print_thermo_header();
  print_thermo(last_thermo);
  write_dumps(ntimestep);
}
// Respa::recurse (:600-741).  `last`: every enclosing loop is in its last trip, i.e. the innermost move that follows is the
// last of this timestep: its kernel also tests the skin/2 displacement the NEXT step's Neighbor::decide asks about (the
// reference decides at the outermost level before any inner level has moved x, on the positions the last step left).
void Engine::respa_recurse(int l, bool last) {
  DeviceState &d = *dev;
  const int top = respa_levels - 1;
  const int nnve = count_nve(this);
  const double triggersq = 0.25 * skin * skin;
  FixLangevin *lg = the_langevin(this);
  const std::vector<int> nbits = nve_bits(this);                              // (fix nve on a group: the respa variants use the same mask)
  launch_flevel_copy(d, respa_flevel[l], false, false);                       // copy_flevel_f
  const TypeTables ttl = level_tables(this, respa_step[l]);
  for (int iloop = 0; iloop < respa_loop[l]; iloop++) {
    const bool last_here = last && iloop == respa_loop[l] - 1;
    stamp();
    for (int k = 0; k < nnve; k++) {                                         // initial_integrate_respa (src/fix_nve.cpp:145-155)
      if (l == 0) {
        const bool will_check = last_here && k == nnve - 1 && neigh_check && (ago + 1 >= neigh_delay) && ((ago + 1) % neigh_every == 0);
        launch_initial_integrate(d, ttl, respa_step[0], triggersq, will_check, nbits[k]);
      } else launch_final_integrate(d, ttl, nbits[k]);
    }
    if (l == top)                                                            // post_integrate_respa: outermost level only
      for (auto &f : fixes) if (f->has_post_integrate) f->post_integrate();
    stamp(T_MODIFY);
    if (l == top && decide()) {
      const bool sort_due = sortfreq > 0 && ntimestep >= nextsort;
      stamp();
      // decomposed: beads may change owner in this rebuild - the level tables are completed on every rank first
      if (world > 1) for (int q = 0; q <= top; q++) dd_gather_rows3(d, *comm, respa_flevel[q]);
      reneighbor(false, sort_due);
      stamp(T_NEIGH);
    } else if (l == 0 && world > 1) {
      stamp();
      halo_exchange();          // the innermost level moved the positions: ghosts follow before this level's forces
      dd_halo_wait(d);
      stamp(T_COMM);
    }
    if (l) respa_recurse(l - 1, last_here);
    stamp();
    respa_level_forces(l);
    stamp(T_PAIR);
    if (l == top && lg) langevin_post_force(this, lg, false);                // post_force_respa (src/fix_langevin.cpp:576-579)
    for (int k = 0; k < nnve; k++) launch_final_integrate(d, ttl, nbits[k]); // final_integrate_respa (src/fix_nve.cpp:159-163)
    stamp(T_MODIFY);
  }
  launch_flevel_copy(d, respa_flevel[l], true, false);                        // copy_f_flevel
}
void Engine::respa_iterate(long nsteps) {                                    // Respa::run (:544-575)
  DeviceState &d = *dev;
  for (long it = 0; it < nsteps; it++) {
    ntimestep++;
    const bool eflag = (ntimestep == endstep) || (thermo_every > 0 && ntimestep % thermo_every == 0);
    const bool restart_now = restart_every > 0 && ntimestep % restart_every == 0;
    const bool dump_now = (!dumps.empty() && dump_due(ntimestep)) || restart_now;
    respa_recurse(respa_levels - 1, true);
    if (eflag || dump_now) {
      stamp();
      if (eflag) {
        compute_forces(true);        // the step's energies / virial: pair at the outer, bond at the last inner evaluation = this state
        last_thermo = eval_thermo();
        thermo_log.push_back(last_thermo);
        print_thermo(last_thermo);
      }
      // sum_flevel_f (:817-843): the total force of the step, for outputs that read it
      launch_flevel_copy(d, respa_flevel[0], false, false);
      for (int l = 1; l < respa_levels; l++) launch_flevel_copy(d, respa_flevel[l], false, true);
      if (dump_now && !dumps.empty() && dump_due(ntimestep)) write_dumps(ntimestep);
      if (restart_now) write_periodic_restart(ntimestep);
      stamp(T_OUTPUT);
    }
  }
  stamp();
  stream_sync(d);
  stamp(T_PAIR);
}

// "MPI task timing breakdown" of src/finish.cpp:318-370 (one task: min = avg = max, %varavg = 0).  Bond is part of the
// Pair row here: pair and bond forces are ONE kernel (k_step / k_force), there is no boundary to stamp between them.
void Engine::print_timing_breakdown(long nsteps) {
  if (timer_level < 2 || nsteps <= 0) return;
  const double tl = loop_time > 0.0 ? loop_time : 1.0;
  std::string out = "\nMPI task timing breakdown:\nSection |  min time  |  avg time  |  max time  |%varavg| %total\n"
                    "---------------------------------------------------------------\n";
  char buf[160];
  double all = 0.0;
  auto row = [&](const char *label, int which) {
    const double t = timers[which];
    all += t;
    snprintf(buf, sizeof buf, "%-8s| %-10.5g | %-10.5g | %-10.5g |%6.1f |%6.2f\n", label, t, t, t, 0.0, t / tl * 100.0);
    out += buf;
  };
  row("Pair", T_PAIR);
  if (atom_style != "atomic") row("Bond", T_BOND);
  row("Neigh", T_NEIGH);
  row("Comm", T_COMM);
  row("Output", T_OUTPUT);
  row("Modify", T_MODIFY);
  const double other = loop_time - all;
  snprintf(buf, sizeof buf, "Other   |            | %-10.4g |            |       |%6.2f\n", other, other / tl * 100.0);
  out += buf;
  if (!timer_sync)   // the reference's own words for its GPU build (src/finish.cpp:411-419)
    out += "(host wall clock per section; the device runs asynchronously behind it - `timer sync` gives the device-accurate split)\n";
  out += "\n";
  say(out);
}

void Engine::run(long nsteps) {
  if (nsteps < 0) throw LammpsError("Invalid run command N value");
  // checks every rank fails identically are made before anything collective starts: they must not cost the communicator
  for (int a = 1; a <= nangletypes && apa > 0 && nangles > 0 && !angle_style_name.empty() && angle_style_name != "none" && angle_style_name != "zero"; a++)
    if (!angtab.style[a]) throw LammpsError("All angle coeffs are not set");
  // a run that ended in an error tore the communicator down (below); halo sequence numbers and arrival counters of the
  // ranks may disagree from then on, so nothing decomposed runs on this handle again
  if (comm && world > 1) comm->require_alive();
  const bool trace = getenv("LAMMPS_LE_TRACE_RUN") != nullptr;
  double tr0 = wall();
  init();
  double tr1 = wall();
  // (a fix on a group that was defined since the last upload: the masks and the member ranks travel with an upload)
  if (dev && dev_current && fixes_on_groups(this) && group_sig != group_signature()) { download(); dev_current = false; }
  if (!dev_current || !dev || !dev->pos) upload();
  group_sig = group_signature();
  if (dev->dd) dd_fast_halo_switch(*dev);
  for (int k = 1; k <= 3; k++) dev->sflag[k] = special_flag(k);
  dev->ident_order = local_order_is_tag_order();
  {
    // bond partner images: frozen at the reneighbor as in the reference (src/ntopo_bond_all.cpp:52-73), unless the minimum
    // image of every step is provably the same thing (device.h bond_minimg)
    double r0max = 0.0;
    bool all_fene = nbondtypes > 0;
    for (int b = 1; b <= nbondtypes; b++) {
      if (bondtab.style[b] == 1) r0max = std::max(r0max, bondtab.p1[b]);
      else if (bondtab.style[b] != 0) all_fene = false;
    }
    const double halfmin = 0.5 * std::min({box.prd[0], box.prd[1], box.prd[2]});
    dev->bond_minimg = (all_fene && 2.0 * r0max * 1.000001 < halfmin && !getenv("LAMMPS_LE_FREEZE_IMAGES")) ? 1 : 0;
  }
  dev->newton_pair = newton_pair ? 1 : 0;
  for (int k = 0; k < 3; k++) {
    const double binsize_optimal = cutneighmax > 0.0 ? 0.5 * cutneighmax : box.prd[0];
    int nb = (int)(box.prd[k] * (1.0 / binsize_optimal));
    if (nb == 0) nb = 1;
    dev->ref_nbin[k] = nb;
    dev->ref_bininv[k] = 1.0 / (box.prd[k] / nb);
  }
  double tr2 = wall();
  le_reneigh_step.assign(fixes.size(), -1);
  dev->le_snapshot = 0;
  for (auto &f : fixes) if (f->force_reneighbor) dev->le_snapshot = 1;
  dev->topo_dirty = true;     // bond tables may have been edited between runs
  dev->bond_pack_dirty = true;
  dev->angle_pack_dirty = true;
  beginstep = ntimestep;
  endstep = ntimestep + nsteps;
  atimestep = ntimestep;            // Integrate::init (src/integrate.cpp:48)
  halo_step = -1;
  host_current = false;
  double t0 = 0.0;
  try {
    if (respa_levels > 0) respa_setup(); else setup();
    for (int k = 0; k < 8; k++) timers[k] = 0.0;     // Timer::init() comes after setup (src/run.cpp:176-181)
    dev->ev_used = 0;
    ktime_counter = 0;
    ktime_every = nsteps <= 64 ? 1 : 16;
    t0 = wall();
    run_wall0 = t0;
    if (respa_levels > 0) respa_iterate(nsteps); else iterate(nsteps);
    loop_time = wall() - t0;
    atime += (double)(ntimestep - atimestep) * dt;       // Update::update_time at the end of a run (src/verlet.cpp:362)
    atimestep = ntimestep;
  } catch (...) {
    // a rank that leaves the loop on an error must not keep its peers inside a collective: tear the communicator down
    // (they end with "communicator aborted" or their own copy of the error; see Comm::wait_stream)
    if (comm && world > 1) { try { comm->abort(); } catch (...) {} }
    throw;
  }
  if (trace) fprintf(stderr, "[run %ld] init %.2f ms, upload %.2f ms, setup %.2f ms, loop %.2f ms\n", nsteps, 1e3 * (tr1 - tr0), 1e3 * (tr2 - tr1), 1e3 * (t0 - tr2), 1e3 * loop_time);
  kstat_ms = 0.0; kstat_n = 0;
  for (size_t k = 0; k < dev->ev_used; k++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, dev->ev0[k], dev->ev1[k]) == hipSuccess) { kstat_ms += ms; kstat_n++; }
  }
  if (kstat_n) kstat_ms /= kstat_n;
  sync_flags(*dev);
  check_device_error(this, *dev);
  char buf[256];
  snprintf(buf, sizeof buf, "Loop time of %g on 1 procs for %ld steps with %d atoms\n\n", loop_time, nsteps, natoms);
  say(buf);
  if (nsteps > 0 && loop_time > 0) {
    double sps = nsteps / loop_time;
    if (units == "lj") snprintf(buf, sizeof buf, "Performance: %.3f tau/day, %.3f timesteps/s\n", 86400.0 * sps * dt, sps);
    else snprintf(buf, sizeof buf, "Performance: %.3f ns/day, %.3f timesteps/s\n", 86400.0 * sps * dt * 1e-6, sps);
    say(buf);
  }
  print_timing_breakdown(nsteps);
  snprintf(buf, sizeof buf, "Neighbor list builds = %ld\nDangerous builds = %ld\n", neigh_builds, neigh_dangerous);
  say(buf);
  if (dev->flags_h[FLAG_FENE_WARN]) {
    snprintf(buf, sizeof buf, "FENE bond too long warnings: %d", dev->flags_h[FLAG_FENE_WARN]);
    warning(buf);
  }
}

}  // namespace lmp_le
