// engine.h — MI355X-native bead-spring MD engine behind the reference's script / C-library boundary.
//
// Host side (C++17) mirrors the reference objects that the hot path talks to
// (/root/reference/src: input.cpp, read_data.cpp, modify.cpp, verlet.cpp, neighbor.cpp,
// pair_lj_cut.cpp, MOLECULE/bond_fene.cpp, fix_nve.cpp, fix_langevin.cpp, USER-LE/fix_*.cpp);
// all per-atom work runs as hand-written HIP kernels for gfx950 (kernels_*.hip).
#pragma once
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace lmp_le {

struct LammpsError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

constexpr int MAXTYPES = 16;       // atom types and bond types handled by the by-value kernel tables
constexpr int MAXBPA = 8;          // bonds per atom the bond-partner table can hold
constexpr int MS_MAX = 32;         // max special neighbors per atom handled by the device rebuild
constexpr int NEIGH_SB_SHIFT = 30; // special-bond bits, as src/lmptype.h:61-62
constexpr int NEIGH_MASK = 0x3FFFFFFF;
constexpr int BOND_TYPE_SHIFT = 26; // bond partner table: (type << 26) | partner index
constexpr int BOND_IDX_MASK = (1 << BOND_TYPE_SHIFT) - 1;
// numneigh word of a bead: list entries in the low 16 bits (bonds + pairs), of which the first (word >> 16) are bond
// entries in the bond-partner table's encoding (kernels_md.hip pair_loop)
constexpr int NN_BOND_SHIFT = 16;
constexpr int NN_COUNT_MASK = (1 << NN_BOND_SHIFT) - 1;
constexpr int NN_NBOND_MASK = 0xFF;          // bond entries of the word: (word >> NN_BOND_SHIFT) & NN_NBOND_MASK
// bit 30 of the word: some bond of this bead reaches its partner through a periodic image.  The image of a bond partner is
// FROZEN at the reneighbor (src/ntopo_bond_all.cpp:52-73: domain->closest_image picks a ghost, whose shift then stays),
// two bits per dimension and bond slot in DeviceState::bshift (0 none, 1 partner at +prd, 2 partner at -prd)
constexpr int NN_SHIFTED_BIT = 1 << 30;
constexpr int BSHIFT_BITS = 6;

// ---------------------------------------------------------------------------------------------
// RanMars in exact integer arithmetic (src/random_mars.cpp:29-95; every value is k * 2^-24)
// ---------------------------------------------------------------------------------------------
struct RanMarsInt {
  // history window w[k] = y_{n-97+k}, k = 0..96, of the lagged-Fibonacci part
  //   y_n = y_{n-97} - y_{n-33}  (mod 2^24),  y_{-k} = u[k] of the reference's seed table
  uint32_t w[97];
  uint64_t n;  // raw index of the NEXT value to generate (the constructor's warm-up draw was n = 0)
  void seed(int seed);
  uint32_t next_raw();  // 24-bit integer of the next uniform(): uniform = value / 2^24
  double uniform() { return next_raw() * (1.0 / 16777216.0); }
  static uint32_t c_of(uint64_t n);  // arithmetic-sequence term used for raw index n
  void jump(uint64_t k);             // advance by k draws in O(97^2 log k)
};
// coefficients a[0..96] of x^k mod (x^97 + x^64 - 1) over Z/2^24: y_{n+k} = sum_j a[j] y_{n+j}
void ranmars_jump_poly(uint64_t k, uint32_t a[97]);

// ---------------------------------------------------------------------------------------------
// device-side views (plain pointers; filled by Engine, consumed by the kernel launchers)
// ---------------------------------------------------------------------------------------------
struct Box {
  double lo[3], hi[3], prd[3], half[3], iprd[3];
};

struct TypeTables {  // by-value kernel argument
  double dtfm[MAXTYPES + 1];   // dtf / mass[type]           (src/fix_nve.cpp:95)
  double g1[MAXTYPES + 1];     // gfactor1[type]             (src/fix_langevin.cpp:296-310)
  double g2[MAXTYPES + 1];     // gfactor2[type] * tsqrt     (src/fix_langevin.cpp:663)
  double mass[MAXTYPES + 1];
};

struct PairTable {  // lj/cut per type pair, row-major (ntypes+1)^2; lives in device memory
  int nt;           // ntypes + 1
  double special_lj[4];
};

struct BondTable {  // by-value kernel argument
  int style[MAXTYPES + 1];  // 0 none/zero, 1 fene, 2 harmonic, 3 morse (unfused force kernel only)
  double p0[MAXTYPES + 1], p1[MAXTYPES + 1], p2[MAXTYPES + 1], p3[MAXTYPES + 1];
};

struct AngleTable {  // by-value kernel argument: angle_style harmonic | cosine (src/MOLECULE/angle_harmonic.cpp, angle_cosine.cpp)
  int style[MAXTYPES + 1];     // 0 none/zero, 1 harmonic, 2 cosine
  double k[MAXTYPES + 1], theta0[MAXTYPES + 1];   // theta0 in radians
};

struct DeviceState;  // defined in device.h (HIP side)
struct Comm;         // defined in comm.h

// ---------------------------------------------------------------------------------------------
// fixes (style registry mirrors src/modify.cpp:93-99 / style names of the reference)
// ---------------------------------------------------------------------------------------------
class Engine;

struct Fix {
  std::string id, group, style;
  int groupbit = 1;          // bit of the fix's group in Engine::gmask (1 = all)
  Engine *eng = nullptr;
  bool force_reneighbor = false;
  virtual ~Fix() {}
  virtual void init() {}
  virtual void setup() {}
  // which hooks this fix has (src/fix.h:247-273 mask bits)
  bool has_initial_integrate = false, has_post_integrate = false, has_post_force = false,
       has_final_integrate = false;
  virtual void initial_integrate() {}
  virtual void post_integrate() {}
  virtual void post_force() {}
  virtual void final_integrate() {}
  virtual double compute_vector(int) { throw LammpsError("Fix does not compute a vector"); }
};

struct FixNVE : Fix {
  FixNVE(Engine *e, const std::vector<std::string> &arg);
};

struct FixLangevin : Fix {
  double t_start, t_stop, t_period;
  int seed;
  RanMarsInt rng;          // host mirror of the stream position (kept in sync by draw counting)
  uint64_t draws = 0;      // Langevin draws consumed so far (3 per atom per post_force call)
  bool dev_ready = false;  // device block states initialised
  std::vector<double> ratio;   // keyword `scale itype ratio`: per-type divisor of the damping time (src/fix_langevin.cpp:135-141)
  bool zeroflag = false;       // keyword `zero yes`: the mean random force of the group is taken off every member (:752-772)
  FixLangevin(Engine *e, const std::vector<std::string> &arg);
};

struct FixExtrusion : Fix {
  int nevery, neutral, ctcf_left, ctcf_right, ctcf_lr, btype;
  double through_prob;
  RanMarsInt rng;
  bool rng_on_device = false;
  int last_break = 0;
  FixExtrusion(Engine *e, const std::vector<std::string> &arg);
  void post_integrate() override;
  double compute_vector(int n) override;
};

struct FixExLoad : Fix {    // also the stock `bond/create` (src/MC/fix_bond_create.cpp), the style ex_load was derived from
  int nevery, iatomtype, jatomtype, btype, imaxbond = 0, inewtype, jmaxbond = 0, jnewtype;
  int atype = 0;             // `atype N`: angles of type N around every new bond, if an angle style is defined (fix_ex_load.cpp:236-254, :855-954)
  bool stock = false;        // bond/create: candidates = every pair of the pair list (not only (i, i+2)), fires at step % N == 0,
  int phase = 3;             //   bonds of btype per bead counted at the first setup and from then on only incremented
  std::vector<int> bondcount;   // [natoms + 2] by tag (stock only)
  bool counted = false;
  double cutsq, fraction = 1.0;
  int seed = 12345;
  RanMarsInt rng;
  bool rng_on_device = false;
  int last_create = 0;
  long total_create = 0;
  FixExLoad(Engine *e, const std::vector<std::string> &arg);
  void init() override;
  void setup() override;
  void post_integrate() override;
  double compute_vector(int n) override;
};

struct FixExUnload : Fix {   // also the stock `bond/break` (src/MC/fix_bond_break.cpp), which differs only in the firing step
  int nevery, btype;
  int angleflag = 0;         // set in init(): the system has angles, broken bonds take theirs along (fix_ex_unload.cpp:149-152)
  void init() override;
  int phase = 2;             // fires when ntimestep % nevery == phase: 2 for ex_unload, 0 for bond/break
  double cutsq, fraction = 1.0;
  int seed = 12345;
  RanMarsInt rng;
  bool rng_on_device = false;
  int last_break = 0;
  long total_break = 0;
  FixExUnload(Engine *e, const std::vector<std::string> &arg);
  void post_integrate() override;
  double compute_vector(int n) override;
};

// ---------------------------------------------------------------------------------------------
// Engine = the LAMMPS instance behind one C-API handle
// ---------------------------------------------------------------------------------------------
struct ThermoRow {
  long step;
  double temp, epair, emol, etotal, press, ke, pe;
  double evdwl, ebond, virial[6];      // (sums over the system; ebond = bonds only)
  long nbonds;
  double eangle = 0.0;                 // sum over the listed angles
  double ptensor[6] = {0, 0, 0, 0, 0, 0};   // pxx pyy pzz pxy pxz pyz (src/compute_pressure.cpp:244-290), when asked for
  bool has_ptensor = false;
};

class Engine {
 public:
  Engine(int argc, char **argv);
  ~Engine();

  // ---- script layer (src/input.cpp:181 file, :327 one, :689 execute_command) ----
  void file(const std::string &path);
  const char *one(const std::string &line);  // returns the command name (borrowed), like Input::one
  void execute(const std::string &cmd, std::vector<std::string> &arg);

  // ---- global state ----
  std::string units = "lj", atom_style = "atomic";
  double boltz = 1, mvv2e = 1, ftm2v = 1, nktv2p = 1, mv2d = 1, dt = 0.005;
  double skin = 0.3;
  int neigh_every = 1, neigh_delay = 10, neigh_check = 1;
  bool newton_pair = true, newton_bond = true;
  int sortfreq = 1000;
  long nextsort = 0;
  double special_lj[4] = {1.0, 0.0, 0.0, 0.0};
  double special_coul[4] = {1.0, 0.0, 0.0, 0.0};   // no Coulomb on this path, but the pair list's special flags depend on it
  // how the pair list treats a special level (src/neighbor.cpp:360-376 special_flag, src/npair.h:112-136 find_special):
  // 0 = pair dropped from the list (lj and coul weight both 0), 1 = stored as an ordinary entry, 2 = stored with the level
  // in its top bits.  The reference also takes 2 when lj == 1 but coul != 1; the factor such an entry carries is 1.0,
  // so it is stored as an ordinary entry here (same list membership, same forces).
  int special_flag(int level) const {
    if (special_lj[level] == 0.0 && special_coul[level] == 0.0) return 0;
    if (special_lj[level] == 1.0) return 1;
    return 2;
  }
  double comm_cutoff = 0.0;
  long ntimestep = 0, beginstep = 0, endstep = 0;
  int thermo_every = 0;
  bool thermo_norm = true;
  std::vector<std::string> thermo_keywords;
  bool box_exist = false;
  Box box{};

  // ---- atoms: host master copies in TAG order (tag t at index t-1) ----
  int natoms = 0, ntypes = 0, nbondtypes = 0;
  int extra_bond = 0, extra_special = 0, bpa = 0, maxspecial = 0;
  long nbonds = 0;
  std::vector<double> mass;                        // [ntypes+1]
  std::vector<int> mass_set;
  std::vector<double> x, v, f;                     // [natoms*3]
  std::vector<int> type, image, molecule;          // image: 3 ints per atom
  std::vector<int> num_bond, bond_type, bond_atom; // [natoms], [natoms*bpa]
  std::vector<int> nspecial, special;              // [natoms*3], [natoms*maxspecial]
  // angles: stored with all three atoms (newton_bond off, src/atom.cpp:1290-1353), atoms as IDs
  int nangletypes = 0, extra_angle = 0, apa = 0;   // apa = angles per atom (0: the system has no angle storage)
  long nangles = 0;
  std::vector<int> num_angle, angle_type, angle_a1, angle_a2, angle_a3;   // [natoms], [natoms*apa]
  std::string angle_style_name;                    // "", "harmonic", "cosine", "zero", "none"
  AngleTable angtab{};
  bool angles_active() const;                      // an angle style with coefficients and angle storage exist
  std::vector<int> crank;                          // canonical (reference local) index of tag t-1
  // the reference's local index of a bead is its ID - 1: true while no Atom::sort ran and the data file listed the atoms
  // in ID order.  Then the order-sensitive kernels index by tag directly instead of through crank[].
  bool local_order_is_tag_order() const;
  bool crank_on_device = false;                    // the device's crank[] is newer than this copy (Atom::sort emulation)
  bool special_built = false;
  bool host_current = true;    // host x/v/f/type/topology reflect the device state
  bool dev_current = false;    // device state reflects the host copies

  // ---- styles ----
  bool pair_lj = false, pair_zero = false;
  double pair_cut_global = 0.0;
  bool pair_shift = false;
  int pair_mix = 0;  // 0 geometric, 1 arithmetic
  std::vector<double> pc_eps, pc_sig, pc_cut;
  std::vector<int> pc_set;
  std::vector<double> lj1, lj2, lj3, lj4, offset, cutsq;
  double cutforcemax = 0.0, cutneighmax = 0.0;
  std::string bond_style_name;                    // "", "fene", "harmonic", "hybrid", "zero"
  std::vector<std::string> bond_hybrid_styles;
  BondTable bondtab{};

  std::vector<std::unique_ptr<Fix>> fixes;
  Fix *find_fix(const std::string &id);

  // ---- run control (src/run.cpp:38-188, src/verlet.cpp) ----
  void run(long nsteps);
  void init();          // lmp->init(): pair coefficients, neighbor cutoffs, fix init, special lists
  void setup();         // Verlet::setup
  void iterate(long n); // Verlet::run
  // run_style respa N loop_1 .. loop_{N-1} [bond L] [pair L] (src/respa.cpp); respa_levels = 0: run_style verlet.  A slow
  // path of unfused kernels (round 3: also decomposed): what the LE fixes' post_integrate_respa hooks need
  int respa_levels = 0, respa_loop[8] = {1, 1, 1, 1, 1, 1, 1, 1}, respa_level_bond = 0, respa_level_pair = 0, respa_level_angle = 0;
  double respa_step[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double *respa_flevel[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // device, by tag
  size_t respa_flevel_n = 0;
  void respa_setup();                          // Respa::setup
  void respa_iterate(long n);                  // Respa::run
  void respa_recurse(int ilevel, bool last);   // Respa::recurse
  void respa_level_forces(int ilevel);
  void compute_forces(bool eflag);
  ThermoRow eval_thermo(bool ke_summed = false);
  void print_thermo_header();
  // one thermo keyword (src/thermo.cpp:1572-2110 compute_*): false when the keyword is unknown.  `r` = the row the energies
  // come from; `isint` = printed with the integer format (BIGINT fields)
  bool thermo_keyword(const ThermoRow &r, const std::string &k, double &val, bool &isint);
  bool want_ptensor = false;           // a pressure-tensor keyword was named: thermo steps also reduce the kinetic tensor
  bool thermo_multi = false;           // thermo_style multi (src/thermo.cpp:115-119, 171-180, 361-366)
  bool thermo_first_line = true;       // the first line of a run prints CPU = 0 (Thermo::firststep)
  double run_wall0 = 0.0;              // wall clock at the start of the run's loop (Timer::TOTAL)
  double atime = 0.0;                  // Update::atime / atimestep: simulation time accumulated over runs with changing dt
  long atimestep = 0;
  void print_thermo(const ThermoRow &);
  ThermoRow last_thermo{};
  std::vector<ThermoRow> thermo_log;
  double loop_time = 0.0;
  long neigh_builds = 0, neigh_dangerous = 0;
  long rng_late_count = 0;     // decomposed runs: late generations of skipped Langevin stream segments (lammps_le_stat)
  int ago = 0;
  // Timer sections of the loop (src/timer.h:25-28); wall clock between stamps as src/timer.cpp:100-135.  The GPU runs
  // asynchronously behind the host, so without `timer sync` a section holds the time the HOST spent in it (waits for
  // the device land where the host first needs a result); `timer sync` drains the stream at every stamp.
  enum { T_PAIR = 0, T_BOND, T_NEIGH, T_COMM, T_OUTPUT, T_MODIFY, T_NSECT };
  double timers[8] = {0};
  int timer_level = 2;         // `timer off|loop|normal|full` (src/timer.cpp:230-300): 0 off, 1 loop, 2 normal, 3 full
  bool timer_sync = false;     // `timer sync|nosync`
  double timer_prev = 0.0;
  void stamp();                // Timer::stamp()      : restart the section clock
  void stamp(int which);       // Timer::stamp(which) : add the time since the previous stamp to a section
  void print_timing_breakdown(long nsteps);   // src/finish.cpp:318-370
  bool kernel_timing = false;  // LAMMPS_LE_KERNEL_TIMING=1: HIP events around sampled step-kernel launches
  double kstat_ms = 0.0;       // mean step-kernel duration of the last run (ms)
  long kstat_n = 0;
  long ktime_counter = 0;      // launches seen by the sampler in this run
  int ktime_every = 16;        // sample every n-th launch (1 for runs of <= 64 steps)
  double stat_neigh_pairs();   // stored full-list entries of the current neighbor list

  // ---- topology on the host (read_data path) ----
  void read_data(const std::string &path);
  void write_data(const std::string &path);
  void velocity(std::vector<std::string> &arg);
  // groups (src/group.cpp): static bit masks by atom; bit 0 = all.  fix nve / fix langevin act on them (unfused kernels)
  std::vector<std::string> group_names = {"all"};
  std::vector<int> gmask;                          // by atom index (ID - 1); empty until a group is defined
  std::string group_sig;                           // group_signature() of the last upload
  std::string group_signature() const;
  int langevin_members = 0;                        // atoms in fix langevin's group (its draws per call / 3); set by upload()
  int group_bit(const std::string &name) const;    // 1 << index, or 0 if there is no such group
  void group_command(std::vector<std::string> &arg);
  // `region ID block | sphere | cylinder | union | intersect ... [side in|out] [units box|lattice]` (src/region*.cpp): static
  // regions, for `group ID region R` and `set region R ...`.  match = !(inside ^ interior), closed boundaries
  struct Region {
    std::string style;
    bool interior = true;
    double p[6] = {0, 0, 0, 0, 0, 0};      // block: xlo xhi ylo yhi zlo zhi; sphere: x y z r; cylinder: c1 c2 r lo hi
    char axis = 'z';
    std::vector<std::string> sub;          // union / intersect
  };
  std::map<std::string, Region> regions;
  void region_command(std::vector<std::string> &arg);
  bool region_match(const Region &r, double x, double y, double z) const;
  void set_command(std::vector<std::string> &arg);   // set atom|type|mol|group ... (src/set.cpp)   // velocity all create|set|scale|zero (src/velocity.cpp)
  // ---- dumps (src/dump_custom.cpp, dump_atom.cpp, dump_local.cpp; compute_property_local.cpp) ----
  struct Dump {
    std::string id, style, path, label = "ENTRIES";
    long every = 0, last = -1;
    int groupbit = 1;              // rows = the members of the dump's group (`mask[i] & groupbit`: src/dump_custom.cpp:607-612,
                                   // dump_atom.cpp:356, dump_dcd.cpp:69,200); dump local takes its rows from the compute's group
    bool unwrap = false;           // dcd: coordinates unwrapped by the image flags (dump_modify unwrap yes)
    int nframes = 0;               // dcd: snapshots written so far (header fields are patched after each one)
    std::vector<std::string> cols;
    FILE *fp = nullptr;
  };
  std::vector<Dump> dumps;
  std::map<std::string, std::vector<std::string>> computes_local;   // compute ID -> property/local attributes
  std::map<std::string, int> computes_local_bit;                    // compute ID -> group bit (both atoms of a bond must be members)
  // ---- restart: own binary format, bit-continuous incl. the RNG streams of the fixes ----
  void write_restart(const std::string &path);
  void read_restart(const std::string &path);
  // `restart N root` | `restart N fileA fileB` | `restart 0` (src/output.cpp:603-700 create_restart, :360-420 write_restart)
  long restart_every = 0;
  std::string restart_a, restart_b;
  int restart_toggle = 0;
  void write_periodic_restart(long step);
  std::map<std::string, std::vector<unsigned char>> restart_fix_state;   // fix ID -> saved state, applied by `fix`
  void apply_restart_state(Fix *f);
  bool dump_due(long step) const;
  void write_dumps(long step);       // downloads, then writes every dump that is due
  void build_special();               // src/special.cpp:55-
  void create_box_atoms_check();

  // ---- ranks (one process per GPU; world > 1 = z-slab spatial decomposition) ----
  Comm *comm = nullptr;
  int world = 1, rank = 0;
  void comm_init(const std::string &backend, int rank, int world, const void *unique_id, const std::string &session);
  void halo_exchange();
  void halo_exchange_once();          // ... unless this step's ghosts are already current (an LE fix asked first)
  long halo_step = -1;

  // ---- device ----
  DeviceState *dev = nullptr;
  void device_init();                 // throws LammpsError if no HIP device
  void upload();                      // host -> device (after read_data / scatter)
  void download();                    // device -> host (x, v, f, type, image, topology)
  void reneighbor(bool defer_check = false, bool sort_now = false);   // pbc + spatial sort + cell lists + neighbor list + bond table
  bool reneigh_pending = false;       // the build's overflow / error flags are published but not yet looked at
  bool finish_reneighbor();           // waits for them; false = a list overflowed (nothing may depend on the lists yet)
  void regrow_lists();                // grow the table and rebuild until every list fits
  bool decide();                      // Neighbor::decide (src/neighbor.cpp:1933-1948)
  void emulate_atom_sort();           // keep `crank` equal to the reference's local order (Atom::sort)
  std::vector<long> le_reneigh_step;  // next_reneighbor per fix

  // ---- output ----
  FILE *screen = stdout, *logfile = nullptr;
  bool echo_screen = false;
  void say(const std::string &s);
  void warning(const std::string &s);
  std::string last_cmd;
  std::map<std::string, std::string> variables;       // name -> current value (index / loop / string) or formula (equal)
  struct VarInfo { std::string style; std::vector<std::string> values; size_t which = 0; };
  std::map<std::string, VarInfo> var_info;            // style + value list of variables made by the `variable` command
  std::string substitute(const std::string &line);
  double evaluate(const std::string &expr);           // equal-style formulas, $(...) and `if` conditions
  double variable_value(const std::string &name);
  // script control flow (src/input.cpp: label :1057, jump :1011, next :1079, if :851, include :989)
  std::string jump_file, jump_label;                  // set by `jump`, consumed by file()
  bool jump_pending = false, jump_skip = false, quit_requested = false;
  int file_depth = 0;

  // error state for the C API (src/library.cpp LAMMPS_EXCEPTIONS behaviour)
  std::string last_error;
  bool has_error = false;

  // extract_* scratch (borrowed pointers handed to the caller)
  std::map<std::string, std::vector<double>> scratch_d;
  std::map<std::string, std::vector<int>> scratch_i;
  std::vector<double *> scratch_rows;
  double scratch_scalar = 0.0;
};

// helpers
double numeric(const std::string &s);
int inumeric(const std::string &s);
std::vector<std::string> split_words(const std::string &line);

}  // namespace lmp_le
