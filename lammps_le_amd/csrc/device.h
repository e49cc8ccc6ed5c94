// device.h — HBM layout of the engine and the kernel launchers (gfx950).
//
// Physical (p-indexed) arrays are kept in CELL order: every reneighbor re-sorts the atoms by
// neighbor cell (ties by tag), so a cell's atoms are one contiguous range and the neighbor
// gathers of the pair kernel stay inside a few L2-resident rows.  Topology (bonds, specials,
// extruder state) stays TAG-indexed and is never permuted; map[tag] -> p links the two.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "engine.h"

namespace lmp_le {

#define HIP_CHECK(x)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess)                                                                         \
      throw LammpsError(std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + \
                        ":" + std::to_string(__LINE__));                                          \
  } while (0)

enum FlagSlot {
  FLAG_MOVED = 0,       // some atom moved more than skin/2 since the last build
  FLAG_ERROR = 1,       // device-side error code (see ERR_*)
  FLAG_NEIGH_OVERFLOW = 2,
  FLAG_FENE_WARN = 3,   // count of "FENE bond too long" warnings
  FLAG_TOPO_CHANGED = 4,
  FLAG_COUNT_A = 5,     // LE fix counters (created / broken)
  FLAG_COUNT_B = 6,
  FLAG_NDRAW = 7,       // number of RNG draws requested by an LE fix
  FLAG_NLIST = 8,       // number of extruder listings
  FLAG_MAXNEIGH = 9,    // largest neighbor count seen at the last build
  FLAG_AUX = 10,
  FLAG_SPECIAL_ASYM = 11,
  FLAG_RECV_UP = 12,    // decomposition: counts received from the upper / lower slab neighbour
  FLAG_RECV_DN = 13,
  FLAG_SEND_BOTH = 14,
  FLAG_GHOST_MIXED = 15, // decomposition: ghosts from below and above are not two separate blocks of the cell order  // decomposition: some bead is in both send lists (slab barely two shells thick)   // some bead's 1-2 list lost an entry its partner still has (see dev_special_remove12)
  FLAG_RNG_MISS = 16,    // decomposition: an owned bead's Langevin draws lie in a stream segment this rank skipped (kernels_rng.hip)
  NFLAGS = 24
};
constexpr int FLAG_SEQ_SLOT = 32;   // the publish sequence number in the mapped host page: a 64-byte sector of its own
enum DevErr {
  ERR_NONE = 0, ERR_BAD_FENE = 1, ERR_BOND_MISSING = 2, ERR_EXT_MULTI = 3, ERR_BPA = 4,
  ERR_SPECIAL = 5, ERR_COUNT_MISMATCH = 6, ERR_NONFINITE = 7, ERR_SPECIAL_SCRATCH = 8, ERR_GHOST_ORDER = 9,
  ERR_HALO_TIMEOUT = 10, ERR_ANGLES = 11
};

// Neighbor cells are cutneigh wide in y and z and cutneigh / CELL_XSPLIT wide in x (the fastest index of the cell
// order): the 2*CELL_XSPLIT+1 x-cells a bead has to look at are still ONE contiguous index range per (y,z) row, but
// cover 2.25 instead of 3 cutoffs -> 25 % fewer candidates in the list build.
constexpr int CELL_XSPLIT = 4;

constexpr int ANGLE_PACK_COLS = 4;
constexpr int LE_MAX_FIXES = 16;   // extrusion / ex_load / ex_unload / bond/create / bond/break instances with a device RanMars state each

struct DeviceState {
  hipStream_t stream = nullptr;
  int n = 0;        // owned atoms
  int npad = 0;     // stride of the column-major (ELL) tables, multiple of 64
  int maxtag = 0;
  int ntypes = 0, bpa = 0, maxspecial = 0;
  Box box{};

  // ---- physical order ----
  double4 *pos = nullptr, *pos_tmp = nullptr;   // x y z type
  double4 *xhold = nullptr;                      // positions at the last build (same order)
  float4 *posf = nullptr;                        // FP32 copy of the positions at the last reneighbor (list-build distance test)
  double *v[3] = {nullptr, nullptr, nullptr}, *v_tmp[3] = {nullptr, nullptr, nullptr};
  double *f[3] = {nullptr, nullptr, nullptr};
  int *tag = nullptr, *tag_tmp = nullptr;
  int *img = nullptr, *img_tmp = nullptr;        // [3][npad]
  // ---- tag order ----
  int *map = nullptr;                            // [maxtag+2] tag -> p
  int *type_t = nullptr;                         // [maxtag+2]
  int *crank = nullptr;                          // [maxtag+2] canonical (reference local) index
  int *num_bond = nullptr, *bond_type = nullptr, *bond_atom = nullptr;   // [(maxtag+2)], [*bpa]
  int *nspecial = nullptr, *special = nullptr;                            // [*3], [*maxspecial]
  // angles by tag (apa = 0: none): every atom holds a copy of each angle it is part of, atoms as IDs
  int apa = 0;
  int *num_angle = nullptr, *angle_type = nullptr, *angle_a1 = nullptr, *angle_a2 = nullptr, *angle_a3 = nullptr;
  double *partial_a = nullptr;     // [nblocks][8] angle energy + virial block sums
  int lg_bit = 1;                  // the group bit of fix langevin (DeviceState::gmask)
  bool lg_grouped = false;         // fix langevin acts on a group: its draws go by lgrank (rank among the members), not crank
  double *lgsum = nullptr;         // fix langevin `zero yes`: [nred_blocks + 1][16] block sums of the random forces + their mean
  // the angle LIST of the last reneighbor as every bead sees it (NTopoAngleAll::build, src/ntopo_angle_all.cpp:37-93): an
  // angle acts - on all three of its atoms - iff the one with the lowest local index holds a copy of it, whatever copies
  // the other two hold (copies go out of step when fix ex_unload's influence rule spares one atom).  eff_*[t]: the listed
  // angles bead t is part of, sorted, so that the force kernel is a per-bead gather with a fixed summation order.
  int ecap = 0;
  int *eff_n = nullptr, *eff_rec = nullptr;      // [npad], int4 [ecap][npad] = (type, i1, i2, i3) with PHYSICAL indices, column-major
  // bond tables as of the last reneighbor = the reference's neighbor->bondlist, which the LE fixes loop over
  // even when another LE fix changed the topology earlier in the same step (fix_ex_unload.cpp:223, fix_extrusion.cpp:368)
  int *num_bond0 = nullptr, *bond_type0 = nullptr, *bond_atom0 = nullptr;
  int le_snapshot = 0;             // keep the snapshot (set when an LE fix exists)
  bool topo_dirty = true;          // bond tables changed since the snapshot was taken
  // one record per tag {num_bond, (type << 26) | partner tag, ..} for the rebuild's bond-partner table: ONE line per bead
  // instead of three (num_bond, bond_type row, bond_atom row) when tag order is not memory order; refreshed when dirty
  int *bond_pack = nullptr;
  int bond_pack_stride = 0;
  bool bond_pack_dirty = true;
  // the first ANGLE_PACK_COLS stored angles of every atom as (type, a1, a2, a3) records, column-major by tag: what the angle
  // listing of every rebuild reads instead of three tables of stride apa (rebuilt after the angle tables changed)
  int *angle_pack = nullptr;
  bool angle_pack_dirty = true;
  // the same records by PHYSICAL index (one GPU, permute pass writes the bond-partner table): moved with the beads by
  // k_permute, so that the table pass reads them next to the bead's other data instead of gathering them by tag
  int *bond_pack_p[2] = {nullptr, nullptr};
  bool bond_pack_p_valid = false;
  // ---- cells / neighbor list ----
  int ncell[3] = {0, 0, 0}, ncells = 0;
  int row_tile = 0;          // (y, z) rows of cells numbered in tiles of this edge (bin_inl.h row_id); 0 = z-major (decomposed runs)
  double cellinv[3] = {0, 0, 0};
  int *cell_of = nullptr, *cell_count = nullptr, *cell_start = nullptr, *cell_fill = nullptr;
  int *scan_tmp = nullptr, *perm = nullptr;
  int maxneigh = 0;
  int *neigh = nullptr;      // [maxneigh][npad] full list, special bits in the top 2 bits
  int *numneigh = nullptr;   // [npad]
  int *bpart = nullptr;      // [bpa][npad] (type << 26) | partner p ; -1 = none
  // [npad] frozen periodic image of each bond partner, BSHIFT_BITS per COMPACTED bond slot (= the bond's row in the bead's
  // list), written by k_bond_table from the positions of the build; 0 for nearly every bead (engine.h NN_SHIFTED_BIT)
  unsigned long long *bshift = nullptr;
  // 1: the minimum image of every step IS the frozen image, provably - every bond style is fene and 2 R0 < half the box,
  // so a bond component cannot reach half the box without `Bad FENE bond` ending the run at that very step (rlogarg <= -3
  // <=> r >= 2 R0); the force kernels then keep their three compares per bond and the rebuild skips the partner gathers
  // (k_bond_table: 25 instead of 56 us per rebuild at 1M beads).  0: images are frozen at the rebuild (bshift).
  int bond_minimg = 0;
  double *pairtab = nullptr; // 6 * nt*nt : cutsq lj1 lj2 lj3 lj4 offset
  int newton_pair = 0;           // `newton on [off]`: which end stores a pair in the reference's half list (ex_load's visit order)
  int ref_nbin[3] = {1, 1, 1};   // the reference's neighbor bins (cutneighmax / 2 fitted to the box, nbin_standard.cpp:53-186)
  double ref_bininv[3] = {1, 1, 1};
  bool ident_order = true;       // the reference's local index of a bead = its ID - 1 (no Atom::sort, data file in ID order)
  int sflag[4] = {1, 1, 1, 1};   // Engine::special_flag per level: 0 dropped from the list, 1 ordinary entry, 2 entry with level bits
  int pair_uniform = 0;      // every type pair has the same coefficients
  double pair_u[6] = {0, 0, 0, 0, 0, 0};
  double cutneigh = 0.0;
  // ---- thermo partial sums ----
  int nred_blocks = 0;
  double *partial = nullptr;     // [nblocks][16]
  double *partial_h = nullptr;   // pinned
  // ---- flags ----
  int *flags = nullptr;          // device
  int *flags_h = nullptr;        // pinned, mapped host copy
  int *flags_h_dev = nullptr;    // device-side address of flags_h
  bool cell_count_dirty = false; // cell_count holds counts that no scan has consumed (and zeroed) yet
  bool bins_ready = false;       // cell_of / cell_count / arrival ranks hold the bins of the CURRENT positions (written by k_step)
  int flags_seq = 0;             // sequence number of the last publish (flags_h[NFLAGS] echoes it)
  // ---- Langevin RNG (block-parallel RanMars) ----
  int rng_B = 0, rng_nblocks = 0;
  uint32_t *rng_state = nullptr;   // [nblocks][97]
  uint32_t *rng_jump = nullptr;    // coefficients of the jump polynomial(s): [97] block mode, [2][97] batch mode (full / last segment)
  uint32_t *rng_out = nullptr;     // [3N] 24-bit draws of the current call, canonical order
  uint32_t *rng_buf[2] = {nullptr, nullptr};   // double buffer: the next call's draws are generated on rng_stream
  int rng_cur = 0;
  hipStream_t rng_stream = nullptr;
  hipEvent_t rng_done[2] = {nullptr, nullptr}, rng_consumed[2] = {nullptr, nullptr};
  bool rng_ahead = false;          // rng_buf[rng_cur ^ 1] already holds the draws of the next call
  // batch generator (default): W whole calls per launch, one wavefront each, two pools alternating
  int rng_mode = 1;                // 1 = batch (k_rng_calls), 0 = block-parallel per call (k_rng_langevin)
  int rng_W = 0;
  long long rng_total = 0;         // draws per call = 3 * beads in the system
  int rng_nseg = 1;                // batch generator: segments per call (one wavefront each) and their length
  long long rng_seglen = 0;
  uint32_t *rng_wstate = nullptr;  // [3][W * S][97] windows in front of each (call, segment) of a batch: batch b reads copy b % 3, writes (b + 1) % 3
  long long rng_batch_no[2] = {0, 0};          // number of the batch each pool holds (selects the window copy it started from)
  // decomposed runs: a rank generates only the stream segments that hold draws of beads it owns or ghosts (kernels_rng.hip)
  int *rng_need = nullptr;         // [S] segments wanted by the batch about to be launched
  int *rng_gen[2] = {nullptr, nullptr};        // [S] segments each pool holds
  int *rng_late = nullptr;         // [2][S] segments the validation found missing (generated late, from the kept windows)
  bool rng_skip = false;           // segment skipping is on (decomposed, batch mode)
  uint32_t *rng_pool[2] = {nullptr, nullptr};   // [W][3N]
  uint64_t rng_batch_raw[2] = {0, 0};           // raw index of the first draw held by each pool (0 = empty)
  int rng_pool_cur = 0;
  RanMarsInt rng_origin;           // host generator at the position given to rng_langevin_setup
  // ---- LE fixes (tag order) ----
  double4 *xt = nullptr;           // [maxtag+2] stored coordinates by tag
  int *le_i[16] = {nullptr};       // integer scratch arrays [maxtag+2]
  double *le_d[2] = {nullptr};     // double scratch [maxtag+2]
  unsigned long long *le_bits = nullptr;
  uint32_t *le_rng_state = nullptr;   // [LE_MAX_FIXES][100]: w[97], n_lo, n_hi per LE fix slot
  uint32_t *le_draws = nullptr;       // [maxtag+2]
  int *le_list = nullptr;             // extruder listings [4][maxtag+2]
  int *le_scan = nullptr;
  // ---- spatial decomposition (z slabs; world > 1) ----
  int dd = 0;                      // 1 = this rank owns a z slab, ghosts at [n, n + nghost)
  int nghost = 0;
  int ntotal = 0;                  // beads in the whole system
  double slab_lo = 0.0, slab_hi = 0.0, cutghost = 0.0, zlo_ext = 0.0;
  int *gcell_start = nullptr, *gcell_count = nullptr;   // ghost ranges per cell (relative to n)
  int *sendlist[2] = {nullptr, nullptr};                // owned indices sent down / up every step
  int *sendlist_alt[2] = {nullptr, nullptr};            // ... and the buffers the next rebuild writes its reordered lists into
  bool bpart_fresh = false;                             // the permute pass of this rebuild has written the bond-partner table
  int *gmask = nullptr;                                 // [maxtag+2] group bits by tag (bit 0 = all); only when a fix acts on a group
  int *lgrank = nullptr;                                // [maxtag+2] rank of a bead among the members of fix langevin's group (local order)
  void *angtab_dev = nullptr;                           // AngleTable in device memory (fused angle step)
  bool ghost_whole_shell = false;                       // every bead within the ghost cutoff of a face is sent (runs with an angle style)
  bool map_stale = true;                                // map[] was not left by a decomposed rebuild: fill it before the next one
  int nsend[2] = {0, 0}, nrecv[2] = {0, 0};
  double4 *sendbuf = nullptr, *recvbuf = nullptr;       // halo staging
  int *gone = nullptr;                                  // [npad] 1 = this bead has just left for another slab
  // halo / compute overlap: beads that are sent to a neighbour or read a ghost form phase 1 of a step, the rest
  // (phase 0) is computed while the ghost positions of the next step travel on comm_stream
  unsigned char *phase = nullptr;                       // [npad]
  int *sendslot = nullptr;                              // [npad] slot of a border bead in its send list (bit 30 = upper list), -1 = none
  bool sendslot_fallback = false;                       // some bead sits in both send lists: pack with the kernel instead
  bool direct_recv = false;                             // send lists are in the receiver's sorted ghost order: a halo lands in place
  bool packed_ahead = false;                            // the step kernel already wrote the border beads' new positions into sendbuf
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_phase1 = nullptr, ev_halo = nullptr;
  bool halo_inflight = false;                           // ev_halo guards ghost slots that comm_stream is filling
  bool halo_ahead = false;                              // ghosts of the coming step were already exchanged
  // Peer windows (kernels_dd.hip "fast halo"): the per-step halo as a side effect of the step kernel.  Every rank owns a
  // window [parity 2][side 2: from below, from above][halo_cap] of positions plus two arrival counters; the neighbours
  // map it (hipIpcOpenMemHandle, or the plain pointer when the ranks are threads of one process) and their step kernels
  // store the new positions of their border beads straight into it, in this rank's sorted ghost order.
  double4 *halo_win = nullptr;
  unsigned *halo_flag = nullptr;                        // [2] arrival counters: [0] written by the rank below, [1] by the rank above
  bool halo_fused = false;                              // counters + window copy in one launch (every neighbour is another GPU)
  size_t halo_cap = 0;
  double4 *peer_win[2] = {nullptr, nullptr};            // the windows of the rank below / above
  unsigned *peer_flag[2] = {nullptr, nullptr};          // ... and the counter of theirs that is mine to write
  void *peer_base[2] = {nullptr, nullptr};              // what hipIpcOpenMemHandle returned (to close), nullptr for plain pointers
  bool fast_halo = false;                               // windows mapped AND switched on for this run (Engine::run re-reads the switch)
  bool halo_mapped = false, halo_verify = false;
  unsigned halo_seq = 0;                                // completed window exchanges (same on every rank)
  int packed_peer = 0;                                  // the last step kernel stored into the neighbours' windows (parity + 1)
  double halo_timeout_s = 120.0;
  int *gdest = nullptr;                                 // arrival order -> sorted ghost slot
  int *gtag_in = nullptr;                               // ghost tags in arrival order
  double *migbuf[2] = {nullptr, nullptr}, *migin = nullptr;   // migrating beads (MIG_W doubles each)
  double4 *xht = nullptr;                               // [maxtag+2] xhold by tag (LE fixes)
  double *gather_send = nullptr, *gather_recv = nullptr;      // whole-system gathers
  size_t gather_cap = 0;
  void *sort_scratch = nullptr;   // kernels_sort.hip: key / value / temp buffers of the Atom::sort emulation
  Comm *comm_watch = nullptr;     // decomposed runs: host waits on the stream go through Comm::wait_stream (time-out + abort)
  // ---- kernel timing (HIP events on the launch stream) ----
  std::vector<hipEvent_t> ev0, ev1;
  size_t ev_used = 0;
};

// ------------------------------- launchers (kernels_*.hip) -------------------------------------
void dev_alloc(DeviceState &d, int n, int maxtag, int ntypes, int bpa, int maxspecial, const Box &box,
               double cutneigh);
void dev_free(DeviceState &d);
void dev_alloc_neigh(DeviceState &d, int maxneigh);

// integrate (kernels_md.hip)
void launch_initial_integrate(DeviceState &d, const TypeTables &tt, double dtv, double triggersq, bool check, int groupbit = 1);
void launch_force(DeviceState &d, const BondTable &bt, const double special_lj[4], bool eflag, bool has_pair, int parts = 3);
void launch_flevel_copy(DeviceState &d, double *flevel, bool to_level, bool add);
void launch_step(DeviceState &d, const BondTable &bt, const double special_lj[4], const TypeTables &tt, bool langevin,
                 bool next, bool ident, bool has_pair, double dtv, double triggersq, bool check, hipEvent_t ev_start = nullptr,
                 hipEvent_t ev_stop = nullptr, int which = -1, bool swap_buffers = true, bool angle_forces = false,
                 bool eflag = false, int nvebit = 1, int lgbit = 1);    // group bits != 1: fix nve / fix langevin on a group (GRP variant)
bool step_fuses_groups(const DeviceState &d, bool has_pair, bool angles);   // fix nve / fix langevin on groups inside the step kernel
bool step_fuses_energy(const DeviceState &d, bool has_pair);   // a thermo step can be one launch of the step kernel's energy variant
void launch_langevin(DeviceState &d, const TypeTables &tt, bool identity_rank, bool fuse_final, int groupbit = 1);
// `zero yes`: after launch_langevin and before the draws are released - the members' mean random force off every member
void launch_langevin_zero(DeviceState &d, const TypeTables &tt, bool identity_rank, int groupbit, long members,
                          double *host_sum3 = nullptr);
void launch_langevin_zero_apply(DeviceState &d, int groupbit, const double *mean3);
void launch_final_integrate(DeviceState &d, const TypeTables &tt, int groupbit = 1);
void launch_ke(DeviceState &d, const TypeTables &tt);
// sum(m v_i v_j) over the owned beads, order xx yy zz xy xz yz (the kinetic part of the pressure tensor); synchronous
void ke_tensor(DeviceState &d, const TypeTables &tt, double *out6);
// angle forces added to f (after launch_force); eflag: energy / virial thirds into partial_a (reduce_angle_partials)
void launch_angle(DeviceState &d, const AngleTable &at, bool eflag, bool overwrite = false);
void launch_angle_list(DeviceState &d);       // at every reneighbor of a run with an angle style
void upload_angle_table(DeviceState &d, const AngleTable &at);   // before the first fused step of a run with an angle style
bool step_fuses_angles(const DeviceState &d, bool has_pair);
void reduce_angle_partials(DeviceState &d, double *out8);
// reductions: returns sums of `partial` columns on the host (synchronises the stream)
void reduce_partials(DeviceState &d, double *out16);

// neighbor (kernels_neigh.hip)
void launch_reneighbor(DeviceState &d, double cutneighsq, const double special_lj[4], bool has_pair);
// (`binned`: cell, arrival order and counts of the m_in slots are in place - the decomposed rebuild's migration pass)
void launch_sort_owned(DeviceState &d, int m_in = -1, int n_out = -1, const int *gone = nullptr, bool binned = false);
void scan_cells(DeviceState &d, int *count, int *start, int nc, int total);   // start[0..nc] := exclusive scan of count[0..nc), count := 0
void launch_lists(DeviceState &d, double cutneighsq, const double special_lj[4], bool has_pair);

// Atom::sort emulation (kernels_sort.hip): crank[tag] := rank in the reference's sorted local order
void launch_atom_sort(DeviceState &d, const int nb[3], const double binv[3], bool by_tag = false);
void sort_scratch_free(DeviceState &d);

// rng (kernels_rng.hip)
void rng_langevin_setup(DeviceState &d, RanMarsInt &host_rng, int natoms);
void launch_rng_langevin(DeviceState &d, uint64_t first_raw);
void rng_langevin_consumed(DeviceState &d);
// decomposed runs, after every change of ownership or of the canonical ranks (rebuild with migration, Atom::sort): do the
// pools hold the draws of every owned bead?  Enqueues the check; rng_late_generate() (after the flags reached the host)
// produces what was missing before the next step consumes it
void rng_validate_owned(DeviceState &d);
// the same check as kernel arguments, for a kernel that walks the owned beads anyway (late == nullptr: nothing to check)
struct RngValidateArgs {
  const int *crank;
  long long seglen;
  int nseg;
  const int *gen0, *gen1;
  int live0, live1;
  int *late;
};
RngValidateArgs rng_validate_args(DeviceState &d);
__device__ __forceinline__ void rng_validate_bead(const RngValidateArgs &V, int t, int *__restrict__ flags) {
  const long long r = V.crank ? V.crank[t] : t - 1;
  const int sg = (int)((3 * r) / V.seglen);
  if (V.live0 && !V.gen0[sg]) { V.late[sg] = 1; flags[FLAG_RNG_MISS] = 1; }
  if (V.live1 && !V.gen1[sg]) { V.late[V.nseg + sg] = 1; flags[FLAG_RNG_MISS] = 1; }
}
void rng_late_generate(DeviceState &d);
int rng_segments_held(DeviceState &d);
void launch_ranmars_gen(DeviceState &d, int slot, const int *count_ptr, uint32_t *out, int maxout);

// LE fixes (kernels_le.hip)
struct ExLoadParams {
  int iatomtype, jatomtype, imaxbond, inewtype, jmaxbond, jnewtype, btype;
  double cutsq, fraction;
  int atype;        // angle type created around every new bond (0: none)
  int groupbit = 1; // both atoms of a candidate pair must be in the fix's group (fix_ex_load.cpp:435,450)
};
struct ExUnloadParams {
  int btype;
  double cutsq, fraction;
  int angleflag;    // angles exist: a broken bond takes the angles it is part of with it (fix_ex_unload.cpp:149-152)
  int groupbit = 1; // both ends of a bond must be in the fix's group (fix_ex_unload.cpp:228-229)
};
struct ExtrusionParams {
  int neutral, ctcf_left, ctcf_right, ctcf_lr, btype;
  double through_prob;
  int groupbit = 1; // both ends of an extruder bond must be in the fix's group (fix_extrusion.cpp:373-376)
};
void le_rng_upload(DeviceState &d, int slot, const RanMarsInt &r);
void le_rng_download(DeviceState &d, int slot, RanMarsInt &r);
// each returns after enqueueing; counters are read back by the caller through flags_h
void launch_topo_snapshot(DeviceState &d);   // num_bond0 / bond_type0 / bond_atom0 := current bond tables
struct Comm;
void launch_ex_load(DeviceState &d, const ExLoadParams &p, int rng_slot, Comm *comm);
// stock fix bond/create: `bondcount` (host, by tag, nt ints) goes up before the launch; bond_create_counts fetches it back
struct Comm;
void launch_bond_create(DeviceState &d, const ExLoadParams &p, int rng_slot, const int *bondcount, int nt, Comm *comm);
void bond_create_counts(DeviceState &d, int *bondcount, int nt);
void launch_ex_unload(DeviceState &d, const ExUnloadParams &p, int rng_slot);
void launch_extrusion(DeviceState &d, const ExtrusionParams &p, int rng_slot);
void stream_sync(DeviceState &d);    // wait for d.stream; with ranks: bounded (Comm::wait_stream)
void sync_flags(DeviceState &d, unsigned reset_mask = 0);   // copy flags to flags_h, zero the masked ones, wait
void publish_flags(DeviceState &d, unsigned reset_mask = 0);   // the same without waiting ...
void wait_flags(DeviceState &d);                               // ... and the wait for the last publish
void scan_exclusive(DeviceState &d, const int *in, int *out, int m, int total_flag);
void dd_alloc(DeviceState &d, int world);
// Decomposed runs, canonical visit order (local index = ID - 1, newton_pair off): the LE fixes work from owner-computed
// bits and O(extruders) position records instead of an all-gather of every bead's (tag, x, xhold) (kernels_dd.hip
// dd_gather_needed, kernels_le.hip launch_ex_load).  Other visit orders keep the whole-system gather.
inline bool dd_le_fast(const DeviceState &d) { return d.dd && d.ident_order && !d.newton_pair && !getenv("LAMMPS_LE_DD_FULL_GATHER"); }

}  // namespace lmp_le
