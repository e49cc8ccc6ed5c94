// kernels_le.hip — the USER-LE fixes (extruder load / step / unload) as data-parallel HIP kernels.
//
// The reference implements each fix as serial loops whose result depends on visit order
// (src/USER-LE/fix_ex_load.cpp:329-655, fix_extrusion.cpp:256-872, fix_ex_unload.cpp:172-372).
// Here every phase is a kernel over beads (tag order, index = atom ID) or over extruder listings,
// re-derived so that the outcome is IDENTICAL to the serial loops at the canonical configuration
// (1 rank, newton off, atom_modify sort 0 0: local index = ID - 1; SURVEY §8a "order dependence"):
//
//  ex_load   accepted(a) = base(a) AND NOT accepted(a-1) for pair a = (a, a+2)  (the partner[mid] test,
//            fix_ex_load.cpp:471-476,484) -> parity of the run of set bits below a in a ballot bitmask;
//            closest-wins partner; k-th bead with a partner gets the k-th RanMars draw (prefix sum).
//  extrusion listings (bond-list entries, double for bonds that straddle a periodic face,
//            ntopo_bond_all.cpp:66-67) in order; RNG draws by prefix sum of the per-listing draw counts
//            (barrier beads only, short-circuit order L then R, fix_extrusion.cpp:406-429); the
//            closest-wins / first-wins writes of phase 1 resolved per target bead from <= 4 claim events;
//            phases 2-4 per bead / per extruder (they only touch their own extruder's beads).
//  ex_unload farthest-wins partner over own bonds with the image frozen at the last reneighbor.
//  special lists: rebuild_special_one / dedup exactly as fix_extrusion.cpp:1045-1135, O(events).
#include "device.h"
#include "comm.h"

namespace lmp_le {

constexpr int BLOCK = 256;
constexpr int SCAN_BLOCK = 1024;
#define BIGD 1.0e20

// integer scratch slots (tag-indexed arrays of length T+2)
enum { I_BC = 0, I_A, I_B, I_C, I_D, I_E, I_F, I_G, I_H, I_J, I_K, I_L, I_M, I_N, I_O, I_P };

struct Topo {   // tag-indexed topology views
  int T, bpa, ms;
  int *num_bond, *bond_type, *bond_atom, *nspecial, *special, *type_t;
  const int *num_bond0, *bond_type0, *bond_atom0;   // bond tables at the last reneighbor (= neighbor->bondlist)
  int apa;                                           // angles per atom (0: no angle storage)
  int *num_angle, *angle_type, *angle_a1, *angle_a2, *angle_a3;
  const int *gmask;                                  // the calling fix's group: bits by tag (nullptr: group all)
  int gbit;
};
static Topo topo_of(DeviceState &d, int groupbit = 1) {
  return Topo{d.maxtag, d.bpa, d.maxspecial, d.num_bond, d.bond_type, d.bond_atom, d.nspecial, d.special, d.type_t,
              d.num_bond0, d.bond_type0, d.bond_atom0, d.apa, d.num_angle, d.angle_type, d.angle_a1, d.angle_a2, d.angle_a3,
              groupbit != 1 ? d.gmask : (const int *)nullptr, groupbit};
}
__device__ __forceinline__ bool in_group(const Topo &tp, int t) { return !tp.gmask || (tp.gmask[t] & tp.gbit); }   // mask[i] & groupbit

// ------------------------------------------------------------------------------------------
// exclusive scan of in[0..m) -> out[0..m], out[m] = total, also written to *total_slot
__global__ __launch_bounds__(SCAN_BLOCK) void k_lscan_local(int m, const int *__restrict__ in, int *__restrict__ out,
                                                            int *__restrict__ blocksum) {
  __shared__ int s[SCAN_BLOCK];
  int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  int v = (i < m) ? in[i] : 0;
  s[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
    int t = (threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
    __syncthreads();
    s[threadIdx.x] += t;
    __syncthreads();
  }
  if (i < m) out[i] = s[threadIdx.x] - v;
  if (threadIdx.x == SCAN_BLOCK - 1) blocksum[blockIdx.x] = s[threadIdx.x];
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_lscan_sums(int nb, int *__restrict__ blocksum, int *__restrict__ total_slot) {
  __shared__ int s[SCAN_BLOCK];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += SCAN_BLOCK) {
    int i = base + threadIdx.x;
    int v = (i < nb) ? blocksum[i] : 0;
    s[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
      int t = (threadIdx.x >= off) ? s[threadIdx.x - off] : 0;
      __syncthreads();
      s[threadIdx.x] += t;
      __syncthreads();
    }
    int c = carry;
    if (i < nb) blocksum[i] = c + s[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == SCAN_BLOCK - 1) carry = c + s[threadIdx.x];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_slot = carry;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_lscan_add(int m, int *__restrict__ out, const int *__restrict__ blocksum,
                                                          const int *__restrict__ total_slot) {
  int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i < m) out[i] += blocksum[blockIdx.x];
}
void scan_exclusive(DeviceState &d, const int *in, int *out, int m, int total_flag);
static void scan_ex(DeviceState &d, const int *in, int *out, int m, int total_flag) { scan_exclusive(d, in, out, m, total_flag); }
void scan_exclusive(DeviceState &d, const int *in, int *out, int m, int total_flag) {
  int sb = (m + SCAN_BLOCK - 1) / SCAN_BLOCK;
  int *tmp = d.le_scan;
  hipLaunchKernelGGL(k_lscan_local, dim3(sb), dim3(SCAN_BLOCK), 0, d.stream, m, in, out, tmp);
  hipLaunchKernelGGL(k_lscan_sums, dim3(1), dim3(SCAN_BLOCK), 0, d.stream, sb, tmp, d.flags + total_flag);
  hipLaunchKernelGGL(k_lscan_add, dim3(sb), dim3(SCAN_BLOCK), 0, d.stream, m, out, tmp, d.flags + total_flag);
}

// exclusive prefix count of flag[t] over the beads in the reference's LOCAL order (crank[t], 0-based): out[t] = number of
// flagged beads with a smaller local index.  That is the order in which the LE fixes hand out RNG draws
// (`for (i = 0; i < nlocal; i++) if (partner[i]) probability[i] = random->uniform()`, fix_ex_load.cpp:517-520) and in which
// NTopoBondAll lists bonds.  With local index = ID - 1 (no Atom::sort, data file in ID order) it is the plain scan by tag.
__global__ __launch_bounds__(BLOCK) void k_by_rank_scatter(int T, const int *__restrict__ crank, const int *__restrict__ flag,
                                                           int *__restrict__ byrank) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t == 0 || t == T + 1) { if (t == 0) byrank[T] = byrank[T + 1] = 0; return; }
  if (t <= T) byrank[crank[t]] = flag[t];
}
__global__ __launch_bounds__(BLOCK) void k_by_rank_gather(int T, const int *__restrict__ crank, const int *__restrict__ scanned,
                                                          int *__restrict__ out) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > T + 1) return;
  out[t] = (t >= 1 && t <= T) ? scanned[crank[t]] : 0;
}
static void scan_local_order(DeviceState &d, const int *flag, int *out, int total_flag, int *tmp_a, int *tmp_b) {
  const int T = d.maxtag, nt = T + 2, nb = (nt + BLOCK - 1) / BLOCK;
  if (d.ident_order) { scan_exclusive(d, flag, out, nt, total_flag); return; }
  hipLaunchKernelGGL(k_by_rank_scatter, dim3(nb), dim3(BLOCK), 0, d.stream, T, d.crank, flag, tmp_a);
  scan_exclusive(d, tmp_a, tmp_b, nt, total_flag);
  hipLaunchKernelGGL(k_by_rank_gather, dim3(nb), dim3(BLOCK), 0, d.stream, T, d.crank, tmp_b, out);
}

// ------------------------------------------------------------------------------------------
// special-list / bond-table edits on one bead (device versions of the reference's in-place code)
__device__ __forceinline__ void dev_delete_bond(const Topo &tp, int i, int partner) {   // fix_extrusion.cpp:656-668
  int b = tp.bpa, nb = tp.num_bond[i];
  for (int m = 0; m < nb; m++)
    if (tp.bond_atom[(size_t)i * b + m] == partner) {
      for (int k = m; k < nb - 1; k++) {
        tp.bond_atom[(size_t)i * b + k] = tp.bond_atom[(size_t)i * b + k + 1];
        tp.bond_type[(size_t)i * b + k] = tp.bond_type[(size_t)i * b + k + 1];
      }
      tp.num_bond[i] = nb - 1;
      break;
    }
}
// NOTE (reference behaviour, reproduced): when `partner` is NOT in the 1-2 block — an unload acting on a bond-list
// entry that an extrusion moved earlier in the same step — the loop below still drops the entry at index n1 and
// decrements all three counters, which can push a genuine 1-2 partner out of the 1-2 block of THIS bead only.
// The reference's half neighbor list then takes the special status of a pair from the lower-index bead's list;
// the flag tells the list build to do the same (k_build_neigh ASYM path).
__device__ __forceinline__ void dev_special_remove12(const Topo &tp, int i, int partner, int *__restrict__ flags) {   // :673-683
  int *slist = tp.special + (size_t)i * tp.ms;
  int n1 = tp.nspecial[3 * (size_t)i], n3 = tp.nspecial[3 * (size_t)i + 2], m;
  for (m = 0; m < n1; m++) if (slist[m] == partner) break;
  if (m == n1) flags[FLAG_SPECIAL_ASYM] = 1;
  for (; m < n3 - 1; m++) slist[m] = slist[m + 1];
  tp.nspecial[3 * (size_t)i]--; tp.nspecial[3 * (size_t)i + 1]--; tp.nspecial[3 * (size_t)i + 2]--;
}
__device__ __forceinline__ bool dev_special_insert12(const Topo &tp, int i, int partner) {   // :748-771
  int *slist = tp.special + (size_t)i * tp.ms;
  int n1 = tp.nspecial[3 * (size_t)i], n2 = tp.nspecial[3 * (size_t)i + 1], n3 = tp.nspecial[3 * (size_t)i + 2], m, n;
  for (m = n1; m < n3; m++) if (slist[m] == partner) break;
  if (m < n3) {
    for (n = m; n < n3 - 1; n++) slist[n] = slist[n + 1];
    n3--;
    if (m < n2) n2--;
  }
  if (n3 == tp.ms) return false;
  for (m = n3; m > n1; m--) slist[m] = slist[m - 1];
  slist[n1] = partner;
  tp.nspecial[3 * (size_t)i] = n1 + 1; tp.nspecial[3 * (size_t)i + 1] = n2 + 1; tp.nspecial[3 * (size_t)i + 2] = n3 + 1;
  return true;
}
__device__ int dev_dedup(int nstart, int nstop, int *copy) {   // fix_extrusion.cpp:1116-1135
  int i, m = nstart;
  while (m < nstop) {
    for (i = 0; i < m; i++)
      if (copy[i] == copy[m]) { copy[m] = copy[nstop - 1]; nstop--; break; }
    if (i == m) m++;
  }
  return nstop;
}
// rebuild_special_one (fix_extrusion.cpp:1045-1108): 1-2 block kept; 1-3 / 1-4 from the 1-2 lists of others
__device__ void dev_rebuild_special_one(const Topo &tp, int m, int *__restrict__ flags) {
  int copy[MS_MAX * MS_MAX + MS_MAX];
  const int ms = tp.ms;
  int *slist = tp.special + (size_t)m * ms;
  int n1 = tp.nspecial[3 * (size_t)m], cn1 = 0, cn2, cn3;
  for (int i = 0; i < n1; i++) copy[cn1++] = slist[i];
  cn2 = cn1;
  for (int i = 0; i < cn1; i++) {
    int n = copy[i];
    const int *sl = tp.special + (size_t)n * ms;
    int nn1 = tp.nspecial[3 * (size_t)n];
    for (int j = 0; j < nn1; j++) if (sl[j] != m) copy[cn2++] = sl[j];
  }
  cn2 = dev_dedup(cn1, cn2, copy);
  if (cn2 > ms) { flags[FLAG_ERROR] = ERR_SPECIAL_SCRATCH; return; }
  cn3 = cn2;
  for (int i = cn1; i < cn2; i++) {
    int n = copy[i];
    const int *sl = tp.special + (size_t)n * ms;
    int nn1 = tp.nspecial[3 * (size_t)n];
    for (int j = 0; j < nn1; j++) if (sl[j] != m) copy[cn3++] = sl[j];
  }
  cn3 = dev_dedup(cn2, cn3, copy);
  if (cn3 > ms) { flags[FLAG_ERROR] = ERR_SPECIAL_SCRATCH; return; }
  // the 1-2 block is unchanged: only the 1-3 / 1-4 blocks are rewritten, so concurrent rebuilds of
  // neighbours (which read only 1-2 blocks) see consistent data.  Its COUNT is stored as the reference stores it
  // (nspecial[m][0] = cn1, fix_extrusion.cpp:1104): a count that the unconditional decrements of a bond removal have
  // pushed below zero (a bead whose 1-2 block had already lost that partner) comes back as 0 - for a concurrent reader
  // -1 and 0 are the same empty block.  Found by the wide fuzz sweep (seeds 184, 215: profiles/r03/fuzz_wide.log).
  if (n1 != cn1) tp.nspecial[3 * (size_t)m] = cn1;
  tp.nspecial[3 * (size_t)m + 1] = cn2;
  tp.nspecial[3 * (size_t)m + 2] = cn3;
  for (int i = cn1; i < cn3; i++) slist[i] = copy[i];
}
// FixExUnload::break_angles (fix_ex_unload.cpp:551-582) for every broken bond (id1, id2 = fin[id1]) that influences atom i:
// an angle copy goes if one of its two bonds is that bond.  The reference walks the broken bonds in list order and
// deletes by shifting; the copies that remain, and their order, do not depend on that order.  `fin[t]` = the partner t lost
// its bond to in this firing (each bead loses at most one bond per firing), 0 otherwise.
__device__ void dev_break_angles(const Topo &tp, int i, const int *__restrict__ fin, int *__restrict__ flags) {
  const int *sl = tp.special + (size_t)i * tp.ms;
  const int n3 = tp.nspecial[3 * (size_t)i + 2];
  auto influences = [&](int id1, int id2) {           // fix_ex_unload.cpp:445-455: the atom is one end, or lists both ends
    if (i == id1 || i == id2) return true;
    int found = 0;
    for (int k = 0; k < n3; k++) if (sl[k] == id1 || sl[k] == id2) found++;
    return found == 2;
  };
  int num = tp.num_angle[i], w = 0, removed = 0;
  int *at = tp.angle_type + (size_t)i * tp.apa, *a1 = tp.angle_a1 + (size_t)i * tp.apa, *a2 = tp.angle_a2 + (size_t)i * tp.apa,
      *a3 = tp.angle_a3 + (size_t)i * tp.apa;
  for (int m = 0; m < num; m++) {
    const int t1 = a1[m], t2 = a2[m], t3 = a3[m];
    const bool gone = (fin[t1] == t2 && influences(t1, t2)) || (fin[t2] == t3 && influences(t2, t3));
    if (gone) { removed++; continue; }
    if (w != m) { at[w] = at[m]; a1[w] = t1; a2[w] = t2; a3[w] = t3; }
    w++;
  }
  tp.num_angle[i] = w;
  if (removed) atomicAdd(&flags[FLAG_COUNT_B], removed);
}
// FixExLoad::create_angles (fix_ex_load.cpp:855-954, newton_bond off) for atom m.  `fin[t]` = the partner of t's new bond
// of this firing (0: none): bond (a, b) is new iff fin[a] == b.  Reads only 1-2 blocks of the special lists, which the
// concurrent rebuilds of other atoms do not touch.
__device__ void dev_create_angles(const Topo &tp, int m, int atype, const int *__restrict__ fin, int *__restrict__ flags) {
  const int ms = tp.ms, apa = tp.apa;
  int num = tp.num_angle[m], made = 0;
  int *at = tp.angle_type + (size_t)m * apa, *a1 = tp.angle_a1 + (size_t)m * apa, *a2 = tp.angle_a2 + (size_t)m * apa,
      *a3 = tp.angle_a3 + (size_t)m * apa;
  bool overflow = false;
  auto add = [&](int i1, int i2, int i3) {
    if (!(fin[i1] == i2 || fin[i2] == i3)) return;     // a new bond must be one of the angle's two bonds
    if (num < apa) { at[num] = atype; a1[num] = i1; a2[num] = i2; a3[num] = i3; num++; made++; }
    else overflow = true;
  };
  const int *s2 = tp.special + (size_t)m * ms;
  const int n2 = tp.nspecial[3 * (size_t)m];
  for (int i = 0; i < n2; i++)                          // atom m central: pairs of its 1-2 neighbours
    for (int j = i + 1; j < n2; j++) add(s2[i], m, s2[j]);
  for (int i = 0; i < n2; i++) {                        // atom m as atom 1 of the angle
    const int i2 = s2[i];
    const int *sl2 = tp.special + (size_t)i2 * ms;
    const int nn = tp.nspecial[3 * (size_t)i2];
    for (int j = 0; j < nn; j++) { const int i3 = sl2[j]; if (i3 != m) add(m, i2, i3); }
  }
  tp.num_angle[m] = num;
  if (made) atomicAdd(&flags[FLAG_COUNT_B], made);
  if (overflow) flags[FLAG_ERROR] = ERR_ANGLES;
}
// influence rules of update_topology: broken (fix_extrusion.cpp:940-969), created (:971-1001)
// `angles`: fix ex_unload / bond/break with angles in the system (extrusion leaves angles alone, fix_extrusion.cpp:924-1002)
__global__ __launch_bounds__(64) void k_topo_broken(Topo tp, const int *__restrict__ fin, int angles, int *__restrict__ flags) {
  int i = blockIdx.x * 64 + threadIdx.x + 1;
  if (i > tp.T) return;
  bool influenced = fin[i] != 0;
  if (!influenced) {
    const int *sl = tp.special + (size_t)i * tp.ms;
    int n = tp.nspecial[3 * (size_t)i + 2];
    for (int k = 0; k < n && !influenced; k++) {
      int p = fin[sl[k]];
      if (p)
        for (int q = 0; q < n; q++) if (sl[q] == p) { influenced = true; break; }
    }
  }
  if (influenced && angles && tp.apa > 0) dev_break_angles(tp, i, fin, flags);     // before the rebuild, as the reference
  if (influenced) dev_rebuild_special_one(tp, i, flags);
}
// `atype` > 0: fix ex_load / bond/create with an angle style defined - angles around the new bonds
__global__ __launch_bounds__(64) void k_topo_created(Topo tp, const int *__restrict__ fin, int atype, int *__restrict__ flags) {
  int i = blockIdx.x * 64 + threadIdx.x + 1;
  if (i > tp.T) return;
  bool influenced = fin[i] != 0;
  if (!influenced) {
    const int *sl = tp.special + (size_t)i * tp.ms;
    int n = tp.nspecial[3 * (size_t)i + 1];
    for (int k = 0; k < n; k++) if (fin[sl[k]]) { influenced = true; break; }
  }
  if (influenced) dev_rebuild_special_one(tp, i, flags);
  if (influenced && atype > 0 && tp.apa > 0) dev_create_angles(tp, i, atype, fin, flags);
}

// ------------------------------------------------------------------------------------------
// stored coordinates by tag + bondcount of `btype`
__global__ __launch_bounds__(BLOCK) void k_le_gather(Topo tp, const double4 *__restrict__ pos,
                                                     const double4 *__restrict__ xhold, const int *__restrict__ map,
                                                     int local, double4 *__restrict__ xt, double4 *__restrict__ xht,
                                                     int btype, int *__restrict__ bondcount, int check_multi,
                                                     int *__restrict__ flags) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > tp.T + 1) return;
  if (t == 0 || t == tp.T + 1) { bondcount[t] = 0; return; }
  // one rank: every bead is local.  Decomposed: xt / xht were filled by the all-gather (dd_gather_positions)
  if (local) { int p = map[t]; xt[t] = pos[p]; xht[t] = xhold[p]; }
  int bc = 0, nb = tp.num_bond[t];
  for (int m = 0; m < nb; m++) if (tp.bond_type[(size_t)t * tp.bpa + m] == btype) bc++;
  bondcount[t] = bc;
  if (check_multi && bc > 1) flags[FLAG_ERROR] = ERR_EXT_MULTI;
}
__device__ __forceinline__ double d2(const double4 &a, const double4 &b) {
  double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return dx * dx + dy * dy + dz * dz;
}
__global__ void k_zero_int(int n, int *a) { int i = blockIdx.x * BLOCK + threadIdx.x; if (i < n) a[i] = 0; }

// NTopoBond::build as the LE fixes see it: the bond tables of this reneighbor, kept until the next one.  One streaming
// kernel for the three tables (three blit copies of 4 + 12 + 12 MB took 255 us at 1M beads, this takes ~15)
__global__ __launch_bounds__(BLOCK) void k_topo_snapshot(size_t n1, size_t n2, const int *__restrict__ nb,
                                                         const int *__restrict__ bt, const int *__restrict__ ba,
                                                         int *__restrict__ nb0, int *__restrict__ bt0, int *__restrict__ ba0) {
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += stride) {
    if (i < n1) nb0[i] = nb[i];
    bt0[i] = bt[i];
    ba0[i] = ba[i];
  }
}
void launch_topo_snapshot(DeviceState &d) {
  const size_t nt = (size_t)d.maxtag + 2, n2 = nt * d.bpa;
  int grid = (int)std::min<size_t>((n2 + BLOCK - 1) / BLOCK, 8192);
  hipLaunchKernelGGL(k_topo_snapshot, dim3(std::max(grid, 1)), dim3(BLOCK), 0, d.stream, nt, n2, d.num_bond, d.bond_type,
                     d.bond_atom, d.num_bond0, d.bond_type0, d.bond_atom0);
}

// ========================================= ex_load ============================================
// base(a): every test of the candidate scan that does not depend on earlier pairs (fix_ex_load.cpp:453-494), AND the
// fact that the scan visits the pair at all: it walks `list->firstneigh` (:427-451), an occasional list that NPairCopy
// (src/npair_copy.cpp) aliases to the pair list of the last reneighbor.  (a, a+2) is an entry there only if it was within
// the list cutoff at that build and its special level is not dropped (npair_half_bin_newtoff.cpp:103-112: weight 0 ->
// no entry; e.g. `special_bonds lj 0 0 1` removes every 1-3 pair and ex_load then loads nothing).  The engine's full
// list holds the same pairs in both directions; the entry is looked up in the list of bead a, the lower local index at
// the canonical order, whose special list the reference consults (ASYM builds apply that rule to both directions).
// LISTSRC: 0 = one rank, every bead and its list are here -> ballot straight into the bitmask; 1 = decomposed: only the
// owner of bead a has its list, write 0/1 per pair for a max-reduction over the ranks (k_exload_bits packs it afterwards).
struct PairListView {
  int n_owned, npad;
  const int *map, *neigh, *numneigh;
  const int *crank;      // nullptr: local index = ID - 1
  // newton_pair on (`newton on off`): the reference's half/bin/newton list stores an owned-owned pair under the atom whose
  // neighbor bin comes first in (z, y, x) order, the lower local index inside one bin (npair_half_bin_newton.cpp:84-149,
  // nstencil_half_bin_3d_newton.cpp); bins of cutneighmax / 2 fitted to the box (nbin_standard.cpp:53-186), positions =
  // those of the last build (xht)
  int newton;
  double lo[3], bininv[3];
  int nbin[3];
  const double4 *xht;
};
__device__ __forceinline__ int local_index(const PairListView &V, int t) { return V.crank ? V.crank[t] : t - 1; }
// true: bead i is the storing end of the pair (i, j)
__device__ __forceinline__ bool stores_pair(const PairListView &V, int i, int j) {
  if (V.newton) {
    const double4 a = V.xht[i], b = V.xht[j];
    const double pa[3] = {a.x, a.y, a.z}, pb[3] = {b.x, b.y, b.z};
#pragma unroll
    for (int d = 2; d >= 0; d--) {
      const int ba = min((int)((pa[d] - V.lo[d]) * V.bininv[d]), V.nbin[d] - 1);
      const int bb = min((int)((pb[d] - V.lo[d]) * V.bininv[d]), V.nbin[d] - 1);
      if (ba != bb) return ba < bb;
    }
  }
  return local_index(V, i) < local_index(V, j);
}
// position of pair a = (a, a+2) in the scan: the local index of its storing end
__device__ __forceinline__ int visit_key(const PairListView &V, int a) {
  return stores_pair(V, a, a + 2) ? local_index(V, a) : local_index(V, a + 2);
}
template <int LISTSRC>
__global__ __launch_bounds__(BLOCK) void k_exload_base(Topo tp, ExLoadParams P, const double4 *__restrict__ xt,
                                                       const int *__restrict__ bc, PairListView V,
                                                       unsigned long long *__restrict__ bits,
                                                       double *__restrict__ rsq_out, int *__restrict__ base_i) {
  int a = blockIdx.x * BLOCK + threadIdx.x;   // pair a = (a, a+2); a = 0 is never valid
  bool base = false;
  if (a >= 1 && a + 2 <= tp.T) {
    int i = a, j = a + 2, mid = a + 1;
    int itype = tp.type_t[i], jtype = tp.type_t[j];
    bool possible = false;
    if (itype == P.iatomtype && jtype == P.jatomtype) {
      if ((P.imaxbond == 0 || bc[i] < P.imaxbond) && (P.jmaxbond == 0 || bc[j] < P.jmaxbond)) possible = true;
    } else if (itype == P.jatomtype && jtype == P.iatomtype) {
      if ((P.jmaxbond == 0 || bc[i] < P.jmaxbond) && (P.imaxbond == 0 || bc[j] < P.imaxbond)) possible = true;
    }
    // the pair is stored under the end with the lower LOCAL index (npair_half_bin_newtoff.cpp:90); its special list
    // and its neighbor list are the ones consulted (fix_ex_load.cpp:486-488)
    const bool swap = (V.crank || V.newton) && !stores_pair(V, i, j);
    const int is = swap ? j : i, js = swap ? i : j;
    possible = possible && in_group(tp, i) && in_group(tp, j);        // fix_ex_load.cpp:435,450
    if (possible && tp.num_bond[i] == 2 && tp.num_bond[j] == 2 && tp.num_bond[mid] == 2) {
      const int *sl = tp.special + (size_t)is * tp.ms;
      int n1 = tp.nspecial[3 * (size_t)is];
      for (int k = 0; k < n1; k++) if (sl[k] == js) possible = false;
      if (possible) {
        double rsq = d2(xt[i], xt[j]);   // stored coordinates, no minimum image (ghost entries are skipped)
        rsq_out[a] = rsq;
        base = rsq < P.cutsq;
      }
    }
    if (base) {                          // few pairs get here: walk bead a's list for a+2
      base = false;
      const int p = V.map[is], q = V.map[js];
      if (p >= 0 && p < V.n_owned && q >= 0) {
        const int w = V.numneigh[p], nn = w & NN_COUNT_MASK;       // (the bead's bond entries come first: skipped)
        for (int k = (w >> NN_BOND_SHIFT) & NN_NBOND_MASK; k < nn; k++)
          if ((V.neigh[(size_t)k * V.npad + p] & NEIGH_MASK) == q) { base = true; break; }
      }
    }
  }
  if (LISTSRC == 1) { if (a <= tp.T + 1) base_i[a] = base ? 1 : 0; return; }
  unsigned long long m = __ballot(base);
  if ((threadIdx.x & 63) == 0) bits[a >> 6] = m;
}
// Decomposed, canonical order: base(a) from what THIS rank holds.  Only the owner of the storing bead a (the lower local
// index) has a's pair list, so only it can say whether (a, a+2) is an entry; it also has a+2's current position - an
// entry of a's list is an owned bead or a ghost of the pair shell, and the halo was exchanged before the fix fired.  The
// rank sets the bit of the pairs it decides, the masks of the ranks are summed (every bit has one owner: sum = or).
// rsq of a pair is stored whenever both beads are here and one is owned: the owner of bead t later compares the two
// pairs t belongs to (closest-wins), and computes the same double from the same two operands as the pair's own owner.
__global__ __launch_bounds__(BLOCK) void k_exload_base_dd(Topo tp, ExLoadParams P, const double4 *__restrict__ pos,
                                                          const int *__restrict__ bc, PairListView V,
                                                          unsigned long long *__restrict__ bits,
                                                          double *__restrict__ rsq_out) {
  int a = blockIdx.x * BLOCK + threadIdx.x;
  bool base = false;
  if (a >= 1 && a + 2 <= tp.T) {
    const int i = a, j = a + 2, mid = a + 1;
    const int pa = V.map[i], pb = V.map[j];
    if (pa >= 0 && pb >= 0 && (pa < V.n_owned || pb < V.n_owned)) {
      int itype = tp.type_t[i], jtype = tp.type_t[j];
      bool possible = false;
      if (itype == P.iatomtype && jtype == P.jatomtype) {
        if ((P.imaxbond == 0 || bc[i] < P.imaxbond) && (P.jmaxbond == 0 || bc[j] < P.jmaxbond)) possible = true;
      } else if (itype == P.jatomtype && jtype == P.iatomtype) {
        if ((P.jmaxbond == 0 || bc[i] < P.jmaxbond) && (P.imaxbond == 0 || bc[j] < P.imaxbond)) possible = true;
      }
      possible = possible && in_group(tp, i) && in_group(tp, j);        // fix_ex_load.cpp:435,450
    if (possible && tp.num_bond[i] == 2 && tp.num_bond[j] == 2 && tp.num_bond[mid] == 2) {
        const int *sl = tp.special + (size_t)i * tp.ms;
        int n1 = tp.nspecial[3 * (size_t)i];
        for (int k = 0; k < n1; k++) if (sl[k] == j) possible = false;
        if (possible) {
          double rsq = d2(pos[pa], pos[pb]);   // a ghost is an unshifted copy of its owner's stored coordinates
          rsq_out[a] = rsq;
          base = rsq < P.cutsq;
        }
      }
      if (base) {
        base = false;
        if (pa < V.n_owned) {
          const int w = V.numneigh[pa], nn = w & NN_COUNT_MASK;
          for (int k = (w >> NN_BOND_SHIFT) & NN_NBOND_MASK; k < nn; k++)
            if ((V.neigh[(size_t)k * V.npad + pa] & NEIGH_MASK) == pb) { base = true; break; }
        }
      }
    }
  }
  unsigned long long m = __ballot(base);
  if ((threadIdx.x & 63) == 0) bits[a >> 6] = m;
}
__global__ __launch_bounds__(BLOCK) void k_exload_bits(int nt, const int *__restrict__ base_i,
                                                       unsigned long long *__restrict__ bits) {
  int a = blockIdx.x * BLOCK + threadIdx.x;
  unsigned long long m = __ballot(a < nt && base_i[a] != 0);
  if ((threadIdx.x & 63) == 0) bits[a >> 6] = m;
}
// Acceptance with a general visit order (local index != ID - 1, i.e. after an Atom::sort).  The scan visits pair a when it
// reaches the storing atom, local index key(a) = min(crank[a], crank[a+2]); pair a is skipped if partner[a+1] was already
// set, i.e. if pair a-1 or pair a+1 was ACCEPTED earlier in the scan.  That is the greedy maximal independent set of the
// path graph of the base pairs in key order; it is resolved in rounds: an undecided pair whose undecided neighbours all
// have larger keys is accepted, a pair next to an accepted one is rejected.  No round without progress can leave an
// undecided pair (keys are distinct), so the loop ends; chains of dependent pairs are as short as runs of base pairs.
// st: 0 undecided, 1 accepted, 2 rejected / not a base pair.
__global__ __launch_bounds__(BLOCK) void k_exload_greedy_init(int nt, const int *__restrict__ base_i, int *__restrict__ st) {
  int a = blockIdx.x * BLOCK + threadIdx.x;
  if (a < nt) st[a] = base_i[a] ? 0 : 2;
}
__global__ __launch_bounds__(BLOCK) void k_exload_keys(int T, PairListView V, int *__restrict__ key) {
  int a = blockIdx.x * BLOCK + threadIdx.x;
  if (a <= T + 1) key[a] = (a >= 1 && a + 2 <= T) ? visit_key(V, a) : 0;
}
__global__ __launch_bounds__(BLOCK) void k_exload_greedy_round(int T, const int *__restrict__ key, int *__restrict__ st,
                                                               int *__restrict__ changed) {
  int a = blockIdx.x * BLOCK + threadIdx.x;
  if (a < 1 || a + 2 > T || st[a] != 0) return;
  const int ka = key[a];
  bool blocked = false, wait = false;
  for (int b = a - 1; b <= a + 1; b += 2) {
    if (b < 1 || b + 2 > T) continue;
    const int sb = st[b];
    if (sb == 1) blocked = true;
    else if (sb == 0 && key[b] < ka) wait = true;
  }
  if (blocked) { st[a] = 2; *changed = 1; }
  else if (!wait) { st[a] = 1; *changed = 1; }
}
__global__ __launch_bounds__(BLOCK) void k_exload_accept_bits(int nt, const int *__restrict__ st,
                                                              unsigned long long *__restrict__ bits) {
  int a = blockIdx.x * BLOCK + threadIdx.x;
  unsigned long long m = __ballot(a < nt && st[a] == 1);
  if ((threadIdx.x & 63) == 0) bits[a >> 6] = m;
}
__device__ __forceinline__ bool bit_at(const unsigned long long *bits, int a) { return (bits[a >> 6] >> (a & 63)) & 1ull; }
// accepted(a) = base(a) && (number of consecutive base bits directly below a is even)
__device__ __forceinline__ bool accepted_at(const unsigned long long *__restrict__ bits, int a) {
  if (a < 1 || !bit_at(bits, a)) return false;
  int run = 0, pos = a - 1;
  while (pos >= 0) {
    int w = pos >> 6, b = pos & 63;                     // examine bits b..0 of word w, from the top
    unsigned long long x = bits[w] << (63 - b);        // bit b moved to the MSB
    unsigned long long inv = ~x;
    int ones = inv ? __clzll((long long)inv) : 64;      // leading ones of x
    if (ones > b + 1) ones = b + 1;
    run += ones;
    if (ones < b + 1) break;
    pos -= ones;
  }
  return (run & 1) == 0;
}
// DIRECT: `bits` already holds the accepted pairs (general visit order); else it holds base(a) and acceptance is the
// parity rule of the ID-order scan.  crank (DIRECT only) orders the two pairs a bead belongs to: the one visited first
// keeps the bead on an exact distance tie (strict `<` in fix_ex_load.cpp:496-503).
template <bool DIRECT>
__global__ __launch_bounds__(BLOCK) void k_exload_partner(int T, const unsigned long long *__restrict__ bits,
                                                          const double *__restrict__ rsq, const int *__restrict__ key,
                                                          int *__restrict__ partner, int *__restrict__ haspartner) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > T + 1) return;
  int p = 0;
  if (t >= 1 && t <= T) {
    bool lo = (t >= 3) && (DIRECT ? bit_at(bits, t - 2) : accepted_at(bits, t - 2));   // pair (t-2, t)
    bool hi = (t + 2 <= T) && (DIRECT ? bit_at(bits, t) : accepted_at(bits, t));       // pair (t, t+2)
    bool lo_first = true;                                                              // ID order: pair t-2 comes first
    if (DIRECT && lo && hi) lo_first = key[t - 2] < key[t];
    if (lo) p = t - 2;
    if (hi && (!lo || (lo_first ? rsq[t] < rsq[t - 2] : !(rsq[t - 2] < rsq[t])))) p = t + 2;
  }
  partner[t] = p;
  haspartner[t] = p != 0;
}
// decomposed: the owner of bead t picks t's partner (it holds the distances of both pairs t can belong to) and publishes
// the choice as one bit in one of two masks (partner below / above), which are summed over the ranks like the base bits
__global__ __launch_bounds__(BLOCK) void k_exload_partner_dd(int T, int n_owned, const int *__restrict__ map,
                                                             const unsigned long long *__restrict__ bits,
                                                             const double *__restrict__ rsq,
                                                             unsigned long long *__restrict__ m_lo,
                                                             unsigned long long *__restrict__ m_hi) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  bool clo = false, chi = false;
  if (t >= 1 && t <= T) {
    const int p = map[t];
    if (p >= 0 && p < n_owned) {
      bool lo = (t >= 3) && accepted_at(bits, t - 2);
      bool hi = (t + 2 <= T) && accepted_at(bits, t);
      clo = lo;
      if (hi && (!lo || rsq[t] < rsq[t - 2])) { chi = true; clo = false; }
    }
  }
  unsigned long long a = __ballot(clo), b = __ballot(chi);
  if ((threadIdx.x & 63) == 0) { m_lo[t >> 6] = a; m_hi[t >> 6] = b; }
}
__global__ __launch_bounds__(BLOCK) void k_exload_partner_unpack(int T, const unsigned long long *__restrict__ m_lo,
                                                                 const unsigned long long *__restrict__ m_hi,
                                                                 int *__restrict__ partner, int *__restrict__ haspartner) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > T + 1) return;
  int p = 0;
  if (t >= 1 && t <= T) { if (bit_at(m_lo, t)) p = t - 2; else if (bit_at(m_hi, t)) p = t + 2; }
  partner[t] = p;
  haspartner[t] = p != 0;
}
__global__ __launch_bounds__(BLOCK) void k_exload_create(Topo tp, ExLoadParams P, const int *__restrict__ partner,
                                                         const int *__restrict__ didx,
                                                         const uint32_t *__restrict__ draws, int *__restrict__ bc,
                                                         int *__restrict__ fin, double4 *__restrict__ pos,
                                                         const int *__restrict__ map, int *__restrict__ flags) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > tp.T + 1) return;
  int f = 0;
  if (t >= 1 && t <= tp.T) {
    int j = partner[t];
    if (j && partner[j] == t) {
      bool ok = true;
      if (P.fraction < 1.0) {
        int lo = t < j ? t : j;
        double prob = (double)draws[didx[lo]] * (1.0 / 16777216.0);
        if (prob >= P.fraction) ok = false;
      }
      if (ok) {
        int nb = tp.num_bond[t];
        if (nb == tp.bpa) { flags[FLAG_ERROR] = ERR_BPA; }
        else {
          tp.bond_type[(size_t)t * tp.bpa + nb] = P.btype;
          tp.bond_atom[(size_t)t * tp.bpa + nb] = j;
          tp.num_bond[t] = nb + 1;
          if (!dev_special_insert12(tp, t, j)) flags[FLAG_ERROR] = ERR_SPECIAL;
          int c = bc[t] + 1;
          bc[t] = c;
          int ty = tp.type_t[t], nty = ty;
          if (ty == P.iatomtype) { if (c == P.imaxbond) nty = P.inewtype; }
          else { if (c == P.jmaxbond) nty = P.jnewtype; }
          if (nty != ty) { tp.type_t[t] = nty; int pp = map[t]; if (pp >= 0) pos[pp].w = (double)nty; }
          f = j;
          if (t < j) atomicAdd(&flags[FLAG_COUNT_A], 1);
        }
      }
    }
  }
  fin[t] = f;
}

void launch_ex_load(DeviceState &d, const ExLoadParams &P, int slot, Comm *comm) {
  Topo tp = topo_of(d, P.groupbit);
  int T = d.maxtag, nt = T + 2, nb = (nt + BLOCK - 1) / BLOCK;
  hipStream_t st = d.stream;
  int *bc = d.le_i[I_BC], *partner = d.le_i[I_A], *has = d.le_i[I_B], *didx = d.le_i[I_C], *fin = d.le_i[I_D];
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_COUNT_A, 0, 4 * sizeof(int), st));   // COUNT_A, COUNT_B, NDRAW, NLIST
  hipLaunchKernelGGL(k_le_gather, dim3(nb), dim3(BLOCK), 0, st, tp, d.pos, d.xhold, d.map, d.dd ? 0 : 1, d.xt, d.xht, P.btype, bc, 0, d.flags);
  int nbw = ((nt + 63) / 64 * 64 + BLOCK - 1) / BLOCK;
  if (!d.neigh || !d.numneigh) throw LammpsError("fix ex_load needs a pair neighbor list");
  PairListView V{d.n, d.npad, d.map, d.neigh, d.numneigh, d.ident_order ? nullptr : d.crank, d.newton_pair,
                 {d.box.lo[0], d.box.lo[1], d.box.lo[2]}, {d.ref_bininv[0], d.ref_bininv[1], d.ref_bininv[2]},
                 {d.ref_nbin[0], d.ref_nbin[1], d.ref_nbin[2]}, d.xht};
  int *tmp_a = d.le_i[I_F], *tmp_b = d.le_i[I_G];
  const bool id_order_scan = d.ident_order && !d.newton_pair;
  if (dd_le_fast(d)) {
    // owner-computed bits instead of every bead's position: two mask reductions of N / 8 and N / 4 bytes per rank
    const size_t words = (size_t)nt / 64 + 16;
    unsigned long long *m_lo = d.le_bits + words, *m_hi = d.le_bits + 2 * words;
    hipLaunchKernelGGL(k_exload_base_dd, dim3(nbw), dim3(BLOCK), 0, st, tp, P, d.pos, bc, V, d.le_bits, d.le_d[0]);
    comm->allreduce_u32_sum(st, (unsigned *)d.le_bits, 2 * (size_t)(nbw * BLOCK / 64));
    hipLaunchKernelGGL(k_exload_partner_dd, dim3(nbw), dim3(BLOCK), 0, st, T, d.n, d.map, d.le_bits, d.le_d[0], m_lo, m_hi);
    comm->allreduce_u32_sum(st, (unsigned *)m_lo, 2 * 2 * words);
    hipLaunchKernelGGL(k_exload_partner_unpack, dim3(nb), dim3(BLOCK), 0, st, T, m_lo, m_hi, partner, has);
  } else if (d.dd && id_order_scan) {
    int *base_i = d.le_i[I_E];
    hipLaunchKernelGGL((k_exload_base<1>), dim3(nbw), dim3(BLOCK), 0, st, tp, P, d.xt, bc, V, d.le_bits, d.le_d[0], base_i);
    comm->allreduce_int_max(st, base_i, nt);       // the owner of bead a knows whether (a, a+2) is in its list
    hipLaunchKernelGGL(k_exload_bits, dim3(nbw), dim3(BLOCK), 0, st, nt, base_i, d.le_bits);
    hipLaunchKernelGGL((k_exload_partner<false>), dim3(nb), dim3(BLOCK), 0, st, T, d.le_bits, d.le_d[0], d.crank, partner, has);
  } else if (id_order_scan) {
    hipLaunchKernelGGL((k_exload_base<0>), dim3(nbw), dim3(BLOCK), 0, st, tp, P, d.xt, bc, V, d.le_bits, d.le_d[0], (int *)nullptr);
    hipLaunchKernelGGL((k_exload_partner<false>), dim3(nb), dim3(BLOCK), 0, st, T, d.le_bits, d.le_d[0], d.crank, partner, has);
  } else {
    // visit order != ID order (Atom::sort ran, unordered data file, or newton_pair on): greedy acceptance in visit
    // order, resolved in rounds
    int *base_i = d.le_i[I_E], *state = tmp_a, *key = tmp_b;
    hipLaunchKernelGGL(k_exload_keys, dim3(nb), dim3(BLOCK), 0, st, T, V, key);
    hipLaunchKernelGGL((k_exload_base<1>), dim3(nbw), dim3(BLOCK), 0, st, tp, P, d.xt, bc, V, d.le_bits, d.le_d[0], base_i);
    if (d.dd) comm->allreduce_int_max(st, base_i, nt);   // (keys and everything after are functions of replicated data: same rounds on every rank)
    hipLaunchKernelGGL(k_exload_greedy_init, dim3(nb), dim3(BLOCK), 0, st, nt, base_i, state);
    // rounds per host look: a round decides at least the pair with the smallest key of every undecided run, so a batch
    // grows with the rounds already spent (8, 16, 32, .. 256) instead of one host round trip per 8 rounds
    bool settled = false;
    for (int batch = 0, rounds = 8; batch < 4096 && !settled; batch++, rounds = std::min(2 * rounds, 256)) {
      HIP_CHECK(hipMemsetAsync(d.flags + FLAG_AUX, 0, sizeof(int), st));
      for (int r = 0; r < rounds; r++)
        hipLaunchKernelGGL(k_exload_greedy_round, dim3(nb), dim3(BLOCK), 0, st, T, key, state, d.flags + FLAG_AUX);
      sync_flags(d);
      // FLAG_AUX = some pair changed state in this batch: one more batch is needed to see a batch without changes
      settled = !d.flags_h[FLAG_AUX];
    }
    // never drop undecided pairs silently: the topology would differ from the reference's without any sign
    if (!settled) throw LammpsError("fix ex_load: the acceptance rounds did not converge (internal error)");
    hipLaunchKernelGGL(k_exload_accept_bits, dim3(nbw), dim3(BLOCK), 0, st, nt, state, d.le_bits);
    hipLaunchKernelGGL((k_exload_partner<true>), dim3(nb), dim3(BLOCK), 0, st, T, d.le_bits, d.le_d[0], key, partner, has);
  }
  if (P.fraction < 1.0) {
    scan_local_order(d, has, didx, FLAG_NDRAW, tmp_a, tmp_b);
    launch_ranmars_gen(d, slot, d.flags + FLAG_NDRAW, d.le_draws, nt);
  }
  hipLaunchKernelGGL(k_exload_create, dim3(nb), dim3(BLOCK), 0, st, tp, P, partner, didx, d.le_draws, bc, fin, d.pos,
                     d.map, d.flags);
  hipLaunchKernelGGL(k_topo_created, dim3((T + 63) / 64), dim3(64), 0, st, tp, fin, P.atype, d.flags);
}

// ========================================= bond/create =========================================
// stock FixBondCreate::post_integrate (src/MC/fix_bond_create.cpp:349-640).  Its candidate scan walks the pair list
// (:422-485): with newton off every owned bead meets each of its candidates, as i or as j, so a bead can pick its
// closest admissible partner from its own FULL list without looking at anyone else's choice.  The list is the one of
// the last reneighboring (the reference rebuilds an occasional list at the firing step; any pair now within the
// fix cutoff <= pair cutoff is in both).  From `partner` on the path is ex_load's (draws, creation, special lists).
__global__ __launch_bounds__(BLOCK) void k_bcreate_partner(Topo tp, ExLoadParams P, Box box, int n, int npad,
                                                           const double4 *__restrict__ pos, const int *__restrict__ tag,
                                                           const int *__restrict__ neigh, const int *__restrict__ numneigh,
                                                           const int *__restrict__ bc, int *__restrict__ partner,
                                                           int *__restrict__ haspartner, const double4 *__restrict__ xt) {
  // xt != nullptr (decomposed): current positions by tag from the all-gather; a ghost's slot in `pos` still holds the
  // previous step's position at this point of the step (the reference forward-communicates first, :368)
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  const int t = tag[p];
  const int itype = tp.type_t[t];
  int best_tag = 0;
  if ((itype == P.iatomtype || itype == P.jatomtype) && in_group(tp, t)) {      // fix_bond_create.cpp:421
    const double4 ri = xt ? xt[t] : pos[p];
    const int bci = bc[t];
    const int *sl = tp.special + (size_t)t * tp.ms;
    const int n1 = tp.nspecial[3 * (size_t)t];
    const int w = numneigh[p], nn = w & NN_COUNT_MASK;             // (the bead's bond entries come first: skipped)
    double best = 1.0e20;
    for (int k = (w >> NN_BOND_SHIFT) & NN_NBOND_MASK; k < nn; k++) {
      const int j = neigh[(size_t)k * npad + p] & NEIGH_MASK;
      const int tj = tag[j];
      if (!in_group(tp, tj)) continue;                                            // :432
      const int jtype = tp.type_t[tj];
      bool possible = false;
      if (itype == P.iatomtype && jtype == P.jatomtype) {
        if ((P.imaxbond == 0 || bci < P.imaxbond) && (P.jmaxbond == 0 || bc[tj] < P.jmaxbond)) possible = true;
      } else if (itype == P.jatomtype && jtype == P.iatomtype) {
        if ((P.jmaxbond == 0 || bci < P.jmaxbond) && (P.imaxbond == 0 || bc[tj] < P.imaxbond)) possible = true;
      }
      if (!possible) continue;
      for (int q = 0; q < n1; q++) if (sl[q] == tj) possible = false;     // :455-458 no duplicate bond
      if (!possible) continue;
      const double4 rj = xt ? xt[tj] : pos[j];
      double dx = ri.x - rj.x, dy = ri.y - rj.y, dz = ri.z - rj.z;
      if (dx > box.half[0]) dx -= box.prd[0]; else if (dx < -box.half[0]) dx += box.prd[0];
      if (dy > box.half[1]) dy -= box.prd[1]; else if (dy < -box.half[1]) dy += box.prd[1];
      if (dz > box.half[2]) dz -= box.prd[2]; else if (dz < -box.half[2]) dz += box.prd[2];
      const double rsq = dx * dx + dy * dy + dz * dz;
      if (rsq >= P.cutsq) continue;
      if (rsq < best) { best = rsq; best_tag = tj; }
    }
  }
  partner[t] = best_tag;
  haspartner[t] = best_tag != 0;
}
__global__ void k_nonzero(int n, const int *__restrict__ a, int *__restrict__ out) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) out[i] = a[i] != 0;
}
void launch_bond_create(DeviceState &d, const ExLoadParams &P, int slot, const int *bondcount, int nt_host, Comm *comm) {
  Topo tp = topo_of(d, P.groupbit);
  int T = d.maxtag, nt = T + 2, nb = (nt + BLOCK - 1) / BLOCK;
  hipStream_t st = d.stream;
  int *bc = d.le_i[I_BC], *partner = d.le_i[I_A], *has = d.le_i[I_B], *didx = d.le_i[I_C], *fin = d.le_i[I_D];
  if (nt_host != nt || (!d.dd && d.n != T)) throw LammpsError("fix bond/create: atom IDs must be 1..N");
  if (!d.neigh) throw LammpsError("fix bond/create needs a pair neighbor list");
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_COUNT_A, 0, 4 * sizeof(int), st));   // COUNT_A, COUNT_B, NDRAW, NLIST
  HIP_CHECK(hipMemcpyAsync(bc, bondcount, (size_t)nt * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipMemsetAsync(partner, 0, (size_t)nt * sizeof(int), st));
  HIP_CHECK(hipMemsetAsync(has, 0, (size_t)nt * sizeof(int), st));
  hipLaunchKernelGGL(k_bcreate_partner, dim3((d.n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, tp, P, d.box, d.n, d.npad, d.pos,
                     d.tag, d.neigh, d.numneigh, bc, partner, has, d.dd ? d.xt : nullptr);
  if (d.dd) {
    // every rank picked partners for the beads it owns (its lists hold the ghosts within the pair shell); the table by
    // tag is completed with a max-reduction (partner tags are > 0, unowned slots 0), the rest runs replicated as ex_load
    comm->allreduce_int_max(st, partner, nt);
    hipLaunchKernelGGL(k_nonzero, dim3(nb), dim3(BLOCK), 0, st, nt, partner, has);
  }
  if (P.fraction < 1.0) {
    scan_local_order(d, has, didx, FLAG_NDRAW, d.le_i[I_F], d.le_i[I_G]);
    launch_ranmars_gen(d, slot, d.flags + FLAG_NDRAW, d.le_draws, nt);
  }
  hipLaunchKernelGGL(k_exload_create, dim3(nb), dim3(BLOCK), 0, st, tp, P, partner, didx, d.le_draws, bc, fin, d.pos,
                     d.map, d.flags);
  hipLaunchKernelGGL(k_topo_created, dim3((T + 63) / 64), dim3(64), 0, st, tp, fin, P.atype, d.flags);
}
void bond_create_counts(DeviceState &d, int *bondcount, int nt) {
  HIP_CHECK(hipMemcpyAsync(bondcount, d.le_i[I_BC], (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, d.stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));
}

// ========================================= ex_unload ==========================================
__global__ __launch_bounds__(BLOCK) void k_exunload_partner(Topo tp, ExUnloadParams P, Box box,
                                                            const double4 *__restrict__ xt,
                                                            const double4 *__restrict__ xht, int *__restrict__ partner,
                                                            int *__restrict__ haspartner) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > tp.T + 1) return;
  int p = 0;
  if (t >= 1 && t <= tp.T) {
    double best = 0.0;
    int nb = tp.num_bond0[t];                       // the bond LIST (last reneighbor), not the current table
    double4 xi = xt[t], hi = xht[t];
    for (int m = 0; m < nb; m++) {
      if (tp.bond_type0[(size_t)t * tp.bpa + m] != P.btype) continue;
      int u = tp.bond_atom0[(size_t)t * tp.bpa + m];
      if (!in_group(tp, t) || !in_group(tp, u)) continue;                       // fix_ex_unload.cpp:228-229
      double4 xj = xt[u], hj = xht[u];
      // partner image frozen at the last reneighbor (closest image then; ntopo_bond_all.cpp:53,64)
      double h0 = hi.x - hj.x, h1 = hi.y - hj.y, h2 = hi.z - hj.z;
      double s0 = (h0 > box.half[0]) ? 1.0 : (h0 < -box.half[0]) ? -1.0 : 0.0;
      double s1 = (h1 > box.half[1]) ? 1.0 : (h1 < -box.half[1]) ? -1.0 : 0.0;
      double s2 = (h2 > box.half[2]) ? 1.0 : (h2 < -box.half[2]) ? -1.0 : 0.0;
      double dx = xi.x - (xj.x + s0 * box.prd[0]), dy = xi.y - (xj.y + s1 * box.prd[1]),
             dz = xi.z - (xj.z + s2 * box.prd[2]);
      double rsq = dx * dx + dy * dy + dz * dz;
      if (rsq <= P.cutsq) continue;
      if (rsq > best) { best = rsq; p = u; }
    }
  }
  partner[t] = p;
  haspartner[t] = p != 0;
}
__global__ __launch_bounds__(BLOCK) void k_exunload_break(Topo tp, ExUnloadParams P, const int *__restrict__ partner,
                                                          const int *__restrict__ didx,
                                                          const uint32_t *__restrict__ draws, int *__restrict__ fin,
                                                          int *__restrict__ flags) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > tp.T + 1) return;
  int f = 0;
  if (t >= 1 && t <= tp.T) {
    int j = partner[t];
    if (j && partner[j] == t) {
      bool ok = true;
      if (P.fraction < 1.0) {
        int lo = t < j ? t : j;
        double prob = (double)draws[didx[lo]] * (1.0 / 16777216.0);
        if (prob >= P.fraction) ok = false;
      }
      if (ok) {
        dev_delete_bond(tp, t, j);
        dev_special_remove12(tp, t, j, flags);
        f = j;
        if (t < j) atomicAdd(&flags[FLAG_COUNT_A], 1);
      }
    }
  }
  fin[t] = f;
}
void launch_ex_unload(DeviceState &d, const ExUnloadParams &P, int slot) {
  Topo tp = topo_of(d, P.groupbit);
  int T = d.maxtag, nt = T + 2, nb = (nt + BLOCK - 1) / BLOCK;
  hipStream_t st = d.stream;
  int *bc = d.le_i[I_BC], *partner = d.le_i[I_A], *has = d.le_i[I_B], *didx = d.le_i[I_C], *fin = d.le_i[I_D];
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_COUNT_A, 0, 4 * sizeof(int), st));
  hipLaunchKernelGGL(k_le_gather, dim3(nb), dim3(BLOCK), 0, st, tp, d.pos, d.xhold, d.map, d.dd ? 0 : 1, d.xt, d.xht, P.btype, bc, 0, d.flags);
  hipLaunchKernelGGL(k_exunload_partner, dim3(nb), dim3(BLOCK), 0, st, tp, P, d.box, d.xt, d.xht, partner, has);
  if (P.fraction < 1.0) {
    scan_local_order(d, has, didx, FLAG_NDRAW, d.le_i[I_F], d.le_i[I_G]);
    launch_ranmars_gen(d, slot, d.flags + FLAG_NDRAW, d.le_draws, nt);
  }
  hipLaunchKernelGGL(k_exunload_break, dim3(nb), dim3(BLOCK), 0, st, tp, P, partner, didx, d.le_draws, fin, d.flags);
  hipLaunchKernelGGL(k_topo_broken, dim3((T + 63) / 64), dim3(64), 0, st, tp, fin, P.angleflag, d.flags);
}

// ========================================= extrusion ==========================================
enum { LF_VALID = 1, LF_SL = 2, LF_SR = 4, LF_DL = 8, LF_DR = 16, LF_DL2 = 32, LF_DR2 = 64 };
enum { CASE_NONE = 0, CASE_BOTH = 1, CASE_LEFT = 2, CASE_RIGHT = 3 };

// which beads head a bond-list entry of type btype (ntopo_bond_all.cpp:52-73 with newton_bond off):
// from t if t < partner (local index = ID-1), or from both ends if the bond straddled a face at the last build
__global__ __launch_bounds__(BLOCK) void k_ext_listflag(Topo tp, int btype, Box box, const double4 *__restrict__ xht,
                                                        const int *__restrict__ crank, int *__restrict__ lflag,
                                                        int *__restrict__ lpart) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t > tp.T + 1) return;
  int fl = 0, u = 0;
  if (t >= 1 && t <= tp.T) {
    int nb = tp.num_bond0[t];                       // bond-list entries (last reneighbor)
    for (int m = 0; m < nb; m++)
      if (tp.bond_type0[(size_t)t * tp.bpa + m] == btype) { u = tp.bond_atom0[(size_t)t * tp.bpa + m]; break; }
    if (u && !(in_group(tp, t) && in_group(tp, u))) u = 0;                      // fix_extrusion.cpp:373-376
    if (u) {
      double4 hi = xht[t], hj = xht[u];
      double h0 = hi.x - hj.x, h1 = hi.y - hj.y, h2 = hi.z - hj.z;
      bool straddle = fabs(h0) > box.half[0] || fabs(h1) > box.half[1] || fabs(h2) > box.half[2];
      // listed from the end with the lower LOCAL index (`newton_bond || i < atom1`, ntopo_bond_all.cpp:66); crank == nullptr: ID order
      const bool first = crank ? crank[t] < crank[u] : t < u;
      fl = (straddle || first) ? 1 : 0;
    }
  }
  lflag[t] = fl;
  lpart[t] = u;
}
// per listing: skip test, structural part of can(L)/can(R), number of RNG draws (fix_extrusion.cpp:398-429)
__global__ __launch_bounds__(BLOCK) void k_ext_prepare(Topo tp, ExtrusionParams P, const int *__restrict__ lflag,
                                                       const int *__restrict__ lidx, const int *__restrict__ lpart,
                                                       const int *__restrict__ bc, int *__restrict__ list_l,
                                                       int *__restrict__ list_r, int *__restrict__ list_f,
                                                       int *__restrict__ ndraw) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t < 1 || t > tp.T || !lflag[t]) return;
  int k = lidx[t], u = lpart[t];
  int l = t < u ? t : u, r = t < u ? u : t;
  int f = 0, nd = 0;
  int nbl = tp.num_bond[l], nbr = tp.num_bond[r];
  if (!(nbl == 1 || nbr == 1 || nbl == 0 || nbr == 0 || bc[l] != 1 || bc[r] != 1)) {
    f |= LF_VALID;
    int L = l - 1, R = r + 1;   // sentinels 0 and T+1 have num_bond = 0 -> structural test fails
    int tyL = tp.type_t[L], tyR = tp.type_t[R];
    bool sL = (tp.num_bond[L] - bc[L] == 2) && bc[L] == 0 &&
              (tyL == P.ctcf_left || tyL == P.ctcf_right || tyL == P.ctcf_lr || tyL == P.neutral);
    bool sR = (tp.num_bond[R] - bc[R] == 2) && bc[R] == 0 &&
              (tyR == P.ctcf_left || tyR == P.ctcf_right || tyR == P.ctcf_lr || tyR == P.neutral);
    // can(X, blk) = .. && (type != blk || p > U()) && (type != ctcf_lr || p > U())  (fix_extrusion.cpp:413-429): one draw per
    // true type test; a roadblock type that EQUALS the side's barrier type makes two chained draws, the second only when the
    // first lets the extruder through (LF_D?2; the offsets of such a firing are walked serially, k_ext_chain_offsets)
    if (sL) {
      f |= LF_SL;
      const int nt = (tyL == P.ctcf_left ? 1 : 0) + (tyL == P.ctcf_lr ? 1 : 0);
      if (nt >= 1) { f |= LF_DL; nd++; }
      if (nt == 2) { f |= LF_DL2; nd++; }
    }
    if (sR) {
      f |= LF_SR;
      const int nt = (tyR == P.ctcf_right ? 1 : 0) + (tyR == P.ctcf_lr ? 1 : 0);
      if (nt >= 1) { f |= LF_DR; nd++; }
      if (nt == 2) { f |= LF_DR2; nd++; }
    }
  }
  list_l[k] = l; list_r[k] = r; list_f[k] = f; ndraw[k] = nd;
}
// per listing: barrier draws -> case and claim distance; register claim events on the free target beads
__global__ __launch_bounds__(BLOCK) void k_ext_claims(int T, ExtrusionParams P, const int *__restrict__ nlist_ptr,
                                                      const int *__restrict__ list_l, const int *__restrict__ list_r,
                                                      const int *__restrict__ list_f, const int *__restrict__ doff,
                                                      const uint32_t *__restrict__ draws,
                                                      const double4 *__restrict__ xt, int *__restrict__ kcase,
                                                      double *__restrict__ kq, int *__restrict__ ev_cnt,
                                                      int *__restrict__ ev_k) {
  int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= *nlist_ptr) return;
  int f = list_f[k], c = CASE_NONE;
  double q = BIGD;
  if (f & LF_VALID) {
    int l = list_l[k], r = list_r[k], L = l - 1, R = r + 1, o = doff[k];
    const double inv = 1.0 / 16777216.0;
    bool canL = (f & LF_SL) != 0, canR = (f & LF_SR) != 0;
    if (f & LF_DL) {
      bool pass = P.through_prob > (double)draws[o] * inv; o++;
      if (pass && (f & LF_DL2)) { pass = P.through_prob > (double)draws[o] * inv; o++; }
      if (!pass) canL = false;
    }
    if (f & LF_DR) {
      bool pass = P.through_prob > (double)draws[o] * inv; o++;
      if (pass && (f & LF_DR2)) { pass = P.through_prob > (double)draws[o] * inv; o++; }
      if (!pass) canR = false;
    }
    if (canL && canR) { c = CASE_BOTH; q = d2(xt[L], xt[R]); }
    else if (canL) { c = CASE_LEFT; q = d2(xt[L], xt[r]); }
    else if (canR) { c = CASE_RIGHT; q = d2(xt[l], xt[R]); }
    if (c == CASE_BOTH || c == CASE_LEFT) { int s = atomicAdd(&ev_cnt[L], 1); if (s < 4) ev_k[(size_t)s * (T + 2) + L] = k; }
    if (c == CASE_BOTH || c == CASE_RIGHT) { int s = atomicAdd(&ev_cnt[R], 1); if (s < 4) ev_k[(size_t)s * (T + 2) + R] = k; }
  }
  kcase[k] = c;
  kq[k] = q;
}
// chained draws (roadblock type == a side's barrier type): how many draws a listing consumes depends on its own first draw,
// so the stream offsets are one serial walk over the listings in bond-list order - a single lane; a firing has as many
// listings as there are extruders.  doff[k] = the listing's first draw, *total = what the firing consumed
__global__ void k_ext_chain_offsets(ExtrusionParams P, const int *__restrict__ nlist_ptr, const int *__restrict__ list_f,
                                    const uint32_t *__restrict__ draws, int *__restrict__ doff, int *__restrict__ total) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const double inv = 1.0 / 16777216.0;
  const int n = *nlist_ptr;
  int o = 0;
  for (int k = 0; k < n; k++) {
    const int f = list_f[k];
    doff[k] = o;
    if (!(f & LF_VALID)) continue;
    if (f & LF_DL) { const bool pass = P.through_prob > (double)draws[o] * inv; o++; if (pass && (f & LF_DL2)) o++; }
    if (f & LF_DR) { const bool pass = P.through_prob > (double)draws[o] * inv; o++; if (pass && (f & LF_DR2)) o++; }
  }
  *total = o;
}
// dc[X] as listing k saw it = min q over earlier claim events on X (strict '<' updates, fix_extrusion.cpp:436-515)
__device__ __forceinline__ double dc_before(int T, int X, int k, const int *ev_cnt, const int *ev_k, const double *kq) {
  double m = BIGD;
  int n = min(ev_cnt[X], 4);
  for (int s = 0; s < n; s++) {
    int e = ev_k[(size_t)s * (T + 2) + X];
    if (e < k && kq[e] < m) m = kq[e];
  }
  return m;
}
__global__ __launch_bounds__(BLOCK) void k_ext_resolve(int T, const int *__restrict__ nlist_ptr,
                                                       const int *__restrict__ list_l, const int *__restrict__ list_r,
                                                       const int *__restrict__ kcase, const double *__restrict__ kq,
                                                       const int *__restrict__ ev_cnt, const int *__restrict__ ev_k,
                                                       int *__restrict__ to_add, int *__restrict__ to_remove) {
  int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= *nlist_ptr) return;
  int c = kcase[k];
  if (c == CASE_NONE) return;
  int l = list_l[k], r = list_r[k], L = l - 1, R = r + 1;
  double q = kq[k];
  bool proceed;
  if (c == CASE_BOTH) proceed = !(q >= dc_before(T, L, k, ev_cnt, ev_k, kq) && q >= dc_before(T, R, k, ev_cnt, ev_k, kq));
  else if (c == CASE_LEFT) proceed = !(q >= dc_before(T, L, k, ev_cnt, ev_k, kq));
  else proceed = !(q >= dc_before(T, R, k, ev_cnt, ev_k, kq));
  if (!proceed) return;
  to_remove[l] = r;
  to_remove[r] = l;
  // the stationary anchor of a one-sided move records its new partner once (dc[anchor] == BIG test)
  if (c == CASE_LEFT) to_add[r] = L;
  if (c == CASE_RIGHT) to_add[l] = R;
}
// final to_add of a claimed free bead: the closest claim, earliest listing on ties
__global__ __launch_bounds__(BLOCK) void k_ext_toadd(int T, const int *__restrict__ ev_cnt, const int *__restrict__ ev_k,
                                                     const double *__restrict__ kq, const int *__restrict__ kcase,
                                                     const int *__restrict__ list_l, const int *__restrict__ list_r,
                                                     int *__restrict__ to_add) {
  int X = blockIdx.x * BLOCK + threadIdx.x;
  if (X < 1 || X > T) return;
  int n = min(ev_cnt[X], 4);
  if (n == 0) return;
  int best = -1;
  double bq = BIGD;
  for (int s = 0; s < n; s++) {
    int e = ev_k[(size_t)s * (T + 2) + X];
    double q = kq[e];
    if (q < bq || (q == bq && e < best)) { bq = q; best = e; }
  }
  if (best < 0) return;
  int l = list_l[best], r = list_r[best], c = kcase[best];
  int val;
  if (X == l - 1) val = (c == CASE_BOTH) ? r + 1 : r;   // claimed as L
  else val = (c == CASE_BOTH) ? l - 1 : l;              // claimed as R
  to_add[X] = val;
}
// phase 2 (fix_extrusion.cpp:517-599): a bead whose chosen partner chose someone else cancels the removal
// of its own extruder; patterns evaluated on the phase-1 snapshot tr0, zeros written to tr1
__global__ __launch_bounds__(BLOCK) void k_ext_losers(int T, const int *__restrict__ to_add, const int *__restrict__ tr0,
                                                      int *__restrict__ tr1) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < 1 || i > T) return;
  int j = to_add[i];
  if (j == 0 || to_add[j] == i) return;
  int lb, rb;
  bool iisleft = i < j;
  if (iisleft) { lb = i + 1; rb = j - 1; } else { lb = j + 1; rb = i - 1; }
  if (iisleft) {
    if (lb == tr0[rb] && tr0[lb] == rb) { tr1[lb] = 0; tr1[rb] = 0; }
    else if (i == tr0[rb] && tr0[i] == rb) { tr1[i] = 0; tr1[rb] = 0; }
    else if (lb == tr0[j] && tr0[lb] == j) { tr1[lb] = 0; tr1[j] = 0; }
    else if (i == tr0[j] && j == tr0[i]) { tr1[i] = 0; tr1[j] = 0; }
  } else {
    if (lb == tr0[rb] && tr0[lb] == rb) { tr1[lb] = 0; tr1[rb] = 0; }
    else if (i == tr0[lb] && tr0[i] == lb) { tr1[i] = 0; tr1[lb] = 0; }
    else if (rb == tr0[j] && tr0[rb] == j) { tr1[rb] = 0; tr1[j] = 0; }
    else if (i == tr0[j] && j == tr0[i]) { tr1[i] = 0; tr1[j] = 0; }
  }
}
// phase 3 (:618-692): one thread per extruder (its lower end) merges two one-sided moves and deletes the bond
__global__ __launch_bounds__(BLOCK) void k_ext_remove(Topo tp, const int *__restrict__ tr, int *__restrict__ to_add,
                                                      int *__restrict__ fin_rm, int *__restrict__ flags) {
  int lb = blockIdx.x * BLOCK + threadIdx.x;
  if (lb < 1 || lb > tp.T) return;
  int rb = tr[lb];
  if (rb == 0 || rb < lb || tr[rb] != lb) return;
  if (to_add[lb - 1] == rb && to_add[rb] == lb - 1 && to_add[lb] == rb + 1 && to_add[rb + 1] == lb) {
    to_add[lb - 1] = rb + 1; to_add[rb + 1] = lb - 1; to_add[lb] = 0; to_add[rb] = 0;
  }
  if ((to_add[lb - 1] == rb && to_add[rb] == lb - 1) || (to_add[lb - 1] == rb + 1 && to_add[rb + 1] == lb - 1) ||
      (to_add[lb] == rb + 1 && to_add[rb + 1] == lb)) {
    dev_delete_bond(tp, lb, rb); dev_special_remove12(tp, lb, rb, flags);
    dev_delete_bond(tp, rb, lb); dev_special_remove12(tp, rb, lb, flags);
    fin_rm[lb] = rb; fin_rm[rb] = lb;
    atomicAdd(&flags[FLAG_COUNT_A], 1);
  }
}
// phase 4 (:699-786)
__global__ __launch_bounds__(BLOCK) void k_ext_create(Topo tp, int btype, const int *__restrict__ tr,
                                                      const int *__restrict__ to_add, int *__restrict__ bc,
                                                      int *__restrict__ fin_add, int *__restrict__ flags) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < 1 || i > tp.T) return;
  int j = to_add[i];
  if (j == 0 || to_add[j] != i) return;
  int nb = tp.num_bond[i];
  if (nb == tp.bpa) return;
  int lb = i < j ? i : j, rb = i < j ? j : i;
  if ((tr[lb + 1] == rb && tr[rb] == lb + 1) || (tr[lb + 1] == rb - 1 && tr[rb - 1] == lb + 1) ||
      (tr[lb] == rb - 1 && tr[rb - 1] == lb)) {
    tp.bond_type[(size_t)i * tp.bpa + nb] = btype;
    tp.bond_atom[(size_t)i * tp.bpa + nb] = j;
    tp.num_bond[i] = nb + 1;
    if (!dev_special_insert12(tp, i, j)) flags[FLAG_ERROR] = ERR_SPECIAL;
    bc[i]++;
    fin_add[i] = j;
    if (i < j) atomicAdd(&flags[FLAG_COUNT_B], 1);
  }
}

void launch_extrusion(DeviceState &d, const ExtrusionParams &P, int slot) {
  Topo tp = topo_of(d, P.groupbit);
  int T = d.maxtag, nt = T + 2, nb = (nt + BLOCK - 1) / BLOCK;
  hipStream_t st = d.stream;
  int *bc = d.le_i[I_BC], *lflag = d.le_i[I_A], *lidx = d.le_i[I_B], *lpart = d.le_i[I_C], *list_l = d.le_i[I_D],
      *list_r = d.le_i[I_E], *list_f = d.le_i[I_F], *ndraw = d.le_i[I_G], *doff = d.le_i[I_H], *kcase = d.le_i[I_J],
      *ev_cnt = d.le_i[I_K], *to_add = d.le_i[I_L], *tr0 = d.le_i[I_M], *tr1 = d.le_i[I_N], *fin_rm = d.le_i[I_O],
      *fin_add = d.le_i[I_P];
  int *ev_k = d.le_list;
  double *kq = d.le_d[0];
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_COUNT_A, 0, 4 * sizeof(int), st));
  for (int *a : {ndraw, ev_cnt, to_add, tr0, fin_rm, fin_add}) HIP_CHECK(hipMemsetAsync(a, 0, (size_t)nt * sizeof(int), st));
  hipLaunchKernelGGL(k_le_gather, dim3(nb), dim3(BLOCK), 0, st, tp, d.pos, d.xhold, d.map, d.dd ? 0 : 1, d.xt, d.xht, P.btype, bc, 1, d.flags);
  hipLaunchKernelGGL(k_ext_listflag, dim3(nb), dim3(BLOCK), 0, st, tp, P.btype, d.box, d.xht,
                     d.ident_order ? (const int *)nullptr : d.crank, lflag, lpart);
  scan_local_order(d, lflag, lidx, FLAG_NLIST, tr1, kcase);       // listing index = position in the bond list (tr1 / kcase: free until later)
  hipLaunchKernelGGL(k_ext_prepare, dim3(nb), dim3(BLOCK), 0, st, tp, P, lflag, lidx, lpart, bc, list_l, list_r, list_f,
                     ndraw);
  scan_ex(d, ndraw, doff, nt, FLAG_NDRAW);    // ndraw is zero beyond the number of listings
  if (P.ctcf_lr == P.ctcf_left || P.ctcf_lr == P.ctcf_right) {
    // chained draws: generate the upper bound from a COPY of the stream state, walk the listings serially for the offsets
    // and the consumed total, then advance the real state by exactly that (the same values again)
    uint32_t *real = d.le_rng_state + slot * 100, *scratch = d.le_rng_state + LE_MAX_FIXES * 100;
    HIP_CHECK(hipMemcpyAsync(scratch, real, 100 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    launch_ranmars_gen(d, LE_MAX_FIXES, d.flags + FLAG_NDRAW, d.le_draws, 2 * nt);
    hipLaunchKernelGGL(k_ext_chain_offsets, dim3(1), dim3(64), 0, st, P, d.flags + FLAG_NLIST, list_f, d.le_draws, doff,
                       d.flags + FLAG_NDRAW);
  }
  launch_ranmars_gen(d, slot, d.flags + FLAG_NDRAW, d.le_draws, 2 * nt);
  hipLaunchKernelGGL(k_ext_claims, dim3(nb), dim3(BLOCK), 0, st, T, P, d.flags + FLAG_NLIST, list_l, list_r, list_f, doff,
                     d.le_draws, d.xt, kcase, kq, ev_cnt, ev_k);
  hipLaunchKernelGGL(k_ext_resolve, dim3(nb), dim3(BLOCK), 0, st, T, d.flags + FLAG_NLIST, list_l, list_r, kcase, kq,
                     ev_cnt, ev_k, to_add, tr0);
  hipLaunchKernelGGL(k_ext_toadd, dim3(nb), dim3(BLOCK), 0, st, T, ev_cnt, ev_k, kq, kcase, list_l, list_r, to_add);
  HIP_CHECK(hipMemcpyAsync(tr1, tr0, (size_t)nt * sizeof(int), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_ext_losers, dim3(nb), dim3(BLOCK), 0, st, T, to_add, tr0, tr1);
  hipLaunchKernelGGL(k_ext_remove, dim3(nb), dim3(BLOCK), 0, st, tp, tr1, to_add, fin_rm, d.flags);
  hipLaunchKernelGGL(k_ext_create, dim3(nb), dim3(BLOCK), 0, st, tp, P.btype, tr1, to_add, bc, fin_add, d.flags);
  hipLaunchKernelGGL(k_topo_broken, dim3((T + 63) / 64), dim3(64), 0, st, tp, fin_rm, 0, d.flags);
  hipLaunchKernelGGL(k_topo_created, dim3((T + 63) / 64), dim3(64), 0, st, tp, fin_add, 0, d.flags);
}

}  // namespace lmp_le
