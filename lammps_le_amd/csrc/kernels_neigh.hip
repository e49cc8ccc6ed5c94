// kernels_neigh.hip — reneighboring on the device: periodic wrap, spatial (cell) sort of the
// physical arrays, bond-partner table, full neighbor list with special-bond encoding.
//
// Reference behaviour restated (the data structures are NOT the reference's):
//   Domain::pbc                      src/domain.cpp:528-645   (wrap + image flags)
//   NBinStandard / NPairHalfBin*     src/nbin_standard.cpp:192-232, src/npair_half_bin_newtoff.cpp:36-128
//   NPair::find_special + flags      src/npair.h:112-136, src/neighbor.cpp:360-376, src/lmptype.h:61-62
//   NTopoBondAll::build              src/ntopo_bond_all.cpp:39-86 (partner = closest image -> min image here)
//   Neighbor::build xhold            src/neighbor.cpp:2048-2052
//
// MI355X design: cells of edge >= cutneigh, atoms re-sorted into cell order at every build
// (deterministic: ties by tag), so a cell is a contiguous range and the 27-cell sweep of the list
// build and the x-gathers of the pair kernel read L2-resident rows.  The list is a FULL list stored
// column-major (neigh[k][p]) so that lane p's k-th neighbor load is coalesced and no force atomics
// are needed.
#include "device.h"
#include "bin_inl.h"

namespace lmp_le {

constexpr int BLOCK = 256;
constexpr int SCAN_BLOCK = 1024;

__global__ __launch_bounds__(BLOCK) void k_wrap_bin(int n, double4 *__restrict__ pos, int *__restrict__ img, int npad,
                                                    Box box, int ncx, int ncy, int ncz, double cix, double ciy,
                                                    double ciz, double zlo_ext, int *__restrict__ cell_of,
                                                    int *__restrict__ cell_count, int *__restrict__ rank,
                                                    int *__restrict__ flags, const int *__restrict__ gone,
                                                    int sentinel, int rtile) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  double4 r = pos[p];
  if (!(isfinite(r.x) && isfinite(r.y) && isfinite(r.z))) {
    flags[FLAG_ERROR] = ERR_NONFINITE;
    cell_of[p] = 0;
    rank[p] = atomicAdd(&cell_count[0], 1);
    return;
  }
  int dix, diy, diz;
  wrap_into_box(r, box, dix, diy, diz);
  if (dix) img[p] += dix;
  if (diy) img[npad + p] += diy;
  if (diz) img[2 * npad + p] += diz;
  pos[p] = r;
  int cell = cell_index(r, box, ncx, ncy, ncz, cix, ciy, ciz, zlo_ext, rtile);
  // decomposed runs: a bead that has just migrated to another slab is binned into a sentinel cell behind all real
  // cells, so the sort that follows also compacts the array (no separate keep/scan/scatter pass)
  if (gone && gone[p]) cell = sentinel;
  cell_of[p] = cell;
  rank[p] = count_into_cell(cell, cell_count);   // arrival order inside the cell; k_sort_cells makes it canonical
}

// ---- exclusive scan of cell_count[0..m) into cell_start[0..m], two small kernels ----
// (the counts are zeroed as they are read: the next binning pass - in the step kernel or in k_wrap_bin - finds a clean
// array without a memset, which as a blit costs two launches, 11 us, between two step kernels)
// (wavefront scans by lane shifts, 16 wave totals through LDS: two barriers per block instead of twenty)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(v, off, 64); if (lane >= off) v += t; }
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_local(int m, int *__restrict__ in, int *__restrict__ out,
                                                           int *__restrict__ blocksum) {
  __shared__ int wtot[SCAN_BLOCK / 64];
  int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  int v = 0;
  if (i < m) { v = in[i]; in[i] = 0; }
  const int inc = wave_inclusive_scan(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  int before = 0;                                  // totals of the wavefronts in front of mine
  for (int k = 0; k < w; k++) before += wtot[k];
  if (i < m) out[i] = before + inc - v;            // exclusive
  if (threadIdx.x == SCAN_BLOCK - 1) blocksum[blockIdx.x] = before + inc;
}
// second and last pass: every block sums the block totals in front of it itself (a few hundred L2-resident words; the
// separate single-block scan of the totals cost a launch) and adds the offset to its slice
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_add(int m, int *__restrict__ out, const int *__restrict__ blocksum,
                                                         int total_slot_n) {
  __shared__ int wtot[SCAN_BLOCK / 64];
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_BLOCK) part += blocksum[b];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = part;
  __syncthreads();
  int offset = 0;
#pragma unroll
  for (int k = 0; k < SCAN_BLOCK / 64; k++) offset += wtot[k];
  int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
  if (i < m) out[i] += offset;
  if (i == 0) out[m] = total_slot_n;
}

__global__ __launch_bounds__(BLOCK) void k_scatter(int n, const int *__restrict__ cell_of,
                                                   const int *__restrict__ cell_start, const int *__restrict__ rank,
                                                   int *__restrict__ perm) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  perm[cell_start[cell_of[p]] + rank[p]] = p;
}
// one thread per cell: order the cell's entries by tag so the final order is deterministic
// (`map` != nullptr: the new index of every bead is known here already - map[tag] is written now, so that the permute pass
//  that follows can translate bond partners while it moves the beads)
__global__ __launch_bounds__(BLOCK) void k_sort_cells(int ncells, const int *__restrict__ cell_start,
                                                      int *__restrict__ perm, const int *__restrict__ tag,
                                                      int *__restrict__ map) {
  int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= ncells) return;
  int b = cell_start[c], e = cell_start[c + 1];
  for (int i = b + 1; i < e; i++) {
    int pi = perm[i], ti = tag[pi], j = i - 1;
    while (j >= b && tag[perm[j]] > ti) { perm[j + 1] = perm[j]; j--; }
    perm[j + 1] = pi;
  }
  if (map)
    for (int i = b; i < e; i++) map[tag[perm[i]]] = i;
}
// bond-partner table of bead s: physical index + type of every stored bond (the step kernel's `bpart` rows)
struct BondTabArgs {
  int bpa, maxtag;
  const int *num_bond, *bond_type, *bond_atom;
  const int *pack;         // packed records by tag (DeviceState::bond_pack), stride ints each
  int stride;
  int *bpart;              // nullptr = nothing to do
  unsigned char *phase;
  const double4 *pos;      // positions of THIS build (wrapped, cell order; ghosts behind the owned beads)
  unsigned long long *bshift;
  double hx, hy, hz;       // half box
};
__device__ __forceinline__ void bond_table_row(int s, int n, int npad, const int *__restrict__ rec, const int *__restrict__ map,
                                               const BondTabArgs &B, int *__restrict__ flags) {
  int nb = rec[0];
  bool ghost = false;
  const bool freeze = B.bshift != nullptr;
  double4 rs = make_double4(0.0, 0.0, 0.0, 0.0);
  if (freeze) rs = B.pos[s];
  unsigned long long code = 0ull;
  int slot = 0;                             // row of this bond in the bead's list (switched-off bonds take none)
  for (int m = 0; m < B.bpa; m++) {
    int e = -1;
    if (m < nb) {
      const int w = rec[1 + m];
      int bt = w >> BOND_TYPE_SHIFT;
      int u = w & BOND_IDX_MASK;
      int q = (u >= 1 && u <= B.maxtag) ? map[u] : -1;
      if (q < 0) flags[FLAG_ERROR] = ERR_BOND_MISSING;
      else if (bt > 0) {
        e = (bt << BOND_TYPE_SHIFT) | q; ghost = ghost || q >= n;
        // Domain::closest_image at the build (src/ntopo_bond_all.cpp:59): the partner's image stays what it is now until
        // the next reneighbor, however the bond stretches in between
        if (freeze) {
          const double4 rq = B.pos[q];
          const double dx = rs.x - rq.x, dy = rs.y - rq.y, dz = rs.z - rq.z;
          const unsigned c = (dx > B.hx ? 1u : dx < -B.hx ? 2u : 0u) | (dy > B.hy ? 4u : dy < -B.hy ? 8u : 0u) |
                             (dz > B.hz ? 16u : dz < -B.hz ? 32u : 0u);
          code |= (unsigned long long)c << (BSHIFT_BITS * slot);
        }
        slot++;
      }
    }
    B.bpart[(size_t)m * npad + s] = e;
  }
  if (freeze) B.bshift[s] = code;
  if (B.phase && ghost) B.phase[s] = 1;     // reads a ghost position: phase 1 of a decomposed step
}
// the beads move into cell order.  `B.bpart` != nullptr (one GPU, bonds whose image needs no freezing): map[] is complete
// (k_sort_cells), and the bond-partner table of the new order is written in the same pass - the scattered reads of the
// packed bond records and of map[] ride under the streaming of the permutation instead of taking a launch of their own
// (k_bond_table: 27 us per rebuild at 1M beads from the scrambled start).
__global__ __launch_bounds__(BLOCK) void k_permute(int n, int npad, const int *__restrict__ perm,
                                                   const double4 *__restrict__ pos, double4 *__restrict__ pos_new,
                                                   double4 *__restrict__ xhold, const double *__restrict__ vx,
                                                   const double *__restrict__ vy, const double *__restrict__ vz,
                                                   double *__restrict__ vxn, double *__restrict__ vyn,
                                                   double *__restrict__ vzn, const int *__restrict__ tag,
                                                   int *__restrict__ tagn, const int *__restrict__ img,
                                                   int *__restrict__ imgn, int *__restrict__ map,
                                                   float4 *__restrict__ posf, int wrap, Box box, BondTabArgs B,
                                                   const int4 *__restrict__ pack_old, int4 *__restrict__ pack_new,
                                                   int *__restrict__ flags) {
  int s = blockIdx.x * BLOCK + threadIdx.x;
  if (s >= n) return;
  int p = perm[s];
  int t = tag[p];
  if (B.bpart) {
    if (pack_old) {     // (records of one int4: up to three bonds per bead) the record travels with the bead
      const int4 r4 = pack_old[p];
      pack_new[s] = r4;
      const int rec[4] = {r4.x, r4.y, r4.z, r4.w};
      bond_table_row(s, n, npad, rec, map, B, flags);
    } else bond_table_row(s, n, npad, B.pack + (size_t)t * B.stride, map, B, flags);
  }
  double4 r = pos[p];
  int dix = 0, diy = 0, diz = 0;
  if (wrap) wrap_into_box(r, box, dix, diy, diz);   // the step kernel binned these positions: Domain::pbc is applied here
  pos_new[s] = r;
  xhold[s] = r;
  vxn[s] = vx[p]; vyn[s] = vy[p]; vzn[s] = vz[p];
  tagn[s] = t;
  imgn[s] = img[p] + dix; imgn[npad + s] = img[npad + p] + diy; imgn[2 * npad + s] = img[2 * npad + p] + diz;
  if (!B.bpart) map[t] = s;
  posf[s] = make_float4((float)r.x, (float)r.y, (float)r.z, 0.f);
}
// packed bond records by tag (see DeviceState::bond_pack); run when the bond tables changed
__global__ __launch_bounds__(BLOCK) void k_bond_pack(int maxtag, int bpa, int stride, const int *__restrict__ num_bond,
                                                     const int *__restrict__ bond_type, const int *__restrict__ bond_atom,
                                                     int *__restrict__ pack) {
  int t = blockIdx.x * BLOCK + threadIdx.x;
  if (t < 1 || t > maxtag) return;
  int *rec = pack + (size_t)t * stride;
  const int nb = num_bond[t];
  rec[0] = nb;
  for (int m = 0; m < bpa; m++)
    rec[1 + m] = (m < nb) ? ((max(bond_type[(size_t)t * bpa + m], 0) << BOND_TYPE_SHIFT) | (bond_atom[(size_t)t * bpa + m] & BOND_IDX_MASK)) : 0;   // (a type <= 0 is a bond that is switched off)
}
__global__ __launch_bounds__(BLOCK) void k_bond_table(int n, int npad, const int *__restrict__ tag,
                                                      const int *__restrict__ map, BondTabArgs B,
                                                      int *__restrict__ flags) {
  int s = blockIdx.x * BLOCK + threadIdx.x;
  if (s >= n) return;
  bond_table_row(s, n, npad, B.pack + (size_t)tag[s] * B.stride, map, B, flags);
}
// the packed records by physical index (after the bond tables or the physical order changed outside k_permute)
__global__ __launch_bounds__(BLOCK) void k_bond_pack_phys(int n, int stride, const int *__restrict__ tag, const int *__restrict__ pack,
                                                          int *__restrict__ pack_p) {
  int i = blockIdx.x * BLOCK + threadIdx.x;      // one lane per word
  if (i >= n * stride) return;
  const int s = i / stride, k = i - s * stride;
  pack_p[i] = pack[(size_t)tag[s] * stride + k];
}

// full neighbor list of atom s.  The 27-cell sweep is done as 9 (dy,dz) row segments: the three x-cells of a
// row are one contiguous index range of the cell-sorted arrays (two ranges when the row wraps around the box),
// so a lane streams ~9 candidates per range with independent loads in flight.  Exclusions are tested on
// physical indices (the bead's own special list is translated through map[] once), the minimum image is
// branch-free and skipped by wavefronts that are wholly interior.  Entries come out in (row segment, index)
// order, which depends only on the sorted positions -> deterministic.
#ifndef BUILD_WAVES_PER_SIMD
#define BUILD_WAVES_PER_SIMD 6
#endif
#ifndef BUILD_STAGE_CAP
#define BUILD_STAGE_CAP 128
#endif
constexpr int STAGE_CAP = BUILD_STAGE_CAP;   // float4 slots of one wavefront's staged row interval (2 KB).  128 slots and 80 VGPRs
                                             // give 6 workgroups per CU: 127.5 vs 141 us per build for 192 slots / 88 VGPRs / 5 (96 slots: 135)
constexpr int SPMAX = 4;   // special entries THAT MATTER (weight != 1) translated to indices and kept in registers
static_assert(SPMAX == 4, "neigh_range compares against spi[0..3]");

// The distance test runs in FP32 on a float4 copy of the positions (half the bytes through the texture-address
// path and a quarter of the FP64 issue cycles) and is DECISIVE outside an error band around cutneigh^2; inside the
// band (a fraction ~1e-4 of the candidates) the FP64 test is repeated on the double positions, so the accepted set is
// exactly the FP64 one.  (A variant that used FP32 only to reject, confirming every survivor in FP64, was slower:
// some lane of a wavefront survives in almost every iteration, so both paths executed.)
// Which end of a pair STORES it in the reference's half neighbor list - the end whose special list gives the pair its
// special status (npair_half_bin_newtoff.cpp:90-108, npair_half_bin_newton.cpp:84-149; only asked when the special lists
// are not symmetric, k_build_neigh ASYM):
//   newton_pair off: an owned-owned pair under the lower LOCAL index (`crank`: Atom::sort / data-file order; nullptr: ID - 1);
//     a pair that interacts through a periodic image is an owned-ghost pair on BOTH sides, each end stores its own copy.
//   newton_pair on: the end whose neighbor bin (cutneighmax / 2 bins fitted to the box, nbin_standard.cpp; a ghost's
//     coordinates lie outside the box, nbin.cpp:120-152) comes first in (z, y, x) order; inside one bin the lower local index,
//     and for an owned-ghost pair the owned end iff the ghost is not below / behind / left of it (:85-91).
struct PairOrder {
  const int *crank;
  int newton;
  double lo[3], hi[3], prd[3], half[3], bininv[3];
  int nbin[3];
};
__device__ __forceinline__ int ref_bin(const PairOrder &O, double x, int d) {      // NBin::coord2bin, one dimension
  if (x >= O.hi[d]) return (int)((x - O.hi[d]) * O.bininv[d]) + O.nbin[d];
  if (x >= O.lo[d]) return min((int)((x - O.lo[d]) * O.bininv[d]), O.nbin[d] - 1);
  return (int)((x - O.lo[d]) * O.bininv[d]) - 1;
}
// true: the force on bead s from its neighbor q takes the pair's special status from q's list, not from its own
__device__ __forceinline__ bool pair_stored_by_other(const PairOrder &O, int ts, int tq, const double4 &ri, const double4 &rj) {
  const double pi[3] = {ri.x, ri.y, ri.z};
  double pj[3] = {rj.x, rj.y, rj.z};
  bool image = false;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const double del = pi[d] - pj[d];
    if (del > O.half[d]) { pj[d] += O.prd[d]; image = true; }
    else if (del < -O.half[d]) { pj[d] -= O.prd[d]; image = true; }
  }
  const int li = O.crank ? O.crank[ts] : ts - 1, lj = O.crank ? O.crank[tq] : tq - 1;
  if (!O.newton) return image ? false : lj < li;
#pragma unroll
  for (int d = 2; d >= 0; d--) {
    const int bi = ref_bin(O, pi[d], d), bj = ref_bin(O, pj[d], d);
    if (bi != bj) return bj < bi;
  }
  if (!image) return lj < li;
  // same bin, q is a ghost of s's rank: s stores the pair unless the ghost lies "below" it
  if (pj[2] < pi[2]) return true;
  if (pj[2] == pi[2]) {
    if (pj[1] < pi[1]) return true;
    if (pj[1] == pi[1] && pj[0] < pi[0]) return true;
  }
  return false;
}

template <bool NOSPECIAL, bool MINIMG, bool ASYM, bool STAGED, bool FRAC>
__device__ __forceinline__ void neigh_range(int s, int b, int e, const double4 &ri, const float4 *__restrict__ posf,
                                            float cutf, const double4 *__restrict__ pos,
                                            const int *__restrict__ tag, const Box &box, double cutneighsq, int n1,
                                            int n2, int kmax, int nrel, const int (&spi)[SPMAX],
                                            const int (&spc)[SPMAX],
                                            const int *__restrict__ slist,
                                            int sf1, int sf2, int sf3, int npad, int maxneigh,
                                            int *__restrict__ neigh, const int *__restrict__ all_nspecial,
                                            const int *__restrict__ all_special, int ms_, int &cnt, float bandf,
                                            const float4 *stg, int stg_base, const PairOrder &order) {
#pragma clang fp contract(fast)
  const float rix = (float)ri.x, riy = (float)ri.y, riz = (float)ri.z;
  const float px = (float)box.prd[0], py = (float)box.prd[1], pz = (float)box.prd[2];
  const float ipx = (float)box.iprd[0], ipy = (float)box.iprd[1], ipz = (float)box.iprd[2];
  const float cut_hi = cutf + bandf, cut_lo = cutf - bandf;
  auto dist2 = [&](int q) {
    const float4 rf = STAGED ? stg[q - stg_base] : posf[q];
    float dxf = rix - rf.x, dyf = riy - rf.y, dzf = riz - rf.z;
    if (MINIMG) {
      dxf -= px * __builtin_rintf(dxf * ipx);
      dyf -= py * __builtin_rintf(dyf * ipy);
      dzf -= pz * __builtin_rintf(dzf * ipz);
    }
    return dxf * dxf + dyf * dyf + dzf * dzf;
  };
  // The loop body is straight-line, predicated code: a wavefront executes every statement of it for every candidate slot
  // anyway (some lane survives the distance test in almost every trip), so what counts is the length of the one path, and
  // nested `continue`s cost mask bookkeeping on top.  Only the rare cases branch: the FP64 re-test inside the error band,
  // beads with more relevant special entries than fit in registers, the asymmetric-special-list variant.
  int *out = neigh + (size_t)cnt * npad + s;     // running store address (one 64-bit add per stored entry)
  for (int q = b; q < e; q++) {
    const float rsqf = dist2(q);
    bool keep = !(rsqf > cut_hi) && q != s;
    if (keep && rsqf >= cut_lo) {               // inside the FP32 error band: repeat the test in FP64
      double4 rj = pos[q];
      double delx = ri.x - rj.x, dely = ri.y - rj.y, delz = ri.z - rj.z;
      if (MINIMG) {
        delx -= box.prd[0] * __builtin_rint(delx * box.iprd[0]);
        dely -= box.prd[1] * __builtin_rint(dely * box.iprd[1]);
        delz -= box.prd[2] * __builtin_rint(delz * box.iprd[2]);
      }
      double rsq = delx * delx + dely * dely + delz * delz;
      keep = !(rsq > cutneighsq);
    }
    int entry = q;
    bool own_list = true;
    if (!NOSPECIAL && ASYM) {
      // half-list semantics of the reference: the special list of the end that STORES the pair decides (pair_stored_by_other)
      if (keep) {
        int ts = tag[s], tq = tag[q];
        if (pair_stored_by_other(order, ts, tq, ri, pos[q])) {
          own_list = false;
          const int *ql = all_special + (size_t)tq * ms_;
          int q1 = all_nspecial[3 * (size_t)tq], q2 = all_nspecial[3 * (size_t)tq + 1], q3 = all_nspecial[3 * (size_t)tq + 2];
          int which = 0;
          for (int k = 0; k < q3; k++)
            if (ql[k] == ts) { which = (k < q1) ? 1 : (k < q2) ? 2 : 3; break; }
          if (which) {
            int sf = (which == 1) ? sf1 : (which == 2) ? sf2 : sf3;
            if (sf == 0) keep = false;
            if (sf == 2) entry = q | (which << NEIGH_SB_SHIFT);
          }
        }
      }
    }
    if (!NOSPECIAL) {
      // special entries kept in registers (unused slots hold -1 and match nothing)
      if (!FRAC) {   // no fractional special weight in this run: every kept entry is an exclusion (weight 0.0)
        const bool hit = (spi[0] == q) | (spi[1] == q) | (spi[2] == q) | (spi[3] == q);
        keep = keep && !(hit && own_list);
      } else {
        int code = 0;   // 0 = not special or weight 1; -1 = weight 0 (excluded); 1..3 = level with a fractional weight
#pragma unroll
        for (int k = 0; k < SPMAX; k++) code = (spi[k] == q) ? spc[k] : code;
        if (own_list) { keep = keep && code >= 0; if (code > 0) entry = q | (code << NEIGH_SB_SHIFT); }
      }
      if (nrel > SPMAX && own_list && keep) {   // too many for the registers: walk the bead's special list by tag
        int tq = tag[q];
        for (int k = 0; k < kmax; k++)
          if (slist[k] == tq) {
            int which = (k < n1) ? 1 : (k < n2) ? 2 : 3;
            int sf = (which == 1) ? sf1 : (which == 2) ? sf2 : sf3;
            if (sf == 0) keep = false;
            if (sf == 2) entry = q | (which << NEIGH_SB_SHIFT);
            break;
          }
      }
    }
    if (keep && cnt < maxneigh) *out = entry;
    if (keep) { out += npad; cnt++; }
  }
}

template <bool NOSPECIAL, bool ASYM, bool FRAC>
__device__ __forceinline__ void build_body(int n, int npad, int maxneigh, const double4 *__restrict__ pos,
                                                       const float4 *__restrict__ posf, float cutf, float bandf,
                                                       const int *__restrict__ tag, const int *__restrict__ map,
                                                       const int *__restrict__ cell_start,
                                                       const int *__restrict__ gcell_start, int dd, double zlo_ext,
                                                       int ncx, int ncy, int ncz, double cix, double ciy, double ciz,
                                                       Box box, double cutneighsq, double margin,
                                                       const int *__restrict__ nspecial,
                                                       const int *__restrict__ special, int ms, int sf1, int sf2,
                                                       int sf3, const int *__restrict__ bondtab, int bpa,
                                                       const unsigned long long *__restrict__ bshift,
                                                       int *__restrict__ neigh, int *__restrict__ numneigh,
                                                       int *__restrict__ flags, int diag, const PairOrder &order) {
  int s = blockIdx.x * BLOCK + threadIdx.x;
  bool active = s < n;
  double4 ri = pos[active ? s : 0];
  // `dd`: 0 = one GPU, rows numbered in tiles (bin_inl.h row_id); 1 = decomposed, z-major rows; 2 = one GPU, z-major rows
  const int rtile = (dd == 0) ? ROW_TILE : 0;
  dd = (dd == 1) ? 1 : 0;
  int cx, cy, cz;
  cell_coords(ri, box, ncx, ncy, ncz, cix, ciy, ciz, zlo_ext, cx, cy, cz);
  // special entries whose level carries a weight != 1 (for `special_bonds fene` the 1-2 partners only): translated
  // to physical indices once, compared as integers in the loop
  int n1 = 0, n2 = 0, kmax = 0, nrel = 0;
  const int *slist = nullptr;
  int spi[SPMAX], spc[SPMAX];
#pragma unroll
  for (int k = 0; k < SPMAX; k++) { spi[k] = -1; spc[k] = 0; }
  if (!NOSPECIAL && active && !(diag & 4) && nspecial == nullptr) {
    // `special_bonds fene`-like flags (only the 1-2 level is dropped) with symmetric special lists: the excluded beads are
    // the bond partners, and k_bond_table has just written their physical indices, coalesced, into bpart (handed in
    // through `special`, row count through `ms`) - no walk through the tag-indexed special tables (5 scattered reads per bead)
    int m = 0;
    for (int k = 0; k < ms; k++) {
      const int e = special[(size_t)k * npad + s];
      if (e < 0) continue;
#pragma unroll
      for (int j = 0; j < SPMAX; j++) if (j == m) { spi[j] = e & BOND_IDX_MASK; spc[j] = -1; }
      m++;
    }
    nrel = m;
  } else if (!NOSPECIAL && active && !(diag & 4)) {
    int t = tag[s];
    // (tag-indexed tables, one or two lines per bead out of 100 MB, read once per rebuild: kept out of the caches)
    n1 = __builtin_nontemporal_load(&nspecial[3 * (size_t)t]); n2 = __builtin_nontemporal_load(&nspecial[3 * (size_t)t + 1]);
    int n3 = __builtin_nontemporal_load(&nspecial[3 * (size_t)t + 2]);
    slist = special + (size_t)t * ms;
    kmax = (sf3 != 1) ? n3 : (sf2 != 1) ? n2 : (sf1 != 1) ? n1 : 0;
    nrel = ((sf1 != 1) ? n1 : 0) + ((sf2 != 1) ? n2 - n1 : 0) + ((sf3 != 1) ? n3 - n2 : 0);
    if (nrel <= SPMAX) {
      int m = 0;
      for (int k = 0; k < kmax; k++) {
        int which = (k < n1) ? 1 : (k < n2) ? 2 : 3;
        int sf = (which == 1) ? sf1 : (which == 2) ? sf2 : sf3;
        if (sf == 1) continue;
        int qi = map[__builtin_nontemporal_load(&slist[k])], code = (sf == 0) ? -1 : which;
#pragma unroll
        for (int j = 0; j < SPMAX; j++) if (j == m) { spi[j] = qi; spc[j] = code; }
        m++;
      }
    }
  }
  // positions were wrapped into the box just before this kernel: interior = farther than cutneigh from all faces
  bool interior = ri.x > box.lo[0] + margin && ri.x < box.hi[0] - margin && ri.y > box.lo[1] + margin &&
                  ri.y < box.hi[1] - margin && ri.z > box.lo[2] + margin && ri.z < box.hi[2] - margin;
  bool all_in = __all(interior || !active);      // (decomposed runs too: ghosts are unshifted copies, only beads near a BOX face meet images)
  // Row-segment staging.  The 64 beads of a wavefront are consecutive in cell order, so for one (dy,dz) offset
  // their candidate ranges are overlapping windows of ONE short index interval [B,E) (~64 + one window).  The
  // wavefront copies that interval's float4 positions into LDS with coalesced loads and every lane then walks its
  // own window in LDS.  Left to per-lane global loads, the same lines are requested by many lanes and wavefronts
  // while still in flight: the vector L1 spent 58 % of the kernel in pending-hit stalls (TCP_PENDING_STALL_CYCLES)
  // and the data-return path was 83 % busy.  Wrapped pieces at the row ends, ghost ranges of decomposed runs and
  // intervals longer than STAGE_CAP take the direct path.
  __shared__ float4 s_stage[BLOCK / 64][STAGE_CAP];
  const int lane = threadIdx.x & 63;
  float4 *stg = s_stage[threadIdx.x >> 6];
  const unsigned long long actmask = __ballot(active);
  if (actmask == 0ull) return;
  if (diag & 8) { if (active) numneigh[s] = 0; return; }                       // lanes past the end of an otherwise live wavefront stay as helpers
  const int last = 63 - __clzll((long long)actmask);
  const int maxneigh_w = (diag & 1) ? 0 : maxneigh;   // diagnostics: bit 0 = no entry stores, bit 1 = no candidate loops
  // the bead's bonds open its list: (type, partner index) words of the bond-partner table k_bond_table has just written,
  // compacted; the step kernel evaluates them inside its pipelined neighbor loop (kernels_md.hip pair_loop)
  int cnt = 0;
  if (active)
    for (int k = 0; k < bpa; k++) {
      const int e = bondtab[(size_t)k * npad + s];
      if (e < 0) continue;
      if (cnt < maxneigh_w) neigh[(size_t)cnt * npad + s] = e;
      cnt++;
    }
  const int nbond = cnt;
  int x0 = cx - CELL_XSPLIT, x1 = cx + CELL_XSPLIT;   // x-cell range (>= cutneigh each way), may stick out of [0, ncx)
#define RANGE_T(B, E, STG, SB)                                                                                    \
  do {                                                                                                            \
    if (all_in) neigh_range<NOSPECIAL, false, ASYM, STG, FRAC>(s, B, E, ri, posf, cutf, pos, tag, box, cutneighsq, n1, n2, kmax, nrel, spi, spc, slist, \
                                              sf1, sf2, sf3, npad, maxneigh_w, neigh, nspecial, special, ms, cnt, bandf, stg, SB, order);  \
    else neigh_range<NOSPECIAL, true, ASYM, STG, FRAC>(s, B, E, ri, posf, cutf, pos, tag, box, cutneighsq, n1, n2, kmax, nrel, spi, spc, slist, sf1, sf2, \
                                      sf3, npad, maxneigh_w, neigh, nspecial, special, ms, cnt, bandf, stg, SB, order);  \
  } while (0)
#define RANGE(B, E) RANGE_T(B, E, false, 0)
  // Pass A: the main index range of all nine (dz,dy) rows, 18 independent loads issued together (the rolled loop
  // below would otherwise expose one load round trip per row, and a second one for the staging copy).
  __shared__ int s_rng[2][9][BLOCK];
  const int lo = max(x0, 0), hi = min(x1, ncx - 1);
  auto row_of = [&](int r, bool &zskip) {
    int az = cz + (r / 3 - 1), ay = cy + (r % 3 - 1);
    zskip = false;
    if (dd) zskip = az < 0 || az >= ncz;                      // slab grid: ghosts pad the z direction
    else { if (az < 0) az += ncz; else if (az >= ncz) az -= ncz; }
    az = min(max(az, 0), ncz - 1);
    if (ay < 0) ay += ncy; else if (ay >= ncy) ay -= ncy;
    return row_id(ay, az, ncy, ncz, rtile) * ncx;
  };
  {
    int vb[9], ve[9];
#pragma unroll
    for (int r = 0; r < 9; r++) {
      bool zskip;
      const int row = row_of(r, zskip);
      const bool live = active && !zskip;
      vb[r] = live ? cell_start[row + lo] : 0;
      ve[r] = live ? cell_start[row + hi + 1] : 0;
    }
#pragma unroll
    for (int r = 0; r < 9; r++) { s_rng[0][r][threadIdx.x] = vb[r]; s_rng[1][r][threadIdx.x] = ve[r]; }
  }
  __builtin_amdgcn_wave_barrier();
  if (diag & 16) { if (active) numneigh[s] = s_rng[0][4][threadIdx.x] & 1; return; }
  // Pass B: row r is processed from LDS while the staging loads of row r+1 are in flight
  auto interval = [&](int r, int &mb, int &me, int &B, int &E, bool &fits) {
    mb = s_rng[0][r][threadIdx.x]; me = s_rng[1][r][threadIdx.x];
    // interval of the wavefront: ranges grow with the lane index, so [first lane's begin, last lane's end)
    B = __shfl(mb, 0, 64); E = __shfl(me, last, 64);
    fits = E - B <= STAGE_CAP && __all(me <= mb || (mb >= B && me <= E));
  };
  int mb, me, B, E;
  bool fits;
  interval(0, mb, me, B, E, fits);
  float4 t0, t1, t2 = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int lm = min(max(E - B - 1, 0), npad - 1 - B);      // (an interval that does not fit is not staged: stay in bounds)
    t0 = posf[B + min(lane, lm)]; t1 = posf[B + min(lane + 64, lm)];
    if (STAGE_CAP > 128) t2 = posf[B + min(lane + 128, lm)];
  }
  for (int r = 0; r < 9; r++) {
    bool zskip;
    const int row = row_of(r, zskip);
    const bool live = active && !zskip && !(diag & 2);   // diag bit 1: ranges and staging stay, candidate loops go
    if (fits && !(diag & 128)) {
      const int len = E - B;
      if (lane < len) stg[lane] = t0;
      if (lane + 64 < len) stg[lane + 64] = t1;
      if (STAGE_CAP > 128 && lane + 128 < len) stg[lane + 128] = t2;
    }
    const int cmb = mb, cme = me, cB = B;
    const bool cfits = fits;
    if (r + 1 < 9) {   // next row: its interval is known (pass A), issue its loads now
      interval(r + 1, mb, me, B, E, fits);
      const int lm = min(max(E - B - 1, 0), npad - 1 - B);
      if (!(diag & 64)) { t0 = posf[B + min(lane, lm)]; t1 = posf[B + min(lane + 64, lm)]; if (STAGE_CAP > 128) t2 = posf[B + min(lane + 128, lm)]; }
    }
    __builtin_amdgcn_wave_barrier();                       // LDS ops of one wavefront execute in order
    if (cfits) { if (live) RANGE_T(cmb, cme, true, cB); }
    else if (live) RANGE(cmb, cme);
    __builtin_amdgcn_wave_barrier();
    if (live) {
      if (x1 >= ncx) RANGE(cell_start[row], cell_start[row + x1 - ncx + 1]);           // cells 0.. (images of ncx..x1)
      if (x0 < 0) RANGE(cell_start[row + ncx + x0], cell_start[row + ncx]);            // cells ..ncx-1 (images of x0..-1)
      if (dd) {   // ghost beads of the same cells, stored (cell-sorted) behind the owned beads
        if (x1 >= ncx) RANGE(n + gcell_start[row], n + gcell_start[row + x1 - ncx + 1]);
        RANGE(n + gcell_start[row + lo], n + gcell_start[row + hi + 1]);
        if (x0 < 0) RANGE(n + gcell_start[row + ncx + x0], n + gcell_start[row + ncx]);
      }
    }
  }
#undef RANGE
#undef RANGE_T
  if (!active) return;
  numneigh[s] = min(cnt, maxneigh) | (min(nbond, maxneigh) << NN_BOND_SHIFT) | ((bshift && bshift[s]) ? NN_SHIFTED_BIT : 0);
  // The longest list is only needed when a list did not fit (the host then grows the table to it).  Recording it
  // unconditionally - one atomicMax per wavefront on ONE address - serialises 15.6k read-modify-writes in a single
  // L2 channel: 170 us of a 255 us kernel at 1M beads.
  if (cnt > maxneigh) { flags[FLAG_NEIGH_OVERFLOW] = 1; atomicMax(&flags[FLAG_MAXNEIGH], cnt); }
}

template <bool NOSPECIAL, bool ASYM, bool FRAC>
__global__ __launch_bounds__(BLOCK, BUILD_WAVES_PER_SIMD) void k_build_neigh(int n, int npad, int maxneigh, const double4 *__restrict__ pos,
                                                       const float4 *__restrict__ posf, float cutf, float bandf,
                                                       const int *__restrict__ tag, const int *__restrict__ map,
                                                       const int *__restrict__ cell_start,
                                                       const int *__restrict__ gcell_start, int dd, double zlo_ext,
                                                       int ncx, int ncy, int ncz, double cix, double ciy, double ciz,
                                                       Box box, double cutneighsq, double margin,
                                                       const int *__restrict__ nspecial,
                                                       const int *__restrict__ special, int ms, int sf1, int sf2,
                                                       int sf3, const int *__restrict__ bondtab, int bpa,
                                                       const unsigned long long *__restrict__ bshift,
                                                       int *__restrict__ neigh, int *__restrict__ numneigh,
                                                       int *__restrict__ flags, int diag) {
  build_body<NOSPECIAL, ASYM, FRAC>(n, npad, maxneigh, pos, posf, cutf, bandf, tag, map, cell_start, gcell_start, dd, zlo_ext, ncx, ncy, ncz, cix, ciy, ciz, box, cutneighsq, margin, nspecial, special, ms, sf1, sf2, sf3, bondtab, bpa, bshift, neigh, numneigh, flags, diag, PairOrder{});
}
// asymmetric special lists: the same body with the order of the reference's half list as an argument (a kernel of its own:
// the argument list of k_build_neigh is part of what its schedule was tuned with)
template <bool FRAC>
__global__ __launch_bounds__(BLOCK, BUILD_WAVES_PER_SIMD) void k_build_neigh_asym(int n, int npad, int maxneigh, const double4 *__restrict__ pos,
                                                       const float4 *__restrict__ posf, float cutf, float bandf,
                                                       const int *__restrict__ tag, const int *__restrict__ map,
                                                       const int *__restrict__ cell_start,
                                                       const int *__restrict__ gcell_start, int dd, double zlo_ext,
                                                       int ncx, int ncy, int ncz, double cix, double ciy, double ciz,
                                                       Box box, double cutneighsq, double margin,
                                                       const int *__restrict__ nspecial,
                                                       const int *__restrict__ special, int ms, int sf1, int sf2,
                                                       int sf3, const int *__restrict__ bondtab, int bpa,
                                                       const unsigned long long *__restrict__ bshift,
                                                       int *__restrict__ neigh, int *__restrict__ numneigh,
                                                       int *__restrict__ flags, PairOrder order) {
  build_body<false, true, FRAC>(n, npad, maxneigh, pos, posf, cutf, bandf, tag, map, cell_start, gcell_start, dd, zlo_ext, ncx, ncy, ncz, cix, ciy, ciz, box, cutneighsq, margin, nspecial, special, ms, sf1, sf2, sf3, bondtab, bpa, bshift, neigh, numneigh, flags, 0, order);
}
// same body under a second name: LAMMPS_LE_DIAG_BUILD re-runs the build into scratch outputs with parts switched
// off, so that a profile of a physically unchanged run shows what each part costs
template <bool NOSPECIAL, bool ASYM, bool FRAC>
__global__ __launch_bounds__(BLOCK, BUILD_WAVES_PER_SIMD) void k_build_neigh_diag(int n, int npad, int maxneigh, const double4 *__restrict__ pos,
                                                       const float4 *__restrict__ posf, float cutf, float bandf,
                                                       const int *__restrict__ tag, const int *__restrict__ map,
                                                       const int *__restrict__ cell_start,
                                                       const int *__restrict__ gcell_start, int dd, double zlo_ext,
                                                       int ncx, int ncy, int ncz, double cix, double ciy, double ciz,
                                                       Box box, double cutneighsq, double margin,
                                                       const int *__restrict__ nspecial,
                                                       const int *__restrict__ special, int ms, int sf1, int sf2,
                                                       int sf3, const int *__restrict__ bondtab, int bpa,
                                                       const unsigned long long *__restrict__ bshift,
                                                       int *__restrict__ neigh, int *__restrict__ numneigh,
                                                       int *__restrict__ flags, int diag) {
  build_body<NOSPECIAL, ASYM, FRAC>(n, npad, maxneigh, pos, posf, cutf, bandf, tag, map, cell_start, gcell_start, dd, zlo_ext, ncx, ncy, ncz, cix, ciy, ciz, box, cutneighsq, margin, nspecial, special, ms, sf1, sf2, sf3, bondtab, bpa, bshift, neigh, numneigh, flags, diag, PairOrder{});
}

// phase 1: wrap owned beads, sort them into cell order (ties by ID), permute the physical arrays.
// Decomposed runs pass m_in = slots to bin (kept + gone + arrived) and `gone`; n_out beads remain afterwards.
static void ensure_bond_pack(DeviceState &d) {
  if (!d.bond_pack_dirty) return;
  hipLaunchKernelGGL(k_bond_pack, dim3((d.maxtag + 1 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, d.stream, d.maxtag, d.bpa, d.bond_pack_stride,
                     d.num_bond, d.bond_type, d.bond_atom, d.bond_pack);
  d.bond_pack_dirty = false;
  d.bond_pack_p_valid = false;
}
void scan_cells(DeviceState &d, int *count, int *start, int nc, int total) {
  const int sb = (nc + SCAN_BLOCK - 1) / SCAN_BLOCK;
  hipLaunchKernelGGL(k_scan_local, dim3(sb), dim3(SCAN_BLOCK), 0, d.stream, nc, count, start, d.scan_tmp);
  hipLaunchKernelGGL(k_scan_add, dim3(sb), dim3(SCAN_BLOCK), 0, d.stream, nc, start, d.scan_tmp, total);
}
void launch_sort_owned(DeviceState &d, int m_in, int n_out, const int *gone, bool binned) {
  if (m_in < 0) m_in = n_out = d.n;
  int nb = std::max(1, (m_in + BLOCK - 1) / BLOCK);
  const int nc = d.ncells + (gone ? 1 : 0);     // + sentinel cell
  hipStream_t st = d.stream;
  // the step kernel that produced these positions may already have binned them (cell_of, arrival ranks, counts): then
  // only Domain::pbc is left, and k_permute applies it while it moves the beads
  const bool prebinned = d.bins_ready && !gone && !d.dd && m_in == d.n;
  d.bins_ready = false;
  if (!prebinned && !binned) {
    if (d.cell_count_dirty) HIP_CHECK(hipMemsetAsync(d.cell_count, 0, (size_t)(nc + 1) * sizeof(int), st));   // bins nobody consumed
    hipLaunchKernelGGL(k_wrap_bin, dim3(nb), dim3(BLOCK), 0, st, m_in, d.pos, d.img, d.npad, d.box, d.ncell[0],
                       d.ncell[1], d.ncell[2], d.cellinv[0], d.cellinv[1], d.cellinv[2], d.zlo_ext, d.cell_of,
                       d.cell_count, d.tag_tmp, d.flags, gone, d.ncells, d.row_tile);
  }
  scan_cells(d, d.cell_count, d.cell_start, nc, m_in);
  d.cell_count_dirty = false;      // k_scan_local left the counts at zero
  hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(BLOCK), 0, st, m_in, d.cell_of, d.cell_start, d.tag_tmp, d.perm);
  // one GPU, bonds that need no frozen image: the permute pass also writes the bond-partner table (see k_permute)
  static const bool no_fuse = getenv("LAMMPS_LE_NO_PERMUTE_BONDS") != nullptr;
  const bool with_bonds = !d.dd && !gone && d.bond_minimg && d.bpa > 0 && d.bpart && !no_fuse;
  if (with_bonds) ensure_bond_pack(d);
  static const bool no_phys = getenv("LAMMPS_LE_NO_PHYS_BOND_PACK") != nullptr;
  const bool phys = with_bonds && d.bond_pack_stride == 4 && !no_phys;
  if (phys && !d.bond_pack_p_valid) {
    const int words = m_in * d.bond_pack_stride;
    hipLaunchKernelGGL(k_bond_pack_phys, dim3((words + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, m_in, d.bond_pack_stride, d.tag, d.bond_pack,
                       d.bond_pack_p[0]);
  }
  d.bond_pack_p_valid = phys;
  hipLaunchKernelGGL(k_sort_cells, dim3((d.ncells + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.ncells,
                     d.cell_start, d.perm, d.tag, with_bonds ? d.map : (int *)nullptr);
  const int n = n_out;
  nb = std::max(1, (n + BLOCK - 1) / BLOCK);
  BondTabArgs BT{d.bpa, d.maxtag, d.num_bond, d.bond_type, d.bond_atom, d.bond_pack, d.bond_pack_stride, with_bonds ? d.bpart : (int *)nullptr,
                 nullptr, nullptr, nullptr, 0.0, 0.0, 0.0};
  hipLaunchKernelGGL(k_permute, dim3(nb), dim3(BLOCK), 0, st, n, d.npad, d.perm, d.pos, d.pos_tmp, d.xhold, d.v[0],
                     d.v[1], d.v[2], d.v_tmp[0], d.v_tmp[1], d.v_tmp[2], d.tag, d.tag_tmp, d.img, d.img_tmp, d.map, d.posf,
                     prebinned ? 1 : 0, d.box, BT, phys ? (const int4 *)d.bond_pack_p[0] : (const int4 *)nullptr, (int4 *)d.bond_pack_p[1], d.flags);
  if (phys) std::swap(d.bond_pack_p[0], d.bond_pack_p[1]);
  d.bpart_fresh = with_bonds;
  std::swap(d.pos, d.pos_tmp);
  for (int k = 0; k < 3; k++) std::swap(d.v[k], d.v_tmp[k]);
  std::swap(d.tag, d.tag_tmp);
  std::swap(d.img, d.img_tmp);
}

// phase 2: bond-partner table + full neighbor list of the owned beads (ghosts, if any, already in place)
void launch_lists(DeviceState &d, double cutneighsq, const double sl[4], bool has_pair) {
  int n = d.n, nb = (n + BLOCK - 1) / BLOCK;
  if (nb == 0) nb = 1;
  hipStream_t st = d.stream;
  ensure_bond_pack(d);
  BondTabArgs BT{d.bpa, d.maxtag, d.num_bond, d.bond_type, d.bond_atom, d.bond_pack, d.bond_pack_stride, d.bpart, d.dd ? d.phase : nullptr,
                 d.pos, d.bond_minimg ? nullptr : d.bshift, d.box.half[0], d.box.half[1], d.box.half[2]};
  // (a launch of its own: folded into the prologue of the list build it made that kernel 48 us slower to save 16, and
  // even the unused extra kernel argument cost the build 32 us)
  if (!d.bpart_fresh) hipLaunchKernelGGL(k_bond_table, dim3(nb), dim3(BLOCK), 0, st, n, d.npad, d.tag, d.map, BT, d.flags);
  d.bpart_fresh = false;      // (a second call for the same order - the list table grew - re-derives the same table)
  if (has_pair) {
    const int sf1 = d.sflag[1], sf2 = d.sflag[2], sf3 = d.sflag[3];
    const int ddcode = d.dd ? 1 : (d.row_tile ? 0 : 2);     // see build_body   // Engine::special_flag (lj AND coul weights)
    double margin = sqrt(cutneighsq) * (1.0 + 1e-12);
    // FP32 test: a float coordinate is off by <= M * 2^-24 (M = largest |coordinate|), a separation component
    // (difference, periodic shift with a float box length) by e_d <= 8 * M * 2^-24, the squared distance of a pair
    // near the cutoff by <= 2 * sqrt(3) * r * e_d + rounding; pairs beyond 1.5 * cutneigh are far outside any band
    double cn = sqrt(cutneighsq), M = 0.0;
    for (int k = 0; k < 3; k++) M = std::max({M, fabs(d.box.lo[k]), fabs(d.box.hi[k])});
    double e_d = 8.0 * M * 5.97e-8;
    float cutf = (float)cutneighsq;
    float bandf = (float)(4.0 * 1.5 * cn * e_d + 3.0 * e_d * e_d + 1e-5 * cutneighsq);
    if (getenv("LAMMPS_LE_BUILD_FP64")) bandf = 1e30f;     // diagnostic: every candidate takes the FP64 test
    const bool frac = sf1 == 2 || sf2 == 2 || sf3 == 2;      // some special weight is neither 0 nor 1
    // exclusions = bond partners (`special_bonds fene`-like flags, symmetric lists): read from the bond-partner table
    const bool from_bpart = sf1 == 0 && sf2 == 1 && sf3 == 1 && d.bpa >= 1 && d.bpa <= SPMAX &&
                            !d.flags_h[FLAG_SPECIAL_ASYM] && !getenv("LAMMPS_LE_NO_BPART_EXCL");
    const int *nsp = from_bpart ? (const int *)nullptr : d.nspecial;
    const int *spl = from_bpart ? (const int *)d.bpart : d.special;
    const int msp = from_bpart ? d.bpa : d.maxspecial;
#define BUILD(NOSP, AS, FR)                                                                                        \
  hipLaunchKernelGGL((k_build_neigh<NOSP, AS, FR>), dim3(nb), dim3(BLOCK), 0, st, n, d.npad, d.maxneigh, d.pos, d.posf, cutf, bandf, d.tag, d.map, \
                     d.cell_start, d.gcell_start, ddcode, d.zlo_ext, d.ncell[0], d.ncell[1], d.ncell[2], d.cellinv[0],  \
                     d.cellinv[1], d.cellinv[2], d.box, cutneighsq, margin, nsp, spl, msp, sf1,  \
                     sf2, sf3, d.bpart, d.bpa, d.bond_minimg ? nullptr : d.bshift, d.neigh, d.numneigh, d.flags, 0)
    if (sf1 == 1 && sf2 == 1 && sf3 == 1) BUILD(true, false, false);
    else if (d.flags_h[FLAG_SPECIAL_ASYM]) {      // sticky flag, read back at the last sync
      PairOrder O{};
      O.crank = d.ident_order ? (const int *)nullptr : d.crank;
      O.newton = d.newton_pair;
      for (int k = 0; k < 3; k++) {
        O.lo[k] = d.box.lo[k]; O.hi[k] = d.box.hi[k]; O.prd[k] = d.box.prd[k]; O.half[k] = d.box.half[k];
        O.bininv[k] = d.ref_bininv[k]; O.nbin[k] = d.ref_nbin[k];
      }
#define BUILD_ASYM(FR)                                                                                             \
  hipLaunchKernelGGL((k_build_neigh_asym<FR>), dim3(nb), dim3(BLOCK), 0, st, n, d.npad, d.maxneigh, d.pos, d.posf, cutf, bandf, d.tag, d.map, \
                     d.cell_start, d.gcell_start, ddcode, d.zlo_ext, d.ncell[0], d.ncell[1], d.ncell[2], d.cellinv[0],  \
                     d.cellinv[1], d.cellinv[2], d.box, cutneighsq, margin, nsp, spl, msp, sf1,  \
                     sf2, sf3, d.bpart, d.bpa, d.bond_minimg ? nullptr : d.bshift, d.neigh, d.numneigh, d.flags, O)
      if (frac) BUILD_ASYM(true); else BUILD_ASYM(false);
#undef BUILD_ASYM
    }
    else if (frac) BUILD(false, false, true);
    else BUILD(false, false, false);
#undef BUILD
    if (const char *dg = getenv("LAMMPS_LE_DIAG_BUILD")) {   // diagnostics: extra launch, entry stores off, scratch counters
      hipLaunchKernelGGL((k_build_neigh_diag<false, false, false>), dim3(nb), dim3(BLOCK), 0, st, n, d.npad, d.maxneigh, d.pos, d.posf, cutf, bandf, d.tag, d.map,
                         d.cell_start, d.gcell_start, ddcode, d.zlo_ext, d.ncell[0], d.ncell[1], d.ncell[2], d.cellinv[0],
                         d.cellinv[1], d.cellinv[2], d.box, cutneighsq, margin, d.nspecial, d.special, d.maxspecial, sf1,
                         sf2, sf3, d.bpart, d.bpa, d.bond_minimg ? nullptr : d.bshift, d.neigh, d.cell_of, d.flags + FLAG_AUX - FLAG_MAXNEIGH, atoi(dg) | 1);
    }
  }
}

void launch_reneighbor(DeviceState &d, double cutneighsq, const double sl[4], bool has_pair) {
  launch_sort_owned(d);
  launch_lists(d, cutneighsq, sl, has_pair);
}

}  // namespace lmp_le
