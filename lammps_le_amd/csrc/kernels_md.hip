// kernels_md.hip — per-timestep kernels: integrate, pair+bond forces, Langevin, kinetic energy.
//
// All arithmetic is IEEE double in the reference's operation order (compiled with
// -ffp-contract=off), so a single call differs from the CPU oracle only through the order in
// which a bead's pair/bond contributions are summed.
//
// Reference semantics restated here:
//   FixNVE::initial_integrate / final_integrate   src/fix_nve.cpp:64-104, :108-141
//   PairLJCut::compute                            src/pair_lj_cut.cpp:68-140
//   BondFENE::compute / BondHarmonic::compute     src/MOLECULE/bond_fene.cpp:52-128, bond_harmonic.cpp:48-101
//   FixLangevin::post_force_templated<0,...>      src/fix_langevin.cpp:585-778
//   Neighbor::check_distance                      src/neighbor.cpp:1962-2014
#include "device.h"
#include "bin_inl.h"
#include <hip/hip_ext.h>

namespace lmp_le {

constexpr int BLOCK = 256;
// the throughput variant of k_step is bound by (resident wavefronts) / (lifetime of one wavefront): occupancy was
// varied on purpose with unused LDS and 16 -> 12 -> 8 waves per CU cost x1.22 and x1.79.  Asking the compiler for five
// waves per SIMD makes it fit 92 VGPRs without scratch (it takes 126 when left alone).
#ifndef STEP_STAGE_W
#define STEP_STAGE_W 4
#endif
#ifndef STEP_WAVES_PER_SIMD
#define STEP_WAVES_PER_SIMD 4
#endif
constexpr int AHEAD_MAX_BEADS = 64000;   // k_step: partner / first-stage loads issued ahead of their use up to this size
constexpr int LPB4_MAX_BEADS = 50000;   // k_step: four lanes per bead up to this many (owned) beads, see k_step
#define TWO_1_3 1.2599210498948732

// XCD-aware block remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD, so give
// each XCD one contiguous slab of atoms; its private L2 then holds only that slab's neighborhood.
__device__ __forceinline__ int logical_block(int nblocks_logical) {
  int per = (nblocks_logical + 7) >> 3;
  return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}
static inline int xcd_grid(int nblocks_logical) { return 8 * ((nblocks_logical + 7) / 8); }

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_initial_integrate(int n, double4 *__restrict__ pos,
                                                             double *__restrict__ vx, double *__restrict__ vy,
                                                             double *__restrict__ vz, const double *__restrict__ fx,
                                                             const double *__restrict__ fy,
                                                             const double *__restrict__ fz,
                                                             const double4 *__restrict__ xhold, TypeTables tt,
                                                             double dtv, double triggersq, int check,
                                                             int *__restrict__ flags, const int *__restrict__ tag,
                                                             const int *__restrict__ gmask, int groupbit) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  double4 r = pos[p];
  // fix nve on a group (`mask[i] & groupbit`, src/fix_nve.cpp:82): the others are not moved by THIS fix - the displacement
  // test below (Neighbor::check_distance, every atom) still looks at them: another fix nve may have moved them
  if (!gmask || (gmask[tag[p]] & groupbit)) {
    double dtfm = tt.dtfm[(int)r.w];
    double a = vx[p], b = vy[p], c = vz[p];
    a += dtfm * fx[p];
    b += dtfm * fy[p];
    c += dtfm * fz[p];
    r.x += dtv * a;
    r.y += dtv * b;
    r.z += dtv * c;
    vx[p] = a; vy[p] = b; vz[p] = c;
    pos[p] = r;
  }
  if (check) {
    double4 h = xhold[p];
    double dx = r.x - h.x, dy = r.y - h.y, dz = r.z - h.z;
    double rsq = dx * dx + dy * dy + dz * dz;
    if (rsq > triggersq) flags[FLAG_MOVED] = 1;
  }
}

__global__ __launch_bounds__(BLOCK) void k_final_integrate(int n, const double4 *__restrict__ pos,
                                                           double *__restrict__ vx, double *__restrict__ vy,
                                                           double *__restrict__ vz, const double *__restrict__ fx,
                                                           const double *__restrict__ fy,
                                                           const double *__restrict__ fz, TypeTables tt,
                                                           const int *__restrict__ tag, const int *__restrict__ gmask,
                                                           int groupbit) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  if (gmask && !(gmask[tag[p]] & groupbit)) return;
  double dtfm = tt.dtfm[(int)pos[p].w];
  vx[p] += dtfm * fx[p];
  vy[p] += dtfm * fy[p];
  vz[p] += dtfm * fz[p];
}

// Langevin post_force, optionally fused with FixNVE::final_integrate
template <bool FUSE_FINAL, bool IDENT>
__global__ __launch_bounds__(BLOCK) void k_langevin(int n, const double4 *__restrict__ pos,
                                                    const int *__restrict__ tag, const int *__restrict__ crank,
                                                    const uint32_t *__restrict__ draws, double *__restrict__ vx,
                                                    double *__restrict__ vy, double *__restrict__ vz,
                                                    double *__restrict__ fx, double *__restrict__ fy,
                                                    double *__restrict__ fz, TypeTables tt, const int *__restrict__ gmask,
                                                    int groupbit) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  int type = (int)pos[p].w;
  int t = tag[p];
  // fix langevin on a group: only members draw (src/fix_langevin.cpp:660-661), in local order - `crank` is then the rank
  // among the members (DeviceState::lgrank)
  if (gmask && !(gmask[t] & groupbit)) return;
  int rank = IDENT ? (t - 1) : crank[t];
  double gamma1 = tt.g1[type], gamma2 = tt.g2[type];
  const double inv = 1.0 / 16777216.0;
  double r0 = (double)draws[3 * (size_t)rank] * inv;
  double r1 = (double)draws[3 * (size_t)rank + 1] * inv;
  double r2 = (double)draws[3 * (size_t)rank + 2] * inv;
  double fran0 = gamma2 * (r0 - 0.5), fran1 = gamma2 * (r1 - 0.5), fran2 = gamma2 * (r2 - 0.5);
  double a = vx[p], b = vy[p], c = vz[p];
  double fdrag0 = gamma1 * a, fdrag1 = gamma1 * b, fdrag2 = gamma1 * c;
  double f0 = fx[p], f1 = fy[p], f2 = fz[p];
  f0 += fdrag0 + fran0;
  f1 += fdrag1 + fran1;
  f2 += fdrag2 + fran2;
  fx[p] = f0; fy[p] = f1; fz[p] = f2;
  if (FUSE_FINAL) {
    double dtfm = tt.dtfm[type];
    a += dtfm * f0;
    b += dtfm * f1;
    c += dtfm * f2;
    vx[p] = a; vy[p] = b; vz[p] = c;
  }
}

// ------------------------------------------------------------------------------------------
// block reduction of NV doubles per thread -> partial[block][16]
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&val)[NV], double *__restrict__ partial, int block,
                                                   int col0) {
  __shared__ double red[BLOCK / 64][NV];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = val[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
    for (int w = 0; w < BLOCK / 64; w++) s += red[w][threadIdx.x];
    partial[(size_t)block * 16 + col0 + threadIdx.x] = s;
  }
}

// column sums of a [nb][W] table of block partials in ONE workgroup, the same order on every run: thread (g, c) adds rows
// g, g + G, g + 2G .. of column c (G = BLOCK / W row groups), then the G group sums are added in group order.  A thermo step
// used to copy the table to the host and add it there (0.5 MB + 65k additions at 1M beads: most of the 190 us a thermo step
// spent outside kernels); now 16 doubles cross
template <int W>
__global__ __launch_bounds__(BLOCK) void k_colsum(int nb, const double *__restrict__ in, double *__restrict__ out) {
  constexpr int G = BLOCK / W;
  __shared__ double part[G][W];
  const int c = threadIdx.x % W, g = threadIdx.x / W;
  double s = 0.0;
  for (int b = g; b < nb; b += G) s += in[(size_t)b * W + c];
  part[g][c] = s;
  __syncthreads();
  if (threadIdx.x < W) {
    double t = 0.0;
    for (int k = 0; k < G; k++) t += part[k][threadIdx.x];
    out[threadIdx.x] = t;
  }
}
__global__ __launch_bounds__(BLOCK) void k_ke(int n, const double4 *__restrict__ pos, const double *__restrict__ vx,
                                              const double *__restrict__ vy, const double *__restrict__ vz,
                                              TypeTables tt, double *__restrict__ partial) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  double val[1] = {0.0};
  if (p < n) {
    double a = vx[p], b = vy[p], c = vz[p];
    val[0] = (a * a + b * b + c * c) * tt.mass[(int)pos[p].w];
  }
  block_reduce_store<1>(val, partial, blockIdx.x, 14);
}

// ------------------------------------------------------------------------------------------
// force on bead p: pair lj/cut over the full ELL neighbor list + its own bonds (no atomics: every pair
// and bond is evaluated from both ends, which is also what keeps the summation order deterministic)
struct ForceArgs {
  int n, npad, nblocks, bpa, nt;
  const double4 *pos;
  const int *neigh, *numneigh, *bpart;
  const unsigned long long *bshift;   // frozen partner images of the bonds (DeviceState::bshift)
  int bond_minimg;                    // DeviceState::bond_minimg: per-step minimum image is provably the frozen image
  const double *pairtab;
  double sl0, sl1, sl2, sl3;
  // all type pairs share one coefficient set (pair_coeff * * ...): scalars instead of the LDS table
  int uniform;
  double u_cutsq, u_lj1, u_lj2, u_lj3, u_lj4, u_off;
  int has_sb;            // some special weight is neither 0 nor 1 -> list entries carry special bits
  double margin;         // beads farther than this from every box face need no minimum image
  int maxrow;            // highest list row a lane may touch before it knows its count: (maxneigh - 1 - sub) / LPB >= this
  int nn_limit;          // diagnostics only (LAMMPS_LE_NN_LIMIT): cap on neighbors visited per bead
  int diag;              // diagnostics only (LAMMPS_LE_DIAG_STEP): extra launch with parts off, see launch_step
  // a launch that tests the skin/2 displacement (a rebuild may follow) also bins the new positions: cell, arrival order
  // inside the cell and the cell counts, i.e. everything k_wrap_bin would produce except Domain::pbc itself (k_permute)
  const float4 *holdf;   // float copy of xhold (= posf of the last build; nullptr: use xhold only)
  double hold_band;
  int bin, ncx, ncy, ncz, rtile;
  double cix, ciy, ciz, zlo_ext;
  int *cell_of, *cell_count, *cell_rank;
  // decomposed runs: border beads also write their new position into the halo send buffer (no pack launch per step)
  const int *sendslot;
  double4 *send_dn, *send_up;   // the staging buffer's two halves, or the neighbours' windows (kernels_dd.hip, fast halo)
};

// reciprocal by v_rcp_f64 + two Newton steps (<= 1 ulp from the IEEE quotient 1/x; an IEEE divide is ~35 instructions)
__device__ __forceinline__ double rcp_nr(double x) {
#pragma clang fp contract(fast)
  double r = __builtin_amdgcn_rcp(x);
  r = r * (2.0 - x * r);
  r = r * (2.0 - x * r);
  return r;
}

// bond coefficients in LDS, one row per bond type, with the per-type constants of BondFENE::compute folded once per
// block: a by-value table indexed by a per-lane type is read through memory (a dependent load for every coefficient)
constexpr int BT_W = 8;   // style, K, 1/R0^2 (fene) or r0 (harmonic), 48 eps, sigma^2, 2^(1/3) sigma^2, R0^2, eps
__device__ __forceinline__ void fill_bond_table(const BondTable &bt, double *__restrict__ s_bt) {
  const int t = threadIdx.x;
  if (t <= MAXTYPES) {
    const int st = bt.style[t];
    const double K = bt.p0[t], R0 = bt.p1[t], epsb = bt.p2[t], sigb = bt.p3[t];
    double *row = s_bt + t * BT_W;
    row[0] = (double)st;
    row[1] = K;
    row[2] = (st == 1) ? 1.0 / (R0 * R0) : R0;     // harmonic: r0; morse: alpha
    row[3] = 48.0 * epsb;
    row[4] = sigb * sigb;
    row[5] = TWO_1_3 * sigb * sigb;
    row[6] = (st == 3) ? epsb : R0 * R0;            // morse: r0 (third coefficient)
    row[7] = epsb;
  }
}

// one pair term.  The hot loop is written for issue-bound FP64 on CDNA4: branch-free minimum image
// (v_rndne), reciprocal by v_rcp_f64 + two Newton steps (<= 1 ulp from the reference's IEEE divide),
// FMA contraction allowed inside this function only (everything else is compiled -ffp-contract=off).
template <bool EFLAG, bool MINIMG, bool UNIFORM, bool HAS_SB>
__device__ __forceinline__ void pair_term(const ForceArgs &A, const Box &box, const double *__restrict__ s_tab,
                                          int itype, const double4 &ri, int jraw, const double4 &rj, bool valid,
                                          double &fxi, double &fyi, double &fzi, double (&e)[14]) {
#pragma clang fp contract(fast)
  double delx = ri.x - rj.x, dely = ri.y - rj.y, delz = ri.z - rj.z;
  if (MINIMG) {
    delx -= box.prd[0] * __builtin_rint(delx * box.iprd[0]);
    dely -= box.prd[1] * __builtin_rint(dely * box.iprd[1]);
    delz -= box.prd[2] * __builtin_rint(delz * box.iprd[2]);
  }
  double rsq = delx * delx + dely * dely + delz * delz;
  double cutsq, lj1, lj2, lj3 = 0.0, lj4 = 0.0, offs = 0.0;
  if (UNIFORM) { cutsq = A.u_cutsq; lj1 = A.u_lj1; lj2 = A.u_lj2; if (EFLAG) { lj3 = A.u_lj3; lj4 = A.u_lj4; offs = A.u_off; } }
  else {
    const int nt2 = A.nt * A.nt, ij = itype * A.nt + (int)rj.w;
    cutsq = s_tab[ij]; lj1 = s_tab[nt2 + ij]; lj2 = s_tab[2 * nt2 + ij];
    if (EFLAG) { lj3 = s_tab[3 * nt2 + ij]; lj4 = s_tab[4 * nt2 + ij]; offs = s_tab[5 * nt2 + ij]; }
  }
  if (rsq < cutsq && valid) {
    double r2inv = __builtin_amdgcn_rcp(rsq);
    r2inv = r2inv * (2.0 - rsq * r2inv);
    r2inv = r2inv * (2.0 - rsq * r2inv);
    double r6inv = r2inv * r2inv * r2inv;
    double forcelj = r6inv * (lj1 * r6inv - lj2);
    double fpair = forcelj * r2inv;
    double factor_lj = 1.0;
    if (HAS_SB) {
      int sb = (jraw >> NEIGH_SB_SHIFT) & 3;
      factor_lj = (sb == 0) ? A.sl0 : (sb == 1) ? A.sl1 : (sb == 2) ? A.sl2 : A.sl3;
      fpair *= factor_lj;
    }
    fxi += delx * fpair; fyi += dely * fpair; fzi += delz * fpair;
    if (EFLAG) {
      double evdwl = r6inv * (lj3 * r6inv - lj4) - offs;
      evdwl *= factor_lj;
      e[0] += 0.5 * evdwl;
      e[2] += 0.5 * delx * delx * fpair; e[3] += 0.5 * dely * dely * fpair; e[4] += 0.5 * delz * delz * fpair;
      e[5] += 0.5 * delx * dely * fpair; e[6] += 0.5 * delx * delz * fpair; e[7] += 0.5 * dely * delz * fpair;
    }
  }
}

// everything a bead's force needs that is addressed by the bead's own index alone: fetched in ONE batch at the top of
// the kernel, before any of it is used.  A small system's step time is the chain of dependent loads one wavefront
// walks (every level costs a trip to the memory-side cache: each kernel starts with a cold L2), so the chain is kept
// at own data -> partners' positions -> next list stage.
struct BeadPre {
  int nall;                // list length word: entries | bond entries << NN_BOND_SHIFT (device.h)
  int j0, j1, j2, j3;      // first list stage (rows below maxneigh always exist; masked once the length is known)
};
template <bool HAS_PAIR, int LPB, bool AHEAD>
__device__ __forceinline__ BeadPre bead_preload(const ForceArgs &A, int p, int sub) {
  BeadPre L;
  if (!AHEAD) {            // throughput-bound sizes: loads stay where they are consumed (measured faster at 1M beads)
    L.nall = HAS_PAIR ? A.numneigh[p] : 0;
    L.j0 = L.j1 = L.j2 = L.j3 = p;
    return L;
  }
  if (HAS_PAIR) {
    const size_t st = (size_t)LPB * A.npad;
    const int *col = A.neigh + p + (size_t)sub * A.npad;
    L.nall = A.numneigh[p];
    L.j0 = col[0]; L.j1 = col[min(1, A.maxrow) * st]; L.j2 = col[min(2, A.maxrow) * st]; L.j3 = col[min(3, A.maxrow) * st];
  } else {
    L.nall = 0; L.j0 = L.j1 = L.j2 = L.j3 = p;
  }
  return L;
}

// one bond term seen from bead p (BondFENE::compute / BondHarmonic::compute; every bond is evaluated from both ends).
// `eb` = (bond type << BOND_TYPE_SHIFT) | partner's index, rj = the partner's position.
// `sh` = the partner's periodic image as frozen at the last reneighbor (BSHIFT_BITS of DeviceState::bshift; 0 for nearly
// every bond): the reference evaluates a bond against the ghost Domain::closest_image picked when the bond list was built
// (src/ntopo_bond_all.cpp:52-73), i.e. x_j + S with a fixed S, not against the minimum image of every step.
template <bool EFLAG, bool ALLSTYLES = false>
__device__ __forceinline__ void bond_term(const Box &box, const double *__restrict__ s_bt, int p, const double4 &ri,
                                          int eb, const double4 &rj, unsigned sh, bool minimg, double &fxi, double &fyi,
                                          double &fzi, double (&e)[14], int *__restrict__ flags) {
  const int q = eb & BOND_IDX_MASK, type = eb >> BOND_TYPE_SHIFT;
  const double *row = s_bt + type * BT_W;
  const int style = (int)row[0];
  if (style == 0) return;
  double xj = rj.x, yj = rj.y, zj = rj.z;
  if (sh) {
    if (sh & 1u) xj += box.prd[0]; else if (sh & 2u) xj -= box.prd[0];
    if (sh & 4u) yj += box.prd[1]; else if (sh & 8u) yj -= box.prd[1];
    if (sh & 16u) zj += box.prd[2]; else if (sh & 32u) zj -= box.prd[2];
  }
  double delx = ri.x - xj, dely = ri.y - yj, delz = ri.z - zj;
  if (minimg) {     // (wave-uniform) all-FENE systems in a large box: see DeviceState::bond_minimg
    const double hx = box.half[0], hy = box.half[1], hz = box.half[2];
    if (delx > hx) delx -= box.prd[0]; else if (delx < -hx) delx += box.prd[0];
    if (dely > hy) dely -= box.prd[1]; else if (dely < -hy) dely += box.prd[1];
    if (delz > hz) delz -= box.prd[2]; else if (delz < -hz) delz += box.prd[2];
  }
  double rsq = delx * delx + dely * dely + delz * delz;
  double fbond, ebond = 0.0;
  if (style == 1) {
    // BondFENE::compute with its four divisions done as products with reciprocals (each <= 1 ulp from the quotient)
    const double K = row[1], inv_r0sq = row[2];
    double rlogarg = 1.0 - rsq * inv_r0sq;
    double sr6 = 0.0;
    if (rlogarg < 0.1) {
      // each bond is visited from both ends: count the warning once (lower index)
      if (p < q) atomicAdd(&flags[FLAG_FENE_WARN], 1);
      if (rlogarg <= -3.0) flags[FLAG_ERROR] = ERR_BAD_FENE;
      rlogarg = 0.1;
    }
    fbond = -K * rcp_nr(rlogarg);
    if (rsq < row[5]) {
      const double rinv = rcp_nr(rsq);
      double sr2 = row[4] * rinv;
      sr6 = sr2 * sr2 * sr2;
      fbond += row[3] * sr6 * (sr6 - 0.5) * rinv;
    }
    if (EFLAG) {
      ebond = -0.5 * K * row[6] * log(rlogarg);
      if (rsq < row[5]) ebond += 4.0 * row[7] * sr6 * (sr6 - 1.0) + row[7];
    }
  } else if (ALLSTYLES && style == 3) {
    // BondMorse::compute (src/MOLECULE/bond_morse.cpp:50-115): D = row[1], alpha = row[2], r0 = row[6]
    const double r = sqrt(rsq), dr = r - row[6], ralpha = exp(-row[2] * dr);
    fbond = (r > 0.0) ? -2.0 * row[1] * row[2] * (1 - ralpha) * ralpha / r : 0.0;
    if (EFLAG) ebond = row[1] * (1 - ralpha) * (1 - ralpha);
  } else {
    double r = sqrt(rsq);
    double dr = r - row[2];
    double rk = row[1] * dr;
    fbond = (r > 0.0) ? -2.0 * rk / r : 0.0;
    if (EFLAG) ebond = rk * dr;
  }
  fxi += delx * fbond; fyi += dely * fbond; fzi += delz * fbond;
  if (EFLAG) {
    e[1] += 0.5 * ebond;
    e[8] += 0.5 * delx * delx * fbond; e[9] += 0.5 * dely * dely * fbond; e[10] += 0.5 * delz * delz * fbond;
    e[11] += 0.5 * delx * dely * fbond; e[12] += 0.5 * delx * delz * fbond; e[13] += 0.5 * dely * delz * fbond;
  }
}

// LPB = lanes per bead: lane `sub` of a bead's LPB lanes takes list entries sub, sub + LPB, ... (see k_step).
// The first `nb` entries of a bead's list are its BONDS ((type << BOND_TYPE_SHIFT) | partner, written by the list build
// from the bond-partner table), the pair entries follow: a bonded partner's position travels through the same pipelined
// gathers as the pair partners', instead of a chain of dependent loads (table entry -> position, one bond after the
// other) behind the pair loop.  Stages that hold a bond entry in some lane (the first one; the second for a wavefront
// with an extruder anchor) run the mixed body, all later ones the plain pair body.
template <bool EFLAG, bool MINIMG, bool UNIFORM, bool HAS_SB, int LPB, bool AHEAD, bool DIAGP = false, bool ALLSTYLES = false>
__device__ __forceinline__ void pair_loop(const ForceArgs &A, const Box &box, const double *__restrict__ s_tab,
                                          const double *__restrict__ s_bt, int p, int sub, const BeadPre &L, int nall,
                                          int nb, unsigned long long bsh, const double4 &ri, double &fxi, double &fyi,
                                          double &fzi, double (&e)[14], int *__restrict__ flags) {
  // software pipeline, W neighbors per stage: while the W position gathers of the current stage are in flight the
  // (coalesced) index loads of the next stage are issued, so a stage costs one exposed round trip.  Lists are
  // consumed in groups of W; slots past the end are predicated off (index = own bead, cached).
  constexpr int W = AHEAD ? 4 : STEP_STAGE_W;
  const int itype = (int)ri.w;
  const size_t npad = (size_t)LPB * A.npad;                                             // stride between a lane's entries
  const int *col = A.neigh + p + (size_t)sub * A.npad;
  const int nn = (LPB == 1) ? nall : (nall > sub ? (nall - sub + LPB - 1) / LPB : 0);   // entries of this lane
  const bool bonds_on = !(DIAGP && (A.diag & 1));
  int j[W];
  if (AHEAD) { j[0] = (0 < nn) ? L.j0 : p; j[1] = (1 < nn) ? L.j1 : p; j[2] = (2 < nn) ? L.j2 : p; j[3] = (3 < nn) ? L.j3 : p; }
  else {
#pragma unroll
    for (int u = 0; u < W; u++) j[u] = (u < nn) ? col[u * npad] : p;
  }
  for (int k = 0; k < nn; k += W) {
    if (DIAGP && (A.diag & 8)) {          // diagnostics: same arithmetic, gathers replaced by coalesced loads
#pragma unroll
      for (int u = 0; u < W; u++) j[u] = min(p + k + u + 1, A.n - 1);
    }
    const int g0 = k * LPB + sub;                                                       // list row of this lane's slot 0
    const bool mixed = __any(g0 < nb);                                                  // (bond entries come first)
    double4 r[W];
    int c[W];
    if (mixed) {
#pragma unroll
      for (int u = 0; u < W; u++) r[u] = A.pos[j[u] & (g0 + u * LPB < nb ? BOND_IDX_MASK : NEIGH_MASK)];
    } else {
#pragma unroll
      for (int u = 0; u < W; u++) r[u] = A.pos[j[u] & NEIGH_MASK];
    }
    const int kn = k + W;
#pragma unroll
    for (int u = 0; u < W; u++) { c[u] = j[u]; j[u] = (kn + u < nn) ? col[(size_t)(kn + u) * npad] : p; }
    if (mixed && AHEAD) {
      // small systems (one wavefront's chain of dependent work is what counts, registers are plentiful): unrolled
#pragma unroll
      for (int u = 0; u < W; u++) {
        if (g0 + u * LPB < nb) { if (bonds_on) bond_term<EFLAG, ALLSTYLES>(box, s_bt, p, ri, c[u], r[u], (unsigned)(bsh >> (BSHIFT_BITS * (g0 + u * LPB))) & 63u, A.bond_minimg != 0, fxi, fyi, fzi, e, flags); }
        else pair_term<EFLAG, MINIMG, UNIFORM, HAS_SB>(A, box, s_tab, itype, ri, c[u], r[u], k + u < nn, fxi, fyi, fzi, e);
      }
    } else if (mixed) {
      // rolled over the slots (one copy of each body in the code; unrolled, the two bodies per slot cost ~20 registers)
#pragma unroll 1
      for (int u = 0; u < W; u++) {
        int cu = c[0];
        double4 ru = r[0];
#pragma unroll
        for (int w = 1; w < W; w++) if (u == w) { cu = c[w]; ru = r[w]; }
        if (g0 + u * LPB < nb) { if (bonds_on) bond_term<EFLAG, ALLSTYLES>(box, s_bt, p, ri, cu, ru, (unsigned)(bsh >> (BSHIFT_BITS * (g0 + u * LPB))) & 63u, A.bond_minimg != 0, fxi, fyi, fzi, e, flags); }
        else pair_term<EFLAG, MINIMG, UNIFORM, HAS_SB>(A, box, s_tab, itype, ri, cu, ru, k + u < nn, fxi, fyi, fzi, e);
      }
    } else {
#pragma unroll
      for (int u = 0; u < W; u++)
        pair_term<EFLAG, MINIMG, UNIFORM, HAS_SB>(A, box, s_tab, itype, ri, c[u], r[u], k + u < nn, fxi, fyi, fzi, e);
    }
  }
}

template <bool EFLAG, bool HAS_PAIR, int LPB, bool DIAG, bool AHEAD, bool ALLSTYLES = false>
__device__ __forceinline__ void bead_force(const ForceArgs &A, const BondTable &bt, const Box &box,
                                           const double *__restrict__ s_tab, const double *__restrict__ s_bt, int p,
                                           int sub, const BeadPre &L, const double4 &ri, double &fxi, double &fyi,
                                           double &fzi, double (&e)[14], int *__restrict__ flags) {
  if (HAS_PAIR) {
    const int nall = (DIAG && (A.diag & 2)) ? 0 : min(L.nall & NN_COUNT_MASK, A.nn_limit);
    const int nb = min((L.nall >> NN_BOND_SHIFT) & NN_NBOND_MASK, nall);
    // frozen partner images: a word of the side table for the few beads that have a bond across a periodic face
    unsigned long long bsh = 0ull;
    if (L.nall & NN_SHIFTED_BIT) bsh = A.bshift[p];
    // wave-uniform choice: a wavefront whose 64 beads all sit deeper than `margin` inside the box skips
    // the minimum-image arithmetic of the pair terms (cell order makes most wavefronts interior)
    const double m = A.margin;
    bool interior = ri.x > box.lo[0] + m && ri.x < box.hi[0] - m && ri.y > box.lo[1] + m && ri.y < box.hi[1] - m &&
                    ri.z > box.lo[2] + m && ri.z < box.hi[2] - m;
    bool all_in = __all(interior);
#define LE_LOOP(MI, UN, SB) pair_loop<EFLAG, MI, UN, SB, LPB, AHEAD, DIAG, ALLSTYLES>(A, box, s_tab, s_bt, p, sub, L, nall, nb, bsh, ri, fxi, fyi, fzi, e, flags)
    if (A.uniform && !A.has_sb) {
      if (all_in) LE_LOOP(false, true, false); else LE_LOOP(true, true, false);
    } else if (!A.has_sb) {
      if (all_in) LE_LOOP(false, false, false); else LE_LOOP(true, false, false);
    } else {
      LE_LOOP(true, false, true);
    }
#undef LE_LOOP
    return;
  }
  // no pair style (no list): the bonds come straight from the bond-partner table
  const int nbond = (DIAG && (A.diag & 1)) ? 0 : A.bpa;
  const unsigned long long bsh = A.bond_minimg ? 0ull : A.bshift[p];
  int slot = 0;                                    // the images are stored by compacted slot (k_bond_table)
  for (int m = 0; m < nbond; m++) {
    const int eb = A.bpart[(size_t)m * A.npad + p];
    if (eb < 0) continue;
    const int myslot = slot++;
    if ((m % LPB) != sub) continue;
    const double4 rj = A.pos[eb & BOND_IDX_MASK];
    bond_term<EFLAG, ALLSTYLES>(box, s_bt, p, ri, eb, rj, (unsigned)(bsh >> (BSHIFT_BITS * myslot)) & 63u, A.bond_minimg != 0, fxi, fyi, fzi, e, flags);
  }
}

// forces only (setup, thermo steps, runs without the standard nve+langevin pair of fixes)
// NOBOND: the bond entries at the head of the lists are skipped (run_style respa with pair and bond forces on different levels)
template <bool EFLAG, bool HAS_PAIR, bool NOBOND = false>
__global__ __launch_bounds__(BLOCK) void k_force(ForceArgs A, BondTable bt, Box box, double *__restrict__ fx,
                                                 double *__restrict__ fy, double *__restrict__ fz,
                                                 double *__restrict__ partial, int *__restrict__ flags) {
  __shared__ double s_tab[6 * (MAXTYPES + 1) * (MAXTYPES + 1)];
  __shared__ double s_bt[(MAXTYPES + 1) * BT_W];
  fill_bond_table(bt, s_bt);
  if (HAS_PAIR && !(A.uniform && !A.has_sb))     // one coefficient set and no fractional special weights: scalars, no table
    for (int k = threadIdx.x; k < 6 * A.nt * A.nt; k += BLOCK) s_tab[k] = A.pairtab[k];
  __syncthreads();
  int lb = logical_block(A.nblocks);
  int p = lb * BLOCK + threadIdx.x;
  double e[14];
#pragma unroll
  for (int k = 0; k < 14; k++) e[k] = 0.0;
  if (lb < A.nblocks && p < A.n) {
    double4 ri = A.pos[p];
    const BeadPre L = bead_preload<HAS_PAIR, 1, true>(A, p, 0);
    double fxi = 0.0, fyi = 0.0, fzi = 0.0;
    bead_force<EFLAG, HAS_PAIR, 1, NOBOND, true, true>(A, bt, box, s_tab, s_bt, p, 0, L, ri, fxi, fyi, fzi, e, flags);   // (DIAG = NOBOND: A.diag = 1 switches the bonds off)
    fx[p] = fxi; fy[p] = fyi; fz[p] = fzi;
  }
  if (EFLAG && lb < A.nblocks) block_reduce_store<14>(e, partial, lb, 0);
}

// the whole force-side of a timestep in ONE pass over the beads:
//   f(x_n) -> [+ Langevin drag/random, 3 draws by canonical rank] -> final half-kick (FixNVE::final_integrate)
//   -> [NEXT: first half-kick + drift of step n+1 (FixNVE::initial_integrate) into the second position
//       buffer + the skin/2 displacement test of Neighbor::check_distance]
// x, v never round-trip through HBM between these stages and f is only stored when a later kernel needs it.
// LPB (lanes per bead) = 4 is the small-system variant: below ~10^5 beads the kernel is not bound by throughput but by
// the chain of dependent loads ONE wavefront walks (index -> position, five stages deep for 18 neighbors, then the
// bonds one after the other: 10 us at 32k beads whatever the chip could stream).  Four lanes share a bead, each takes
// every fourth list entry and every fourth bond slot, the partial forces meet in a two-step butterfly and lane 0
// integrates: the chain shrinks to one or two stages.
// ANG: runs with an angle style - the angle forces of this step were written into fx / fy / fz by k_angle<.., OVERWRITE>
// right before this launch and are added to the bead's sums (a template parameter, not a run-time test: the step kernel's
// schedule is sensitive - a pointer test here cost every run 2 us per launch)
#define ANGLE_SMALL 0.001
// the listed angles of one bead (records of k_angle_list): its own share of every one - f1 as atom 1, -(f1 + f3) as the
// centre, f3 as atom 3 - and, EFLAG, a third of the energy / virial each (Angle::ev_tally, newton_bond off)
template <bool EFLAG, typename TABLE>
__device__ __forceinline__ void bead_angles(int p, const double4 &rp, int na, const int4 *__restrict__ rec, int npad,
                                            const double4 *__restrict__ pos, const Box &box, const TABLE &at, double &f0, double &f1v,
                                            double &f2, double (&acc)[8]) {
  for (int m = 0; m < na; m++) {
    const int4 r = rec[(size_t)m * npad + p];
    const int type = r.x;
    if (type <= 0 || at.style[type] == 0) continue;
    const int p1 = r.y, p2 = r.z, p3 = r.w;
    const double4 r1 = (p1 == p) ? rp : pos[p1], r2 = (p2 == p) ? rp : pos[p2], r3 = (p3 == p) ? rp : pos[p3];
    double delx1 = r1.x - r2.x, dely1 = r1.y - r2.y, delz1 = r1.z - r2.z;
    double delx2 = r3.x - r2.x, dely2 = r3.y - r2.y, delz2 = r3.z - r2.z;
    const double hx = box.half[0], hy = box.half[1], hz = box.half[2];
    if (delx1 > hx) delx1 -= box.prd[0]; else if (delx1 < -hx) delx1 += box.prd[0];
    if (dely1 > hy) dely1 -= box.prd[1]; else if (dely1 < -hy) dely1 += box.prd[1];
    if (delz1 > hz) delz1 -= box.prd[2]; else if (delz1 < -hz) delz1 += box.prd[2];
    if (delx2 > hx) delx2 -= box.prd[0]; else if (delx2 < -hx) delx2 += box.prd[0];
    if (dely2 > hy) dely2 -= box.prd[1]; else if (dely2 < -hy) dely2 += box.prd[1];
    if (delz2 > hz) delz2 -= box.prd[2]; else if (delz2 < -hz) delz2 += box.prd[2];
    const double rsq1 = delx1 * delx1 + dely1 * dely1 + delz1 * delz1, ra = sqrt(rsq1);
    const double rsq2 = delx2 * delx2 + dely2 * dely2 + delz2 * delz2, rb = sqrt(rsq2);
    double cs = delx1 * delx2 + dely1 * dely2 + delz1 * delz2;
    cs /= ra * rb;
    if (cs > 1.0) cs = 1.0;
    if (cs < -1.0) cs = -1.0;
    double a, eangle = 0.0;
    if (at.style[type] == 1) {
      double sn = sqrt(1.0 - cs * cs);
      if (sn < ANGLE_SMALL) sn = ANGLE_SMALL;
      sn = 1.0 / sn;
      const double dtheta = acos(cs) - at.theta0[type], tk = at.k[type] * dtheta;
      if (EFLAG) eangle = tk * dtheta;
      a = -2.0 * tk * sn;
    } else {
      if (EFLAG) eangle = at.k[type] * (1.0 + cs);
      a = at.k[type];
    }
    const double a11 = a * cs / rsq1, a12 = -a / (ra * rb), a22 = a * cs / rsq2;
    const double f1x = a11 * delx1 + a12 * delx2, f1y = a11 * dely1 + a12 * dely2, f1z = a11 * delz1 + a12 * delz2;
    const double f3x = a22 * delx2 + a12 * delx1, f3y = a22 * dely2 + a12 * dely1, f3z = a22 * delz2 + a12 * delz1;
    if (p == p1) { f0 += f1x; f1v += f1y; f2 += f1z; }
    else if (p == p2) { f0 -= f1x + f3x; f1v -= f1y + f3y; f2 -= f1z + f3z; }
    else { f0 += f3x; f1v += f3y; f2 += f3z; }
    if (EFLAG) {
      const double third = 1.0 / 3.0;
      acc[0] += third * eangle;
      acc[1] += third * (delx1 * f1x + delx2 * f3x); acc[2] += third * (dely1 * f1y + dely2 * f3y);
      acc[3] += third * (delz1 * f1z + delz2 * f3z); acc[4] += third * (delx1 * f1y + delx2 * f3y);
      acc[5] += third * (delx1 * f1z + delx2 * f3z); acc[6] += third * (dely1 * f1z + dely2 * f3z);
    }
  }
}

// EF: a thermo step - the same pass also sums the energies and the virial of the pair and bond terms into `partial`
// (rows of 16 per block, as k_force<EFLAG> writes them), so that a step with thermo output is not a pass of k_force
// plus a Langevin / integrate kernel.  Only for NEXT = false, one lane per bead (the velocities thermo reads are those
// after final_integrate); the block totals travel in through `pos_next`, which a NEXT = false launch does not use.
// GRP: fix nve and / or fix langevin act on a group (`mask[i] & groupbit`, src/fix_nve.cpp:82, src/fix_langevin.cpp:661): a bead
// outside fix nve's group keeps its position and velocity (its force is still computed - others feel it), a bead outside fix
// langevin's group gets neither drag nor noise and draws nothing; `crank` is then the rank table the draws go by (the rank among
// the members when the thermostat is on a group)
template <bool LANGEVIN, bool NEXT, bool IDENT, bool HAS_PAIR, int LPB, bool DIAG, bool AHEAD, bool ANG, bool EF, bool GRP>
__device__ __forceinline__ void step_body(const ForceArgs &A, const BondTable &bt, const Box &box, const TypeTables &tt,
                                          const double *s_tab, const double *s_bt,
                                          const int *__restrict__ tag, const int *__restrict__ crank,
                                          const uint32_t *__restrict__ draws, double *__restrict__ vx,
                                          double *__restrict__ vy, double *__restrict__ vz,
                                          double *__restrict__ fx, double *__restrict__ fy,
                                          double *__restrict__ fz, double4 *__restrict__ pos_next,
                                          const double4 *__restrict__ xhold, double dtv, double triggersq,
                                          int check, int *__restrict__ flags,
                                          const unsigned char *__restrict__ phase, int which, double (&e)[14], double &ke,
                                          const int *__restrict__ gmask, int nvebit, int lgbit) {
  int lb = logical_block(A.nblocks);
  const int sub = (LPB == 1) ? 0 : (int)(threadIdx.x % LPB);
  int p = lb * (BLOCK / LPB) + threadIdx.x / LPB;
  if (lb >= A.nblocks || p >= A.n) return;         // (the LPB lanes of a bead always leave together)
  if (which >= 0 && phase[p] != which) return;     // decomposed runs: this launch handles one phase of the step
  // a list of the build this launch follows did not fit: the host has not looked yet (Engine::reneighbor defers the
  // check behind this kernel); store nothing, it will rebuild and launch again
  // ---- level 0: all loads addressed by p, issued back to back before any of them is used ----
  double4 ri = A.pos[p];
  double a = vx[p], b = vy[p], c = vz[p];
  int t = 0;
  if (LANGEVIN || GRP) t = tag[p];
  const BeadPre L = bead_preload<HAS_PAIR, LPB, AHEAD>(A, p, sub);
  double4 hold = ri;
  if (AHEAD && NEXT && check) hold = xhold[p];
  const int poisoned = flags[FLAG_NEIGH_OVERFLOW];
  // ---- level 1: the draws (by canonical rank), then - inside bead_force - the partners' positions ----
  uint32_t d0 = 0, d1 = 0, d2 = 0;
  bool m_nve = true, m_lg = true;
  if (GRP) {
    const int gm = gmask[t];
    m_nve = nvebit == 1 || (gm & nvebit);
    m_lg = lgbit == 1 || (gm & lgbit);
  }
  if (LANGEVIN && !(DIAG && (A.diag & 4)) && (!GRP || m_lg)) {
    int rank = IDENT ? (t - 1) : crank[t];
    d0 = draws[3 * (size_t)rank]; d1 = draws[3 * (size_t)rank + 1]; d2 = draws[3 * (size_t)rank + 2];
  }
  double f0 = 0.0, f1 = 0.0, f2 = 0.0;
  bead_force<EF, HAS_PAIR, LPB, DIAG, AHEAD>(A, bt, box, s_tab, s_bt, p, sub, L, ri, f0, f1, f2, e, flags);
  if (LPB > 1) {
#pragma unroll
    for (int o = 1; o < LPB; o <<= 1) { f0 += __shfl_xor(f0, o); f1 += __shfl_xor(f1, o); f2 += __shfl_xor(f2, o); }
    if (sub) return;
  }
  if (ANG && !NEXT) { f0 += fx[p]; f1 += fy[p]; f2 += fz[p]; }     // Angle::compute follows Bond::compute (k_angle wrote them)
  if (ANG && NEXT) {
    // whole steps of a run with an angle style evaluate the bead's listed angles right here; a NEXT launch stores no forces,
    // so its three force pointers carry the angle data instead: records, counts, coefficient table (launch_step)
    double acc8[8];
    const AngleTable &at = *reinterpret_cast<const AngleTable *>(fz);
    bead_angles<false>(p, ri, reinterpret_cast<const int *>(fy)[p], reinterpret_cast<const int4 *>(fx), A.npad, A.pos, box, at, f0, f1, f2, acc8);
  }
  const int type = (int)ri.w;
  if (LANGEVIN && (!GRP || m_lg)) {
    double gamma1 = tt.g1[type], gamma2 = tt.g2[type];
    const double inv = 1.0 / 16777216.0;
    double r0 = (double)d0 * inv;
    double r1 = (double)d1 * inv;
    double r2 = (double)d2 * inv;
    double fran0 = gamma2 * (r0 - 0.5), fran1 = gamma2 * (r1 - 0.5), fran2 = gamma2 * (r2 - 0.5);
    double fdrag0 = gamma1 * a, fdrag1 = gamma1 * b, fdrag2 = gamma1 * c;
    f0 += fdrag0 + fran0;
    f1 += fdrag1 + fran1;
    f2 += fdrag2 + fran2;
  }
  if (poisoned) return;
  const double dtfm = tt.dtfm[type];
  if (!GRP || m_nve) { a += dtfm * f0; b += dtfm * f1; c += dtfm * f2; }          // final_integrate of this step
  if (EF) ke = (a * a + b * b + c * c) * tt.mass[type];    // the kinetic energy thermo reads (k_ke's term, column 14)
  if (NEXT) {
    if (!GRP || m_nve) {
      a += dtfm * f0; b += dtfm * f1; c += dtfm * f2;        // initial_integrate of the next step
      ri.x += dtv * a; ri.y += dtv * b; ri.z += dtv * c;
    }
    pos_next[p] = ri;
    if (A.sendslot) {
      const int sl = A.sendslot[p];
      if (sl >= 0) ((sl >> 30) ? A.send_up : A.send_dn)[sl & ((1 << 30) - 1)] = ri;
    }
    if (check) {
      // Neighbor::check_distance.  Throughput shape: the float copy of the build-time positions (posf, 16 B, what the
      // list build reads anyway) decides unless the squared displacement lies within an error band of skin^2/4; only
      // then is the double reference fetched, so the flag is the one the FP64 test sets (32 MB less per test at 1M).
      bool moved;
      if (!AHEAD && A.holdf) {
        const float4 hf = A.holdf[p];
        double dx = ri.x - (double)hf.x, dy = ri.y - (double)hf.y, dz = ri.z - (double)hf.z;
        const double dsq = dx * dx + dy * dy + dz * dz;
        moved = dsq > triggersq;
        if (fabs(dsq - triggersq) < A.hold_band) {
          const double4 h = xhold[p];
          dx = ri.x - h.x; dy = ri.y - h.y; dz = ri.z - h.z;
          moved = dx * dx + dy * dy + dz * dz > triggersq;
        }
      } else {
        const double4 h = AHEAD ? hold : xhold[p];
        double dx = ri.x - h.x, dy = ri.y - h.y, dz = ri.z - h.z;
        moved = dx * dx + dy * dy + dz * dz > triggersq;
      }
      if (moved && !(DIAG && A.diag)) flags[FLAG_MOVED] = 1;
      if (!DIAG && A.bin) {     // (wave-uniform; the lanes still here: every LPB-th one up to the end of the array)
        double4 w = ri;
        int cell = 0;
        if (isfinite(w.x) && isfinite(w.y) && isfinite(w.z)) {
          int dix, diy, diz;
          wrap_into_box(w, box, dix, diy, diz);
          cell = cell_index(w, box, A.ncx, A.ncy, A.ncz, A.cix, A.ciy, A.ciz, A.zlo_ext, A.rtile);
        } else flags[FLAG_ERROR] = ERR_NONFINITE;
        A.cell_of[p] = cell;
        A.cell_rank[p] = count_into_cell<LPB>(cell, A.cell_count);
      }
    }
  } else {
    fx[p] = f0; fy[p] = f1; fz[p] = f2;
  }
  if (DIAG && (A.diag & 64)) { if (a == 1.2345e300) vx[p] = a; return; }   // diagnostic launch: v is left alone (the value is still computed)
  vx[p] = a; vy[p] = b; vz[p] = c;
}
template <bool LANGEVIN, bool NEXT, bool IDENT, bool HAS_PAIR, int LPB, bool DIAG, bool AHEAD, bool ANG = false, bool EF = false,
          bool GRP = false>
__global__ __launch_bounds__(BLOCK, ((AHEAD || EF) ? 1 : STEP_WAVES_PER_SIMD)) void k_step(ForceArgs A, BondTable bt, Box box, TypeTables tt,
                                                const int *__restrict__ tag, const int *__restrict__ crank,
                                                const uint32_t *__restrict__ draws, double *__restrict__ vx,
                                                double *__restrict__ vy, double *__restrict__ vz,
                                                double *__restrict__ fx, double *__restrict__ fy,
                                                double *__restrict__ fz, double4 *__restrict__ pos_next,
                                                const double4 *__restrict__ xhold, double dtv, double triggersq,
                                                int check, int *__restrict__ flags,
                                                const unsigned char *__restrict__ phase, int which,
                                                const int *__restrict__ gmask, int nvebit, int lgbit) {
  // (the three group arguments trail the list: instantiations without GRP never load them, their code is the one it was)
  static_assert(!EF || (!NEXT && LPB == 1), "energy variant: NEXT = false, one lane per bead");
  static_assert(!GRP || (!IDENT && !EF && !AHEAD && LPB == 1), "group variant: plain shape, ranks from a table");
  __shared__ double s_tab[6 * (MAXTYPES + 1) * (MAXTYPES + 1)];
  __shared__ double s_bt[(MAXTYPES + 1) * BT_W];
  fill_bond_table(bt, s_bt);
  if (HAS_PAIR && !(A.uniform && !A.has_sb))     // one coefficient set and no fractional special weights: scalars, no table
    for (int k = threadIdx.x; k < 6 * A.nt * A.nt; k += BLOCK) s_tab[k] = A.pairtab[k];
  __syncthreads();
  double e[14];
  double kev[1] = {0.0};
  if (EF) {
#pragma unroll
    for (int k = 0; k < 14; k++) e[k] = 0.0;
  }
  step_body<LANGEVIN, NEXT, IDENT, HAS_PAIR, LPB, DIAG, AHEAD, ANG, EF, GRP>(A, bt, box, tt, s_tab, s_bt, tag, crank, draws, vx, vy, vz, fx, fy, fz,
                                                                             pos_next, xhold, dtv, triggersq, check, flags, phase, which, e, kev[0],
                                                                             gmask, nvebit, lgbit);
  if (EF) {
    const int lb = logical_block(A.nblocks);
    if (lb < A.nblocks) {
      block_reduce_store<14>(e, reinterpret_cast<double *>(pos_next), lb, 0);
      block_reduce_store<1>(kev, reinterpret_cast<double *>(pos_next), lb, 14);
    }
  }
}

// ------------------------------------------------------------------------------------------
// `groupbit` != 1: the fix acts on a group (DeviceState::gmask by tag)
void launch_initial_integrate(DeviceState &d, const TypeTables &tt, double dtv, double triggersq, bool check, int groupbit) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  d.bins_ready = false;     // the positions move: bins a step kernel may have left behind are stale
  hipLaunchKernelGGL(k_initial_integrate, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.v[0], d.v[1], d.v[2],
                     d.f[0], d.f[1], d.f[2], d.xhold, tt, dtv, triggersq, check ? 1 : 0, d.flags, d.tag,
                     groupbit != 1 ? d.gmask : (const int *)nullptr, groupbit);
}
void launch_final_integrate(DeviceState &d, const TypeTables &tt, int groupbit) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_final_integrate, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.v[0], d.v[1], d.v[2],
                     d.f[0], d.f[1], d.f[2], tt, d.tag, groupbit != 1 ? d.gmask : (const int *)nullptr, groupbit);
}
// fix langevin `zero yes` (src/fix_langevin.cpp:725-729, 752-772): the members' random forces summed (block sums, then one
// workgroup adds them in block order and divides by the member count), and the mean taken off every member's force
template <bool IDENT>
__global__ __launch_bounds__(BLOCK) void k_langevin_fsum(int n, const double4 *__restrict__ pos, const int *__restrict__ tag,
                                                         const int *__restrict__ crank, const uint32_t *__restrict__ draws,
                                                         TypeTables tt, const int *__restrict__ gmask, int groupbit,
                                                         double *__restrict__ partial) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  double val[3] = {0.0, 0.0, 0.0};
  if (p < n) {
    int t = tag[p];
    if (!gmask || (gmask[t] & groupbit)) {
      int rank = IDENT ? (t - 1) : crank[t];
      double gamma2 = tt.g2[(int)pos[p].w];
      const double inv = 1.0 / 16777216.0;
      for (int k = 0; k < 3; k++) val[k] = gamma2 * ((double)draws[3 * (size_t)rank + k] * inv - 0.5);
    }
  }
  block_reduce_store<3>(val, partial, blockIdx.x, 0);
}
__global__ __launch_bounds__(BLOCK) void k_langevin_fsum_total(int nblocks, double inv_count, double *__restrict__ partial) {
  // one workgroup: thread k < 3 walks column k in block order (the same sum on every run)
  if (threadIdx.x < 3) {
    double s = 0.0;
    for (int b = 0; b < nblocks; b++) s += partial[(size_t)b * 16 + threadIdx.x];
    partial[(size_t)nblocks * 16 + threadIdx.x] = s * inv_count;
  }
}
__global__ __launch_bounds__(BLOCK) void k_langevin_zero(int n, const int *__restrict__ tag, const int *__restrict__ gmask,
                                                         int groupbit, const double *__restrict__ mean,
                                                         double *__restrict__ fx, double *__restrict__ fy,
                                                         double *__restrict__ fz) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  if (gmask && !(gmask[tag[p]] & groupbit)) return;
  fx[p] -= mean[0]; fy[p] -= mean[1]; fz[p] -= mean[2];
}
// `host_sum3` (decomposed runs): this rank's three sums come back to the host instead, nothing is applied; the caller adds
// the ranks' sums and hands the mean to launch_langevin_zero_apply
void launch_langevin_zero(DeviceState &d, const TypeTables &tt, bool ident, int groupbit, long members, double *host_sum3) {
  int nb = std::max(1, (d.n + BLOCK - 1) / BLOCK);
  const int *gm = groupbit != 1 ? d.gmask : (const int *)nullptr;
  const int *rk = groupbit != 1 ? d.lgrank : d.crank;
  if (groupbit != 1) ident = false;
  if (ident)
    hipLaunchKernelGGL((k_langevin_fsum<true>), dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.tag, rk, d.rng_out, tt, gm,
                       groupbit, d.lgsum);
  else
    hipLaunchKernelGGL((k_langevin_fsum<false>), dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.tag, rk, d.rng_out, tt, gm,
                       groupbit, d.lgsum);
  hipLaunchKernelGGL(k_langevin_fsum_total, dim3(1), dim3(BLOCK), 0, d.stream, nb, host_sum3 ? 1.0 : 1.0 / (double)members, d.lgsum);
  if (host_sum3) {
    HIP_CHECK(hipMemcpyAsync(d.partial_h, d.lgsum + (size_t)nb * 16, 3 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
    stream_sync(d);
    for (int k = 0; k < 3; k++) host_sum3[k] = d.partial_h[k];
    return;
  }
  hipLaunchKernelGGL(k_langevin_zero, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.tag, gm, groupbit,
                     d.lgsum + (size_t)nb * 16, d.f[0], d.f[1], d.f[2]);
}
void launch_langevin_zero_apply(DeviceState &d, int groupbit, const double *mean3) {
  int nb = std::max(1, (d.n + BLOCK - 1) / BLOCK);
  const int *gm = groupbit != 1 ? d.gmask : (const int *)nullptr;
  for (int k = 0; k < 3; k++) d.partial_h[k] = mean3[k];
  HIP_CHECK(hipMemcpyAsync(d.lgsum + (size_t)nb * 16, d.partial_h, 3 * sizeof(double), hipMemcpyHostToDevice, d.stream));
  hipLaunchKernelGGL(k_langevin_zero, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.tag, gm, groupbit,
                     d.lgsum + (size_t)nb * 16, d.f[0], d.f[1], d.f[2]);
  stream_sync(d);          // (partial_h is reused by the next reduction)
}
void launch_langevin(DeviceState &d, const TypeTables &tt, bool ident, bool fuse_final, int groupbit) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  const int *gm = groupbit != 1 ? d.gmask : (const int *)nullptr;
  const int *rk = groupbit != 1 ? d.lgrank : d.crank;       // rank among the members of the group
  if (groupbit != 1) ident = false;
#define LGV(F, I)                                                                                               \
  hipLaunchKernelGGL((k_langevin<F, I>), dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.tag, rk,       \
                     d.rng_out, d.v[0], d.v[1], d.v[2], d.f[0], d.f[1], d.f[2], tt, gm, groupbit)
  if (fuse_final) { if (ident) LGV(true, true); else LGV(true, false); }
  else { if (ident) LGV(false, true); else LGV(false, false); }
#undef LGV
}
// run_style respa: FixRespa's per-level force arrays (src/fix_respa.cpp), kept by tag so that the cell sort of a rebuild
// does not have to carry them; to_level: flevel[tag[p]] = f[p], else f[p] = flevel[tag[p]] (+= when `add`)
__global__ __launch_bounds__(BLOCK) void k_flevel_copy(int n, const int *__restrict__ tag, double *__restrict__ fx,
                                                       double *__restrict__ fy, double *__restrict__ fz,
                                                       double *__restrict__ flevel, int to_level, int add) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  double *row = flevel + 3 * (size_t)tag[p];
  if (to_level) { row[0] = fx[p]; row[1] = fy[p]; row[2] = fz[p]; }
  else if (add) { fx[p] += row[0]; fy[p] += row[1]; fz[p] += row[2]; }
  else { fx[p] = row[0]; fy[p] = row[1]; fz[p] = row[2]; }
}
void launch_flevel_copy(DeviceState &d, double *flevel, bool to_level, bool add) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_flevel_copy, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.tag, d.f[0], d.f[1], d.f[2], flevel,
                     to_level ? 1 : 0, add ? 1 : 0);
}
// kinetic-energy tensor sum(m v_i v_j), order xx yy zz xy xz yz (ComputeTemp::compute_vector, src/compute_temp.cpp:108-140):
// block sums into the scratch rows of DeviceState::lgsum, added on the host in block order
__global__ __launch_bounds__(BLOCK) void k_ke_tensor(int n, const double4 *__restrict__ pos, const double *__restrict__ vx,
                                                     const double *__restrict__ vy, const double *__restrict__ vz,
                                                     TypeTables tt, double *__restrict__ partial) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  double val[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (p < n) {
    double a = vx[p], b = vy[p], c = vz[p], m = tt.mass[(int)pos[p].w];
    val[0] = m * a * a; val[1] = m * b * b; val[2] = m * c * c;
    val[3] = m * a * b; val[4] = m * a * c; val[5] = m * b * c;
  }
  block_reduce_store<6>(val, partial, blockIdx.x, 0);
}
void ke_tensor(DeviceState &d, const TypeTables &tt, double *out6) {
  const int nb = std::max(1, (d.n + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL(k_ke_tensor, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.v[0], d.v[1], d.v[2], tt, d.lgsum);
  hipLaunchKernelGGL((k_colsum<16>), dim3(1), dim3(BLOCK), 0, d.stream, nb, d.lgsum, d.lgsum + (size_t)nb * 16);
  HIP_CHECK(hipMemcpyAsync(d.partial_h, d.lgsum + (size_t)nb * 16, 16 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
  stream_sync(d);
  for (int k = 0; k < 6; k++) out6[k] = d.partial_h[k];
}
void launch_ke(DeviceState &d, const TypeTables &tt) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_ke, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.v[0], d.v[1], d.v[2], tt, d.partial);
}
static ForceArgs force_args(DeviceState &d, const double sl[4]) {
  ForceArgs A;
  A.n = d.n; A.npad = d.npad; A.nblocks = (d.n + BLOCK - 1) / BLOCK; A.bpa = d.bpa; A.nt = d.ntypes + 1;
  A.pos = d.pos; A.neigh = d.neigh; A.numneigh = d.numneigh; A.bpart = d.bpart; A.bshift = d.bshift; A.bond_minimg = d.bond_minimg; A.pairtab = d.pairtab;
  A.sl0 = sl[0]; A.sl1 = sl[1]; A.sl2 = sl[2]; A.sl3 = sl[3];
  A.uniform = d.pair_uniform; A.u_cutsq = d.pair_u[0]; A.u_lj1 = d.pair_u[1]; A.u_lj2 = d.pair_u[2];
  A.u_lj3 = d.pair_u[3]; A.u_lj4 = d.pair_u[4]; A.u_off = d.pair_u[5];
  A.has_sb = 0;     // some list entries carry a special level whose factor is not 1 (incl. lj weight 0 kept for its coul weight)
  for (int k = 1; k <= 3; k++) if (d.sflag[k] == 2) A.has_sb = 1;
  A.margin = d.cutneigh;
  static int lim = getenv("LAMMPS_LE_NN_LIMIT") ? atoi(getenv("LAMMPS_LE_NN_LIMIT")) : (1 << 30);
  A.nn_limit = lim;
  A.maxrow = d.maxneigh - 1;
  A.diag = 0;
  A.sendslot = nullptr; A.send_dn = A.send_up = nullptr;
  {
    // posf = (float)xhold: a coordinate is off by <= M * 2^-24, a squared displacement d^2 <= skin^2/4 .. by
    // <= 2 * sqrt(3) * |d| * e + 3 e^2; |d| <= ~skin near the threshold.  Band = four times that bound.
    double M = 0.0;
    for (int k = 0; k < 3; k++) M = std::max({M, fabs(d.box.lo[k]), fabs(d.box.hi[k])});
    const double e = M * 5.97e-8, dmax = 2.0 * d.cutneigh;
    A.holdf = (d.posf && !d.dd) ? d.posf : nullptr;
    A.hold_band = 4.0 * (2.0 * 1.7320508 * dmax * e + 3.0 * e * e);
  }
  A.bin = 0; A.ncx = d.ncell[0]; A.ncy = d.ncell[1]; A.ncz = d.ncell[2]; A.rtile = d.row_tile;
  A.cix = d.cellinv[0]; A.ciy = d.cellinv[1]; A.ciz = d.cellinv[2]; A.zlo_ext = d.zlo_ext;
  A.cell_of = d.cell_of; A.cell_count = d.cell_count; A.cell_rank = d.tag_tmp;
  return A;
}
// `parts`: 3 = pair + bond (the default), 1 = pair only, 2 = bond only, 0 = nothing (forces zeroed); run_style respa
void launch_force(DeviceState &d, const BondTable &bt, const double sl[4], bool eflag, bool has_pair, int parts) {
  ForceArgs A = force_args(d, sl);
  int grid = xcd_grid(A.nblocks);
  if (!has_pair) parts &= 2;
  if (parts != 3) {
    if (parts == 0) {
      for (int k = 0; k < 3; k++) HIP_CHECK(hipMemsetAsync(d.f[k], 0, (size_t)d.n * sizeof(double), d.stream));
    } else if (parts == 2) {      // bonds only: straight from the bond-partner table
      if (eflag) hipLaunchKernelGGL((k_force<true, false>), dim3(grid), dim3(BLOCK), 0, d.stream, A, bt, d.box, d.f[0], d.f[1], d.f[2], d.partial, d.flags);
      else hipLaunchKernelGGL((k_force<false, false>), dim3(grid), dim3(BLOCK), 0, d.stream, A, bt, d.box, d.f[0], d.f[1], d.f[2], d.partial, d.flags);
    } else {                      // pairs only
      A.diag = 1;
      if (eflag) hipLaunchKernelGGL((k_force<true, true, true>), dim3(grid), dim3(BLOCK), 0, d.stream, A, bt, d.box, d.f[0], d.f[1], d.f[2], d.partial, d.flags);
      else hipLaunchKernelGGL((k_force<false, true, true>), dim3(grid), dim3(BLOCK), 0, d.stream, A, bt, d.box, d.f[0], d.f[1], d.f[2], d.partial, d.flags);
    }
    return;
  }
#define FRC(E, P)                                                                                            \
  hipLaunchKernelGGL((k_force<E, P>), dim3(grid), dim3(BLOCK), 0, d.stream, A, bt, d.box, d.f[0], d.f[1],   \
                     d.f[2], d.partial, d.flags)
  if (eflag) { if (has_pair) FRC(true, true); else FRC(true, false); }
  else { if (has_pair) FRC(false, true); else FRC(false, false); }
#undef FRC
}
// which shape of the step kernel a system of this size takes (environment overrides for experiments)
static bool step_lpb4(const DeviceState &d) {
  static const int lpb_env = getenv("LAMMPS_LE_LPB") ? atoi(getenv("LAMMPS_LE_LPB")) : 0;
  static const int lpb_max_n = getenv("LAMMPS_LE_LPB_MAX_N") ? atoi(getenv("LAMMPS_LE_LPB_MAX_N")) : LPB4_MAX_BEADS;
  return lpb_env ? lpb_env == 4 : d.n <= lpb_max_n;
}
static bool step_ahead(const DeviceState &d) {
  static const int ahead_max_n = getenv("LAMMPS_LE_AHEAD_MAX_N") ? atoi(getenv("LAMMPS_LE_AHEAD_MAX_N")) : AHEAD_MAX_BEADS;
  return d.n <= ahead_max_n;
}
// a thermo step as ONE pass (k_step<.., EF>): one GPU, a pair style, the throughput shape of the kernel (one lane per bead)
bool step_fuses_energy(const DeviceState &d, bool has_pair) {
  static const bool off = getenv("LAMMPS_LE_NO_FUSED_THERMO") != nullptr;
  return has_pair && !d.dd && !step_lpb4(d) && !step_ahead(d) && !off;
}
// fix nve / fix langevin on groups inside the step kernel (GRP instantiations): a pair style; with an angle style only the
// throughput shape (the look-ahead shape of small systems has no group + angle instantiation)
bool step_fuses_groups(const DeviceState &d, bool has_pair, bool angles) {
  static const bool off = getenv("LAMMPS_LE_NO_FUSED_GROUPS") != nullptr;
  return has_pair && !off && !(angles && step_ahead(d));
}
// fused force + Langevin + final_integrate [+ next initial_integrate]; swaps the position buffers when `next`
void launch_step(DeviceState &d, const BondTable &bt, const double sl[4], const TypeTables &tt, bool langevin,
                 bool next, bool ident, bool has_pair, double dtv, double triggersq, bool check, hipEvent_t ev_start,
                 hipEvent_t ev_stop, int which, bool swap_buffers, bool angle_forces, bool eflag, int nvebit, int lgbit) {
  ForceArgs A = force_args(d, sl);
  const bool grp = nvebit != 1 || lgbit != 1;        // fix nve / fix langevin on a group: the GRP instantiations
  if (grp && (eflag || !has_pair || !d.gmask)) throw LammpsError("internal: group variant of the fused step asked for a launch it does not cover");
  if (d.dd && next && d.sendslot && !d.sendslot_fallback) {
    A.sendslot = d.sendslot;
    if (d.fast_halo && d.direct_recv && which < 0) {
      // the neighbours' windows, in their sorted ghost order: what I send down arrives there "from above" and vice versa
      const int parity = (int)((d.halo_seq + 1u) & 1u);
      A.send_dn = d.peer_win[0] + (size_t)(parity * 2 + 1) * d.halo_cap;
      A.send_up = d.peer_win[1] + (size_t)(parity * 2 + 0) * d.halo_cap;
      d.packed_peer = parity + 1;
      d.packed_ahead = false;
    } else {
      A.send_dn = d.sendbuf; A.send_up = d.sendbuf + d.nsend[0];
      d.packed_ahead = true;
      d.packed_peer = 0;
    }
  }
  // lanes per bead: 4 while the launch is latency-bound (few wavefronts per SIMD), 1 once it is throughput-bound
  const bool lpb4 = step_lpb4(d) && !angle_forces && !grp;     // (angle / group runs: one lane per bead, see step_fuses_angles)
  if (grp && angle_forces && step_ahead(d)) throw LammpsError("internal: group variant with angles covers the throughput shape only");
  if (angle_forces && !has_pair) throw LammpsError("internal: fused angle step without a pair style");
  const bool ahead = step_ahead(d) && !grp;
  if (eflag && (next || angle_forces || which >= 0 || !step_fuses_energy(d, has_pair)))
    throw LammpsError("internal: energy variant of the fused step asked for a launch it does not cover");
  // diagnostics only (LAMMPS_LE_STEP_LDS_PAD=bytes): unused dynamic LDS per workgroup, to lower the occupancy on purpose
  static const unsigned lds_pad = getenv("LAMMPS_LE_STEP_LDS_PAD") ? (unsigned)atoi(getenv("LAMMPS_LE_STEP_LDS_PAD")) : 0u;
  if (lpb4) { A.nblocks = (d.n + BLOCK / 4 - 1) / (BLOCK / 4); A.maxrow = (d.maxneigh - 4) / 4; }
  // positions binned by this launch: single GPU, whole-step launch with a displacement test in it, one lane per bead (with
  // four lanes per bead - small systems - it was measured slower than the separate k_wrap_bin: 62.2k vs 64.5k steps/s at 32k)
  static const bool no_fused_bin = getenv("LAMMPS_LE_NO_FUSED_BIN") != nullptr;
  const bool bin = check && next && !d.dd && which < 0 && !lpb4 && !no_fused_bin && d.cell_count && d.ncells > 0;
  if (bin) {
    // the counts are zero unless an earlier launch binned and no rebuild consumed them (the scan zeroes what it reads)
    if (d.cell_count_dirty) HIP_CHECK(hipMemsetAsync(d.cell_count, 0, (size_t)(d.ncells + 1) * sizeof(int), d.stream));
    d.cell_count_dirty = true;
    A.bin = 1;
  }
  d.bins_ready = bin;
  int grid = xcd_grid(A.nblocks);
  // ev_start / ev_stop (sampled launches only) take the kernel's own begin / end timestamps from its dispatch packet,
  // the same clock rocprofv3 --kernel-trace reports
  // a whole step of a run with an angle style (NEXT) evaluates the listed angles inside the kernel: records, counts and the
  // coefficient table travel in the three force pointers, which a NEXT launch does not otherwise use (step_body)
  double *pf0 = d.f[0], *pf1 = d.f[1], *pf2 = d.f[2];
  if (angle_forces && next) {
    if (!d.eff_rec || !d.eff_n || !d.angtab_dev) throw LammpsError("internal: fused angle step without angle records");
    pf0 = reinterpret_cast<double *>(d.eff_rec); pf1 = reinterpret_cast<double *>(d.eff_n); pf2 = reinterpret_cast<double *>(d.angtab_dev);
  }
#define STPL(L, N, I, P, W, D, H) STPA(L, N, I, P, W, D, H, false)
#define STPA(L, N, I, P, W, D, H, G)                                                                         \
  hipExtLaunchKernelGGL((k_step<L, N, I, P, W, D, H, G>), dim3(grid), dim3(BLOCK), lds_pad, d.stream, ev_start, ev_stop, 0, A, bt, \
                        d.box, tt, d.tag, d.crank, d.rng_out, d.v[0], d.v[1], d.v[2], pf0, pf1, pf2,    \
                        d.pos_tmp, d.xhold, dtv, triggersq, check ? 1 : 0, d.flags, d.phase, which, (const int *)nullptr, 1, 1)
#define STP(L, N, I, P) do { if (angle_forces) { if (ahead) STPA(L, N, I, true, 1, false, true, true); else STPA(L, N, I, true, 1, false, false, true); } \
    else if (lpb4) STPL(L, N, I, P, 4, false, true); else if (ahead) STPL(L, N, I, P, 1, false, true); else STPL(L, N, I, P, 1, false, false); } while (0)
  int key = (langevin ? 8 : 0) | (next ? 4 : 0) | (ident ? 2 : 0) | (has_pair ? 1 : 0);
  if (eflag) {      // thermo step: energies and virial in the same pass (block totals through the unused pos_next argument)
#define STPE(L, I)                                                                                              \
  hipExtLaunchKernelGGL((k_step<L, false, I, true, 1, false, false, false, true>), dim3(grid), dim3(BLOCK), lds_pad, d.stream, ev_start, \
                        ev_stop, 0, A, bt, d.box, tt, d.tag, d.crank, d.rng_out, d.v[0], d.v[1], d.v[2], d.f[0], d.f[1], d.f[2],  \
                        reinterpret_cast<double4 *>(d.partial), d.xhold, dtv, triggersq, 0, d.flags, d.phase, which, (const int *)nullptr, 1, 1)
    if (langevin) { if (ident) STPE(true, true); else STPE(true, false); }
    else { if (ident) STPE(false, true); else STPE(false, false); }
#undef STPE
    return;
  }
  // LAMMPS_LE_DIAG_STEP=bits: the same kernel is launched once more BEFORE the real launch with parts switched off
  // (1 bonds, 2 pair loop, 4 draws, 8 pair gathers replaced by coalesced loads, 128 nothing); it writes only the second position buffer, which the real launch
  // overwrites, so the run is physically unchanged and a kernel trace shows what each part costs
  static const int diag_step = getenv("LAMMPS_LE_DIAG_STEP") ? atoi(getenv("LAMMPS_LE_DIAG_STEP")) : 0;
  if (diag_step && key == 15 && which < 0 && !lpb4) {
    ForceArgs R = A;
    A.diag = diag_step | 64; A.bin = 0;
    hipEvent_t e0 = ev_start, e1 = ev_stop;
    ev_start = ev_stop = nullptr;
    STPL(true, true, true, true, 1, true, false);
    A = R; ev_start = e0; ev_stop = e1;
  }
  if (grp) {      // one lane per bead, ranks from a table: the draws of a thermostat on a group go by the rank among its members
    const int *ranks = d.lg_grouped ? d.lgrank : d.crank;
#define STPG(L, N, G)                                                                                           \
  hipExtLaunchKernelGGL((k_step<L, N, false, true, 1, false, false, G, false, true>), dim3(grid), dim3(BLOCK), lds_pad, d.stream, \
                        ev_start, ev_stop, 0, A, bt, d.box, tt, d.tag, ranks, d.rng_out, d.v[0], d.v[1], d.v[2], pf0, pf1, \
                        pf2, d.pos_tmp, d.xhold, dtv, triggersq, check ? 1 : 0, d.flags, d.phase, which, d.gmask, nvebit, lgbit)
#define STPGA(L, N) do { if (angle_forces) STPG(L, N, true); else STPG(L, N, false); } while (0)
    if (langevin) { if (next) STPGA(true, true); else STPGA(true, false); }
    else { if (next) STPGA(false, true); else STPGA(false, false); }
#undef STPGA
#undef STPG
    if (next && swap_buffers) std::swap(d.pos, d.pos_tmp);
    return;
  }
  switch (key) {
    case 0: STP(false, false, false, false); break;  case 1: STP(false, false, false, true); break;
    case 2: STP(false, false, true, false); break;   case 3: STP(false, false, true, true); break;
    case 4: STP(false, true, false, false); break;   case 5: STP(false, true, false, true); break;
    case 6: STP(false, true, true, false); break;    case 7: STP(false, true, true, true); break;
    case 8: STP(true, false, false, false); break;   case 9: STP(true, false, false, true); break;
    case 10: STP(true, false, true, false); break;   case 11: STP(true, false, true, true); break;
    case 12: STP(true, true, false, false); break;   case 13: STP(true, true, false, true); break;
    case 14: STP(true, true, true, false); break;    case 15: STP(true, true, true, true); break;
  }
#undef STP
#undef STPL
#undef STPA
  if (next && swap_buffers) std::swap(d.pos, d.pos_tmp);
}

// ------------------------------------------------------------------------------------------
// Angle forces (angle_style harmonic | cosine): AngleHarmonic::compute / AngleCosine::compute (src/MOLECULE/
// angle_harmonic.cpp:53-147, angle_cosine.cpp:49-121) as one thread per bead over the LISTED angles the bead is part of
// (k_angle_list above).  A bead evaluates every such angle and keeps
// only its own share - f1 as atom 1, -(f1 + f3) as the centre, f3 as atom 3 - so no atomics are needed and the order of
// a bead's sum is fixed; the two other atoms are found through map[] (tag -> index).  Energy and virial: a third of each
// per stored copy, which is Angle::ev_tally with newton_bond off (src/angle.cpp:164-250).  The two arms are taken by
// minimum image (they are a bond long; the reference uses the images closest to the listing atom at the last reneighbor).
// The angle list (NTopoAngleAll::build with newton_bond off, src/ntopo_angle_all.cpp:55-76): the copy atom i stores is
// listed iff i has the lowest local index of the three atoms, and a listed angle moves all three.  One thread per atom
// appends its listed copies to the three atoms' records; a second kernel sorts each atom's records (central atom, ends,
// type), so that what a bead sums, and in which order, depends on nothing but the topology.  An angle across a periodic
// face: seen from the listing atom i the atoms on the other side are ghost images (Domain::closest_image, index >= nlocal),
// which never block a listing and which the listing does not move (newton_bond off: forces go to owned atoms only,
// angle_harmonic.cpp:121-137) - so the copy on i is listed iff i has the lowest local index among the atoms on ITS side, and
// its record goes to those atoms only; the atoms on the other side are moved by the listings of their own copies.  With all
// copies in place that adds up to one evaluation per atom, with copies out of step it is what the reference computes
// (found by the mixed sweep, tests/test_gpu_fuzz2.py seeds 17, 74, 125).
// (type, a1, a2, a3) of the first ANGLE_PACK_COLS stored angles of every atom, column-major by tag (empty slots: type 0)
__global__ __launch_bounds__(BLOCK) void k_angle_pack(int T, int apa, int stride, const int *__restrict__ num_angle,
                                                      const int *__restrict__ angle_type, const int *__restrict__ a1,
                                                      const int *__restrict__ a2, const int *__restrict__ a3, int4 *__restrict__ pack) {
  const int i = blockIdx.x * BLOCK + threadIdx.x + 1;
  if (i > T) return;
  const int na = num_angle[i];
  for (int m = 0; m < ANGLE_PACK_COLS; m++) {
    const size_t c = (size_t)i * apa + m;
    pack[(size_t)m * stride + i] = m < na ? make_int4(angle_type[c], a1[c], a2[c], a3[c]) : make_int4(0, 0, 0, 0);
  }
}
__global__ __launch_bounds__(BLOCK) void k_angle_list(int T, int apa, int ecap, int npad, const int *__restrict__ crank,
                                                      const int *__restrict__ map, const float4 *__restrict__ pos, Box box, int n_owned,
                                                      const int *__restrict__ num_angle, const int *__restrict__ angle_type,
                                                      const int *__restrict__ a1, const int *__restrict__ a2,
                                                      const int *__restrict__ a3, int *__restrict__ eff_n,
                                                      int4 *__restrict__ eff_rec, int *__restrict__ flags,
                                                      const int4 *__restrict__ pack, int pack_stride) {
  const int i = blockIdx.x * BLOCK + threadIdx.x + 1;
  if (i > T) return;
  const int na = num_angle[i];
  const int li = crank ? crank[i] : i;
  for (int m = 0; m < na; m++) {
    const size_t c = (size_t)i * apa + m;
    // (the first columns come from the packed records - coalesced 16-byte loads -, the tables of stride apa only beyond them)
    int4 src;
    if (m < ANGLE_PACK_COLS) src = pack[(size_t)m * pack_stride + i];
    else src = make_int4(angle_type[c], a1[c], a2[c], a3[c]);
    const int t[3] = {src.y, src.z, src.w};
    if (n_owned < 0) {
      // one GPU: two of an angle's three stored copies are not listed (the listing atom must have the lowest local index among
      // the atoms on its side), and which ones is known from the ranks alone unless an atom with a lower rank sits across a
      // periodic face - so only those atoms are looked up before the copy is dropped (the kernel is bound by its gathers)
      int rk[3];
      bool lower = false;
      for (int q = 0; q < 3; q++) { rk[q] = crank ? crank[t[q]] : t[q]; lower = lower || li > rk[q]; }
      const int pi = map[i];
      if (pi < 0) { flags[FLAG_ERROR] = ERR_BOND_MISSING; continue; }
      const float4 ri = pos[pi];
      int p[3] = {-1, -1, -1};
      bool ghost[3] = {false, false, false}, known[3] = {false, false, false}, listed = true;
      auto look = [&](int q) {
        p[q] = (t[q] == i) ? pi : map[t[q]];
        known[q] = true;
        if (p[q] < 0) return false;
        const float4 rq = pos[p[q]];
        ghost[q] = fabs((double)ri.x - (double)rq.x) > box.half[0] || fabs((double)ri.y - (double)rq.y) > box.half[1] ||
                   fabs((double)ri.z - (double)rq.z) > box.half[2];
        return true;
      };
      bool missing = false;
      if (lower)
        for (int q = 0; q < 3 && listed; q++)
          if (li > rk[q]) { if (!look(q)) { missing = true; break; } listed = ghost[q]; }
      if (missing) { flags[FLAG_ERROR] = ERR_BOND_MISSING; continue; }
      if (!listed) continue;
      for (int q = 0; q < 3; q++) if (!known[q] && !look(q)) missing = true;
      if (missing) { flags[FLAG_ERROR] = ERR_BOND_MISSING; continue; }
      const int4 rec = make_int4(src.x, p[0], p[1], p[2]);
      for (int q = 0; q < 3; q++) {
        if (ghost[q]) continue;
        const int slot = atomicAdd(&eff_n[p[q]], 1);
        if (slot >= ecap) { flags[FLAG_ERROR] = ERR_ANGLES; continue; }
        eff_rec[(size_t)slot * npad + p[q]] = rec;
      }
      continue;
    }
    // records hold PHYSICAL indices (this list lives until the next reneighbor, like the indices) and sit column-major by
    // the bead's own index: the force kernel reads them coalesced and gathers positions without a tag -> index lookup
    const int p[3] = {map[t[0]], map[t[1]], map[t[2]]};
    const int pi = map[i];
    // decomposed runs (n_owned >= 0; the angle tables are replicated): only angles that move a bead this rank OWNS matter here;
    // all three atoms of such an angle must be present (owned or ghost: the ghost shell of a run with angles holds every bead
    // within the ghost cutoff of a face, k_dd_borders)
    if (n_owned >= 0 && !((p[0] >= 0 && p[0] < n_owned) || (p[1] >= 0 && p[1] < n_owned) || (p[2] >= 0 && p[2] < n_owned))) continue;
    if (p[0] < 0 || p[1] < 0 || p[2] < 0 || pi < 0) { flags[FLAG_ERROR] = ERR_BOND_MISSING; continue; }
    // (the float copy of the build-time positions decides "more than half a box away": an angle's arms are a few sigma long)
    const float4 ri = pos[pi];
    bool listed = true, ghost[3];
    for (int q = 0; q < 3; q++) {
      const float4 rq = pos[p[q]];
      ghost[q] = fabs((double)ri.x - (double)rq.x) > box.half[0] || fabs((double)ri.y - (double)rq.y) > box.half[1] ||
                 fabs((double)ri.z - (double)rq.z) > box.half[2];
      listed = listed && (ghost[q] || li <= (crank ? crank[t[q]] : t[q]));
    }
    if (!listed) continue;
    const int4 rec = make_int4(src.x, p[0], p[1], p[2]);
    for (int q = 0; q < 3; q++) {
      if (ghost[q] || (n_owned >= 0 && p[q] >= n_owned)) continue;
      const int slot = atomicAdd(&eff_n[p[q]], 1);
      if (slot >= ecap) { flags[FLAG_ERROR] = ERR_ANGLES; continue; }
      eff_rec[(size_t)slot * npad + p[q]] = rec;
    }
  }
}
__global__ __launch_bounds__(BLOCK) void k_angle_sort(int n, int ecap, int npad, int *__restrict__ eff_n, int4 *__restrict__ eff_rec) {
  const int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  const int cnt = min(eff_n[p], ecap);
  eff_n[p] = cnt;
  if (cnt < 2) return;
  auto less = [](const int4 &x, const int4 &y) {       // (type, i1, i2, i3) = (x, y, z, w): centre, ends, type
    const int xl = min(x.y, x.w), xh = max(x.y, x.w), yl = min(y.y, y.w), yh = max(y.y, y.w);
    if (x.z != y.z) return x.z < y.z;
    if (xl != yl) return xl < yl;
    if (xh != yh) return xh < yh;
    return x.x < y.x;
  };
  for (int a = 1; a < cnt; a++) {
    const int4 key = eff_rec[(size_t)a * npad + p];
    int b = a - 1;
    while (b >= 0) {
      const int4 cur = eff_rec[(size_t)b * npad + p];
      if (!less(key, cur)) break;
      eff_rec[(size_t)(b + 1) * npad + p] = cur;
      b--;
    }
    eff_rec[(size_t)(b + 1) * npad + p] = key;
  }
}
void launch_angle_list(DeviceState &d) {
  if (d.apa <= 0) return;
  const int T = d.maxtag, nb = std::max(1, (T + BLOCK - 1) / BLOCK);
  HIP_CHECK(hipMemsetAsync(d.eff_n, 0, (size_t)d.npad * sizeof(int), d.stream));
  const int pack_stride = T + 2;
  if (d.angle_pack_dirty) {       // (the angle tables change at an LE firing with `atype` / angle breaking, not at a rebuild)
    hipLaunchKernelGGL(k_angle_pack, dim3(nb), dim3(BLOCK), 0, d.stream, T, d.apa, pack_stride, d.num_angle, d.angle_type, d.angle_a1,
                       d.angle_a2, d.angle_a3, (int4 *)d.angle_pack);
    d.angle_pack_dirty = false;
  }
  hipLaunchKernelGGL(k_angle_list, dim3(nb), dim3(BLOCK), 0, d.stream, T, d.apa, d.ecap, d.npad, d.ident_order ? (const int *)nullptr : d.crank,
                     d.map, d.posf, d.box, d.dd ? d.n : -1, d.num_angle, d.angle_type, d.angle_a1, d.angle_a2, d.angle_a3, d.eff_n, (int4 *)d.eff_rec, d.flags,
                     (const int4 *)d.angle_pack, pack_stride);
  hipLaunchKernelGGL(k_angle_sort, dim3(std::max(1, (d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, d.stream, d.n, d.ecap, d.npad, d.eff_n,
                     (int4 *)d.eff_rec);
}

template <bool EFLAG, bool OVERWRITE = false>
__global__ __launch_bounds__(BLOCK) void k_angle(int n, int ecap, Box box, AngleTable at, const double4 *__restrict__ pos,
                                                 int npad, const int *__restrict__ num_angle, const int4 *__restrict__ rec,
                                                 double *__restrict__ fx,
                                                 double *__restrict__ fy, double *__restrict__ fz,
                                                 double *__restrict__ partial_a, int *__restrict__ flags) {
  const int p = blockIdx.x * BLOCK + threadIdx.x;
  double acc[8];
#pragma unroll
  for (int k = 0; k < 8; k++) acc[k] = 0.0;
  if (p < n) {
    const int na = num_angle[p];
    const double4 rp = pos[p];
    double f0 = 0.0, f1v = 0.0, f2 = 0.0;
    bead_angles<EFLAG>(p, rp, na, rec, npad, pos, box, at, f0, f1v, f2, acc);
    if (OVERWRITE) { fx[p] = f0; fy[p] = f1v; fz[p] = f2; }      // the fused step kernel adds them to its own sums
    else { fx[p] += f0; fy[p] += f1v; fz[p] += f2; }
  }
  if (EFLAG) {
    __shared__ double red[BLOCK / 64][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      double v = acc[k];
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      double v = 0.0;
      for (int w = 0; w < BLOCK / 64; w++) v += red[w][threadIdx.x];
      partial_a[(size_t)blockIdx.x * 8 + threadIdx.x] = v;
    }
  }
}
// the coefficient table in device memory, for the fused step (kernel arguments of k_step are not to be touched)
void upload_angle_table(DeviceState &d, const AngleTable &at) {
  if (!d.angtab_dev) HIP_CHECK(hipMalloc((void **)&d.angtab_dev, sizeof(AngleTable)));
  HIP_CHECK(hipMemcpyAsync(d.angtab_dev, &at, sizeof(AngleTable), hipMemcpyHostToDevice, d.stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));      // (`at` is the caller's object)
}
void launch_angle(DeviceState &d, const AngleTable &at, bool eflag, bool overwrite) {
  if (d.apa <= 0) return;
  const int nb = std::max(1, (d.n + BLOCK - 1) / BLOCK);
  if (overwrite)
    hipLaunchKernelGGL((k_angle<false, true>), dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.ecap, d.box, at, d.pos, d.npad, d.eff_n,
                       (const int4 *)d.eff_rec, d.f[0], d.f[1], d.f[2], d.partial_a, d.flags);
  else if (eflag)
    hipLaunchKernelGGL((k_angle<true>), dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.ecap, d.box, at, d.pos, d.npad, d.eff_n,
                       (const int4 *)d.eff_rec, d.f[0], d.f[1], d.f[2], d.partial_a, d.flags);
  else
    hipLaunchKernelGGL((k_angle<false>), dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.ecap, d.box, at, d.pos, d.npad, d.eff_n,
                       (const int4 *)d.eff_rec, d.f[0], d.f[1], d.f[2], d.partial_a, d.flags);
}
void reduce_angle_partials(DeviceState &d, double *out8) {
  const int nb = std::max(1, (d.n + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL((k_colsum<8>), dim3(1), dim3(BLOCK), 0, d.stream, nb, d.partial_a, d.partial_a + (size_t)nb * 8);
  HIP_CHECK(hipMemcpyAsync(d.partial_h, d.partial_a + (size_t)nb * 8, 8 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
  stream_sync(d);
  for (int k = 0; k < 8; k++) out8[k] = d.partial_h[k];
}

// can the fused step kernel take the angle forces of a run (launch_step's angle_forces)?  Needs the pair-style instantiations
bool step_fuses_angles(const DeviceState &d, bool has_pair) { (void)d; return has_pair; }

// sum the per-block partials in a fixed order (deterministic); the totals land in the table's spare row nb
void reduce_partials(DeviceState &d, double *out16) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL((k_colsum<16>), dim3(1), dim3(BLOCK), 0, d.stream, nb, d.partial, d.partial + (size_t)nb * 16);
  HIP_CHECK(hipMemcpyAsync(d.partial_h, d.partial + (size_t)nb * 16, 16 * sizeof(double), hipMemcpyDeviceToHost, d.stream));
  stream_sync(d);
  for (int k = 0; k < 16; k++) out16[k] = d.partial_h[k];
}

}  // namespace lmp_le
