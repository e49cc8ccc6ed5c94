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

namespace lmp_le {

constexpr int BLOCK = 256;
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
                                                             int *__restrict__ flags) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  double4 r = pos[p];
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
                                                           const double *__restrict__ fz, TypeTables tt) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
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
                                                    double *__restrict__ fz, TypeTables tt) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  int type = (int)pos[p].w;
  int t = tag[p];
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
// pair lj/cut over the full ELL neighbor list + per-atom bonds; one thread per bead, no atomics.
template <bool EFLAG, bool HAS_PAIR>
__global__ __launch_bounds__(BLOCK) void k_force(int n, int npad, int nblocks, const double4 *__restrict__ pos,
                                                 const int *__restrict__ neigh, const int *__restrict__ numneigh,
                                                 const int *__restrict__ bpart, int bpa,
                                                 const double *__restrict__ pairtab, int nt, double sl0, double sl1,
                                                 double sl2, double sl3, BondTable bt, Box box,
                                                 double *__restrict__ fx, double *__restrict__ fy,
                                                 double *__restrict__ fz, double *__restrict__ partial,
                                                 int *__restrict__ flags) {
  __shared__ double s_tab[6 * (MAXTYPES + 1) * (MAXTYPES + 1)];
  int nt2 = nt * nt;
  if (HAS_PAIR)
    for (int k = threadIdx.x; k < 6 * nt2; k += BLOCK) s_tab[k] = pairtab[k];
  __syncthreads();
  int lb = logical_block(nblocks);
  int p = lb * BLOCK + threadIdx.x;
  double e[14];
#pragma unroll
  for (int k = 0; k < 14; k++) e[k] = 0.0;
  bool active = (lb < nblocks) && (p < n);
  if (active) {
    double4 ri = pos[p];
    int itype = (int)ri.w;
    double fxi = 0.0, fyi = 0.0, fzi = 0.0;
    const double hx = box.half[0], hy = box.half[1], hz = box.half[2];
    const double px = box.prd[0], py = box.prd[1], pz = box.prd[2];
    if (HAS_PAIR) {
      const double *cutsq = s_tab, *lj1 = s_tab + nt2, *lj2 = s_tab + 2 * nt2, *lj3 = s_tab + 3 * nt2,
                   *lj4 = s_tab + 4 * nt2, *offs = s_tab + 5 * nt2;
      int nn = numneigh[p];
      const int *col = neigh + p;
      int jnext = (nn > 0) ? col[0] : 0;
      for (int k = 0; k < nn; k++) {
        int jraw = jnext;
        if (k + 1 < nn) jnext = col[(size_t)(k + 1) * npad];
        int j = jraw & NEIGH_MASK;
        int sb = (jraw >> NEIGH_SB_SHIFT) & 3;
        double factor_lj = (sb == 0) ? sl0 : (sb == 1) ? sl1 : (sb == 2) ? sl2 : sl3;
        double4 rj = pos[j];
        double delx = ri.x - rj.x, dely = ri.y - rj.y, delz = ri.z - rj.z;
        if (delx > hx) delx -= px; else if (delx < -hx) delx += px;
        if (dely > hy) dely -= py; else if (dely < -hy) dely += py;
        if (delz > hz) delz -= pz; else if (delz < -hz) delz += pz;
        double rsq = delx * delx + dely * dely + delz * delz;
        int ij = itype * nt + (int)rj.w;
        if (rsq < cutsq[ij]) {
          double r2inv = 1.0 / rsq;
          double r6inv = r2inv * r2inv * r2inv;
          double forcelj = r6inv * (lj1[ij] * r6inv - lj2[ij]);
          double fpair = factor_lj * forcelj * r2inv;
          fxi += delx * fpair;
          fyi += dely * fpair;
          fzi += delz * fpair;
          if (EFLAG) {
            double evdwl = r6inv * (lj3[ij] * r6inv - lj4[ij]) - offs[ij];
            evdwl *= factor_lj;
            e[0] += 0.5 * evdwl;
            e[2] += 0.5 * delx * delx * fpair; e[3] += 0.5 * dely * dely * fpair; e[4] += 0.5 * delz * delz * fpair;
            e[5] += 0.5 * delx * dely * fpair; e[6] += 0.5 * delx * delz * fpair; e[7] += 0.5 * dely * delz * fpair;
          }
        }
      }
    }
    for (int m = 0; m < bpa; m++) {
      int eb = bpart[(size_t)m * npad + p];
      if (eb < 0) continue;
      int q = eb & BOND_IDX_MASK, type = eb >> BOND_TYPE_SHIFT;
      int style = bt.style[type];
      if (style == 0) continue;
      double4 rj = pos[q];
      double delx = ri.x - rj.x, dely = ri.y - rj.y, delz = ri.z - rj.z;
      if (delx > hx) delx -= px; else if (delx < -hx) delx += px;
      if (dely > hy) dely -= py; else if (dely < -hy) dely += py;
      if (delz > hz) delz -= pz; else if (delz < -hz) delz += pz;
      double rsq = delx * delx + dely * dely + delz * delz;
      double fbond, ebond = 0.0;
      if (style == 1) {
        double K = bt.p0[type], R0 = bt.p1[type], epsb = bt.p2[type], sigb = bt.p3[type];
        double r0sq = R0 * R0;
        double rlogarg = 1.0 - rsq / r0sq;
        double sr6 = 0.0;
        if (rlogarg < 0.1) {
          // each bond is visited from both ends: count the warning once (lower index)
          if (p < q) atomicAdd(&flags[FLAG_FENE_WARN], 1);
          if (rlogarg <= -3.0) flags[FLAG_ERROR] = ERR_BAD_FENE;
          rlogarg = 0.1;
        }
        fbond = -K / rlogarg;
        if (rsq < TWO_1_3 * sigb * sigb) {
          double sr2 = sigb * sigb / rsq;
          sr6 = sr2 * sr2 * sr2;
          fbond += 48.0 * epsb * sr6 * (sr6 - 0.5) / rsq;
        }
        if (EFLAG) {
          ebond = -0.5 * K * r0sq * log(rlogarg);
          if (rsq < TWO_1_3 * sigb * sigb) ebond += 4.0 * epsb * sr6 * (sr6 - 1.0) + epsb;
        }
      } else {
        double r = sqrt(rsq);
        double dr = r - bt.p1[type];
        double rk = bt.p0[type] * dr;
        fbond = (r > 0.0) ? -2.0 * rk / r : 0.0;
        if (EFLAG) ebond = rk * dr;
      }
      fxi += delx * fbond;
      fyi += dely * fbond;
      fzi += delz * fbond;
      if (EFLAG) {
        e[1] += 0.5 * ebond;
        e[8] += 0.5 * delx * delx * fbond; e[9] += 0.5 * dely * dely * fbond; e[10] += 0.5 * delz * delz * fbond;
        e[11] += 0.5 * delx * dely * fbond; e[12] += 0.5 * delx * delz * fbond; e[13] += 0.5 * dely * delz * fbond;
      }
    }
    fx[p] = fxi; fy[p] = fyi; fz[p] = fzi;
  }
  if (EFLAG && lb < nblocks) block_reduce_store<14>(e, partial, lb, 0);
}

// ------------------------------------------------------------------------------------------
void launch_initial_integrate(DeviceState &d, const TypeTables &tt, double dtv, double triggersq, bool check) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_initial_integrate, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.v[0], d.v[1], d.v[2],
                     d.f[0], d.f[1], d.f[2], d.xhold, tt, dtv, triggersq, check ? 1 : 0, d.flags);
}
void launch_final_integrate(DeviceState &d, const TypeTables &tt) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_final_integrate, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.v[0], d.v[1], d.v[2],
                     d.f[0], d.f[1], d.f[2], tt);
}
void launch_langevin(DeviceState &d, const TypeTables &tt, bool ident, bool fuse_final) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
#define LGV(F, I)                                                                                               \
  hipLaunchKernelGGL((k_langevin<F, I>), dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.tag, d.crank,       \
                     d.rng_out, d.v[0], d.v[1], d.v[2], d.f[0], d.f[1], d.f[2], tt)
  if (fuse_final) { if (ident) LGV(true, true); else LGV(true, false); }
  else { if (ident) LGV(false, true); else LGV(false, false); }
#undef LGV
}
void launch_ke(DeviceState &d, const TypeTables &tt) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_ke, dim3(nb), dim3(BLOCK), 0, d.stream, d.n, d.pos, d.v[0], d.v[1], d.v[2], tt, d.partial);
}
void launch_force(DeviceState &d, const BondTable &bt, const double sl[4], bool eflag, bool has_pair) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  int grid = xcd_grid(nb);
  int nt = d.ntypes + 1;
#define FRC(E, P)                                                                                               \
  hipLaunchKernelGGL((k_force<E, P>), dim3(grid), dim3(BLOCK), 0, d.stream, d.n, d.npad, nb, d.pos, d.neigh,   \
                     d.numneigh, d.bpart, d.bpa, d.pairtab, nt, sl[0], sl[1], sl[2], sl[3], bt, d.box, d.f[0],  \
                     d.f[1], d.f[2], d.partial, d.flags)
  if (eflag) { if (has_pair) FRC(true, true); else FRC(true, false); }
  else { if (has_pair) FRC(false, true); else FRC(false, false); }
#undef FRC
}

// sum the per-block partials on the host in block order (deterministic)
void reduce_partials(DeviceState &d, double *out16) {
  int nb = (d.n + BLOCK - 1) / BLOCK;
  HIP_CHECK(hipMemcpyAsync(d.partial_h, d.partial, (size_t)nb * 16 * sizeof(double), hipMemcpyDeviceToHost,
                           d.stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));
  for (int k = 0; k < 16; k++) out16[k] = 0.0;
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 16; k++) out16[k] += d.partial_h[(size_t)b * 16 + k];
}

}  // namespace lmp_le
