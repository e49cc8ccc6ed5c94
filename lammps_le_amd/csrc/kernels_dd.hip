// kernels_dd.hip — spatial decomposition across the GPUs of one node.
//
// Reference counterpart: CommBrick::exchange / borders / forward_comm (src/comm_brick.cpp:452-876).  The design is
// not the reference's 6-swap brick scheme:
//   * z-slab decomposition: with <= 8 GPUs and the ghost cutoff of this model a slab's halo is as large as a
//     2x2x2 brick's, and every rank has exactly two neighbours -> two grouped ncclSend/ncclRecv pairs per step
//     over direct xGMI links, no forwarding of corner ghosts.
//   * ghosts carry their owner's wrapped coordinates unshifted; the force kernels apply the minimum image
//     (x, y are fully periodic on every rank), so no shifted copies and no per-swap PBC arithmetic.
//   * topology (bonds, specials, extruder table) is replicated and tag-indexed; only positions travel.
//   * migration and ghost lists are rebuilt at reneighbor time on the device (flag + scan + scatter).
#include <unistd.h>

#include <cstring>

#include "bin_inl.h"
#include "comm.h"
#include "device.h"

namespace lmp_le {

constexpr int BLOCK = 256;
constexpr int MIG_W = 12;    // doubles per migrating bead: x y z type vx vy vz tag ix iy iz pad
constexpr int GATH_W = 14;   // doubles per bead in whole-system gathers: tag x y z type vx vy vz fx fy fz ix iy iz
constexpr int GATH_LE_W = 8; // ... of the LE fixes' firing-step gather: tag x y z type xhold[3]

void dd_halo_wait(DeviceState &d);

// signed z offset from the bottom of my slab, wrapped to [-Lz/2, Lz/2)
__device__ __forceinline__ double zrel_slab(double z, double slab_lo, const Box &box) {
  double zc = z - slab_lo;
  if (zc >= box.half[2]) zc -= box.prd[2];
  if (zc < -box.half[2]) zc += box.prd[2];
  return zc;
}

// slot for every lane with `pred` set: one atomic per wavefront (lanes of a wavefront get consecutive slots)
__device__ __forceinline__ int wave_append(bool pred, int *__restrict__ counter) {
  unsigned long long m = __ballot(pred);
  if (m == 0ull) return 0;
  int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1, base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader, 64);
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

// ---- migration: wrap owned beads (Domain::pbc, src/domain.cpp:528-645); a bead whose slab is no longer mine is
// packed for the lower / upper neighbour (slots from wave-aggregated atomics: leavers are few) and marked `gone`.
// Kept beads are not moved here: they are binned (cell, arrival order, counts) for the cell sort that follows, the
// `gone` ones into a sentinel cell behind all real cells, which compacts the arrays for free.  The same pass takes the
// tags of every slot this rank held - owned and ghost - out of map[]: what the rebuild then puts back (k_permute the owned
// beads, k_dd_ghost_place the ghosts) is exactly the new set, without a fill of the whole tag range per rebuild.
// counters: flags[COUNT_B] = sent down, [NDRAW] = sent up.
struct BinArgs {
  int ncx, ncy, ncz;
  double cix, ciy, ciz, zlo_ext;
  int *cell_of, *cell_count, *rank;
  int sentinel;
};
__global__ __launch_bounds__(BLOCK) void k_dd_leave(int n, int nslots, int npad, int migcap, Box box, double slab_lo, double width,
                                                    int me, int P, double4 *__restrict__ pos,
                                                    const double *__restrict__ vx, const double *__restrict__ vy,
                                                    const double *__restrict__ vz, const int *__restrict__ tag,
                                                    int *__restrict__ img, double *__restrict__ mig_dn,
                                                    double *__restrict__ mig_up, int *__restrict__ gone,
                                                    int *__restrict__ map, BinArgs B, int *__restrict__ flags) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  bool active = p < n;
  if (p < nslots) map[tag[p]] = -1;
  double4 r = pos[active ? p : 0];
  int im[3] = {0, 0, 0};
  bool godn = false, goup = false;
  if (active) {
    bool bad = !(isfinite(r.x) && isfinite(r.y) && isfinite(r.z));
    if (bad) flags[FLAG_ERROR] = ERR_NONFINITE;
    double *c = &r.x;
    bool wrapped = false;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      double x = c[d];
      int i = img[d * npad + p];
      if (x < box.lo[d]) { x += box.prd[d]; i--; wrapped = true; }
      if (x >= box.hi[d]) { x -= box.prd[d]; x = fmax(x, box.lo[d]); i++; wrapped = true; }
      c[d] = x;
      im[d] = i;
    }
    if (wrapped) { pos[p] = r; img[p] = im[0]; img[npad + p] = im[1]; img[2 * npad + p] = im[2]; }
    if (!bad) {
      // owner = slab index from one expression that every rank evaluates identically
      int owner = (int)((r.z - box.lo[2]) / width);
      owner = min(max(owner, 0), P - 1);
      double zc = zrel_slab(r.z, slab_lo, box);      // direction of travel for a bead that left
      godn = owner != me && zc < 0.0;
      goup = owner != me && zc >= 0.0;
    }
    gone[p] = (godn || goup) ? 1 : 0;
    // (active lanes are a prefix of the wavefront, as count_into_cell asks)
    int cell = bad ? 0 : cell_index(r, box, B.ncx, B.ncy, B.ncz, B.cix, B.ciy, B.ciz, B.zlo_ext, 0);
    if (godn || goup) cell = B.sentinel;
    B.cell_of[p] = cell;
    B.rank[p] = count_into_cell(cell, B.cell_count);
  }
  int sd = wave_append(godn, &flags[FLAG_COUNT_B]);
  int su = wave_append(goup, &flags[FLAG_NDRAW]);
  if (godn || goup) {
    int slot = godn ? sd : su;
    if (slot >= migcap) return;                    // reported by the host from the counters
    double *b = (godn ? mig_dn : mig_up) + (size_t)slot * MIG_W;
    b[0] = r.x; b[1] = r.y; b[2] = r.z; b[3] = r.w; b[4] = vx[p]; b[5] = vy[p]; b[6] = vz[p];
    b[7] = (double)tag[p]; b[8] = (double)im[0]; b[9] = (double)im[1]; b[10] = (double)im[2];
    b[11] = 0.0;
  }
}
__global__ __launch_bounds__(BLOCK) void k_dd_arrive(int narr, int base, int npad, const double *__restrict__ in,
                                                     double4 *__restrict__ pos, double *__restrict__ vx,
                                                     double *__restrict__ vy, double *__restrict__ vz,
                                                     int *__restrict__ tag, int *__restrict__ img,
                                                     int *__restrict__ gone, Box box, BinArgs B) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= narr) return;
  gone[base + i] = 0;
  const double *b = in + (size_t)i * MIG_W;
  int s = base + i;
  const double4 r = make_double4(b[0], b[1], b[2], b[3]);     // (wrapped by its sender)
  pos[s] = r;
  vx[s] = b[4]; vy[s] = b[5]; vz[s] = b[6];
  tag[s] = (int)b[7];
  img[s] = (int)b[8]; img[npad + s] = (int)b[9]; img[2 * npad + s] = (int)b[10];
  const int cell = cell_index(r, box, B.ncx, B.ncy, B.ncz, B.cix, B.ciy, B.ciz, B.zlo_ext, 0);
  B.cell_of[s] = cell;
  B.rank[s] = count_into_cell(cell, B.cell_count);
}

// ---- borders (after the cell sort): the send lists hold (a) every owned bead within the PAIR shell (rc + skin) of
// my lower / upper z face and (b) beads within the full ghost cutoff (`comm_modify cutoff`, which the reference
// needs for long extruder bonds) that actually have a bond partner on another rank.  The reference ghosts the
// whole `comm_modify cutoff` shell (src/comm_brick.cpp:600-720); with cutoff 5.0 and 13-sigma slabs that is 76 %
// of a slab per step over xGMI, against 23 % here — the forces are the same because only bond partners are ever
// looked up beyond the pair shell.  Slots: wave-aggregated atomics; the receiver sorts its ghosts by cell and ID,
// so the list order is free.
__global__ __launch_bounds__(BLOCK) void k_dd_borders(int n, const double4 *__restrict__ pos, Box box, double slab_lo,
                                                      double width, double cutpair, double cutghost, int bpa,
                                                      const int *__restrict__ tag, const int *__restrict__ map,
                                                      const int *__restrict__ num_bond,
                                                      const int *__restrict__ bond_atom, int *__restrict__ list_dn,
                                                      int *__restrict__ list_up, int *__restrict__ flags,
                                                      unsigned char *__restrict__ phase, int *__restrict__ sendslot,
                                                      RngValidateArgs V, int whole_shell) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  bool active = p < n;
  if (active && V.late) rng_validate_bead(V, tag[p], flags);      // do this rank's pools hold the draws of what it owns now?
  double zc = active ? zrel_slab(pos[p].z, slab_lo, box) : 0.0;
  bool dn = active && zc < cutpair, up = active && zc >= width - cutpair;
  bool far_dn = active && !dn && zc < cutghost, far_up = active && !up && zc >= width - cutghost;
  if (far_dn || far_up) {
    int t = tag[p], nb = num_bond[t];
    // (runs with an angle style: the whole ghost-cutoff shell, as the reference sends it - an angle's far end is two bonds
    //  away from the bead it moves and need not be bonded to anything on the other side)
    bool remote = whole_shell != 0;      // map[] holds owned beads only at this point
    for (int m = 0; m < nb; m++) remote = remote || map[bond_atom[(size_t)t * bpa + m]] < 0;
    dn = dn || (far_dn && remote);
    up = up || (far_up && remote);
  }
  int sd = wave_append(dn, &flags[FLAG_COUNT_A]);
  int su = wave_append(up, &flags[FLAG_COUNT_B]);
  if (dn) list_dn[sd] = p;
  if (up) list_up[su] = p;
  // the fused step kernel packs a border bead's new position itself (no pack launch per step); a slab is at least two
  // ghost shells thick, so a bead is in at most one list - if it ever is in both, -2 makes the halo fall back to packing
  if (active) sendslot[p] = (dn && up) ? -2 : dn ? sd : up ? (su | (1 << 30)) : -1;
  if (dn && up) flags[FLAG_ERROR] = ERR_GHOST_ORDER;
  // sent beads are phase 1: a bead with a ghost NEIGHBOR lies within the pair shell of a face and is therefore sent;
  // the bond-table kernel adds the few beads whose bond partner is a ghost
  if (active) phase[p] = (dn || up) ? 1 : 0;
}
// halo pack of both send lists in one launch (also used for the initial border exchange together with the tags)
__global__ __launch_bounds__(BLOCK) void k_dd_pack(int m0, int m1, const int *__restrict__ list0,
                                                   const int *__restrict__ list1, const double4 *__restrict__ pos,
                                                   double4 *__restrict__ out) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < m0 + m1) out[i] = pos[i < m0 ? list0[i] : list1[i - m0]];
}
// the first exchange after a rebuild: positions and tags of both send lists
__global__ __launch_bounds__(BLOCK) void k_dd_pack_xt(int m0, int m1, const int *__restrict__ list0,
                                                      const int *__restrict__ list1, const double4 *__restrict__ pos,
                                                      const int *__restrict__ tag, double4 *__restrict__ out,
                                                      int *__restrict__ out_tag) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m0 + m1) return;
  const int p = i < m0 ? list0[i] : list1[i - m0];
  out[i] = pos[p];
  out_tag[i] = tag[p];
}
__global__ __launch_bounds__(BLOCK) void k_dd_unpack(int m, int base, const int *__restrict__ gdest,
                                                     const double4 *__restrict__ in, double4 *__restrict__ pos) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < m) pos[base + gdest[i]] = in[i];
}

// ---- ghosts into cell order ----
__global__ __launch_bounds__(BLOCK) void k_dd_ghost_bin(int m, const double4 *__restrict__ in, Box box, int ncx, int ncy,
                                                        int ncz, double cix, double ciy, double ciz, double zlo_ext,
                                                        int *__restrict__ cell_of, int *__restrict__ count,
                                                        int *__restrict__ rank) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  double4 r = in[i];
  double zrel = r.z - zlo_ext;
  if (zrel < 0.0) zrel += box.prd[2];
  if (zrel >= box.prd[2]) zrel -= box.prd[2];
  int cx = (int)((r.x - box.lo[0]) * cix), cy = (int)((r.y - box.lo[1]) * ciy), cz = (int)(zrel * ciz);
  cx = min(max(cx, 0), ncx - 1); cy = min(max(cy, 0), ncy - 1); cz = min(max(cz, 0), ncz - 1);
  int c = (cz * ncy + cy) * ncx + cx;
  cell_of[i] = c;
  rank[i] = atomicAdd(&count[c], 1);
}
// slot of arrival i inside its cell = number of same-cell arrivals with a smaller tag (deterministic)
__global__ __launch_bounds__(BLOCK) void k_dd_ghost_slot(int m, const int *__restrict__ cell_of,
                                                         const int *__restrict__ gstart, const int *__restrict__ rank,
                                                         int *__restrict__ perm) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < m) perm[gstart[cell_of[i]] + rank[i]] = i;
}
__global__ __launch_bounds__(BLOCK) void k_dd_ghost_sort(int ncells, const int *__restrict__ gstart, int *__restrict__ perm,
                                                         const int *__restrict__ tag_in) {
  int c = blockIdx.x * BLOCK + threadIdx.x;
  if (c >= ncells) return;
  int b = gstart[c], e = gstart[c + 1];
  for (int i = b + 1; i < e; i++) {
    int pi = perm[i], ti = tag_in[pi], j = i - 1;
    while (j >= b && tag_in[perm[j]] > ti) { perm[j + 1] = perm[j]; j--; }
    perm[j + 1] = pi;
  }
}
// The ghosts that came from below fill the low z layers of the local grid and those from above the high ones, so
// the cell-sorted ghost array is [from below | from above].  Each sender is told where its k-th border bead ended up
// inside its block (`rel`); it then reorders its send list accordingly, and from then on a halo message is received
// STRAIGHT into the ghost slots: no unpack kernel per step.  arrival layout of the first exchange: [above | below].
__global__ __launch_bounds__(BLOCK) void k_dd_ghost_place(int m, int base, int nabove, int nbelow, const int *__restrict__ perm,
                                                          const double4 *__restrict__ in, const int *__restrict__ tag_in,
                                                          double4 *__restrict__ pos, int *__restrict__ tag,
                                                          int *__restrict__ gdest, int *__restrict__ map,
                                                          float4 *__restrict__ posf, int *__restrict__ rel,
                                                          int *__restrict__ flags) {
  int s = blockIdx.x * BLOCK + threadIdx.x;
  if (s >= m) return;
  int i = perm[s];
  double4 r = in[i];
  pos[base + s] = r;
  posf[base + s] = make_float4((float)r.x, (float)r.y, (float)r.z, 0.f);
  int t = tag_in[i];
  tag[base + s] = t;
  gdest[i] = s;
  map[t] = base + s;
  if (rel) {
    const bool above = i < nabove;               // arrival block
    int q = above ? s - nbelow : s;              // slot inside its sorted block
    if (q < 0 || q >= (above ? nabove : nbelow)) { flags[FLAG_ERROR] = ERR_GHOST_ORDER; q = 0; }
    rel[i] = q;
  }
}
__global__ __launch_bounds__(BLOCK) void k_dd_reorder_sends(int m0, int m1, const int *__restrict__ rel,
                                                            const int *__restrict__ list0, const int *__restrict__ list1,
                                                            int *__restrict__ new0, int *__restrict__ new1,
                                                            int *__restrict__ sendslot) {
  int k = blockIdx.x * BLOCK + threadIdx.x;
  if (k >= m0 + m1) return;
  bool up = k >= m0;
  int p = up ? list1[k - m0] : list0[k], r = rel[k];
  (up ? new1 : new0)[r] = p;
  if (sendslot[p] >= 0) sendslot[p] = up ? (r | (1 << 30)) : r;
}
__global__ __launch_bounds__(BLOCK) void k_fill_int(int n, int *a, int v) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i < n) a[i] = v;
}
__global__ __launch_bounds__(BLOCK) void k_dd_map_owned(int n, const int *__restrict__ tag, int *__restrict__ map) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p < n) map[tag[p]] = p;
}

// ---- whole-system gathers (LE firing steps, host download) ----
__global__ __launch_bounds__(BLOCK) void k_dd_gather_pack(int n, int npad, int stride, const double4 *__restrict__ pos,
                                                          const double4 *__restrict__ xhold, const double *__restrict__ vx,
                                                          const double *__restrict__ vy, const double *__restrict__ vz,
                                                          const double *__restrict__ fx, const double *__restrict__ fy,
                                                          const double *__restrict__ fz, const int *__restrict__ tag,
                                                          const int *__restrict__ img, int mode, double *__restrict__ out) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= stride) return;
  double *b = out + (size_t)p * (mode == 0 ? GATH_LE_W : GATH_W);
  if (p >= n) { b[0] = 0.0; return; }
  double4 r = pos[p];
  b[0] = (double)tag[p]; b[1] = r.x; b[2] = r.y; b[3] = r.z; b[4] = r.w;
  if (mode == 0) {            // LE fixes: current + held positions
    double4 h = xhold[p];
    b[5] = h.x; b[6] = h.y; b[7] = h.z;
  } else {                    // download: v f image
    b[5] = vx[p]; b[6] = vy[p]; b[7] = vz[p]; b[8] = fx[p]; b[9] = fy[p]; b[10] = fz[p];
    b[11] = (double)img[p]; b[12] = (double)img[npad + p]; b[13] = (double)img[2 * npad + p];
  }
}
__global__ __launch_bounds__(BLOCK) void k_dd_scatter_xt(long total, const double *__restrict__ in, double4 *__restrict__ xt,
                                                         double4 *__restrict__ xht) {
  long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= total) return;
  const double *b = in + (size_t)i * GATH_LE_W;
  int t = (int)b[0];
  if (t <= 0) return;
  xt[t] = make_double4(b[1], b[2], b[3], b[4]);
  xht[t] = make_double4(b[5], b[6], b[7], 0.0);
}

// =============================================================================================
static void ensure_gather(DeviceState &d, size_t doubles_per_rank, int world) {
  size_t need = doubles_per_rank * (size_t)(world + 1);
  if (need > d.gather_cap) {
    // Regrow: no stream of this process may still touch the old buffer.  A peer rank that is a thread of this process copies
    // out of it on ITS stream (comm.cpp local_exchange); the sender's host has seen that copy enqueued (the acknowledgement)
    // but not finished, so the whole device is drained before the buffer goes: this is the one buffer of a decomposed run
    // that is re-allocated while a run is live.  (The crash records of the in-process transport are not this: comm.cpp.)
    if (d.gather_send) { HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(d.gather_send); }
    HIP_CHECK(hipMalloc(&d.gather_send, need * sizeof(double)));
    d.gather_cap = need;
  }
  d.gather_recv = d.gather_send + doubles_per_rank;      // (per call: the two gathers differ in their row width)
}

// ------------------------------------------------------------------------------------------------------------------
// Fast halo: the per-step forward communication (CommBrick::forward_comm, src/comm_brick.cpp:452-512) without a transport
// call.  A grouped RCCL exchange costs ~25 us per step against a ~12 us step kernel at 125k beads per GPU; here the step
// kernel of a rank stores the new positions of its border beads straight into the neighbours' windows (peer memory over
// xGMI, mapped once per allocation through hipIpcOpenMemHandle), and what is left per step is a single-wavefront kernel
// that publishes "my stores are complete" to both neighbours and waits for theirs, plus the copy of the window into the
// ghost slots - one launch for both when every neighbour is another GPU (k_halo_exchange_win).
// Protocol, per rank and exchange number s (parity s & 1; all ranks count the same exchanges):
//   step kernel:  stores into the neighbours' window[s & 1]                     (it follows this rank's unpack of s - 1)
//   k_halo_sync:  release-store s into my counter at each neighbour, then spin until both of my counters reach s
//   unpack:       window[s & 1] -> ghost slots
// A neighbour writes my window[s & 1] again only in exchange s + 2, after it has seen my counter s + 1, which I publish
// after my unpack of s: one counter per direction orders both the data (read after write) and the reuse of the buffer.
// The rebuild keeps its variable-size exchanges on the transport, which also fence the windows across a rebuild.
// The spin is bounded (halo_timeout_s, the transports' own limit): every wave reaches its exit, a dead neighbour ends
// in ERR_HALO_TIMEOUT -> LammpsError -> Comm::abort on this rank.
// the wait of one lane for both arrival counters.  `vstate` (verify mode only: [0] mismatch count, [1] "windows broken"): a
// counter that does not arrive in time is then not an error - the transport's copy of this halo follows and wins - but one
// more mismatch, and later exchanges do not wait again (a first multi-GPU run whose windows do not work loses a few seconds
// of its pre-roll and carries on over the transport, instead of ending in ERR_HALO_TIMEOUT).
__device__ __forceinline__ bool halo_wait_counters(const unsigned *__restrict__ mine, unsigned seq, long long timeout_ticks,
                                                   int *__restrict__ flags, unsigned *__restrict__ vstate, bool count_it) {
  if (vstate && __hip_atomic_load(vstate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
  const long long t0 = wall_clock64();
  for (int side = 0; side < 2; side++)
    while ((int)(__hip_atomic_load(mine + side, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > timeout_ticks) {
        if (vstate) {
          __hip_atomic_store(vstate + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (count_it) atomicAdd(vstate, 1u);
        } else flags[FLAG_ERROR] = ERR_HALO_TIMEOUT;
        return false;
      }
    }
  return true;
}
__global__ __launch_bounds__(64) void k_halo_sync(unsigned *__restrict__ to_dn, unsigned *__restrict__ to_up,
                                                  const unsigned *__restrict__ mine, unsigned seq, long long timeout_ticks,
                                                  int *__restrict__ flags, unsigned *__restrict__ vstate) {
  if (threadIdx.x != 0) return;
  __threadfence_system();
  __hip_atomic_store(to_dn, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(to_up, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  (void)halo_wait_counters(mine, seq, timeout_ticks, flags, vstate, true);
}
// window -> ghost slots [from below | from above]; the window is read with system-scope loads (never from a stale line)
__global__ __launch_bounds__(BLOCK) void k_halo_unpack_win(int n0, int n1, const double4 *__restrict__ from_below,
                                                           const double4 *__restrict__ from_above,
                                                           double4 *__restrict__ ghost, int corrupt) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n0 + n1) return;
  const unsigned long long *src = (const unsigned long long *)(i < n0 ? from_below + i : from_above + (i - n0));
  unsigned long long w[4];
#pragma unroll
  for (int k = 0; k < 4; k++) w[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  double4 r;
  r.x = __longlong_as_double((long long)w[0]); r.y = __longlong_as_double((long long)w[1]);
  r.z = __longlong_as_double((long long)w[2]); r.w = __longlong_as_double((long long)w[3]);
  if (corrupt && i == 0) r.x += 0.25;      // test hook (LAMMPS_LE_TEST_HALO_CORRUPT): the verify mode must notice and repair
  ghost[i] = r;
}
// The same in ONE launch (ranks on different GPUs): block 0 publishes, every block waits for the two counters itself (polls of
// two uncached words) and then copies its part of the window.  Not used when a neighbour shares this GPU (one-GPU rehearsals):
// there a grid of waiting workgroups holds up the very kernels it waits for (measured, 4 ranks on one GPU: 55 us per
// exchange against 13 + 8 us for the two launches above).
__global__ __launch_bounds__(BLOCK) void k_halo_exchange_win(unsigned *__restrict__ to_dn, unsigned *__restrict__ to_up,
                                                             const unsigned *__restrict__ mine, unsigned seq,
                                                             long long timeout_ticks, int *__restrict__ flags,
                                                             unsigned *__restrict__ vstate, int n0, int n1,
                                                             const double4 *__restrict__ from_below,
                                                             const double4 *__restrict__ from_above,
                                                             double4 *__restrict__ ghost, int corrupt) {
  __shared__ int s_ok;
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) {
      __threadfence_system();
      __hip_atomic_store(to_dn, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(to_up, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    s_ok = halo_wait_counters(mine, seq, timeout_ticks, flags, vstate, blockIdx.x == 0) ? 1 : 0;
  }
  __syncthreads();
  if (!s_ok) return;
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n0 + n1) return;
  const unsigned long long *src = (const unsigned long long *)(i < n0 ? from_below + i : from_above + (i - n0));
  unsigned long long w[4];
#pragma unroll
  for (int k = 0; k < 4; k++) w[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  double4 r;
  r.x = __longlong_as_double((long long)w[0]); r.y = __longlong_as_double((long long)w[1]);
  r.z = __longlong_as_double((long long)w[2]); r.w = __longlong_as_double((long long)w[3]);
  if (corrupt && i == 0) r.x += 0.25;      // test hook (LAMMPS_LE_TEST_HALO_CORRUPT): the verify mode must notice and repair
  ghost[i] = r;
}

// LAMMPS_LE_FAST_HALO_VERIFY=1: the same halo once more through the transport; differences are counted and the transport's
// copy wins (a first multi-GPU run checks the windows against RCCL for as long as it likes before it relies on them)
__global__ __launch_bounds__(BLOCK) void k_halo_verify(int m, double4 *__restrict__ ghost, const double4 *__restrict__ ref,
                                                       unsigned *__restrict__ mismatches) {
  int i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= m) return;
  const double4 a = ghost[i], b = ref[i];
  const bool same = __double_as_longlong(a.x) == __double_as_longlong(b.x) && __double_as_longlong(a.y) == __double_as_longlong(b.y) &&
                    __double_as_longlong(a.z) == __double_as_longlong(b.z) && __double_as_longlong(a.w) == __double_as_longlong(b.w);
  if (!same) { atomicAdd(mismatches, 1u); ghost[i] = b; }
}

struct PeerInfo {
  long long pid;
  unsigned long long raw;        // the window's address in its owner's address space (ranks that are threads of one process)
  int exported;
  int pad;
  char busid[32];                // PCI bus id of the rank's GPU: a neighbour on MY GPU means a one-GPU rehearsal
  hipIpcMemHandle_t handle;
};
static size_t halo_flag_offset(size_t cap) { return 4 * cap * sizeof(double4); }
constexpr int HALO_MISMATCH_SLOT = 16;    // (behind the two arrival counters, local use only)
unsigned dd_halo_mismatches(DeviceState &d) {
  unsigned v = 0;
  if (d.halo_flag) HIP_CHECK(hipMemcpy(&v, d.halo_flag + HALO_MISMATCH_SLOT, sizeof v, hipMemcpyDeviceToHost));
  return v;
}

// per run: LAMMPS_LE_FAST_HALO=0 keeps the mapped windows idle, LAMMPS_LE_FAST_HALO_VERIFY=1 checks every window halo against
// the transport (same values on every rank: the exchanges are counted in step)
void dd_fast_halo_switch(DeviceState &d) {
  const char *env = getenv("LAMMPS_LE_FAST_HALO"), *ver = getenv("LAMMPS_LE_FAST_HALO_VERIFY");
  d.fast_halo = d.halo_mapped && !(env && atoi(env) == 0);
  d.halo_verify = ver && atoi(ver) != 0;
  d.packed_peer = 0;
}

void dd_fast_halo_free(DeviceState &d) {
  for (int k = 0; k < 2; k++) {
    if (d.peer_base[k] && !(k == 1 && d.peer_base[1] == d.peer_base[0])) (void)hipIpcCloseMemHandle(d.peer_base[k]);
    d.peer_base[k] = nullptr; d.peer_win[k] = nullptr; d.peer_flag[k] = nullptr;
  }
  if (d.halo_win) (void)hipFree(d.halo_win);
  d.halo_win = nullptr; d.halo_flag = nullptr; d.halo_cap = 0;
  d.fast_halo = d.halo_mapped = false; d.packed_peer = 0; d.halo_seq = 0;
}

// collective: allocate this rank's window, exchange its address / IPC handle, map the two neighbours' windows.
// LAMMPS_LE_FAST_HALO=0 switches it off; with RCCL it is opt-in (=1) until it has run on a multi-GPU node.
void dd_fast_halo_setup(DeviceState &d, Comm &comm) {
  comm.barrier();                // nobody stores into a window that is about to go
  dd_fast_halo_free(d);
  const char *env = getenv("LAMMPS_LE_FAST_HALO");
  // (not with the in-process transport: its ranks are threads whose streams share the few hardware queues of ONE process, and
  //  a counter spin queued in front of the very kernel it waits for would never end)
  const bool want = comm.backend == Comm::LOCAL ? false : env ? atoi(env) != 0 : comm.backend == Comm::SHM;
  if (!want || comm.world < 2) return;
  const int P = comm.world, me = comm.rank, nbr[2] = {(me + P - 1) % P, (me + 1) % P};
  d.halo_cap = (size_t)d.npad;
  const size_t bytes = halo_flag_offset(d.halo_cap) + 256;
  long bad = 0;
  PeerInfo mine{};
  // Uncached (else fine-grained) device memory, as RCCL allocates the buffers its peers write: a neighbour GPU stores into
  // this window while kernels of this GPU are running, which ordinary (coarse-grained) device memory is only coherent
  // for at kernel boundaries of ONE device - this GPU's L2 could keep serving a line of the previous exchange.
  if (hipExtMallocWithFlags((void **)&d.halo_win, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    d.halo_win = nullptr;
    if (hipExtMallocWithFlags((void **)&d.halo_win, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      d.halo_win = nullptr;
      bad = 1;
    }
  }
  if (!bad) {
    HIP_CHECK(hipMemset(d.halo_win, 0, bytes));
    HIP_CHECK(hipStreamSynchronize(nullptr));
    d.halo_flag = (unsigned *)((char *)d.halo_win + halo_flag_offset(d.halo_cap));
    {
      int dev = 0;
      HIP_CHECK(hipGetDevice(&dev));
      if (hipDeviceGetPCIBusId(mine.busid, (int)sizeof mine.busid, dev) != hipSuccess) { (void)hipGetLastError(); mine.busid[0] = 0; }
    }
    mine.pid = (long long)getpid();
    mine.raw = (unsigned long long)(uintptr_t)d.halo_win;
    mine.exported = hipIpcGetMemHandle(&mine.handle, d.halo_win) == hipSuccess ? 1 : 0;
    if (!mine.exported) (void)hipGetLastError();
  }
  std::vector<PeerInfo> all((size_t)P);
  comm.allgather_host(&mine, all.data(), sizeof(PeerInfo));
  for (int k = 0; k < 2 && !bad; k++) {
    const PeerInfo &pi = all[(size_t)nbr[k]];
    char *base = nullptr;
    if (pi.raw == 0) bad = 1;
    else if (pi.pid == mine.pid) base = (char *)(uintptr_t)pi.raw;
    else if (k == 1 && nbr[1] == nbr[0]) base = (char *)d.peer_base[0];          // two ranks: one neighbour, mapped once
    else if (pi.exported) {
      void *ptr = nullptr;
      if (hipIpcOpenMemHandle(&ptr, pi.handle, hipIpcMemLazyEnablePeerAccess) == hipSuccess) base = (char *)ptr;
      else (void)hipGetLastError();
      d.peer_base[k] = ptr;
    }
    if (k == 1 && nbr[1] == nbr[0]) d.peer_base[1] = d.peer_base[0];
    if (!base) { bad = 1; break; }
    d.peer_win[k] = (double4 *)base;
    // my counter at the rank below is its "from above" one, at the rank above its "from below" one
    d.peer_flag[k] = (unsigned *)(base + halo_flag_offset(d.halo_cap)) + (k == 0 ? 1 : 0);
  }
  bad = comm.allreduce_host_max(bad);           // all ranks or none
  if (bad) { dd_fast_halo_free(d); return; }
  // one launch per exchange when no neighbour shares this GPU (LAMMPS_LE_HALO_FUSED=0 / 1 overrides: tests run the fused
  // kernel between processes on one GPU)
  long shared = 0;
  for (int k = 0; k < 2; k++) {
    const PeerInfo &pi = all[(size_t)nbr[k]];
    if (!mine.busid[0] || !pi.busid[0] || strncmp(mine.busid, pi.busid, sizeof mine.busid) == 0) shared = 1;
  }
  shared = comm.allreduce_host_max(shared);
  const char *fz = getenv("LAMMPS_LE_HALO_FUSED");
  d.halo_fused = fz ? atoi(fz) != 0 : !shared;
  d.halo_timeout_s = comm.timeout_s;
  d.halo_mapped = true;
  dd_fast_halo_switch(d);
  comm.barrier();
}

void dd_alloc(DeviceState &d, int world) {
  size_t np = d.npad;
  auto al = [](auto *&p, size_t bytes) {
    if (p) (void)hipFree(p);
    HIP_CHECK(hipMalloc((void **)&p, bytes));
    HIP_CHECK(hipMemset(p, 0, bytes));
    HIP_CHECK(hipStreamSynchronize(nullptr));   // see dalloc (device.cpp)
  };
  al(d.gcell_start, ((size_t)d.ncells + 2) * sizeof(int));
  al(d.gcell_count, ((size_t)d.ncells + 2) * sizeof(int));
  for (int k = 0; k < 2; k++) { al(d.sendlist[k], np * sizeof(int)); al(d.sendlist_alt[k], np * sizeof(int)); al(d.migbuf[k], np * MIG_W * sizeof(double) / 4 + 1024); }
  d.map_stale = true;
  al(d.migin, np * MIG_W * sizeof(double) / 2 + 1024);
  al(d.sendbuf, np * sizeof(double4));
  al(d.recvbuf, np * sizeof(double4));
  al(d.gdest, np * sizeof(int));
  al(d.gone, np * sizeof(int));
  al(d.phase, np);
  al(d.sendslot, np * sizeof(int));
  if (!d.comm_stream) {
    HIP_CHECK(hipStreamCreateWithFlags(&d.comm_stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&d.ev_phase1, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&d.ev_halo, hipEventDisableTiming));
  }
  al(d.gtag_in, np * sizeof(int));
  (void)world;
}

// Rebuild ownership and ghosts on this rank, then sort and build lists.  Collective over all ranks.
void dd_reneighbor(DeviceState &d, Comm &comm, double cutneighsq, const double sl[4], bool has_pair, bool build_lists) {
  hipStream_t st = d.stream;
  dd_halo_wait(d);
  d.halo_ahead = false;
  d.packed_ahead = false;
  d.packed_peer = 0;       // (a halo the step kernel pushed into the neighbours' windows before a rebuild is never consumed)
  const int P = comm.world, me = comm.rank, dn_rank = (me + P - 1) % P, up_rank = (me + 1) % P;
  const double width = d.box.prd[2] / P;   // the SAME expression on every rank and in Engine::upload (owner of a bead)
  int n = d.n, nb = std::max(1, (n + BLOCK - 1) / BLOCK);
  const int migcap = (int)(((size_t)d.npad * MIG_W / 4) / MIG_W);
  // counts travel rank-to-rank on the device and reach the host together with this rank's own counters: one
  // host synchronisation per phase (migration, borders) instead of three
  const unsigned counters = (1u << FLAG_COUNT_A) | (1u << FLAG_COUNT_B) | (1u << FLAG_NDRAW) | (1u << FLAG_SEND_BOTH);
  auto swap_counts = [&](int slot_dn, int slot_up) {
    comm.exchange(st, {{d.flags + slot_dn, sizeof(int), dn_rank}, {d.flags + slot_up, sizeof(int), up_rank}},
                  {{d.flags + FLAG_RECV_UP, sizeof(int), up_rank}, {d.flags + FLAG_RECV_DN, sizeof(int), dn_rank}});
    sync_flags(d, counters);     // published, then zeroed for the next phase
  };
  // ---- 1. migration (+ binning of what stays, + the old slots' tags out of map[]) ----
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_COUNT_A, 0, 4 * sizeof(int), st));   // COUNT_A, COUNT_B, NDRAW, NLIST
  if (d.map_stale) {      // first rebuild on freshly uploaded arrays: map[] may hold anything
    hipLaunchKernelGGL(k_fill_int, dim3((d.maxtag + 2 + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.maxtag + 2, d.map, -1);
    d.map_stale = false;
  }
  if (d.cell_count_dirty) { HIP_CHECK(hipMemsetAsync(d.cell_count, 0, ((size_t)d.ncells + 2) * sizeof(int), st)); d.cell_count_dirty = false; }
  const BinArgs BA{d.ncell[0], d.ncell[1], d.ncell[2], d.cellinv[0], d.cellinv[1], d.cellinv[2], d.zlo_ext, d.cell_of, d.cell_count,
                   d.tag_tmp, d.ncells};
  {
    const int nslots = n + d.nghost, gl = std::max(1, (nslots + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_dd_leave, dim3(gl), dim3(BLOCK), 0, st, n, nslots, d.npad, migcap, d.box, d.slab_lo, width, me, P, d.pos,
                       d.v[0], d.v[1], d.v[2], d.tag, d.img, d.migbuf[0], d.migbuf[1], d.gone, d.map, BA, d.flags);
  }
  swap_counts(FLAG_COUNT_B, FLAG_NDRAW);      // what I send down arrives as the lower rank's "from above"
  int ndn = d.flags_h[FLAG_COUNT_B], nup = d.flags_h[FLAG_NDRAW];
  int recvc[2] = {d.flags_h[FLAG_RECV_DN], d.flags_h[FLAG_RECV_UP]};   // [0] from below (their up), [1] from above
  if (ndn > migcap || nup > migcap || recvc[0] + recvc[1] > 2 * migcap)
    throw LammpsError("too many beads migrate between slabs in one reneighbor (down " + std::to_string(ndn) + ", up " +
                      std::to_string(nup) + ", arriving " + std::to_string(recvc[0] + recvc[1]) + ", capacity " +
                      std::to_string(migcap) + ")");
  comm.exchange(st, {{d.migbuf[0], (size_t)ndn * MIG_W * sizeof(double), dn_rank},
                     {d.migbuf[1], (size_t)nup * MIG_W * sizeof(double), up_rank}},
                {{d.migin, (size_t)recvc[1] * MIG_W * sizeof(double), up_rank},
                 {d.migin + (size_t)recvc[1] * MIG_W, (size_t)recvc[0] * MIG_W * sizeof(double), dn_rank}});
  int narr = recvc[0] + recvc[1];
  if (n + narr > d.npad - 64) throw LammpsError("slab overflow: more beads than the allocation of this rank");
  if (narr)
    hipLaunchKernelGGL(k_dd_arrive, dim3((narr + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, narr, n, d.npad, d.migin,
                       d.pos, d.v[0], d.v[1], d.v[2], d.tag, d.img, d.gone, d.box, BA);
  // ---- 2. cell sort of kept + arrived beads (sets map for them; the gone ones drop off the end) ----
  launch_sort_owned(d, n + narr, n + narr - ndn - nup, d.gone, true);
  d.n = n = n + narr - ndn - nup;
  nb = std::max(1, (n + BLOCK - 1) / BLOCK);
  // ---- 3. borders ----
  hipLaunchKernelGGL(k_dd_borders, dim3(nb), dim3(BLOCK), 0, st, n, d.pos, d.box, d.slab_lo, width,
                     std::min(sqrt(cutneighsq), d.cutghost), d.cutghost, d.bpa, d.tag, d.map, d.num_bond, d.bond_atom,
                     d.sendlist[0], d.sendlist[1], d.flags, d.phase, d.sendslot, rng_validate_args(d), d.ghost_whole_shell ? 1 : 0);
  swap_counts(FLAG_COUNT_A, FLAG_COUNT_B);
  d.nsend[0] = d.flags_h[FLAG_COUNT_A];
  d.nsend[1] = d.flags_h[FLAG_COUNT_B];
  d.sendslot_fallback = false;
  d.nrecv[0] = d.flags_h[FLAG_RECV_DN];
  d.nrecv[1] = d.flags_h[FLAG_RECV_UP];
  d.nghost = d.nrecv[0] + d.nrecv[1];
  if (n + d.nghost > d.npad - 64) throw LammpsError("ghost overflow on this rank");
  // first exchange: positions and tags (arrival order: from above first, then from below — same as every step)
  int *tagsend = d.le_i[6], *tagrecv = d.gtag_in;
  int nsall = d.nsend[0] + d.nsend[1];
  if (nsall)
    hipLaunchKernelGGL(k_dd_pack_xt, dim3((nsall + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.nsend[0], d.nsend[1], d.sendlist[0],
                       d.sendlist[1], d.pos, d.tag, d.sendbuf, tagsend);
  comm.exchange(st, {{d.sendbuf, (size_t)d.nsend[0] * sizeof(double4), dn_rank},
                     {d.sendbuf + d.nsend[0], (size_t)d.nsend[1] * sizeof(double4), up_rank},
                     {tagsend, (size_t)d.nsend[0] * sizeof(int), dn_rank},
                     {tagsend + d.nsend[0], (size_t)d.nsend[1] * sizeof(int), up_rank}},
                {{d.recvbuf, (size_t)d.nrecv[1] * sizeof(double4), up_rank},
                 {d.recvbuf + d.nrecv[1], (size_t)d.nrecv[0] * sizeof(double4), dn_rank},
                 {tagrecv, (size_t)d.nrecv[1] * sizeof(int), up_rank},
                 {tagrecv + d.nrecv[1], (size_t)d.nrecv[0] * sizeof(int), dn_rank}});
  // ---- 4. ghosts into cell order behind the owned beads (the scan leaves the counts at zero for the next rebuild) ----
  int m = d.nghost, gb = std::max(1, (m + BLOCK - 1) / BLOCK);
  int *gcell_of = d.le_i[7], *grank = d.le_i[8], *gperm = d.le_i[9];
  static const bool no_direct = getenv("LAMMPS_LE_NO_DIRECT_RECV") != nullptr;
  int *rel_out = d.le_i[10], *rel_in = d.le_i[11];
  if (m)
    hipLaunchKernelGGL(k_dd_ghost_bin, dim3(gb), dim3(BLOCK), 0, st, m, d.recvbuf, d.box, d.ncell[0], d.ncell[1],
                       d.ncell[2], d.cellinv[0], d.cellinv[1], d.cellinv[2], d.zlo_ext, gcell_of, d.gcell_count, grank);
  scan_cells(d, d.gcell_count, d.gcell_start, d.ncells, m);
  if (m) {
    hipLaunchKernelGGL(k_dd_ghost_slot, dim3(gb), dim3(BLOCK), 0, st, m, gcell_of, d.gcell_start, grank, gperm);
    hipLaunchKernelGGL(k_dd_ghost_sort, dim3((d.ncells + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.ncells, d.gcell_start,
                       gperm, tagrecv);
    hipLaunchKernelGGL(k_dd_ghost_place, dim3(gb), dim3(BLOCK), 0, st, m, n, d.nrecv[1], d.nrecv[0], gperm, d.recvbuf, tagrecv,
                       d.pos, d.tag, d.gdest, d.map, d.posf, no_direct ? (int *)nullptr : rel_out, d.flags);
  }
  // ---- 4b. tell the senders the sorted order of what they sent; they reorder their lists (direct receive from now on)
  d.direct_recv = false;
  if (!no_direct) {
    // my "from above" block came from up_rank's lower list, my "from below" block from dn_rank's upper list
    comm.exchange(st, {{rel_out, (size_t)d.nrecv[1] * sizeof(int), up_rank},
                       {rel_out + d.nrecv[1], (size_t)d.nrecv[0] * sizeof(int), dn_rank}},
                  {{rel_in, (size_t)d.nsend[0] * sizeof(int), dn_rank},
                   {rel_in + d.nsend[0], (size_t)d.nsend[1] * sizeof(int), up_rank}});
    // (slabs are at least two ghost shells thick, so the two blocks cannot interleave and no bead is in both send
    // lists; FLAG_GHOST_MIXED / FLAG_SEND_BOTH would be an internal error and are reported with the build's flags)
    if (nsall) {
      hipLaunchKernelGGL(k_dd_reorder_sends, dim3((nsall + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.nsend[0], d.nsend[1],
                         rel_in, d.sendlist[0], d.sendlist[1], d.sendlist_alt[0], d.sendlist_alt[1], d.sendslot);
      std::swap(d.sendlist[0], d.sendlist_alt[0]);
      std::swap(d.sendlist[1], d.sendlist_alt[1]);
    }
    d.direct_recv = true;
  }
  // ---- 5. lists (the engine builds them itself when an Atom::sort emulation has to come in between) ----
  if (build_lists) launch_lists(d, cutneighsq, sl, has_pair);
}

// per-step forward communication of ghost positions (CommBrick::forward_comm, src/comm_brick.cpp:452-512): owned
// border beads of `src` are packed, exchanged and scattered into the ghost slots of `dst`, all on stream `st`.
void dd_halo(DeviceState &d, Comm &comm, hipStream_t st, const double4 *src, double4 *dst) {
  const int P = comm.world, me = comm.rank, dn_rank = (me + P - 1) % P, up_rank = (me + 1) % P;
  int nsall = d.nsend[0] + d.nsend[1];
  if (d.packed_peer) {     // the step kernel stored this halo into the neighbours' windows: counters, then window -> ghost slots
    const int parity = d.packed_peer - 1;
    const unsigned seq = d.halo_seq + 1u;
    static int clock_khz = 0;
    if (!clock_khz) {
      int dev = 0;
      HIP_CHECK(hipGetDevice(&dev));
      if (hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || clock_khz <= 0) clock_khz = 100000;
    }
    // (verify mode: a window that does not deliver is rejected after 5 s, not waited for as long as the transports wait)
    const long long ticks = (long long)((d.halo_verify ? std::min(d.halo_timeout_s, 5.0) : d.halo_timeout_s) * 1e3 * (double)clock_khz);
    unsigned *vstate = d.halo_verify ? d.halo_flag + HALO_MISMATCH_SLOT : (unsigned *)nullptr;
    // test hook (LAMMPS_LE_TEST_HALO_MUTE): my "stores complete" never reaches the neighbours - windows that do not deliver
    const bool mute = getenv("LAMMPS_LE_TEST_HALO_MUTE") != nullptr;
    unsigned *to_dn = mute ? d.halo_flag + HALO_MISMATCH_SLOT + 4 : d.peer_flag[0], *to_up = mute ? d.halo_flag + HALO_MISMATCH_SLOT + 5 : d.peer_flag[1];
    static const int corrupt = getenv("LAMMPS_LE_TEST_HALO_CORRUPT") ? 1 : 0;
    if (d.halo_fused)
      hipLaunchKernelGGL(k_halo_exchange_win, dim3(std::max(1, (d.nghost + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, to_dn,
                         to_up, d.halo_flag, seq, ticks, d.flags, vstate, d.nrecv[0], d.nrecv[1],
                         d.halo_win + (size_t)(parity * 2 + 0) * d.halo_cap, d.halo_win + (size_t)(parity * 2 + 1) * d.halo_cap,
                         dst + d.n, corrupt);
    else {
      hipLaunchKernelGGL(k_halo_sync, dim3(1), dim3(64), 0, st, to_dn, to_up, d.halo_flag, seq, ticks, d.flags, vstate);
      if (d.nghost)
        hipLaunchKernelGGL(k_halo_unpack_win, dim3((d.nghost + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.nrecv[0], d.nrecv[1],
                           d.halo_win + (size_t)(parity * 2 + 0) * d.halo_cap, d.halo_win + (size_t)(parity * 2 + 1) * d.halo_cap,
                           dst + d.n, corrupt);
    }
    d.halo_seq = seq;
    d.packed_peer = 0;
    if (d.halo_verify && nsall + d.nghost > 0) {
      hipLaunchKernelGGL(k_dd_pack, dim3((std::max(nsall, 1) + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.nsend[0], d.nsend[1],
                         d.sendlist[0], d.sendlist[1], src, d.sendbuf);
      comm.exchange(st, {{d.sendbuf, (size_t)d.nsend[0] * sizeof(double4), dn_rank},
                         {d.sendbuf + d.nsend[0], (size_t)d.nsend[1] * sizeof(double4), up_rank}},
                    {{d.recvbuf + d.nrecv[0], (size_t)d.nrecv[1] * sizeof(double4), up_rank},
                     {d.recvbuf, (size_t)d.nrecv[0] * sizeof(double4), dn_rank}});
      if (d.nghost)
        hipLaunchKernelGGL(k_halo_verify, dim3((d.nghost + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.nghost, dst + d.n,
                           d.recvbuf, d.halo_flag + HALO_MISMATCH_SLOT);
    }
    return;
  }
  if (nsall && !d.packed_ahead)
    hipLaunchKernelGGL(k_dd_pack, dim3((nsall + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.nsend[0], d.nsend[1],
                       d.sendlist[0], d.sendlist[1], src, d.sendbuf);
  d.packed_ahead = false;
  if (d.direct_recv) {   // the senders pack in my sorted ghost order [from below | from above]: receive in place
    comm.exchange(st, {{d.sendbuf, (size_t)d.nsend[0] * sizeof(double4), dn_rank},
                       {d.sendbuf + d.nsend[0], (size_t)d.nsend[1] * sizeof(double4), up_rank}},
                  {{dst + d.n + d.nrecv[0], (size_t)d.nrecv[1] * sizeof(double4), up_rank},
                   {dst + d.n, (size_t)d.nrecv[0] * sizeof(double4), dn_rank}});
    return;
  }
  comm.exchange(st, {{d.sendbuf, (size_t)d.nsend[0] * sizeof(double4), dn_rank},
                     {d.sendbuf + d.nsend[0], (size_t)d.nsend[1] * sizeof(double4), up_rank}},
                {{d.recvbuf, (size_t)d.nrecv[1] * sizeof(double4), up_rank},
                 {d.recvbuf + d.nrecv[1], (size_t)d.nrecv[0] * sizeof(double4), dn_rank}});
  if (d.nghost)
    hipLaunchKernelGGL(k_dd_unpack, dim3((d.nghost + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, d.nghost, d.n, d.gdest,
                       d.recvbuf, dst);
}
void dd_halo(DeviceState &d, Comm &comm) { dd_halo(d, comm, d.stream, d.pos, d.pos); }
// the main stream may not touch ghost slots, send/receive buffers or the communicator while comm_stream works
void dd_halo_wait(DeviceState &d) {
  if (!d.halo_inflight) return;
  HIP_CHECK(hipStreamWaitEvent(d.stream, d.ev_halo, 0));
  d.halo_inflight = false;
}

// every rank receives (tag, x, xhold) of all beads: xt / xht by tag for the replicated LE kernels
void dd_gather_positions(DeviceState &d, Comm &comm) {
  dd_halo_wait(d);
  long maxn = comm.allreduce_host_max(d.n);
  int stride = (int)maxn;
  ensure_gather(d, (size_t)stride * GATH_LE_W, comm.world);
  hipLaunchKernelGGL(k_dd_gather_pack, dim3((stride + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, d.stream, d.n, d.npad, stride,
                     d.pos, d.xhold, d.v[0], d.v[1], d.v[2], d.f[0], d.f[1], d.f[2], d.tag, d.img, 0, d.gather_send);
  comm.allgather(d.stream, d.gather_send, d.gather_recv, (size_t)stride * GATH_LE_W * sizeof(double));
  long total = (long)stride * comm.world;
  hipLaunchKernelGGL(k_dd_scatter_xt, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, d.stream, total,
                     d.gather_recv, d.xt, d.xht);
}

// ---- run_style respa in a decomposed run: the per-level force tables (kept by tag, engine.cpp respa_*) follow their beads ----
// A bead that migrates at a rebuild needs its rows on the new owner.  The slow path's answer: before the rebuild every rank
// packs (tag, f[3]) of the beads it owns, the lists are all-gathered (padded to the longest, tag 0 = padding) and every rank
// writes all rows into its table - the table is then complete on every rank, whoever owns a bead next.
__global__ __launch_bounds__(BLOCK) void k_dd_rows_pack(int n, int stride, const int *__restrict__ tag,
                                                        const double *__restrict__ table, double *__restrict__ out) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= stride) return;
  double *b = out + (size_t)p * 4;
  if (p < n) {
    const int t = tag[p];
    const double *row = table + 3 * (size_t)t;
    b[0] = (double)t; b[1] = row[0]; b[2] = row[1]; b[3] = row[2];
  } else b[0] = 0.0;
}
__global__ __launch_bounds__(BLOCK) void k_dd_rows_scatter(long total, const double *__restrict__ in, double *__restrict__ table) {
  long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= total) return;
  const double *b = in + (size_t)i * 4;
  const int t = (int)b[0];
  if (t <= 0) return;
  double *row = table + 3 * (size_t)t;
  row[0] = b[1]; row[1] = b[2]; row[2] = b[3];
}
void dd_gather_rows3(DeviceState &d, Comm &comm, double *table_by_tag) {
  dd_halo_wait(d);
  const int stride = (int)comm.allreduce_host_max(d.n);
  ensure_gather(d, (size_t)stride * 4, comm.world);
  hipLaunchKernelGGL(k_dd_rows_pack, dim3((stride + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, d.stream, d.n, stride, d.tag, table_by_tag,
                     d.gather_send);
  comm.allgather(d.stream, d.gather_send, d.gather_recv, (size_t)stride * 4 * sizeof(double));
  const long total = (long)stride * comm.world;
  hipLaunchKernelGGL(k_dd_rows_scatter, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, d.stream, total, d.gather_recv,
                     table_by_tag);
}

// ---- what a firing of fix extrusion / ex_unload needs instead of every bead's position (canonical visit order) ----
// The replicated kernels of these two fixes read stored coordinates only at the ends of the extruder bonds of the last
// bond list (table 0) and, for extrusion, at their chain neighbours t - 1 / t + 1 (the beads an end can step to,
// fix_extrusion.cpp:406-515).  The bond tables are replicated, so every rank knows which tags those are; each packs
// (tag, x, type, xhold) of the ones it OWNS, the variable-length lists are all-gathered (padded to the longest; a padding
// row has tag 0) and scattered into xt / xht by tag: O(extruders) rows of 64 bytes instead of N.  The shape of the
// reference's own pack / unpack of touched atoms, src/USER-LE/fix_extrusion.cpp:1147-1419.
__global__ __launch_bounds__(BLOCK) void k_dd_need_pack(int n, int T, int bpa, int btype, int with_nbrs, int cap,
                                                        const int *__restrict__ tag, const int *__restrict__ num_bond0,
                                                        const int *__restrict__ bond_type0, const double4 *__restrict__ pos,
                                                        const double4 *__restrict__ xhold, double *__restrict__ out,
                                                        int *__restrict__ counter) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  bool need = false;
  int t = 0;
  if (p < n) {
    t = tag[p];
    auto has = [&](int u) {          // (tags 0 and T + 1 are sentinels with no bonds)
      if (u < 1 || u > T) return false;
      const int nb = num_bond0[u];
      for (int m = 0; m < nb; m++) if (bond_type0[(size_t)u * bpa + m] == btype) return true;
      return false;
    };
    need = has(t) || (with_nbrs && (has(t - 1) || has(t + 1)));
  }
  const int slot = wave_append(need, counter);
  if (need && slot < cap) {
    double *b = out + (size_t)slot * GATH_LE_W;
    const double4 r = pos[p], h = xhold[p];
    b[0] = (double)t; b[1] = r.x; b[2] = r.y; b[3] = r.z; b[4] = r.w; b[5] = h.x; b[6] = h.y; b[7] = h.z;
  }
}
__global__ __launch_bounds__(BLOCK) void k_dd_need_pad(int from, int to, double *__restrict__ out) {
  int i = from + blockIdx.x * BLOCK + threadIdx.x;
  if (i < to) out[(size_t)i * GATH_LE_W] = 0.0;
}
void dd_gather_needed(DeviceState &d, Comm &comm, int btype, bool with_nbrs) {
  dd_halo_wait(d);
  hipStream_t st = d.stream;
  // room for every owned bead: allocated once per system size (no regrow under a copy another rank's stream may still run)
  ensure_gather(d, (size_t)d.npad * GATH_LE_W / std::max(1, comm.world) + (size_t)4096 * GATH_LE_W, comm.world);
  const int cap = (int)((d.gather_recv - d.gather_send) / GATH_LE_W);
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_AUX, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_dd_need_pack, dim3(std::max(1, (d.n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, d.n, d.maxtag, d.bpa, btype,
                     with_nbrs ? 1 : 0, cap, d.tag, d.num_bond0, d.bond_type0, d.pos, d.xhold, d.gather_send, d.flags + FLAG_AUX);
  sync_flags(d);
  const long mine = d.flags_h[FLAG_AUX];
  const long most = comm.allreduce_host_max(mine);
  if (most > cap) throw LammpsError("more extruder beads on one rank than the gather buffer holds (" + std::to_string(most) + ")");
  if (most == 0) return;
  if (mine < most)
    hipLaunchKernelGGL(k_dd_need_pad, dim3((unsigned)((most - mine + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, (int)mine, (int)most,
                       d.gather_send);
  comm.allgather(st, d.gather_send, d.gather_recv, (size_t)most * GATH_LE_W * sizeof(double));
  const long total = most * comm.world;
  hipLaunchKernelGGL(k_dd_scatter_xt, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, total, d.gather_recv, d.xt,
                     d.xht);
}

// host download of the whole system: rows of GATH_W doubles for every bead of every rank
void dd_gather_all(DeviceState &d, Comm &comm, std::vector<double> &rows, int &stride_out) {
  dd_halo_wait(d);
  long maxn = comm.allreduce_host_max(d.n);
  int stride = (int)maxn;
  ensure_gather(d, (size_t)stride * GATH_W, comm.world);
  hipLaunchKernelGGL(k_dd_gather_pack, dim3((stride + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, d.stream, d.n, d.npad, stride,
                     d.pos, d.xhold, d.v[0], d.v[1], d.v[2], d.f[0], d.f[1], d.f[2], d.tag, d.img, 1, d.gather_send);
  comm.allgather(d.stream, d.gather_send, d.gather_recv, (size_t)stride * GATH_W * sizeof(double));
  rows.resize((size_t)stride * comm.world * GATH_W);
  HIP_CHECK(hipMemcpyAsync(rows.data(), d.gather_recv, rows.size() * sizeof(double), hipMemcpyDeviceToHost, d.stream));
  stream_sync(d);
  stride_out = stride;
}

}  // namespace lmp_le
