// device.cpp — HBM allocation for one engine instance (288 GB per MI355X: everything stays resident).
#include "comm.h"
#include "device.h"
#include "bin_inl.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>

namespace lmp_le {

template <class T>
static void dalloc(T *&p, size_t count) {
  HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
  // null-stream memset + wait: the engine's streams are non-blocking, i.e. NOT ordered behind the null stream, and
  // a memset that is still pending when the first kernel writes the buffer would wipe that kernel's output
  HIP_CHECK(hipMemset(p, 0, count * sizeof(T)));
  HIP_CHECK(hipStreamSynchronize(nullptr));
}
template <class T>
static void dfree(T *&p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

void dev_alloc_neigh(DeviceState &d, int maxneigh) {
  // (a bead's count word keeps the entries in 16 bits next to the number of bond entries: engine.h NN_BOND_SHIFT)
  if (maxneigh > NN_COUNT_MASK) throw LammpsError("Neighbor list overflow: more than 65535 neighbors per bead");
  dfree(d.neigh);
  d.maxneigh = maxneigh;
  dalloc(d.neigh, (size_t)maxneigh * d.npad);
}

void dev_alloc(DeviceState &d, int n, int maxtag, int ntypes, int bpa, int maxspecial, const Box &box,
               double cutneigh) {
  if (!d.stream) HIP_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
  d.n = n;
  d.npad = ((n + 63) / 64) * 64 + 64;
  d.maxtag = maxtag;
  d.ntypes = ntypes;
  d.bpa = bpa;
  d.maxspecial = maxspecial;
  d.box = box;
  d.ntotal = maxtag;
  if (!d.dd) d.zlo_ext = box.lo[2];
  d.row_tile = (d.dd || getenv("LAMMPS_LE_NO_ROW_TILES")) ? 0 : ROW_TILE;      // (row_id relies on it being ROW_TILE or 0)
  size_t np = d.npad, nt = (size_t)maxtag + 2;
  dalloc(d.pos, np); dalloc(d.pos_tmp, np); dalloc(d.xhold, np); dalloc(d.posf, np);
  for (int k = 0; k < 3; k++) { dalloc(d.v[k], np); dalloc(d.v_tmp[k], np); dalloc(d.f[k], np); }
  dalloc(d.tag, np); dalloc(d.tag_tmp, np);
  dalloc(d.img, 3 * np); dalloc(d.img_tmp, 3 * np);
  dalloc(d.map, nt); dalloc(d.type_t, nt); dalloc(d.crank, nt);
  dalloc(d.num_bond, nt); dalloc(d.bond_type, nt * bpa); dalloc(d.bond_atom, nt * bpa);
  dalloc(d.nspecial, nt * 3); dalloc(d.special, nt * (size_t)maxspecial);
  dalloc(d.num_bond0, nt); dalloc(d.bond_type0, nt * bpa); dalloc(d.bond_atom0, nt * bpa);
  if (d.apa > 0) {
    dalloc(d.angle_pack, (size_t)ANGLE_PACK_COLS * nt * 4);
    d.angle_pack_dirty = true;
    dalloc(d.num_angle, nt); dalloc(d.angle_type, nt * d.apa); dalloc(d.angle_a1, nt * d.apa); dalloc(d.angle_a2, nt * d.apa);
    dalloc(d.angle_a3, nt * d.apa);
    d.ecap = d.apa + 8;
    dalloc(d.eff_n, np); dalloc(d.eff_rec, np * (size_t)d.ecap * 4);       // by the bead's physical index, records column-major
  }
  if (maxtag >= (1 << BOND_TYPE_SHIFT)) throw LammpsError("MI355X engine: atom IDs must stay below 2^26");
  d.bond_pack_stride = ((1 + bpa) + 3) & ~3;
  dalloc(d.bond_pack, nt * (size_t)d.bond_pack_stride);
  for (int k = 0; k < 2; k++) dalloc(d.bond_pack_p[k], np * (size_t)d.bond_pack_stride);
  d.bond_pack_dirty = true;
  d.bond_pack_p_valid = false;
  // cells of edge >= cutneigh
  d.ncells = 1;
  for (int k = 0; k < 3; k++) {
    double extent = (k == 2 && d.dd) ? (d.slab_hi - d.slab_lo) + 2.0 * d.cutghost : box.prd[k];
    d.ncell[k] = cutneigh > 0.0 ? (int)(extent / (k == 0 ? cutneigh / CELL_XSPLIT : cutneigh)) : 1;
    if (d.ncell[k] < 1) d.ncell[k] = 1;
    // keep cells from getting needlessly tiny for bond-only runs
    d.cellinv[k] = d.ncell[k] / extent;
    d.ncells *= d.ncell[k];
  }
  dalloc(d.cell_of, np); dalloc(d.cell_count, (size_t)d.ncells + 2); dalloc(d.cell_start, (size_t)d.ncells + 2);
  dalloc(d.cell_fill, (size_t)d.ncells + 2); dalloc(d.scan_tmp, (size_t)d.ncells / 1024 + 2); dalloc(d.perm, np);
  double vol = box.prd[0] * box.prd[1] * box.prd[2];
  double expect = (double)n / vol * 4.18879020478639 * cutneigh * cutneigh * cutneigh;
  int mn = (int)(expect * 1.5) + 24;
  dalloc(d.numneigh, np);
  dalloc(d.bpart, (size_t)std::max(bpa, 1) * np);   // >= 1 row: the step kernel loads before it masks
  dalloc(d.bshift, np);
  dev_alloc_neigh(d, mn);
  dalloc(d.pairtab, (size_t)6 * (ntypes + 1) * (ntypes + 1));
  d.nred_blocks = (n + 255) / 256 + 8;
  dalloc(d.partial, (size_t)d.nred_blocks * 16);
  dalloc(d.partial_a, (size_t)d.nred_blocks * 8);
  dalloc(d.lgsum, ((size_t)d.nred_blocks + 1) * 16);
  HIP_CHECK(hipHostMalloc((void **)&d.partial_h, (size_t)d.nred_blocks * 16 * sizeof(double)));
  dalloc(d.flags, NFLAGS);
  HIP_CHECK(hipHostMalloc((void **)&d.flags_h, (FLAG_SEQ_SLOT + 16) * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
  for (int k = 0; k < FLAG_SEQ_SLOT + 16; k++) d.flags_h[k] = 0;
  d.flags_seq = 0;
  d.bins_ready = false;
  d.cell_count_dirty = false;    // (freshly allocated arrays are zeroed)
  HIP_CHECK(hipHostGetDevicePointer((void **)&d.flags_h_dev, d.flags_h, 0));
  // LE fix scratch
  dalloc(d.xt, nt);
  dalloc(d.xht, nt);
  for (int k = 0; k < 16; k++) dalloc(d.le_i[k], nt);
  for (int k = 0; k < 2; k++) dalloc(d.le_d[k], nt);
  dalloc(d.le_bits, 3 * (nt / 64 + 16));     // three masks: base / accepted pairs, partner below, partner above
  dalloc(d.le_rng_state, (LE_MAX_FIXES + 1) * 100);      // (+ a scratch copy: chained barrier draws of fix extrusion, kernels_le.hip)
  dalloc(d.le_draws, 2 * nt);      // (fix extrusion with chained barrier draws: up to four per listing, a listing per two beads)
  dalloc(d.le_list, 4 * nt);
  dalloc(d.le_scan, std::max(nt, (size_t)d.ncells + 2) / 1024 + 16);
}

void dd_fast_halo_free(DeviceState &d);   // kernels_dd.hip
void dev_free(DeviceState &d) {
  dd_fast_halo_free(d);
  if (d.rng_stream) {
    (void)hipStreamSynchronize(d.rng_stream);
    if (d.stream) (void)hipStreamSynchronize(d.stream);
    for (int k = 0; k < 2; k++) { (void)hipEventDestroy(d.rng_done[k]); (void)hipEventDestroy(d.rng_consumed[k]); }
    (void)hipStreamDestroy(d.rng_stream);
    d.rng_stream = nullptr;
  }
  dfree(d.pos); dfree(d.pos_tmp); dfree(d.xhold); dfree(d.posf);
  for (int k = 0; k < 3; k++) { dfree(d.v[k]); dfree(d.v_tmp[k]); dfree(d.f[k]); }
  dfree(d.tag); dfree(d.tag_tmp); dfree(d.img); dfree(d.img_tmp);
  dfree(d.map); dfree(d.type_t); dfree(d.crank);
  dfree(d.num_bond); dfree(d.bond_type); dfree(d.bond_atom); dfree(d.nspecial); dfree(d.special); dfree(d.num_bond0); dfree(d.bond_type0); dfree(d.bond_atom0); dfree(d.bond_pack); dfree(d.bond_pack_p[0]); dfree(d.bond_pack_p[1]);
  if (d.angtab_dev) { (void)hipFree(d.angtab_dev); d.angtab_dev = nullptr; }
  dfree(d.gmask); dfree(d.lgrank);
  dfree(d.cell_of); dfree(d.cell_count); dfree(d.cell_start); dfree(d.cell_fill); dfree(d.scan_tmp); dfree(d.perm);
  dfree(d.neigh); dfree(d.numneigh); dfree(d.bpart); dfree(d.bshift); dfree(d.pairtab); dfree(d.partial); dfree(d.partial_a); dfree(d.lgsum);
  dfree(d.angle_pack); dfree(d.num_angle); dfree(d.angle_type); dfree(d.angle_a1); dfree(d.angle_a2); dfree(d.angle_a3); dfree(d.eff_n); dfree(d.eff_rec);
  if (d.partial_h) (void)hipHostFree(d.partial_h);
  d.partial_h = nullptr;
  dfree(d.flags);
  if (d.flags_h) (void)hipHostFree(d.flags_h);
  d.flags_h = nullptr;
  dfree(d.rng_state); dfree(d.rng_jump); dfree(d.rng_buf[0]); dfree(d.rng_buf[1]); d.rng_out = nullptr;
  dfree(d.rng_pool[0]); dfree(d.rng_pool[1]); dfree(d.rng_wstate); d.rng_W = 0; d.rng_batch_raw[0] = d.rng_batch_raw[1] = 0;
  dfree(d.xt); dfree(d.xht);
  dfree(d.gcell_start); dfree(d.gcell_count); dfree(d.sendlist[0]); dfree(d.sendlist[1]); dfree(d.migbuf[0]);
  dfree(d.migbuf[1]); dfree(d.migin); dfree(d.sendbuf); dfree(d.recvbuf); dfree(d.gdest); dfree(d.gtag_in); dfree(d.gone); dfree(d.phase); dfree(d.sendslot);
  dfree(d.gather_send); d.gather_recv = nullptr; d.gather_cap = 0;
  for (int k = 0; k < 16; k++) dfree(d.le_i[k]);
  for (int k = 0; k < 2; k++) dfree(d.le_d[k]);
  dfree(d.le_bits); dfree(d.le_rng_state); dfree(d.le_draws); dfree(d.le_list); dfree(d.le_scan);
  sort_scratch_free(d);
  if (d.stream) (void)hipStreamDestroy(d.stream);
  d.stream = nullptr;
}

// flags reach the host through a mapped pinned page written by a one-wave kernel (a blit-copy of 64 bytes costs
// ~18 us on this stack, a kernel + sync ~5 us)
// `reset` = bit mask of flags that are zeroed right after they were published (saves one memset launch per flag
// and phase: the consumer of a flag is always the host, which reads the published copy)
// The host does not call hipStreamSynchronize for these hand-overs (its wake-up costs 30-50 us, twice per rebuild):
// the kernel writes a sequence number behind the flags and the host spins on the mapped page.
__global__ void k_publish_flags(int *__restrict__ flags, int *__restrict__ host, unsigned reset, int seq) {
  // ONE wavefront: lane k writes flag k, then - behind the barrier and a system-scope release, which on this
  // hardware waits for every outstanding store of the wave - lane 0 writes the sequence number the host spins on.
  // The host must never see the new number next to old flags: flags and number sit in different 64-byte sectors
  // of the mapped page, and stores of one instruction have no order among themselves, hence the separate store.
  // (Sixteen serial stores from one lane cost 10 us over PCIe; this is one round trip.)
  const int k = threadIdx.x;
  if (k < NFLAGS) {
    int v = flags[k];
    __hip_atomic_store(&host[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((reset >> k) & 1u) flags[k] = 0;
  }
  __syncthreads();
  if (k == 0) __hip_atomic_store(&host[FLAG_SEQ_SLOT], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void stream_sync(DeviceState &d) {
  if (d.comm_watch) d.comm_watch->wait_stream(d.stream);
  else HIP_CHECK(hipStreamSynchronize(d.stream));
}
void publish_flags(DeviceState &d, unsigned reset) {
  const int seq = ++d.flags_seq;
  hipLaunchKernelGGL(k_publish_flags, dim3(1), dim3(64), 0, d.stream, d.flags, d.flags_h_dev, reset, seq);
}
void sync_flags(DeviceState &d, unsigned reset) {
  publish_flags(d, reset);
  wait_flags(d);
}
// waits for the LAST publish_flags (work enqueued behind it keeps running)
void wait_flags(DeviceState &d) {
  const int seq = d.flags_seq;
  static const bool spin = !getenv("LAMMPS_LE_NO_SPIN");
  if (spin) {
    volatile int *h = d.flags_h;
    auto t0 = std::chrono::steady_clock::now();
    long it = 0;
    while (h[FLAG_SEQ_SLOT] != seq) {
      if ((++it & 0xFFF) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;   // long-running
    }                                                                            // work (or a fault): block instead
    if (h[FLAG_SEQ_SLOT] == seq) { std::atomic_thread_fence(std::memory_order_acquire); return; }   // pairs with the release store
  }
  stream_sync(d);
}

}  // namespace lmp_le
