// bin_inl.h — cell of a position and the wave-aggregated cell counter, shared by the reneighbor kernels and the step
// kernel (which bins the positions it has just produced when a rebuild may follow, see k_step).
#pragma once
#include "device.h"

namespace lmp_le {

// Numbering of the (y, z) rows of cells.  A row's x-cells are always consecutive cell ids (the list build and the cell
// sort rely on one contiguous bead range per row); the ROWS are numbered in tiles of ROW_TILE x ROW_TILE instead of plain
// z-major order, so that the beads a group of concurrently running workgroups needs - their own rows and the rows +-1 in
// y and z - are a compact ~18 x 18 bundle of rows (2.6 MB of positions at 8M beads) instead of three whole z-layers
// (5.5 MB at 8M beads, more than an XCD's 4 MB L2).  tile = 0: plain z-major numbering (decomposed runs: the slab code
// relies on ghosts from below / above forming the first / last cell layers).
#ifndef ROW_TILE_SIZE
#define ROW_TILE_SIZE 16
#endif
constexpr int ROW_TILE = ROW_TILE_SIZE;   // measured at 8M beads: k_step 517 / 514 / 521 / 534 us for tiles of 8 / 16 / 32 / 64 rows (1M: no difference)
// (`tile` is 0 or ROW_TILE: the tiled branch divides by the compile-time constant - shifts; an integer division by a
// run-time value is ~40 instructions, and the list build numbers 18 rows per bead)
__device__ __forceinline__ int row_id(int ay, int az, int ncy, int ncz, int tile) {
  if (tile == 0) return az * ncy + ay;
  static_assert((ROW_TILE & (ROW_TILE - 1)) == 0, "ROW_TILE must be a power of two");
  const int ty = (int)((unsigned)ay / (unsigned)ROW_TILE), tz = (int)((unsigned)az / (unsigned)ROW_TILE);
  const int hy = min(ROW_TILE, ncy - ty * ROW_TILE), hz = min(ROW_TILE, ncz - tz * ROW_TILE);
  return tz * ROW_TILE * ncy + ty * ROW_TILE * hz + (az - tz * ROW_TILE) * hy + (ay - ty * ROW_TILE);
}
// cell coordinates of a (wrapped) position.  z cells are counted from zlo_ext (the bottom of this rank's slab + ghost
// shell; = box.lo[2] on one rank) with a periodic wrap, so owned and ghost beads of a slab land in one local grid.
__device__ __forceinline__ void cell_coords(const double4 &r, const Box &box, int ncx, int ncy, int ncz, double cix,
                                            double ciy, double ciz, double zlo_ext, int &cx, int &cy, int &cz) {
  double zrel = r.z - zlo_ext;
  if (zrel < 0.0) zrel += box.prd[2];
  if (zrel >= box.prd[2]) zrel -= box.prd[2];
  cx = (int)((r.x - box.lo[0]) * cix); cy = (int)((r.y - box.lo[1]) * ciy); cz = (int)(zrel * ciz);
  cx = min(max(cx, 0), ncx - 1); cy = min(max(cy, 0), ncy - 1); cz = min(max(cz, 0), ncz - 1);
}
__device__ __forceinline__ int cell_index(const double4 &r, const Box &box, int ncx, int ncy, int ncz, double cix,
                                          double ciy, double ciz, double zlo_ext, int tile) {
  int cx, cy, cz;
  cell_coords(r, box, ncx, ncy, ncz, cix, ciy, ciz, zlo_ext, cx, cy, cz);
  return row_id(cy, cz, ncy, ncz, tile) * ncx + cx;
}

// Domain::pbc for one coordinate triple (src/domain.cpp:528-645, orthogonal box): wrapped copy + image deltas
__device__ __forceinline__ void wrap_into_box(double4 &r, const Box &box, int &dix, int &diy, int &diz) {
  double *c = &r.x;
  int di[3] = {0, 0, 0};
#pragma unroll
  for (int d = 0; d < 3; d++) {
    double x = c[d];
    if (x < box.lo[d]) { x += box.prd[d]; di[d]--; }
    if (x >= box.hi[d]) { x -= box.prd[d]; x = fmax(x, box.lo[d]); di[d]++; }
    c[d] = x;
  }
  dix = di[0]; diy = di[1]; diz = di[2];
}

// arrival order of a bead inside its cell + the cell's count.  Beads arrive nearly cell-sorted: one returning atomic per
// run of equal cells inside the wavefront.  The lanes that call this are every STRIDE-th lane of the wavefront, from lane 0
// up to wherever the array ends (STRIDE = lanes per bead of the calling kernel); all of them must call it.
template <int STRIDE = 1>
__device__ __forceinline__ int count_into_cell(int cell, int *__restrict__ cell_count) {
  const int lane = threadIdx.x & 63;
  int prev = __shfl_up(cell, STRIDE, 64);                      // the previous calling lane's cell
  bool head = (lane < STRIDE) || (prev != cell);
  unsigned long long heads = __ballot(head);                   // (only calling lanes vote)
  unsigned long long act = __ballot(true);
  const unsigned long long upto = (2ull << lane) - 1ull;       // lanes 0..lane
  unsigned long long below = heads & upto;                     // heads at or below my lane
  int hl = 63 - __clzll((long long)below);                     // lane of my run's head
  unsigned long long after = heads & ~upto;
  int endl = after ? (__ffsll((long long)after) - 1) : 64;     // first head past my run
  const unsigned long long from_head = ~((1ull << hl) - 1ull);
  const unsigned long long run = act & from_head & (endl >= 64 ? ~0ull : ((1ull << endl) - 1ull));
  int base = 0;
  if (head) base = atomicAdd(&cell_count[cell], __popcll(run));
  base = __shfl(base, hl, 64);
  return base + __popcll(run & ((1ull << lane) - 1ull));       // calling lanes of my run in front of me
}

}  // namespace lmp_le
