// kernels_sort.hip — Atom::sort on the device.
//
// The reference re-sorts its owned atoms every `atom_modify sort N` steps (default 1000) on the first reneighbor
// at/after `nextsort` (src/atom.cpp:2003-2094): a stable counting sort by bin (bins of 1/2 cutneighmax over the box,
// setup_sort_bins :2100-2208; bin index = iz*nby*nbx + iy*nbx + ix, :2060-2075), atoms of one bin keep their previous
// relative order.  The engine never moves its arrays for this - its physical order is its own cell order - but the
// reference's local index decides which Langevin draws a bead gets (draw 3*i+c of a call goes to local atom i,
// src/fix_langevin.cpp:670-674) and in which order the LE fixes visit things, so the engine keeps `crank[tag]` = the
// local index the reference would have.  Round 1 did this on the host (download + std::sort, ~150 ms per sort at 1M
// beads); here it is one key kernel, one device radix sort (rocPRIM) on (bin, previous rank) and one scatter.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "device.h"

namespace lmp_le {

constexpr int BLOCK = 256;

__global__ __launch_bounds__(BLOCK) void k_sort_keys(int n, const double4 *__restrict__ pos, const int *__restrict__ tag,
                                                     const int *__restrict__ crank, Box box, int nbx, int nby, int nbz,
                                                     double bix, double biy, double biz, unsigned long long ntot,
                                                     unsigned long long *__restrict__ keys, int *__restrict__ vals) {
  int p = blockIdx.x * BLOCK + threadIdx.x;
  if (p >= n) return;
  const double4 r = pos[p];
  const int t = tag ? tag[p] : p + 1;       // (decomposed runs: `pos` is the gathered table by tag, entry p belongs to tag p + 1)
  // src/atom.cpp:2060-2071: ix = static_cast<int>((x - bboxlo) * bininv), clamped to [0, nbin - 1]
  int ix = (int)((r.x - box.lo[0]) * bix), iy = (int)((r.y - box.lo[1]) * biy), iz = (int)((r.z - box.lo[2]) * biz);
  ix = min(max(ix, 0), nbx - 1); iy = min(max(iy, 0), nby - 1); iz = min(max(iz, 0), nbz - 1);
  const unsigned long long ibin = ((unsigned long long)iz * nby + iy) * nbx + ix;
  keys[p] = ibin * ntot + (unsigned long long)crank[t];     // stable within a bin = ordered by the previous local index
  vals[p] = t;
}
__global__ __launch_bounds__(BLOCK) void k_sort_ranks(int n, const int *__restrict__ sorted_tags, int *__restrict__ crank) {
  int r = blockIdx.x * BLOCK + threadIdx.x;
  if (r < n) crank[sorted_tags[r]] = r;
}

// fix langevin on a group: the rank of every member among the members in the new local order (its draws are handed out in
// that order, src/fix_langevin.cpp:660-674).  flag[r] = is the bead at local rank r a member -> exclusive scan -> lgrank[tag]
__global__ __launch_bounds__(BLOCK) void k_sort_member_flags(int n, const int *__restrict__ sorted_tags, const int *__restrict__ gmask,
                                                             int bit, int *__restrict__ flag) {
  int r = blockIdx.x * BLOCK + threadIdx.x;
  if (r < n) flag[r] = (gmask[sorted_tags[r]] & bit) ? 1 : 0;
}
__global__ __launch_bounds__(BLOCK) void k_sort_member_ranks(int n, const int *__restrict__ sorted_tags, const int *__restrict__ flag,
                                                             const int *__restrict__ scan, int *__restrict__ lgrank) {
  int r = blockIdx.x * BLOCK + threadIdx.x;
  if (r < n) lgrank[sorted_tags[r]] = flag[r] ? scan[r] : 0;
}

struct SortScratch {
  unsigned long long *keys[2] = {nullptr, nullptr};
  int *vals[2] = {nullptr, nullptr};
  void *temp = nullptr;
  size_t temp_bytes = 0;
  int cap = 0;
};
static SortScratch &scratch_of(DeviceState &d) {
  if (!d.sort_scratch) d.sort_scratch = new SortScratch();
  return *(SortScratch *)d.sort_scratch;
}
void sort_scratch_free(DeviceState &d) {
  if (!d.sort_scratch) return;
  SortScratch &s = *(SortScratch *)d.sort_scratch;
  for (int k = 0; k < 2; k++) { if (s.keys[k]) (void)hipFree(s.keys[k]); if (s.vals[k]) (void)hipFree(s.vals[k]); }
  if (s.temp) (void)hipFree(s.temp);
  delete &s;
  d.sort_scratch = nullptr;
}

// crank[tag] := position of the bead in the reference's freshly sorted local order.  Positions must be the wrapped ones
// of the reneighbor this sort belongs to (the caller runs it right behind Engine::reneighbor).
// `by_tag` (decomposed runs): every rank sorts ALL beads from the all-gathered positions by tag (d.xt, filled by
// dd_gather_positions right before), so that the replicated `crank` is the one-rank order on every rank.
void launch_atom_sort(DeviceState &d, const int nb[3], const double binv[3], bool by_tag) {
  const int n = by_tag ? d.maxtag : d.n;
  SortScratch &s = scratch_of(d);
  if (s.cap < n) {
    for (int k = 0; k < 2; k++) {
      if (s.keys[k]) (void)hipFree(s.keys[k]);
      if (s.vals[k]) (void)hipFree(s.vals[k]);
      HIP_CHECK(hipMalloc((void **)&s.keys[k], (size_t)d.npad * sizeof(unsigned long long)));
      HIP_CHECK(hipMalloc((void **)&s.vals[k], (size_t)d.npad * sizeof(int)));
    }
    s.cap = d.npad;
    if (s.temp) (void)hipFree(s.temp);
    s.temp = nullptr; s.temp_bytes = 0;
  }
  const unsigned long long ntot = (unsigned long long)d.maxtag + 1;
  const unsigned long long nbins = (unsigned long long)nb[0] * nb[1] * nb[2];
  int bits = 1;
  while (bits < 64 && ((nbins * ntot) >> bits) != 0ull) bits++;
  const int grid = (n + BLOCK - 1) / BLOCK;
  hipLaunchKernelGGL(k_sort_keys, dim3(grid), dim3(BLOCK), 0, d.stream, n, by_tag ? d.xt + 1 : d.pos, by_tag ? (const int *)nullptr : d.tag, d.crank, d.box, nb[0], nb[1], nb[2],
                     binv[0], binv[1], binv[2], ntot, s.keys[0], s.vals[0]);
  size_t need = 0;
  HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, s.keys[0], s.keys[1], s.vals[0], s.vals[1], (size_t)n, 0u, (unsigned)bits, d.stream));
  if (need > s.temp_bytes) {
    if (s.temp) (void)hipFree(s.temp);
    HIP_CHECK(hipMalloc(&s.temp, need));
    s.temp_bytes = need;
  }
  HIP_CHECK(rocprim::radix_sort_pairs(s.temp, need, s.keys[0], s.keys[1], s.vals[0], s.vals[1], (size_t)n, 0u, (unsigned)bits, d.stream));
  hipLaunchKernelGGL(k_sort_ranks, dim3(grid), dim3(BLOCK), 0, d.stream, n, s.vals[1], d.crank);
  if (d.lg_grouped) {      // (keys[0] / vals[0] are free again: flags and their scan)
    int *flag = s.vals[0], *scan = reinterpret_cast<int *>(s.keys[0]);
    hipLaunchKernelGGL(k_sort_member_flags, dim3(grid), dim3(BLOCK), 0, d.stream, n, s.vals[1], d.gmask, d.lg_bit, flag);
    size_t need2 = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, need2, flag, scan, 0, (size_t)n, rocprim::plus<int>(), d.stream));
    if (need2 > s.temp_bytes) {
      if (s.temp) (void)hipFree(s.temp);
      HIP_CHECK(hipMalloc(&s.temp, need2));
      s.temp_bytes = need2;
    }
    HIP_CHECK(rocprim::exclusive_scan(s.temp, need2, flag, scan, 0, (size_t)n, rocprim::plus<int>(), d.stream));
    hipLaunchKernelGGL(k_sort_member_ranks, dim3(grid), dim3(BLOCK), 0, d.stream, n, s.vals[1], flag, scan, d.lgrank);
  }
}

}  // namespace lmp_le
