// tools.cpp — host-side generator of synthetic start configurations (counterpart of the reference's tools/chain.f).
//
// tools/chain.f lays phantom random walks into the box (bond 0.97, rho* = 0.8442) and leaves the overlaps to a soft
// push-off run with styles this engine does not have.  What the benchmarks need from such a start is that chain
// neighbours are NOT memory neighbours: the serpentine-lattice start (lammps_le_amd.synth.lattice_chains) makes tag
// order = space order, which flatters every tag-indexed gather of the step kernel.  This generator produces an
// overlap-free, locally DISORDERED space-filling walk instead: the L^3 lattice is cut into B^3 blocks, the blocks are
// visited in serpentine order, and inside every block the path is a random Hamiltonian path produced by "backbite"
// moves (Mansfield 1982): pick an end of the path, pick a lattice neighbour v of it that is not its path neighbour,
// connect end-v and cut the link that closes the ring; the result is again a Hamiltonian path with a new end.  The
// start of a block's path is pinned to the site that faces the previous block's last site, its end to the face of
// the next block.  Consecutive sites are lattice neighbours everywhere, so every bond has the lattice spacing.
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace {
struct Rng {
  uint64_t s;
  uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); }
  int below(int n) { return (int)(((uint64_t)next() * (uint64_t)n) >> 32); }
};
struct Block {
  int B, M;
  std::vector<int> path, where;     // path[k] = site, where[site] = k ; site = (z*B + y)*B + x
  explicit Block(int b) : B(b), M(b * b * b), path(M), where(M) {
    int k = 0;
    for (int z = 0; z < B; z++)
      for (int yy = 0; yy < B; yy++) {
        int y = (z & 1) ? B - 1 - yy : yy;
        for (int xx = 0; xx < B; xx++) {
          int x = ((z * B + yy) & 1) ? B - 1 - xx : xx;
          path[k] = (z * B + y) * B + x; where[path[k]] = k; k++;
        }
      }
  }
  int neighbours(int site, int *out) const {
    int x = site % B, y = (site / B) % B, z = site / (B * B), n = 0;
    if (x > 0) out[n++] = site - 1;
    if (x < B - 1) out[n++] = site + 1;
    if (y > 0) out[n++] = site - B;
    if (y < B - 1) out[n++] = site + B;
    if (z > 0) out[n++] = site - B * B;
    if (z < B - 1) out[n++] = site + B * B;
    return n;
  }
  void reverse(int a, int b) {   // path[a..b]
    while (a < b) { int t = path[a]; path[a] = path[b]; path[b] = t; where[path[a]] = a; where[path[b]] = b; a++; b--; }
    if (a == b) where[path[a]] = a;
  }
  void backbite(Rng &r, bool at_end) {
    int nb[6];
    if (at_end) {
      int e = path[M - 1], n = neighbours(e, nb), v = nb[r.below(n)];
      if (v == path[M - 2]) return;
      reverse(where[v] + 1, M - 1);
    } else {
      int s = path[0], n = neighbours(s, nb), v = nb[r.below(n)];
      if (v == path[1]) return;
      reverse(0, where[v] - 1);
    }
  }
};
}  // namespace

// xyz[3*k .. 3*k+2] = lattice site of path position k, k < L^3.  L must be a multiple of the (even) block edge B.
extern "C" int lammps_le_tool_scrambled_path(int L, int B, int seed, int sweeps, int *xyz) {
  if (B < 2 || (B & 1) || L % B) return 1;
  const int nb = L / B, M = B * B * B;
  Rng rng{0x9E3779B97F4A7C15ull ^ ((uint64_t)(uint32_t)seed * 0xD1B54A32D192ED03ull)};
  for (int k = 0; k < 8; k++) rng.next();
  // serpentine order over the blocks: consecutive blocks share a face
  std::vector<int> border;
  for (int z = 0; z < nb; z++)
    for (int yy = 0; yy < nb; yy++) {
      int y = (z & 1) ? nb - 1 - yy : yy;
      for (int xx = 0; xx < nb; xx++) {
        int x = ((z * nb + yy) & 1) ? nb - 1 - xx : xx;
        border.push_back((z * nb + y) * nb + x);
      }
    }
  long long out = 0;
  int prev_site[3] = {-1, -1, -1};     // global lattice coordinates of the last site written
  for (size_t kb = 0; kb < border.size(); kb++) {
    const int b = border[kb], bx = b % nb, by = (b / nb) % nb, bz = b / (nb * nb);
    int dir[3] = {0, 0, 0};              // direction to the next block
    const bool last = kb + 1 == border.size();
    if (!last) {
      const int c = border[kb + 1];
      dir[0] = c % nb - bx; dir[1] = (c / nb) % nb - by; dir[2] = c / (nb * nb) - bz;
    }
    Block blk(B);
    for (long long m = 0; m < (long long)sweeps * M; m++) blk.backbite(rng, rng.next() & 1u);
    auto coords = [&](int site, int *g) { g[0] = bx * B + site % B; g[1] = by * B + (site / B) % B; g[2] = bz * B + site / (B * B); };
    long long guard = 0;
    if (kb > 0) {
      // pin the start: the site of this block that is a lattice neighbour of the previous block's last site
      for (;;) {
        int g[3]; coords(blk.path[0], g);
        const int d = abs(g[0] - prev_site[0]) + abs(g[1] - prev_site[1]) + abs(g[2] - prev_site[2]);
        if (d == 1) break;
        blk.backbite(rng, false);
        if (++guard > 4000000ll * M) return 2;
      }
    }
    if (!last) {
      // pin the end to the face of the next block, moving only the end (the start stays where it is)
      guard = 0;
      for (;;) {
        const int e = blk.path[M - 1], x = e % B, y = (e / B) % B, z = e / (B * B);
        const bool on_face = (dir[0] > 0 && x == B - 1) || (dir[0] < 0 && x == 0) || (dir[1] > 0 && y == B - 1) ||
                             (dir[1] < 0 && y == 0) || (dir[2] > 0 && z == B - 1) || (dir[2] < 0 && z == 0);
        if (on_face) break;
        blk.backbite(rng, true);
        if (++guard > 4000000ll * M) return 3;
      }
    }
    for (int k = 0; k < M; k++) {
      int g[3]; coords(blk.path[k], g);
      xyz[3 * out] = g[0]; xyz[3 * out + 1] = g[1]; xyz[3 * out + 2] = g[2];
      out++;
      if (k == M - 1) { prev_site[0] = g[0]; prev_site[1] = g[1]; prev_site[2] = g[2]; }
    }
  }
  return 0;
}
