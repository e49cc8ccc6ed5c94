// kernels_rng.hip — RanMars streams on the device, bit-exact with src/random_mars.cpp:81-95.
//
// FixLangevin draws 3 uniforms per bead per call from ONE serial generator
// (src/fix_langevin.cpp:670-674); the value used for component c of the bead at canonical
// position r in the k-th call is draw 3N*k + 3r + c of the stream.  The stream is cut into
// blocks of B draws; block b keeps the 97-value history window in front of its first draw.  One
// wavefront per block (a) generates its B draws 33 at a time (the lag-33 term forbids more),
// (b) writes them as 24-bit integers, (c) advances its window by the fixed per-call jump 3N with
// the precomputed polynomial x^(3N) mod (x^97 + x^64 - 1) (97 dot products of length 97).
#include "device.h"
#include <cstring>
#include <vector>
#include <algorithm>

namespace lmp_le {

static constexpr uint32_t M24 = 0xFFFFFFu;

__device__ __forceinline__ uint32_t c_of_dev(unsigned long long n) {
  const unsigned long long CM = 16777213ull, CD = 7654321ull, C0 = 362436ull;
  unsigned long long dec = (((n + 1) % CM) * CD) % CM;
  return (uint32_t)((C0 + CM - dec) % CM);
}

constexpr int RNG_MAXB = 3072;

__global__ __launch_bounds__(64) void k_rng_langevin(int B, long long total, unsigned long long first_raw,
                                                     uint32_t *__restrict__ state, const uint32_t *__restrict__ jump,
                                                     uint32_t *__restrict__ out) {
  __shared__ uint32_t buf[97 + RNG_MAXB];
  __shared__ uint32_t a[97];
  int b = blockIdx.x, lane = threadIdx.x;
  long long off = (long long)b * B;
  int nout = (int)min((long long)B, total - off);
  int G = max(nout, 96);
  for (int k = lane; k < 97; k += 64) { buf[k] = state[(size_t)b * 97 + k]; a[k] = jump[k]; }
  __syncthreads();
  for (int base = 0; base < G; base += 33) {
    int i = base + lane;
    if (lane < 33 && i < G) buf[97 + i] = (buf[i] - buf[i + 64]) & M24;
    __syncthreads();
  }
  {
    // arithmetic sequence term: one exact evaluation per lane, then steps of 64 draws (c -= 64*cd mod cm)
    const int CM = 16777213, STEP64 = (int)((64ull * 7654321ull) % 16777213ull);
    int c = (int)c_of_dev(first_raw + (unsigned long long)(off + lane));
    for (int i = lane; i < nout; i += 64) {
      out[off + i] = (buf[97 + i] - (uint32_t)c) & M24;
      c -= STEP64;
      if (c < 0) c += CM;
    }
  }
  // window for the next call: y'[i] = sum_j a[j] * y[i + j]
  uint32_t acc0 = 0, acc1 = 0;
  int i1 = lane + 64;
  for (int j = 0; j < 97; j++) {
    uint32_t aj = a[j];
    acc0 += aj * buf[lane + j];
    if (i1 < 97) acc1 += aj * buf[i1 + j];
  }
  state[(size_t)b * 97 + lane] = acc0 & M24;
  if (i1 < 97) state[(size_t)b * 97 + i1] = acc1 & M24;
}

// ------------------------------------------------------------------------------------------
// Batch generator (default).  The draw stream does not depend on the simulation state, so it is produced W calls
// ahead: wavefront w of a batch generates ALL 3N draws of call (base + w) serially — no per-block jump — with the
// twice-substituted recurrence  y_n = y_{n-97} - y_{n-130} + y_{n-66}  (min lag 66 >= 64: one full wavefront per
// dependent step), then moves its window (W-1) calls ahead with x^((W-1)*3N) mod (x^97 + x^64 - 1).  A batch is a
// latency-bound trickle (one wave per block, ~1 KB LDS) that runs on rng_stream underneath the step kernels of
// the previous batch; per consumed call nothing is launched at all.
// ring index of y_i (i = draw index within this wave's call) is (i + 97) & 255.
// Decomposed runs (`need` != nullptr): a segment none of this rank's owned or ghost beads draws from is SKIPPED - its
// window is moved W calls ahead with x^(W * total) straight from the segment's start (96 recurrence values + the same
// 97 x 97 products), nothing is generated or stored.  `st_out` == nullptr: late generation of segments a validation found
// missing (a bead migrated in from beyond the ghost shell): generate from the kept input windows, leave the state alone.
__global__ __launch_bounds__(64) void k_rng_calls(long long total, long long seglen, int nseg,
                                                  unsigned long long base_raw, const uint32_t *__restrict__ wstate,
                                                  uint32_t *__restrict__ st_out_all,
                                                  const uint32_t *__restrict__ jump, uint32_t *__restrict__ pool,
                                                  const int *__restrict__ need, int *__restrict__ genmask) {
  __shared__ uint32_t ring[256];
  __shared__ uint32_t a[97];
  const int CM = 16777213, STEP64 = (int)((64ull * 7654321ull) % 16777213ull);
  // One wavefront per (call, segment): a call of 3N draws is cut into `nseg` segments of `seglen` draws (the last one
  // shorter), each with its own 97-value window in front of its first draw, so that the generator's throughput does not
  // hang on ONE wavefront per call when N is large (an 8M-bead call is 24 M dependent draws: 0.66 ms).
  const int w = blockIdx.x / nseg, sg = blockIdx.x % nseg, lane = threadIdx.x;
  const long long seg0 = (long long)sg * seglen;
  const long long len = min(seglen, total - seg0);             // draws of this segment
  const uint32_t *st = wstate + (size_t)blockIdx.x * 97;
  uint32_t *st_out = st_out_all ? st_out_all + (size_t)blockIdx.x * 97 : nullptr;
  uint32_t *out = pool + (size_t)w * total + seg0;
  const bool gen = need == nullptr || need[sg] != 0;
  if (!st_out && gen && w == 0 && lane == 0) genmask[sg] = 1;     // late generation: the pool now holds this segment too
  if (!gen) {
    if (!st_out) return;
    // skip: y_0 .. y_95 behind the window (three steps of the plain recurrence), then the window W calls later
    for (int k = lane; k < 97; k += 64) { ring[k] = st[k]; a[k] = jump[2 * 97 + k]; }
    __syncthreads();
    for (int base = 0; base < 96; base += 33) {
      const int i = base + lane;
      if (lane < 33 && i < 96) ring[97 + i] = (ring[i] - ring[i + 64]) & M24;
      __syncthreads();
    }
    uint32_t s0 = 0, s1 = 0;
    const int j1 = lane + 64;
    for (int j = 0; j < 97; j++) {
      const uint32_t aj = a[j];
      s0 += aj * ring[lane + j];
      if (j1 < 97) s1 += aj * ring[j1 + j];
    }
    st_out[lane] = s0 & M24;
    if (j1 < 97) st_out[j1] = s1 & M24;
    return;
  }
  unsigned long long raw0 = base_raw + (unsigned long long)w * (unsigned long long)total + (unsigned long long)seg0;
  const uint32_t *jp = jump + (sg == nseg - 1 ? 97 : 0);        // the last segment is shorter: its own jump distance
  for (int k = lane; k < 97; k += 64) { ring[k] = st[k]; a[k] = jp[k]; }
  __syncthreads();
  if (lane < 33) {   // draws 0..32 with the plain recurrence (the 64-wide form needs y_{i-130}, i >= 33)
    uint32_t y = (ring[lane] - ring[lane + 64]) & M24;
    ring[lane + 97] = y;
    if (lane < len) __builtin_nontemporal_store((uint32_t)((y - c_of_dev(raw0 + lane)) & M24), &out[lane]);
  }
  __syncthreads();
  long long G = len + 96;   // 96 values past the end feed the jump below
  int c = (int)c_of_dev(raw0 + 33ull + (unsigned long long)lane);
  for (long long base = 33; base < G; base += 64) {
    long long i = base + lane;
    int r = (int)(i & 255);
    uint32_t y = (ring[r] - ring[(r - 33) & 255] + ring[(r + 31) & 255]) & M24;   // y_{i-97} - y_{i-130} + y_{i-66}
    ring[(r + 97) & 255] = y;
    if (i < len) __builtin_nontemporal_store((uint32_t)((y - (uint32_t)c) & M24), &out[i]);   // 6 GB per batch, read once 1..512 steps later: keep it out of the caches the step kernel lives in
    c -= STEP64;
    if (c < 0) c += CM;
    __syncthreads();
  }
  // window of the same segment W calls later: y'[k] = sum_j a[j] * y_{len - 97 + k + j} with a = x^(W*total - len);
  // ring index of y_{len-97+m} is (len+m)&255
  int t0 = (int)(len & 255);
  uint32_t acc0 = 0, acc1 = 0;
  int i1 = lane + 64;
  for (int j = 0; j < 97; j++) {
    uint32_t aj = a[j];
    acc0 += aj * ring[(t0 + lane + j) & 255];
    if (i1 < 97) acc1 += aj * ring[(t0 + i1 + j) & 255];
  }
  if (st_out) {
    st_out[lane] = acc0 & M24;
    if (i1 < 97) st_out[i1] = acc1 & M24;
  }
}

// segments that hold draws of the beads in [0, m) (owned, then ghosts): 3 draws per bead at 3 * canonical rank
__global__ __launch_bounds__(256) void k_rng_mark(int m, const int *__restrict__ tag, const int *__restrict__ crank,
                                                  long long seglen, int *__restrict__ need) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= m) return;
  const int t = tag[p];
  const long long r = crank ? crank[t] : t - 1;
  need[(int)((3 * r) / seglen)] = 1;
}
// owned beads whose segment a live pool does not hold: wanted[pool][seg] = 1, FLAG_RNG_MISS
__global__ __launch_bounds__(256) void k_rng_validate(int n, const int *__restrict__ tag, const int *__restrict__ crank,
                                                      long long seglen, int nseg, const int *__restrict__ gen0,
                                                      const int *__restrict__ gen1, int live0, int live1,
                                                      int *__restrict__ late, int *__restrict__ flags) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  rng_validate_bead(RngValidateArgs{crank, seglen, nseg, gen0, gen1, live0, live1, late}, tag[p], flags);
}

// the table a bead's draws are addressed by: canonical rank (nullptr: tag - 1), or - fix langevin on a group - the rank among
// the group's members (non-members hold 0 there: they mark / check the first segment, which costs nothing but that segment)
static const int *rng_rank_table(const DeviceState &d) {
  if (d.lg_grouped) return d.lgrank;
  return d.ident_order ? (const int *)nullptr : d.crank;
}
static void rng_free_batch(DeviceState &d) {
  for (int k = 0; k < 2; k++) if (d.rng_pool[k]) { (void)hipFree(d.rng_pool[k]); d.rng_pool[k] = nullptr; }
  if (d.rng_wstate) { (void)hipFree(d.rng_wstate); d.rng_wstate = nullptr; }
  if (d.rng_need) { (void)hipFree(d.rng_need); d.rng_need = nullptr; }
  if (d.rng_late) { (void)hipFree(d.rng_late); d.rng_late = nullptr; }
  for (int k = 0; k < 2; k++) if (d.rng_gen[k]) { (void)hipFree(d.rng_gen[k]); d.rng_gen[k] = nullptr; }
  d.rng_batch_raw[0] = d.rng_batch_raw[1] = 0;
}

void rng_langevin_setup(DeviceState &d, RanMarsInt &host_rng, int natoms) {
  long long total = 3ll * natoms;
  if (!d.rng_stream) {
    // lowest priority: the generator is a background trickle and must not delay the step kernels' workgroups
    int prio_lo = 0, prio_hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIP_CHECK(hipStreamCreateWithPriority(&d.rng_stream, hipStreamNonBlocking, prio_lo));
    for (int k = 0; k < 2; k++) {
      HIP_CHECK(hipEventCreateWithFlags(&d.rng_done[k], hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&d.rng_consumed[k], hipEventDisableTiming));
    }
  }
  HIP_CHECK(hipStreamSynchronize(d.rng_stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));
  d.rng_origin = host_rng;
  d.rng_out = nullptr;
  const char *mode = getenv("LAMMPS_LE_RNG_MODE");
  d.rng_mode = (mode && !strcmp(mode, "block")) ? 0 : 1;
  if (d.rng_mode == 1) {
    // calls per batch: as many wavefronts as hide each other's latency, bounded by 2 pools of W*3N draws <= 16 GiB.
    // Decomposed runs: every rank generates the WHOLE stream (3 * N_total draws per call, one wavefront per call), while
    // its steps get shorter with the rank count - at 8M beads a batch of 89 calls takes 59 ms, i.e. 1.5 calls per ms,
    // the rate ONE GPU consumes them at; 64 GiB of pools (of 288) let 4x as many calls run side by side.  (Generating only
    // the stream segments that hold a rank's own tags is the real fix: DESIGN.md section 6.)
    const long long pool_cap = d.dd ? (64ll << 30) : (16ll << 30);
    long long w = pool_cap / (8 * total);
    int W = (int)std::min<long long>(512, std::max<long long>(8, w));
    if (getenv("LAMMPS_LE_RNG_W")) W = std::max(2, atoi(getenv("LAMMPS_LE_RNG_W")));
    // decomposed runs cut a call into segments of <= 65535 draws and generate only the ones that hold draws of owned or
    // ghost beads (the rest of the stream is jumped over); one GPU generates everything, in ~1024 wavefronts per batch
    const bool skip = d.dd && !getenv("LAMMPS_LE_RNG_NO_SKIP");
    if (W != d.rng_W || total != d.rng_total || skip != d.rng_skip) {
      d.rng_skip = skip;
      rng_free_batch(d);
      d.rng_W = W; d.rng_total = total;
      // segments per call: ~1024 wavefronts per batch, segments no shorter than 64k draws (measured: 8M beads 1389 / 1430 /
      // 1437 / 1433 timesteps/s with 1 / 4 / 8 / 23 segments of W = 89 calls; 1M beads, W = 512: 12.20k / 12.24k / 11.93k with 1 / 2 / 4)
      int S = (int)std::min<long long>(std::max<long long>(1, 1024 / W), std::max<long long>(1, total / 65536));
      if (getenv("LAMMPS_LE_RNG_SEGMENTS")) S = std::max(1, atoi(getenv("LAMMPS_LE_RNG_SEGMENTS")));
      if ((long long)S * 256 > total) S = 1;
      d.rng_nseg = S;
      d.rng_seglen = (total + S - 1) / S;
      if (skip && !getenv("LAMMPS_LE_RNG_SEGMENTS")) { d.rng_seglen = std::min<long long>(65535, total); d.rng_nseg = S = (int)((total + d.rng_seglen - 1) / d.rng_seglen); }
      d.rng_seglen = (d.rng_seglen + 2) / 3 * 3;       // a bead's three draws never straddle two segments
      d.rng_nseg = S = (int)((total + d.rng_seglen - 1) / d.rng_seglen);
      for (int k = 0; k < 2; k++) HIP_CHECK(hipMalloc(&d.rng_pool[k], (size_t)W * total * sizeof(uint32_t)));
      HIP_CHECK(hipMalloc(&d.rng_wstate, 3 * (size_t)W * S * 97 * sizeof(uint32_t)));
      HIP_CHECK(hipMalloc(&d.rng_need, 2 * (size_t)S * sizeof(int)));      // one per pool
      HIP_CHECK(hipMalloc(&d.rng_late, 2 * (size_t)S * sizeof(int)));
      for (int k = 0; k < 2; k++) HIP_CHECK(hipMalloc(&d.rng_gen[k], (size_t)S * sizeof(int)));
      HIP_CHECK(hipMemset(d.rng_late, 0, 2 * (size_t)S * sizeof(int)));
      if (d.rng_jump) { (void)hipFree(d.rng_jump); d.rng_jump = nullptr; }
      HIP_CHECK(hipMalloc(&d.rng_jump, 3 * 97 * sizeof(uint32_t)));
      // a segment's window goes from the end of the segment to the start of the same segment W calls later; a skipped
      // segment's from its start (third polynomial)
      uint32_t a[3 * 97];
      const long long len_last = total - (long long)(S - 1) * d.rng_seglen;
      ranmars_jump_poly((uint64_t)total * (uint64_t)W - (uint64_t)d.rng_seglen, a);
      ranmars_jump_poly((uint64_t)total * (uint64_t)W - (uint64_t)len_last, a + 97);
      ranmars_jump_poly((uint64_t)total * (uint64_t)W, a + 2 * 97);
      HIP_CHECK(hipMemcpyAsync(d.rng_jump, a, sizeof a, hipMemcpyHostToDevice, d.stream));
      HIP_CHECK(hipStreamSynchronize(d.stream));
    }
    d.rng_batch_raw[0] = d.rng_batch_raw[1] = 0;   // the first call seeds the wave windows from rng_origin
    return;
  }
  rng_free_batch(d);
  d.rng_W = 0;
  // block length: the per-block cost is (B/33 dependent generation steps) + (a fixed 97x97 jump); small systems
  // are latency-bound -> short blocks, large ones amortise the jump over long blocks
  d.rng_B = natoms < 200000 ? 192 : 3072;
  if (getenv("LAMMPS_LE_RNG_B")) d.rng_B = std::min(RNG_MAXB, std::max(96, atoi(getenv("LAMMPS_LE_RNG_B"))));
  d.rng_nblocks = (int)((total + d.rng_B - 1) / d.rng_B);
  if (d.rng_state) { (void)hipFree(d.rng_state); d.rng_state = nullptr; }
  if (d.rng_jump) { (void)hipFree(d.rng_jump); d.rng_jump = nullptr; }
  for (int k = 0; k < 2; k++) if (d.rng_buf[k]) { (void)hipFree(d.rng_buf[k]); d.rng_buf[k] = nullptr; }
  d.rng_out = nullptr;
  HIP_CHECK(hipMalloc(&d.rng_state, (size_t)d.rng_nblocks * 97 * sizeof(uint32_t)));
  HIP_CHECK(hipMalloc(&d.rng_jump, 97 * sizeof(uint32_t)));
  for (int k = 0; k < 2; k++) HIP_CHECK(hipMalloc(&d.rng_buf[k], (size_t)total * sizeof(uint32_t)));
  d.rng_cur = 0; d.rng_out = d.rng_buf[0]; d.rng_ahead = false;
  std::vector<uint32_t> st((size_t)d.rng_nblocks * 97);
  RanMarsInt r = host_rng;   // positioned at the first draw of the next call
  for (int b = 0; b < d.rng_nblocks; b++) {
    for (int k = 0; k < 97; k++) st[(size_t)b * 97 + k] = r.w[k];
    if (b + 1 < d.rng_nblocks) for (int k = 0; k < d.rng_B; k++) r.next_raw();
  }
  uint32_t a[97];
  ranmars_jump_poly((uint64_t)total, a);
  HIP_CHECK(hipMemcpyAsync(d.rng_state, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
  HIP_CHECK(hipMemcpyAsync(d.rng_jump, a, sizeof a, hipMemcpyHostToDevice, d.stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));
}

// Draws for the call whose first raw index is `first_raw` end up in d.rng_out before the consumer kernel runs
// on d.stream.  The generator runs on its own stream one call AHEAD (the stream is deterministic: the next
// call always starts 3N draws later), so it overlaps the force kernel of the previous step; the two draw
// buffers alternate and events order producer / consumer.
static void rng_generate(DeviceState &d, int buf, uint64_t first_raw) {
  long long total = 3ll * d.ntotal;
  // the buffer may still be read by the consumer of two calls ago
  HIP_CHECK(hipStreamWaitEvent(d.rng_stream, d.rng_consumed[buf], 0));
  hipLaunchKernelGGL(k_rng_langevin, dim3(d.rng_nblocks), dim3(64), 0, d.rng_stream, d.rng_B, total,
                     (unsigned long long)first_raw, d.rng_state, d.rng_jump, d.rng_buf[buf]);
  HIP_CHECK(hipEventRecord(d.rng_done[buf], d.rng_stream));
}
static hipEvent_t rng_mark_event() {
  static thread_local hipEvent_t ev = nullptr;
  if (!ev) HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  return ev;
}
// `batch_no`: 0, 1, 2, .. since the last seeding; batch b starts from window copy b % 3 and leaves copy (b + 1) % 3, so
// the input windows of the two live batches survive for a late generation
static void rng_gen_batch(DeviceState &d, int pool, uint64_t base_raw, bool wait_consumed, long long batch_no) {
  const size_t copy = (size_t)d.rng_W * d.rng_nseg * 97;
  int *need = d.rng_need + (size_t)pool * d.rng_nseg;
  if (d.rng_skip) {
    // what this rank will draw from, as of now: owned beads and ghosts (a bead beyond the ghost shell that arrives while
    // the batch is live is caught by rng_validate_owned at the rebuild that brings it).  The pool's previous batch has
    // been consumed, i.e. the main stream is past it: its `need` / `gen` words are free.  gen is written on the MAIN
    // stream, where the validation reads it (the batch itself runs on rng_stream for milliseconds).
    HIP_CHECK(hipMemsetAsync(need, 0, (size_t)d.rng_nseg * sizeof(int), d.stream));
    // (test hook LAMMPS_LE_TEST_RNG_NO_GHOST_MARK: owned beads only, so that every bead that migrates in misses its segment
    //  and the validation + late generation path has to produce it)
    const bool owned_only = getenv("LAMMPS_LE_TEST_RNG_NO_GHOST_MARK") != nullptr;      // (read per batch: tests share a process)
    const int m = d.n + (owned_only ? 0 : d.nghost);
    hipLaunchKernelGGL(k_rng_mark, dim3(std::max(1, (m + 255) / 256)), dim3(256), 0, d.stream, m, d.tag,
                       rng_rank_table(d), d.rng_seglen, need);
    HIP_CHECK(hipMemcpyAsync(d.rng_gen[pool], need, (size_t)d.rng_nseg * sizeof(int), hipMemcpyDeviceToDevice, d.stream));
    HIP_CHECK(hipEventRecord(rng_mark_event(), d.stream));
    HIP_CHECK(hipStreamWaitEvent(d.rng_stream, rng_mark_event(), 0));
  }
  if (wait_consumed) HIP_CHECK(hipStreamWaitEvent(d.rng_stream, d.rng_consumed[pool], 0));
  hipLaunchKernelGGL(k_rng_calls, dim3(d.rng_W * d.rng_nseg), dim3(64), 0, d.rng_stream, d.rng_total, d.rng_seglen, d.rng_nseg,
                     (unsigned long long)base_raw, d.rng_wstate + (size_t)(batch_no % 3) * copy,
                     d.rng_wstate + (size_t)((batch_no + 1) % 3) * copy, d.rng_jump, d.rng_pool[pool],
                     d.rng_skip ? need : (const int *)nullptr, (int *)nullptr);
  HIP_CHECK(hipEventRecord(d.rng_done[pool], d.rng_stream));
  d.rng_batch_raw[pool] = base_raw;
  d.rng_batch_no[pool] = batch_no;
}
int rng_segments_held(DeviceState &d) {     // segments of a call the current pool holds (= all of them unless skipping)
  if (!d.rng_skip || !d.rng_gen[0]) return d.rng_nseg;
  std::vector<int> g((size_t)d.rng_nseg);
  HIP_CHECK(hipStreamSynchronize(d.stream));
  HIP_CHECK(hipMemcpy(g.data(), d.rng_gen[d.rng_pool_cur], g.size() * sizeof(int), hipMemcpyDeviceToHost));
  int c = 0;
  for (int v : g) c += v != 0;
  return c;
}
void rng_validate_owned(DeviceState &d) {
  if (!d.rng_skip || d.rng_mode != 1 || !d.rng_gen[0] || (!d.rng_batch_raw[0] && !d.rng_batch_raw[1])) return;
  hipLaunchKernelGGL(k_rng_validate, dim3(std::max(1, (d.n + 255) / 256)), dim3(256), 0, d.stream, d.n, d.tag,
                     rng_rank_table(d), d.rng_seglen, d.rng_nseg, d.rng_gen[0], d.rng_gen[1],
                     d.rng_batch_raw[0] ? 1 : 0, d.rng_batch_raw[1] ? 1 : 0, d.rng_late, d.flags);
}
RngValidateArgs rng_validate_args(DeviceState &d) {
  if (!d.rng_skip || d.rng_mode != 1 || !d.rng_gen[0] || (!d.rng_batch_raw[0] && !d.rng_batch_raw[1]))
    return RngValidateArgs{nullptr, 1, 0, nullptr, nullptr, 0, 0, nullptr};
  return RngValidateArgs{rng_rank_table(d), d.rng_seglen, d.rng_nseg, d.rng_gen[0], d.rng_gen[1],
                         d.rng_batch_raw[0] ? 1 : 0, d.rng_batch_raw[1] ? 1 : 0, d.rng_late};
}
void rng_late_generate(DeviceState &d) {
  // (rare: a bead reached this slab from beyond the ghost shell within the life of a batch, or the canonical ranks changed)
  const size_t copy = (size_t)d.rng_W * d.rng_nseg * 97;
  for (int pool = 0; pool < 2; pool++) {
    if (!d.rng_batch_raw[pool]) continue;
    HIP_CHECK(hipStreamWaitEvent(d.stream, d.rng_done[pool], 0));
    hipLaunchKernelGGL(k_rng_calls, dim3(d.rng_W * d.rng_nseg), dim3(64), 0, d.stream, d.rng_total, d.rng_seglen, d.rng_nseg,
                       (unsigned long long)d.rng_batch_raw[pool], d.rng_wstate + (size_t)(d.rng_batch_no[pool] % 3) * copy,
                       (uint32_t *)nullptr, d.rng_jump, d.rng_pool[pool], d.rng_late + (size_t)pool * d.rng_nseg, d.rng_gen[pool]);
  }
  HIP_CHECK(hipMemsetAsync(d.rng_late, 0, 2 * (size_t)d.rng_nseg * sizeof(int), d.stream));
  HIP_CHECK(hipMemsetAsync(d.flags + FLAG_RNG_MISS, 0, sizeof(int), d.stream));
}
// wave windows in front of calls first_raw + w*3N, w = 0..W-1, from the host generator; then two batches
static void rng_seed_batches(DeviceState &d, uint64_t first_raw) {
  HIP_CHECK(hipStreamSynchronize(d.rng_stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));
  RanMarsInt r = d.rng_origin;
  if (first_raw < r.n) throw LammpsError("internal: Langevin stream asked to go backwards");
  r.jump(first_raw - r.n);
  uint32_t a[97], al[97], y[193];
  ranmars_jump_poly((uint64_t)d.rng_total, a);          // call -> next call
  ranmars_jump_poly((uint64_t)d.rng_seglen, al);        // segment -> next segment of the same call
  auto jumped = [&](const uint32_t *win, const uint32_t *poly, uint32_t *res) {
    std::memcpy(y, win, 97 * sizeof(uint32_t));
    for (int i = 97; i < 193; i++) y[i] = (y[i - 97] - y[i - 33]) & M24;
    for (int i = 0; i < 97; i++) {
      uint32_t acc = 0;
      for (int j = 0; j < 97; j++) acc += poly[j] * y[i + j];
      res[i] = acc & M24;
    }
  };
  const int S = d.rng_nseg;
  std::vector<uint32_t> st((size_t)d.rng_W * S * 97);
  uint32_t cur[97], nxt[97];
  for (int w = 0; w < d.rng_W; w++) {
    std::memcpy(cur, r.w, sizeof cur);
    for (int sg = 0; sg < S; sg++) {
      std::memcpy(&st[((size_t)w * S + sg) * 97], cur, sizeof cur);
      if (sg + 1 < S) { jumped(cur, al, nxt); std::memcpy(cur, nxt, sizeof cur); }
    }
    if (w + 1 == d.rng_W) break;
    jumped(r.w, a, nxt);
    std::memcpy(r.w, nxt, sizeof nxt);
  }
  HIP_CHECK(hipMemcpyAsync(d.rng_wstate, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice, d.rng_stream));
  HIP_CHECK(hipStreamSynchronize(d.rng_stream));
  uint64_t span = (uint64_t)d.rng_total * (uint64_t)d.rng_W;
  rng_gen_batch(d, 0, first_raw, false, 0);
  rng_gen_batch(d, 1, first_raw + span, false, 1);
  d.rng_pool_cur = 0;
  HIP_CHECK(hipStreamWaitEvent(d.stream, d.rng_done[0], 0));
}
void launch_rng_langevin(DeviceState &d, uint64_t first_raw) {
  if (d.rng_mode == 1) {
    uint64_t total = (uint64_t)d.rng_total, span = total * (uint64_t)d.rng_W;
    auto inside = [&](int q) {
      uint64_t b = d.rng_batch_raw[q];
      return b && first_raw >= b && first_raw < b + span && (first_raw - b) % total == 0;
    };
    int p = d.rng_pool_cur;
    if (!inside(p)) {
      int q = p ^ 1;
      if (inside(q)) {
        // every consumer of pool p is already enqueued on d.stream: refill it with the batch after q
        HIP_CHECK(hipEventRecord(d.rng_consumed[p], d.stream));
        rng_gen_batch(d, p, d.rng_batch_raw[q] + span, true, d.rng_batch_no[q] + 1);
        HIP_CHECK(hipStreamWaitEvent(d.stream, d.rng_done[q], 0));
        d.rng_pool_cur = p = q;
      } else {
        rng_seed_batches(d, first_raw);
        p = 0;
      }
    }
    d.rng_out = d.rng_pool[p] + (first_raw - d.rng_batch_raw[p]);
    return;
  }
  if (d.ntotal < 200000) {
    // small systems are launch-bound: generate in order on the main stream (no events, no second stream)
    long long total = 3ll * d.ntotal;
    d.rng_out = d.rng_buf[0];
    hipLaunchKernelGGL(k_rng_langevin, dim3(d.rng_nblocks), dim3(64), 0, d.stream, d.rng_B, total,
                       (unsigned long long)first_raw, d.rng_state, d.rng_jump, d.rng_out);
    return;
  }
  if (!d.rng_ahead) rng_generate(d, d.rng_cur, first_raw);
  else d.rng_cur ^= 1;                                   // generated ahead during the previous call
  d.rng_out = d.rng_buf[d.rng_cur];
  HIP_CHECK(hipStreamWaitEvent(d.stream, d.rng_done[d.rng_cur], 0));
  // run ahead: the following call's draws
  rng_generate(d, d.rng_cur ^ 1, first_raw + 3ull * (uint64_t)d.ntotal);
  d.rng_ahead = true;
}
// to be called right after the consumer kernel of the current draws has been enqueued on d.stream
void rng_langevin_consumed(DeviceState &d) {
  if (d.rng_mode == 1 || d.ntotal < 200000) return;
  HIP_CHECK(hipEventRecord(d.rng_consumed[d.rng_cur], d.stream));
}

// ------------------------------------------------------------------------------------------
// serial-stream generator for the LE fixes: `count` (device-resident) draws from one RanMars state,
// one wavefront, 33 values per dependent step through a 256-entry LDS ring.
// state layout: w[0..96] = y_{n-97..n-1}, [97] = n low, [98] = n high
__global__ __launch_bounds__(64) void k_ranmars_gen(uint32_t *__restrict__ state, const int *__restrict__ count_ptr,
                                                    uint32_t *__restrict__ out, int maxout) {
  __shared__ uint32_t ring[256];
  int lane = threadIdx.x;
  int count = min(*count_ptr, maxout);
  unsigned long long n0 = (unsigned long long)state[97] | ((unsigned long long)state[98] << 32);
  for (int k = lane; k < 97; k += 64) ring[k] = state[k];    // position i holds y_{n0-97+i}
  __syncthreads();
  // the first 33 draws with the plain recurrence, then 64 per dependent step with the twice-substituted one
  // (y_n = y_{n-97} - y_{n-130} + y_{n-66}, minimum lag 66: see k_rng_calls) and the arithmetic-sequence term advanced by
  // 64 steps at a time instead of two 64-bit modulo operations per draw: an ex_load firing at 1M beads draws ~350k values
  // from ONE serial stream (320 us with 33 draws per step)
  if (lane < 33 && lane < count) {
    uint32_t y = (ring[lane] - ring[lane + 64]) & M24;
    ring[lane + 97] = y;
    out[lane] = (y - c_of_dev(n0 + (unsigned long long)lane)) & M24;
  }
  __syncthreads();
  {
    const int CM = 16777213, STEP64 = (int)((64ull * 7654321ull) % 16777213ull);
    int c = (int)c_of_dev(n0 + 33ull + (unsigned long long)lane);
    for (int base = 33; base < count; base += 64) {
      const int i = base + lane, r = i & 255;
      const uint32_t y = (ring[r] - ring[(r - 33) & 255] + ring[(r + 31) & 255]) & M24;   // ring index of y_i is (i + 97) & 255
      if (i < count) { ring[(r + 97) & 255] = y; out[i] = (y - (uint32_t)c) & M24; }
      c -= STEP64;
      if (c < 0) c += CM;
      __syncthreads();
    }
  }
  uint32_t w0 = 0, w1 = 0;
  w0 = ring[(count + lane) & 255];
  if (lane + 64 < 97) w1 = ring[(count + lane + 64) & 255];
  __syncthreads();
  state[lane] = w0;
  if (lane + 64 < 97) state[lane + 64] = w1;
  if (lane == 0) {
    unsigned long long n1 = n0 + (unsigned long long)count;
    state[97] = (uint32_t)(n1 & 0xFFFFFFFFull);
    state[98] = (uint32_t)(n1 >> 32);
  }
}

void launch_ranmars_gen(DeviceState &d, int slot, const int *count_ptr, uint32_t *out, int maxout) {
  hipLaunchKernelGGL(k_ranmars_gen, dim3(1), dim3(64), 0, d.stream, d.le_rng_state + slot * 100, count_ptr, out,
                     maxout);
}
void le_rng_upload(DeviceState &d, int slot, const RanMarsInt &r) {
  uint32_t st[100] = {0};
  for (int k = 0; k < 97; k++) st[k] = r.w[k];
  st[97] = (uint32_t)(r.n & 0xFFFFFFFFull);
  st[98] = (uint32_t)(r.n >> 32);
  HIP_CHECK(hipMemcpyAsync(d.le_rng_state + slot * 100, st, sizeof st, hipMemcpyHostToDevice, d.stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));
}
void le_rng_download(DeviceState &d, int slot, RanMarsInt &r) {
  uint32_t st[100];
  HIP_CHECK(hipMemcpyAsync(st, d.le_rng_state + slot * 100, sizeof st, hipMemcpyDeviceToHost, d.stream));
  HIP_CHECK(hipStreamSynchronize(d.stream));
  for (int k = 0; k < 97; k++) r.w[k] = st[k];
  r.n = (uint64_t)st[97] | ((uint64_t)st[98] << 32);
}

}  // namespace lmp_le

// test hook: `count` draws of RanMars(seed) after skipping `skip`, generated by the device serial-stream
// kernel in `ncalls` consecutive calls (exercises the state write-back)
extern "C" int lammps_le_test_device_ranmars(int seed, long long skip, int count, int ncalls, double *out) {
  using namespace lmp_le;
  try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return 1;
    DeviceState d;
    HIP_CHECK(hipStreamCreate(&d.stream));
    HIP_CHECK(hipMalloc(&d.le_rng_state, 300 * sizeof(uint32_t)));
    uint32_t *draws; int *cnt;
    HIP_CHECK(hipMalloc(&draws, (size_t)count * sizeof(uint32_t)));
    HIP_CHECK(hipMalloc(&cnt, sizeof(int)));
    RanMarsInt r; r.seed(seed); r.jump((uint64_t)skip);
    le_rng_upload(d, 1, r);
    std::vector<uint32_t> h(count);
    int done = 0;
    for (int c = 0; c < ncalls; c++) {
      int n = (c == ncalls - 1) ? count - done : count / ncalls;
      HIP_CHECK(hipMemcpy(cnt, &n, sizeof(int), hipMemcpyHostToDevice));
      launch_ranmars_gen(d, 1, cnt, draws, count);
      HIP_CHECK(hipMemcpy(h.data() + done, draws, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
      done += n;
    }
    for (int i = 0; i < count; i++) out[i] = h[i] * (1.0 / 16777216.0);
    (void)hipFree(draws); (void)hipFree(cnt); (void)hipFree(d.le_rng_state); (void)hipStreamDestroy(d.stream);
    return 0;
  } catch (...) { return 2; }
}
