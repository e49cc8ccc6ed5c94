// ranmars.cpp — RanMars in exact integer arithmetic + polynomial jump-ahead.
//
// src/random_mars.cpp:29-95 of the reference: uniform() = (y_n - c_n) mod 1 with
//   y_n = y_{n-97} - y_{n-33} (mod 1),  y_{-k} = u[k] (seed table, k = 1..97)
//   c_n = c_{n-1} - 7654321/2^24 (mod 16777213/2^24), c_{-1} = 362436/2^24
// and every quantity an integer multiple of 2^-24, so 24-bit integer arithmetic is bit-exact.
// The lagged-Fibonacci part is linear over Z/2^24: E^97 = 1 - E^64 for the shift operator E, so
// y_{n+k} = sum_j a_j y_{n+j} with a(x) = x^k mod (x^97 + x^64 - 1).  That is what lets the device
// generate the 3N Langevin draws of a step in parallel blocks (kernels_rng.hip).
#include <cstring>

#include "engine.h"

namespace lmp_le {

static constexpr uint32_t M24 = 0xFFFFFFu;
static constexpr uint64_t CM = 16777213ull, CD = 7654321ull, C0 = 362436ull;

void RanMarsInt::seed(int seed) {
  if (seed <= 0 || seed > 900000000) throw LammpsError("Invalid seed for Marsaglia random # generator");
  int ij = (seed - 1) / 30082;
  int kl = (seed - 1) - 30082 * ij;
  int i = (ij / 177) % 177 + 2;
  int j = ij % 177 + 2;
  int k = (kl / 169) % 178 + 1;
  int l = kl % 169;
  uint32_t u[98];
  for (int ii = 1; ii <= 97; ii++) {
    uint32_t s = 0;
    for (int jj = 1; jj <= 24; jj++) {
      int m = ((i * j) % 179) * k % 179;
      i = j; j = k; k = m;
      l = (53 * l + 1) % 169;
      if ((l * m) % 64 >= 32) s |= 1u << (24 - jj);
    }
    u[ii] = s;
  }
  for (int q = 0; q < 97; q++) w[q] = u[97 - q];   // w[q] = y_{n-97+q}, y_{-m} = u[m]
  n = 0;
  next_raw();   // the constructor's warm-up uniform()
}

uint32_t RanMarsInt::c_of(uint64_t n) {
  uint64_t dec = (((n + 1) % CM) * CD) % CM;
  return (uint32_t)((C0 + CM - dec) % CM);
}

uint32_t RanMarsInt::next_raw() {
  uint32_t y = (w[0] - w[64]) & M24;
  std::memmove(w, w + 1, 96 * sizeof(uint32_t));
  w[96] = y;
  uint32_t out = (y - c_of(n)) & M24;
  n++;
  return out;
}

static void poly_mulmod(const uint32_t *a, const uint32_t *b, uint32_t *out) {
  uint32_t c[193];
  std::memset(c, 0, sizeof c);
  for (int i = 0; i < 97; i++) {
    if (!a[i]) continue;
    for (int j = 0; j < 97; j++) c[i + j] += a[i] * b[j];
  }
  for (int d = 192; d >= 97; d--) {   // x^97 = 1 - x^64
    uint32_t v = c[d];
    if (!v) continue;
    c[d - 97] += v;
    c[d - 33] -= v;
    c[d] = 0;
  }
  for (int i = 0; i < 97; i++) out[i] = c[i] & M24;
}

void ranmars_jump_poly(uint64_t k, uint32_t a[97]) {
  uint32_t res[97], base[97], tmp[97];
  std::memset(res, 0, sizeof res);
  std::memset(base, 0, sizeof base);
  res[0] = 1;
  base[1] = 1;
  while (k) {
    if (k & 1) { poly_mulmod(res, base, tmp); std::memcpy(res, tmp, sizeof res); }
    k >>= 1;
    if (k) { poly_mulmod(base, base, tmp); std::memcpy(base, tmp, sizeof base); }
  }
  std::memcpy(a, res, sizeof res);
}

void RanMarsInt::jump(uint64_t k) {
  if (k == 0) return;
  if (k < 4096) { for (uint64_t i = 0; i < k; i++) next_raw(); return; }
  uint32_t a[97], y[193];
  ranmars_jump_poly(k, a);
  std::memcpy(y, w, sizeof w);
  for (int i = 97; i < 193; i++) y[i] = (y[i - 97] - y[i - 33]) & M24;
  for (int i = 0; i < 97; i++) {
    uint32_t acc = 0;
    for (int j = 0; j < 97; j++) acc += a[j] * y[i + j];
    w[i] = acc & M24;
  }
  n += k;
}

}  // namespace lmp_le
