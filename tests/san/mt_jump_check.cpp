// The GF(2) polynomial arithmetic behind the MT19937 jump-ahead (irspack_amd/csrc/mt_jump.hpp), compiled on
// its own: products mod the characteristic polynomial against a bit-by-bit reference, the tabulated
// polynomial against Berlekamp-Massey, and jumps against the engine's own sequence.  Built twice by
// tests/test_host_sanitizers.py: with the carry-less multiplier (where the host has one) and with
// -DIRS_MTJUMP_NO_CLMUL (the portable product a host without PCLMULQDQ takes).
#include <cstdio>
#include <cstring>
#include <random>

#include "../../irspack_amd/csrc/mt_jump.hpp"

using namespace irs::mtjump;

#define CHECK(c)                                                  \
  do {                                                            \
    if (!(c)) {                                                   \
      std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
      return 1;                                                   \
    }                                                             \
  } while (0)

static Poly mulmod_reference(const Poly &a, const Poly &b) {
  std::vector<uint64_t> c(2 * PW + 2, 0);
  for (int i = 0; i < DEG; i++)
    if ((a[i >> 6] >> (i & 63)) & 1u)
      for (int j = 0; j < DEG; j++)
        if ((b[j >> 6] >> (j & 63)) & 1u) {
          const int k = i + j;
          c[k >> 6] ^= uint64_t(1) << (k & 63);
        }
  const Poly &phi = phi_low_computed();
  for (int i = 2 * DEG - 2; i >= DEG; i--)
    if ((c[i >> 6] >> (i & 63)) & 1u) {
      c[i >> 6] ^= uint64_t(1) << (i & 63);
      for (int j = 0; j < DEG; j++)
        if ((phi[j >> 6] >> (j & 63)) & 1u) {
          const int k = i - DEG + j;
          c[k >> 6] ^= uint64_t(1) << (k & 63);
        }
    }
  Poly o;
  std::memcpy(o.data(), c.data(), PW * sizeof(uint64_t));
  o[PW - 1] &= (uint64_t(1) << (DEG & 63)) - 1;
  return o;
}

int main() {
  CHECK(phi_low() == phi_low_computed());
  CHECK(phi_terms().size() == 134 && phi_terms().front() == 0 && DEG - phi_terms().back() >= 64);
  std::mt19937_64 g(11);
  for (int rep = 0; rep < 3; rep++) {
    Poly a, b;
    for (auto &w : a) w = g();
    for (auto &w : b) w = g();
    if (rep == 2) {  // sparse operands and the top coefficient
      a.fill(0);
      a[PW - 1] = uint64_t(1) << ((DEG - 1) & 63);
      a[0] = 1;
    }
    a[PW - 1] &= (uint64_t(1) << (DEG & 63)) - 1;
    b[PW - 1] &= (uint64_t(1) << (DEG & 63)) - 1;
    CHECK(mulmod(a, b) == mulmod_reference(a, b));
  }
  // jumps of j blocks of 624 * 2^b words land on the engine's own state
  uint32_t w[N];
  w[0] = 5489u;
  for (uint32_t i = 1; i < N; i++) w[i] = 1812433253u * (w[i - 1] ^ (w[i - 1] >> 30)) + i;
  std::vector<uint32_t> x;
  const size_t blocks_checked = 41;
  raw_sequence(w, N * (blocks_checked + 1), x);
  for (const uint64_t j : {uint64_t(1), uint64_t(2), uint64_t(5), uint64_t(16), uint64_t(37)}) {
    for (const int b : {0, 1, 3}) {
      if ((j << b) > blocks_checked - 1) continue;
      uint32_t v[N];
      std::memcpy(v, w, sizeof(w));
      irs::mtjump::apply(pow_blocks(b, j), v);
      CHECK(std::memcmp(v, x.data() + (j << b) * N, sizeof(v)) == 0);
    }
  }
  std::printf("mt_jump_check ok (%s product)\n", have_clmul() ? "carry-less" : "portable");
  return 0;
}
