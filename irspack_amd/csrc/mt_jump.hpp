// Jump-ahead for MT19937 (no HIP dependency): the state the engine has after J more words, without
// generating them - what lets every host thread produce ITS block of std::mt19937's stream
// (host_prep.hpp: draw_factor; Solver::initialize, IALSTrainer.hpp:64-76, draws 1.3 G variates from one
// sequential engine at the 10 M x 1 M shape).
//
// The engine is linear over GF(2): one step F maps the 19937-bit state (x_k's top bit, x_{k+1} ..
// x_{k+623}) to the next; F^J = g_J(F) with g_J(t) = t^J mod phi(t), phi the characteristic polynomial
// of F (degree 19937).  (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead
// for F2-linear random number generators", 2008.)
//   phi      134 terms below t^19937, tabulated (phi_terms); phi_low_computed() derives it: Berlekamp-
//            Massey on 2 x 19937 bits of the engine's own output (one fixed bit of successive state
//            words), the connection polynomial reversed - the sanitizer harness compares the two
//   t^J      J = 624 2^b: t^624 needs no reduction, then b squarings mod phi; a thread's offset k J is a
//            product of the cached powers g^(2^i)
//   apply    Horner: h <- F(h) + g_i state, i = 19936 .. 0 (19937 single-word steps on a sliding
//            window, ~10 k state additions)
// Polynomials: 312 64-bit words, bit i of word i / 64 = coefficient of t^i.  Multiplication: 312 x 312
// carry-less 64-bit products (PCLMULQDQ, 0.1 ms) where the host has them, else for every set bit of a
// one of 64 pre-shifted copies of b is added at the bit's word offset (1.4 ms); reduction: phi is
// sparse, so a whole word of high coefficients is folded onto 134 lower places at once (0.1 ms).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>
#if defined(__x86_64__) && !defined(IRS_MTJUMP_NO_CLMUL)
#include <immintrin.h>
#endif

namespace irs {
namespace mtjump {

constexpr int N = 624, M = 397, DEG = 19937, PW = (DEG + 63) / 64;  // 312 words per polynomial
using Poly = std::array<uint64_t, PW>;

inline uint32_t mix(uint32_t hi, uint32_t lo, uint32_t far) {
  const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
  return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// the raw state words x_0, x_1, ... of an engine whose window is `w` (x_0 .. x_623)
inline void raw_sequence(const uint32_t *w, size_t count, std::vector<uint32_t> &x) {
  x.assign(w, w + N);
  x.resize(std::max<size_t>(count, N));
  for (size_t k = N; k < count; k++) x[k] = mix(x[k - N], x[k - N + 1], x[k - N + M]);
}

// phi(t) without its leading term t^19937 (bits 0 .. 19936), COMPUTED (17 ms): what the table below was
// printed from and what tests/san/host_prep_san.cpp checks it against
inline const Poly &phi_low_computed() {
  static Poly phi;
  static std::once_flag once;
  std::call_once(once, [] {
    // any non-degenerate window will do: the characteristic polynomial is the engine's, not the seed's
    uint32_t w[N];
    w[0] = 19650218u;
    for (uint32_t i = 1; i < N; i++) w[i] = 1812433253u * (w[i - 1] ^ (w[i - 1] >> 30)) + i;
    const size_t LEN = 2 * DEG + 64;
    std::vector<uint32_t> x;
    raw_sequence(w, LEN + N, x);
    // s_n = bit 0 of x_{n + 624} (a word that depends on the whole state); Berlekamp-Massey over GF(2),
    // polynomials packed 64 coefficients per word: C(D) = 1 + c_1 D + ..., s_n = sum_i c_i s_{n - i}
    const size_t WORDS = (LEN + 63) / 64 + 1;
    std::vector<uint64_t> s(WORDS, 0), C(WORDS, 0), B(WORDS, 0), T(WORDS, 0);
    for (size_t n = 0; n < LEN; n++)
      if (x[n + N] & 1u) s[n >> 6] |= uint64_t(1) << (n & 63);
    // the window s_{n-L} .. s_n is read REVERSED against c_0 .. c_L; keep a reversed copy of s so that the
    // discrepancy is a popcount of an AND of two forward runs: r[j] = s_{LEN - 1 - j}
    std::vector<uint64_t> r(WORDS + 1, 0);
    for (size_t n = 0; n < LEN; n++)
      if ((s[n >> 6] >> (n & 63)) & 1u) {
        const size_t j = LEN - 1 - n;
        r[j >> 6] |= uint64_t(1) << (j & 63);
      }
    C[0] = B[0] = 1;
    size_t L = 0, m = 1;
    for (size_t n = 0; n < LEN; n++) {
      // d = sum_{i = 0..L} c_i s_{n - i} = sum_i c_i r[(LEN - 1 - n) + i]
      const size_t off = LEN - 1 - n, ow = off >> 6, ob = off & 63;
      uint64_t acc = 0;
      const size_t lw = (L >> 6) + 1;
      for (size_t q = 0; q < lw; q++) {
        const uint64_t lo = r[ow + q] >> ob;
        const uint64_t hi = ob ? (r[ow + q + 1] << (64 - ob)) : 0;
        acc ^= C[q] & (lo | hi);
      }
      const bool d = __builtin_parityll(acc);
      if (!d) {
        m++;
        continue;
      }
      const bool grow = 2 * L <= n;
      if (grow) T = C;
      // C <- C + D^m B
      const size_t sw = m >> 6, sb = m & 63;
      for (size_t q = 0; q + sw < WORDS; q++) {
        const uint64_t v = B[q];
        if (!v) continue;
        C[q + sw] ^= v << sb;
        if (sb && q + sw + 1 < WORDS) C[q + sw + 1] ^= v >> (64 - sb);
      }
      if (grow) {
        L = n + 1 - L;
        B = T;
        m = 1;
      } else {
        m++;
      }
    }
    // L must be 19937; phi(t) = t^L + c_1 t^{L-1} + ... + c_L: coefficient of t^j is c_{L - j}
    phi.fill(0);
    if (L == static_cast<size_t>(DEG)) {
      for (int j = 0; j < DEG; j++) {
        const size_t i = DEG - j;
        if ((C[i >> 6] >> (i & 63)) & 1u) phi[j >> 6] |= uint64_t(1) << (j & 63);
      }
    }
  });
  return phi;
}

// The exponents of phi's 134 terms below t^19937, ascending: a constant of the engine (the characteristic
// polynomial of MT19937's state transition), tabulated so that the first trainer of a process does not
// spend 17 ms on Berlekamp-Massey; `phi_low_computed()` reproduces it (checked in the sanitizer harness).
inline const std::vector<int> &phi_terms() {
  static const std::vector<int> terms = {
      0, 1189, 1416, 1585, 1643, 1870, 2493, 2773, 3000, 3227, 3454, 3681, 3908, 4135, 4362, 4753,
      5661, 6337, 6569, 7129, 7477, 7525, 7583, 7752, 7979, 8206, 9505, 9901, 9969, 10128, 10693, 10761,
      10920, 11089, 11147, 11157, 11215, 11321, 11374, 11384, 11485, 11611, 11712, 11717, 11838, 11881, 11944, 11997,
      12277, 12335, 12393, 12504, 12509, 12620, 12673, 12731, 12736, 12789, 12905, 12958, 12963, 13137, 13185, 13190,
      13243, 13301, 13412, 13528, 13533, 13639, 13697, 13760, 13813, 13866, 14093, 14151, 14209, 14320, 14325, 14436,
      14547, 14552, 14605, 14721, 14774, 14779, 14953, 15001, 15006, 15059, 15117, 15228, 15344, 15349, 15455, 15513,
      15576, 15629, 15682, 15909, 15967, 16025, 16136, 16141, 16252, 16363, 16368, 16421, 16537, 16590, 16595, 16817,
      16822, 16875, 16933, 17044, 17160, 17271, 17329, 17445, 17498, 17725, 17783, 17841, 17952, 18068, 18179, 18237,
      18406, 18633, 18691, 18860, 19087, 19314};
  return terms;
}

inline const Poly &phi_low() {
  static const Poly phi = [] {
    Poly p;
    p.fill(0);
    for (int j : phi_terms()) p[j >> 6] |= uint64_t(1) << (j & 63);
    return p;
  }();
  return phi;
}

// c (bits 0 .. 2 DEG - 2) mod phi, in place, a word of high coefficients at a time: t^i = sum_k t^(i - DEG + e_k)
// for every term e_k of phi.  A word's 64 coefficients are folded together - the targets of word w lie
// below it as long as DEG - e_max >= 64, which is checked - so the reduction is 312 words x 135 terms
// shifted XORs instead of one 313-word XOR per set coefficient (10 k of them: 1 ms per product).
// Returns false when phi does not allow it (never for MT19937); the caller then reduces bit by bit.
inline bool reduce_sparse(uint64_t *c, size_t words) {
  const std::vector<int> &e = phi_terms();
  if (e.empty() || DEG - e.back() < 64) return false;
  const int top = static_cast<int>(words) - 1;
  for (int w = top; w >= (DEG >> 6); w--) {
    uint64_t v = c[w];
    if (w == (DEG >> 6)) v &= ~((uint64_t(1) << (DEG & 63)) - 1);  // (only the coefficients >= DEG of this word)
    if (!v) continue;
    c[w] ^= v;
    const int base = 64 * w - DEG;  // coefficient 64 w + q goes to base + q + e_k
    for (int ek : e) {
      const int sft = base + ek;  // >= 0: w >= DEG / 64 and the masked bits are >= DEG
      // (for the partial word `base` is negative by up to DEG & 63 bits, but v's low bits are zero: shift v down)
      if (sft >= 0) {
        const int q = sft >> 6, r = sft & 63;
        c[q] ^= v << r;
        if (r) c[q + 1] ^= v >> (64 - r);
      } else {
        c[0] ^= v >> (-sft);  // (-sft < 64; the bits shifted out are the zeros below DEG)
      }
    }
  }
  return true;
}

#if defined(__x86_64__) && !defined(IRS_MTJUMP_NO_CLMUL)
// (IRS_MTJUMP_NO_CLMUL: the portable product only - tests/san/mt_jump_check.cpp builds both)
// the 39,873-bit product with the carry-less multiplier (312 x 312 64-bit products, 0.1 ms) where the
// host has one; the shifted-copy loop below (one 313-word XOR per set coefficient of a: 1.4 ms) elsewhere
__attribute__((target("pclmul,sse2"))) inline void product_clmul(const Poly &a, const Poly &b, uint64_t *c) {
  for (int i = 0; i < PW; i++) {
    if (!a[i]) continue;
    const __m128i ai = _mm_set_epi64x(0, static_cast<long long>(a[i]));
    uint64_t carry = 0;  // (the high half of the product before: every word of c is touched once per i)
    for (int j = 0; j < PW; j++) {
      const __m128i p = _mm_clmulepi64_si128(ai, _mm_set_epi64x(0, static_cast<long long>(b[j])), 0x00);
      c[i + j] ^= static_cast<uint64_t>(_mm_cvtsi128_si64(p)) ^ carry;
      carry = static_cast<uint64_t>(_mm_cvtsi128_si64(_mm_unpackhi_epi64(p, p)));
    }
    c[i + PW] ^= carry;
  }
}
inline bool have_clmul() {
  static const bool has = __builtin_cpu_supports("pclmul") != 0;
  return has;
}
#else
inline bool have_clmul() { return false; }
#endif

// a * b mod phi
inline Poly mulmod(const Poly &a, const Poly &b) {
  // 64 shifted copies of an operand: sh[k][q] = word q of (p << k), PW + 1 words
  struct Shifted {
    std::vector<uint64_t> v;  // 64 x (PW + 1)
    explicit Shifted(const Poly &p, bool top_bit) : v(64 * (PW + 1), 0) {
      for (int k = 0; k < 64; k++) {
        uint64_t *d = v.data() + k * (PW + 1);
        for (int q = 0; q < PW; q++) {
          d[q] ^= p[q] << k;
          if (k) d[q + 1] ^= p[q] >> (64 - k);
        }
        if (top_bit) {  // + t^19937 << k
          const int bit = DEG + k;
          d[bit >> 6] ^= uint64_t(1) << (bit & 63);
        }
      }
    }
    const uint64_t *at(int k) const { return v.data() + k * (PW + 1); }
  };
  std::vector<uint64_t> c(2 * PW + 2, 0);
#if defined(__x86_64__) && !defined(IRS_MTJUMP_NO_CLMUL)
  if (have_clmul()) {
    product_clmul(a, b, c.data());
  } else
#endif
  {
    const Shifted bs(b, false);
    for (int q = 0; q < PW; q++) {
      uint64_t w = a[q];
      while (w) {
        const int k = __builtin_ctzll(w);
        w &= w - 1;
        const uint64_t *src = bs.at(k);
        uint64_t *dst = c.data() + q;
        for (int j = 0; j <= PW; j++) dst[j] ^= src[j];
      }
    }
  }
  static const Shifted phis(phi_low(), true);  // phi itself (with its leading term), shifted: the fallback reduction
  if (!reduce_sparse(c.data(), c.size()))
    for (int i = 2 * DEG - 2; i >= DEG; i--) {
      if (!((c[i >> 6] >> (i & 63)) & 1u)) continue;
      const int s = i - DEG;  // c += phi << s
      const uint64_t *src = phis.at(s & 63);
      uint64_t *dst = c.data() + (s >> 6);
      for (int j = 0; j <= PW; j++) dst[j] ^= src[j];
    }
  Poly out;
  std::memcpy(out.data(), c.data(), PW * sizeof(uint64_t));
  out[PW - 1] &= (uint64_t(1) << (DEG & 63)) - 1;  // (bits >= 19937 are zero by now)
  return out;
}

// t^(624 * 2^b) mod phi for b = 0 .. 40, built on demand (thread-safe)
inline Poly pow_block(int b) {
  static std::vector<Poly> table;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (table.empty()) {
    Poly p;
    p.fill(0);
    p[N >> 6] = uint64_t(1) << (N & 63);  // t^624
    table.push_back(p);
  }
  while (static_cast<int>(table.size()) <= b) table.push_back(mulmod(table.back(), table.back()));
  return table[b];
}

// the window after `blocks` blocks of 624 * 2^b words: g = (t^(624 2^b))^blocks
inline Poly pow_blocks(int b, uint64_t blocks) {
  Poly acc;
  acc.fill(0);
  acc[0] = 1;
  bool one = true;
  for (int i = 0; blocks; i++, blocks >>= 1) {
    if (!(blocks & 1)) continue;
    const Poly f = pow_block(b + i);
    acc = one ? f : mulmod(acc, f);
    one = false;
  }
  return acc;
}

// window <- g(F) window
inline void apply(const Poly &g, uint32_t *window) {
  std::vector<uint32_t> buf(N + DEG + 1, 0);
  uint32_t *h = buf.data();  // the sliding window h[0 .. 623]
  for (int i = DEG - 1; i >= 0; i--) {
    // h <- F(h): the word that follows the window, then the window moves on by one
    h[N] = mix(h[0], h[1], h[M]);
    h++;
    if ((g[i >> 6] >> (i & 63)) & 1u)
      for (int j = 0; j < N; j++) h[j] ^= window[j];
  }
  std::memcpy(window, h, N * sizeof(uint32_t));
}

}  // namespace mtjump
}  // namespace irs
