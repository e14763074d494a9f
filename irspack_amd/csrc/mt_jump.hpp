// Jump-ahead for MT19937 (no HIP dependency): the state the engine has after J more words, without
// generating them - what lets every host thread produce ITS block of std::mt19937's stream
// (host_prep.hpp: draw_factor; Solver::initialize, IALSTrainer.hpp:64-76, draws 1.3 G variates from one
// sequential engine at the 10 M x 1 M shape).
//
// The engine is linear over GF(2): one step F maps the 19937-bit state (x_k's top bit, x_{k+1} ..
// x_{k+623}) to the next; F^J = g_J(F) with g_J(t) = t^J mod phi(t), phi the characteristic polynomial
// of F (degree 19937).  (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead
// for F2-linear random number generators", 2008.)
//   phi      Berlekamp-Massey on 2 x 19937 bits of the engine's own output (one fixed bit of
//            successive state words): its connection polynomial reversed; computed once per process
//   t^J      J = 624 2^b: t^624 needs no reduction, then b squarings mod phi; a thread's offset k J is a
//            product of the cached powers g^(2^i)
//   apply    Horner: h <- F(h) + g_i state, i = 19936 .. 0 (19937 single-word steps on a sliding
//            window, ~10 k state additions)
// Polynomials: 312 64-bit words, bit i of word i / 64 = coefficient of t^i.  Multiplication: for every
// set bit of a, one of 64 pre-shifted copies of b is added at the bit's word offset; reduction: from
// the top bit down, a pre-shifted copy of phi is added - both plain XOR runs the compiler vectorises.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

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

// phi(t) without its leading term t^19937 (bits 0 .. 19936)
inline const Poly &phi_low() {
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
  static const Shifted phis(phi_low(), true);  // phi itself (with its leading term), shifted
  const Shifted bs(b, false);
  std::vector<uint64_t> c(2 * PW + 2, 0);
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
