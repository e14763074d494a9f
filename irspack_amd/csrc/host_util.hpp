// Host-only helpers (no HIP dependency): shared by the library and by the sanitizer build of
// the multi-threaded host preparation (tests/san/host_prep_san.cpp).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

namespace irs {

// (the literal overload matters: a std::string parameter would be constructed - and heap
// allocated - on every call, 160 ms for a per-entry check over 20 M entries)
inline void check_arg(bool cond, const char *msg) {
  if (!cond) throw std::invalid_argument(msg);
}
inline void check_arg(bool cond, const std::string &msg) {
  if (!cond) throw std::invalid_argument(msg);
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// an IRSPACK_AMD_* switch: unset = `dflt`, else its integer value != 0
inline bool env_flag(const char *name, bool dflt) {
  const char *e = std::getenv(name);
  return e ? std::atoi(e) != 0 : dflt;
}

// std::vector whose resize() leaves trivially constructible elements uninitialised: the big
// host staging arrays (hundreds of MB) are written once right after they are sized, and the
// value-initialising resize touched every page on one thread first (40 ms per 240 MB).
template <class T> struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
  template <class U, class... A> void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0)
      ::new (static_cast<void *>(p)) U;
    else
      ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
  }
};
template <class T> using RawVector = std::vector<T, NoInitAlloc<T>>;

// body(k) for k in [0, n_thr) on n_thr host threads (k = 0 on the calling one); body must not throw.
// The threads are started as a binary tree - thread k starts 2 k + 1 and 2 k + 2 before its own share -
// so that the last one starts after log2(n_thr) thread creations instead of n_thr (30 us each: half a
// millisecond for 20 threads, as much as a 4 M-entry pass takes).
template <class F> inline void run_on_threads(int n_thr, F &&body) {
  if (n_thr <= 1) {
    body(0);
    return;
  }
  struct Node {
    static void run(int k, int n, F &body) {
      // A thread that cannot be started (std::system_error: the process is out of threads) must not end
      // in std::terminate inside a worker - its subtree runs here, on this thread, instead.
      std::thread left, right;
      bool left_here = false, right_here = false;
      if (2 * k + 1 < n) {
        try {
          left = std::thread([&body, k, n] { run(2 * k + 1, n, body); });
        } catch (const std::system_error &) {
          left_here = true;
        }
      }
      if (2 * k + 2 < n) {
        try {
          right = std::thread([&body, k, n] { run(2 * k + 2, n, body); });
        } catch (const std::system_error &) {
          right_here = true;
        }
      }
      body(k);
      if (left_here) run(2 * k + 1, n, body);
      if (right_here) run(2 * k + 2, n, body);
      if (left.joinable()) left.join();
      if (right.joinable()) right.join();
    }
  };
  Node::run(0, n_thr, body);
}

// fn(begin, end) over [0, n) on up to `max_threads` host threads (one call on this thread when n is
// small); fn must not throw.
template <class F> inline void parallel_ranges(int64_t n, F &&fn, int max_threads = 16, int64_t min_per_thread = 250000) {
  const int64_t hw = static_cast<int64_t>(std::thread::hardware_concurrency());
  const int n_thr = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({max_threads, hw, n / min_per_thread})));
  if (n_thr <= 1) {
    fn(int64_t(0), n);
    return;
  }
  run_on_threads(n_thr, [&](int k) { fn(n * k / n_thr, n * (k + 1) / n_thr); });
}

}  // namespace irs
