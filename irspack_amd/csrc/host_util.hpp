// Host-only helpers (no HIP dependency): shared by the library and by the sanitizer build of
// the multi-threaded host preparation (tests/san/host_prep_san.cpp).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
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

}  // namespace irs
