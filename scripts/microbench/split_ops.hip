// Microbenchmark (development): issue rate and exactness of the candidates for the bf16x3 split's residual
// step r = x - bf16(x):  (a) v_and/v_lshl + v_sub_f32 (round 5/6),  (b) v_dot2c_f32_bf16 with a {-1, 0} /
// {0, -1} selector (one instruction: r = x + (-1)*h.lo + 0*h.hi),  (c) v_pk_add_f32 on register pairs.
// Build: hipcc --offload-arch=gfx950 -O3 -o split_ops split_ops.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      std::exit(1);                                                                    \
    }                                                                                  \
  } while (0)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// the selectors {-1, 0} / {0, -1} come through scalar registers the compiler cannot see into: written as
// constants, hipcc 7.2 folds 0x0000bf80 to the INLINE constant -1.0, which the hardware supplies as the fp32
// pattern 0xbf800000 = {0, -1} (measured: r0 = x0 + h.hi)
__device__ __forceinline__ unsigned opaque(unsigned v) {
  unsigned r;
  asm volatile("s_mov_b32 %0, %1" : "=s"(r) : "i"(v));
  return r;
}
#define SEL_LO opaque(0x0000bf80u)
#define SEL_HI opaque(0xbf800000u)
__device__ __forceinline__ float sub_lo(float x, unsigned h, unsigned sel) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, h), __builtin_bit_cast(bf16x2, sel), x, false);
}
#define sub_hi sub_lo

// exactness: out[3 i ..] = (hi, mid, lo) packed pairs by both forms
__global__ void split_check(const float *x, unsigned *a, unsigned *b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  {
    const unsigned h = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    const unsigned mi = pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(mi << 16), s1 = r1 - __uint_as_float(mi & 0xffff0000u);
    a[3 * i] = h; a[3 * i + 1] = mi; a[3 * i + 2] = pack_bf16(s0, s1);
  }
  {
    const unsigned h = pack_bf16(x0, x1);
    const unsigned sl = SEL_LO, sh = SEL_HI;
    const float r0 = sub_lo(x0, h, sl), r1 = sub_hi(x1, h, sh);
    const unsigned mi = pack_bf16(r0, r1);
    const float s0 = sub_lo(r0, mi, sl), s1 = sub_hi(r1, mi, sh);
    b[3 * i] = h; b[3 * i + 1] = mi; b[3 * i + 2] = pack_bf16(s0, s1);
  }
}

template <int FORM>
__global__ __launch_bounds__(256) void rate_kernel(const float *x, float *out, int iters) {
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = x[threadIdx.x + 64 * k];
  unsigned acc = 0;
  const unsigned sl = SEL_LO, sh = SEL_HI;
  (void)sl; (void)sh;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      if (FORM == 0) {
        const unsigned h = pack_bf16(v[k], v[k + 1]);
        const float r0 = v[k] - __uint_as_float(h << 16), r1 = v[k + 1] - __uint_as_float(h & 0xffff0000u);
        const unsigned mi = pack_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(mi << 16), s1 = r1 - __uint_as_float(mi & 0xffff0000u);
        acc += h ^ mi ^ pack_bf16(s0, s1);
        v[k] += s0; v[k + 1] += s1;
      } else if (FORM == 1) {
        const unsigned h = pack_bf16(v[k], v[k + 1]);
        const float r0 = sub_lo(v[k], h, sl), r1 = sub_hi(v[k + 1], h, sh);
        const unsigned mi = pack_bf16(r0, r1);
        const float s0 = sub_lo(r0, mi, sl), s1 = sub_hi(r1, mi, sh);
        acc += h ^ mi ^ pack_bf16(s0, s1);
        v[k] += s0; v[k + 1] += s1;
      } else {
        f32x2 p = {v[k], v[k + 1]};
        f32x2 q = {v[(k + 2) & 15], v[(k + 3) & 15]};
#pragma unroll
        for (int r = 0; r < 6; r++) { f32x2 t; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(t) : "v"(p), "v"(q)); p = t; }
        v[k] = p.x; v[k + 1] = p.y;
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; k++) s += v[k];
  if (s == 123.456f || acc == 0x12345u) out[0] = s + acc;
}

int main() {
  const int n = 1 << 22;
  std::vector<float> hx(2 * n);
  std::mt19937_64 rng(7);
  std::normal_distribution<float> nd(0.f, 0.1f);
  for (int i = 0; i < 2 * n; i++) hx[i] = nd(rng);
  // special values: tiny, denormal, huge, exact bf16, zero, negative zero
  const float sp[] = {0.f, -0.f, 1.f, -1.f, 1e-38f, 1e-40f, -3e-39f, 1e38f, 3.3e38f, 1.17549435e-38f, 0.1f, 1.0f + 1.19e-7f,
                      1e-30f, -1e-30f, 255.99998f, 1.9999999f};
  for (size_t i = 0; i < sizeof sp / sizeof sp[0]; i++) hx[i] = sp[i];
  for (int i = 64; i < 4096; i++) hx[i] = std::ldexp(nd(rng), -100 - (i % 50));  // results near / below the denormal range
  float *dx; unsigned *da, *db; float *dout;
  CK(hipMalloc(&dx, sizeof(float) * 2 * n)); CK(hipMalloc(&da, 12u * n)); CK(hipMalloc(&db, 12u * n)); CK(hipMalloc(&dout, 64));
  CK(hipMemcpy(dx, hx.data(), sizeof(float) * 2 * n, hipMemcpyHostToDevice));
  split_check<<<n / 256, 256>>>(dx, da, db, n);
  std::vector<unsigned> ha(3 * n), hb(3 * n);
  CK(hipMemcpy(ha.data(), da, 12u * n, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), db, 12u * n, hipMemcpyDeviceToHost));
  long diff = 0, first = -1;
  for (long i = 0; i < 3L * n; i++) if (ha[i] != hb[i]) { if (first < 0) first = i; diff++; }
  std::printf("split words differing: %ld of %ld", diff, 3L * n);
  if (first >= 0) std::printf("  first at pair %ld word %ld: x = %g %g  and/sub %08x  dot2 %08x", first / 3, first % 3, hx[2 * (first / 3)], hx[2 * (first / 3) + 1], ha[first], hb[first]);
  std::printf("\n");
  // does the sum of the three pieces reproduce x (where the pieces are normal)?
  long inexact = 0;
  for (long i = 0; i < n; i++) for (int s = 0; s < 2; s++) {
    auto part = [&](unsigned w) { unsigned u = s ? (w & 0xffff0000u) : (w << 16); float f; std::memcpy(&f, &u, 4); return f; };
    const double sum = double(part(hb[3 * i])) + double(part(hb[3 * i + 1])) + double(part(hb[3 * i + 2]));
    if (sum != double(hx[2 * i + s]) && std::fabs(hx[2 * i + s]) > 1e-30f) inexact++;
  }
  std::printf("dot2 form: values (|x| > 1e-30) not reproduced exactly by hi + mid + lo: %ld\n", inexact);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4096, blocks = 256 * 8;
  auto run = [&](auto kern, const char *name, double ops_per_iter) {
    kern<<<blocks, 256>>>(dx, dout, 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<<<blocks, 256>>>(dx, dout, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // waves per SIMD: blocks * 4 waves / (256 CUs * 4 SIMDs)
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double cycles = ms * 1e-3 * 2.4e9;
    std::printf("%-28s %8.3f ms   %.2f cycles per pair-split (or per 6 pk_adds) per wave at 2.4 GHz\n", name, ms,
                cycles / (iters * 8.0 * waves_per_simd));
    (void)ops_per_iter;
  };
  run(rate_kernel<0>, "and/shift + sub (13 VALU)", 0);
  run(rate_kernel<1>, "dot2c (9 VALU)", 0);
  run(rate_kernel<2>, "6 x v_pk_add_f32", 0);
  return 0;
}
