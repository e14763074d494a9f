// ORACLE - TEST INFRASTRUCTURE ONLY.  Sanitizer driver of the multi-threaded CPU oracle
// (`make -C oracle SAN=thread` / `SAN=address` builds it against the sanitized sources and
// tests/test_host_sanitizers.py runs it): two iALS epochs per solver (Cholesky, CG, iALS++) and
// compute_loss on 4 threads, on a small random matrix.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct ModelConfig {
  uint64_t K;
  float alpha0, reg, nu, init_stdev;
  int32_t random_seed;
  int32_t loss_type;
};
struct SolverConfig {
  uint64_t n_threads;
  int32_t solver_type;
  uint64_t max_cg_steps, ialspp_subspace_dimension, ialspp_iteration;
};
extern "C" {
void *orc_ials_create(const ModelConfig *, int64_t, int64_t, const int64_t *, const int32_t *, const float *);
void orc_ials_destroy(void *);
int orc_ials_step(void *, const SolverConfig *);
int orc_ials_compute_loss(void *, const SolverConfig *, float *);
float *orc_ials_user_ptr(void *);
}

int main() {
  const int64_t U = 400, I = 300;
  std::mt19937_64 rng(3);
  std::vector<int64_t> indptr(U + 1, 0);
  std::vector<int32_t> indices;
  std::vector<float> data;
  for (int64_t u = 0; u < U; u++) {
    for (int32_t i = 0; i < I; i++)
      if (rng() % 10 == 0) {
        indices.push_back(i);
        data.push_back(1.0f + static_cast<float>(rng() % 3));
      }
    indptr[u + 1] = static_cast<int64_t>(indices.size());
  }
  ModelConfig mc{24, 0.1f, 0.05f, 1.0f, 0.1f, 42, 1};
  for (int solver = 0; solver < 3; solver++) {
    void *t = orc_ials_create(&mc, U, I, indptr.data(), indices.data(), data.data());
    SolverConfig sc{4, solver, 3, 8, 2};
    for (int e = 0; e < 2; e++)
      if (orc_ials_step(t, &sc) != 0) return 1;
    float loss = 0;
    if (orc_ials_compute_loss(t, &sc, &loss) != 0 || !(loss == loss)) return 1;
    orc_ials_destroy(t);
  }
  std::puts("oracle_san ok");
  return 0;
}
