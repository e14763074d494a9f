// Host side of the iALS path: trainer object, task lists, launches, C ABI.
// Mirrors irspack::ials::IALSTrainer (/root/reference/cpp_source/als/
// IALSTrainer.hpp:709-984) behind include/irspack_amd.h.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <random>
#include <sstream>
#include <future>
#include <thread>

#include "common.hpp"
#include "comm.hpp"
#include "host_prep.hpp"

#include <hip/hip_ext.h>
#include "ials_kernels.hpp"
#include "ials_chol16.hpp"
#include "ials_wg_kernels.hpp"
#include "ials_wg16_kernels.hpp"
#include "ials_pp_kernels.hpp"
#include "ials_feature_kernels.hpp"
#include "ials_short_kernels.hpp"
#include "ials_gk_kernels.hpp"
#include "ials_eig_kernels.hpp"
#include "ials_mf_kernels.hpp"

namespace irs {

std::string &last_error() {
  static thread_local std::string e;
  return e;
}

void require_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw std::runtime_error(
        "irspack_amd: no HIP device is visible; the gfx950 kernels cannot run "
        "(there is no CPU fallback).");
  if (device < 0 || device >= n)
    throw std::invalid_argument("irspack_amd: device index out of range.");
  IRS_HIP(hipSetDevice(device));
}

namespace ials {

static int padded_k(int64_t K) {
  if (K <= 16) return 16;
  if (K <= 32) return 32;
  if (K <= 64) return 64;
  return static_cast<int>(ceil_div(K, 64) * 64);
}

static int chunk_size() {
  static int v = [] {
    const char *e = std::getenv("IRSPACK_AMD_IALS_CHUNK");
    int c = e ? std::atoi(e) : 1024;
    if (c < 64) c = 64;
    return (c + 3) & ~3;
  }();
  return v;
}

static int max_chunks() { return 256; }

// One CSR orientation resident on the device, with its longest-first task list.
struct Side {
  int64_t n_rows = 0, n_other = 0, row_begin = 0, row_end = 0, nnz = 0;
  DeviceBuffer<int32_t> indptr, indices;
  DeviceBuffer<float> data, reg;
  DeviceBuffer<Task> tasks;
  DeviceBuffer<SplitRow> split;
  DeviceBuffer<FoldGroup> fold;  // groups of partial Gramians summed before the split rows are finished
  int32_t n_fold = 0;
  DeviceBuffer<int32_t> rows_by_len;  // rows [row_begin, row_end) longest first (iALS++ launch order)
  int32_t n_tasks = 0, n_split = 0, n_slots = 0, n_long = 0;
  // tasks at the end of the (longest-first) list with <= SHORT_MAX (n_short) / <= 16 entries
  int32_t n_short = 0, n_short16 = 0, n_short8 = 0;
  int64_t long_entries = 0;  // stored entries of the tasks [0, n_tasks - n_short)
  // matrix-free CG at 128 < K <= 256 (ials_mf_kernels.hpp): rows_by_len is cut into the
  // level-synchronous rows (more than MF_NCAP entries, chunked) and the resident classes of
  // MF_CAPS; mf_class[c] = first index of class c in rows_by_len, [MF_CLASSES] = end
  DeviceBuffer<MfLongRow> mf_lrows;
  DeviceBuffer<MfChunk> mf_chunks;
  int32_t mf_n_lrows = 0, mf_n_chunks = 0, mf_class[MF_CLASSES + 1] = {};
  bool unit = false;  // every stored confidence is exactly 1 (UNIT kernels)
  bool positive = false;  // every stored confidence is > 0 (eigenbasis short-row kernels)
  float reg_min = 0.f;    // smallest per-row regulariser of the rows [row_begin, row_end)
  bool entries_ready = false, indptr_ready = false;  // the device arrays were filled ahead of build()

  // The gather pipeline loads whole 64-entry blocks up to two blocks past a row's end: 320 zero entries
  // behind the arrays (set on the device; no padded host copies).
  void alloc_entries(size_t ne, hipStream_t s) {
    indices.alloc(ne + 320);
    data.alloc(ne + 320);
    IRS_HIP(hipMemsetAsync(indices.ptr + ne, 0, 320 * sizeof(int32_t), s));
    IRS_HIP(hipMemsetAsync(data.ptr + ne, 0, 320 * sizeof(float), s));
    entries_ready = true;
  }
  void upload_entries(const HostCsr &m, hipStream_t s) {
    const size_t ne = static_cast<size_t>(m.indptr[m.rows]);
    alloc_entries(ne, s);
    if (ne) {
      IRS_HIP(hipMemcpyAsync(indices.ptr, m.idx(), ne * sizeof(int32_t), hipMemcpyHostToDevice, s));
      if (m.data.size() == ne) {
        IRS_HIP(hipMemcpyAsync(data.ptr, m.data.data(), ne * sizeof(float), hipMemcpyHostToDevice, s));
      } else {  // all ones (host_csr carried no values): written on the device, 80 MB less over PCIe per side
        check_arg(m.flags_known && m.unit, "internal: a value array is missing.");
        IRS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(data.ptr), 0x3f800000, ne, s));
      }
    }
    IRS_HIP(hipStreamSynchronize(s));
  }
  void upload_indptr(const HostCsr &m, hipStream_t s) {
    std::vector<int32_t> ip32(m.rows + 1);
    for (int64_t r = 0; r <= m.rows; r++) ip32[r] = static_cast<int32_t>(m.indptr[r]);
    indptr.upload(ip32, s);
    IRS_HIP(hipStreamSynchronize(s));
    indptr_ready = true;
  }

  // `share`: a Side already built from the same matrix - its device copies of the CSR arrays and
  // of the regulariser are borrowed (row chunks of a shard, irs_ials_sharded_step)
  void build(const HostCsr &m, int64_t rb, int64_t re, const irs_ials_model_config &cfg,
             hipStream_t s, const Side *share = nullptr) {
    n_rows = m.rows;
    n_other = m.cols;
    row_begin = rb;
    row_end = re;
    nnz = m.indptr[m.rows];
    if (share) {
      unit = share->unit;
      positive = share->positive;
    } else if (m.flags_known) {  // (host_csr classified the values on its validation pass)
      unit = m.unit;
      positive = m.unit || m.positive;
    } else {
      unit = std::all_of(m.data.begin(), m.data.end(), [](float v) { return v == 1.0f; });
      positive = unit || std::all_of(m.data.begin(), m.data.end(), [](float v) { return v > 0.0f; });
    }
    std::vector<int32_t> ip32(m.rows + 1);
    std::vector<float> regs(m.rows);
    ip32[m.rows] = static_cast<int32_t>(m.indptr[m.rows]);
    parallel_ranges(m.rows, [&](int64_t r0, int64_t r1) {  // (10 M powf at the 10 M x 1 M shape)
      for (int64_t r = r0; r < r1; r++) {
        ip32[r] = static_cast<int32_t>(m.indptr[r]);
        // Solver::compute_reg, hpp:117-120, evaluated in float like the reference
        const int64_t nz = m.indptr[r + 1] - m.indptr[r];
        regs[r] = cfg.reg * std::pow(cfg.alpha0 * m.cols + nz, cfg.nu);
      }
    });
    reg_min = 0.f;
    for (int64_t r = rb; r < re; r++) reg_min = r == rb ? regs[r] : std::min(reg_min, regs[r]);
    const int CH = chunk_size();
    // (read per build, not once per process: the tests lower it to fold rows of 10^4 entries)
    const char *long_env = std::getenv("IRSPACK_AMD_IALS_CHUNK_LONG");
    const int32_t chunk_long = std::max(CH, long_env ? std::atoi(long_env) : 16384);
    std::vector<Task> tk;
    std::vector<SplitRow> sp;
    std::vector<FoldGroup> fg;
    tk.reserve(re - rb);
    int32_t slots = 0;
    for (int64_t r = rb; r < re; r++) {
      const int32_t b = ip32[r], e = ip32[r + 1], nz = e - b;
      if (nz > CH) {
        // a bounded number of chunks per row: the second kernel adds a row's partials one after
        // the other, and a chunk is one wave's serial work (the longest chunk is the floor of
        // the half-step: 4.4 M entries in 32 chunks were 16 ms of one wave at K = 128)
        // (up to 32 chunks of any length, more only to keep a chunk below 16 K entries: the
        // ML-20M shape's longest row - 116 k entries - stays at 32 chunks and needs no fold pass)
        const int32_t nch = std::min<int32_t>(
            (nz + CH - 1) / CH, std::max<int32_t>(32, std::min<int32_t>(max_chunks(), (nz + chunk_long - 1) / chunk_long)));
        const int32_t per = (((nz + nch - 1) / nch) + 3) & ~3;
        SplitRow sr{static_cast<int32_t>(r), slots, 0, nz, 1};
        for (int32_t c = b; c < e; c += per) {
          tk.push_back(Task{static_cast<int32_t>(r), c, std::min(c + per, e), slots++});
          sr.n_slots++;
        }
        if (sr.n_slots > FOLD_MIN) {
          for (int32_t g = 0; g < sr.n_slots; g += FOLD_GROUP)
            fg.push_back(FoldGroup{sr.first_slot + g, std::min(FOLD_GROUP, sr.n_slots - g)});
          sr.n_slots = (sr.n_slots + FOLD_GROUP - 1) / FOLD_GROUP;
          sr.slot_stride = FOLD_GROUP;
        }
        sp.push_back(sr);
      } else {
        tk.push_back(Task{static_cast<int32_t>(r), b, e, -1});
      }
    }
    {  // longest first, ties in list order: a counting sort by length (stable, O(n + max length))
      int32_t max_len = 0;
      for (const Task &a : tk) max_len = std::max(max_len, a.end - a.begin);
      std::vector<int32_t> start(static_cast<size_t>(max_len) + 2, 0);
      for (const Task &a : tk) start[max_len - (a.end - a.begin) + 1]++;
      for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
      std::vector<Task> sorted(tk.size());
      for (const Task &a : tk) sorted[start[max_len - (a.end - a.begin)]++] = a;
      tk.swap(sorted);
    }
    // (dealing length strata round-robin so that neighbouring waves sit in different phases
    // was tried: 3-5 % slower, and 2x slower when the stratum count shares a factor with the
    // 8 XCDs the dispatcher deals workgroups to - longest-first keeps the XCDs balanced)
    std::stable_sort(sp.begin(), sp.end(),
                     [](const SplitRow &a, const SplitRow &b) { return a.n_slots > b.n_slots; });
    n_tasks = static_cast<int32_t>(tk.size());
    n_split = static_cast<int32_t>(sp.size());
    n_short = n_short16 = n_short8 = 0;
    while (n_short < n_tasks && tk[n_tasks - 1 - n_short].slot < 0 &&
           tk[n_tasks - 1 - n_short].end - tk[n_tasks - 1 - n_short].begin <= SHORT_MAX) {
      if (tk[n_tasks - 1 - n_short].end - tk[n_tasks - 1 - n_short].begin <= 16) n_short16++;
      if (tk[n_tasks - 1 - n_short].end - tk[n_tasks - 1 - n_short].begin <= 8) n_short8++;
      n_short++;
    }
    long_entries = 0;  // entries of the tasks in front of the short ones (the rank-update kernels' work)
    for (int32_t i = 0; i < n_tasks - n_short; i++) long_entries += tk[i].end - tk[i].begin;
    n_slots = slots;
    if (share) {
      indptr.borrow(share->indptr);
      indices.borrow(share->indices);
      data.borrow(share->data);
    } else {
      if (!indptr_ready) indptr.upload(ip32, s);
      if (!entries_ready) upload_entries(m, s);
    }
    {
      std::vector<int32_t> order(re - rb);
      {  // rows by length, longest first, ties in row order (counting sort)
        int32_t max_len = 0;
        for (int64_t r = rb; r < re; r++) max_len = std::max(max_len, ip32[r + 1] - ip32[r]);
        std::vector<int32_t> start(static_cast<size_t>(max_len) + 2, 0);
        for (int64_t r = rb; r < re; r++) start[max_len - (ip32[r + 1] - ip32[r]) + 1]++;
        for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
        for (int64_t r = rb; r < re; r++)
          order[start[max_len - (ip32[r + 1] - ip32[r])]++] = static_cast<int32_t>(r);
      }
      rows_by_len.upload(order, s);
      if (cfg.K > 128 && cfg.K <= 256) {  // the classes of the matrix-free CG kernels
        size_t i = 0;
        for (int c = 0; c < MF_CLASSES; c++) {
          mf_class[c] = static_cast<int32_t>(i);
          const int32_t lower = c + 1 < MF_CLASSES ? MF_CAPS[c + 1] : -1;  // class c: lower < length <= MF_CAPS[c]
          while (i < order.size() && ip32[order[i] + 1] - ip32[order[i]] > lower) i++;
        }
        mf_class[MF_CLASSES] = static_cast<int32_t>(order.size());
        std::vector<MfLongRow> lr;
        std::vector<MfChunk> ch;
        for (int32_t k = 0; k < mf_class[1]; k++) {
          const int32_t r = order[k], b = ip32[r], e = ip32[r + 1], nz = e - b;
          const int32_t nch = (nz + MF_CHUNK - 1) / MF_CHUNK;
          const int32_t per = (((nz + nch - 1) / nch) + 15) & ~15;  // whole 16-entry groups
          MfLongRow L{r, static_cast<int32_t>(ch.size()), 0, 0};
          for (int32_t c = b; c < e; c += per) {
            ch.push_back(MfChunk{k, c, std::min(c + per, e), 0});
            L.n_chunks++;
          }
          lr.push_back(L);
        }
        mf_n_lrows = static_cast<int32_t>(lr.size());
        mf_n_chunks = static_cast<int32_t>(ch.size());
        mf_lrows.upload(lr, s);
        mf_chunks.upload(ch, s);
        IRS_HIP(hipStreamSynchronize(s));
      }
      n_long = 0;  // rows long enough for a whole workgroup (ialspp_long_kernel)
      while (n_long < static_cast<int32_t>(order.size()) &&
             ip32[order[n_long] + 1] - ip32[order[n_long]] > 2048)
        n_long++;
    }
    if (share) reg.borrow(share->reg);
    else reg.upload(regs, s);
    tasks.upload(tk, s);
    split.upload(sp, s);
    n_fold = static_cast<int32_t>(fg.size());
    fold.upload(fg, s);
    IRS_HIP(hipStreamSynchronize(s));  // host vectors go out of scope
  }
};

struct Profiler {
  struct Rec {
    const char *name;
    hipEvent_t a, b;
  };
  bool enabled = false;
  // Events only on the solve kernel of the larger side ("ials_solve_*_user" when there are more users
  // than items: the dominant kernel).  A start / stop pair on every launch of an epoch costs 0.05 ms
  // of 2.1 (one box: 2.075 ms without events, 2.127 with ten pairs, 2.11 with the two solves', less
  // with one) - a timed run needs the dominant kernel's duration and nothing else (bench.py: timed
  // steps in this mode, every other kernel's duration from extra untimed epochs).
  bool dominant_only = false;
  const char *dominant_suffix = "_user";  // the side with more rows (its solve has the larger flop count)
  bool skip(const char *name) const {
    if (!enabled) return true;
    if (!dominant_only) return false;
    const size_t n = std::strlen(name), m = std::strlen(dominant_suffix);
    return std::strncmp(name, "ials_solve_", 11) != 0 || n < m || std::strcmp(name + n - m, dominant_suffix) != 0;
  }
  bool region_open = false;
  std::vector<Rec> recs;
  std::map<std::string, std::pair<double, int64_t>> totals;
  void begin(const char *name, hipStream_t s) {
    region_open = !skip(name);
    if (!region_open) return;
    Rec r{name, nullptr, nullptr};
    IRS_HIP(hipEventCreate(&r.a));
    IRS_HIP(hipEventCreate(&r.b));
    IRS_HIP(hipEventRecord(r.a, s));
    recs.push_back(r);
  }
  void end(hipStream_t s) {
    if (!region_open) return;
    region_open = false;
    IRS_HIP(hipEventRecord(recs.back().b, s));
  }
  // A region that is ONE kernel launch: the two events ride on the launch itself
  // (hipExtLaunchKernel: start / stop timestamps of the dispatch, no marker packets between the
  // kernels - the separate records cost 1.5 % of a 2.4 ms epoch).
  template <class... Args, class F = void (*)(Args...)>
  void launch(const char *name, F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    if (skip(name)) {
      hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
      return;
    }
    Rec r{name, nullptr, nullptr};
    IRS_HIP(hipEventCreate(&r.a));
    IRS_HIP(hipEventCreate(&r.b));
    if (dominant_only) {
      // one measured launch among plain ones: two marker records around it (the dispatch-attached
      // pair switches the queue's profiling on and off around the launch, ~20 us each way)
      IRS_HIP(hipEventRecord(r.a, s));
      hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
      IRS_HIP(hipEventRecord(r.b, s));
    } else {
      hipExtLaunchKernelGGL(kernel, grid, block, static_cast<uint32_t>(lds), s, r.a, r.b, 0, args...);
    }
    recs.push_back(r);
  }
  void collect() {
    for (auto &r : recs) {
      IRS_HIP(hipEventSynchronize(r.b));
      float ms = 0;
      IRS_HIP(hipEventElapsedTime(&ms, r.a, r.b));
      auto &t = totals[r.name];
      t.first += ms;
      t.second += 1;
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
  void clear() {
    collect();
    totals.clear();
  }
};

}  // namespace ials
}  // namespace irs

using namespace irs;
using namespace irs::ials;

struct irs_ials_trainer {
  irs_ials_model_config cfg;
  int64_t K = 0, n_users = 0, n_items = 0;
  int KP = 0, T = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // iALS++: the short-row launch beside the long-row launch
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t stream3 = nullptr;  // eigenbasis short rows: every other pass
  hipEvent_t ev_fork3 = nullptr, ev_join3 = nullptr;
  bool has_X = false;
  bool whole = true;  // unsharded: both CSR orientations are complete on this device
  irs_ials_shard shard{0, 0, 0, 0};
  DeviceBuffer<float> factor[2];                  // 0 user, 1 item
  Side side[2];                                   // 0: X (solve users), 1: X^T (solve items)
  DeviceBuffer<float> P_raw[2], P[2], P_acc[2], P_accL[2];   // [s]: Gramian used by the solve of side s
  DeviceBuffer<float> gram_partial, split_partial, row_loss;
  DeviceBuffer<double> loss_sum;
  DeviceBuffer<int32_t> err_flag;
  DeviceBuffer<float> prior[2];        // feature prior of the next half step(s), [rows, KP]
  bool has_prior[2] = {false, false};
  // feature matrices (CSR and CSR of the transpose), weights and ridge right-hand sides
  int64_t n_feat[2] = {0, 0};
  DeviceBuffer<int32_t> f_indptr[2], f_indices[2], ft_indptr[2], ft_indices[2];
  DeviceBuffer<float> f_data[2], ft_data[2], f_W[2], f_rhs[2], f_part[2];
  int f_chunks[2] = {1, 1};  // feature_rhs_kernel: chunks of FEATURE_RHS_CHUNK stored rows per feature
  DeviceBuffer<float> pp_pred;         // iALS++ prediction cache (CSR-indexed, padded)
  DeviceBuffer<int32_t> pp_llt_sink;   // iALS++ does not test the LLT status (hpp:495-497)
  DeviceBuffer<float> pp_pblk;         // iALS++ chain path: blocks of P in accumulator layout
  DeviceBuffer<float> pp_px;           // iALS++ one-block sweep on the solve kernel: target @ P
  DeviceBuffer<float> gk_sys, gk_delta;  // general-size path (ials_gk_kernels.hpp): scratch systems
  // eigenbasis short-row path (ials_eig_kernels.hpp)
  DeviceBuffer<float> eig_Qrows, eig_Qcols, eig_lam, eig_stats, eig_table, eig_xt, eig_xt2;
  DeviceBuffer<double> eig_Qd[2];   // per solved side: float64 eigenvectors, the next call's warm start
  bool eig_warm[2] = {false, false};
  int *eig_resident = nullptr;  // host-pinned: the decomposition kernel reports that it has a CU
  int eig_token = 0;
  float eig_stats_host[4] = {0, 0, 0, 0};
  bool opt_eig = true;  // IRSPACK_AMD_IALS_EIG
  // matrix-free CG at 128 < K <= 256 (ials_mf_kernels.hpp): state of the level-synchronous rows
  DeviceBuffer<float> mf_vec, mf_xs, mf_rs, mf_partial, mf_r2;
  DeviceBuffer<int32_t> mf_done;
  bool opt_mf = true;   // IRSPACK_AMD_IALS_MF
  // irs_ials_sharded_step: the Gramian of side s has been all-reduced ahead of its half-epoch
  bool gram_prefetched[2] = {false, false};
  // ... and the rows of the shard cut into chunks that are solved and exchanged one after the
  // other (IRSPACK_AMD_SHARD_CHUNKS at creation; empty: one solve, one exchange)
  std::vector<std::unique_ptr<Side>> chunk_side[2];
  int32_t eig_last = 0; // 1: the last half-step took the eigenbasis path (diagnostics)
  bool gk() const { return KP > 256; }   // K > 256: every size is a run-time value
  Profiler prof;
  bool opt_wave128 = true, opt_unit = true, opt_short = true, opt_wg16 = true;  // read_switches()
  bool opt_short2 = true;  // two short rows per wave (IRSPACK_AMD_IALS_SHORT2)
  bool opt_pp_direct = true;  // iALS++ with one block = the direct solve (IRSPACK_AMD_IALSPP_DIRECT)
  bool opt_pp_chain = true;   // iALS++ prediction passes merged into the rank updates (IRSPACK_AMD_IALSPP_CHAIN)
  bool opt_pp_fork = true;    // iALS++ short-row launch on a second stream (IRSPACK_AMD_IALSPP_FORK)
  bool opt_bf16x3 = false;

  int64_t rows_of(int which) const { return which == 0 ? n_users : n_items; }
};

namespace {

// T = KP / 16.  T <= 4: one wave per row (ials_kernels.hpp); T = 8 / 12 / 16: one
// workgroup per row (ials_wg_kernels.hpp).
#define IRS_DISPATCH_T(t, ...)                                                      \
  switch (t) {                                                                      \
    case 1: { constexpr int TT = 1; __VA_ARGS__; } break;                           \
    case 2: { constexpr int TT = 2; __VA_ARGS__; } break;                           \
    case 4: { constexpr int TT = 4; __VA_ARGS__; } break;                           \
    default:                                                                        \
      throw std::invalid_argument("irspack_amd: unsupported padded latent dimension."); \
  }
#define IRS_DISPATCH_T8(t, ...)                                                     \
  switch (t) {                                                                      \
    case 1: { constexpr int TT = 1; __VA_ARGS__; } break;                           \
    case 2: { constexpr int TT = 2; __VA_ARGS__; } break;                           \
    case 4: { constexpr int TT = 4; __VA_ARGS__; } break;                           \
    case 8: { constexpr int TT = 8; __VA_ARGS__; } break;                           \
    default:                                                                        \
      throw std::invalid_argument("irspack_amd: unsupported padded latent dimension."); \
  }
#define IRS_DISPATCH_TW(t, ...)                                                     \
  switch (t) {                                                                      \
    case 8: { constexpr int TT = 8; __VA_ARGS__; } break;                           \
    case 12: { constexpr int TT = 12; __VA_ARGS__; } break;                         \
    case 16: { constexpr int TT = 16; __VA_ARGS__; } break;                         \
    default:                                                                        \
      throw std::invalid_argument("irspack_amd: unsupported padded latent dimension."); \
  }
#define IRS_DISPATCH_ANY(t, ...)                                                    \
  switch (t) {                                                                      \
    case 1: { constexpr int TT = 1; __VA_ARGS__; } break;                           \
    case 2: { constexpr int TT = 2; __VA_ARGS__; } break;                           \
    case 4: { constexpr int TT = 4; __VA_ARGS__; } break;                           \
    case 8: { constexpr int TT = 8; __VA_ARGS__; } break;                           \
    case 12: { constexpr int TT = 12; __VA_ARGS__; } break;                         \
    case 16: { constexpr int TT = 16; __VA_ARGS__; } break;                         \
    default:                                                                        \
      throw std::invalid_argument("irspack_amd: unsupported padded latent dimension."); \
  }

void validate_config(const irs_ials_model_config &c) {
  check_arg(c.K >= 1, "K must be positive.");
  // (no upper limit, like the reference: above 256 the general-size kernels of
  // ials_gk_kernels.hpp take over; 2^15 keeps K * K and the tile offsets inside int32)
  check_arg(c.K <= 32768, "irspack_amd: n_components above 32768 is not supported.");
  check_arg(c.loss_type == IRS_LOSS_ORIGINAL || c.loss_type == IRS_LOSS_IALSPP,
            "unknown loss_type.");
}

void read_switches(irs_ials_trainer *t);

void alloc_common(irs_ials_trainer *t) {
  read_switches(t);
  t->KP = padded_k(t->K);
  t->T = t->KP / 16;
  for (int w = 0; w < 2; w++) {
    // rows padded to a multiple of 8 (zero, never solved, never gathered) so that 1 / 2 / 4 / 8
    // equal row shards tile the buffer exactly and one in-place all-gather can move them
    // rows padded to a multiple of 8, plus 8 rows that stay zero (UNIT kernels gather row
    // `padded rows` for the entries past a row's end)
    t->factor[w].alloc(static_cast<size_t>(ceil_div(t->rows_of(w), 8) * 8 + 8) * t->KP);
    t->factor[w].zero(t->stream);
    t->P_raw[w].alloc(t->KP * t->KP);
    t->P[w].alloc(t->KP * t->KP);
    t->P_acc[w].alloc(t->KP * t->KP);
    t->P_accL[w].alloc(t->KP * t->KP);
    t->P_accL[w].zero(t->stream);
    t->P_raw[w].zero(t->stream);
    t->P[w].zero(t->stream);
    t->P_acc[w].zero(t->stream);
  }
  t->err_flag.alloc(1);
  t->err_flag.zero(t->stream);
  t->loss_sum.alloc(2);
}

// `s`: the stream of the copy (default: the trainer's; the constructor uploads beside its other work)
void upload_factor(irs_ials_trainer *t, int which, const float *host, hipStream_t s = nullptr) {
  if (!s) s = t->stream;
  t->gram_prefetched[0] = t->gram_prefetched[1] = false;  // (of irs_ials_sharded_step: the factors change)
  const int64_t n = t->rows_of(which);
  t->factor[which].zero(s);
  if (n > 0)
    IRS_HIP(hipMemcpy2DAsync(t->factor[which].ptr, t->KP * sizeof(float), host,
                             t->K * sizeof(float), t->K * sizeof(float), n,
                             hipMemcpyHostToDevice, s));
  IRS_HIP(hipStreamSynchronize(s));
}

void download_factor(irs_ials_trainer *t, const float *dev, int64_t n, float *host) {
  if (n > 0)
    IRS_HIP(hipMemcpy2DAsync(host, t->K * sizeof(float), dev, t->KP * sizeof(float),
                             t->K * sizeof(float), n, hipMemcpyDeviceToHost, t->stream));
  IRS_HIP(hipStreamSynchronize(t->stream));
}

void init_factor(irs_ials_trainer *t, int which) {
  const RawVector<float> h = draw_factor(t->cfg.init_stdev, t->cfg.random_seed, t->K, t->rows_of(which));
  upload_factor(t, which, h.data());
}

// Gramian of `which` factors over rows [rb, re) into P_raw[dst] (unscaled).
// `finish`: the reduce kernel also writes what launch_finish_gramian would (K <= 64; the caller then
// skips that launch) - only where nothing (an all-reduce over ranks) sits between the two
void launch_partial_gramian(irs_ials_trainer *t, int which, int64_t rb, int64_t re, int dst, bool finish = false) {
  const int64_t n = std::max<int64_t>(re - rb, 0);
  if (t->gk()) {  // K > 256: (64 x 64 block pair, row slab) units, slabs summed in order
    const int KP = t->KP, nb = KP / 64, nbp = nb * (nb + 1) / 2;
    const int n_slabs = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(64, ceil_div(n, 256))));
    const int64_t per = ceil_div(ceil_div(std::max<int64_t>(n, 1), n_slabs), 4) * 4;
    t->gram_partial.alloc(static_cast<size_t>(n_slabs) * KP * KP);
    t->prof.begin("gramian_partial", t->stream);
    hipLaunchKernelGGL(gk_gramian_partial_kernel, dim3(ceil_div(static_cast<int64_t>(n_slabs) * nbp, 4)),
                       dim3(256), 0, t->stream, static_cast<const float *>(t->factor[which].ptr), KP, rb,
                       re, per, n_slabs, t->gram_partial.ptr);
    t->prof.end(t->stream);
    t->prof.begin("gramian_reduce", t->stream);
    hipLaunchKernelGGL(gk_gramian_reduce_kernel, dim3(ceil_div(static_cast<int64_t>(KP) * KP, 256)),
                       dim3(256), 0, t->stream, static_cast<const float *>(t->gram_partial.ptr), KP,
                       n_slabs, t->P_raw[dst].ptr);
    t->prof.end(t->stream);
    IRS_HIP(hipGetLastError());
    return;
  }
  int64_t n_waves = std::min<int64_t>(1024, std::max<int64_t>(1, ceil_div(n, 64)));
  n_waves = ceil_div(n_waves, 4) * 4;
  int64_t per = ceil_div(std::max<int64_t>(n, 1), n_waves);
  per = ceil_div(per, 4) * 4;
  if (t->T <= 4) {
    IRS_DISPATCH_T(t->T, {
      using G = Geo<TT>;
      t->gram_partial.alloc(static_cast<size_t>(n_waves / 4) * G::NT * 256);
      t->prof.launch("gramian_partial", gramian_partial_kernel<TT>, dim3(n_waves / 4), dim3(256), 0,
                     t->stream, static_cast<const float *>(t->factor[which].ptr), rb, re, per,
                     t->gram_partial.ptr);
      if (finish)
        t->prof.launch("gramian_reduce", gramian_reduce_kernel<TT, true>, dim3(ceil_div(G::NT * 256, 64)),
                       dim3(64 * GRAM_CHAINS), 0, t->stream, static_cast<const float *>(t->gram_partial.ptr),
                       n_waves / 4, t->P_raw[dst].ptr, t->cfg.alpha0, t->P[dst].ptr, t->P_acc[dst].ptr,
                       t->P_accL[dst].ptr);
      else
        t->prof.launch("gramian_reduce", gramian_reduce_kernel<TT>, dim3(ceil_div(G::NT * 256, 64)),
                       dim3(64 * GRAM_CHAINS), 0, t->stream, static_cast<const float *>(t->gram_partial.ptr),
                       n_waves / 4, t->P_raw[dst].ptr, 0.f, static_cast<float *>(nullptr),
                       static_cast<float *>(nullptr), static_cast<float *>(nullptr));
    });
  } else {
    // four waves of a block share a slab of rows and split the tiles between them
    // (a wave has ONE 4-row load in flight per step: with one block per CU ten million rows took
    // 5.6 ms, 0.9 TB/s; more resident blocks hide the latency - up to eight per CU for long sides)
    const int64_t n_blocks = std::min<int64_t>(std::max<int64_t>(256, std::min<int64_t>(2048, n / 1024)),
                                               std::max<int64_t>(1, ceil_div(n, 64)));
    int64_t per_block = ceil_div(std::max<int64_t>(n, 1), n_blocks);
    per_block = ceil_div(per_block, 4) * 4;
    IRS_DISPATCH_TW(t->T, {
      using G = Geo<TT>;
      t->gram_partial.alloc(static_cast<size_t>(n_blocks) * G::NT * 256);
      t->prof.begin("gramian_partial", t->stream);
      hipLaunchKernelGGL((gramian_partial_wg_kernel<TT>), dim3(n_blocks), dim3(256), 0, t->stream,
                         t->factor[which].ptr, rb, re, per_block, t->gram_partial.ptr);
      t->prof.end(t->stream);
      t->prof.begin("gramian_reduce", t->stream);
      hipLaunchKernelGGL((gramian_reduce_kernel<TT>), dim3(ceil_div(G::NT * 256, 64)), dim3(64 * GRAM_CHAINS),
                         0, t->stream, t->gram_partial.ptr, n_blocks, t->P_raw[dst].ptr, 0.f,
                         static_cast<float *>(nullptr), static_cast<float *>(nullptr), static_cast<float *>(nullptr));
      t->prof.end(t->stream);
    });
  }
  IRS_HIP(hipGetLastError());
}

void launch_finish_gramian(irs_ials_trainer *t, int dst) {
  if (t->gk()) {
    const int64_t n = static_cast<int64_t>(t->KP) * t->KP;
    t->prof.begin("gramian_finish", t->stream);
    hipLaunchKernelGGL(gk_gramian_finish_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, t->stream,
                       static_cast<const float *>(t->P_raw[dst].ptr), t->cfg.alpha0, n, t->P[dst].ptr);
    t->prof.end(t->stream);
    IRS_HIP(hipGetLastError());
    return;
  }
  IRS_DISPATCH_ANY(t->T, {
    using G = Geo<TT>;
    t->prof.launch("gramian_finish", gramian_finish_kernel<TT>, dim3(ceil_div(G::KP * G::KP, 256)),
                   dim3(256), 0, t->stream, static_cast<const float *>(t->P_raw[dst].ptr),
                   t->cfg.alpha0, t->P[dst].ptr, t->P_acc[dst].ptr, t->P_accL[dst].ptr);
  });
  IRS_HIP(hipGetLastError());
}

void check_solver(const irs_ials_solver_config *sc) {
  check_arg(sc != nullptr, "solver_config is null.");
  check_arg(sc->n_threads > 0, "n_threads must be strictly positive.");  // hpp:81-83
  check_arg(sc->solver_type == IRS_SOLVER_CHOLESKY || sc->solver_type == IRS_SOLVER_CG ||
                sc->solver_type == IRS_SOLVER_IALSPP,
            "unknown solver_type.");
}

// ---- general-size path (ials_gk_kernels.hpp): K > 256 and iALS++ blocks wider than 64 ----
GkParams gk_params(irs_ials_trainer *t, Side &sd, const float *other, float *target, int pidx,
                   const float *prior) {
  GkParams p{};
  p.rows = sd.rows_by_len.ptr;
  p.row_first = 0;
  p.n_rows = static_cast<int32_t>(sd.row_end - sd.row_begin);
  p.indptr = sd.indptr.ptr;
  p.indices = sd.indices.ptr;
  p.data = sd.data.ptr;
  p.pred = nullptr;
  p.other = other;
  p.ld_other = t->KP;
  p.target = target;
  p.ld_target = t->KP;
  p.reg = sd.reg.ptr;
  p.P = t->P[pidx].ptr;
  p.ldP = t->KP;
  p.prior = prior;
  p.c0 = 0;
  p.D = static_cast<int32_t>(t->K);
  p.Np = static_cast<int32_t>(ceil_div(t->K, 64) * 64);
  p.K = static_cast<int32_t>(t->K);
  p.bias = t->cfg.loss_type == IRS_LOSS_IALSPP ? 0.0f : t->cfg.alpha0;
  p.mode = 0;
  p.err_flag = t->err_flag.ptr;
  return p;
}

// rows of one pass through the scratch systems (at most IRSPACK_AMD_GK_SCRATCH_MB, default 2048)
int32_t gk_batch_rows(irs_ials_trainer *t, GkParams &p) {
  const int64_t nt = p.Np / 16;
  p.sys_floats = nt * (nt + 1) / 2 * 256 + p.Np;
  static const int64_t budget_mb = [] {
    const char *e = std::getenv("IRSPACK_AMD_GK_SCRATCH_MB");
    return e ? std::max<int64_t>(1, std::atoll(e)) : int64_t(2048);
  }();
  const int64_t fit = (budget_mb << 20) / (p.sys_floats * static_cast<int64_t>(sizeof(float)));
  const int32_t B = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(p.n_rows, fit)));
  t->gk_sys.alloc(static_cast<size_t>(B) * p.sys_floats);
  p.sys = t->gk_sys.ptr;
  return B;
}

// rank update + blocked Cholesky of the rows [first, first + m) of the launch order
void gk_build_and_solve(irs_ials_trainer *t, GkParams p, int32_t first, int32_t m) {
  p.row_first = first;
  p.n_rows = m;
  const int nb = p.Np / 64, nbp = nb * (nb + 1) / 2;
  if (p.mode == 1)
    hipLaunchKernelGGL(gk_block_rhs0_kernel, dim3(m), dim3(256), static_cast<size_t>(p.ldP) * sizeof(float),
                       t->stream, p);
  hipLaunchKernelGGL(gk_syrk_kernel, dim3(ceil_div(static_cast<int64_t>(m) * nbp, 4)), dim3(256), 0,
                     t->stream, p);
  const size_t lds = (4 * 2 * 16 * 17 + p.Np + 16) * sizeof(float);
  hipLaunchKernelGGL(gk_chol_kernel, dim3(m), dim3(256), lds, t->stream, p);
}

// Solver::step_cholesky / step_cg for K > 256
void launch_gk_solve(irs_ials_trainer *t, Side &sd, const float *other, float *target, int pidx,
                     const irs_ials_solver_config *sc, const float *prior, int32_t *err_flag) {
  GkParams p = gk_params(t, sd, other, target, pidx, prior);
  p.err_flag = err_flag;
  const int32_t n = p.n_rows;
  if (n <= 0) return;
  if (sc->solver_type == IRS_SOLVER_CG) {
    const int steps = sc->max_cg_steps == 0 ? static_cast<int>(t->K)
                                            : static_cast<int>(std::min<uint64_t>(sc->max_cg_steps, 1u << 20));
    const size_t lds = (4 * static_cast<size_t>(t->KP) + 8) * sizeof(float) +
                       4 * static_cast<size_t>(t->KP) * sizeof(double);  // vectors + float64 partial sums
    // a row's CG vectors live in the workgroup's LDS (160 KB per compute unit on gfx950): say so
    // instead of letting the launch fail (Cholesky and iALS++ work from global scratch at any K)
    if (lds > 160u * 1024u)
      throw std::invalid_argument(
          "solver_type = CG keeps 48 bytes of a row's vectors per (padded) factor dimension in LDS: "
          "n_components = " + std::to_string(t->K) + " needs " + std::to_string(lds) +
          " bytes, the device has 163840 (n_components <= 3392 works); use CHOLESKY or IALSPP above.");
    IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gk_cg_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    t->prof.begin(pidx == 0 ? "ials_solve_cg_user" : "ials_solve_cg_item", t->stream);
    hipLaunchKernelGGL(gk_cg_kernel, dim3(n), dim3(256), lds, t->stream, p, steps, 1);
    t->prof.end(t->stream);
  } else {
    const int32_t B = gk_batch_rows(t, p);
    t->prof.begin(pidx == 0 ? "ials_solve_cholesky_user" : "ials_solve_cholesky_item", t->stream);
    for (int32_t first = 0; first < n; first += B) gk_build_and_solve(t, p, first, std::min(B, n - first));
    t->prof.end(t->stream);
  }
  IRS_HIP(hipGetLastError());
}

// Solver::step_ialspp (hpp:516-535) with blocks wider than 64 dims, or at K > 256: per block the
// Newton step of _step_dimrange (hpp:436-502) through the scratch systems; the prediction cache
// (hpp:410-413) is corrected after every block but the last (hpp:500-506).
void launch_ialspp_general(irs_ials_trainer *t, Side &sd, const float *other, float *target, int pidx,
                           const irs_ials_solver_config *sc) {
  GkParams base = gk_params(t, sd, other, target, pidx, nullptr);
  if (!t->pp_llt_sink.ptr) {
    t->pp_llt_sink.alloc(1);
    t->pp_llt_sink.zero(t->stream);
  }
  base.err_flag = t->pp_llt_sink.ptr;  // the reference does not test the LLT status here (hpp:495-497)
  const int32_t n = base.n_rows;
  if (n <= 0) return;
  t->pp_pred.alloc(static_cast<size_t>(sd.nnz) + 320);
  const int32_t K = static_cast<int32_t>(t->K);
  const int32_t sub = static_cast<int32_t>(std::min<uint64_t>(std::max<uint64_t>(1, sc->ialspp_subspace_dimension), t->K));
  base.mode = 1;
  base.pred = t->pp_pred.ptr;
  for (uint64_t it = 0; it < sc->ialspp_iteration; it++) {
    t->prof.begin(pidx == 0 ? "ials_ialspp_user" : "ials_ialspp_item", t->stream);
    hipLaunchKernelGGL(gk_pred_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, t->stream, base, nullptr,
                       t->pp_pred.ptr, 0);
    for (int32_t c0 = 0; c0 < K; c0 += sub) {
      GkParams p = base;
      p.c0 = c0;
      p.D = std::min(sub, K - c0);
      p.Np = static_cast<int32_t>(ceil_div(p.D, 64) * 64);
      const int32_t B = gk_batch_rows(t, p);
      const bool last = c0 + sub >= K;
      if (!last) t->gk_delta.alloc(static_cast<size_t>(B) * p.D);
      for (int32_t first = 0; first < n; first += B) {
        const int32_t m = std::min(B, n - first);
        GkParams q = p;
        q.row_first = first;
        q.n_rows = m;
        const int64_t nd = static_cast<int64_t>(m) * p.D;
        if (!last)
          hipLaunchKernelGGL(gk_block_delta_kernel, dim3(ceil_div(nd, 256)), dim3(256), 0, t->stream, q,
                             t->gk_delta.ptr, 0);
        gk_build_and_solve(t, p, first, m);
        if (!last) {
          hipLaunchKernelGGL(gk_block_delta_kernel, dim3(ceil_div(nd, 256)), dim3(256), 0, t->stream, q,
                             t->gk_delta.ptr, 1);
          hipLaunchKernelGGL(gk_pred_kernel, dim3(ceil_div(m, 4)), dim3(256), 0, t->stream, q,
                             static_cast<const float *>(t->gk_delta.ptr), t->pp_pred.ptr, 1);
        }
      }
    }
    t->prof.end(t->stream);
  }
  IRS_HIP(hipGetLastError());
}

// Solver::step_ialspp / step_icd (hpp:516-630): `ialspp_iteration` sweeps, one wave per row.
void launch_ialspp(irs_ials_trainer *t, Side &sd, const float *other, float *target, int pidx,
                   const irs_ials_solver_config *sc) {
  static const char *kNames[2] = {"ials_ialspp_user", "ials_ialspp_item"};
  const int32_t n_rows = static_cast<int32_t>(sd.row_end - sd.row_begin);
  if (n_rows <= 0) return;
  // blocks wider than 64 dims: run-time sizes (ials_gk_kernels.hpp).  Blocks up to 64 dims keep
  // the tuned kernels at every K: they only hold the x row (KP floats) per wave beyond the block.
  if (sc->ialspp_subspace_dimension > 64 || t->KP > 2048) {
    launch_ialspp_general(t, sd, other, target, pidx, sc);
    return;
  }
  t->pp_pred.alloc(static_cast<size_t>(sd.nnz) + 320);
  if (!t->pp_llt_sink.ptr) {
    t->pp_llt_sink.alloc(1);
    t->pp_llt_sink.zero(t->stream);
  }
  PpParams p;
  p.indptr = sd.indptr.ptr;
  p.indices = sd.indices.ptr;
  p.data = sd.data.ptr;
  p.rows = sd.rows_by_len.ptr;
  p.n_rows = n_rows;
  p.other = other;
  p.target = target;
  p.reg = sd.reg.ptr;
  p.P = t->P[pidx].ptr;
  p.pred = t->pp_pred.ptr;
  p.ignored_flag = t->pp_llt_sink.ptr;
  p.bias = t->cfg.loss_type == IRS_LOSS_IALSPP ? 0.0f : t->cfg.alpha0;  // hpp:431-432
  p.K = static_cast<int32_t>(t->K);
  p.KP = t->KP;
  p.xs_extra = std::max(0, t->KP - 256);
  // a subspace dimension of 0 or 1 is the iCD branch (hpp:673-677)
  p.sub = static_cast<int32_t>(std::max<uint64_t>(1, sc->ialspp_subspace_dimension));
  p.zero_start = 0;
  p.chain = t->opt_pp_chain ? 1 : 0;
  p.P_blk = nullptr;
  if (p.chain && p.sub == 64 && p.K > 64) {  // the blocks of P in accumulator layout
    const int nb = (p.K + 63) / 64;
    t->pp_pblk.alloc(static_cast<size_t>(nb) * 10 * 256);
    hipLaunchKernelGGL(pp_pack_p_kernel, dim3(nb), dim3(64), 0, t->stream, t->P[pidx].ptr, p.K, p.KP,
                       t->pp_pblk.ptr);
    p.P_blk = t->pp_pblk.ptr;
  }
  const int D = std::min<int>(p.sub, p.K);
  const int TS = D <= 16 ? 1 : D <= 32 ? 2 : 4;
  const bool aligned = p.sub % TS == 0;
  // rows above 2048 stored entries get a whole workgroup each (they come first in
  // rows_by_len), the rest one wave each
  const int32_t n_long = std::min(sd.n_long, n_rows);
  auto launch = [&](auto long_kernel, size_t long_floats, auto kernel, size_t lds_floats) {
    const size_t lds = 4 * (lds_floats + p.xs_extra) * sizeof(float),
                 lds_long = (long_floats + p.xs_extra) * sizeof(float);
    IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(long_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds_long)));
    // The two launches touch different rows.  The long-row launch ends with a tail (a few rows
    // of 10^5 entries on a few CUs), so the one-wave-per-row launch runs beside it on a second
    // stream (fork / join by events) instead of behind it.
    const bool fork = n_long > 0 && n_rows > n_long && t->opt_pp_fork;
    if (fork && !t->stream2) {
      IRS_HIP(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
      IRS_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
      IRS_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
    }
    for (uint64_t it = 0; it < sc->ialspp_iteration; it++) {
      t->prof.begin(kNames[pidx], t->stream);
      hipStream_t s2 = t->stream;
      if (fork) {
        s2 = t->stream2;
        IRS_HIP(hipEventRecord(t->ev_fork, t->stream));
        IRS_HIP(hipStreamWaitEvent(s2, t->ev_fork, 0));
      }
      if (n_long > 0) {
        PpParams pl = p;
        pl.n_rows = n_long;
        hipLaunchKernelGGL(long_kernel, dim3(n_long), dim3(64 * PP_LONG_WAVES), lds_long, t->stream,
                           pl);
      }
      if (n_rows > n_long) {
        PpParams ps = p;
        ps.rows = p.rows + n_long;
        ps.n_rows = n_rows - n_long;
        hipLaunchKernelGGL(kernel, dim3(ceil_div(ps.n_rows, 4)), dim3(256), lds, s2, ps);
      }
      if (fork) {
        IRS_HIP(hipEventRecord(t->ev_join, s2));
        IRS_HIP(hipStreamWaitEvent(t->stream, t->ev_join, 0));
      }
      t->prof.end(t->stream);
    }
  };
  // the form of the sweep is a template parameter (each form gets its own register budget)
  const bool single = p.sub >= p.K;
  const bool chained = !single && TS == 4 && aligned && p.chain && p.sub == 64 && p.K > 64;
#define IRS_PP_LAUNCH(TSV, AL)                                                                        \
  do {                                                                                                \
    if (single)                                                                                       \
      launch(ialspp_long_kernel<TSV, AL, 1>, PpLongGeo<TSV>::LDS_FLOATS, ialspp_kernel<TSV, AL, 1>,   \
             PpGeo<TSV>::LDS_FLOATS);                                                                 \
    else                                                                                              \
      launch(ialspp_long_kernel<TSV, AL, 0>, PpLongGeo<TSV>::LDS_FLOATS, ialspp_kernel<TSV, AL, 0>,   \
             PpGeo<TSV>::LDS_FLOATS);                                                                 \
  } while (0)
  switch (TS * 2 + (aligned ? 1 : 0)) {
    case 2:  // TS = 1 is always aligned
    case 3: IRS_PP_LAUNCH(1, true); break;
    case 4: IRS_PP_LAUNCH(2, false); break;
    case 5: IRS_PP_LAUNCH(2, true); break;
    case 8: IRS_PP_LAUNCH(4, false); break;
    default:
      if (chained)
        launch(ialspp_long_kernel<4, true, 2>, PpLongGeo<4>::LDS_FLOATS, ialspp_kernel<4, true, 2>,
               PpGeo<4>::LDS_FLOATS);
      else
        IRS_PP_LAUNCH(4, true);
      break;
  }
#undef IRS_PP_LAUNCH
  IRS_HIP(hipGetLastError());
}

// Development switches, read when a trainer is created (so one process can A/B two trainers):
//   IRSPACK_AMD_IALS_WAVE128=0  sends K in (64, 128] to the workgroup-per-row kernels
//   IRSPACK_AMD_IALS_UNIT=0     keeps binary interactions on the general rank-update code
//   IRSPACK_AMD_IALS_SHORT=0    sends the short rows of a CG step through the general kernels
//   IRSPACK_AMD_IALS_WG16=0     Cholesky at K > 64 on the first-generation kernels (4-row panels)
//   IRSPACK_AMD_IALS_BF16X3=0   binary interactions, Cholesky and CG, 48 < K <= 128: the rank update back on the
//                               fp32-input matrix instruction.  Default (round 6): the bf16 matrix cores
//                               on exact three-way splits of the fp32 values, six fp32-exact partial
//                               products per product, fp32 accumulate (syrk_gather_bf16x3, ials_kernels.hpp)
void read_switches(irs_ials_trainer *t) {
  t->opt_wave128 = env_flag("IRSPACK_AMD_IALS_WAVE128", true);
  t->opt_unit = env_flag("IRSPACK_AMD_IALS_UNIT", true);
  t->opt_short = env_flag("IRSPACK_AMD_IALS_SHORT", true);
  t->opt_short2 = env_flag("IRSPACK_AMD_IALS_SHORT2", true);
  t->opt_pp_direct = env_flag("IRSPACK_AMD_IALSPP_DIRECT", true);
  t->opt_pp_chain = env_flag("IRSPACK_AMD_IALSPP_CHAIN", true);
  t->opt_pp_fork = env_flag("IRSPACK_AMD_IALSPP_FORK", true);
  t->opt_wg16 = env_flag("IRSPACK_AMD_IALS_WG16", true);
  t->opt_bf16x3 = env_flag("IRSPACK_AMD_IALS_BF16X3", true);
  t->opt_eig = env_flag("IRSPACK_AMD_IALS_EIG", true);
  t->opt_mf = env_flag("IRSPACK_AMD_IALS_MF", true);
}

// Eigen-decomposition of P[pidx] (row-major [KP, KP]) into the trainer's eig_* buffers.
void launch_eigen(irs_ials_trainer *t, int pidx, hipStream_t stream) {
  const int KP = t->KP;
  t->eig_Qrows.alloc(static_cast<size_t>(KP) * KP);
  t->eig_Qcols.alloc(static_cast<size_t>(KP) * KP);
  t->eig_lam.alloc(KP);
  t->eig_stats.alloc(4);
  t->eig_Qd[pidx].alloc(static_cast<size_t>(KP) * KP);
  if (!t->eig_resident) {
    IRS_HIP(hipHostMalloc(reinterpret_cast<void **>(&t->eig_resident), sizeof(int), hipHostMallocCoherent));
    *t->eig_resident = 0;
  }
  EigOut o{t->eig_Qrows.ptr, t->eig_Qcols.ptr, t->eig_lam.ptr, t->eig_stats.ptr, t->eig_Qd[pidx].ptr,
           t->eig_resident, ++t->eig_token};
  const int warm = t->eig_warm[pidx] ? 1 : 0;
  t->eig_warm[pidx] = true;
  const size_t lds = static_cast<size_t>(KP) * KP * sizeof(double);
  auto launch = [&](auto kernel) {
    IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    t->prof.begin("eig_jacobi", stream);
    hipLaunchKernelGGL(kernel, dim3(1), dim3(1024), lds, stream,
                       static_cast<const float *>(t->P[pidx].ptr), static_cast<int>(t->K), o, warm);
    t->prof.end(stream);
  };
  if (KP == 128) launch(eig_jacobi_kernel<128>);
  else launch(eig_jacobi_kernel<64>);
  IRS_HIP(hipGetLastError());
}

// Short rows (<= 32 stored entries, the tail of the task list) in the eigenbasis of P
// (ials_eig_kernels.hpp), on a SECOND stream beside the kernels of the longer rows: the
// eigen-decomposition is one workgroup (1 - 3 ms of one CU) and must not leave the other 255 idle.
//   eig_begin   decides from host-side facts whether the path is a candidate and, if so, forks the
//               second stream and starts the decomposition + the copy of its statistics;
//   eig_finish  (after the caller has launched the long rows on the main stream) waits for the
//               statistics, checks the conditioning, runs table product / short-row kernels /
//               rotation on the second stream and joins it.  Returns false - nothing launched
//               that changes the factors - when the path declines; the caller then runs its
//               usual kernels over those rows.
bool eig_begin(irs_ials_trainer *t, Side &sd, int pidx, bool cg) {
  t->eig_last = 0;
  const int KP = t->KP;
  if (!t->opt_eig || (KP != 64 && KP != 128) || !sd.positive || sd.n_short <= 0) return false;
  // worth one eigen-decomposition (milliseconds of ONE compute unit) and the product that takes
  // every gathered row into the eigenbasis (n_other x KP x KP multiply-adds, 8 KP bytes per row)?
  // configs[3]: the user side (9.7 M short rows, 10^6 gathered rows) yes, the item side (0.8 M
  // short rows, 10^7 gathered rows: the product alone would cost more than those rows) no.
  if (static_cast<double>(sd.n_short) * KP * KP * KP < 3e11) return false;
  // Cholesky saves a dense KP^3 / 3 factorisation per short row, CG only its products with P, and
  // the table product competes with the long rows' kernels beside it.  configs[3], item side
  // (0.8 M short rows, 10^7 gathered rows) on this path: Cholesky epoch 60.2 -> 52.1 ms, CG epoch
  // 48.8 -> 50.9 ms.
  const double ratio = cg ? 64.0 : 8.0;
  if (static_cast<double>(sd.n_short) * KP < ratio * static_cast<double>(sd.n_other)) return false;
  if (!t->stream2) {
    IRS_HIP(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
    IRS_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
    IRS_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
  }
  IRS_HIP(hipEventRecord(t->ev_fork, t->stream));  // (the Gramian of this half-step is on t->stream)
  IRS_HIP(hipStreamWaitEvent(t->stream2, t->ev_fork, 0));
  launch_eigen(t, pidx, t->stream2);
  // the decomposition needs a CU to itself (its LDS): the long rows' kernels, which the caller
  // launches next, must not reach the device before it is resident.  It reports that through a
  // host-pinned word; the wait is the time the stream's earlier work (the Gramian) still needs,
  // bounded so that a lost report costs speed, not progress.
  // NOTE (blocking): on a side that takes this path half_step_async / irs_ials_step are therefore NOT
  // asynchronous - the host waits here for the Gramian to finish (then yields its core instead of
  // spinning hot) and again in eig_finish for the decomposition's statistics (the conditioning
  // decision is taken on the host).  A report that never arrives is logged: it costs 200 ms per
  // half-step, silently it would look like a slow kernel.
  const auto t0 = std::chrono::steady_clock::now();
  bool seen = false;
  int spins = 0;
  while (!(seen = __atomic_load_n(t->eig_resident, __ATOMIC_ACQUIRE) == t->eig_token) &&
         std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(200))
    if (++spins > 2000) std::this_thread::yield();
  if (!seen) {
    static std::atomic<int> warned{0};
    if (warned.fetch_add(1) < 3)
      fprintf(stderr, "irspack_amd: the eigen-decomposition kernel did not report residency within 200 ms "
                      "(side %d); the long rows' kernels are launched without waiting for it\n", pidx);
  }
  return true;
}

bool eig_finish(irs_ials_trainer *t, Side &sd, const float *other, float *target, int pidx, bool cg,
                int max_cg_steps, int32_t *err_flag) {
  const int KP = t->KP;
  hipStream_t s2 = t->stream2;
  IRS_HIP(hipMemcpyAsync(t->eig_stats_host, t->eig_stats.ptr, 3 * sizeof(float), hipMemcpyDeviceToHost, s2));
  IRS_HIP(hipStreamSynchronize(s2));
  const float *st = t->eig_stats_host;
  // M = P + reg_r I must be well conditioned for EVERY row: float32 carries Q and 1 / (lambda + reg)
  const double lo = std::max(0.0, static_cast<double>(st[1])) + sd.reg_min;
  const double hi = static_cast<double>(st[0]) + sd.reg_min;
  if (!(lo > 0.0) || !(hi <= 1e4 * lo) || !std::isfinite(hi)) return false;
  // V~ = other Q  (every gathered row once per half-step)
  const int64_t n_other = sd.n_other;
  t->eig_table.alloc(static_cast<size_t>(std::max<int64_t>(n_other, 1)) * KP);
  check_arg(n_other < (int64_t(1) << 31) - 64, "eigenbasis table: too many gathered rows");
  t->prof.begin("eig_table", s2);
  if (KP == 128)
    hipLaunchKernelGGL((rows_times_matT_kernel<128>), dim3(ceil_div(n_other, 256)), dim3(256), 0, s2, other,
                       static_cast<const Task *>(nullptr), t->eig_table.ptr, static_cast<const Task *>(nullptr),
                       static_cast<const float *>(t->eig_Qrows.ptr), static_cast<int>(n_other));
  else
    hipLaunchKernelGGL((rows_times_matT_kernel<64>), dim3(ceil_div(n_other, 256)), dim3(256), 0, s2, other,
                       static_cast<const Task *>(nullptr), t->eig_table.ptr, static_cast<const Task *>(nullptr),
                       static_cast<const float *>(t->eig_Qrows.ptr), static_cast<int>(n_other));
  t->prof.end(s2);
  const int32_t first = sd.n_tasks - sd.n_short;
  // rows per pass: x~ scratch of B x KP floats (IRSPACK_AMD_IALS_EIG_PASS_ROWS: tests shrink it)
  const char *pass_env = std::getenv("IRSPACK_AMD_IALS_EIG_PASS_ROWS");
  const int32_t B = std::min<int32_t>(sd.n_short, pass_env ? std::max(64, std::atoi(pass_env)) : 1 << 20);
  t->eig_xt.alloc(static_cast<size_t>(B) * KP);
  EigShortParams p{};
  p.indices = sd.indices.ptr;
  p.data = sd.data.ptr;
  p.table = t->eig_table.ptr;
  p.lam = t->eig_lam.ptr;
  p.reg = sd.reg.ptr;
  p.xt = t->eig_xt.ptr;
  p.bias = t->cfg.loss_type == IRS_LOSS_IALSPP ? 0.0f : t->cfg.alpha0;
  p.K = static_cast<int32_t>(t->K);
  p.max_cg_steps = max_cg_steps;
  p.warm_start = 1;
  p.err_flag = err_flag;
  // passes alternate between two streams (each with its own x~ scratch): the rotation of one pass
  // (matrix cores, streaming) runs beside the short-row kernel of the next (gather latency)
  const bool two = sd.n_short > B;
  hipStream_t sb = s2;
  float *xt = t->eig_xt.ptr;
  if (two) {
    t->eig_xt2.alloc(static_cast<size_t>(B) * KP);
    if (!t->stream3) {
      IRS_HIP(hipStreamCreateWithFlags(&t->stream3, hipStreamNonBlocking));
      IRS_HIP(hipEventCreateWithFlags(&t->ev_fork3, hipEventDisableTiming));
      IRS_HIP(hipEventCreateWithFlags(&t->ev_join3, hipEventDisableTiming));
    }
    IRS_HIP(hipEventRecord(t->ev_fork3, s2));  // (after the table product)
    IRS_HIP(hipStreamWaitEvent(t->stream3, t->ev_fork3, 0));
  }
  auto rotate = [&](const float *src, const Task *src_rows, float *dst, const Task *dst_rows, const float *M,
                    int n) {
    if (KP == 128)
      hipLaunchKernelGGL((rows_times_matT_kernel<128>), dim3(ceil_div(n, 256)), dim3(256), 0, sb, src,
                         src_rows, dst, dst_rows, M, n);
    else
      hipLaunchKernelGGL((rows_times_matT_kernel<64>), dim3(ceil_div(n, 256)), dim3(256), 0, sb, src,
                         src_rows, dst, dst_rows, M, n);
  };
  const char *kname = cg ? (pidx == 0 ? "ials_short_cg_user" : "ials_short_cg_item")
                         : (pidx == 0 ? "ials_short_cholesky_user" : "ials_short_cholesky_item");
  // the list is longest first: [17..32 entries | 9..16 entries | <= 8 entries]
  const int32_t n32 = sd.n_short - sd.n_short16;
  constexpr bool rows8 = true;  // (the 8-entry forms: four waves per SIMD for the tail of the list)
  const int32_t n16 = sd.n_short - (rows8 ? sd.n_short8 : 0);         // rows before the <= 8 class
  for (int32_t b0 = 0; b0 < sd.n_short; b0 += B) {
    const int32_t m = std::min(B, sd.n_short - b0);
    const Task *tasks = sd.tasks.ptr + first + b0;
    const bool odd = two && ((b0 / B) & 1);
    sb = odd ? t->stream3 : s2;
    xt = odd ? t->eig_xt2.ptr : t->eig_xt.ptr;
    p.tasks = tasks;
    p.n_tasks = m;
    // rows of this pass with 17..32 entries: [0, m2) (one row per wave), the rest have <= 16
    // (four rows per wave, 16 lanes each)
    const int32_t m2 = std::max(0, std::min(b0 + m, n32) - b0);
    const int32_t m3 = std::max(m2, std::min(b0 + m, n16) - b0);  // [m2, m3): 9..16 entries, [m3, m): <= 8
    auto part = [&](auto kernel, int32_t off, int32_t cnt, int rows_per_block, size_t lds) {
      if (cnt <= 0) return;
      EigShortParams q = p;
      q.tasks = tasks + off;
      q.n_tasks = cnt;
      q.xt = xt + static_cast<size_t>(off) * KP;
      hipLaunchKernelGGL(kernel, dim3(ceil_div(cnt, rows_per_block)), dim3(256), lds, sb, q);
    };
    constexpr bool rows16 = true;  // (four rows of <= 16 entries per wave)
    if (cg) {
      t->prof.begin("eig_rotate", sb);
      rotate(target, tasks, xt, nullptr, t->eig_Qrows.ptr, m);  // x~0 = Q^T x0
      t->prof.end(sb);
    }
    t->prof.begin(kname, sb);
    if (cg) {
      if (!rows16) {
        if (KP == 128) part(ials_cg_eig_short_kernel<128>, 0, m, 4, 0);
        else part(ials_cg_eig_short_kernel<64>, 0, m, 4, 0);
      } else if (KP == 128) {
        part(ials_cg_eig16_kernel<128, 32>, 0, m2, 8, 0);   // 17..32 entries: 32 lanes per row
        part(ials_cg_eig16_kernel<128, 16>, m2, m3 - m2, 16, 0);
        part(ials_cg_eig16_kernel<128, 16, 8>, m3, m - m3, 16, 0);
      } else {
        part(ials_cg_eig16_kernel<64, 32>, 0, m2, 8, 0);
        part(ials_cg_eig16_kernel<64, 16>, m2, m3 - m2, 16, 0);
        part(ials_cg_eig16_kernel<64, 16, 8>, m3, m - m3, 16, 0);
      }
    } else {
      const size_t lds1 = 4 * (Chol16Geo<1>::LDS_FLOATS + 32) * sizeof(float);
      const size_t lds2 = 4 * (Chol16Geo<2>::LDS_FLOATS + 64) * sizeof(float);
      // 17..32 entries: the MFMA kernel (one row per wave) by default; the 32-lanes-per-row form of
      // the register kernel is slower for Cholesky (its n x n factorisation is n^2 / 2 lane
      // broadcasts: configs[3] epoch 69 -> 77 ms) although it is the faster one for CG (68 -> 63)
      constexpr bool rows32 = false;
      if (KP == 128) {
        if (rows16 && rows32) part(ials_wb_eig16_kernel<128, 32>, 0, m2, 8, 0);
        else part(ials_wb_short_kernel<128, 2>, 0, m2, 4, lds2);
        if (rows16) {
          part(ials_wb_eig16_kernel<128, 16>, m2, m3 - m2, 16, 0);
          part(ials_wb_eig16_kernel<128, 16, 8>, m3, m - m3, 16, 0);
        } else {
          part(ials_wb_short_kernel<128, 1>, m2, m - m2, 4, lds1);
        }
      } else {
        if (rows16 && rows32) part(ials_wb_eig16_kernel<64, 32>, 0, m2, 8, 0);
        else part(ials_wb_short_kernel<64, 2>, 0, m2, 4, lds2);
        if (rows16) {
          part(ials_wb_eig16_kernel<64, 16>, m2, m3 - m2, 16, 0);
          part(ials_wb_eig16_kernel<64, 16, 8>, m3, m - m3, 16, 0);
        } else {
          part(ials_wb_short_kernel<64, 1>, m2, m - m2, 4, lds1);
        }
      }
    }
    t->prof.end(sb);
    t->prof.begin("eig_rotate", sb);
    rotate(xt, nullptr, target, tasks, t->eig_Qcols.ptr, m);  // x = Q x~
    t->prof.end(sb);
  }
  if (two) {
    IRS_HIP(hipEventRecord(t->ev_join3, t->stream3));
    IRS_HIP(hipStreamWaitEvent(s2, t->ev_join3, 0));
  }
  IRS_HIP(hipGetLastError());
  IRS_HIP(hipEventRecord(t->ev_join, s2));
  IRS_HIP(hipStreamWaitEvent(t->stream, t->ev_join, 0));
  t->eig_last = 1;
  return true;
}

// step_cg (hpp:199-264) at 128 < K <= 256, matrix free (ials_mf_kernels.hpp): the resident classes
// on the trainer's stream, the level-synchronous rows (one pair of launches per product) on the
// second stream beside them.
void launch_mf_cg(irs_ials_trainer *t, Side &sd, const float *other, float *target, int pidx,
                  const irs_ials_solver_config *sc) {
  MfParams p{};
  p.rows = sd.rows_by_len.ptr;
  p.indptr = sd.indptr.ptr;
  p.indices = sd.indices.ptr;
  p.data = sd.data.ptr;
  p.other = other;
  p.target = target;
  p.reg = sd.reg.ptr;
  p.P = t->P[pidx].ptr;
  p.bias = t->cfg.loss_type == IRS_LOSS_IALSPP ? 0.0f : t->cfg.alpha0;  // hpp:190-191
  p.K = static_cast<int32_t>(t->K);
  p.max_cg_steps = sc->max_cg_steps == 0 ? static_cast<int32_t>(t->K)  // hpp:232-234
                                         : static_cast<int32_t>(std::min<uint64_t>(sc->max_cg_steps, 1u << 20));
  p.err_flag = t->err_flag.ptr;
  const int32_t n_long = sd.mf_n_lrows;
  static const bool fork_ok = env_flag("IRSPACK_AMD_MF_FORK", true);  // 0: one stream (A/B, profiling)
  const bool fork = fork_ok && n_long > 0 && sd.mf_class[MF_CLASSES] > sd.mf_class[1];
  hipStream_t ls = t->stream;
  if (fork) {
    if (!t->stream2) {
      IRS_HIP(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
      IRS_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
      IRS_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
    }
    IRS_HIP(hipEventRecord(t->ev_fork, t->stream));
    IRS_HIP(hipStreamWaitEvent(t->stream2, t->ev_fork, 0));
    ls = t->stream2;
  }
  if (n_long > 0) {
    const size_t KP = static_cast<size_t>(t->KP);
    t->mf_vec.alloc(n_long * KP);
    t->mf_xs.alloc(n_long * KP);
    t->mf_rs.alloc(n_long * KP);
    t->mf_partial.alloc(static_cast<size_t>(sd.mf_n_chunks) * 2 * KP);
    t->mf_r2.alloc(n_long);
    t->mf_done.alloc(n_long);
    p.lrows = sd.mf_lrows.ptr;
    p.chunks = sd.mf_chunks.ptr;
    p.n_lrows = n_long;
    p.n_chunks = sd.mf_n_chunks;
    p.vec = t->mf_vec.ptr;
    p.xs = t->mf_xs.ptr;
    p.rs = t->mf_rs.ptr;
    p.partial = t->mf_partial.ptr;
    p.r2 = t->mf_r2.ptr;
    p.done = t->mf_done.ptr;
    t->prof.begin(pidx == 0 ? "ials_long_cg_user" : "ials_long_cg_item", ls);
    auto chain = [&](auto init, auto chunk, auto rowk) {
      hipLaunchKernelGGL(init, dim3(n_long), dim3(256), 0, ls, p);
      // 2 (steps + 1) + 1 launches: 9 at the reference's default of three steps, 2 K + 3 for
      // max_cg_steps = 0 (= K steps, hpp:232-234).  A caller's own, larger count is honoured - rows leave
      // the iteration through the 1e-20 exits (hpp:238, 258) long before - but not blindly: every 64
      // steps the rows' `done` flags come back, and the loop ends when every row has left.
      std::vector<int32_t> done_host;
      for (int step = 0; step <= p.max_cg_steps; step++) {
        hipLaunchKernelGGL(chunk, dim3(sd.mf_n_chunks), dim3(256), 0, ls, p, step == 0 ? 1 : 0);
        hipLaunchKernelGGL(rowk, dim3(n_long), dim3(256), 0, ls, p, step);
        if (step % 64 == 63 && step + 1 <= p.max_cg_steps) {
          done_host.resize(static_cast<size_t>(n_long));
          IRS_HIP(hipMemcpyAsync(done_host.data(), p.done, sizeof(int32_t) * static_cast<size_t>(n_long),
                                 hipMemcpyDeviceToHost, ls));
          IRS_HIP(hipStreamSynchronize(ls));
          if (std::all_of(done_host.begin(), done_host.end(), [](int32_t d) { return d != 0; })) break;
        }
      }
    };
    if (t->T == 12) chain(mf_long_init_kernel<192>, mf_chunk_kernel<192>, mf_row_kernel<192>);
    else chain(mf_long_init_kernel<256>, mf_chunk_kernel<256>, mf_row_kernel<256>);
    t->prof.end(ls);
  }
  t->prof.begin(pidx == 0 ? "ials_solve_cg_user" : "ials_solve_cg_item", t->stream);
  auto resident = [&](auto kernel, int cls, int rows_per_wg, int waves) {
    const int32_t first = sd.mf_class[cls], count = sd.mf_class[cls + 1] - first;
    if (count <= 0) return;
    MfParams q = p;
    q.row_first = first;
    q.n_rows = count;
    hipLaunchKernelGGL(kernel, dim3(ceil_div(count, rows_per_wg)), dim3(64 * waves), 0, t->stream, q);
  };
  // classes 1 .. 4 of MF_CAPS: <= 320 (eight waves on one row), 160 (four), 80 (two rows of two
  // waves), 40 (four rows of one wave)
  if (t->T == 12) {
    resident(mf_cg_rows_kernel<192, 1, 8, MF_J>, 1, 1, 8);
    resident(mf_cg_rows_kernel<192, 1, 4, MF_J>, 2, 1, 4);
    resident(mf_cg_rows_kernel<192, 2, 2, MF_J>, 3, 2, 4);
    resident(mf_cg_rows_kernel<192, 4, 1, MF_J>, 4, 4, 4);
  } else {
    resident(mf_cg_rows_kernel<256, 1, 8, MF_J>, 1, 1, 8);
    resident(mf_cg_rows_kernel<256, 1, 4, MF_J>, 2, 1, 4);
    resident(mf_cg_rows_kernel<256, 2, 2, MF_J>, 3, 2, 4);
    resident(mf_cg_rows_kernel<256, 4, 1, MF_J>, 4, 4, 4);
  }
  t->prof.end(t->stream);
  if (fork) {
    IRS_HIP(hipEventRecord(t->ev_join, ls));
    IRS_HIP(hipStreamWaitEvent(t->stream, t->ev_join, 0));
  }
  IRS_HIP(hipGetLastError());
}

// Solver::step (hpp:664-679) for side `s` over the rows of `sd`, writing `target`.
void launch_solve(irs_ials_trainer *t, Side &sd, const float *other, float *target, int pidx,
                  const irs_ials_solver_config *sc, const float *prior = nullptr) {
  static const char *kNames[2][2][2] = {
      {{"ials_solve_cholesky_user", "ials_solve_cholesky_item"},
       {"ials_split_cholesky_user", "ials_split_cholesky_item"}},
      {{"ials_solve_cg_user", "ials_solve_cg_item"},
       {"ials_split_cg_user", "ials_split_cg_item"}}};
  bool pp_direct = false;
  if (sc->solver_type == IRS_SOLVER_IALSPP) {
    if (prior)  // hpp:659-661
      throw std::invalid_argument("Feature-aware iALS does not support IALSPP.");
    // One block that covers every dimension (subspace dimension >= K, the default at K <= 64):
    // the block step of hpp:436-502 is a Newton step of a quadratic, x - A^-1 (A x - b) with the
    // A and b of step_cholesky (hpp:289-324; the gradient P x + reg x + sum (c (x.v - 1) - bias) v
    // is A x - b).  It runs on the tuned rank-update + Cholesky kernel in the reference's own
    // arithmetic FORM (round 5; RESID kernels, ials_kernels.hpp): the right-hand side is the
    // negative gradient at the current row, built from the predictions v.x like hpp:455-474, the
    // solve yields the step, the row moves by it.  (Rounds 2-4 computed the exact minimiser
    // A^-1 b instead: the same number in exact arithmetic, but with kappa * 2^-24 of forward error
    // where the reference's form has none - at the reference's DEFAULT alpha0 = 0 a row with
    // fewer entries than K was up to 0.19 relative away from float64 where the reference is
    // 3e-3, tests/test_gpu_operating_point.py.)  One sweep is then exactly the reference's; more
    // sweeps repeat it like the reference does.  Like the reference on this path (hpp:495-497)
    // a failed factorisation is not reported.
    pp_direct = t->opt_pp_direct && sc->ialspp_subspace_dimension > 1 &&
                static_cast<uint64_t>(sc->ialspp_subspace_dimension) >= t->K &&
                sc->ialspp_iteration >= 1 && t->T <= 4 && target == t->factor[pidx].ptr;
    if (!pp_direct) {
      launch_ialspp(t, sd, other, target, pidx, sc);
      return;
    }
  }
  if (t->gk()) {
    int32_t *flag = t->err_flag.ptr;
    if (pp_direct) {
      if (!t->pp_llt_sink.ptr) {
        t->pp_llt_sink.alloc(1);
        t->pp_llt_sink.zero(t->stream);
      }
      flag = t->pp_llt_sink.ptr;
    }
    irs_ials_solver_config eff = *sc;
    if (pp_direct) eff.solver_type = IRS_SOLVER_CHOLESKY;
    launch_gk_solve(t, sd, other, target, pidx, &eff, prior, flag);
    return;
  }
  if (sc->solver_type == IRS_SOLVER_CG && t->T > 8 && prior == nullptr && t->opt_mf) {
    launch_mf_cg(t, sd, other, target, pidx, sc);
    return;
  }
  SolveParams p;
  p.prior = prior;
  p.tasks = sd.tasks.ptr;
  p.n_tasks = sd.n_tasks;
  p.split_rows = sd.split.ptr;
  p.n_split = sd.n_split;
  p.indices = sd.indices.ptr;
  p.data = sd.data.ptr;
  p.other = other;
  p.target = target;
  p.reg = sd.reg.ptr;
  p.P_acc = t->P_acc[pidx].ptr;
  p.P_accL = t->P_accL[pidx].ptr;
  p.err_flag = t->err_flag.ptr;
  if (pp_direct) {
    if (!t->pp_llt_sink.ptr) {
      t->pp_llt_sink.alloc(1);
      t->pp_llt_sink.zero(t->stream);
    }
    p.err_flag = t->pp_llt_sink.ptr;
  }
  p.bias = t->cfg.loss_type == IRS_LOSS_IALSPP ? 0.0f : t->cfg.alpha0;  // hpp:190-191
  p.K = static_cast<int32_t>(t->K);
  p.max_cg_steps = sc->max_cg_steps == 0 ? static_cast<int32_t>(t->K)
                                         : static_cast<int32_t>(std::min<uint64_t>(
                                               sc->max_cg_steps, 1u << 20));
  p.warm_start = 1;
  p.zero_row = static_cast<int32_t>(ceil_div(sd.n_other, 8) * 8);
  // (the UNIT kernels address the gathered table with 32-bit byte offsets)
  const bool unit = sd.unit && t->opt_unit && other == t->factor[1 - pidx].ptr &&
                    static_cast<uint64_t>(p.zero_row + 8) * t->KP * sizeof(float) < (uint64_t(1) << 32);
  const bool cg = sc->solver_type == IRS_SOLVER_CG;
  auto pp_gramian_term = [&]() {
    // the Gramian term of every row's gradient at once: (target @ P)[r] = P x_r (P is symmetric; its
    // KP rows play the items of user_scores_kernel); R K^2 multiply-adds, ~1 % of the half-step
    // Only the rows this Side solves, [row_begin, row_end): a rank of a sharded run (or a row chunk of
    // irs_ials_sharded_step) must not repeat the product for the whole side, nor hold [n_rows, KP] for it.
    // resid_finish_rhs reads px at the ABSOLUTE row, hence the base pointer shifted back by row_begin rows.
    const int64_t rb = sd.row_begin, R = sd.row_end - sd.row_begin, KPl = t->KP;
    t->pp_px.alloc(static_cast<size_t>(std::max<int64_t>(R, 1)) * KPl);
    if (R > 0) {
      const int64_t waves = ceil_div(R, 64) * ceil_div(KPl, 64);
      IRS_DISPATCH_T(t->T, {
        t->prof.launch("ials_ialspp_px", user_scores_kernel<16 * TT>, dim3(ceil_div(waves, 4)), dim3(256), 0, t->stream,
                       static_cast<const float *>(target), static_cast<const float *>(t->P[pidx].ptr), rb, R,
                       KPl, t->pp_px.ptr);
      });
    }
    p.px = t->pp_px.ptr - rb * KPl;
  };
  if (pp_direct) pp_gramian_term();
  // CG, short rows: the matrix-free kernel takes the tail of the longest-first task list
  int n_regular = sd.n_tasks;
  // short rows (<= 32 entries) of a side that has enough of them: Cholesky in its low-rank form /
  // CG with a diagonal P, both in the eigenbasis of P (ials_eig_kernels.hpp)
  const bool short_cg_ok = cg && prior == nullptr && sd.n_short > 0 && (t->T <= 4 || t->T == 8) && t->opt_short;
  hipStream_t short_stream = t->stream;
  auto launch_short_cg = [&]() {  // the matrix-free short-row CG kernels over the tail of the task list
    const int n_regular = sd.n_tasks - sd.n_short;
    int n_cu = 0;
    IRS_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, t->device));
    t->prof.begin(pidx == 0 ? "ials_short_cg_user" : "ials_short_cg_item", short_stream);
    IRS_DISPATCH_T8(t->T, {
      constexpr int KPP = 16 * TT;
      const size_t lds = ShortGeo<KPP>::LDS_BYTES;
      auto launch = [&](auto kernel, int first, int count, int nr) {
        if (count <= 0) return;
        IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        const int grid = static_cast<int>(
            std::min<int64_t>(ceil_div(count, SHORT_WAVES * nr), 2 * std::max(n_cu, 1)));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * SHORT_WAVES), lds, short_stream, p,
                           t->P[pidx].ptr, first, count);
      };
      // the list is longest first: [general | 17..32 entries | <= 16 entries]
      if (t->opt_short2) {
        launch(ials_cg_short_kernel<KPP, 32, 1>, n_regular, sd.n_short - sd.n_short16, 1);
        launch(ials_cg_short_kernel<KPP, 16, 2>, sd.n_tasks - sd.n_short16, sd.n_short16, 2);
      } else {
        launch(ials_cg_short_kernel<KPP, 32, 1>, n_regular, sd.n_short - sd.n_short16, 1);
        launch(ials_cg_short_kernel<KPP, 16, 1>, sd.n_tasks - sd.n_short16, sd.n_short16, 1);
      }
    });
    t->prof.end(short_stream);
  };
  const bool eig_cand = prior == nullptr && !pp_direct && other == t->factor[1 - pidx].ptr &&
                        eig_begin(t, sd, pidx, cg);
  bool short_forked = false;
  if (eig_cand) {
    n_regular = sd.n_tasks - sd.n_short;  // (the long rows start now, beside the decomposition)
  } else if (short_cg_ok) {
    // the short-row kernels (latency bound, low issue rate) run BESIDE the kernels of the long rows
    // (matrix-core bound) on the second stream when there are enough of them to matter
    if (sd.n_short >= 65536) {
      if (!t->stream2) {
        IRS_HIP(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
        IRS_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
        IRS_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
      }
      IRS_HIP(hipEventRecord(t->ev_fork, t->stream));
      IRS_HIP(hipStreamWaitEvent(t->stream2, t->ev_fork, 0));
      short_stream = t->stream2;
      short_forked = true;
    }
    launch_short_cg();
    short_stream = t->stream;
    n_regular = sd.n_tasks - sd.n_short;
  }
  // the kernels of the K x K systems over `count` tasks starting at `tasks_begin`
  // rows cut into many chunks: their partial Gramians are summed in groups first (ials_kernels.hpp)
  auto fold_partials = [&](int partial_floats) {
    if (sd.n_fold <= 0) return;
    t->prof.launch("ials_fold_partials", fold_partials_kernel, dim3(sd.n_fold, ceil_div(partial_floats / 4, 256)),
                   dim3(256), 0, t->stream, t->split_partial.ptr, partial_floats, sd.fold.ptr);
  };
  auto launch_dense = [&](const Task *tasks_begin, int count, bool with_split) {
  const int n_regular = count;
  p.tasks = tasks_begin;
  p.n_tasks = count;
  if (!cg && (t->T >= 12 || (t->T == 8 && !t->opt_wave128)) && t->opt_wg16) {
    // 128 < K <= 256, Cholesky: one workgroup per row, 16-row block steps on the matrix cores
    // (ials_wg16_kernels.hpp).  K <= 128 stays on the one-wave-per-row kernel, whose solve is the
    // same block algorithm in one wave (ials_chol16.hpp): ML-20M shape, K = 128: 8.7 ms against
    // 12.5 ms per epoch, because four waves per row repeat the gather's vector work four times.
    IRS_DISPATCH_TW(t->T, {
      using G = Geo<TT>;
      constexpr size_t lds = WgChol<TT>::LDS_FLOATS * sizeof(float);
      t->split_partial.alloc(static_cast<size_t>(std::max(sd.n_slots, 1)) * G::PARTIAL_FLOATS);
      p.partials = t->split_partial.ptr;
      auto launch = [&](auto kernel, int n_items, const char *name) {
        if (n_items <= 0) return;
        IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        t->prof.begin(name, t->stream);
        hipLaunchKernelGGL(kernel, dim3(n_items), dim3(256), lds, t->stream, p);
        t->prof.end(t->stream);
      };
      launch(ials_wg16_cholesky_kernel<TT, 0>, n_regular, kNames[0][0][pidx]);
      if (with_split) fold_partials(G::PARTIAL_FLOATS);
      if (with_split) launch(ials_wg16_cholesky_kernel<TT, 1>, sd.n_split, kNames[0][1][pidx]);
    });
  } else if (t->T == 8 && t->opt_wave128) {
    // 64 < K <= 128: the 36 tiles still fit one wave's 512 registers (144 of them
    // accumulators), so the one-wave-per-task kernel is reused with the MFMA panel Cholesky /
    // a two-rows-per-lane CG (46 KB LDS per wave for the spilled tiles: three waves per CU)
    using G = Geo<8>;
    t->split_partial.alloc(static_cast<size_t>(std::max(sd.n_slots, 1)) * G::PARTIAL_FLOATS);
    p.partials = t->split_partial.ptr;
    if (n_regular > 0) {
      t->prof.begin(kNames[cg][0][pidx], t->stream);
      // (unit confidences: the bf16x3 rank update of ials_kernels.hpp, round 6 also at T = 8 - 216 bf16 matrix
      // instructions of 16 cycles per 32 entries instead of 288 fp32-input ones of 32)
      // The bf16x3 kernels hold ~410-440 registers (one wave per SIMD, like the CG kernels always did); the
      // fp32-input Cholesky kernel holds 236 (two).  Under Cholesky the second wave is worth more than the
      // faster rank update when rows are short - ML-20M user half, 144 entries per task: 4.8 ms fp32-input,
      // 5.3 ms bf16x3; item half, 544 per task: 3.1 -> 2.5 ms - so the choice goes by the mean task length.
      // (`unit` also requires a gathered table below 4 GB - the fp32-input UNIT kernels address it with
      // 32-bit byte offsets; the T = 8 bf16x3 gather uses 64-bit row addresses and has no such limit)
      const bool x3_any = sd.unit && t->opt_unit && other == t->factor[1 - pidx].ptr && t->opt_bf16x3;
      const bool x3 = x3_any && (cg || sd.long_entries >= int64_t(320) * std::max(n_regular, 1));
      if (cg && x3)
        hipLaunchKernelGGL((ials_solve_kernel<8, 1, 0, true, true>), dim3(n_regular), dim3(64), 0, t->stream, p);
      else if (cg && unit)
        hipLaunchKernelGGL((ials_solve_kernel<8, 1, 0, true>), dim3(n_regular), dim3(64), 0, t->stream, p);
      else if (cg)
        hipLaunchKernelGGL((ials_solve_kernel<8, 1, 0>), dim3(n_regular), dim3(64), 0, t->stream, p);
      else if (x3)
        hipLaunchKernelGGL((ials_solve_kernel<8, 0, 0, true, true>), dim3(n_regular), dim3(64), 0, t->stream, p);
      else if (unit)
        hipLaunchKernelGGL((ials_solve_kernel<8, 0, 0, true>), dim3(n_regular), dim3(64), 0, t->stream, p);
      else
        hipLaunchKernelGGL((ials_solve_kernel<8, 0, 0>), dim3(n_regular), dim3(64), 0, t->stream, p);
      t->prof.end(t->stream);
    }
    if (with_split && sd.n_split > 0) {
      fold_partials(G::PARTIAL_FLOATS);
      t->prof.begin(kNames[cg][1][pidx], t->stream);
      if (cg && sd.unit && t->opt_unit && other == t->factor[1 - pidx].ptr && t->opt_bf16x3)  // (lower-form partials)
        hipLaunchKernelGGL((ials_solve_kernel<8, 1, 1, false, true>), dim3(sd.n_split), dim3(64), 0, t->stream, p);
      else if (cg)
        hipLaunchKernelGGL((ials_solve_kernel<8, 1, 1>), dim3(sd.n_split), dim3(64), 0, t->stream, p);
      else
        hipLaunchKernelGGL((ials_solve_kernel<8, 0, 1>), dim3(sd.n_split), dim3(64), 0, t->stream, p);
      t->prof.end(t->stream);
    }
  } else if (t->T <= 4) {
    IRS_DISPATCH_T(t->T, {
      using G = Geo<TT>;
      t->split_partial.alloc(static_cast<size_t>(std::max(sd.n_slots, 1)) * G::PARTIAL_FLOATS);
      p.partials = t->split_partial.ptr;
      if (n_regular > 0) {
        const dim3 grid(ceil_div(n_regular, SOLVE_WAVES)), block(64 * SOLVE_WAVES);
        const char *name = kNames[cg][0][pidx];
        if (pp_direct && unit && t->opt_bf16x3 && TT == 4)  // (the gradient form on the bf16x3 rank update, round 6)
          t->prof.launch(name, ials_solve_kernel<4, 0, 0, true, true, true>, grid, block, 0, t->stream, p);
        else if (pp_direct && unit)
          t->prof.launch(name, ials_solve_kernel<TT, 0, 0, true, false, true>, grid, block, 0, t->stream, p);
        else if (pp_direct)
          t->prof.launch(name, ials_solve_kernel<TT, 0, 0, false, false, true>, grid, block, 0, t->stream, p);
        else if (cg && unit && t->opt_bf16x3 && TT == 4)
          t->prof.launch(name, ials_solve_kernel<4, 1, 0, true, true>, grid, block, 0, t->stream, p);
        else if (cg && unit)
          t->prof.launch(name, ials_solve_kernel<TT, 1, 0, true>, grid, block, 0, t->stream, p);
        else if (cg)
          t->prof.launch(name, ials_solve_kernel<TT, 1, 0>, grid, block, 0, t->stream, p);
        else if (unit && t->opt_bf16x3 && TT == 4)
          t->prof.launch(name, ials_solve_kernel<4, 0, 0, true, true>, grid, block, 0, t->stream, p);
        else if (unit)
          t->prof.launch(name, ials_solve_kernel<TT, 0, 0, true>, grid, block, 0, t->stream, p);
        else
          t->prof.launch(name, ials_solve_kernel<TT, 0, 0>, grid, block, 0, t->stream, p);
      }
      if (with_split && sd.n_split > 0) {
        fold_partials(G::PARTIAL_FLOATS);
        const dim3 grid(ceil_div(sd.n_split, SOLVE_WAVES)), block(64 * SOLVE_WAVES);
        if (pp_direct)
          t->prof.launch(kNames[cg][1][pidx], ials_solve_kernel<TT, 0, 1, false, false, true>, grid, block, 0,
                         t->stream, p);
        else if (cg && unit && t->opt_bf16x3 && TT == 4)  // (the bf16x3 chunks left lower-form partials)
          t->prof.launch(kNames[cg][1][pidx], ials_solve_kernel<4, 1, 1, false, true>, grid, block, 0, t->stream, p);
        else if (cg)
          t->prof.launch(kNames[cg][1][pidx], ials_solve_kernel<TT, 1, 1>, grid, block, 0, t->stream, p);
        else
          t->prof.launch(kNames[cg][1][pidx], ials_solve_kernel<TT, 0, 1>, grid, block, 0, t->stream, p);
      }
    });
  } else {
    IRS_DISPATCH_TW(t->T, {
      using G = Geo<TT>;
      constexpr size_t lds = WgGeo<TT>::LDS_FLOATS * sizeof(float);
      t->split_partial.alloc(static_cast<size_t>(std::max(sd.n_slots, 1)) * G::PARTIAL_FLOATS);
      p.partials = t->split_partial.ptr;
      auto launch = [&](auto kernel, int n_items, const char *name) {
        IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    static_cast<int>(lds)));
        t->prof.begin(name, t->stream);
        hipLaunchKernelGGL(kernel, dim3(n_items), dim3(256), lds, t->stream, p);
        t->prof.end(t->stream);
      };
      if (n_regular > 0) {
        if (cg)
          launch(ials_wg_solve_kernel<TT, 1, 0>, n_regular, kNames[1][0][pidx]);
        else
          launch(ials_wg_solve_kernel<TT, 0, 0>, n_regular, kNames[0][0][pidx]);
      }
      if (with_split && sd.n_split > 0) {
        fold_partials(G::PARTIAL_FLOATS);
        if (cg)
          launch(ials_wg_solve_kernel<TT, 1, 1>, sd.n_split, kNames[1][1][pidx]);
        else
          launch(ials_wg_solve_kernel<TT, 0, 1>, sd.n_split, kNames[0][1][pidx]);
      }
    });
  }
  IRS_HIP(hipGetLastError());
  };  // launch_dense
  launch_dense(sd.tasks.ptr, n_regular, true);
  if (pp_direct) {  // the further sweeps of hpp:516-530, each a step from where the last one ended
    for (uint64_t it = 1; it < sc->ialspp_iteration; it++) {
      pp_gramian_term();
      launch_dense(sd.tasks.ptr, n_regular, true);
    }
  }
  if (short_forked) {
    IRS_HIP(hipEventRecord(t->ev_join, t->stream2));
    IRS_HIP(hipStreamWaitEvent(t->stream, t->ev_join, 0));
  }
  if (eig_cand && !eig_finish(t, sd, other, target, pidx, cg, p.max_cg_steps, p.err_flag)) {
    // the eigenbasis path declined (ill-conditioned P + reg I): the short rows through the usual kernels
    p.tasks = sd.tasks.ptr;
    p.n_tasks = sd.n_tasks;
    if (short_cg_ok) launch_short_cg();
    else launch_dense(sd.tasks.ptr + n_regular, sd.n_short, false);
  }
}

// Raise what the reference would have thrown from inside the worker threads.
void sync_and_check(irs_ials_trainer *t) {
  int32_t flag = 0;
  IRS_HIP(hipMemcpyAsync(&flag, t->err_flag.ptr, sizeof(flag), hipMemcpyDeviceToHost, t->stream));
  IRS_HIP(hipStreamSynchronize(t->stream));
  if (flag) {
    t->err_flag.zero(t->stream);
    IRS_HIP(hipStreamSynchronize(t->stream));
    if (flag & 1) throw std::runtime_error("Cholesky decomposition failed.");  // hpp:318
    if (flag & 2) throw std::runtime_error("Cholesky solve failed.");          // hpp:322
    if (flag & 8)  // comm.hip: a peer's rows or Gramian did not arrive (IRSPACK_AMD_PEER_TIMEOUT_S)
      throw std::runtime_error("The sharded step timed out waiting for a peer's stores.");
    throw std::runtime_error(
        "Conjugate-gradient solver encountered a singular system.");  // hpp:252-253
  }
}

void require_X(irs_ials_trainer *t) {
  if (!t->has_X)
    throw std::runtime_error(
        "This IALSTrainer was restored from factors and holds no interaction matrix.");
}

void full_gramian(irs_ials_trainer *t, int side) {
  t->gram_prefetched[side] = false;
  // side 0 (user solve) sums item rows; side 1 sums user rows.
  const int other = 1 - side;
  const bool fused = t->T <= 4;  // (the one-wave-per-row family: reduce and finish in one launch)
  launch_partial_gramian(t, other, 0, t->rows_of(other), side, fused);
  if (!fused) launch_finish_gramian(t, side);
}

}  // namespace

extern "C" {

irs_status irs_ials_scores_device_(irs_ials_trainer *t, int64_t begin, int64_t end,
                                   float *device_out, void **stream_out, int32_t *device_index);
irs_status irs_ials_factors_device_(irs_ials_trainer *t, const float **user, const float **item,
                                    int32_t *KP, int64_t *n_users, int64_t *n_items,
                                    void **stream_out, int32_t *device_index);
irs_status irs_ials_scores_prefix_device_(irs_ials_trainer *t, int64_t begin, int64_t end,
                                          int64_t n_prefix, const float *item_rows,
                                          const float *user_rows, float *device_out);

const char *irs_last_error(void) { return irs::last_error().c_str(); }
int32_t irs_abi_version(void) { return IRS_ABI_VERSION; }
int32_t irs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

irs_status irs_ials_create(const irs_ials_model_config *config, int64_t n_users,
                           int64_t n_items, const int64_t *indptr, const int32_t *indices,
                           const float *data, int32_t device, const irs_ials_shard *shard,
                           irs_ials_trainer **out) {
  return guard([&] {
    check_arg(config && out, "null argument.");
    validate_config(*config);
    const bool timing = std::getenv("IRSPACK_AMD_IALS_TIMING") != nullptr;
    auto tm0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
      if (!timing) return;
      const auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "ials create phase %-16s %8.2f ms\n", what,
              std::chrono::duration<double, std::milli>(t1 - tm0).count());
      tm0 = t1;
    };
    // the initial factors (a sequential libstdc++ random stream, ~0.4 s for 10 M values) and the
    // transposed matrix are prepared on two host threads while this one sets the device up
    // ... and uploaded by that same thread as soon as the device buffers exist, beside the preparation
    // and the uploads of the two CSR orientations (5.6 GB of factors at the 10 M x 1 M shape: 0.26 s
    // that used to follow everything else)
    RawVector<float> init_draw;
    std::promise<irs_ials_trainer *> buffers_ready;  // null: the construction failed before the allocation
    std::shared_future<irs_ials_trainer *> buffers = buffers_ready.get_future().share();
    std::exception_ptr draw_error;
    std::thread draw_thread([&, buffers] {
      try {
        init_draw = draw_factor(config->init_stdev, config->random_seed, static_cast<int64_t>(config->K),
                                std::max(n_users, n_items));
        irs_ials_trainer *tr = buffers.get();
        if (!tr) return;
        IRS_HIP(hipSetDevice(device));
        hipStream_t us = nullptr;
        IRS_HIP(hipStreamCreateWithFlags(&us, hipStreamNonBlocking));
        struct StreamGuard {
          hipStream_t s;
          ~StreamGuard() { (void)hipStreamDestroy(s); }
        } guard{us};
        upload_factor(tr, 0, init_draw.data(), us);  // hpp:718-719: both sides from the same seed
        upload_factor(tr, 1, init_draw.data(), us);
      } catch (...) {
        draw_error = std::current_exception();
      }
    });
    struct Joiner {
      std::thread &t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } draw_join{draw_thread};
    struct ReleaseWaiter {  // (declared AFTER the joiner: runs first, so that a throw below cannot leave the
      std::promise<irs_ials_trainer *> &p;  // draw thread waiting for buffers that will never come)
      bool done = false;
      ~ReleaseWaiter() {
        if (!done) p.set_value(nullptr);
      }
    } release{buffers_ready};
    // A rank of a sharded run prepares only its own user rows of X and item rows of X^T (the
    // ranks together do the work once, not once each); the unsharded trainer keeps everything.
    const irs_ials_shard sh = shard ? *shard : irs_ials_shard{0, n_users, 0, n_items};
    check_arg(0 <= sh.user_begin && sh.user_begin <= sh.user_end && sh.user_end <= n_users &&
                  0 <= sh.item_begin && sh.item_begin <= sh.item_end && sh.item_end <= n_items,
              "shard out of range.");
    const bool whole = sh.user_begin == 0 && sh.user_end == n_users && sh.item_begin == 0 &&
                       sh.item_end == n_items;
    HostCsr X = whole ? host_csr(n_users, n_items, indptr, indices, data, /*view=*/true)
                      : host_csr_rows(n_users, n_items, indptr, indices, data, sh.user_begin,
                                      sh.user_end);
    mark("copy + validate");
    // X^T (hpp:713).  An unsharded trainer uploads X once and transposes it ON THE DEVICE (round 5:
    // device_sort.hip - a stable radix sort of the entry numbers by column + one gather; the host's
    // 16-thread counting sort and the second 80 MB upload were ~15 of the 34 ms of an ML-20M
    // construction); a rank of a sharded run keeps the host pass over its own columns.
    // IRSPACK_AMD_IALS_HOST_TRANSPOSE=1: the host path for the unsharded trainer too (tests: the two
    // must give the same X^T, entry for entry).
    const bool device_transpose = whole && !env_flag("IRSPACK_AMD_IALS_HOST_TRANSPOSE", false) &&
                                  X.indptr[n_users] > 0 && n_items > 0;
    HostCsr Xt;
    std::exception_ptr transpose_error;
    std::thread transpose_thread([&] {
      try {
        if (device_transpose) return;
        Xt = whole ? transpose(X)
                   : transpose_cols(n_users, n_items, indptr, indices, data, sh.item_begin,
                                    sh.item_end);
      } catch (...) {
        transpose_error = std::current_exception();
      }
    });
    Joiner transpose_join{transpose_thread};
    require_device(device);
    auto t = std::make_unique<irs_ials_trainer>();
    // (destroyed before `t`: the upload must not outlive its buffers.  It is also destroyed before `release`
    // above, so when alloc_common throws - hipMalloc out of memory at the 10 M x 1 M shape - it has to
    // fulfil the promise ITSELF before joining: the draw thread is blocked on it.)
    struct ReleaseThenJoin {
      std::thread &t;
      ReleaseWaiter &r;
      ~ReleaseThenJoin() {
        if (!r.done) {
          r.p.set_value(nullptr);
          r.done = true;
        }
        if (t.joinable()) t.join();
      }
    } draw_join_before_trainer{draw_thread, release};
    t->cfg = *config;
    t->K = static_cast<int64_t>(config->K);
    t->n_users = n_users;
    t->n_items = n_items;
    t->device = device;
    t->shard = sh;
    t->whole = whole;
    // (test hook: an allocation failure here must come back as IRS_RUNTIME_ERROR, not hang the draw thread)
    if (env_flag("IRSPACK_AMD_TEST_FAIL_ALLOC", false)) throw std::runtime_error("injected allocation failure.");
    alloc_common(t.get());
    buffers_ready.set_value(t.get());
    release.done = true;
    mark("device alloc");
    if (device_transpose) {  // X on the device first: the item thread transposes it there
      t->side[0].upload_entries(X, t->stream);
      t->side[0].upload_indptr(X, t->stream);
      mark("upload X");
    }
    // the two orientations are prepared and uploaded side by side (host preparation of one
    // overlaps the copies of the other)
    std::exception_ptr item_error;
    std::thread item_thread([&] {
      try {
        transpose_thread.join();
        if (transpose_error) std::rethrow_exception(transpose_error);
        IRS_HIP(hipSetDevice(device));
        if (device_transpose) {
          const int64_t nnz = X.indptr[n_users];
          hipStream_t ts = nullptr;
          IRS_HIP(hipStreamCreateWithFlags(&ts, hipStreamNonBlocking));
          struct StreamGuard {
            hipStream_t s;
            ~StreamGuard() { (void)hipStreamDestroy(s); }
          } guard{ts};
          Side &s0 = t->side[0], &s1 = t->side[1];
          const auto tt0 = std::chrono::steady_clock::now();
          s1.alloc_entries(static_cast<size_t>(nnz), ts);
          std::vector<int32_t> count;
          DeviceBuffer<char> scratch;
          const bool values = !(X.flags_known && X.unit);
          transpose_csr_device(s0.indptr.ptr, s0.indices.ptr, values ? s0.data.ptr : nullptr, n_users, n_items, nnz,
                               s1.indices.ptr, values ? s1.data.ptr : nullptr, count, scratch, ts);
          if (!values) {
            IRS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s1.data.ptr), 0x3f800000,
                                      static_cast<size_t>(nnz), ts));
            IRS_HIP(hipStreamSynchronize(ts));
          }
          Xt.rows = n_items;
          Xt.cols = n_users;
          Xt.flags_known = X.flags_known;
          Xt.unit = X.unit;
          Xt.positive = X.positive;
          Xt.indptr.assign(static_cast<size_t>(n_items) + 1, 0);
          for (int64_t c = 0; c < n_items; c++) Xt.indptr[c + 1] = Xt.indptr[c] + count[c];
          check_arg(Xt.indptr[n_items] == nnz, "internal: the device transpose lost entries.");
          if (timing)
            fprintf(stderr, "ials create (item thread) device transpose %8.2f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tt0).count());
          if (!X.flags_known) {  // (host_csr always classifies; kept for callers that do not)
            Xt.flags_known = true;
            Xt.unit = false;
            Xt.positive = false;
          }
        }
        const auto tb0 = std::chrono::steady_clock::now();
        t->side[1].build(Xt, t->shard.item_begin, t->shard.item_end, t->cfg, t->stream);
        if (timing)
          fprintf(stderr, "ials create (item thread) side 1 build     %8.2f ms\n",
                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
      } catch (...) {
        item_error = std::current_exception();
      }
    });
    Joiner item_join{item_thread};
    t->side[0].build(X, t->shard.user_begin, t->shard.user_end, t->cfg, t->stream);
    mark("side 0 build");
    item_thread.join();
    if (item_error) std::rethrow_exception(item_error);
    mark("both sides");
    if (const char *ce = std::getenv("IRSPACK_AMD_SHARD_CHUNKS")) {
      // row chunks of the shard for irs_ials_sharded_step: chunk c of a side is solved while chunk
      // c - 1 is on the wire.  Same kernels on shorter task lists; the CSR arrays are shared.
      const int C = std::max(1, std::min(16, std::atoi(ce)));
      for (int w = 0; w < 2 && C > 1; w++) {
        const int64_t rb = w == 0 ? t->shard.user_begin : t->shard.item_begin;
        const int64_t re = w == 0 ? t->shard.user_end : t->shard.item_end;
        for (int c = 0; c < C; c++) {
          auto sd = std::make_unique<Side>();
          sd->build(w == 0 ? X : Xt, rb + (re - rb) * c / C, rb + (re - rb) * (c + 1) / C, t->cfg, t->stream,
                    &t->side[w]);
          t->chunk_side[w].push_back(std::move(sd));
        }
      }
      mark("row chunks");
    }
    t->has_X = true;
    draw_thread.join();
    if (draw_error) std::rethrow_exception(draw_error);
    mark("draw + factors");
    {  // the host staging goes back to the kernel on a thread of its own (munmap of ~0.3 GB - and of the
       // 5 GB of initial factors at the 10 M x 1 M shape: 0.4 s that the caller used to wait for)
      struct Junk {
        HostCsr a, b;
        RawVector<float> draw;
      };
      auto *junk = new Junk();
      junk->a.indices.swap(X.indices);
      junk->a.data.swap(X.data);
      junk->b.indices.swap(Xt.indices);
      junk->b.data.swap(Xt.data);
      junk->draw.swap(init_draw);
      std::thread([junk] { delete junk; }).detach();
    }
    mark("hand-off");
    *out = t.release();
  });
}

irs_status irs_ials_create_from_factors(const irs_ials_model_config *config, int64_t n_users,
                                        int64_t n_items, const float *user, const float *item,
                                        int32_t device, irs_ials_trainer **out) {
  return guard([&] {
    check_arg(config && out, "null argument.");
    validate_config(*config);
    require_device(device);
    auto t = std::make_unique<irs_ials_trainer>();
    t->cfg = *config;
    t->K = static_cast<int64_t>(config->K);
    t->n_users = n_users;
    t->n_items = n_items;
    t->device = device;
    t->shard = irs_ials_shard{0, n_users, 0, n_items};
    alloc_common(t.get());
    upload_factor(t.get(), 0, user);
    upload_factor(t.get(), 1, item);
    full_gramian(t.get(), 0);  // hpp:754-755
    full_gramian(t.get(), 1);
    IRS_HIP(hipStreamSynchronize(t->stream));
    *out = t.release();
  });
}

irs_status irs_ials_destroy(irs_ials_trainer *t) {
  return guard([&] {
    if (t) {
      (void)hipSetDevice(t->device);
      t->prof.clear();
      if (t->stream2) {
        (void)hipStreamSynchronize(t->stream2);
        (void)hipStreamDestroy(t->stream2);
        (void)hipEventDestroy(t->ev_fork);
        (void)hipEventDestroy(t->ev_join);
      }
      if (t->stream3) {
        (void)hipStreamSynchronize(t->stream3);
        (void)hipStreamDestroy(t->stream3);
        (void)hipEventDestroy(t->ev_fork3);
        (void)hipEventDestroy(t->ev_join3);
      }
      if (t->eig_resident) (void)hipHostFree(t->eig_resident);
      delete t;
    }
  });
}

irs_status irs_ials_set_stream(irs_ials_trainer *t, void *hip_stream) {
  return guard([&] {
    check_arg(t != nullptr, "null trainer.");
    IRS_HIP(hipSetDevice(t->device));
    IRS_HIP(hipStreamSynchronize(t->stream));
    t->stream = static_cast<hipStream_t>(hip_stream);
  });
}

irs_status irs_ials_device_buffer(irs_ials_trainer *t, int32_t which, void **device_ptr,
                                  int64_t *rows, int64_t *ld) {
  return guard([&] {
    check_arg(t && device_ptr && rows && ld, "null argument.");
    check_arg(which >= 0 && which <= 3, "which must be in 0..3.");
    if (which < 2) {
      *device_ptr = t->factor[which].ptr;
      *rows = ceil_div(t->rows_of(which), 8) * 8;  // allocated rows (zero padding at the end)
    } else {
      *device_ptr = t->P_raw[which - 2].ptr;
      *rows = t->KP;
    }
    *ld = t->KP;
  });
}

irs_status irs_ials_copy_rows_async(irs_ials_trainer *t, int32_t which, int64_t row_begin,
                                    int64_t row_end, void *ext, int32_t to_ext) {
  return guard([&] {
    check_arg(t && ext, "null argument.");
    check_arg(which >= 0 && which <= 3, "which must be in 0..3.");
    const int64_t rows = which < 2 ? t->rows_of(which) : t->KP;
    check_arg(0 <= row_begin && row_begin <= row_end && row_end <= rows, "row range out of bounds.");
    IRS_HIP(hipSetDevice(t->device));
    float *base = which < 2 ? t->factor[which].ptr : t->P_raw[which - 2].ptr;
    float *mine = base + row_begin * t->KP;
    const size_t bytes = static_cast<size_t>(row_end - row_begin) * t->KP * sizeof(float);
    if (bytes == 0) return;
    if (!to_ext) t->gram_prefetched[0] = t->gram_prefetched[1] = false;
    IRS_HIP(hipMemcpyAsync(to_ext ? ext : static_cast<void *>(mine),
                           to_ext ? static_cast<const void *>(mine) : ext, bytes,
                           hipMemcpyDeviceToDevice, t->stream));
  });
}

irs_status irs_ials_partial_gramian_async(irs_ials_trainer *t, int32_t side) {
  return guard([&] {
    check_arg(t && (side == 0 || side == 1), "bad argument.");
    IRS_HIP(hipSetDevice(t->device));
    const int other = 1 - side;
    const int64_t rb = other == 0 ? t->shard.user_begin : t->shard.item_begin;
    const int64_t re = other == 0 ? t->shard.user_end : t->shard.item_end;
    t->gram_prefetched[side] = false;
    launch_partial_gramian(t, other, rb, re, side);
  });
}

irs_status irs_ials_finish_gramian_async(irs_ials_trainer *t, int32_t side) {
  return guard([&] {
    check_arg(t && (side == 0 || side == 1), "bad argument.");
    IRS_HIP(hipSetDevice(t->device));
    launch_finish_gramian(t, side);
  });
}

irs_status irs_ials_gramian_async(irs_ials_trainer *t, int32_t side) {
  return guard([&] {
    check_arg(t && (side == 0 || side == 1), "bad argument.");
    const int other = 1 - side;
    const int64_t rb = other == 0 ? t->shard.user_begin : t->shard.item_begin;
    const int64_t re = other == 0 ? t->shard.user_end : t->shard.item_end;
    check_arg(rb == 0 && re == t->rows_of(other),
              "irs_ials_gramian_async needs every row of the other side on this trainer; a shard sums its "
              "own rows (irs_ials_partial_gramian_async), all-reduces, then finishes.");
    IRS_HIP(hipSetDevice(t->device));
    full_gramian(t, side);
  });
}

irs_status irs_ials_half_step_async(irs_ials_trainer *t, int32_t side,
                                    const irs_ials_solver_config *sc) {
  return guard([&] {
    check_arg(t && (side == 0 || side == 1), "bad argument.");
    check_solver(sc);
    require_X(t);
    IRS_HIP(hipSetDevice(t->device));
    t->gram_prefetched[1 - side] = false;  // (the rows its prefetched Gramian was summed from change)
    launch_solve(t, t->side[side], t->factor[1 - side].ptr, t->factor[side].ptr, side, sc,
                 t->has_prior[side] ? t->prior[side].ptr : nullptr);
  });
}

irs_status irs_ials_synchronize(irs_ials_trainer *t) {
  return guard([&] {
    check_arg(t != nullptr, "null trainer.");
    IRS_HIP(hipSetDevice(t->device));
    sync_and_check(t);
  });
}

// IALSTrainer::step, hpp:784-788
irs_status irs_ials_step(irs_ials_trainer *t, const irs_ials_solver_config *sc) {
  return guard([&] {
    check_arg(t != nullptr, "null trainer.");
    check_solver(sc);
    require_X(t);
    IRS_HIP(hipSetDevice(t->device));
    check_arg(t->shard.user_begin == 0 && t->shard.user_end == t->n_users &&
                  t->shard.item_begin == 0 && t->shard.item_end == t->n_items,
              "irs_ials_step needs an unsharded trainer; sharded runs drive the half "
              "steps from the host loop.");
    full_gramian(t, 0);
    launch_solve(t, t->side[0], t->factor[1].ptr, t->factor[0].ptr, 0, sc);
    full_gramian(t, 1);
    launch_solve(t, t->side[1], t->factor[0].ptr, t->factor[1].ptr, 1, sc);
    sync_and_check(t);
  });
}

// ---------------------------------------------------------------- row-sharded epoch (comm.hip)
irs_status irs_comm_export(irs_comm *c, irs_ials_trainer *t, void *handle256) {
  return guard([&] {
    check_arg(c && t && handle256, "null argument.");
    check_arg(t->device == c->device, "trainer and communicator live on different devices.");
    comm_export(c, t->factor[0].ptr, t->factor[1].ptr, t->KP, handle256);
  });
}

irs_status irs_comm_attach(irs_comm *c, irs_ials_trainer *t, const void *handles) {
  return guard([&] {
    check_arg(c && t && handles, "null argument.");
    check_arg(t->device == c->device, "trainer and communicator live on different devices.");
    comm_attach(c, t->factor[0].ptr, t->factor[1].ptr, t->KP, handles);
  });
}

// IALSTrainer::step (hpp:784-788) over the ranks of `comm`, rows sharded (sharding.py's loop in
// one call).  Per half-epoch: partial Gramian of the rank's own rows of the other side ->
// all-reduce (K x K, trainer stream) -> finish -> solve the rank's rows -> exchange of the solved
// rows (the communicator's own stream; comm.hip: in-place all-gather / grouped broadcasts / one
// send-receive group over the mesh / peer stores - no staging copy in any of them).  While the rows
// travel, the partial Gramian of the NEXT half-epoch - it needs only the rows this rank has just
// solved - and its all-reduce are already running on the trainer's stream.
//
// Errors: everything that can be checked is checked BEFORE the first collective (a rank that
// throws while the others wait inside a collective would hang the job); what the solves report on
// the device (hpp:317-323, 250-254: one rank's rows) is all-reduced at the end, so every rank
// raises the same exception.
irs_status irs_ials_sharded_step(irs_ials_trainer *t, const irs_ials_solver_config *sc, irs_comm *c,
                                 const int64_t *user_bounds, const int64_t *item_bounds) {
  return guard([&] {
    check_arg(t && c && user_bounds && item_bounds, "null argument.");
    check_solver(sc);
    require_X(t);
    check_arg(t->device == c->device, "trainer and communicator live on different devices.");
    check_arg(!t->has_prior[0] && !t->has_prior[1] && t->n_feat[0] == 0 && t->n_feat[1] == 0,
              "the sharded step does not cover the feature-aware model.");
    const int64_t *bounds[2] = {user_bounds, item_bounds};
    for (int w = 0; w < 2; w++) {
      const int64_t n = t->rows_of(w);
      check_arg(bounds[w][0] == 0 && bounds[w][c->world] == n, "row bounds must cover every row.");
      for (int r = 0; r < c->world; r++) check_arg(bounds[w][r] <= bounds[w][r + 1], "row bounds must not decrease.");
    }
    check_arg(user_bounds[c->rank] == t->shard.user_begin && user_bounds[c->rank + 1] == t->shard.user_end &&
                  item_bounds[c->rank] == t->shard.item_begin && item_bounds[c->rank + 1] == t->shard.item_end,
              "the trainer was created for another shard than row_bounds[rank].");
    check_arg(!(c->peer_rows() || c->peer_gram()) || (c->attached && c->KP == t->KP),
              "peer stores need irs_comm_export + irs_comm_attach on this trainer before the first step.");
    IRS_HIP(hipSetDevice(t->device));
    const size_t KP = static_cast<size_t>(t->KP);
    struct Unwind {  // a throw from here on leaves no half-made prefetch behind
      irs_ials_trainer *t;
      bool armed = true;
      ~Unwind() {
        if (armed) t->gram_prefetched[0] = t->gram_prefetched[1] = false;
      }
    } unwind{t};
    auto reduce_gramian = [&](int side) {  // own rows of the other side, summed over the ranks
      const int other = 1 - side;
      launch_partial_gramian(t, other, other == 0 ? t->shard.user_begin : t->shard.item_begin,
                             other == 0 ? t->shard.user_end : t->shard.item_end, side);
      comm_allreduce(c, t->P_raw[side].ptr, KP * KP, t->stream);
    };
    // chunk k of n_chunks of every rank's rows (n_chunks == 1: the whole shards): rank r's chunk k =
    // rows [b_r + len_r k / C, b_r + len_r (k + 1) / C), the formula the chunk task lists were cut with
    auto exchange_rows = [&](int side, int k, int n_chunks) {
      const int64_t *b = bounds[side];
      int64_t lo[COMM_MAX_WORLD], hi[COMM_MAX_WORLD];
      for (int r = 0; r < c->world; r++) {
        const int64_t len = b[r + 1] - b[r];
        lo[r] = b[r] + len * k / n_chunks;
        hi[r] = b[r] + len * (k + 1) / n_chunks;
      }
      comm_exchange_rows(c, side, t->factor[side].ptr, KP, lo, hi, t->rows_of(side), t->stream, n_chunks == 1);
    };
    for (int side = 0; side < 2; side++) {
      if (!t->gram_prefetched[side]) reduce_gramian(side);
      t->gram_prefetched[side] = false;
      launch_finish_gramian(t, side);
      const int n_chunks = static_cast<int>(t->chunk_side[side].size());
      if (n_chunks > 1) {
        for (int k = 0; k < n_chunks; k++) {  // chunk k travels while chunk k + 1 is solved
          launch_solve(t, *t->chunk_side[side][k], t->factor[1 - side].ptr, t->factor[side].ptr, side, sc);
          exchange_rows(side, k, n_chunks);
        }
      } else {
        launch_solve(t, t->side[side], t->factor[1 - side].ptr, t->factor[side].ptr, side, sc);
        exchange_rows(side, 0, 1);
      }
      // the next half-epoch's Gramian (side 1 now, side 0 of the next call) from the rows just solved
      reduce_gramian(1 - side);
      t->gram_prefetched[1 - side] = true;
      // its solve gathers every row of `side`: wait for them
      IRS_HIP(hipStreamWaitEvent(t->stream, c->ev_rows, 0));
    }
    comm_allreduce_flag(c, t->err_flag.ptr, t->stream);
    IRS_HIP(hipStreamSynchronize(c->stream));
    sync_and_check(t);
    unwind.armed = false;
  });
}

irs_status irs_ials_get_factor(irs_ials_trainer *t, int32_t which, float *out) {
  return guard([&] {
    check_arg(t && out && (which == 0 || which == 1), "bad argument.");
    IRS_HIP(hipSetDevice(t->device));
    download_factor(t, t->factor[which].ptr, t->rows_of(which), out);
  });
}

irs_status irs_ials_set_factor(irs_ials_trainer *t, int32_t which, const float *in,
                               int64_t rows, int64_t cols) {
  return guard([&] {
    check_arg(t && in && (which == 0 || which == 1), "bad argument.");
    check_arg(rows == t->rows_of(which) && cols == t->K,
              "factor matrix shape does not match the trainer.");
    IRS_HIP(hipSetDevice(t->device));
    upload_factor(t, which, in);
  });
}

irs_status irs_ials_user_scores(irs_ials_trainer *t, int64_t begin, int64_t end,
                                const irs_ials_solver_config *sc, float *out) {
  return guard([&] {
    check_arg(t && sc, "null argument.");
    check_arg(sc->n_threads > 0, "n_threads must be strictly positive.");  // hpp:944-951
    check_arg(end >= begin, "userblock_end must be greater than or equal to userblock_begin");
    check_arg(begin >= 0, "userblock_begin must be non-negative");
    check_arg(t->n_users >= end, "userblock_end must be smaller than or equal to n_users");
    const int64_t m = end - begin;
    if (m == 0 || t->n_items == 0) return;
    IRS_HIP(hipSetDevice(t->device));
    DeviceBuffer<float> d;
    d.alloc(static_cast<size_t>(m) * t->n_items);
    if (irs_ials_scores_device_(t, begin, end, d.ptr, nullptr, nullptr) != IRS_OK)
      throw std::runtime_error(irs::last_error());
    IRS_HIP(hipMemcpyAsync(out, d.ptr, static_cast<size_t>(m) * t->n_items * sizeof(float),
                           hipMemcpyDeviceToHost, t->stream));
    IRS_HIP(hipStreamSynchronize(t->stream));
  });
}

// Internal hook for evaluator.hip's fused path: user[begin:end] @ item^T into a
// device buffer on the trainer's stream (hpp:942-984).  Not part of the public ABI.
irs_status irs_ials_scores_device_(irs_ials_trainer *t, int64_t begin, int64_t end,
                                   float *device_out, void **stream_out, int32_t *device_index) {
  return guard([&] {
    check_arg(t != nullptr, "null trainer.");
    check_arg(0 <= begin && begin <= end && end <= t->n_users, "user block out of range.");
    IRS_HIP(hipSetDevice(t->device));
    if (stream_out) *stream_out = t->stream;
    if (device_index) *device_index = t->device;
    const int64_t m = end - begin;
    if (m == 0 || t->n_items == 0) return;
    check_arg(device_out != nullptr, "null output.");
    const int64_t waves = ceil_div(m, 64) * ceil_div(t->n_items, 64);
    t->prof.begin("user_scores", t->stream);
    if (t->gk()) {
      hipLaunchKernelGGL(gk_user_scores_kernel, dim3(ceil_div(waves, 4)), dim3(256), 0, t->stream,
                         static_cast<const float *>(t->factor[0].ptr),
                         static_cast<const float *>(t->factor[1].ptr), t->KP, begin, m, t->n_items,
                         device_out);
    } else {
      IRS_DISPATCH_ANY(t->T, {
        hipLaunchKernelGGL((user_scores_kernel<16 * TT>), dim3(ceil_div(waves, 4)), dim3(256), 0,
                           t->stream, t->factor[0].ptr, t->factor[1].ptr, begin, m, t->n_items,
                           device_out);
      });
    }
    t->prof.end(t->stream);
    IRS_HIP(hipGetLastError());
  });
}

// Internal hook for evaluator.hip's sample pass: user[begin:end] @ item[:n_prefix]^T into a
// device buffer [end - begin, n_prefix] on the trainer's stream; `item_rows` (device, [n_prefix,
// KP]) replaces the leading item rows when given, `user_rows` (device, [>= end, KP]) the user
// factors.  Not part of the public ABI.
irs_status irs_ials_scores_prefix_device_(irs_ials_trainer *t, int64_t begin, int64_t end,
                                          int64_t n_prefix, const float *item_rows,
                                          const float *user_rows, float *device_out) {
  return guard([&] {
    check_arg(t && device_out, "null argument.");
    check_arg(0 <= begin && begin <= end && end <= t->n_users && 0 < n_prefix &&
                  n_prefix <= t->n_items, "block out of range.");
    IRS_HIP(hipSetDevice(t->device));
    const int64_t m = end - begin;
    if (m == 0) return;
    const int64_t waves = ceil_div(m, 64) * ceil_div(n_prefix, 64);
    t->prof.begin("user_scores", t->stream);
    if (t->gk()) {
      hipLaunchKernelGGL(gk_user_scores_kernel, dim3(ceil_div(waves, 4)), dim3(256), 0, t->stream,
                         user_rows ? user_rows : static_cast<const float *>(t->factor[0].ptr),
                         item_rows ? item_rows : static_cast<const float *>(t->factor[1].ptr), t->KP,
                         begin, m, n_prefix, device_out);
    } else {
      IRS_DISPATCH_ANY(t->T, {
        hipLaunchKernelGGL((user_scores_kernel<16 * TT>), dim3(ceil_div(waves, 4)), dim3(256), 0,
                           t->stream, user_rows ? user_rows : t->factor[0].ptr,
                           item_rows ? item_rows : t->factor[1].ptr,
                           begin, m, n_prefix, device_out);
      });
    }
    t->prof.end(t->stream);
    IRS_HIP(hipGetLastError());
  });
}

// Internal hook for evaluator.hip's fused scoring + ranking kernel: the factor buffers where
// they live ([rows, KP] row-major, padded columns zero).  Not part of the public ABI.
irs_status irs_ials_factors_device_(irs_ials_trainer *t, const float **user, const float **item,
                                    int32_t *KP, int64_t *n_users, int64_t *n_items,
                                    void **stream_out, int32_t *device_index) {
  return guard([&] {
    check_arg(t && user && item && KP && n_users && n_items, "null argument.");
    *user = t->factor[0].ptr;
    *item = t->factor[1].ptr;
    *KP = t->KP;
    *n_users = t->n_users;
    *n_items = t->n_items;
    if (stream_out) *stream_out = t->stream;
    if (device_index) *device_index = t->device;
  });
}

namespace {

// X_to_vector / X_to_vector_with_prior (hpp:122-168): fold-in of new rows against the stored
// other-side factors.  Without a prior the rows start from 0; with one they start from the
// prior (hpp:156) and the solve adds reg_r * prior_r to the right-hand side.
void transform_impl(irs_ials_trainer *t, int32_t side, int64_t rows, int64_t cols,
                    const int64_t *indptr, const int32_t *indices, const float *data,
                    const float *prior, const irs_ials_solver_config *sc, float *out) {
  check_arg(t && out && (side == 0 || side == 1), "bad argument.");
  check_solver(sc);
  HostCsr X = host_csr(rows, cols, indptr, indices, data);
  IRS_HIP(hipSetDevice(t->device));
  full_gramian(t, side);  // hpp:793 / :799
  if (side == 1) X = transpose(X);  // hpp:800
  const int64_t n_other = t->rows_of(1 - side);
  if (X.cols != n_other) {  // hpp:126-131
    std::stringstream ss;
    ss << "Shape mismatch: X.cols() = " << X.cols
       << " but other.factor.rows() = " << n_other << ".";
    throw std::invalid_argument(ss.str());
  }
  if (prior && sc->solver_type == IRS_SOLVER_IALSPP)
    throw std::invalid_argument("Feature-aware iALS does not support IALSPP.");
  Side tmp;
  tmp.build(X, 0, X.rows, t->cfg, t->stream);
  DeviceBuffer<float> result, d_prior;  // DenseMatrix::Zero(X.rows(), K), hpp:132
  result.alloc(static_cast<size_t>(X.rows) * t->KP);
  result.zero(t->stream);
  if (prior) {
    std::vector<float> padded(static_cast<size_t>(X.rows) * t->KP, 0.0f);
    for (int64_t r = 0; r < X.rows; r++)
      std::copy(prior + r * t->K, prior + (r + 1) * t->K, padded.begin() + r * t->KP);
    d_prior.upload(padded, t->stream);
    IRS_HIP(hipMemcpyAsync(result.ptr, d_prior.ptr, padded.size() * sizeof(float),
                           hipMemcpyDeviceToDevice, t->stream));
    IRS_HIP(hipStreamSynchronize(t->stream));  // `padded` goes out of scope
  }
  launch_solve(t, tmp, t->factor[1 - side].ptr, result.ptr, side, sc,
               prior ? d_prior.ptr : nullptr);
  sync_and_check(t);
  download_factor(t, result.ptr, X.rows, out);
}

}  // namespace

irs_status irs_ials_transform(irs_ials_trainer *t, int32_t side, int64_t rows, int64_t cols,
                              const int64_t *indptr, const int32_t *indices, const float *data,
                              const irs_ials_solver_config *sc, float *out) {
  return guard([&] { transform_impl(t, side, rows, cols, indptr, indices, data, nullptr, sc, out); });
}

irs_status irs_ials_transform_with_prior(irs_ials_trainer *t, int32_t side, int64_t rows,
                                         int64_t cols, const int64_t *indptr,
                                         const int32_t *indices, const float *data,
                                         const float *prior, const irs_ials_solver_config *sc,
                                         float *out) {
  return guard([&] {
    check_arg(prior != nullptr, "prior is null.");
    transform_impl(t, side, rows, cols, indptr, indices, data, prior, sc, out);
  });
}

irs_status irs_ials_set_prior(irs_ials_trainer *t, int32_t which, const float *prior) {
  return guard([&] {
    check_arg(t && (which == 0 || which == 1), "bad argument.");
    if (prior == nullptr) {
      t->has_prior[which] = false;
      return;
    }
    IRS_HIP(hipSetDevice(t->device));
    const int64_t n = t->rows_of(which);
    std::vector<float> padded(static_cast<size_t>(n) * t->KP, 0.0f);
    for (int64_t r = 0; r < n; r++)
      std::copy(prior + r * t->K, prior + (r + 1) * t->K, padded.begin() + r * t->KP);
    t->prior[which].upload(padded, t->stream);
    IRS_HIP(hipStreamSynchronize(t->stream));
    t->has_prior[which] = true;
  });
}

constexpr int64_t FEATURE_RHS_CHUNK = 2048;

irs_status irs_ials_set_features(irs_ials_trainer *t, int32_t which, int64_t rows, int64_t n_feat,
                                 const int64_t *indptr, const int32_t *indices,
                                 const float *data) {
  return guard([&] {
    check_arg(t && (which == 0 || which == 1), "bad argument.");
    if (rows != t->rows_of(which))  // initialize_feature_aware, hpp:1005-1007
      throw std::invalid_argument("Feature matrix row count mismatch.");
    check_arg(t->whole, "feature-aware iALS needs an unsharded trainer.");
    HostCsr F = host_csr(rows, n_feat, indptr, indices, data);
    HostCsr Ft = transpose(F);
    IRS_HIP(hipSetDevice(t->device));
    auto to32 = [](const std::vector<int64_t> &v) {
      std::vector<int32_t> o(v.size());
      for (size_t i = 0; i < v.size(); i++) o[i] = static_cast<int32_t>(v[i]);
      return o;
    };
    hipStream_t s = t->stream;
    t->f_indptr[which].upload(to32(F.indptr), s);
    t->f_indices[which].upload(F.indices, s);
    // host_csr / transpose carry NO values for an all-ones matrix (one-hot features: the usual input);
    // feature_prior_kernel / feature_rhs_kernel read data[q] per stored entry, so the ones are made here
    const size_t f_nnz = static_cast<size_t>(F.indptr[rows]);
    std::vector<float> ones;
    if (F.unit && f_nnz > 0) ones.assign(f_nnz, 1.0f);
    if (F.unit) t->f_data[which].upload(ones, s);
    else t->f_data[which].upload(F.data, s);
    t->ft_indptr[which].upload(to32(Ft.indptr), s);
    t->ft_indices[which].upload(Ft.indices, s);
    if (F.unit) t->ft_data[which].upload(ones, s);
    else t->ft_data[which].upload(Ft.data, s);
    t->f_W[which].alloc(static_cast<size_t>(std::max<int64_t>(n_feat, 1)) * t->KP);
    t->f_rhs[which].alloc(static_cast<size_t>(std::max<int64_t>(n_feat, 1)) * t->KP);
    int64_t longest = 1;
    for (int64_t f = 0; f < n_feat; f++) longest = std::max(longest, Ft.indptr[f + 1] - Ft.indptr[f]);
    t->f_chunks[which] = static_cast<int>(ceil_div(longest, FEATURE_RHS_CHUNK));
    t->f_part[which].alloc(static_cast<size_t>(t->f_chunks[which]) * std::max<int64_t>(n_feat, 1) *
                           t->KP);
    IRS_HIP(hipStreamSynchronize(s));  // host vectors go out of scope
    t->n_feat[which] = n_feat;
  });
}

irs_status irs_ials_apply_feature_prior(irs_ials_trainer *t, int32_t which, const float *weight) {
  return guard([&] {
    check_arg(t && weight && (which == 0 || which == 1), "bad argument.");
    check_arg(t->n_feat[which] > 0, "no feature matrix was set for this side.");
    IRS_HIP(hipSetDevice(t->device));
    const int64_t F = t->n_feat[which], n = t->rows_of(which);
    std::vector<float> padded(static_cast<size_t>(F) * t->KP, 0.0f);
    for (int64_t f = 0; f < F; f++)
      std::copy(weight + f * t->K, weight + (f + 1) * t->K, padded.begin() + f * t->KP);
    t->f_W[which].upload(padded, t->stream);
    t->prior[which].alloc(static_cast<size_t>(ceil_div(n, 8) * 8) * t->KP);
    t->prof.begin(which == 0 ? "feature_prior_user" : "feature_prior_item", t->stream);
    hipLaunchKernelGGL(feature_prior_kernel, dim3(ceil_div(n, 4), ceil_div(t->KP, 256)), dim3(256), 0, t->stream,
                       t->f_indptr[which].ptr, t->f_indices[which].ptr, t->f_data[which].ptr,
                       t->f_W[which].ptr, n, t->KP, t->prior[which].ptr);
    t->prof.end(t->stream);
    IRS_HIP(hipGetLastError());
    IRS_HIP(hipStreamSynchronize(t->stream));  // `padded` goes out of scope
    t->has_prior[which] = true;
  });
}

irs_status irs_ials_feature_rhs(irs_ials_trainer *t, int32_t which, float *out) {
  return guard([&] {
    check_arg(t && out && (which == 0 || which == 1), "bad argument.");
    check_arg(t->n_feat[which] > 0, "no feature matrix was set for this side.");
    require_X(t);
    IRS_HIP(hipSetDevice(t->device));
    const int64_t F = t->n_feat[which];
    t->prof.begin(which == 0 ? "feature_rhs_user" : "feature_rhs_item", t->stream);
    const int nc = t->f_chunks[which];
    hipLaunchKernelGGL(feature_rhs_kernel, dim3(F, nc, ceil_div(t->KP, 256)), dim3(256), 0, t->stream,
                       t->ft_indptr[which].ptr, t->ft_indices[which].ptr, t->ft_data[which].ptr,
                       t->side[which].reg.ptr, t->factor[which].ptr, t->KP,
                       static_cast<int>(FEATURE_RHS_CHUNK), static_cast<int>(F),
                       t->f_part[which].ptr);
    hipLaunchKernelGGL(feature_rhs_reduce_kernel, dim3(ceil_div(F * t->KP, 256)), dim3(256), 0,
                       t->stream, t->f_part[which].ptr, nc, F * t->KP, t->f_rhs[which].ptr);
    t->prof.end(t->stream);
    IRS_HIP(hipGetLastError());
    IRS_HIP(hipMemcpy2DAsync(out, t->K * sizeof(float), t->f_rhs[which].ptr,
                             t->KP * sizeof(float), t->K * sizeof(float), F,
                             hipMemcpyDeviceToHost, t->stream));
    IRS_HIP(hipStreamSynchronize(t->stream));
  });
}

irs_status irs_ials_compute_loss(irs_ials_trainer *t, const irs_ials_solver_config *sc,
                                 float *out) {
  return guard([&] {
    check_arg(t && out, "null argument.");
    check_solver(sc);
    require_X(t);
    check_arg(t->whole, "compute_loss needs an unsharded trainer (a shard holds only its rows).");
    IRS_HIP(hipSetDevice(t->device));
    full_gramian(t, 0);  // hpp:837-838
    full_gramian(t, 1);
    double loss = 0.0;
    if (t->cfg.alpha0 != 0.0f) {  // hpp:840-844
      std::vector<float> pu(t->KP * t->KP), pi(t->KP * t->KP);
      IRS_HIP(hipMemcpyAsync(pu.data(), t->P[0].ptr, pu.size() * sizeof(float),
                             hipMemcpyDeviceToHost, t->stream));
      IRS_HIP(hipMemcpyAsync(pi.data(), t->P[1].ptr, pi.size() * sizeof(float),
                             hipMemcpyDeviceToHost, t->stream));
      IRS_HIP(hipStreamSynchronize(t->stream));
      double s = 0.0;
      for (size_t i = 0; i < pu.size(); i++) s += static_cast<double>(pu[i]) * pi[i];
      loss = s / t->cfg.alpha0;
    }
    const float bias = t->cfg.loss_type == IRS_LOSS_IALSPP ? 0.0f : t->cfg.alpha0;
    t->row_loss.alloc(static_cast<size_t>(std::max(t->n_users, t->n_items)));
    double h[2] = {0.0, 0.0};
    for (int s = 0; s < 2; s++) {
      const int64_t n = t->rows_of(s);
      if (n == 0) continue;
      if (t->gk()) {
        t->prof.begin("loss_rows", t->stream);
        hipLaunchKernelGGL(gk_loss_rows_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, t->stream,
                           static_cast<const float *>(t->factor[s].ptr),
                           static_cast<const float *>(t->factor[1 - s].ptr), t->KP,
                           static_cast<const int32_t *>(t->side[s].indptr.ptr),
                           static_cast<const int32_t *>(t->side[s].indices.ptr),
                           static_cast<const float *>(t->side[s].data.ptr),
                           static_cast<const float *>(t->side[s].reg.ptr), n, bias, s == 0 ? 1 : 0,
                           t->row_loss.ptr);
        t->prof.end(t->stream);
      } else {
        IRS_DISPATCH_ANY(t->T, {
          t->prof.begin("loss_rows", t->stream);
          hipLaunchKernelGGL((loss_rows_kernel<TT>), dim3(ceil_div(n, 4)), dim3(256), 0,
                             t->stream, t->factor[s].ptr, t->factor[1 - s].ptr,
                             t->side[s].indptr.ptr, t->side[s].indices.ptr, t->side[s].data.ptr,
                             t->side[s].reg.ptr, n, bias, s == 0 ? 1 : 0, t->row_loss.ptr);
          t->prof.end(t->stream);
        });
      }
      hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, t->stream, t->row_loss.ptr, n,
                         t->loss_sum.ptr + s);
      IRS_HIP(hipGetLastError());
      IRS_HIP(hipMemcpyAsync(&h[s], t->loss_sum.ptr + s, sizeof(double), hipMemcpyDeviceToHost,
                             t->stream));
      IRS_HIP(hipStreamSynchronize(t->stream));
    }
    *out = static_cast<float>((loss + h[0] + h[1]) / 2.0);  // hpp:939
  });
}

// Test hook (declared in the header, no reference counterpart): the eigen-decomposition kernel of
// ials_eig_kernels.hpp on a host matrix.  P: [K, K] row-major symmetric; out: Qrows [K, K] (row k
// = eigenvector k), lam [K], stats [3] (largest, smallest eigenvalue, sweeps).
irs_status irs_ials_eigen_debug(const float *P, int64_t K, int32_t device, float *Qrows, float *lam,
                                 float *stats, const float *P_prev) {
  return guard([&] {
    check_arg(P && Qrows && lam && stats && K >= 1 && K <= 128, "bad argument.");
    require_device(device);
    const int KP = K <= 64 ? 64 : 128;
    std::vector<float> padded(static_cast<size_t>(KP) * KP, 0.0f);
    for (int64_t r = 0; r < K; r++) std::copy(P + r * K, P + (r + 1) * K, padded.begin() + r * KP);
    DeviceBuffer<float> dP, dQr, dQc, dl, ds;
    DeviceBuffer<double> dQd;
    hipStream_t s = nullptr;
    dP.upload(padded, s);
    dQr.alloc(padded.size());
    dQc.alloc(padded.size());
    dl.alloc(KP);
    ds.alloc(4);
    dQd.alloc(padded.size());
    EigOut o{dQr.ptr, dQc.ptr, dl.ptr, ds.ptr, dQd.ptr};
    const size_t lds = padded.size() * sizeof(double);
    // P_prev given: decompose it first, then P warm-started from its eigenvectors
    for (int pass = P_prev ? 0 : 1; pass < 2; pass++) {
    const int warm = (pass == 1 && P_prev) ? 1 : 0;
    if (pass == 0) {
      std::vector<float> prev(padded.size(), 0.0f);
      for (int64_t r = 0; r < K; r++) std::copy(P_prev + r * K, P_prev + (r + 1) * K, prev.begin() + r * KP);
      dP.upload(prev, s);
      IRS_HIP(hipStreamSynchronize(s));
    } else if (P_prev) {
      dP.upload(padded, s);
    }
    if (KP == 128) {
      IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(eig_jacobi_kernel<128>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL(eig_jacobi_kernel<128>, dim3(1), dim3(1024), lds, s,
                         static_cast<const float *>(dP.ptr), static_cast<int>(K), o, warm);
    } else {
      IRS_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(eig_jacobi_kernel<64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL(eig_jacobi_kernel<64>, dim3(1), dim3(1024), lds, s,
                         static_cast<const float *>(dP.ptr), static_cast<int>(K), o, warm);
    }
    IRS_HIP(hipGetLastError());
    IRS_HIP(hipStreamSynchronize(s));
    }
    std::vector<float> q(padded.size()), l(KP);
    IRS_HIP(hipMemcpyAsync(q.data(), dQr.ptr, q.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    IRS_HIP(hipMemcpyAsync(l.data(), dl.ptr, KP * sizeof(float), hipMemcpyDeviceToHost, s));
    IRS_HIP(hipMemcpyAsync(stats, ds.ptr, 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    IRS_HIP(hipStreamSynchronize(s));
    for (int64_t k = 0; k < K; k++) {
      std::copy(q.begin() + k * KP, q.begin() + k * KP + K, Qrows + k * K);
      lam[k] = l[k];
    }
  });
}

// Diagnostics / tests: which rows of the last half-step were solved in the eigenbasis of the
// Gramian (ials_eig_kernels.hpp); bit 0 = the short rows (<= 32 stored entries).
int32_t irs_ials_last_eigenbasis(irs_ials_trainer *t) { return t ? t->eig_last : 0; }

irs_status irs_ials_profile(irs_ials_trainer *t, int32_t enable) {
  return guard([&] {
    check_arg(t != nullptr, "null trainer.");
    IRS_HIP(hipSetDevice(t->device));
    t->prof.clear();
    t->prof.enabled = enable != 0;
    t->prof.dominant_only = enable == 2;
    t->prof.dominant_suffix = t->n_users >= t->n_items ? "_user" : "_item";
  });
}

irs_status irs_ials_profile_read(irs_ials_trainer *t, int32_t cap, char (*names)[48],
                                 double *ms, int64_t *launches, int32_t *count) {
  return guard([&] {
    check_arg(t && names && ms && launches && count, "null argument.");
    IRS_HIP(hipSetDevice(t->device));
    t->prof.collect();
#ifdef IRS_IALS_PHASES
    {
      std::vector<unsigned long long> h(8 * 4096), z(8 * 4096, 0);
      IRS_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(ials_phase_clk), h.size() * 8));
      IRS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ials_phase_clk), z.data(), z.size() * 8));
      unsigned long long ph[8] = {0};
      for (int q = 0; q < 8; q++)
        for (int i = 0; i < 4096; i++) ph[q] += h[q * 4096 + i];
      fprintf(stderr, "ials phases (shader cycles summed over waves): gather+syrk %llu diag %llu trsm %llu trailing %llu backsub %llu\n",
              ph[0], ph[1], ph[2], ph[3], ph[4]);
    }
#endif
    int32_t i = 0;
    for (auto &kv : t->prof.totals) {
      if (i >= cap) break;
      std::strncpy(names[i], kv.first.c_str(), 47);
      names[i][47] = 0;
      ms[i] = kv.second.first;
      launches[i] = kv.second.second;
      i++;
    }
    *count = i;
  });
}

}  // extern "C"
