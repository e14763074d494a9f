// Transport of the row-sharded iALS epoch: RCCL communicators and the peer-store path (comm.hpp).
// No reference counterpart (the reference has no distributed layer); the sharded
// unit is IALSTrainer::step, IALSTrainer.hpp:758-789.
#include "comm.hpp"

#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

using namespace irs;

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PeerPtrs {
  char *p[COMM_MAX_WORLD];
};

// Stores src[0:n16) (16-byte units) at the same offset of every peer's buffer: blockIdx.y walks
// the peers starting behind this rank, so that at any moment the ranks write to different targets
// and every xGMI link of the mesh carries one stream.  dst[q] already points at the destination
// of the first unit.
__global__ __launch_bounds__(256) void peer_push_kernel(const u32x4 *__restrict__ src, PeerPtrs dst, size_t n16,
                                                        int rank, int world) {
  const int q = (rank + 1 + static_cast<int>(blockIdx.y)) % world;
  u32x4 *__restrict__ out = reinterpret_cast<u32x4 *>(dst.p[q]);
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride)
    __builtin_nontemporal_store(src[i], out + i);
}

// word `rank` of the flag line `kind` of every peer <- seq.  Launched behind the push on the same
// stream: the end of the push kernel has released its stores at system scope, so a peer that sees
// the number sees the data.
__global__ void peer_signal_kernel(PeerPtrs box, int kind, int rank, int world, uint32_t seq) {
  const int q = static_cast<int>(threadIdx.x);
  if (q >= world || q == rank) return;
  uint32_t *flag = reinterpret_cast<uint32_t *>(box.p[q]) + kind * 64 + rank;
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Waits until every peer's word of the own flag line `kind` has reached seq (wrap-safe).  One
// wave; lane q polls peer q with a sleep in between.  timeout_ticks of the 100 MHz clock without
// progress -> bit 3 of the trainer's error flag and exit (a lost peer must not hang the GPU).
__global__ void peer_wait_kernel(char *box, int kind, int rank, int world, uint32_t seq, int64_t timeout_ticks,
                                 int32_t *err_flag) {
  const int q = static_cast<int>(threadIdx.x);
  if (q < world && q != rank) {
    const uint32_t *flag = reinterpret_cast<const uint32_t *>(box) + kind * 64 + q;
    const int64_t t0 = static_cast<int64_t>(wall_clock64());
    while (static_cast<int32_t>(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(32);
      if (static_cast<int64_t>(wall_clock64()) - t0 > timeout_ticks) {
        atomicOr(err_flag, 8);
        break;
      }
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// buf[i] <- sum over ranks r = 0 .. world - 1 (in that order: the same bits on every rank) of slot r
__global__ __launch_bounds__(256) void mailbox_sum_kernel(const float *__restrict__ slots, size_t slot_floats,
                                                          int world, float *__restrict__ buf, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.0f;
  for (int r = 0; r < world; r++)
    acc += __builtin_nontemporal_load(slots + static_cast<size_t>(r) * slot_floats + i);
  buf[i] = acc;
}

// bits[b] <- bit b of the solver's error flag (bit 3: a peer wait of THIS communicator timed out;
// that word lives behind the eight floats and is cleared here)
__global__ void flag_expand_kernel(const int32_t *flag, int32_t *timeout_word, float *bits) {
  const int32_t f = *flag | *timeout_word;
  bits[threadIdx.x] = ((f >> threadIdx.x) & 1) ? 1.0f : 0.0f;
  __syncthreads();
  if (threadIdx.x == 0) *timeout_word = 0;
}

__global__ void flag_compact_kernel(const float *bits, int32_t *flag) {
  int32_t f = 0;
  for (int b = 0; b < 8; b++)
    if (bits[b] > 0.0f) f |= 1 << b;
  *flag = f;
}

int64_t timeout_ticks() {
  static const int64_t ticks = [] {
    const char *e = std::getenv("IRSPACK_AMD_PEER_TIMEOUT_S");
    const double s = e ? std::max(0.001, std::atof(e)) : 20.0;
    return static_cast<int64_t>(s * 1e8);  // wall_clock64: 100 MHz
  }();
  return ticks;
}

size_t slot_floats_of(int KP) { return static_cast<size_t>(KP) * KP; }

// the two halves of the mailbox alternate between consecutive reductions: a rank that is already
// one reduction ahead cannot overwrite what a slower rank is still summing (two ahead is
// impossible: a row exchange, which needs every rank, lies in between)
float *mailbox_slot(char *box, int KP, int world, uint32_t seq, int r) {
  float *base = reinterpret_cast<float *>(box + COMM_FLAG_BYTES);
  return base + (static_cast<size_t>(seq & 1) * world + r) * slot_floats_of(KP);
}

void peer_signal_and_wait(irs_comm *c, int kind, uint32_t seq, hipStream_t s, int32_t *err_flag) {
  PeerPtrs boxes{};
  for (int r = 0; r < c->world; r++) boxes.p[r] = c->peer_box[r];
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, s, boxes, kind, c->rank, c->world, seq);
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, s, c->box, kind, c->rank, c->world, seq,
                     timeout_ticks(), err_flag);
  IRS_HIP(hipGetLastError());
}

}  // namespace

irs_comm::~irs_comm() {
  (void)hipSetDevice(device);
  if (stream) (void)hipStreamSynchronize(stream);
  if (rows || gram) {
    try {
      auto &api = RcclApi::get();
      if (rows) (void)api.CommDestroy(rows);
      if (gram) (void)api.CommDestroy(gram);
    } catch (...) {
    }
  }
  for (int i = 0; i < n_opened; i++)
    if (opened[i]) (void)hipIpcCloseMemHandle(opened[i]);
  if (box) (void)hipFree(box);
  if (err_vec) (void)hipFree(err_vec);
  if (ev_solved) (void)hipEventDestroy(ev_solved);
  if (ev_rows) (void)hipEventDestroy(ev_rows);
  if (stream) (void)hipStreamDestroy(stream);
}

namespace irs {

void comm_export(irs_comm *c, float *factor0, float *factor1, int KP, void *handle256) {
  check_arg(c && handle256 && factor0 && factor1 && KP > 0, "bad argument.");
  IRS_HIP(hipSetDevice(c->device));
  const size_t bytes = COMM_FLAG_BYTES + 2 * static_cast<size_t>(c->world) * slot_floats_of(KP) * sizeof(float);
  if (!c->box || c->box_bytes < bytes || c->KP != KP) {
    check_arg(!c->attached, "the communicator is already attached to a trainer of another size.");
    if (c->box) IRS_HIP(hipFree(c->box));
    c->box = nullptr;
    void *p = nullptr;
    // flags and mailbox are polled / read while remote ranks store into them: uncached
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      IRS_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
    }
    c->box = static_cast<char *>(p);
    c->box_bytes = bytes;
    c->KP = KP;
    IRS_HIP(hipMemset(c->box, 0, bytes));
  }
  CommHandle h;
  std::memset(&h, 0, sizeof(h));
  h.pid = static_cast<int64_t>(getpid());
  h.raw[0] = reinterpret_cast<uint64_t>(factor0);
  h.raw[1] = reinterpret_cast<uint64_t>(factor1);
  h.raw[2] = reinterpret_cast<uint64_t>(c->box);
  h.KP = KP;
  h.rank = c->rank;
  if (c->world > 1) {
    IRS_HIP(hipIpcGetMemHandle(&h.factor[0], factor0));
    IRS_HIP(hipIpcGetMemHandle(&h.factor[1], factor1));
    IRS_HIP(hipIpcGetMemHandle(&h.box, c->box));
  }
  std::memcpy(handle256, &h, sizeof(h));
}

void comm_attach(irs_comm *c, float *factor0, float *factor1, int KP, const void *handles) {
  check_arg(c && handles && factor0 && factor1, "bad argument.");
  check_arg(!c->attached, "the communicator is already attached.");
  check_arg(c->box != nullptr && c->KP == KP, "irs_comm_export must be called (on this trainer) before irs_comm_attach.");
  IRS_HIP(hipSetDevice(c->device));
  const CommHandle *hs = static_cast<const CommHandle *>(handles);
  const int64_t me = static_cast<int64_t>(getpid());
  for (int r = 0; r < c->world; r++) {
    const CommHandle &h = hs[r];
    check_arg(h.rank == r, "handles must be given in rank order.");
    check_arg(h.KP == KP, "every rank must run the same latent dimension.");
    if (r == c->rank) {
      c->peer_factor[0][r] = factor0;
      c->peer_factor[1][r] = factor1;
      c->peer_box[r] = c->box;
      continue;
    }
    if (h.pid == me) {  // ranks living in one process (tests): the pointers themselves
      c->peer_factor[0][r] = reinterpret_cast<float *>(h.raw[0]);
      c->peer_factor[1][r] = reinterpret_cast<float *>(h.raw[1]);
      c->peer_box[r] = reinterpret_cast<char *>(h.raw[2]);
      continue;
    }
    void *p[3] = {nullptr, nullptr, nullptr};
    const hipIpcMemHandle_t *src[3] = {&h.factor[0], &h.factor[1], &h.box};
    for (int k = 0; k < 3; k++) {
      IRS_HIP(hipIpcOpenMemHandle(&p[k], *src[k], hipIpcMemLazyEnablePeerAccess));
      c->opened[c->n_opened++] = p[k];
    }
    c->peer_factor[0][r] = static_cast<float *>(p[0]);
    c->peer_factor[1][r] = static_cast<float *>(p[1]);
    c->peer_box[r] = static_cast<char *>(p[2]);
  }
  c->attached = true;
}

void comm_allreduce(irs_comm *c, float *buf, size_t n, hipStream_t s) {
  if (!c->peer_gram()) {
    // (also at world size 1, where it is the identity: the one-GPU tests then run the very calls,
    // buffers and counts of a multi-GPU epoch)
    IRS_RCCL(RcclApi::get().AllReduce(buf, buf, n, ncclFloat, ncclSum, c->gram, s));
    return;
  }
  check_arg(c->attached, "a local communicator needs irs_comm_attach before the first step.");
  check_arg(n <= slot_floats_of(c->KP), "all-reduce larger than the mailbox slot.");
  const uint32_t seq = ++c->seq[COMM_FLAG_GRAM];
  const size_t bytes = n * sizeof(float);
  // own partial into slot[rank] of every mailbox (its own included) ...
  for (int q = 0; q < c->world; q++)
    IRS_HIP(hipMemcpyAsync(mailbox_slot(c->peer_box[q], c->KP, c->world, seq, c->rank), buf, bytes,
                           hipMemcpyDeviceToDevice, s));
  // ... the number behind it, wait for everybody's, sum in rank order
  peer_signal_and_wait(c, COMM_FLAG_GRAM, seq, s, reinterpret_cast<int32_t *>(c->err_vec + 8));
  hipLaunchKernelGGL(mailbox_sum_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s,
                     mailbox_slot(c->box, c->KP, c->world, seq, 0), slot_floats_of(c->KP), c->world, buf, n);
  IRS_HIP(hipGetLastError());
}

void comm_exchange_rows(irs_comm *c, int which, float *F, size_t KP, const int64_t *lo, const int64_t *hi,
                        int64_t n_rows, hipStream_t solve_stream, bool whole_shards) {
  IRS_HIP(hipEventRecord(c->ev_solved, solve_stream));
  IRS_HIP(hipStreamWaitEvent(c->stream, c->ev_solved, 0));
  const int me = c->rank, W = c->world;
  auto at = [&](int64_t row) { return F + static_cast<size_t>(row) * KP; };
  auto count = [&](int r) { return static_cast<size_t>(hi[r] - lo[r]) * KP; };
  if (c->peer_rows()) {
    check_arg(c->attached, "IRS_EXCHANGE_PEER needs irs_comm_attach before the first step.");
    const uint32_t seq = ++c->seq[COMM_FLAG_ROWS];
    if (W > 1 && hi[me] > lo[me]) {
      PeerPtrs dst{};
      for (int q = 0; q < W; q++)
        dst.p[q] = reinterpret_cast<char *>(c->peer_factor[which][q] + static_cast<size_t>(lo[me]) * KP);
      const size_t n16 = count(me) * sizeof(float) / 16;  // KP is a multiple of 16 floats
      const unsigned bx = static_cast<unsigned>(std::min<size_t>((n16 + 255) / 256, 512));
      hipLaunchKernelGGL(peer_push_kernel, dim3(bx, W - 1), dim3(256), 0, c->stream,
                         reinterpret_cast<const u32x4 *>(at(lo[me])), dst, n16, me, W);
    }
    peer_signal_and_wait(c, COMM_FLAG_ROWS, seq, c->stream, reinterpret_cast<int32_t *>(c->err_vec + 8));
  } else {
    check_arg(c->rows != nullptr, "a local communicator moves rows by peer stores only (IRS_EXCHANGE_PEER).");
    auto &api = RcclApi::get();
    // equal blocks of the row-padded buffer: one in-place all-gather
    const int64_t padded = (n_rows + 7) / 8 * 8, S = padded / W;
    bool equal = c->exchange == IRS_EXCHANGE_AUTO && whole_shards && padded % W == 0;
    for (int r = 0; equal && r < W; r++) equal = lo[r] == std::min<int64_t>(r * S, n_rows);
    if (c->exchange == IRS_EXCHANGE_MESH) {
      // own block to every peer, every peer's block into place: W - 1 sends and W - 1 receives in
      // ONE group, point to point - on the xGMI mesh every link carries one block at a time
      IRS_RCCL(api.GroupStart());
      for (int d = 1; d < W; d++) {
        const int to = (me + d) % W, from = (me - d + W) % W;
        if (hi[me] > lo[me]) IRS_RCCL(api.Send(at(lo[me]), count(me), ncclFloat, to, c->rows, c->stream));
        if (hi[from] > lo[from]) IRS_RCCL(api.Recv(at(lo[from]), count(from), ncclFloat, from, c->rows, c->stream));
      }
      IRS_RCCL(api.GroupEnd());
    } else if (equal) {
      IRS_RCCL(api.AllGather(at(static_cast<int64_t>(me) * S), F, static_cast<size_t>(S) * KP, ncclFloat, c->rows,
                             c->stream));
    } else {
      IRS_RCCL(api.GroupStart());
      for (int r = 0; r < W; r++)
        if (hi[r] > lo[r]) IRS_RCCL(api.Broadcast(at(lo[r]), at(lo[r]), count(r), ncclFloat, r, c->rows, c->stream));
      IRS_RCCL(api.GroupEnd());
    }
  }
  IRS_HIP(hipEventRecord(c->ev_rows, c->stream));
}

void comm_allreduce_flag(irs_comm *c, int32_t *flag, hipStream_t s) {
  // a time-out of this rank's peer waits (bit 3, kept behind err_vec) joins the solver's bits
  hipLaunchKernelGGL(flag_expand_kernel, dim3(1), dim3(8), 0, s, flag, reinterpret_cast<int32_t *>(c->err_vec + 8),
                     c->err_vec);
  IRS_HIP(hipGetLastError());
  comm_allreduce(c, c->err_vec, 8, s);
  hipLaunchKernelGGL(flag_compact_kernel, dim3(1), dim3(1), 0, s, c->err_vec, flag);
  IRS_HIP(hipGetLastError());
}

}  // namespace irs

extern "C" {

irs_status irs_comm_unique_id(void *id256) {
  return guard([&] {
    check_arg(id256 != nullptr, "null argument.");
    auto &api = RcclApi::get();
    ncclUniqueId ids[2];
    IRS_RCCL(api.GetUniqueId(&ids[0]));
    IRS_RCCL(api.GetUniqueId(&ids[1]));
    static_assert(sizeof(ids) == 256, "two 128-byte ids");
    std::memcpy(id256, ids, sizeof(ids));
  });
}

static std::unique_ptr<irs_comm> new_comm(int32_t rank, int32_t world, int32_t device) {
  check_arg(world >= 1 && world <= COMM_MAX_WORLD && rank >= 0 && rank < world, "rank out of range.");
  require_device(device);
  IRS_HIP(hipSetDevice(device));
  auto c = std::make_unique<irs_comm>();  // (its destructor releases whatever a later throw leaves behind)
  c->rank = rank;
  c->world = world;
  c->device = device;
  IRS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  IRS_HIP(hipEventCreateWithFlags(&c->ev_solved, hipEventDisableTiming));
  IRS_HIP(hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming));
  IRS_HIP(hipMalloc(reinterpret_cast<void **>(&c->err_vec), 16 * sizeof(float)));
  IRS_HIP(hipMemset(c->err_vec, 0, 16 * sizeof(float)));
  return c;
}

irs_status irs_comm_create(const void *id256, int32_t rank, int32_t world, int32_t device,
                           irs_comm **out) {
  return guard([&] {
    check_arg(id256 && out, "null argument.");
    auto c = new_comm(rank, world, device);
    auto &api = RcclApi::get();
    ncclUniqueId ids[2];
    std::memcpy(ids, id256, sizeof(ids));
    IRS_RCCL(api.CommInitRank(&c->rows, world, ids[0], rank));
    IRS_RCCL(api.CommInitRank(&c->gram, world, ids[1], rank));
    *out = c.release();
  });
}

irs_status irs_comm_create_local(int32_t rank, int32_t world, int32_t device, irs_comm **out) {
  return guard([&] {
    check_arg(out != nullptr, "null argument.");
    auto c = new_comm(rank, world, device);
    c->exchange = IRS_EXCHANGE_PEER;
    *out = c.release();
  });
}

irs_status irs_comm_set_exchange(irs_comm *c, int32_t mode) {
  return guard([&] {
    check_arg(c != nullptr, "null communicator.");
    check_arg(mode >= IRS_EXCHANGE_AUTO && mode <= IRS_EXCHANGE_PEER, "unknown exchange mode.");
    check_arg(mode == IRS_EXCHANGE_PEER || c->rows != nullptr,
              "a local communicator moves rows by peer stores only (IRS_EXCHANGE_PEER).");
    check_arg(mode != IRS_EXCHANGE_PEER || c->attached, "IRS_EXCHANGE_PEER needs irs_comm_attach first.");
    IRS_HIP(hipSetDevice(c->device));
    IRS_HIP(hipStreamSynchronize(c->stream));
    c->exchange = mode;
  });
}

int32_t irs_comm_get_exchange(irs_comm *c) { return c ? c->exchange : -1; }

irs_status irs_comm_destroy(irs_comm *c) {
  return guard([&] { delete c; });
}

}  // extern "C"
