// Transport of one rank of a row-sharded run (one process per GPU).
// The reference has no distributed layer; this is the transport of the two exchanges its
// epoch needs when rows are sharded (sharding.py): the K x K all-reduce of the Gramian and the
// exchange of the freshly solved factor rows, issued from INSIDE the library
// (irs_ials_sharded_step, ials.hip) so that an epoch is one call, not ten Python / torch calls.
//
// Two transports (comm.hip):
//   * RCCL (irs_comm_create): librccl.so is opened at run time (dlopen) - a single-GPU box
//     without RCCL still loads the library; irs_comm_create then fails with a clear message.
//     The rows travel by ncclAllGather / grouped ncclBroadcast (RCCL picks the algorithm, a ring
//     on xGMI) or by ONE group of ncclSend / ncclRecv per exchange - own block to every peer,
//     every peer's block into place: the full mesh, every link busy at once (IRS_EXCHANGE_MESH).
//   * peer stores (irs_comm_attach): every rank maps every other rank's factor buffers, a K x K
//     mailbox and a line of flags (hipIpcGetMemHandle / hipIpcOpenMemHandle) and a copy kernel
//     STORES the solved rows into all replicas at once over xGMI; arrival is signalled by
//     sequence numbers in uncached flag words that a one-wave kernel of the receiver waits for.
//     The Gramian all-reduce then is: own partial into slot[rank] of every mailbox, flags, a
//     sum over the slots in rank order - bitwise identical on every rank.  With peer stores on
//     both paths no RCCL communicator is needed at all (irs_comm_create_local), which is also
//     how two processes sharing ONE GPU exercise the whole sharded epoch in the tests.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <string>

#include "common.hpp"

namespace irs {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  void *handle = nullptr;

  static RcclApi &get() {
    static RcclApi api = [] {
      RcclApi a;
      for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a.handle) break;
      }
      if (!a.handle) return a;
      auto sym = [&](const char *n) { return dlsym(a.handle, n); };
      a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
      a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
      a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
      a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(sym("ncclCommAbort"));
      a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
      a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
      a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(sym("ncclBroadcast"));
      a.Send = reinterpret_cast<decltype(a.Send)>(sym("ncclSend"));
      a.Recv = reinterpret_cast<decltype(a.Recv)>(sym("ncclRecv"));
      a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
      a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
      a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
      return a;
    }();
    if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce ||
        !api.AllGather || !api.Broadcast || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
      throw std::runtime_error("librccl.so could not be opened (or lacks the nccl* entry points): "
                               "the RCCL transport of the sharded iALS step needs it.");
    return api;
  }
};

#define IRS_RCCL(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if (_r != ncclSuccess) {                                                                    \
      const auto gs = irs::RcclApi::get().GetErrorString;                                       \
      throw std::runtime_error(std::string("RCCL error: ") + (gs ? gs(_r) : "?") + " at " +     \
                               __FILE__ + ":" + std::to_string(__LINE__) + " (" #expr ")");     \
    }                                                                                           \
  } while (0)

constexpr int COMM_MAX_WORLD = 16;
// one flag line per kind of exchange; word r of a line is written by rank r only
constexpr int COMM_FLAG_ROWS = 0, COMM_FLAG_GRAM = 1, COMM_FLAG_KINDS = 2;
constexpr size_t COMM_FLAG_BYTES = 4096;  // flag lines at the head of the mailbox allocation

// what one rank hands to the others (irs_comm_export): 256 bytes
struct CommHandle {
  hipIpcMemHandle_t factor[2];  // user, item factor buffers [rows padded, KP]
  hipIpcMemHandle_t box;        // flags + mailbox
  int64_t pid;                  // ranks of ONE process (tests) exchange raw pointers instead
  uint64_t raw[3];              // factor[0], factor[1], box as pointers of that process
  int32_t KP, rank;
  char pad[256 - 3 * sizeof(hipIpcMemHandle_t) - sizeof(int64_t) - 3 * sizeof(uint64_t) - 2 * sizeof(int32_t)];
};
static_assert(sizeof(CommHandle) == 256, "the handle blob of irs_comm_export is 256 bytes");

}  // namespace irs

// RCCL: two communicators over the same ranks - `rows` carries the exchange of the solved rows on
// the communicator's own stream, `gram` the K x K all-reduces on the trainer's stream (the Gramian
// of the NEXT half-epoch needs only the rows this rank has just solved, so its all-reduce runs
// while the rows are in flight; two collectives may not be in flight on one communicator).
struct irs_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t rows = nullptr, gram = nullptr;      // null: a local (peer-store only) communicator
  hipStream_t stream = nullptr;                   // of the row exchange
  hipEvent_t ev_solved = nullptr, ev_rows = nullptr;
  int exchange = IRS_EXCHANGE_AUTO;
  // ---- peer stores
  bool attached = false;
  int KP = 0;
  char *box = nullptr;               // own flags + mailbox (uncached device memory)
  size_t box_bytes = 0;
  float *peer_factor[2][irs::COMM_MAX_WORLD] = {};   // [which][rank]; own entry = own buffer
  char *peer_box[irs::COMM_MAX_WORLD] = {};
  void *opened[3 * irs::COMM_MAX_WORLD] = {};        // what hipIpcOpenMemHandle returned (to close)
  int n_opened = 0;
  uint32_t seq[irs::COMM_FLAG_KINDS] = {0, 0};
  float *err_vec = nullptr;          // 8 floats: the bits of the solver's error flag, summed over ranks
  ~irs_comm();
  bool peer_rows() const { return exchange == IRS_EXCHANGE_PEER; }
  bool peer_gram() const { return gram == nullptr; }
};

namespace irs {

// ---- comm.hip: what irs_ials_sharded_step (ials.hip) calls
// buf[0:n] <- sum over ranks, on `s` (RCCL all-reduce on `gram`, or the mailbox exchange); n <= KP * KP
void comm_allreduce(irs_comm *c, float *buf, size_t n, hipStream_t s);
// rows [lo_r, hi_r) of every rank r (lo / hi: world entries) of the factor matrix `F` (= this rank's
// buffer of side `which`, leading dimension KP) into every replica, on c->stream behind
// whatever `solve_stream` holds; records c->ev_rows behind it
void comm_exchange_rows(irs_comm *c, int which, float *F, size_t KP, const int64_t *lo, const int64_t *hi,
                        int64_t n_rows, hipStream_t solve_stream, bool whole_shards);
// the solver's error flag (bit mask) of every rank OR-ed together on every rank, on `s`
void comm_allreduce_flag(irs_comm *c, int32_t *flag, hipStream_t s);
void comm_export(irs_comm *c, float *factor0, float *factor1, int KP, void *handle256);
void comm_attach(irs_comm *c, float *factor0, float *factor1, int KP, const void *handles);

}  // namespace irs
