// RCCL communicators of one rank of a row-sharded run (one process per GPU).
// The reference has no distributed layer; this is the transport of the two exchanges its
// epoch needs when rows are sharded (sharding.py): the K x K all-reduce of the Gramian and the
// all-gather of the freshly solved factor rows, issued from INSIDE the library
// (irs_ials_sharded_step, ials.hip) so that an epoch is one call, not ten Python / torch calls.
//
// librccl.so is opened at run time (dlopen): a single-GPU box without RCCL still loads the
// library; irs_comm_* then fail with a clear message.
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
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
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
      a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
      a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
      a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(sym("ncclBroadcast"));
      a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
      a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
      a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
      return a;
    }();
    if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce ||
        !api.AllGather || !api.Broadcast || !api.GroupStart || !api.GroupEnd)
      throw std::runtime_error("librccl.so could not be opened (or lacks the nccl* entry points): "
                               "the sharded iALS step needs RCCL.");
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

}  // namespace irs

// Two communicators over the same ranks: `rows` carries the all-gather of the solved rows on the
// communicator's own stream, `gram` the K x K all-reduces on the trainer's stream - the Gramian of
// the NEXT half-epoch needs only the rows this rank has just solved, so its all-reduce runs while
// the all-gather is in flight; two collectives may not be in flight on one communicator.
struct irs_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t rows = nullptr, gram = nullptr;
  hipStream_t stream = nullptr;                   // of the row exchange
  hipEvent_t ev_solved = nullptr, ev_rows = nullptr;
};
