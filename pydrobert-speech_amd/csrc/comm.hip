// Multi-GPU gather of feature rows over RCCL (include/pds_amd.h, "Multi-GPU").  RCCL is loaded with
// dlopen on first use, so libpds_amd.so carries no load-time dependency on it and the single-GPU
// entry points work where it is absent.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "pds_internal.h"

struct pds_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
};

namespace pds {
namespace {

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl *rccl() {
  static Rccl lib;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib.handle) break;
    }
    if (!lib.handle) {
      lib.error = std::string("RCCL is not available: ") + dlerror();
      return;
    }
#define PDS_SYM(field, symbol)                                                     \
  lib.field = reinterpret_cast<decltype(lib.field)>(dlsym(lib.handle, #symbol)); \
  if (!lib.field) lib.error = "RCCL lacks " #symbol;
    PDS_SYM(GetUniqueId, ncclGetUniqueId)
    PDS_SYM(CommInitRank, ncclCommInitRank)
    PDS_SYM(CommInitAll, ncclCommInitAll)
    PDS_SYM(CommDestroy, ncclCommDestroy)
    PDS_SYM(AllGather, ncclAllGather)
    PDS_SYM(Broadcast, ncclBroadcast)
    PDS_SYM(AllReduce, ncclAllReduce)
    PDS_SYM(GroupStart, ncclGroupStart)
    PDS_SYM(GroupEnd, ncclGroupEnd)
    PDS_SYM(GetErrorString, ncclGetErrorString)
#undef PDS_SYM
  });
  if (!lib.error.empty()) {
    set_error(lib.error);
    return nullptr;
  }
  return &lib;
}

int32_t rccl_fail(Rccl *lib, ncclResult_t res, const char *what) {
  set_error(std::string(what) + ": " + lib->GetErrorString(res));
  return PDS_ERR_HIP;
}

#define PDS_RCCL(lib, call)                                      \
  do {                                                           \
    ncclResult_t res_ = (call);                                  \
    if (res_ != ncclSuccess) return rccl_fail(lib, res_, #call); \
  } while (0)

int32_t invalid_comm(const char *msg) {
  set_error(msg);
  return PDS_ERR_INVALID;
}

}  // namespace
}  // namespace pds

using namespace pds;

extern "C" {

static_assert(sizeof(ncclUniqueId) == PDS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

int32_t pds_comm_unique_id(void *id128) {
  if (!id128) return invalid_comm("comm_unique_id: null pointer");
  Rccl *lib = rccl();
  if (!lib) return PDS_ERR_INVALID;
  PDS_RCCL(lib, lib->GetUniqueId(static_cast<ncclUniqueId *>(id128)));
  return PDS_OK;
}

int32_t pds_comm_init_rank(const void *id128, int32_t world, int32_t rank, pds_comm **comm_out) {
  if (!id128 || !comm_out) return invalid_comm("comm_init_rank: null pointer");
  *comm_out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return invalid_comm("comm_init_rank: rank outside [0, world)");
  Rccl *lib = rccl();
  if (!lib) return PDS_ERR_INVALID;
  return no_throw("comm_init_rank", [&]() -> int32_t {
    auto comm = new pds_comm();
    {
      const hipError_t err = hipGetDevice(&comm->device);
      if (err != hipSuccess) {
        delete comm;
        return pds::hip_fail(err, "hipGetDevice");
      }
    }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    const ncclResult_t res = lib->CommInitRank(&comm->comm, world, id, rank);
    if (res != ncclSuccess) {
      delete comm;
      return rccl_fail(lib, res, "ncclCommInitRank");
    }
    comm->world = world;
    comm->rank = rank;
    *comm_out = comm;
    return PDS_OK;
  });
}

int32_t pds_comm_init_all(int32_t ndev, const int32_t *devices, pds_comm **comms_out) {
  if (!comms_out || ndev < 1) return invalid_comm("comm_init_all: need an array for ndev >= 1 handles");
  for (int i = 0; i < ndev; ++i) comms_out[i] = nullptr;
  Rccl *lib = rccl();
  if (!lib) return PDS_ERR_INVALID;
  return no_throw("comm_init_all", [&]() -> int32_t {
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) devs[i] = devices ? devices[i] : i;
    std::vector<ncclComm_t> raw(ndev, nullptr);
    PDS_RCCL(lib, lib->CommInitAll(raw.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; ++i) {
      auto comm = new pds_comm();
      comm->comm = raw[i];
      comm->world = ndev;
      comm->rank = i;
      comm->device = devs[i];
      comms_out[i] = comm;
    }
    return PDS_OK;
  });
}

int32_t pds_comm_world(const pds_comm *comm) { return comm ? comm->world : 0; }
int32_t pds_comm_rank(const pds_comm *comm) { return comm ? comm->rank : -1; }

void pds_comm_destroy(pds_comm *comm) {
  if (!comm) return;
  Rccl *lib = rccl();
  if (lib && comm->comm) (void)lib->CommDestroy(comm->comm);
  delete comm;
}

int32_t pds_comm_group_start(void) {
  Rccl *lib = rccl();
  if (!lib) return PDS_ERR_INVALID;
  PDS_RCCL(lib, lib->GroupStart());
  return PDS_OK;
}

int32_t pds_comm_group_end(void) {
  Rccl *lib = rccl();
  if (!lib) return PDS_ERR_INVALID;
  PDS_RCCL(lib, lib->GroupEnd());
  return PDS_OK;
}

int32_t pds_gather_rows(pds_comm *comm, const void *d_local, const int64_t *rows_per_rank, int64_t row_bytes,
                        void *d_out, void *stream) {
  if (!comm || !rows_per_rank) return invalid_comm("gather_rows: null communicator or row counts");
  if (row_bytes <= 0) return invalid_comm("gather_rows: row_bytes must be positive");
  Rccl *lib = rccl();
  if (!lib) return PDS_ERR_INVALID;
  int64_t total = 0;
  bool equal = true;
  for (int r = 0; r < comm->world; ++r) {
    if (rows_per_rank[r] < 0) return invalid_comm("gather_rows: negative row count");
    equal = equal && rows_per_rank[r] == rows_per_rank[0];
    total += rows_per_rank[r];
  }
  if (total == 0) return PDS_OK;
  if (!d_out || (rows_per_rank[comm->rank] > 0 && !d_local)) return invalid_comm("gather_rows: null buffer");
  hipStream_t s = (hipStream_t)stream;
  if (equal) {
    PDS_RCCL(lib, lib->AllGather(d_local, d_out, (size_t)(rows_per_rank[0] * row_bytes), ncclChar, comm->comm, s));
    return PDS_OK;
  }
  // ragged shards: every rank's block is broadcast into its place, all in one group
  PDS_RCCL(lib, lib->GroupStart());
  int64_t at = 0;
  ncclResult_t first_bad = ncclSuccess;
  for (int r = 0; r < comm->world; ++r) {
    const size_t bytes = (size_t)(rows_per_rank[r] * row_bytes);
    char *dst = static_cast<char *>(d_out) + at * row_bytes;
    if (bytes) {
      const ncclResult_t res = lib->Broadcast(r == comm->rank ? d_local : dst, dst, bytes, ncclChar, r, comm->comm, s);
      if (res != ncclSuccess && first_bad == ncclSuccess) first_bad = res;
    }
    at += rows_per_rank[r];
  }
  const ncclResult_t end = lib->GroupEnd();
  if (first_bad != ncclSuccess) return rccl_fail(lib, first_bad, "ncclBroadcast");
  if (end != ncclSuccess) return rccl_fail(lib, end, "ncclGroupEnd");
  return PDS_OK;
}

int32_t pds_allreduce_sum_f64(pds_comm *comm, double *d_table, int64_t count, void *stream) {
  if (!comm) return invalid_comm("allreduce_sum: null communicator");
  if (count < 0) return invalid_comm("allreduce_sum: negative count");
  if (count == 0) return PDS_OK;
  if (!d_table) return invalid_comm("allreduce_sum: null buffer");
  Rccl *lib = rccl();
  if (!lib) return PDS_ERR_INVALID;
  PDS_RCCL(lib, lib->AllReduce(d_table, d_table, (size_t)count, ncclDouble, ncclSum, comm->comm, (hipStream_t)stream));
  return PDS_OK;
}

}  // extern "C"
