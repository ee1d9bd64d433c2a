// RCCL collectives of the data-parallel path behind the C ABI: the flat-gradient all-reduce, and the small
// reductions that make a ray-sharded step equal to the 1-GPU step on the concatenated batch (sampler convergence flag,
// loss denominators).  The reference has no distributed code (main_recon.py:111-112 `strategy=None`).
//
// RCCL is bound at run time (dlopen): in a torch process `librccl.so.1` is already mapped (torch/lib), and the library proper
// keeps no link-time dependency on it -- single-GPU users never load it.  One communicator per process/GPU, created from a
// unique id that rank 0 makes and the caller distributes (i2sdf_amd/dist.py uses the torch.distributed store for those 128 bytes).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string.h>
#include <string>
#include "../../include/i2sdf.h"

int i2sdf_hip_check(hipError_t e, const char* what);

namespace {

struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) return;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.h, "ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast && r.GetErrorString;
  });
  return r;
}

thread_local std::string g_comm_err;

int nccl_check(ncclResult_t e, const char* what) {
  if (e == ncclSuccess) return I2SDF_OK;
  g_comm_err = std::string(what) + ": " + (rccl().ok ? rccl().GetErrorString(e) : "RCCL not loaded");
  return I2SDF_ECOMM;
}

}  // namespace

struct i2sdf_comm {
  ncclComm_t nccl = nullptr;
  int rank = 0, nranks = 1;
};

extern "C" const char* i2sdf_last_comm_error(void) { return g_comm_err.c_str(); }

// 1 if librccl could be bound in this process (no communicator is created): lets the ranks of a job agree on the transport BEFORE any
// of them enters the collective ncclCommInitRank, where a rank that cannot follow would leave the others waiting
extern "C" int32_t i2sdf_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int i2sdf_comm_unique_id(void* out, int64_t out_bytes) {
  if (!out || out_bytes < (int64_t)sizeof(ncclUniqueId)) return I2SDF_EINVAL;
  if (!rccl().ok) { g_comm_err = "librccl.so.1 could not be loaded"; return I2SDF_ECOMM; }
  ncclUniqueId id;
  int rc = nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  if (rc) return rc;
  memcpy(out, &id, sizeof(id));
  return I2SDF_OK;
}

extern "C" int i2sdf_comm_init_rank(const void* unique_id, int32_t nranks, int32_t rank, i2sdf_comm** out) {
  if (!unique_id || !out || nranks < 1 || rank < 0 || rank >= nranks) return I2SDF_EINVAL;
  if (!rccl().ok) { g_comm_err = "librccl.so.1 could not be loaded"; return I2SDF_ECOMM; }
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  i2sdf_comm* c = new i2sdf_comm();
  c->rank = rank; c->nranks = nranks;
  int rc = nccl_check(rccl().CommInitRank(&c->nccl, nranks, id, rank), "ncclCommInitRank");
  if (rc) { delete c; return rc; }
  *out = c;
  return I2SDF_OK;
}

extern "C" void i2sdf_comm_destroy(i2sdf_comm* c) {
  if (!c) return;
  if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
  delete c;
}

extern "C" int32_t i2sdf_comm_size(const i2sdf_comm* c) { return c ? c->nranks : 1; }
extern "C" int32_t i2sdf_comm_rank(const i2sdf_comm* c) { return c ? c->rank : 0; }

static int comm_allreduce(const i2sdf_comm* c, void* buf, int64_t n, int dtype, int op, hipStream_t st) {
  if (!c || !buf || n < 0) return I2SDF_EINVAL;
  if (n == 0) return I2SDF_OK;
  const ncclDataType_t dt = dtype == I2SDF_XCHG_I32 ? ncclInt32 : ncclFloat32;
  const ncclRedOp_t ro = op == I2SDF_XCHG_MAX ? ncclMax : (op == I2SDF_XCHG_AVG ? ncclAvg : ncclSum);
  return nccl_check(rccl().AllReduce(buf, buf, (size_t)n, dt, ro, c->nccl, st), "ncclAllReduce");
}

static int exchange_via_rccl(void* ctx, void* buf, int64_t n, int32_t dtype, int32_t op, void* stream) {
  return comm_allreduce((const i2sdf_comm*)ctx, buf, n, dtype, op, (hipStream_t)stream);
}

extern "C" int i2sdf_comm_as_exchange(const i2sdf_comm* comm, i2sdf_exchange* out) {
  if (!comm || !out) return I2SDF_EINVAL;
  out->allreduce = exchange_via_rccl;
  out->ctx = (void*)comm;
  return I2SDF_OK;
}

// i2sdf_allreduce_grads(flat, n, comm, stream).  In place; every rank ends up with the MEAN over ranks
// (DistributedDataParallel's convention: the loss of a rank is a mean over its own rays).
extern "C" int i2sdf_allreduce_grads(float* flat, int64_t n, const i2sdf_comm* comm, void* stream) {
  return comm_allreduce(comm, flat, n, I2SDF_XCHG_F32, I2SDF_XCHG_AVG, (hipStream_t)stream);
}

extern "C" int i2sdf_allreduce_max_i32(int32_t* flags, int64_t n, const i2sdf_comm* comm, void* stream) {
  return comm_allreduce(comm, flags, n, I2SDF_XCHG_I32, I2SDF_XCHG_MAX, (hipStream_t)stream);
}

extern "C" int i2sdf_broadcast(void* buf, int64_t bytes, int32_t root, const i2sdf_comm* comm, void* stream) {
  if (!comm || !buf || bytes < 0 || root < 0 || root >= comm->nranks) return I2SDF_EINVAL;
  if (bytes == 0) return I2SDF_OK;
  return nccl_check(rccl().Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, comm->nccl, (hipStream_t)stream), "ncclBroadcast");
}
