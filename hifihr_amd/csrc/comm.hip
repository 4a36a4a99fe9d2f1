// RCCL wrapper of the C ABI (include/hifihr.h, "gradient exchange"): the one collective of the data-parallel step, for callers that
// do not go through torch.distributed.  Replaces what nn.DataParallel's gather / reduce does in the reference (reference
// train_hrnet.py:560): one process per GPU, SUM all-reduce of the flat fp32 gradient buffer over
// xGMI, parameters broadcast once at start.
//
// librccl is resolved at the first call with dlopen (the copy already mapped into the process -- e.g. torch's -- is reused):
// libhifihr.so itself has no link-time dependency on it, so a single-GPU user never loads it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/hifihr.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char err[256] = {0};
};

// Resolved once, thread-safely (function-local static initialised by a lambda: C++11 guarantees one initialisation; the round-2
// version set a plain `tried` flag before the symbols were resolved, so a second thread could see a half-filled table).
static void rccl_load(Rccl& r);
Rccl& rccl_table() {
  static Rccl r = [] { Rccl t; rccl_load(t); return t; }();
  return r;
}
Rccl* rccl() {
  Rccl& r = rccl_table();
  return r.lib ? &r : nullptr;
}
static void rccl_load(Rccl& r) {
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {                        // a copy that is already mapped wins (one RCCL per process)
    r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (r.lib) break;
  }
  for (int i = 0; !r.lib && i < 3; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) { const char* e = dlerror(); snprintf(r.err, sizeof(r.err), "librccl not found: %s", e ? e : "(no dlerror text)"); return; }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
  r.Broadcast = (decltype(r.Broadcast))dlsym(r.lib, "ncclBroadcast");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Broadcast) {
    snprintf(r.err, sizeof(r.err), "librccl lacks an expected symbol (ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / Broadcast)");
    dlclose(r.lib);
    r.lib = nullptr;
  }
}

thread_local char g_comm_err[320];
int comm_fail(int code, const char* what, ncclResult_t res = ncclSuccess) {
  Rccl* r = rccl();
  if (r == nullptr)                                      // the library itself is missing / incomplete: say why (dlopen / dlsym text)
    snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", what, rccl_table().err);
  else
    snprintf(g_comm_err, sizeof(g_comm_err), "%s%s%s", what, res != ncclSuccess ? ": " : "",
             res != ncclSuccess && r->GetErrorString ? r->GetErrorString(res) : "");
  return code;
}

}  // namespace

struct hifihr_comm {
  ncclComm_t comm;
  int rank, world;
};

extern "C" {

const char* hifihr_comm_last_error(void) { return g_comm_err; }

int hifihr_comm_get_unique_id(hifihr_comm_uid* out) {
  static_assert(sizeof(hifihr_comm_uid) == sizeof(ncclUniqueId), "unique id size");
  if (!out) return comm_fail(HIFIHR_EINVAL, "hifihr_comm_get_unique_id: null argument");
  Rccl* r = rccl();
  if (!r) return comm_fail(HIFIHR_EHIP, "hifihr_comm_get_unique_id: RCCL is not available");
  ncclUniqueId id;
  const ncclResult_t res = r->GetUniqueId(&id);
  if (res != ncclSuccess) return comm_fail(HIFIHR_EHIP, "ncclGetUniqueId", res);
  memcpy(out, &id, sizeof(id));
  return HIFIHR_OK;
}

int hifihr_comm_init(hifihr_comm** out, int rank, int world, const hifihr_comm_uid* uid) {
  if (!out || !uid || world < 1 || rank < 0 || rank >= world) return comm_fail(HIFIHR_EINVAL, "hifihr_comm_init: bad argument");
  Rccl* r = rccl();
  if (!r) return comm_fail(HIFIHR_EHIP, "hifihr_comm_init: RCCL is not available");
  ncclUniqueId id;
  memcpy(&id, uid, sizeof(id));
  ncclComm_t c;
  const ncclResult_t res = r->CommInitRank(&c, world, id, rank);        // binds to the calling thread's current HIP device
  if (res != ncclSuccess) return comm_fail(HIFIHR_EHIP, "ncclCommInitRank", res);
  hifihr_comm* h = new (std::nothrow) hifihr_comm{c, rank, world};
  if (!h) { r->CommDestroy(c); return comm_fail(HIFIHR_EHIP, "hifihr_comm_init: out of memory"); }
  *out = h;
  return HIFIHR_OK;
}

int hifihr_comm_destroy(hifihr_comm* h) {
  if (!h) return HIFIHR_OK;
  Rccl* r = rccl();
  if (r) r->CommDestroy(h->comm);
  delete h;
  return HIFIHR_OK;
}

int hifihr_comm_allreduce_f32(hifihr_comm* h, float* buf_d, size_t n, void* stream) {
  if (!h || !buf_d) return comm_fail(HIFIHR_EINVAL, "hifihr_comm_allreduce_f32: null argument");
  Rccl* r = rccl();
  if (!r) return comm_fail(HIFIHR_EHIP, "hifihr_comm_allreduce_f32: RCCL is not available");
  const ncclResult_t res = r->AllReduce(buf_d, buf_d, n, ncclFloat32, ncclSum, h->comm, (hipStream_t)stream);
  return res == ncclSuccess ? HIFIHR_OK : comm_fail(HIFIHR_EHIP, "ncclAllReduce", res);
}

int hifihr_comm_broadcast_f32(hifihr_comm* h, float* buf_d, size_t n, int root, void* stream) {
  if (!h || !buf_d || root < 0 || root >= h->world) return comm_fail(HIFIHR_EINVAL, "hifihr_comm_broadcast_f32: bad argument");
  Rccl* r = rccl();
  if (!r) return comm_fail(HIFIHR_EHIP, "hifihr_comm_broadcast_f32: RCCL is not available");
  const ncclResult_t res = r->Broadcast(buf_d, buf_d, n, ncclFloat32, root, h->comm, (hipStream_t)stream);
  return res == ncclSuccess ? HIFIHR_OK : comm_fail(HIFIHR_EHIP, "ncclBroadcast", res);
}

}  // extern "C"
