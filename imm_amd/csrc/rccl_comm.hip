// rccl_comm.hip — the gradient all-reduce of the data-parallel step as C-ABI entry points (SURVEY.md §8b export set:
// imm_rccl_{unique_id,init,allreduce,destroy}).
//
// Reference: imm/train/cnn_train_multi.py:66-106 `average_gradients` — the per-variable mean over the towers of one
// process, computed on the CPU that hosts the variables.  MI355X-native: one process per GPU, ONE sum all-reduce of the flat
// f32 gradient buffer (or of a bucket of it) over RCCL / xGMI, enqueued on the caller's HIP stream (so it can be captured
// into the step's HIP graph); the 1/N of the mean is folded into imm_clip_adam_step (grad_scale), which keeps the
// reference's order "average, then clip".
//
// RCCL is bound at RUN time (dlopen of the librccl the process already has — PyTorch-ROCm ships one — else the system
// one): libimm_hip.so does not link it, so the single-GPU path has no dependency on it and both users share one HIP runtime.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
// the slice of the NCCL/RCCL API used here (rccl.h: ncclUniqueId is 128 opaque bytes; ncclFloat = 7, ncclSum = 0)
typedef struct { char internal[128]; } imm_nccl_uid;
typedef void* nccl_comm_t;
typedef int (*fn_get_uid)(imm_nccl_uid*);
typedef int (*fn_comm_init_rank)(nccl_comm_t*, int, imm_nccl_uid, int);
typedef int (*fn_comm_destroy)(nccl_comm_t);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_err_string)(int);
typedef int (*fn_comm_count)(nccl_comm_t, int*);

struct Rccl {
  void* h = nullptr;
  fn_get_uid get_uid = nullptr;
  fn_comm_init_rank init_rank = nullptr;
  fn_comm_destroy destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_err_string err = nullptr;
  fn_comm_count count = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.h) return 0;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // the copy already in the process (PyTorch's), if any
    if (h) break;
  }
  for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) return imm_fail(IMM_E_UNSUPPORTED, "rccl: librccl.so not found (%s)", dlerror());
  Rccl r;
  r.h = h;
  r.get_uid = (fn_get_uid)dlsym(h, "ncclGetUniqueId");
  r.init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
  r.destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
  r.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
  r.err = (fn_err_string)dlsym(h, "ncclGetErrorString");
  r.count = (fn_comm_count)dlsym(h, "ncclCommCount");
  if (!r.get_uid || !r.init_rank || !r.destroy || !r.all_reduce || !r.err || !r.count)
    return imm_fail(IMM_E_UNSUPPORTED, "rccl: librccl.so lacks an NCCL entry point");
  g_rccl = r;
  return 0;
}

int nccl_fail(const char* what, int rc) { return imm_fail(IMM_E_HIP, "%s: %s", what, g_rccl.err ? g_rccl.err(rc) : "?"); }
}  // namespace

// 128 bytes that rank 0 creates and every rank passes to imm_rccl_init (exchange them with any host-side channel)
extern "C" int imm_rccl_unique_id(void* out128_host) {
  IMM_REQUIRE(out128_host, "rccl_unique_id: null");
  if (load_rccl()) return IMM_E_UNSUPPORTED;
  imm_nccl_uid id;
  const int rc = g_rccl.get_uid(&id);
  if (rc) return nccl_fail("ncclGetUniqueId", rc);
  memcpy(out128_host, &id, sizeof(id));
  return 0;
}

// Communicator of `world` ranks for the CURRENT device (one process per GPU); collective: every rank must call it.
extern "C" int imm_rccl_init(int rank, int world, const void* unique_id128_host, void** comm_out_host) {
  IMM_REQUIRE(unique_id128_host && comm_out_host && world >= 1 && rank >= 0 && rank < world, "rccl_init: args");
  if (load_rccl()) return IMM_E_UNSUPPORTED;
  imm_nccl_uid id;
  memcpy(&id, unique_id128_host, sizeof(id));
  nccl_comm_t comm = nullptr;
  const int rc = g_rccl.init_rank(&comm, world, id, rank);
  if (rc) return nccl_fail("ncclCommInitRank", rc);
  int n = 0;
  if (g_rccl.count(comm, &n) || n != world) {
    g_rccl.destroy(comm);
    return imm_fail(IMM_E_HIP, "rccl_init: communicator reports %d ranks, expected %d", n, world);
  }
  *comm_out_host = comm;
  return 0;
}

// buf[i] <- sum over ranks of buf[i], f32, in place, enqueued on `stream` (asynchronous; capturable into a HIP graph)
extern "C" int imm_rccl_allreduce(void* comm, float* buf, int64_t count, void* stream) {
  IMM_REQUIRE(comm && buf && count > 0, "rccl_allreduce: args");
  if (load_rccl()) return IMM_E_UNSUPPORTED;
  const int rc = g_rccl.all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, (nccl_comm_t)comm, (hipStream_t)stream);
  if (rc) return nccl_fail("ncclAllReduce", rc);
  return 0;
}

extern "C" int imm_rccl_destroy(void* comm) {
  IMM_REQUIRE(comm, "rccl_destroy: null");
  if (load_rccl()) return IMM_E_UNSUPPORTED;
  const int rc = g_rccl.destroy((nccl_comm_t)comm);
  if (rc) return nccl_fail("ncclCommDestroy", rc);
  return 0;
}
