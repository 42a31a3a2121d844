// Data-parallel exchange in the C ABI (train-step contract: `ssak_allreduce`): a sum all-reduce of a
// range of a device buffer over RCCL -- one process per GPU, xGMI between them -- for hosts that do not bring torch.distributed.
// librccl.so is dlopen'ed on first use: libssak_hip.so keeps no link-time dependency on it (a single-GPU host never loads it).
// The reference's counterpart is torch.nn.DataParallel's gather + loss.mean() inside HF Trainer
// (docker/transformers_modified/trainer.py:2532-2533; per-device batch = batch_size // num_devices,
// ssak/train/transformers/wav2vec_train.py:349,356).
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.h"

namespace {

// the few RCCL declarations used (rccl.h is not included: nothing of RCCL is needed to BUILD this library)
typedef struct {
  char internal[128];
} rcclUniqueId;  // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128
typedef void* rcclComm_t;
enum { RCCL_SUM = 0, RCCL_FLOAT32 = 7, RCCL_BFLOAT16 = 9 };  // ncclRedOp_t / ncclDataType_t values (stable across NCCL 2.x / RCCL)

struct Rccl {
  void* so = nullptr;
  int (*get_unique_id)(rcclUniqueId*) = nullptr;
  int (*comm_init_rank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*comm_destroy)(rcclComm_t) = nullptr;
  const char* (*error_string)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.so) return SSAK_OK;
  // An RCCL that the process has already mapped comes first (RTLD_NOLOAD): a torch process carries its own copy, and a second,
  // different librccl next to it would run two RCCL instances on one device.  Only then a fresh load.
  void* so = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so"}) {
    so = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (so) break;
  }
  if (!so)
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
  if (!so) {
    ssak_set_error("ssak_comm: librccl.so not found (%s)", dlerror());
    return SSAK_ERR_STATE;
  }
  Rccl r;
  r.so = so;
  r.get_unique_id = (int (*)(rcclUniqueId*))dlsym(so, "ncclGetUniqueId");
  r.comm_init_rank = (int (*)(rcclComm_t*, int, rcclUniqueId, int))dlsym(so, "ncclCommInitRank");
  r.all_reduce = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(so, "ncclAllReduce");
  r.comm_destroy = (int (*)(rcclComm_t))dlsym(so, "ncclCommDestroy");
  r.error_string = (const char* (*)(int))dlsym(so, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
    ssak_set_error("ssak_comm: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    dlclose(so);
    return SSAK_ERR_STATE;
  }
  g_rccl = r;
  return SSAK_OK;
}

int check_rccl(int rc, const char* what) {
  if (rc == 0) return SSAK_OK;
  ssak_set_error("ssak_comm: %s failed: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
  return SSAK_ERR_LAUNCH;
}

}  // namespace

struct ssak_comm {
  rcclComm_t comm = nullptr;
  int world = 1, rank = 0;
};

extern "C" int ssak_comm_unique_id(void* id128) {
  SSAK_REQUIRE(id128, "ssak_comm_unique_id: null pointer");
  if (int rc = load_rccl()) return rc;
  rcclUniqueId id;
  if (int rc = check_rccl(g_rccl.get_unique_id(&id), "ncclGetUniqueId")) return rc;
  memcpy(id128, id.internal, sizeof(id.internal));
  return SSAK_OK;
}

extern "C" int ssak_comm_create(ssak_comm** out, int world, int rank, const void* id128) {
  SSAK_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "ssak_comm_create: bad arguments (world %d, rank %d)", world, rank);
  if (int rc = load_rccl()) return rc;
  rcclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  ssak_comm* c = new ssak_comm;
  c->world = world;
  c->rank = rank;
  if (int rc = check_rccl(g_rccl.comm_init_rank(&c->comm, world, id, rank), "ncclCommInitRank")) {
    delete c;
    return rc;
  }
  *out = c;
  return SSAK_OK;
}

extern "C" int ssak_allreduce(ssak_comm* c, void* buf, long offset, long count, int dtype, void* stream) {
  SSAK_REQUIRE(c && c->comm && buf && offset >= 0 && count >= 0, "ssak_allreduce: bad arguments");
  SSAK_REQUIRE(dtype == SSAK_DTYPE_F32 || dtype == SSAK_DTYPE_BF16, "ssak_allreduce: dtype must be SSAK_DTYPE_F32 or SSAK_DTYPE_BF16");
  if (count == 0) return SSAK_OK;
  char* p = (char*)buf + (size_t)offset * (dtype == SSAK_DTYPE_F32 ? 4 : 2);
  return check_rccl(g_rccl.all_reduce(p, p, (size_t)count, dtype == SSAK_DTYPE_F32 ? RCCL_FLOAT32 : RCCL_BFLOAT16, RCCL_SUM, c->comm, (hipStream_t)stream),
                    "ncclAllReduce");
}

extern "C" int ssak_comm_destroy(ssak_comm* c) {
  if (!c) return SSAK_OK;
  int rc = SSAK_OK;
  if (c->comm) rc = check_rccl(g_rccl.comm_destroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}
