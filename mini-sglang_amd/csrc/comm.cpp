// Tensor-parallel communicator over RCCL (xGMI).  C-ABI: msgl_comm_* in include/msgl_hip.h.
//
// Same contract as the reference's NCCLWrapper (C/src/pynccl.cu:72-175): in-place SUM
// all-reduce of fp16/bf16 tensors, all-gather into a rank-ordered destination, both enqueued
// on the caller's stream (capturable in a hipGraph), never synchronising.
//
// The reference stages messages <= max_bytes through an ncclMemAlloc + symmetric-window buffer
// (pynccl.cu:81-90, 105-123).  Here the default is the direct in-place collective on the
// caller's tensor (no extra HBM round trip); staging through a buffer registered with
// ncclCommRegister is kept as an opt-in (MSGL_COMM_STAGE=1) until it can be measured on an
// 8-GPU xGMI node -- this round only had 1-GPU boxes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/msgl_hip.h"

static_assert(MSGL_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {

thread_local char g_err[512] = "";

void set_err(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define COMM_NCCL(call, what)                                        \
  do {                                                               \
    ncclResult_t r_ = (call);                                        \
    if (r_ != ncclSuccess) {                                         \
      set_err("%s: %s", what, ncclGetErrorString(r_));               \
      return MSGL_ECOMM;                                             \
    }                                                                \
  } while (0)

#define COMM_HIP(call, what)                                         \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) {                                          \
      set_err("%s: %s", what, hipGetErrorString(e_));                \
      return MSGL_ELAUNCH;                                           \
    }                                                                \
  } while (0)

bool to_nccl_dtype(int dtype, ncclDataType_t* out) {
  if (dtype == MSGL_BF16) { *out = ncclBfloat16; return true; }
  if (dtype == MSGL_FP16) { *out = ncclFloat16; return true; }
  return false;  // the reference supports fp16/bf16 only (pynccl.cu:59-63)
}

}  // namespace

struct msgl_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  size_t max_bytes = 0;
  void* buf = nullptr;       // staging buffer (device)
  void* reg_handle = nullptr;  // ncclCommRegister handle, may stay null
  bool stage = false;          // MSGL_COMM_STAGE=1
  bool shortcut1 = true;       // world == 1: skip the library call (MSGL_COMM_NO_SHORTCUT=1 keeps it, so that
                               // a 1-GPU box can exercise RCCL's enqueue path, e.g. under hipGraph capture)
};

extern "C" {

const char* msgl_comm_last_error(void) { return g_err; }

int msgl_comm_unique_id(char out_id[MSGL_UNIQUE_ID_BYTES]) {
  if (!out_id) { set_err("comm_unique_id: null pointer"); return MSGL_EINVAL; }
  ncclUniqueId id;
  COMM_NCCL(ncclGetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out_id, id.internal, NCCL_UNIQUE_ID_BYTES);
  return MSGL_OK;
}

int msgl_comm_create(msgl_comm_t* out, int rank, int world_size, const char id[MSGL_UNIQUE_ID_BYTES],
                     size_t max_bytes) {
  if (!out || !id || world_size < 1 || rank < 0 || rank >= world_size) {
    set_err("comm_create: bad arguments (rank %d of %d)", rank, world_size);
    return MSGL_EINVAL;
  }
  msgl_comm* c = new msgl_comm();
  c->rank = rank;
  c->world = world_size;
  c->max_bytes = max_bytes;
  ncclUniqueId uid;
  memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = ncclCommInitRank(&c->comm, world_size, uid, rank);
  if (r != ncclSuccess) {
    set_err("ncclCommInitRank: %s", ncclGetErrorString(r));
    delete c;
    return MSGL_ECOMM;
  }
  const char* st = getenv("MSGL_COMM_STAGE");
  c->stage = st && st[0] == '1';
  const char* ns = getenv("MSGL_COMM_NO_SHORTCUT");
  c->shortcut1 = !(ns && ns[0] == '1');
  if (max_bytes > 0) {
    if (hipMalloc(&c->buf, max_bytes) != hipSuccess) {
      (void)hipGetLastError();
      c->buf = nullptr;
      c->max_bytes = 0;
    } else if (ncclCommRegister(c->comm, c->buf, max_bytes, &c->reg_handle) != ncclSuccess) {
      c->reg_handle = nullptr;  // best effort
    }
  }
  *out = c;
  return MSGL_OK;
}

int msgl_comm_all_reduce_sum(msgl_comm_t c, void* data, size_t count, int dtype, void* stream) {
  if (!c || !data) { set_err("comm_all_reduce: null pointer"); return MSGL_EINVAL; }
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt)) { set_err("comm_all_reduce: dtype code %d unsupported", dtype); return MSGL_EINVAL; }
  if (count == 0 || (c->world == 1 && c->shortcut1)) return MSGL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t bytes = count * 2;
  if (c->stage && c->buf && c->reg_handle && bytes <= c->max_bytes && data != c->buf) {
    COMM_HIP(hipMemcpyAsync(c->buf, data, bytes, hipMemcpyDeviceToDevice, s), "stage in");
    COMM_NCCL(ncclAllReduce(c->buf, c->buf, count, dt, ncclSum, c->comm, s), "ncclAllReduce");
    COMM_HIP(hipMemcpyAsync(data, c->buf, bytes, hipMemcpyDeviceToDevice, s), "stage out");
  } else {
    COMM_NCCL(ncclAllReduce(data, data, count, dt, ncclSum, c->comm, s), "ncclAllReduce");
  }
  return MSGL_OK;
}

int msgl_comm_all_gather(msgl_comm_t c, void* dst, const void* src, size_t count, int dtype, void* stream) {
  if (!c || !dst || !src) { set_err("comm_all_gather: null pointer"); return MSGL_EINVAL; }
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt)) { set_err("comm_all_gather: dtype code %d unsupported", dtype); return MSGL_EINVAL; }
  if (count == 0) return MSGL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->world == 1 && c->shortcut1) {
    if (dst != src) COMM_HIP(hipMemcpyAsync(dst, src, count * 2, hipMemcpyDeviceToDevice, s), "all_gather copy");
    return MSGL_OK;
  }
  COMM_NCCL(ncclAllGather(src, dst, count, dt, c->comm, s), "ncclAllGather");
  return MSGL_OK;
}

void* msgl_comm_get_buffer(msgl_comm_t c) { return c ? c->buf : nullptr; }

// What the library itself reports for this communicator (not what the caller passed to create): bench.py prints
// nranks as `rccl_ranks_seen`, so that an N-GPU line shows the collective really spans N ranks on N devices.
int msgl_comm_info(msgl_comm_t c, int* nranks, int* rank, int* device) {
  if (!c || !c->comm) { set_err("comm_info: null communicator"); return MSGL_EINVAL; }
  int n = 0, r = 0, d = 0;
  COMM_NCCL(ncclCommCount(c->comm, &n), "ncclCommCount");
  COMM_NCCL(ncclCommUserRank(c->comm, &r), "ncclCommUserRank");
  COMM_NCCL(ncclCommCuDevice(c->comm, &d), "ncclCommCuDevice");
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (device) *device = d;
  return MSGL_OK;
}

int msgl_comm_destroy(msgl_comm_t c) {
  if (!c) return MSGL_OK;
  if (c->reg_handle) (void)ncclCommDeregister(c->comm, c->reg_handle);
  if (c->buf) (void)hipFree(c->buf);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return MSGL_OK;
}

}  // extern "C"
