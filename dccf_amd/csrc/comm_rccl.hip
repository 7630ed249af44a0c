// comm_rccl.hip — the row-sharded trainer's collectives straight on RCCL (xGMI inside a node): the all-to-all of row payloads
// as ONE group of ncclSend / ncclRecv per peer on the caller's stream, and the all-reduce of [dW | db].  torch.distributed issues
// the same RCCL calls, but every c10d call costs 25-40 us of host time — four of them per 0.2 ms step made the sharded step
// host-bound.  No reference counterpart (the reference is single-GPU, src/main.py:106,153-155).
//
// RCCL is NOT a link-time dependency of this library: the process is PyTorch-ROCm, whose wheel ships its own librccl.so.1, and
// two RCCL copies in one process must not happen.  The entry points are resolved at run time from the copy the process has
// already mapped (dlopen with RTLD_NOLOAD only; torch.distributed loads it) — never from the loader's search path.
#include "common.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*GroupStart)();
  ncclResult_t (*GroupEnd)();
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
  bool ok;
};
const Api& api() {
  static const Api a = [] {
    Api q;
    memset(&q, 0, sizeof(q));
    // only the copy the process has ALREADY mapped (PyTorch-ROCm's, loaded with torch.distributed's nccl backend): a plain
    // dlopen could map a second RCCL next to it — exactly what must not happen.  Not mapped = "RCCL not available": the
    // caller's torch.distributed branch does the same work.
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) return q;
#define SYM(field, name) q.field = (decltype(q.field))dlsym(h, name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.GroupStart && q.GroupEnd && q.Send && q.Recv && q.AllReduce;
    return q;
  }();
  return a;
}
struct Comm {
  ncclComm_t c;
  int world, rank;
};
}  // namespace

#define NCCL_TRY(x)                                                                                                   \
  do {                                                                                                                \
    const ncclResult_t r_ = (x);                                                                                      \
    if (r_ != ncclSuccess) {                                                                                          \
      snprintf(g_dccf_err, sizeof(g_dccf_err), "%s failed: %s (%s:%d)", #x,                                        \
               api().GetErrorString ? api().GetErrorString(r_) : "?", __FILE__, __LINE__);                           \
      return 1000 + (int)r_;                                                                                          \
    }                                                                                                                 \
  } while (0)

extern "C" int dccf_comm_unique_id(uint8_t* out128) {
  ARG_CHECK(out128 != nullptr, "NULL output");
  ARG_CHECK(api().ok, "RCCL is not available in this process (no librccl.so.1 mapped: create the torch.distributed nccl process group first)");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCL_TRY(api().GetUniqueId(&id));
  memcpy(out128, &id, 128);
  return 0;
}

extern "C" int dccf_comm_create(void** comm, const uint8_t* id128, int32_t world, int32_t rank) {
  ARG_CHECK(comm && id128 && world >= 1 && rank >= 0 && rank < world, "bad arguments");
  ARG_CHECK(api().ok, "RCCL is not available in this process (no librccl.so.1 mapped: create the torch.distributed nccl process group first)");
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  Comm* c = new Comm();
  c->world = world;
  c->rank = rank;
  const ncclResult_t r = api().CommInitRank(&c->c, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    snprintf(g_dccf_err, sizeof(g_dccf_err), "ncclCommInitRank failed: %s", api().GetErrorString ? api().GetErrorString(r) : "?");
    return 1000 + (int)r;
  }
  *comm = c;
  return 0;
}

extern "C" int dccf_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* c = (Comm*)comm;
  if (api().ok) (void)api().CommDestroy(c->c);
  delete c;
  return 0;
}

// All-to-all of rows of `width` floats: peer q gets send_rows[q] rows (contiguous, peers in rank order in `send`) and delivers
// recv_rows[q] rows (contiguous, in rank order in `recv`).  Up to two payloads of different row widths travel in ONE RCCL group
// on `stream` (the sharded step's forward exchange: embedding-side rows and 768-d feature rows).
static int a2a_enqueue(Comm* c, const float* send, const int64_t* send_rows, float* recv, const int64_t* recv_rows, int64_t width,
                       hipStream_t st) {
  int64_t so = 0, ro = 0;
  for (int q = 0; q < c->world; ++q) {
    if (send_rows[q] > 0) NCCL_TRY(api().Send(send + so * width, (size_t)(send_rows[q] * width), ncclFloat, q, c->c, st));
    if (recv_rows[q] > 0) NCCL_TRY(api().Recv(recv + ro * width, (size_t)(recv_rows[q] * width), ncclFloat, q, c->c, st));
    so += send_rows[q];
    ro += recv_rows[q];
  }
  return 0;
}
static int a2a_check(const Comm* c, const float* send, const int64_t* send_rows, const float* recv, const int64_t* recv_rows,
                     int64_t width) {
  ARG_CHECK(send_rows && recv_rows && width >= 1, "bad arguments");
  for (int q = 0; q < c->world; ++q) {
    ARG_CHECK(send_rows[q] >= 0 && recv_rows[q] >= 0, "negative row count");
    ARG_CHECK((send_rows[q] == 0 || send) && (recv_rows[q] == 0 || recv), "NULL buffer");
  }
  return 0;
}
extern "C" int dccf_comm_all_to_all_rows(void* comm, const float* send, const int64_t* send_rows, float* recv,
                                         const int64_t* recv_rows, int64_t width, void* stream) {
  ARG_CHECK(comm != nullptr, "NULL communicator");
  Comm* c = (Comm*)comm;
  if (int e = a2a_check(c, send, send_rows, recv, recv_rows, width)) return e;
  NCCL_TRY(api().GroupStart());
  const int e = a2a_enqueue(c, send, send_rows, recv, recv_rows, width, (hipStream_t)stream);
  const ncclResult_t r = api().GroupEnd();
  if (e) return e;
  NCCL_TRY(r);
  return 0;
}
extern "C" int dccf_comm_all_to_all_rows2(void* comm, const float* send_a, const int64_t* send_rows_a, float* recv_a,
                                          const int64_t* recv_rows_a, int64_t width_a, const float* send_b,
                                          const int64_t* send_rows_b, float* recv_b, const int64_t* recv_rows_b, int64_t width_b,
                                          void* stream) {
  ARG_CHECK(comm != nullptr, "NULL communicator");
  Comm* c = (Comm*)comm;
  if (int e = a2a_check(c, send_a, send_rows_a, recv_a, recv_rows_a, width_a)) return e;
  if (int e = a2a_check(c, send_b, send_rows_b, recv_b, recv_rows_b, width_b)) return e;
  NCCL_TRY(api().GroupStart());
  int e = a2a_enqueue(c, send_a, send_rows_a, recv_a, recv_rows_a, width_a, (hipStream_t)stream);
  if (!e) e = a2a_enqueue(c, send_b, send_rows_b, recv_b, recv_rows_b, width_b, (hipStream_t)stream);
  const ncclResult_t r = api().GroupEnd();
  if (e) return e;
  NCCL_TRY(r);
  return 0;
}

extern "C" int dccf_comm_all_reduce_sum(void* comm, float* buf, int64_t n, void* stream) {
  ARG_CHECK(comm && (buf || n == 0) && n >= 0, "bad arguments");
  if (n == 0) return 0;
  Comm* c = (Comm*)comm;
  NCCL_TRY(api().AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->c, (hipStream_t)stream));
  return 0;
}
