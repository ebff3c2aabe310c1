// rcclmock.cpp -- TEST INFRASTRUCTURE: a one-rank stand-in for librccl.so.1 (built into a scratch directory that is put in front of
// LD_LIBRARY_PATH), so that brk_exchange_transport / brk_comm_* of libbrk.so run on a CPU under sanitizers next to hipmock.cpp.
// Sends and receives are matched in issue order at ncclGroupEnd, the only peer is rank 0 itself.
#include <rccl/rccl.h> // the real prototypes: UBSan checks the type of a function called through a pointer

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct ncclComm {
  int nranks, rank;
};
typedef ncclComm Comm;
struct Op {
  void *ptr;
  size_t bytes;
};
static thread_local int g_depth = 0;
static thread_local std::vector<Op> g_send, g_recv;
typedef ncclUniqueId Id;

static ncclResult_t flush() {
  if (g_send.size() != g_recv.size()) return ncclInvalidUsage;
  for (size_t q = 0; q < g_send.size(); ++q) {
    if (g_send[q].bytes != g_recv[q].bytes) return ncclInvalidUsage;
    std::memcpy(g_recv[q].ptr, g_send[q].ptr, g_send[q].bytes);
  }
  g_send.clear(), g_recv.clear();
  return ncclSuccess;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0x5a, sizeof *id);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *c, int nranks, ncclUniqueId, int rank) {
  if (nranks != 1 || rank != 0) return ncclInvalidArgument; // the mock has one rank
  *c = new ncclComm{nranks, rank};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) {
  *r = c->rank;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) {
  *n = c->nranks;
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() { return --g_depth == 0 ? flush() : ncclSuccess; }
ncclResult_t ncclSend(const void *p, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t) {
  if (peer != c->rank) return ncclInvalidArgument;
  g_send.push_back({const_cast<void *>(p), count});
  return g_depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void *p, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t) {
  if (peer != c->rank) return ncclInvalidArgument;
  g_recv.push_back({p, count});
  return g_depth ? ncclSuccess : flush();
}
const char *ncclGetErrorString(ncclResult_t e) { return e == ncclSuccess ? "no error" : "rcclmock error"; }
}
