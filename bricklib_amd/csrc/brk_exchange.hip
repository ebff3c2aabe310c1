// brk_exchange.hip -- exchange views: a plan (brk_xplan.cpp) bound to one brick storage on one GPU, executed as
//   begin      one brk_link_copy launch: same-GPU ghost links + packing of everything bound for each peer
//   transport  ncclGroupStart; ncclRecv / ncclSend per message; ncclGroupEnd   (RCCL over xGMI, on the caller's stream)
//   end        one brk_link_copy launch: unpacking
// This is the GPU-resident replacement of BrickDecomp::exchange / ExchangeView::exchange / MultiStageExchangeView::exchange
// (reference include/brick-mpi.h:82-165,466-495) and of the strong driver's pack -> MPI -> unpack (strong/main.cu:367-482).
// librccl is bound on first use: single-GPU users and CPU-only hosts never need it.  ONE RCCL per process: an RCCL that is
// already mapped (PyTorch ships its own librccl.so under torch/lib and maps it at import) is the one this library binds to --
// dlopen(RTLD_NOLOAD) over the candidate names, then dlsym(RTLD_DEFAULT) -- and only when there is none is librccl.so.1 loaded.
// brk_comm_library() says which file answered (dladdr).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// building without the RCCL development header: the few types of its C API this file names (values as in nccl.h); the
// library itself is still only looked for at run time, and brk_comm_* / the transport report BRK_ECOMM when it is absent
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
#endif

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "brk_internal.h"

#define HIP_TRY(expr)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return brk_fail(BRK_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

namespace {

// ---- RCCL entry points, resolved once -----------------------------------------------------------------------------
struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  char why[256] = "";
  char path[512] = ""; // the file ncclCommInitRank was bound from
};

Rccl &rccl_state() {
  static Rccl R;
  return R;
}

Rccl *rccl() {
  Rccl &R = rccl_state();
  static std::once_flag once;
  std::call_once(once, [&R] {
    // (1) an RCCL instance this process has mapped already: same communicator pool, one set of proxy threads, one IPC registry
    static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *name : names) {
      R.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (R.handle) break;
    }
    if (!R.handle && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
      // mapped under another name or path (torch/lib/librccl.so loaded by path): the global scope answers; find its file and
      // take a handle of exactly that object so that every symbol below comes from the same library
      Dl_info info;
      if (dladdr(dlsym(RTLD_DEFAULT, "ncclCommInitRank"), &info) && info.dli_fname) R.handle = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    }
    // (2) none mapped: load one
    for (size_t q = 0; !R.handle && q < sizeof names / sizeof *names; ++q) R.handle = dlopen(names[q], RTLD_NOW | RTLD_GLOBAL);
    if (!R.handle) {
      snprintf(R.why, sizeof R.why, "librccl could not be loaded: %s", dlerror());
      return;
    }
    bool ok = true;
    auto sym = [&](const char *n) {
      void *p = dlsym(R.handle, n);
      if (!p) {
        ok = false;
        snprintf(R.why, sizeof R.why, "librccl lacks %s", n);
      }
      return p;
    };
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
    R.CommUserRank = reinterpret_cast<decltype(R.CommUserRank)>(sym("ncclCommUserRank"));
    R.CommCount = reinterpret_cast<decltype(R.CommCount)>(sym("ncclCommCount"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
    R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
    R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
      dlclose(R.handle);
      R.handle = nullptr;
      return;
    }
    Dl_info info;
    if (dladdr(reinterpret_cast<void *>(R.CommInitRank), &info) && info.dli_fname) snprintf(R.path, sizeof R.path, "%s", info.dli_fname);
  });
  return R.handle ? &R : nullptr;
}

int rccl_missing() { return brk_fail(BRK_ECOMM, "RCCL is not available: %s", rccl_state().why); }

#define NCCL_TRY(R, expr)                                                                                   \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) return brk_fail(BRK_ECOMM, "%s failed: %s", #expr, (R)->GetErrorString(r_));     \
  } while (0)

} // namespace

// ---- views ----------------------------------------------------------------------------------------------------------
struct brk_exchange_view {
  int my_rank = 0, device = 0;
  uint32_t nstages = 0;
  char *storage = nullptr;
  char *arena = nullptr; // every send / receive buffer of the view, 256-B aligned pieces of one allocation
  brk_link *links_dev = nullptr;
  struct Buf {
    char *send = nullptr, *recv = nullptr;
    uint64_t send_bytes = 0, recv_bytes = 0;
  };
  struct Msg {
    int peer;
    bool is_send;
    char *ptr;
    uint64_t bytes;
  };
  struct Stage {
    uint32_t off[2] = {0, 0}, n[2] = {0, 0}; // [phase]: slice of links_dev
    uint64_t max_bytes[2] = {0, 0};
    std::vector<Msg> msgs;
    std::map<int, Buf> buf;
  };
  std::vector<Stage> stages;
};

extern "C" int brk_exchange_view_create(const brk_xplan *p, void *storage_dev, brk_exchange_view **out) {
  if (!p || !storage_dev || !out) return brk_fail(BRK_EINVAL, "brk_exchange_view_create: null argument");
  brk_exchange_view *V = new (std::nothrow) brk_exchange_view();
  if (!V) return brk_fail(BRK_ENOMEM, "brk_exchange_view_create: out of memory");
  auto fail = [&](int rc) {
    brk_exchange_view_free(V);
    return rc;
  };
  V->my_rank = brk_xplan_my_rank(p);
  V->nstages = brk_xplan_nstages(p);
  V->storage = static_cast<char *>(storage_dev);
  V->stages.resize(V->nstages);
  if (hipGetDevice(&V->device) != hipSuccess) return fail(brk_fail(BRK_EHIP, "hipGetDevice failed"));
  const brk_xcopy *copies = brk_xplan_copies(p);
  const brk_xmsg *msgs = brk_xplan_msgs(p);
  const uint32_t nc = brk_xplan_ncopies(p), nm = brk_xplan_nmsgs(p);
  // buffers: every (stage, peer) that a copy or a message names
  uint64_t total = 0;
  auto want = [&](uint32_t stage, int peer) {
    brk_exchange_view::Buf &b = V->stages[stage].buf[peer];
    b.send_bytes = brk_xplan_buffer_bytes(p, stage, peer, BRK_SP_SEND);
    b.recv_bytes = brk_xplan_buffer_bytes(p, stage, peer, BRK_SP_RECV);
  };
  for (uint32_t q = 0; q < nc; ++q) {
    if (copies[q].src_peer >= 0) want(copies[q].stage, copies[q].src_peer);
    if (copies[q].dst_peer >= 0) want(copies[q].stage, copies[q].dst_peer);
  }
  for (uint32_t q = 0; q < nm; ++q)
    if (msgs[q].space != BRK_SP_STORAGE) want(msgs[q].stage, msgs[q].peer);
  auto round = [](uint64_t x) { return (x + 255) / 256 * 256; };
  for (auto &S : V->stages)
    for (auto &kv : S.buf) total += round(kv.second.send_bytes) + round(kv.second.recv_bytes);
  if (total) {
    if (hipMalloc(&V->arena, total) != hipSuccess) return fail(brk_fail(BRK_ENOMEM, "exchange view: %llu bytes of buffers", (unsigned long long)total));
    uint64_t o = 0;
    for (auto &S : V->stages)
      for (auto &kv : S.buf) {
        kv.second.send = kv.second.send_bytes ? V->arena + o : nullptr;
        o += round(kv.second.send_bytes);
        kv.second.recv = kv.second.recv_bytes ? V->arena + o : nullptr;
        o += round(kv.second.recv_bytes);
      }
  }
  auto addr = [&](uint32_t stage, uint32_t space, int peer, uint64_t off) -> char * {
    if (space == BRK_SP_STORAGE) return V->storage + off;
    const brk_exchange_view::Buf &b = V->stages[stage].buf[peer];
    return (space == BRK_SP_SEND ? b.send : b.recv) + off;
  };
  // link tables: per stage, phase 0 then phase 1, plan order kept inside a phase
  std::vector<brk_link> host;
  for (uint32_t st = 0; st < V->nstages; ++st)
    for (uint32_t ph = 0; ph < 2; ++ph) {
      brk_exchange_view::Stage &S = V->stages[st];
      S.off[ph] = (uint32_t)host.size();
      for (uint32_t q = 0; q < nc; ++q)
        if (copies[q].stage == st && copies[q].phase == ph && copies[q].bytes) {
          host.push_back({addr(st, copies[q].src_space, copies[q].src_peer, copies[q].src_off),
                          addr(st, copies[q].dst_space, copies[q].dst_peer, copies[q].dst_off), copies[q].bytes});
          S.max_bytes[ph] = std::max(S.max_bytes[ph], copies[q].bytes);
        }
      S.n[ph] = (uint32_t)host.size() - S.off[ph];
      if (S.n[ph] > 65535) return fail(brk_fail(BRK_EINVAL, "exchange view: %u copies in one phase exceed one launch", S.n[ph]));
    }
  if (!host.empty()) {
    if (hipMalloc(&V->links_dev, host.size() * sizeof(brk_link)) != hipSuccess) return fail(brk_fail(BRK_ENOMEM, "exchange view: link table"));
    if (hipMemcpy(V->links_dev, host.data(), host.size() * sizeof(brk_link), hipMemcpyHostToDevice) != hipSuccess)
      return fail(brk_fail(BRK_EHIP, "exchange view: link table upload failed"));
  }
  for (uint32_t q = 0; q < nm; ++q)
    V->stages[msgs[q].stage].msgs.push_back({msgs[q].peer, msgs[q].is_send != 0, addr(msgs[q].stage, msgs[q].space, msgs[q].peer, msgs[q].off), msgs[q].bytes});
  *out = V;
  return BRK_OK;
}

extern "C" void brk_exchange_view_free(brk_exchange_view *v) {
  if (!v) return;
  int cur = 0;
  const bool sw = hipGetDevice(&cur) == hipSuccess && cur != v->device && hipSetDevice(v->device) == hipSuccess;
  if (v->arena) (void)hipFree(v->arena);
  if (v->links_dev) (void)hipFree(v->links_dev);
  if (sw) (void)hipSetDevice(cur);
  delete v;
}

// a view's buffers and link tables live on the device it was created on
static int check_device(const brk_exchange_view *v) {
  int cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (cur != v->device) return brk_fail(BRK_EINVAL, "exchange: view belongs to device %d, current device is %d", v->device, cur);
  return BRK_OK;
}

static int run_phase(brk_exchange_view *v, uint32_t stage, int ph, brk_stream s) {
  if (!v || stage >= v->nstages) return brk_fail(BRK_EINVAL, "exchange: bad view / stage");
  if (int rc = check_device(v)) return rc;
  const brk_exchange_view::Stage &S = v->stages[stage];
  if (!S.n[ph]) return BRK_OK;
  return brk_link_copy(v->links_dev + S.off[ph], S.n[ph], S.max_bytes[ph], s);
}

extern "C" int brk_exchange_begin(brk_exchange_view *v, uint32_t stage, brk_stream s) { return run_phase(v, stage, 0, s); }
extern "C" int brk_exchange_end(brk_exchange_view *v, uint32_t stage, brk_stream s) { return run_phase(v, stage, 1, s); }

extern "C" int brk_exchange_transport(brk_exchange_view *v, uint32_t stage, void *comm, brk_stream s) {
  if (!v || stage >= v->nstages) return brk_fail(BRK_EINVAL, "exchange: bad view / stage");
  const brk_exchange_view::Stage &S = v->stages[stage];
  if (S.msgs.empty()) return BRK_OK;
  if (!comm) return brk_fail(BRK_EINVAL, "exchange: %zu messages to move but no communicator", S.msgs.size());
  if (int rc = check_device(v)) return rc;
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  hipStream_t st = static_cast<hipStream_t>(s);
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  NCCL_TRY(R, R->GroupStart());
  ncclResult_t bad = ncclSuccess;
  for (const brk_exchange_view::Msg &m : S.msgs) {
    const ncclResult_t r = m.is_send ? R->Send(m.ptr, m.bytes, ncclUint8, m.peer, c, st) : R->Recv(m.ptr, m.bytes, ncclUint8, m.peer, c, st);
    if (r != ncclSuccess && bad == ncclSuccess) bad = r;
  }
  const ncclResult_t ge = R->GroupEnd(); // always close the group, also after a failed call inside it
  if (bad != ncclSuccess) return brk_fail(BRK_ECOMM, "ncclSend/ncclRecv failed: %s", R->GetErrorString(bad));
  NCCL_TRY(R, ge);
  return BRK_OK;
}

extern "C" int brk_exchange_run(brk_exchange_view *v, void *comm, brk_stream s) {
  if (!v) return brk_fail(BRK_EINVAL, "exchange: null view");
  for (uint32_t st = 0; st < v->nstages; ++st) {
    if (int rc = brk_exchange_begin(v, st, s)) return rc;
    if (int rc = brk_exchange_transport(v, st, comm, s)) return rc;
    if (int rc = brk_exchange_end(v, st, s)) return rc;
  }
  return BRK_OK;
}

// Same-GPU transport between views: the k-th send of views[a] to rank(views[b]) in `stage` pairs with the k-th receive of
// views[b] from rank(views[a]) -- the matching rule of the grouped ncclSend/ncclRecv calls.
extern "C" int brk_exchange_loopback(brk_exchange_view *const *views, uint32_t nviews, uint32_t stage, brk_stream s) {
  if (!views || !nviews) return brk_fail(BRK_EINVAL, "loopback: no views");
  std::map<int, brk_exchange_view *> by_rank;
  for (uint32_t a = 0; a < nviews; ++a) {
    if (!views[a] || stage >= views[a]->nstages) return brk_fail(BRK_EINVAL, "loopback: bad view / stage");
    if (!by_rank.emplace(views[a]->my_rank, views[a]).second) return brk_fail(BRK_EINVAL, "loopback: two views of rank %d", views[a]->my_rank);
  }
  hipStream_t st = static_cast<hipStream_t>(s);
  for (uint32_t a = 0; a < nviews; ++a) {
    std::map<int, size_t> next; // per destination: receives of that view from us consumed so far
    for (const brk_exchange_view::Msg &m : views[a]->stages[stage].msgs) {
      if (!m.is_send) continue;
      auto it = by_rank.find(m.peer);
      if (it == by_rank.end()) return brk_fail(BRK_EINVAL, "loopback: rank %d sends to rank %d, which has no view here", views[a]->my_rank, m.peer);
      const auto &rm = it->second->stages[stage].msgs;
      size_t &k = next[m.peer];
      while (k < rm.size() && (rm[k].is_send || rm[k].peer != views[a]->my_rank)) ++k;
      if (k == rm.size() || rm[k].bytes != m.bytes)
        return brk_fail(BRK_EINVAL, "loopback: send of %llu bytes from rank %d has no matching receive on rank %d",
                        (unsigned long long)m.bytes, views[a]->my_rank, m.peer);
      HIP_TRY(hipMemcpyAsync(rm[k].ptr, m.ptr, m.bytes, hipMemcpyDeviceToDevice, st));
      ++k;
    }
  }
  return BRK_OK;
}

extern "C" void *brk_exchange_view_buffer(const brk_exchange_view *v, uint32_t stage, int peer, uint32_t which) {
  if (!v || stage >= v->nstages) return nullptr;
  auto it = v->stages[stage].buf.find(peer);
  if (it == v->stages[stage].buf.end()) return nullptr;
  return which == BRK_SP_SEND ? it->second.send : (which == BRK_SP_RECV ? it->second.recv : nullptr);
}

// ---- the reference-shaped call ---------------------------------------------------------------------------------------
extern "C" int brk_exchange(brk_decomp *d, double *storage_dev, size_t step_elems, void *comm, const int ranks[27], brk_stream s) {
  if (!d || !storage_dev || !ranks) return brk_fail(BRK_EINVAL, "brk_exchange: null argument");
  int me = 0;
  if (comm) {
    Rccl *R = rccl();
    if (!R) return rccl_missing();
    NCCL_TRY(R, R->CommUserRank(static_cast<ncclComm_t>(comm), &me));
  } else {
    me = ranks[13]; // no communicator: legal only when every neighbour is this rank (index 13 = self)
  }
  brk_exchange_view *view = nullptr;
  {
    std::lock_guard<std::mutex> lock(d->xmu);
    for (const brk_xcache_entry &e : d->xcache)
      if (e.storage == storage_dev && e.comm == comm && e.step == step_elems && std::memcmp(e.ranks, ranks, sizeof e.ranks) == 0) view = e.view;
    if (!view) {
      brk_xcache_entry e{};
      e.storage = storage_dev, e.comm = comm, e.step = step_elems;
      std::memcpy(e.ranks, ranks, sizeof e.ranks);
      if (int rc = brk_xplan_build_decomp(d, ranks, me, BRK_XCHG_PACKED, 0, step_elems, &e.plan)) return rc;
      if (int rc = brk_exchange_view_create(e.plan, storage_dev, &e.view)) {
        brk_xplan_free(e.plan);
        return rc;
      }
      d->xcache.push_back(e);
      view = e.view;
    }
  }
  return brk_exchange_run(view, comm, s);
}

// ---- communicators ---------------------------------------------------------------------------------------------------
static_assert(sizeof(ncclUniqueId) == 128, "brk.h promises a 128-byte id");

extern "C" int brk_comm_unique_id(void *id128) {
  if (!id128) return brk_fail(BRK_EINVAL, "brk_comm_unique_id: null");
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  ncclUniqueId id;
  NCCL_TRY(R, R->GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof id);
  return BRK_OK;
}

extern "C" int brk_comm_init_rank(void **comm, int nranks, const void *id128, int rank) {
  if (!comm || !id128 || nranks <= 0 || rank < 0 || rank >= nranks) return brk_fail(BRK_EINVAL, "brk_comm_init_rank: bad arguments");
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  ncclComm_t c = nullptr;
  NCCL_TRY(R, R->CommInitRank(&c, nranks, id, rank));
  *comm = c;
  return BRK_OK;
}

extern "C" int brk_comm_destroy(void *comm) {
  if (!comm) return BRK_OK;
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  NCCL_TRY(R, R->CommDestroy(static_cast<ncclComm_t>(comm)));
  return BRK_OK;
}

// the RCCL library the transport is bound to (file name as dladdr reports it), or NULL + error when none can be bound
extern "C" const char *brk_comm_library(void) {
  Rccl *R = rccl();
  if (!R) {
    (void)rccl_missing();
    return nullptr;
  }
  return R->path;
}

extern "C" int brk_comm_rank(void *comm, int *rank, int *nranks) {
  if (!comm) return brk_fail(BRK_EINVAL, "brk_comm_rank: null communicator");
  Rccl *R = rccl();
  if (!R) return rccl_missing();
  if (rank) NCCL_TRY(R, R->CommUserRank(static_cast<ncclComm_t>(comm), rank));
  if (nranks) NCCL_TRY(R, R->CommCount(static_cast<ncclComm_t>(comm), nranks));
  return BRK_OK;
}
