// brk_xplan.cpp -- exchange plans: the host-side bookkeeping of the ghost-zone exchange (include/brk.h, "ghost-zone
// exchange").  A plan says which byte ranges are copied where before the transport (same-GPU links, packing), which
// messages travel, and which copies follow (unpacking).  The rules come from the reference -- ghost run i <- skin run i of
// the rank behind ghost[i].neighbor, message order = run order (include/brick-mpi.h:466-495); the three-stage shift
// exchange (:587-681) -- the code is written fresh.  Nothing here touches the GPU: views (brk_exchange.hip) execute plans
// on the device, the Python CPU tests interpret the same tables over gloo.
#include "brk_internal.h"

#include <algorithm>
#include <map>
#include <new>
#include <set>
#include <utility>
#include <vector>

using namespace brk_dir;

struct brk_xplan {
  int my_rank = 0;
  uint32_t nstages = 0;
  std::vector<brk_xcopy> copies;
  std::vector<brk_xmsg> msgs;
  // (stage, peer) -> {send bytes, recv bytes}
  std::map<std::pair<uint32_t, int>, std::pair<uint64_t, uint64_t>> buf;
  uint64_t bytes_remote = 0, bytes_local = 0;
};

namespace {

// sizes the per-peer buffers from the tables and validates them
int finish(brk_xplan *P) {
  P->nstages = 0;
  P->buf.clear();
  P->bytes_remote = P->bytes_local = 0;
  auto grow = [&](uint32_t stage, int peer, uint32_t space, uint64_t end) {
    auto &b = P->buf[{stage, peer}];
    uint64_t &x = space == BRK_SP_SEND ? b.first : b.second;
    x = std::max(x, end);
  };
  for (const brk_xcopy &c : P->copies) {
    if (c.phase > 1 || c.src_space > BRK_SP_RECV || c.dst_space > BRK_SP_RECV)
      return brk_fail(BRK_EINVAL, "exchange plan: bad phase/space in a copy");
    if (c.bytes % 16 || c.src_off % 16 || c.dst_off % 16)
      return brk_fail(BRK_EINVAL, "exchange plan: copy offsets and sizes must be multiples of 16 bytes");
    if ((c.src_space == BRK_SP_STORAGE) != (c.src_peer < 0) || (c.dst_space == BRK_SP_STORAGE) != (c.dst_peer < 0))
      return brk_fail(BRK_EINVAL, "exchange plan: a peer must be named for buffer spaces and only for them");
    if (c.dst_space == BRK_SP_RECV && c.phase == 0)
      return brk_fail(BRK_EINVAL, "exchange plan: a begin-phase copy cannot write a receive buffer");
    P->nstages = std::max(P->nstages, c.stage + 1);
    if (c.src_space != BRK_SP_STORAGE) grow(c.stage, c.src_peer, c.src_space, c.src_off + c.bytes);
    if (c.dst_space != BRK_SP_STORAGE) grow(c.stage, c.dst_peer, c.dst_space, c.dst_off + c.bytes);
    P->bytes_local += c.bytes;
  }
  for (const brk_xmsg &m : P->msgs) {
    if (m.peer < 0 || m.space > BRK_SP_RECV || m.is_send > 1)
      return brk_fail(BRK_EINVAL, "exchange plan: bad message");
    if ((m.space == BRK_SP_SEND && !m.is_send) || (m.space == BRK_SP_RECV && m.is_send))
      return brk_fail(BRK_EINVAL, "exchange plan: sends leave from send buffers, receives land in receive buffers");
    if (m.bytes == 0 || m.bytes % 16 || m.off % 16)
      return brk_fail(BRK_EINVAL, "exchange plan: message offsets and sizes must be non-zero multiples of 16 bytes");
    P->nstages = std::max(P->nstages, m.stage + 1);
    if (m.space != BRK_SP_STORAGE) grow(m.stage, m.peer, m.space, m.off + m.bytes);
    if (m.is_send) P->bytes_remote += m.bytes;
  }
  return BRK_OK;
}

int index_of_set(const std::array<uint64_t, 27> &nbs, uint64_t set) {
  for (int q = 0; q < 27; ++q)
    if (nbs[q] == set) return q;
  return -1;
}

// ghost run q <- skin run q of the neighbour: all runs at once (packed or direct)
int build_runs(const brk_decomp *d, const int ranks[27], int me, int mode, unsigned flags, uint64_t brick_bytes, brk_xplan *P) {
  std::array<uint64_t, 27> nbs;
  neighbors3(nbs);
  const size_t nr = d->ghost.size();
  std::vector<int> src(nr), dst(nr);
  for (size_t q = 0; q < nr; ++q) {
    const int gi = index_of_set(nbs, d->ghost[q].neighbor), si = index_of_set(nbs, d->skin[q].neighbor);
    if (gi < 0 || si < 0) return brk_fail(BRK_EINVAL, "exchange plan: run %zu names an unknown neighbour", q);
    src[q] = ranks[gi];
    dst[q] = ranks[si];
    if (src[q] < 0 || dst[q] < 0) return brk_fail(BRK_EINVAL, "exchange plan: negative rank for run %zu", q);
  }
  const bool via_comm = (flags & BRK_XCHG_SELF_VIA_COMM) != 0;
  auto remote = [&](int r) { return r != me || via_comm; };
  // same-rank runs: my skin run q IS the neighbour's skin run q
  for (size_t q = 0; q < nr; ++q)
    if (!remote(src[q]))
      P->copies.push_back({0, 0, BRK_SP_STORAGE, BRK_SP_STORAGE, -1, -1, (uint64_t)d->skin[q].pos * brick_bytes,
                           (uint64_t)d->ghost[q].pos * brick_bytes, (uint64_t)d->ghost[q].len * brick_bytes});
  if (mode == BRK_XCHG_DIRECT) {
    // one message per run, the same run order on every rank: pairs match by issue order per peer
    // (a run can be empty -- a section with no bricks, e.g. an axis exactly two ghost depths wide under skin3d_normal / _bad --
    //  and then sends nothing: ghost[q].len == skin[q].len on every rank, so both sides skip it)
    for (size_t q = 0; q < nr; ++q) {
      if (d->ghost[q].len == 0) continue;
      if (remote(src[q]))
        P->msgs.push_back({0, 0, src[q], BRK_SP_STORAGE, (uint64_t)d->ghost[q].pos * brick_bytes, (uint64_t)d->ghost[q].len * brick_bytes});
      if (remote(dst[q]))
        P->msgs.push_back({0, 1, dst[q], BRK_SP_STORAGE, (uint64_t)d->skin[q].pos * brick_bytes, (uint64_t)d->skin[q].len * brick_bytes});
    }
    return BRK_OK;
  }
  std::set<int> peers;
  for (size_t q = 0; q < nr; ++q) {
    if (remote(src[q])) peers.insert(src[q]);
    if (remote(dst[q])) peers.insert(dst[q]);
  }
  for (int p : peers) {
    uint64_t so = 0, ro = 0;
    for (size_t q = 0; q < nr; ++q) { // run-index order on both sides: the buffers need no header
      if (remote(dst[q]) && dst[q] == p) {
        const uint64_t n = (uint64_t)d->skin[q].len * brick_bytes;
        P->copies.push_back({0, 0, BRK_SP_STORAGE, BRK_SP_SEND, -1, p, (uint64_t)d->skin[q].pos * brick_bytes, so, n});
        so += n;
      }
      if (remote(src[q]) && src[q] == p) {
        const uint64_t n = (uint64_t)d->ghost[q].len * brick_bytes;
        P->copies.push_back({0, 1, BRK_SP_RECV, BRK_SP_STORAGE, p, -1, ro, (uint64_t)d->ghost[q].pos * brick_bytes, n});
        ro += n;
      }
    }
    if (ro) P->msgs.push_back({0, 0, p, BRK_SP_RECV, 0, ro});
    if (so) P->msgs.push_back({0, 1, p, BRK_SP_SEND, 0, so});
  }
  return BRK_OK;
}

// maximal pieces of two equally long id sequences that advance by one on both sides: (src id, dst id, bricks)
struct Piece {
  uint32_t src, dst, len;
};
std::vector<Piece> pieces(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) {
  std::vector<Piece> out;
  for (size_t q = 0; q < a.size(); ++q) {
    if (!out.empty() && a[q] == out.back().src + out.back().len && b[q] == out.back().dst + out.back().len)
      ++out.back().len;
    else
      out.push_back({a[q], b[q], 1});
  }
  return out;
}

// three-stage shift exchange: in stage a every rank trades its two boundary layers along axis a with its -a / +a
// neighbours; the layers span the ghosts already received along the earlier axes, so edge and corner ghosts arrive in
// two / three hops (MultiStageExchangeView, include/brick-mpi.h:587-681)
int build_shift(const brk_decomp *d, const int ranks[27], int me, unsigned flags, uint64_t brick_bytes, brk_xplan *P) {
  std::array<uint64_t, 27> nbs;
  neighbors3(nbs);
  const bool via_comm = (flags & BRK_XCHG_SELF_VIA_COMM) != 0;
  const uint32_t *T = d->tdims;
  auto box_ids = [&](int a, uint32_t lo_a, uint32_t hi_a) {
    uint32_t lo[3] = {0, 0, 0}, hi[3] = {T[0], T[1], T[2]};
    for (int b = 0; b < 3; ++b)
      if (b > a) lo[b] = d->depth[b], hi[b] = T[b] - d->depth[b]; // not exchanged yet: the domain only
    lo[a] = lo_a, hi[a] = hi_a;
    std::vector<uint32_t> ids;
    for (uint32_t k = lo[2]; k < hi[2]; ++k)
      for (uint32_t j = lo[1]; j < hi[1]; ++j)
        for (uint32_t i = lo[0]; i < hi[0]; ++i) ids.push_back(d->grid[(size_t)k * d->stride[2] + (size_t)j * d->stride[1] + i]);
    return ids;
  };
  for (int a = 0; a < 3; ++a) {
    const uint32_t gb = d->depth[a];
    const int qm = index_of_set(nbs, dir_bit(-(a + 1))), qp = index_of_set(nbs, dir_bit(a + 1));
    const int minus = ranks[qm], plus = ranks[qp];
    if (minus < 0 || plus < 0) return brk_fail(BRK_EINVAL, "exchange plan: negative rank for a face neighbour");
    struct Layer {
      std::vector<uint32_t> ids;
      int peer;
    };
    // my low layer is what my -a neighbour receives as its high ghost, and the other way round
    Layer send[2] = {{box_ids(a, gb, 2 * gb), minus}, {box_ids(a, T[a] - 2 * gb, T[a] - gb), plus}};
    Layer recv[2] = {{box_ids(a, T[a] - gb, T[a]), plus}, {box_ids(a, 0, gb), minus}};
    std::map<int, uint64_t> so, ro;
    for (int l = 0; l < 2; ++l) {
      if (send[l].peer == me && !via_comm) { // the neighbour is this rank: link the layer straight into the ghost
        for (const Piece &pc : pieces(send[l].ids, recv[l].ids))
          P->copies.push_back({(uint32_t)a, 0, BRK_SP_STORAGE, BRK_SP_STORAGE, -1, -1, (uint64_t)pc.src * brick_bytes,
                               (uint64_t)pc.dst * brick_bytes, (uint64_t)pc.len * brick_bytes});
        continue;
      }
      const int p = send[l].peer;
      uint32_t start = 0;
      std::vector<uint32_t> seq(send[l].ids.size());
      for (size_t q = 0; q < seq.size(); ++q) seq[q] = start + (uint32_t)q;
      for (const Piece &pc : pieces(send[l].ids, seq)) {
        P->copies.push_back({(uint32_t)a, 0, BRK_SP_STORAGE, BRK_SP_SEND, -1, p, (uint64_t)pc.src * brick_bytes, so[p],
                             (uint64_t)pc.len * brick_bytes});
        so[p] += (uint64_t)pc.len * brick_bytes;
      }
    }
    for (int l = 0; l < 2; ++l) {
      if (recv[l].peer == me && !via_comm) continue;
      const int p = recv[l].peer;
      std::vector<uint32_t> seq(recv[l].ids.size());
      for (size_t q = 0; q < seq.size(); ++q) seq[q] = (uint32_t)q;
      for (const Piece &pc : pieces(recv[l].ids, seq)) {
        P->copies.push_back({(uint32_t)a, 1, BRK_SP_RECV, BRK_SP_STORAGE, p, -1, ro[p], (uint64_t)pc.src * brick_bytes,
                             (uint64_t)pc.len * brick_bytes});
        ro[p] += (uint64_t)pc.len * brick_bytes;
      }
    }
    std::set<int> peers;
    for (auto &kv : so) peers.insert(kv.first);
    for (auto &kv : ro) peers.insert(kv.first);
    for (int p : peers) {
      if (ro.count(p)) P->msgs.push_back({(uint32_t)a, 0, p, BRK_SP_RECV, 0, ro[p]});
      if (so.count(p)) P->msgs.push_back({(uint32_t)a, 1, p, BRK_SP_SEND, 0, so[p]});
    }
  }
  return BRK_OK;
}

} // namespace

extern "C" int brk_xplan_build_decomp(const brk_decomp *d, const int ranks[27], int my_rank, int mode, unsigned flags,
                                      size_t step_elems, brk_xplan **out) {
  if (!d || !ranks || !out) return brk_fail(BRK_EINVAL, "brk_xplan_build_decomp: null argument");
  if (step_elems < 512 || step_elems % 2) return brk_fail(BRK_EINVAL, "brk_xplan_build_decomp: step_elems %zu", step_elems);
  if (mode != BRK_XCHG_PACKED && mode != BRK_XCHG_DIRECT && mode != BRK_XCHG_SHIFT)
    return brk_fail(BRK_EINVAL, "brk_xplan_build_decomp: unknown mode %d", mode);
  if (my_rank < 0) return brk_fail(BRK_EINVAL, "brk_xplan_build_decomp: negative rank");
  brk_xplan *P = new (std::nothrow) brk_xplan();
  if (!P) return brk_fail(BRK_ENOMEM, "brk_xplan_build_decomp: out of memory");
  P->my_rank = my_rank;
  const uint64_t brick_bytes = (uint64_t)step_elems * sizeof(double);
  int rc = mode == BRK_XCHG_SHIFT ? build_shift(d, ranks, my_rank, flags, brick_bytes, P)
                                  : build_runs(d, ranks, my_rank, mode, flags, brick_bytes, P);
  if (rc == BRK_OK) rc = finish(P);
  if (rc != BRK_OK) {
    delete P;
    return rc;
  }
  if (mode == BRK_XCHG_SHIFT) P->nstages = 3;
  if (P->nstages == 0) P->nstages = 1;
  *out = P;
  return BRK_OK;
}

extern "C" int brk_xplan_build_tables(int my_rank, uint32_t ncopies, const brk_xcopy *copies, uint32_t nmsgs, const brk_xmsg *msgs,
                                      brk_xplan **out) {
  if (!out || (ncopies && !copies) || (nmsgs && !msgs) || my_rank < 0) return brk_fail(BRK_EINVAL, "brk_xplan_build_tables: bad arguments");
  brk_xplan *P = new (std::nothrow) brk_xplan();
  if (!P) return brk_fail(BRK_ENOMEM, "brk_xplan_build_tables: out of memory");
  P->my_rank = my_rank;
  P->copies.assign(copies, copies + ncopies);
  P->msgs.assign(msgs, msgs + nmsgs);
  const int rc = finish(P);
  if (rc != BRK_OK) {
    delete P;
    return rc;
  }
  if (P->nstages == 0) P->nstages = 1;
  *out = P;
  return BRK_OK;
}

extern "C" void brk_xplan_free(brk_xplan *p) { delete p; }
extern "C" int brk_xplan_my_rank(const brk_xplan *p) { return p->my_rank; }
extern "C" uint32_t brk_xplan_nstages(const brk_xplan *p) { return p->nstages; }
extern "C" uint32_t brk_xplan_ncopies(const brk_xplan *p) { return (uint32_t)p->copies.size(); }
extern "C" const brk_xcopy *brk_xplan_copies(const brk_xplan *p) { return p->copies.data(); }
extern "C" uint32_t brk_xplan_nmsgs(const brk_xplan *p) { return (uint32_t)p->msgs.size(); }
extern "C" const brk_xmsg *brk_xplan_msgs(const brk_xplan *p) { return p->msgs.data(); }
extern "C" uint64_t brk_xplan_bytes_remote(const brk_xplan *p) { return p->bytes_remote; }
extern "C" uint64_t brk_xplan_bytes_local(const brk_xplan *p) { return p->bytes_local; }
extern "C" uint64_t brk_xplan_buffer_bytes(const brk_xplan *p, uint32_t stage, int peer, uint32_t which) {
  auto it = p->buf.find({stage, peer});
  if (it == p->buf.end()) return 0;
  return which == BRK_SP_SEND ? it->second.first : (which == BRK_SP_RECV ? it->second.second : 0);
}
