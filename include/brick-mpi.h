// brick-mpi.h -- BrickDecomp and friends (reference include/brick-mpi.h:178-753, src/brick-mpi.cpp:9-52) on top of the
// layout builder of libbrk.so (brk_decomp_*): region-ordered brick numbering, ghost / skin run tables, adjacency, the
// neighbour -> rank map, and the ghost-zone exchange:
//   * ExchangeView / MultiStageExchangeView (reference :82-165, created by exchangeView / multiStageExchangeView :525-681)
//     and BrickDecomp::exchange(bStorage) (:466-495) on DEVICE storages: GPU-resident pack -> ncclSend/ncclRecv in one
//     group -> unpack, all inside libbrk.so (brk_exchange_*), on the communicator in BrickDecomp::rccl_comm;
//   * when an MPI implementation is present at compile time and no RCCL communicator has been set: the reference's
//     run-by-run exchange over MPI (host storages, or device pointers with a GPU-aware MPI as weak/main.cu does).
// Provided: 3 dimensions, 8x8x8 bricks (what the reference's drivers use), any order of the surface sections (skin3d_good / _normal / _bad or a list of one's own).
#pragma once

#include <ostream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "bitset.h"
#include "brick.h"
#include "brk.h"

#if __has_include(<mpi.h>) && !defined(BRK_NO_MPI)
#include <mpi.h>
#define BRK_HAVE_MPI 1
#endif

// The exchange of one storage as an object (reference ExchangeView :82-123 and MultiStageExchangeView :131-165: built once,
// exchange() every time step).  Here it owns an exchange plan and a view of libbrk.so: per-peer buffers and link tables on the
// GPU; exchange() enqueues begin (links + pack) -> ncclSend/ncclRecv group -> end (unpack) for every stage on `stream`.
struct ExchangeView {
  brk_xplan *plan = nullptr;
  brk_exchange_view *view = nullptr;
  void *rccl_comm = nullptr;  // ncclComm_t; may be null when every neighbour is this rank
  void *stream = nullptr;     // hipStream_t the exchange is enqueued on (null stream by default, like the reference's drivers)
  std::vector<size_t> seclen; // bytes of every message this rank sends (reference member of the same name)

  ExchangeView() = default;
  ExchangeView(const ExchangeView &) = delete;
  ExchangeView &operator=(const ExchangeView &) = delete;
  ExchangeView(ExchangeView &&o) noexcept { *this = std::move(o); }
  ExchangeView &operator=(ExchangeView &&o) noexcept {
    std::swap(plan, o.plan), std::swap(view, o.view), std::swap(rccl_comm, o.rccl_comm), std::swap(stream, o.stream);
    seclen.swap(o.seclen);
    return *this;
  }
  ~ExchangeView() {
    brk_exchange_view_free(view);
    brk_xplan_free(plan);
  }
  void exchange() {
    if (brk_exchange_run(view, rccl_comm, stream) != BRK_OK) throw std::runtime_error(brk_last_error());
  }
};
typedef ExchangeView MultiStageExchangeView; // the three-stage form is the same object over a BRK_XCHG_SHIFT plan

// timing accumulators the reference's drivers print (src/brick-mpi.cpp:7); bricktime.h's time_mpi zeroes them when it sees this macro
#define BRK_HAVE_TIMERS 1
inline double packtime = 0, calltime = 0, waittime = 0, movetime = 0, calctime = 0;

// all 3^dim subsets of {+-1..+-dim} with at most one sign per axis, in the reference's order (src/brick-mpi.cpp:9-23);
// 3-D only: the order comes from the library
inline void allneighbors(BitSet cur, long idx, long dim, std::vector<BitSet> &neighbors) {
  if (dim != 3 || idx != 1 || cur.set != 0) throw std::runtime_error("allneighbors: only (0, 1, 3, out) is provided");
  uint64_t raw[27];
  brk_allneighbors3(raw);
  for (uint64_t r : raw) neighbors.emplace_back(r);
}

template <unsigned dim, unsigned... BDims>
class BrickDecomp {
  static_assert(dim == 3 && sizeof...(BDims) == 3, "BrickDecomp<3, 8, 8, 8> is the provided configuration");

public:
  struct g_region {
    BitSet neighbor;                                      // direction set of the neighbour this run talks to
    unsigned skin_st, skin_ed, pos, len, first_pad, last_pad;
  };
  std::vector<g_region> ghost, skin;                      // run i: ghost[i] <- skin[i] of rank_map[ghost[i].neighbor]
  unsigned sep_pos[3];                                    // [1,sep0) inner, [sep0,sep1) surface, [sep1,sep2) ghost
  std::vector<BitSet> skinlist;
  std::vector<long> skin_size;
  std::vector<unsigned> dims, t_dims, g_depth, stride;    // in bricks: domain, domain + ghost, ghost depth, row/plane strides
#ifdef BRK_HAVE_MPI
  MPI_Comm comm = MPI_COMM_NULL;
#endif
  std::unordered_map<uint64_t, int> rank_map;             // BitSet::set -> rank, filled by populate()
  void *rccl_comm = nullptr;                              // ncclComm_t for the device exchange (brkCommFromMPI / brk_comm_init_rank)

  BrickDecomp(const std::vector<unsigned> &dims_cells, unsigned depth_cells, unsigned numfield = 1) : numfield_(numfield) {
    constexpr unsigned bd[3] = {BDims...};
    if (bd[0] != 8 || bd[1] != 8 || bd[2] != 8) throw std::runtime_error("BrickDecomp: 8x8x8 bricks only");
    if (dims_cells.size() != 3) throw std::runtime_error("BrickDecomp: three extents expected");
    for (int a = 0; a < 3; ++a) cells_[a] = dims_cells[a];
    depth_ = depth_cells;
  }
  BrickDecomp(const BrickDecomp &) = delete;
  BrickDecomp &operator=(const BrickDecomp &) = delete;
  ~BrickDecomp() { brk_decomp_free(h_); }

  // the order of the surface sections (reference :322): skin3d_good (recommended), skin3d_normal, skin3d_bad or any list in
  // which each of the 26 sections appears once (empty sets allowed as run separators)
  void initialize(const std::vector<BitSet> &order) {
    std::vector<uint64_t> raw;
    for (const BitSet &b : order) raw.push_back(b.set);
    if (brk_decomp_build_order(cells_, depth_, raw.data(), (uint32_t)raw.size(), &h_) != BRK_OK) throw std::runtime_error(brk_last_error());
    const uint32_t ns = brk_decomp_nskin(h_);
    skinlist = order;
    const int64_t *ss = brk_decomp_skin_size(h_);
    skin_size.assign(ss, ss + ns);
    uint32_t sp[3], td[3];
    brk_decomp_sep_pos(h_, sp);
    brk_decomp_tdims(h_, td);
    for (int a = 0; a < 3; ++a) sep_pos[a] = sp[a];
    t_dims.assign(td, td + 3);
    dims.resize(3);
    g_depth.assign(3, depth_ / 8);
    stride.resize(3);
    for (int a = 0; a < 3; ++a) dims[a] = cells_[a] / 8;
    stride[0] = 1, stride[1] = td[0], stride[2] = td[0] * td[1];
    const uint32_t nr = brk_decomp_nruns(h_);
    auto conv = [](const brk_region &r) { return g_region{BitSet(r.neighbor), r.skin_st, r.skin_ed, r.pos, r.len, r.first_pad, r.last_pad}; };
    const brk_region *g = brk_decomp_ghost(h_), *k = brk_decomp_skin(h_);
    for (uint32_t q = 0; q < nr; ++q) ghost.push_back(conv(g[q])), skin.push_back(conv(k[q]));
  }

  // logical brick coordinates (ghost included) -> brick id: bDecomp[k][j][i]
  struct Row {
    const uint32_t *p;
    unsigned operator[](int i) const { return p[i]; }
  };
  struct Plane {
    const uint32_t *p;
    unsigned t0;
    Row operator[](int j) const { return Row{p + (size_t)j * t0}; }
  };
  Plane operator[](int k) const { return Plane{brk_decomp_grid(h_) + (size_t)k * t_dims[0] * t_dims[1], t_dims[0]}; }

  // adjacency of every brick; the caller frees .adj as with init_grid (docs/usingbricks.md:42-47 of the reference)
  BrickInfo<dim> getBrickInfo() const {
    BrickInfo<dim> info(brk_decomp_nbricks(h_));
    const uint32_t *a = brk_decomp_adj(h_);
    unsigned *dst = &info.adj[0][0];
    for (size_t q = 0; q < (size_t)info.nbricks * 27; ++q) dst[q] = a[q];
    return info;
  }

  unsigned nbricks() const { return brk_decomp_nbricks(h_); }
  const brk_decomp *handle() const { return h_; }

  // neighbour ranks in the order the C ABI wants them (brk_allneighbors3)
  void ranks27(int out[27]) const {
    uint64_t nb[27];
    brk_allneighbors3(nb);
    for (int q = 0; q < 27; ++q) {
      auto it = rank_map.find(nb[q]);
      if (it == rank_map.end()) throw std::runtime_error("BrickDecomp: populate() has not filled rank_map");
      out[q] = it->second;
    }
  }

  // the communicator's idea of this rank must be the rank map's (ranks[13] = self): a communicator made from another MPI
  // communicator than the one populate() mapped (a reordered cart vs MPI_COMM_WORLD) would send to the wrong peers silently
  void check_comm_rank(const int ranks[27]) const {
    if (!rccl_comm) return;
    int r = -1;
    if (brk_comm_rank(rccl_comm, &r, nullptr) != BRK_OK) throw std::runtime_error(brk_last_error());
    if (r != ranks[13])
      throw std::runtime_error("BrickDecomp: rccl_comm says rank " + std::to_string(r) + ", populate() mapped this rank as " +
                               std::to_string(ranks[13]) + " -- make the communicator from the communicator populate() was given");
  }

  // exchangeView(bStorage) of the reference (:525-585): the exchange of one DEVICE storage as an object.  mode: BRK_XCHG_PACKED
  // (one message per peer, default), BRK_XCHG_DIRECT (one per run, the reference's message pattern), BRK_XCHG_SHIFT.
  // flags: BRK_XCHG_SELF_VIA_COMM sends runs whose peer is this rank through the communicator too (single-GPU tests of the transport).
  ExchangeView exchangeView(const BrickStorage &bStorage_dev, int mode = BRK_XCHG_PACKED, unsigned flags = 0) const {
    int ranks[27];
    ranks27(ranks);
    check_comm_rank(ranks);
    ExchangeView ev;
    ev.rccl_comm = rccl_comm;
    if (brk_xplan_build_decomp(h_, ranks, ranks[13], mode, flags, bStorage_dev.step, &ev.plan) != BRK_OK ||
        brk_exchange_view_create(ev.plan, bStorage_dev.dat.get(), &ev.view) != BRK_OK)
      throw std::runtime_error(brk_last_error());
    const brk_xmsg *m = brk_xplan_msgs(ev.plan);
    for (uint32_t q = 0; q < brk_xplan_nmsgs(ev.plan); ++q)
      if (m[q].is_send) ev.seclen.push_back((size_t)m[q].bytes);
    return ev;
  }
  // multiStageExchangeView(bStorage) of the reference (:587-681): three stages of <= 2 messages
  MultiStageExchangeView multiStageExchangeView(const BrickStorage &bStorage_dev) const { return exchangeView(bStorage_dev, BRK_XCHG_SHIFT); }

  // bDecomp.exchange(bStorage) on a DEVICE storage through libbrk.so: pack -> RCCL -> unpack (plan and view are cached in the
  // library per storage).  Without a communicator this is legal only when every neighbour is this rank (1-rank periodic runs).
  void exchange_device(BrickStorage &bStorage_dev, void *stream = nullptr) {
    int ranks[27];
    ranks27(ranks);
    check_comm_rank(ranks);
    if (brk_exchange(h_, bStorage_dev.dat.get(), bStorage_dev.step, rccl_comm, ranks, stream) != BRK_OK)
      throw std::runtime_error(brk_last_error());
  }
#ifndef BRK_HAVE_MPI
  // no MPI at compile time: exchange() IS the device exchange (reference signature, include/brick-mpi.h:466)
  void exchange(BrickStorage &bStorage) { exchange_device(bStorage); }
#endif

#ifdef BRK_HAVE_MPI
  // ghost run i <- skin run i of the rank behind ghost[i].neighbor, tag i (reference include/brick-mpi.h:466-495);
  // with an RCCL communicator set, the storage is taken to live on the GPU and goes through libbrk.so instead
  void exchange(BrickStorage &bStorage) {
    if (rccl_comm) return exchange_device(bStorage);
    std::vector<MPI_Request> req(ghost.size() * 2);
    bElem *dat = bStorage.dat.get();
    const size_t bytes_per_brick = bStorage.step * sizeof(bElem);
    for (size_t q = 0; q < ghost.size(); ++q) {
      MPI_Irecv(dat + (size_t)ghost[q].pos * bStorage.step, (int)(ghost[q].len * bytes_per_brick), MPI_CHAR,
                rank_map[ghost[q].neighbor.set], (int)q, comm, &req[2 * q]);
      MPI_Isend(dat + (size_t)skin[q].pos * bStorage.step, (int)(skin[q].len * bytes_per_brick), MPI_CHAR,
                rank_map[skin[q].neighbor.set], (int)q, comm, &req[2 * q + 1]);
    }
    MPI_Waitall((int)req.size(), req.data(), MPI_STATUSES_IGNORE);
  }

  // one-sided form (reference include/brick-mpi.h:688-712, marked "don't use" there): between two fences every ghost run is
  // fetched with MPI_Get from the same-index skin run of the rank behind it; `win` exposes the peer's whole storage in bytes.
  // Host storages (or device pointers under a GPU-aware MPI); wait / call times go to the reference's accumulators.
  void exchange(BrickStorage bStorage, MPI_Win &win) {
    double st = MPI_Wtime(), ed;
    MPI_Win_fence(0, win);
    ed = MPI_Wtime();
    waittime += ed - st;
    st = ed;
    bElem *dat = bStorage.dat.get();
    const size_t bytes_per_brick = bStorage.step * sizeof(bElem);
    for (size_t q = 0; q < ghost.size(); ++q) {
      const int len = (int)(ghost[q].len * bytes_per_brick);
      MPI_Get(dat + (size_t)ghost[q].pos * bStorage.step, len, MPI_CHAR, rank_map[ghost[q].neighbor.set],
              (MPI_Aint)((size_t)skin[q].pos * bytes_per_brick), len, MPI_CHAR, win);
    }
    ed = MPI_Wtime();
    calltime += ed - st;
    st = ed;
    MPI_Win_fence(0, win);
    waittime += MPI_Wtime() - st;
  }
#endif

private:
  brk_decomp *h_ = nullptr;
  uint32_t cells_[3] = {0, 0, 0}, depth_ = 0;
  unsigned numfield_;
};

// min / avg / max / sigma of one number per rank, gathered on rank 0 (reference include/brick-mpi.h:758-793)
struct mpi_stats {
  double min, max, avg, sigma;
};
inline std::ostream &operator<<(std::ostream &os, const mpi_stats &s) {
  return os << "[" << s.min << ", " << s.avg << ", " << s.max << "] (σ: " << s.sigma << ")";
}
#ifdef BRK_HAVE_MPI
inline mpi_stats mpi_statistics(double value, MPI_Comm comm) {
  mpi_stats out{0, 0, 0, 0};
  if (comm == MPI_COMM_NULL) return out;
  int size = 1;
  MPI_Comm_size(comm, &size);
  double sum = 0, sumsq = 0, sq = value * value;
  MPI_Reduce(&value, &sum, 1, MPI_DOUBLE, MPI_SUM, 0, comm);
  MPI_Reduce(&value, &out.max, 1, MPI_DOUBLE, MPI_MAX, 0, comm);
  MPI_Reduce(&value, &out.min, 1, MPI_DOUBLE, MPI_MIN, 0, comm);
  MPI_Reduce(&sq, &sumsq, 1, MPI_DOUBLE, MPI_SUM, 0, comm);
  out.avg = sum / size;
  const double var = (sumsq - sum * sum / size) / (size > 1 ? size - 1 : 1);   // sample variance
  out.sigma = var > 0 ? __builtin_sqrt(var) : 0.0;
  return out;
}
#endif

// the three orders of the surface sections the reference ships (src/brick-mpi.cpp:25-136), from the library
inline std::vector<BitSet> brk_skin3d_list(int which) {
  uint64_t raw[BRK_MAX_SKIN];
  uint32_t n = 0;
  std::vector<BitSet> out;
  if (brk_skin3d_order(which, raw, &n) == BRK_OK)
    for (uint32_t q = 0; q < n; ++q) out.emplace_back(raw[q]);
  return out;
}
inline const std::vector<BitSet> &skin3d_good_list() {
  static const std::vector<BitSet> v = brk_skin3d_list(BRK_SKIN3D_GOOD);
  return v;
}
inline const std::vector<BitSet> &skin3d_normal_list() {
  static const std::vector<BitSet> v = brk_skin3d_list(BRK_SKIN3D_NORMAL);
  return v;
}
inline const std::vector<BitSet> &skin3d_bad_list() {
  static const std::vector<BitSet> v = brk_skin3d_list(BRK_SKIN3D_BAD);
  return v;
}
#define skin3d_good (skin3d_good_list())
#define skin3d_normal (skin3d_normal_list())
#define skin3d_bad (skin3d_bad_list())

// neighbour -> rank on a periodic Cartesian grid with MPI_Cart_rank's row-major rank order (reference populate(),
// include/brick-mpi.h:730-753, including its orientation: the "+d" neighbour sits at coordinate - 1)
template <unsigned dim, unsigned... BDims>
void populate_cart(BrickDecomp<dim, BDims...> &bDecomp, const int cart_dims[3], const int coo[3]) {
  int out[27];
  if (brk_populate_rank_map(cart_dims, coo, out) != BRK_OK) throw std::runtime_error(brk_last_error());
  uint64_t nb[27];
  brk_allneighbors3(nb);
  for (int q = 0; q < 27; ++q) bDecomp.rank_map[nb[q]] = out[q];
}
#ifdef BRK_HAVE_MPI
// An RCCL communicator over the ranks of an MPI communicator: rank 0's ncclUniqueId travels by MPI_Bcast, every rank joins
// (one rank per GPU; call after hipSetDevice).  The result goes into BrickDecomp::rccl_comm.
inline void *brkCommFromMPI(MPI_Comm comm) {
  int rank = 0, size = 1;
  MPI_Comm_rank(comm, &rank);
  MPI_Comm_size(comm, &size);
  // rank 0 ALWAYS takes part in the broadcast: a status byte travels with the id, and every rank throws after it when rank 0
  // could not make one (librccl missing) -- nobody is left waiting inside MPI_Bcast
  char msg[129] = {0};
  std::string why;
  if (rank == 0 && brk_comm_unique_id(msg + 1) != BRK_OK) {
    msg[0] = 1;
    why = brk_last_error();
  }
  MPI_Bcast(msg, 129, MPI_CHAR, 0, comm);
  if (msg[0]) throw std::runtime_error("brkCommFromMPI: rank 0 could not create an RCCL unique id" + (why.empty() ? std::string() : ": " + why));
  const char *id = msg + 1;
  void *c = nullptr;
  if (brk_comm_init_rank(&c, size, id, rank) != BRK_OK) throw std::runtime_error(brk_last_error());
  return c;
}

template <unsigned dim, unsigned... BDims>
void populate(MPI_Comm &comm, BrickDecomp<dim, BDims...> &bDecomp, BitSet neighbor, int d, int *coo) {
  if (neighbor.set != 0 || d != 1) throw std::runtime_error("populate: call with (comm, bDecomp, 0, 1, coo)");
  int cart[3], periods[3], mine[3];
  MPI_Cart_get(comm, 3, cart, periods, mine);
  bDecomp.comm = comm;
  populate_cart(bDecomp, cart, coo);
}
#endif
