// brk_host.cpp -- host-side layout builders behind the C ABI (include/brk.h).
//
// Brick numbering, adjacency, ghost/skin run tables, neighbour->rank maps and Z-Morton ownership are
// integer bookkeeping done once at setup; they stay on the host.  The rules restated here come from the
// reference (citations per function, paths relative to the reference tree); the code is written fresh.
#include "brk_internal.h"

#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

// ---------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

int brk_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *brk_last_error(void) { return g_err; }
static thread_local const char *g_last_kernel = "";
extern "C" void brki_note_kernel(const char *name) { g_last_kernel = name; }
extern "C" const char *brk_last_kernel(void) { return g_last_kernel; }
extern "C" int brk_abi_version(void) { return 1013; } // 1001: brk_stencil_apply_taps; 1002: BRK_MPI7_X2, BRK_7PT_X2; 1003: exchange; 1004: brk_adjacency_register; 1005: brk_stencil_apply_*_ex; 1006: brk_comm_library, brk_set_variant left the ABI; 1007: brk_storage_alloc / _free, brk_placement_probe; 1008: brk_finalize, brk_last_kernel; 1009: brk_storage_alloc_ex; 1010: brk_stencil_apply_expr; 1011: brk_expr_compile / _apply / _release / _source; 1012: brk_expr_kernel_from_disk; 1013: brk_expr_apply_box

// ---------------------------------------------------------------------------------------------- stencil facts
extern "C" int brk_stencil_radius(int id) {
  static const int r[BRK_NSTENCIL] = {1, 1, 1, 2, 4, 2, 1, 2, 2};
  return (id >= 0 && id < BRK_NSTENCIL) ? r[id] : BRK_EINVAL;
}
extern "C" int brk_stencil_ncoeff(int id) {
  static const int n[BRK_NSTENCIL] = {7, 27, 2, 3, 5, 10, 7, 2, 7};
  return (id >= 0 && id < BRK_NSTENCIL) ? n[id] : BRK_EINVAL;
}
// ST_ITER = ghost depth (8 cells) / radius: stencils/fake.h:39,72,88,344.  The single-GPU coefficient
// stencils have no exchange cycle; they report 1.
extern "C" int brk_stencil_iters(int id) {
  static const int n[BRK_NSTENCIL] = {1, 1, 8, 4, 2, 4, 1, 4, 1}; // the fused double sweep: 8 / 2 passes per exchange
  return (id >= 0 && id < BRK_NSTENCIL) ? n[id] : BRK_EINVAL;
}

// ---------------------------------------------------------------------------------------------- direction sets
// A direction set is the reference's BitSet (include/bitset.h:20-30): axis p in 1..3, "+p" is bit p, "-p" is
// bit 31+p.  Negating all directions swaps the two halves (bitset.h:119-126).  Helpers: brk_internal.h.
using namespace brk_dir;
namespace {
uint64_t make_set(std::initializer_list<int> l) {
  uint64_t s = 0;
  for (int p : l) s ^= dir_bit(p);
  return s;
}

// src/brick-mpi.cpp:25-52 (skin3d_good): the order in which the 26 surface sections are laid out.
const std::vector<uint64_t> &surface_order() {
  static const std::vector<uint64_t> v = {
      make_set({1}),          make_set({1, -3}),     make_set({1, 2, -3}),  make_set({1, 2}),      make_set({1, 2, 3}),
      make_set({2, 3}),       make_set({2}),         make_set({2, -3}),     make_set({-1, 2, -3}), make_set({-1, 2}),
      make_set({-1, 2, 3}),   make_set({-1, 3}),     make_set({-1}),        make_set({-3}),        make_set({-1, -3}),
      make_set({-1, -2, -3}), make_set({-1, -2}),    make_set({-1, -2, 3}), make_set({-2, 3}),     make_set({-2}),
      make_set({-2, -3}),     make_set({1, -2, -3}), make_set({1, -2}),     make_set({1, -2, 3}),  make_set({1, 3}),
      make_set({3})};
  return v;
}

// src/brick-mpi.cpp:54-82 (skin3d_normal): the 27 sections in lexicographic order of (axis 3, axis 2, axis 1) in {-, none, +},
// the empty set (no surface section: a run separator) in the middle
std::vector<uint64_t> surface_order_normal() {
  std::vector<uint64_t> v;
  const int pick[3] = {-1, 0, +1};
  for (int c : pick)
    for (int b : pick)
      for (int a : pick) {
        uint64_t s = 0;
        if (a) s |= dir_bit(a * 1);
        if (b) s |= dir_bit(b * 2);
        if (c) s |= dir_bit(c * 3);
        v.push_back(s);
      }
  return v;
}
// src/brick-mpi.cpp:84-136 (skin3d_bad): the same 26 sections with an empty set between any two, so that no two sections
// are ever contiguous for a neighbour (one run per section: 98 runs)
std::vector<uint64_t> surface_order_bad() {
  std::vector<uint64_t> v;
  for (uint64_t s : surface_order_normal())
    if (s) {
      if (!v.empty()) v.push_back(0);
      v.push_back(s);
    }
  return v;
}
} // namespace

extern "C" int brk_skin3d_order(int which, uint64_t *out, uint32_t *n) {
  if (!out || !n) return brk_fail(BRK_EINVAL, "brk_skin3d_order: null argument");
  std::vector<uint64_t> v;
  if (which == BRK_SKIN3D_GOOD) v = surface_order();
  else if (which == BRK_SKIN3D_NORMAL) v = surface_order_normal();
  else if (which == BRK_SKIN3D_BAD) v = surface_order_bad();
  else return brk_fail(BRK_EINVAL, "brk_skin3d_order: unknown order %d", which);
  std::copy(v.begin(), v.end(), out);
  *n = (uint32_t)v.size();
  return BRK_OK;
}

// src/brick-mpi.cpp:9-23 -- for axis 1,2,3 in turn choose '+', nothing, '-' (axis 1 varies slowest).
void brk_dir::neighbors3(std::array<uint64_t, 27> &out) {
  int q = 0;
  const int pick[3] = {+1, 0, -1};
  for (int a : pick)
    for (int b : pick)
      for (int c : pick) {
        uint64_t s = 0;
        if (a) s |= dir_bit(a * 1);
        if (b) s |= dir_bit(b * 2);
        if (c) s |= dir_bit(c * 3);
        out[q++] = s;
      }
}

extern "C" void brk_allneighbors3(uint64_t out[27]) {
  std::array<uint64_t, 27> a;
  neighbors3(a);
  std::copy(a.begin(), a.end(), out);
}

// ---------------------------------------------------------------------------------------------- init_grid
// include/bricksetup.h:30-90.  Lexicographic ids; a neighbour exists iff its FLAT position is inside the grid
// (bricksetup.h:44) -- bricks on the outer ring therefore see wrapped neighbours, as in the reference.
extern "C" int brk_init_grid(const long dims[3], uint32_t *grid, uint32_t *adj) {
  if (!dims || !grid || !adj || dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0)
    return brk_fail(BRK_EINVAL, "brk_init_grid: bad arguments");
  const long n = dims[0] * dims[1] * dims[2];
  if (n > 0xFFFFFFFFl)
    return brk_fail(BRK_EINVAL, "brk_init_grid: %ld bricks do not fit 32-bit ids", n);
  const long sj = dims[0], sk = dims[0] * dims[1];
  for (long p = 0; p < n; ++p)
    grid[p] = (uint32_t)p;
  for (long p = 0; p < n; ++p) {
    uint32_t *row = adj + (size_t)p * 27;
    int d = 0;
    for (long ok : {-sk, 0l, sk})
      for (long oj : {-sj, 0l, sj})
        for (long oi : {-1l, 0l, 1l}) {
          long q = p + ok + oj + oi;
          row[d++] = (q >= 0 && q < n) ? (uint32_t)q : 0u;
        }
  }
  return BRK_OK;
}

// ---------------------------------------------------------------------------------------------- BrickDecomp
// include/brick-mpi.h:178-460.  Brick ids are handed out region by region so that everything one neighbour
// needs (its ghost on our side, our skin on its side) is a few contiguous id ranges:
//   [1, sep0)      interior bricks that touch no surface, lexicographic           (brick-mpi.h:382)
//   [sep0, sep1)   26 surface sections in skin3d_good order                        (:389-398)
//   [sep1, sep2)   ghost: per neighbour n, the sections of n's surface that contain -n, in the same order (:404-452)
// id 0 is a dummy brick that every out-of-domain adjacency entry points to (:275-276,:353).
// Brick size (4096 B) equals the page size, so the page-padding logic (:342-368) contributes nothing.

namespace {
// Number the bricks of section `region` of the subdomain displaced by `owner` (0 = ours), k slowest, i fastest.
// A '+' on an axis selects the last `depth` bricks of that subdomain, '-' the first, neither the middle.
void number_section(brk_decomp &D, uint64_t owner, uint64_t region, uint32_t &next) {
  long lo[3], hi[3];
  for (int a = 0; a < 3; ++a) {
    long base = D.depth[a]; // our subdomain starts `depth` bricks into the padded grid
    if (has(owner, a + 1)) base += D.dims[a];
    if (has(owner, -(a + 1))) base -= D.dims[a];
    if (has(region, a + 1)) {
      lo[a] = base + D.dims[a] - D.depth[a];
      hi[a] = base + D.dims[a];
    } else if (has(region, -(a + 1))) {
      lo[a] = base;
      hi[a] = base + D.depth[a];
    } else {
      lo[a] = base + D.depth[a];
      hi[a] = base + D.dims[a] - D.depth[a];
    }
  }
  for (long k = lo[2]; k < hi[2]; ++k)
    for (long j = lo[1]; j < hi[1]; ++j)
      for (long i = lo[0]; i < hi[0]; ++i)
        D.grid[(size_t)k * D.stride[2] + (size_t)j * D.stride[1] + (size_t)i] = next++;
}
} // namespace

extern "C" int brk_decomp_build(const uint32_t dims_cells[3], uint32_t ghost_cells, brk_decomp **out) {
  return brk_decomp_build_order(dims_cells, ghost_cells, surface_order().data(), (uint32_t)surface_order().size(), out);
}

// initialize(skinlist) of the reference for ANY order of the surface sections (include/brick-mpi.h:322-460): every one of the 26
// sections must appear exactly once; empty sets may stand anywhere (they number nothing and end a neighbour's run, :393,:424-446)
extern "C" int brk_decomp_build_order(const uint32_t dims_cells[3], uint32_t ghost_cells, const uint64_t *order, uint32_t norder,
                                      brk_decomp **out) {
  if (!dims_cells || !out || !order)
    return brk_fail(BRK_EINVAL, "brk_decomp_build: null argument");
  {
    std::array<uint64_t, 27> all;
    neighbors3(all);
    if (norder > BRK_MAX_SKIN) return brk_fail(BRK_EINVAL, "brk_decomp_build: %u surface sections exceed %d", norder, BRK_MAX_SKIN);
    for (uint64_t want : all) {
      if (!want) continue;
      if (std::count(order, order + norder, want) != 1)
        return brk_fail(BRK_EINVAL, "brk_decomp_build: every one of the 26 surface sections must be listed exactly once");
    }
    for (uint32_t q = 0; q < norder; ++q)
      if (order[q] && std::find(all.begin(), all.end(), order[q]) == all.end())
        return brk_fail(BRK_EINVAL, "brk_decomp_build: entry %u of the order is not a surface section", q);
  }
  if (ghost_cells == 0 || ghost_cells % 8)
    return brk_fail(BRK_EINVAL, "brk_decomp_build: ghost depth %u is not a positive multiple of the brick edge 8", ghost_cells);
  for (int a = 0; a < 3; ++a)
    if (dims_cells[a] % 8 || dims_cells[a] < 2 * ghost_cells)
      return brk_fail(BRK_EINVAL, "brk_decomp_build: extent %u must be a multiple of 8 and >= 2*ghost", dims_cells[a]);
  brk_decomp *D = new (std::nothrow) brk_decomp();
  if (!D)
    return brk_fail(BRK_ENOMEM, "brk_decomp_build: out of memory");
  uint64_t total = 1;
  for (int a = 0; a < 3; ++a) {
    D->dims[a] = dims_cells[a] / 8;
    D->depth[a] = ghost_cells / 8;
    D->tdims[a] = D->dims[a] + 2 * D->depth[a];
    D->stride[a] = (uint32_t)total;
    total *= D->tdims[a];
  }
  if (total >= 0xFFFFFFFFull) {
    delete D;
    return brk_fail(BRK_EINVAL, "brk_decomp_build: %llu bricks do not fit 32-bit ids", (unsigned long long)total);
  }
  D->grid.assign((size_t)total, 0u);
  D->skinlist.assign(order, order + norder);

  uint32_t next = 1;
  number_section(*D, 0, 0, next);
  D->sep[0] = next;
  std::vector<uint32_t> sec_start;
  for (uint64_t sec : D->skinlist) {
    sec_start.push_back(next);
    if (sec) number_section(*D, 0, sec, next); // an empty set numbers nothing (:393)
    D->skin_size.push_back((int64_t)next - (int64_t)sec_start.back());
  }
  sec_start.push_back(next);
  D->sep[1] = next;

  std::array<uint64_t, 27> nbs;
  neighbors3(nbs);
  const int nsec = (int)D->skinlist.size();
  for (uint64_t n : nbs) {
    if (!n) continue;
    const uint64_t toward_us = negate(n);
    int run_first = -1;
    uint32_t run_pos = 0;
    auto close_run = [&](int end) {
      brk_region g{n, (uint32_t)run_first, (uint32_t)end, run_pos, next - run_pos, 0, 0};
      brk_region s{toward_us, (uint32_t)run_first, (uint32_t)end, sec_start[run_first],
                   sec_start[end] - sec_start[run_first], 0, 0};
      D->ghost.push_back(g);
      D->skin.push_back(s);
      run_first = -1;
    };
    for (int l = 0; l < nsec; ++l) {
      if (D->skinlist[l] && contains(D->skinlist[l], toward_us)) {
        if (run_first < 0) {
          run_first = l;
          run_pos = next;
        }
        number_section(*D, n, D->skinlist[l], next);
      } else if (run_first >= 0) {
        close_run(l);
      }
    }
    if (run_first >= 0) close_run(nsec);
  }
  D->sep[2] = next;
  D->nbricks = next;

  // adjacency from the logical grid (brick-mpi.h:266-291,455-459); out of the padded grid -> 0
  D->adj.assign((size_t)next * 27, 0u);
  for (uint32_t k = 0; k < D->tdims[2]; ++k)
    for (uint32_t j = 0; j < D->tdims[1]; ++j)
      for (uint32_t i = 0; i < D->tdims[0]; ++i) {
        uint32_t *row = &D->adj[(size_t)D->grid[(size_t)k * D->stride[2] + (size_t)j * D->stride[1] + i] * 27];
        int d = 0;
        for (int dk = -1; dk <= 1; ++dk)
          for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di, ++d) {
              long kk = (long)k + dk, jj = (long)j + dj, ii = (long)i + di;
              bool in = kk >= 0 && kk < D->tdims[2] && jj >= 0 && jj < D->tdims[1] && ii >= 0 && ii < D->tdims[0];
              row[d] = in ? D->grid[(size_t)kk * D->stride[2] + (size_t)jj * D->stride[1] + (size_t)ii] : 0u;
            }
      }
  *out = D;
  return BRK_OK;
}

extern "C" void brk_decomp_free(brk_decomp *d) {
  if (!d) return;
  for (auto &e : d->xcache) { // what brk_exchange() built for this decomposition
    brk_exchange_view_free(e.view);
    brk_xplan_free(e.plan);
  }
  delete d;
}
extern "C" uint32_t brk_decomp_nbricks(const brk_decomp *d) { return d->nbricks; }
extern "C" uint32_t brk_decomp_grid_size(const brk_decomp *d) { return (uint32_t)d->grid.size(); }
extern "C" void brk_decomp_tdims(const brk_decomp *d, uint32_t out[3]) { std::copy(d->tdims, d->tdims + 3, out); }
extern "C" void brk_decomp_sep_pos(const brk_decomp *d, uint32_t out[3]) { std::copy(d->sep, d->sep + 3, out); }
extern "C" const uint32_t *brk_decomp_grid(const brk_decomp *d) { return d->grid.data(); }
extern "C" const uint32_t *brk_decomp_adj(const brk_decomp *d) { return d->adj.data(); }
extern "C" uint32_t brk_decomp_nruns(const brk_decomp *d) { return (uint32_t)d->ghost.size(); }
extern "C" const brk_region *brk_decomp_ghost(const brk_decomp *d) { return d->ghost.data(); }
extern "C" const brk_region *brk_decomp_skin(const brk_decomp *d) { return d->skin.data(); }
extern "C" uint32_t brk_decomp_nskin(const brk_decomp *d) { return (uint32_t)d->skinlist.size(); }
extern "C" const uint64_t *brk_decomp_skinlist(const brk_decomp *d) { return d->skinlist.data(); }
extern "C" const int64_t *brk_decomp_skin_size(const brk_decomp *d) { return d->skin_size.data(); }

// ---------------------------------------------------------------------------------------------- rank maps
// include/brick-mpi.h:730-753.  Axis d (1 = contiguous) lives in cart coordinate coo[3-d]; direction "+d"
// is the rank at coo-1 and "-d" the rank at coo+1 (the reference's orientation).  Ranks are row-major over
// (coo[0],coo[1],coo[2]) with periodic wrap, as MPI_Cart_rank.
extern "C" int brk_populate_rank_map(const int cart[3], const int coo[3], int out[27]) {
  if (!cart || !coo || !out || cart[0] <= 0 || cart[1] <= 0 || cart[2] <= 0)
    return brk_fail(BRK_EINVAL, "brk_populate_rank_map: bad arguments");
  std::array<uint64_t, 27> nbs;
  neighbors3(nbs);
  for (int q = 0; q < 27; ++q) {
    int c[3] = {coo[0], coo[1], coo[2]};
    for (int d = 1; d <= 3; ++d) {
      if (has(nbs[q], d)) c[3 - d] -= 1;
      if (has(nbs[q], -d)) c[3 - d] += 1;
    }
    for (int a = 0; a < 3; ++a)
      c[a] = ((c[a] % cart[a]) + cart[a]) % cart[a];
    out[q] = (c[0] * cart[1] + c[1]) * cart[2] + c[2];
  }
  return BRK_OK;
}

// MPI_Dims_create(size, 3, {0,0,0}): factors as evenly as possible, non-increasing (weak/args.cpp:105).
extern "C" int brk_dims_create(int size, int out[3]) {
  if (size <= 0 || !out)
    return brk_fail(BRK_EINVAL, "brk_dims_create: bad arguments");
  int best[3] = {size, 1, 1};
  for (int a = 1; a <= size; ++a) {
    if (size % a) continue;
    for (int b = 1; b <= size / a; ++b) {
      if ((size / a) % b) continue;
      int c = size / a / b;
      int t[3] = {a, b, c};
      std::sort(t, t + 3, [](int x, int y) { return x > y; });
      if (t[0] - t[2] < best[0] - best[2] || (t[0] - t[2] == best[0] - best[2] && t[0] < best[0]))
        std::copy(t, t + 3, best);
    }
  }
  std::copy(best, best + 3, out);
  return BRK_OK;
}

// ---------------------------------------------------------------------------------------------- Z-Morton / strong
// include/zmort.h:66-99: coordinate d of a 3-D Z-Morton id is the bits d, d+3, d+6, ...
extern "C" uint64_t brk_zmort_get(uint64_t id, uint64_t d) {
  uint64_t v = 0;
  id >>= d;
  for (int bit = 0; id; ++bit, id >>= 3)
    v |= (id & 1ull) << bit;
  return v;
}

extern "C" uint64_t brk_zmort_set(uint64_t id, uint64_t d, uint64_t p) {
  uint64_t mask = 0;
  for (uint64_t bit = d; bit < 64; bit += 3)
    mask |= 1ull << bit;
  uint64_t spread = 0;
  for (uint64_t bit = d; p && bit < 64; bit += 3, p >>= 1)
    spread |= (p & 1ull) << bit;
  return (id & ~mask) | spread;
}

// strong/args.cpp:104-113: contiguous Z-Morton ranges, the first allsubs%size ranks own one extra subdomain.
extern "C" void brk_strong_partition(uint64_t allsubs, int size, int rank, uint64_t out[2]) {
  uint64_t base = allsubs / (uint64_t)size, extra = allsubs % (uint64_t)size;
  uint64_t r = (uint64_t)rank;
  out[0] = r * base + std::min(r, extra);
  out[1] = out[0] + base + (r < extra ? 1 : 0);
}

// strong/args.cpp:36-56: neighbour subdomain in direction set n, periodic in block coordinates
// ("+d" -> coordinate+1 here, unlike populate()), then its owner rank and slot on that rank.
extern "C" uint64_t brk_strong_neighbor(uint64_t n, uint64_t zid, uint64_t subdim, uint64_t allsubs, int size, int *owner,
                                        int *slot) {
  for (int d = 0; d < 3; ++d) {
    long off = has(n, d + 1) ? 1 : (has(n, -(d + 1)) ? -1 : 0);
    if (!off) continue;
    long c = (long)brk_zmort_get(zid, (uint64_t)d);
    c = (c + (long)subdim + off) % (long)subdim;
    zid = brk_zmort_set(zid, (uint64_t)d, (uint64_t)c);
  }
  uint64_t base = allsubs / (uint64_t)size, extra = allsubs % (uint64_t)size;
  uint64_t split = extra * (base + 1);
  uint64_t o, s;
  if (zid < split) {
    o = zid / (base + 1);
    s = zid % (base + 1);
  } else {
    o = extra + (zid - split) / base;
    s = (zid - split) % base;
  }
  if (owner) *owner = (int)o;
  if (slot) *slot = (int)s;
  return zid;
}
