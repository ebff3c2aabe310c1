// The DEVICE-facing host code of libbrk.so -- exchange views (create / begin / loopback / transport / end / free), brk_exchange with
// its per-decomposition cache, the box-list cache (replacement + retirement), registered adjacency tables, every launch path --
// driven on a CPU under AddressSanitizer + UBSan against tests/hipmock (malloc-backed HIP stand-in whose emulated kernels check
// every address a launch would touch).  Built by tests/hipmock/build.sh, run by tests/test_sanitized_host_paths.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "brk.h"

extern "C" unsigned long hipmock_live_allocations(void);
extern "C" unsigned long hipmock_launches(void);

#define REQ(x)                                                                                 \
  do {                                                                                         \
    if (!(x)) {                                                                                \
      fprintf(stderr, "probe: line %d: %s failed (%s)\n", __LINE__, #x, brk_last_error());     \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

static double *dev_doubles(size_t n, double fill) {
  double *p = nullptr;
  if (hipMalloc((void **)&p, n * 8) != hipSuccess) abort();
  for (size_t q = 0; q < n; ++q) p[q] = fill; // "device" memory is host memory under the mock
  return p;
}
static uint32_t *dev_u32(const uint32_t *src, size_t n) {
  uint32_t *p = nullptr;
  if (hipMalloc((void **)&p, n * 4) != hipSuccess) abort();
  std::memcpy(p, src, n * 4);
  return p;
}

// ---- 1. exchange between `world` virtual ranks of a periodic grid: every mode, through views and the loopback transport
static int exchange_virtual_ranks(const uint32_t shape[3], int world, size_t step) {
  brk_decomp *d = nullptr;
  REQ(brk_decomp_build(shape, 8, &d) == BRK_OK);
  const uint32_t nb = brk_decomp_nbricks(d), nr = brk_decomp_nruns(d);
  const brk_region *g = brk_decomp_ghost(d), *k = brk_decomp_skin(d);
  uint64_t nbset[27];
  brk_allneighbors3(nbset);
  int cart[3];
  REQ(brk_dims_create(world, cart) == BRK_OK);
  for (int mode : {BRK_XCHG_PACKED, BRK_XCHG_DIRECT, BRK_XCHG_SHIFT}) {
    std::vector<double *> st(world);
    std::vector<brk_xplan *> plans(world);
    std::vector<brk_exchange_view *> views(world);
    std::vector<std::vector<int>> rmap(world, std::vector<int>(27));
    for (int r = 0; r < world; ++r) {
      const int coo[3] = {r / (cart[1] * cart[2]), (r / cart[2]) % cart[1], r % cart[2]};
      REQ(brk_populate_rank_map(cart, coo, rmap[r].data()) == BRK_OK);
      REQ(rmap[r][13] == r);
      st[r] = dev_doubles((size_t)nb * step, -1.0);
      for (uint32_t b = 0; b < nb; ++b)
        for (size_t e = 0; e < step; ++e) st[r][(size_t)b * step + e] = r * 1e6 + b + e * 1e-4;
      REQ(brk_xplan_build_decomp(d, rmap[r].data(), r, mode, 0, step, &plans[r]) == BRK_OK);
      REQ(brk_exchange_view_create(plans[r], st[r], &views[r]) == BRK_OK);
    }
    const uint32_t nstages = brk_xplan_nstages(plans[0]);
    for (uint32_t s = 0; s < nstages; ++s) {
      for (int r = 0; r < world; ++r) REQ(brk_exchange_begin(views[r], s, nullptr) == BRK_OK);
      REQ(brk_exchange_loopback(views.data(), (uint32_t)world, s, nullptr) == BRK_OK);
      for (int r = 0; r < world; ++r) REQ(brk_exchange_end(views[r], s, nullptr) == BRK_OK);
    }
    // ghost run q of rank r == skin run q of the rank behind ghost[q].neighbor (reference include/brick-mpi.h:466-495)
    for (int r = 0; r < world; ++r)
      for (uint32_t q = 0; q < nr; ++q) {
        int gi = -1;
        for (int t = 0; t < 27; ++t)
          if (nbset[t] == g[q].neighbor) gi = t;
        REQ(gi >= 0);
        const int src = rmap[r][gi];
        for (uint32_t t = 0; t < g[q].len; ++t)
          for (size_t e = 0; e < step; e += 37) {
            const double want = src * 1e6 + (k[q].pos + t) + e * 1e-4;
            REQ(st[r][(size_t)(g[q].pos + t) * step + e] == want);
          }
      }
    for (int r = 0; r < world; ++r) {
      brk_exchange_view_free(views[r]);
      brk_xplan_free(plans[r]);
      REQ(hipFree(st[r]) == hipSuccess);
    }
  }
  brk_decomp_free(d);
  return 0;
}

// ---- 2. brk_exchange(): the reference-shaped call with its cache of (storage, step, rank map, communicator) -> plan + view
static int exchange_cached(void *comm) {
  const uint32_t shape[3] = {32, 24, 16};
  brk_decomp *d = nullptr;
  REQ(brk_decomp_build(shape, 8, &d) == BRK_OK);
  const uint32_t nb = brk_decomp_nbricks(d);
  int ranks[27];
  for (int &r : ranks) r = 0;
  double *a = dev_doubles((size_t)nb * 512, 1.0), *b = dev_doubles((size_t)nb * 1024, 2.0);
  for (int rep = 0; rep < 3; ++rep) {
    REQ(brk_exchange(d, a, 512, comm, ranks, nullptr) == BRK_OK);
    REQ(brk_exchange(d, b, 1024, comm, ranks, nullptr) == BRK_OK);
  }
  brk_decomp_free(d); // frees the two cached views and plans
  REQ(hipFree(a) == hipSuccess && hipFree(b) == hipSuccess);
  return 0;
}

// ---- 3. launch paths over an init_grid numbering: range / list / k-paired list / box (list cache: > 16 geometries, both rounds),
//         registered and unregistered adjacency, taps, array <-> brick, compare
static int launches() {
  const long dims[3] = {7, 6, 9};
  const uint32_t tot[3] = {7, 6, 9};
  const size_t nbr = 7 * 6 * 9;
  std::vector<uint32_t> grid(nbr), adj(nbr * 27);
  REQ(brk_init_grid(dims, grid.data(), adj.data()) == BRK_OK);
  uint32_t *adj_dev = dev_u32(adj.data(), adj.size());
  double *in = dev_doubles(nbr * 512, 0.0), *out = dev_doubles(nbr * 512, 0.0);
  for (size_t q = 0; q < nbr * 512; ++q) in[q] = std::sin(0.001 * (double)q);
  double coeff[27];
  for (int q = 0; q < 27; ++q) coeff[q] = 0.01 * (q + 1);
  const int stencils[] = {BRK_7PT_COEFF, BRK_27PT_COEFF, BRK_MPI7, BRK_MPI13, BRK_MPI25, BRK_MPI125, BRK_COND7, BRK_MPI7_X2, BRK_7PT_X2};
  for (int pass = 0; pass < 2; ++pass) { // pass 1: over a registered table
    if (pass == 1) {
      REQ(brk_adjacency_register(adj_dev, (uint32_t)nbr) == BRK_OK);
      REQ(brk_adjacency_register(adj_dev, (uint32_t)nbr) != BRK_OK); // twice: refused
    }
    for (int id : stencils) {
      const double *cf = (id == BRK_7PT_COEFF || id == BRK_27PT_COEFF || id == BRK_COND7 || id == BRK_7PT_X2) ? coeff : nullptr;
      // interior box of the grid, every lo/cnt combination that keeps the halo inside: 2 * 2 * 3 * ... > 16 geometries
      int geoms = 0;
      for (uint32_t ci = 1; ci <= 5; ci += 2)
        for (uint32_t cj = 1; cj <= 4; cj += 3)
          for (uint32_t ck = 2; ck <= 7; ck += 2) {
            const uint32_t lo[3] = {1, 1, 1}, cnt[3] = {ci, cj, ck};
            for (uint32_t flags : {0u, (uint32_t)BRK_SWEEP_REVERSE})
              REQ(brk_stencil_apply_box_ex(id, adj_dev, in, out, 512, 0, tot, lo, cnt, cf, flags, nullptr) == BRK_OK);
            ++geoms;
          }
      REQ(geoms > 16);
      // id range and lists: ragged, padded with skips, k-paired declaration
      REQ(brk_stencil_apply_range(id, adj_dev, in, out, 512, 7 * 6 + 7 + 1, 7 * 6 + 7 + 6, cf, nullptr) == BRK_OK);
      std::vector<uint32_t> ids;
      for (uint32_t k = 1; k < 8; ++k)
        for (uint32_t j = 1; j < 5; ++j) {
          for (uint32_t i = 1; i < 6; ++i) ids.push_back((k * 6 + j) * 7 + i);
          ids.push_back(BRK_LIST_SKIP);
        }
      uint32_t *ids_dev = dev_u32(ids.data(), ids.size());
      for (uint32_t n : {(uint32_t)ids.size(), 1u, 5u, 13u, 0u}) {
        REQ(brk_stencil_apply_list(id, adj_dev, in, out, 512, ids_dev, n, cf, nullptr) == BRK_OK);
        REQ(brk_stencil_apply_list_ex(id, adj_dev, in, out, 512, ids_dev, n, cf, BRK_SWEEP_KPAIRED | BRK_SWEEP_REVERSE, nullptr) == BRK_OK);
      }
      REQ(hipFree(ids_dev) == hipSuccess);
    }
    if (pass == 1) {
      REQ(brk_adjacency_release(adj_dev) == BRK_OK);
      REQ(brk_adjacency_release(adj_dev) != BRK_OK);
    }
  }
  // the 125-point kernel at four waves per SIMD: box entry point, registered table, rows of 8 bricks, every brick regular -- and the
  // same box over a table with ONE irregular brick (the reduction over the mask must send it back to the general kernel)
  {
    const long d2[3] = {10, 4, 5};
    const uint32_t t2[3] = {10, 4, 5}, lo2[3] = {1, 1, 1}, c2[3] = {8, 2, 3};
    const size_t nb2 = 10 * 4 * 5;
    std::vector<uint32_t> g2(nb2), a2(nb2 * 27);
    REQ(brk_init_grid(d2, g2.data(), a2.data()) == BRK_OK);
    for (int twist = 0; twist < 2; ++twist) {
      if (twist) a2[(size_t)((2 * 4 + 1) * 10 + 4) * 27 + 26] = 0; // one corner neighbour of an interior brick points at the dummy brick
      uint32_t *adj2 = dev_u32(a2.data(), a2.size());
      double *i2 = dev_doubles(nb2 * 512, 0.0), *o2 = dev_doubles(nb2 * 512, 0.0);
      for (size_t q = 0; q < nb2 * 512; ++q) i2[q] = std::cos(0.002 * (double)q);
      REQ(brk_adjacency_register(adj2, (uint32_t)nb2) == BRK_OK);
      for (uint32_t flags : {0u, (uint32_t)BRK_SWEEP_REVERSE, 0u})
        REQ(brk_stencil_apply_box_ex(BRK_MPI125, adj2, i2, o2, 512, 0, t2, lo2, c2, nullptr, flags, nullptr) == BRK_OK);
      const std::string kn = brk_last_kernel();
      REQ(twist ? kn.find("k_cube_dense") == 0 : kn.find("k_cube125_q") == 0);
      REQ(brk_adjacency_release(adj2) == BRK_OK);
      REQ(hipFree(adj2) == hipSuccess && hipFree(i2) == hipSuccess && hipFree(o2) == hipSuccess);
    }
  }
  // stencil given as taps
  {
    const int32_t off[] = {0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 2, 0, 0, 0, -2, 1, 1, 1};
    const double w[] = {0.5, 0.1, 0.1, 0.05, 0.05, 0.2};
    REQ(brk_stencil_apply_taps(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 3, 6, off, w, nullptr) == BRK_OK);
    const int32_t far[] = {9, 0, 0};
    REQ(brk_stencil_apply_taps(adj_dev, in, out, 512, nullptr, 1, 1, 1, far, w, nullptr) != BRK_OK);
    // star-shaped and cube-shaped tables reach radius 4 (the latter with the table in device memory, kept per content)
    const int32_t star4[] = {0, 0, 0, 4, 0, 0, -3, 0, 0, 0, 4, 0, 0, -4, 0, 0, 0, 3};
    REQ(brk_stencil_apply_taps(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 3, 6, star4, w, nullptr) == BRK_OK);
    const int32_t cube3[] = {3, 1, 0, -2, 3, -3};
    REQ(brk_stencil_apply_taps(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 1, 2, cube3, w, nullptr) == BRK_OK);
    REQ(std::string(brk_last_kernel()) == "k_taps_far<3,2>");
    REQ(brk_stencil_apply_taps(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 2, 2, cube3, w, nullptr) == BRK_OK); // the cached table
    const int32_t cube4[] = {4, -4, 1};
    REQ(brk_stencil_apply_taps(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 1, 1, cube4, w, nullptr) == BRK_OK);
    REQ(std::string(brk_last_kernel()) == "k_taps_far<4,1>");
    const int32_t cube5[] = {5, 1, 0};
    REQ(brk_stencil_apply_taps(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 1, 1, cube5, w, nullptr) != BRK_OK);
  }
  // stencil given as a register program: |max(in, 0) * in(i+2, j-1, k+1) - 0.5|, then the argument checks
  {
    brk_expr_op ops[7] = {};
    ops[0].code = BRK_EX_TAP, ops[0].dst = 0;
    ops[1].code = BRK_EX_CONST, ops[1].dst = 1, ops[1].imm = 0.0;
    ops[2].code = BRK_EX_MAX, ops[2].dst = 0, ops[2].a = 0, ops[2].b = 1;
    ops[3].code = BRK_EX_TAP, ops[3].dst = 1, ops[3].offset[0] = 2, ops[3].offset[1] = -1, ops[3].offset[2] = 1;
    ops[4].code = BRK_EX_MUL, ops[4].dst = 0, ops[4].a = 0, ops[4].b = 1;
    ops[5].code = BRK_EX_CONST, ops[5].dst = 1, ops[5].imm = 0.5;
    ops[6].code = BRK_EX_SUB, ops[6].dst = 2, ops[6].a = 0, ops[6].b = 1;
    REQ(brk_stencil_apply_expr(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 3, ops, 7, 2, nullptr) == BRK_OK);
    REQ(std::string(brk_last_kernel()) == "k_expr<2,4>");
    // the same program as generated source (compiling it needs a device: tests/test_gpu_parity.py)
    size_t need = 0;
    REQ(brk_expr_source(ops, 7, 2, nullptr, 0, &need) == BRK_OK && need > 500);
    std::vector<char> text(need);
    REQ(brk_expr_source(ops, 7, 2, text.data(), text.size(), nullptr) == BRK_OK);
    REQ(std::string(text.data()).find("c[88 + col[2]]") != std::string::npos && std::strlen(text.data()) + 1 == need); // radius 2, off-axis: the cube kernel's cell update (plane pitch 96)
    REQ(brk_expr_source(ops, 7, 2, text.data(), 40, nullptr) == BRK_OK && std::strlen(text.data()) == 39); // truncated, terminated
    REQ(brk_expr_release(nullptr) == BRK_OK);
    REQ(brk_stencil_apply_expr(adj_dev, in, out, 512, nullptr, 0, 3, ops, 7, 3, nullptr) != BRK_OK); // register 3 is never written
    ops[4].b = 9;
    REQ(brk_stencil_apply_expr(adj_dev, in, out, 512, nullptr, 0, 3, ops, 7, 2, nullptr) != BRK_OK); // read before written
    ops[4].b = 1, ops[3].offset[2] = 3;
    REQ(brk_stencil_apply_expr(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 2, ops, 7, 2, nullptr) == BRK_OK && std::string(brk_last_kernel()) == "k_expr<3,2>");
    ops[3].offset[0] = -4;
    REQ(brk_stencil_apply_expr(adj_dev, in, out, 512, nullptr, 7 * 6 * 2 + 7 * 2 + 2, 2, ops, 7, 2, nullptr) == BRK_OK && std::string(brk_last_kernel()) == "k_expr<4,1>");
    ops[3].offset[1] = 5;
    REQ(brk_stencil_apply_expr(adj_dev, in, out, 512, nullptr, 0, 3, ops, 7, 2, nullptr) != BRK_OK); // beyond radius 4
  }
  // array <-> brick, compare
  {
    const long dl[3] = {40, 32, 56}, pad[3] = {0, 0, 0}, gh[3] = {8, 8, 8};
    const size_t ne = (size_t)56 * 48 * 72;
    double *arr = dev_doubles(ne, 0.0), *arr2 = dev_doubles(ne, 0.0);
    for (size_t q = 0; q < ne; ++q) arr[q] = (double)q;
    uint32_t *grid_dev = dev_u32(grid.data(), grid.size());
    uint64_t *res = nullptr;
    REQ(hipMalloc((void **)&res, 16) == hipSuccess);
    REQ(brk_array_to_brick(dl, pad, gh, arr, grid_dev, in, 512, nullptr) == BRK_OK);
    REQ(brk_brick_to_array(dl, pad, gh, arr2, grid_dev, in, 512, nullptr) == BRK_OK);
    REQ(brk_compare_brick(dl, pad, gh, arr, grid_dev, in, 512, 1e-12, res, nullptr) == BRK_OK);
    REQ(res[0] == 0);
    REQ(hipFree(arr) == hipSuccess && hipFree(arr2) == hipSuccess && hipFree(grid_dev) == hipSuccess && hipFree(res) == hipSuccess);
  }
  REQ(hipFree(adj_dev) == hipSuccess && hipFree(in) == hipSuccess && hipFree(out) == hipSuccess);
  return 0;
}

// brk_storage_alloc / brk_storage_free / brk_placement_probe: small storages (plain allocation), measured ones (candidates, all but one
// freed again), storages of different sizes as partners, double free refused
static int placement() {
  void *small = nullptr, *a = nullptr, *b = nullptr, *c = nullptr;
  brk_placement rep;
  const size_t big = (size_t)65 << 20;
  const unsigned long live0 = hipmock_live_allocations();
  // by default storages below 512 MiB are not measured (the probe would run out of the Infinity Cache): plain allocations
  REQ(brk_storage_alloc(big, &a, &rep) == BRK_OK && a && rep.candidates == 0);
  REQ(brk_storage_alloc(big, &b, &rep) == BRK_OK && b && rep.candidates == 0);
  REQ(brk_storage_free(a) == BRK_OK && brk_storage_free(b) == BRK_OK);
  a = b = nullptr;
  setenv("BRK_PLACEMENT_MIN_BYTES", "67108864", 1); // the machinery on storages of 65 MiB
  REQ(brk_storage_alloc(4096, &small, &rep) == BRK_OK && small && rep.candidates == 0);
  REQ(brk_storage_alloc(big, &a, &rep) == BRK_OK && a && rep.candidates == 0); // nothing to place it against yet
  REQ(brk_storage_alloc(big + 4096, &b, &rep) == BRK_OK && b && rep.candidates >= 2 && rep.chosen >= 0 && rep.chosen < rep.candidates &&
      rep.ms_chosen > 0 && rep.ms_chosen <= rep.ms_worst && rep.probe_bricks == 23 * 23 * 23); // 65 MiB hold 25^3 bricks
  REQ(brk_storage_alloc(big - 4096 + 24, &c, nullptr) == BRK_OK && c); // a size that is no multiple of a brick, nor of 16 bytes
  REQ(hipmock_live_allocations() == live0 + 4); // the rejected candidates are gone
  { // named partners (brk_storage_alloc_ex): a second READ field of a's kind, an output against both; refused: foreign pointers, relations
    void *d = nullptr, *e = nullptr;
    brk_place_partner same[1] = {{a, BRK_PLACE_SAME}}, both[2] = {{a, BRK_PLACE_OPPOSITE}, {b, BRK_PLACE_OPPOSITE}};
    REQ(brk_storage_alloc_ex(big, same, 1, nullptr, &d, &rep) == BRK_OK && d && rep.candidates >= 2 && rep.ms_chosen > 0);
    REQ(brk_storage_alloc_ex(big, both, 2, nullptr, &e, &rep) == BRK_OK && e && rep.candidates >= 2 && rep.ms_chosen <= rep.ms_worst);
    brk_place_partner bad[1] = {{(char *)a + 4096, BRK_PLACE_OPPOSITE}}, rel[1] = {{a, 5}};
    void *f = nullptr;
    REQ(brk_storage_alloc_ex(big, bad, 1, nullptr, &f, nullptr) != BRK_OK && !f);
    REQ(brk_storage_alloc_ex(big, rel, 1, nullptr, &f, nullptr) != BRK_OK && !f);
    REQ(brk_storage_alloc_ex(big, nullptr, 3, nullptr, &f, nullptr) != BRK_OK);
    REQ(brk_storage_free(d) == BRK_OK && brk_storage_free(e) == BRK_OK);
    REQ(hipmock_live_allocations() == live0 + 4);
  }
  double ms = 0.0;
  REQ(brk_placement_probe(a, b, big, nullptr, &ms) == BRK_OK && ms > 0.0);
  REQ(brk_placement_probe(a, (char *)b + 8, big, nullptr, &ms) != BRK_OK); // misaligned
  REQ(brk_placement_probe(a, b, 26 * 4096, nullptr, &ms) != BRK_OK);         // fewer than 3 x 3 x 3 bricks
  REQ(brk_placement_probe(nullptr, b, 4096, nullptr, &ms) != BRK_OK);
  REQ(brk_storage_free(b) == BRK_OK && brk_storage_free(b) != BRK_OK); // second free: not ours any more
  REQ(brk_storage_free(a) == BRK_OK && brk_storage_free(c) == BRK_OK && brk_storage_free(small) == BRK_OK && brk_storage_free(nullptr) == BRK_OK);
  REQ(brk_storage_alloc(0, &a, nullptr) != BRK_OK && brk_storage_alloc(4096, nullptr, nullptr) != BRK_OK);
  REQ(hipmock_live_allocations() == live0);
  unsetenv("BRK_PLACEMENT_MIN_BYTES");
  return 0;
}

int main() {
  if (placement()) return 1;
  const uint32_t shapes[][3] = {{16, 16, 16}, {32, 16, 24}, {24, 40, 16}};
  for (auto &sh : shapes)
    for (int world : {1, 2, 3, 4, 5, 8, 12})
      for (size_t step : {(size_t)512, (size_t)1024})
        if (exchange_virtual_ranks(sh, world, step)) return 1;
  if (exchange_cached(nullptr)) return 1;
  // the same through a (one-rank, mocked) RCCL communicator: brk_comm_*, SELF_VIA_COMM plans, brk_exchange_transport
  {
    char id[128];
    void *comm = nullptr;
    REQ(brk_comm_unique_id(id) == BRK_OK);
    REQ(brk_comm_init_rank(&comm, 1, id, 0) == BRK_OK);
    int r = -1, n = -1;
    REQ(brk_comm_rank(comm, &r, &n) == BRK_OK && r == 0 && n == 1);
    if (exchange_cached(comm)) return 1;
    const uint32_t shape[3] = {24, 16, 32};
    brk_decomp *d = nullptr;
    REQ(brk_decomp_build(shape, 8, &d) == BRK_OK);
    const uint32_t nb = brk_decomp_nbricks(d);
    int ranks[27] = {0};
    for (int mode : {BRK_XCHG_PACKED, BRK_XCHG_DIRECT, BRK_XCHG_SHIFT}) {
      double *st = dev_doubles((size_t)nb * 512, 0.0);
      for (size_t q = 0; q < (size_t)nb * 512; ++q) st[q] = (double)q;
      brk_xplan *p = nullptr;
      brk_exchange_view *v = nullptr;
      REQ(brk_xplan_build_decomp(d, ranks, 0, mode, BRK_XCHG_SELF_VIA_COMM, 512, &p) == BRK_OK);
      REQ(brk_exchange_view_create(p, st, &v) == BRK_OK);
      REQ(brk_exchange_run(v, comm, nullptr) == BRK_OK);
      const brk_region *g = brk_decomp_ghost(d), *k = brk_decomp_skin(d);
      for (uint32_t q = 0; q < brk_decomp_nruns(d); ++q)
        for (size_t e = 0; e < (size_t)g[q].len * 512; e += 101) REQ(st[(size_t)g[q].pos * 512 + e] == (double)((size_t)k[q].pos * 512 + e));
      REQ(brk_exchange_run(v, nullptr, nullptr) != BRK_OK); // messages but no communicator: refused
      brk_exchange_view_free(v);
      brk_xplan_free(p);
      REQ(hipFree(st) == hipSuccess);
    }
    brk_decomp_free(d);
    REQ(brk_comm_destroy(comm) == BRK_OK);
  }
  if (launches()) return 1;
  printf("probe_asan_device_paths: ok, %lu launches emulated, %lu device allocations still live (the box-list cache)\n", hipmock_launches(),
         hipmock_live_allocations());
  return 0;
}
