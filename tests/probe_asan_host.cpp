// Host-side bookkeeping of libbrk.so (brk_host.cpp + brk_xplan.cpp: init_grid, BrickDecomp tables, rank maps, exchange plans,
// Z-Morton / strong partition) compiled with AddressSanitizer + UBSan and driven over a range of shapes.  GPU sanitizers are not
// available on the test pool; this covers the part of the library that is plain C++.  Built and run by tests/test_host_logic.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "brk.h"

// the one device-side symbol the host files reference (cached exchange views are freed with their decomposition): never reached
// here because no view is ever created
extern "C" void brk_exchange_view_free(brk_exchange_view *) {}

#define REQ(x)                                                              \
  do {                                                                      \
    if (!(x)) {                                                             \
      fprintf(stderr, "probe: %s failed (%s)\n", #x, brk_last_error());     \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main() {
  unsigned long long digest = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { digest = (digest ^ v) * 1099511628211ull; };
  // init_grid
  for (long n : {3L, 4L, 7L, 18L}) {
    const long dims[3] = {n, n + 1, n + 2};
    std::vector<uint32_t> grid((size_t)dims[0] * dims[1] * dims[2]), adj(grid.size() * 27);
    REQ(brk_init_grid(dims, grid.data(), adj.data()) == BRK_OK);
    for (uint32_t a : adj) mix(a);
  }
  // decompositions, rank maps, plans
  const uint32_t shapes[][3] = {{16, 16, 16}, {32, 16, 24}, {24, 40, 16}, {64, 64, 64}, {48, 16, 16}, {128, 16, 32}};
  for (auto &sh : shapes) {
    brk_decomp *d = nullptr;
    REQ(brk_decomp_build(sh, 8, &d) == BRK_OK);
    const uint32_t nb = brk_decomp_nbricks(d), nr = brk_decomp_nruns(d);
    const uint32_t *adj = brk_decomp_adj(d), *grid = brk_decomp_grid(d);
    uint32_t td[3], sp[3];
    brk_decomp_tdims(d, td);
    brk_decomp_sep_pos(d, sp);
    REQ(sp[2] == nb);
    for (size_t q = 0; q < (size_t)nb * 27; ++q) mix(adj[q]);
    for (size_t q = 0; q < (size_t)td[0] * td[1] * td[2]; ++q) mix(grid[q]);
    const brk_region *g = brk_decomp_ghost(d), *k = brk_decomp_skin(d);
    for (uint32_t q = 0; q < nr; ++q) mix(g[q].pos), mix(g[q].len), mix(k[q].pos), mix(k[q].len), mix(g[q].neighbor);
    for (int world : {1, 2, 3, 4, 6, 8, 12, 27}) {
      int cart[3];
      REQ(brk_dims_create(world, cart) == BRK_OK);
      for (int r = 0; r < world; ++r) {
        const int coo[3] = {r % cart[0], (r / cart[0]) % cart[1], r / (cart[0] * cart[1])};
        int ranks[27];
        REQ(brk_populate_rank_map(cart, coo, ranks) == BRK_OK);
        for (int mode : {BRK_XCHG_PACKED, BRK_XCHG_DIRECT, BRK_XCHG_SHIFT})
          for (unsigned flags : {0u, (unsigned)BRK_XCHG_SELF_VIA_COMM}) {
            brk_xplan *p = nullptr;
            REQ(brk_xplan_build_decomp(d, ranks, ranks[13], mode, flags, 512, &p) == BRK_OK);
            const brk_xcopy *c = brk_xplan_copies(p);
            const brk_xmsg *m = brk_xplan_msgs(p);
            for (uint32_t q = 0; q < brk_xplan_ncopies(p); ++q) mix(c[q].src_off), mix(c[q].dst_off), mix(c[q].bytes);
            for (uint32_t q = 0; q < brk_xplan_nmsgs(p); ++q) {
              mix(m[q].off), mix(m[q].bytes), mix((uint64_t)m[q].peer);
              mix(brk_xplan_buffer_bytes(p, m[q].stage, m[q].peer, m[q].is_send ? BRK_SP_SEND : BRK_SP_RECV));
            }
            mix(brk_xplan_bytes_remote(p)), mix(brk_xplan_bytes_local(p)), mix(brk_xplan_nstages(p));
            brk_xplan_free(p);
          }
      }
    }
    brk_decomp_free(d);
  }
  // error paths must fail cleanly
  {
    brk_decomp *d = nullptr;
    const uint32_t bad[3] = {12, 16, 16};
    REQ(brk_decomp_build(bad, 8, &d) != BRK_OK && d == nullptr);
    brk_xplan *p = nullptr;
    brk_xcopy c{0, 0, BRK_SP_STORAGE, BRK_SP_SEND, -1, 1, 8, 0, 4096};
    REQ(brk_xplan_build_tables(0, 1, &c, 0, nullptr, &p) != BRK_OK);
    int cart[3];
    REQ(brk_dims_create(0, cart) != BRK_OK);
  }
  // strong bookkeeping
  for (uint64_t subdim : {1ull, 2ull, 4ull, 8ull}) {
    const uint64_t all = subdim * subdim * subdim;
    uint64_t nbset[27];
    brk_allneighbors3(nbset);
    for (int size : {1, 2, 3, 5, 8}) {
      if ((uint64_t)size > all) continue;
      for (int r = 0; r < size; ++r) {
        uint64_t lr[2];
        brk_strong_partition(all, size, r, lr);
        mix(lr[0]), mix(lr[1]);
      }
      for (uint64_t z = 0; z < all; ++z)
        for (uint64_t s : nbset) {
          int owner = -1, slot = -1;
          mix(brk_strong_neighbor(s, z, subdim, all, size, &owner, &slot));
          mix((uint64_t)owner), mix((uint64_t)slot);
        }
    }
    for (uint64_t z = 0; z < all; ++z)
      for (uint64_t a = 0; a < 3; ++a) mix(brk_zmort_set(z, a, (brk_zmort_get(z, a) + 1) % subdim));
  }
  printf("probe_asan_host: ok digest %016llx\n", digest);
  return 0;
}
