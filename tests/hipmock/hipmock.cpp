// hipmock.cpp -- TEST INFRASTRUCTURE: a malloc-backed stand-in for the two dozen HIP runtime entry points libbrk.so and the C++
// example drivers use, so that their HOST code (views, plans, caches, launch paths, table construction) runs on a CPU under
// AddressSanitizer / UBSan.  GPU sanitizers are not available on the test pool; the kernels themselves are not what is tested
// here.  The product never links this file: tests/test_sanitized_host_paths.py compiles the .hip translation units with
// `clang++ -x hip --cuda-host-only -fsanitize=address,undefined` (the real HIP headers, host pass only) and links them with this
// file instead of libamdhip64.
//
//   * "device" memory is malloc memory, every allocation is registered: the emulated kernels check that whatever a launch would
//     touch (list entries, adjacency rows, bricks of `in` / `out`, link ranges, derived tables) lies inside ONE registered
//     allocation -- on top of what ASan sees of the accesses themselves;
//   * a launch arrives as hipLaunchKernel(stub address, grid, block, args); __hipRegisterFunction told us the kernel's mangled
//     name for every stub, the emulation is chosen by name: k_link_copy = memcpy per link (overlap = ASan report), the stencil
//     kernels = the oracle's brick sweep over the same list (oracle/brick_oracle.c, so the drivers' own validation passes),
//     k_face_table / k_link_mask / k_regular_mask / k_iter restated as host loops;
//   * streams are synchronous, events are heap objects (a double destroy or a leak is reported by ASan / LSan).
#include <hip/hip_runtime_api.h>

#include <cxxabi.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "brick-device.h"
#include "brk.h"

extern "C" {
// oracle/brick_oracle.c (test infrastructure)
void orc_brick_stencil(int id, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, long n, uint32_t first,
                       const double *coeff);
void orc_brick_taps(const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, long n, int ntaps, const int *offsets,
                    const double *weights);
}

namespace {

// registries live behind function-local statics: the module constructors of the kernel translation units call
// __hipRegisterFunction before this file's own globals would have been constructed
std::mutex &mu() {
  static std::mutex *m = new std::mutex;
  return *m;
}
std::map<uintptr_t, size_t> &allocs() { // base -> bytes
  static auto *m = new std::map<uintptr_t, size_t>;
  return *m;
}
std::map<const void *, std::string> &kernels() { // stub address -> demangled name
  static auto *m = new std::map<const void *, std::string>;
  return *m;
}
#define g_mu mu()
#define g_alloc allocs()
#define g_kernels kernels()
thread_local hipError_t g_last = hipSuccess;
struct Cfg {
  dim3 grid, block;
  size_t shmem;
  hipStream_t st;
};
thread_local std::vector<Cfg> g_cfg;
unsigned long g_launches = 0;

[[noreturn]] void die(const char *what, const void *p, size_t bytes) {
  fprintf(stderr, "hipmock: %s: %p + %zu is not inside one device allocation\n", what, p, bytes);
  abort();
}

void need(const char *what, const void *p, size_t bytes) {
  if (bytes == 0) return;
  std::lock_guard<std::mutex> lock(g_mu);
  const uintptr_t a = (uintptr_t)p;
  auto it = g_alloc.upper_bound(a);
  if (it == g_alloc.begin()) die(what, p, bytes);
  --it;
  if (a + bytes > it->first + it->second) die(what, p, bytes);
}

template <typename T>
T arg(void **args, int q) {
  T v;
  std::memcpy(&v, args[q], sizeof v);
  return v;
}

// template arguments of "…::k_name<a, b, c>(…)" as integers
std::vector<long> targs(const std::string &name, const char *kernel) {
  std::vector<long> out;
  size_t p = name.find(std::string(kernel) + "<");
  if (p == std::string::npos) return out;
  p += std::strlen(kernel) + 1;
  while (p < name.size() && name[p] != '>') {
    while (p < name.size() && !(name[p] == '-' || (name[p] >= '0' && name[p] <= '9')) && name[p] != '>') ++p; // skips "(brklib::Kind)"-style casts
    if (name[p] == '>') break;
    out.push_back(std::strtol(name.c_str() + p, nullptr, 10));
    while (p < name.size() && name[p] != ',' && name[p] != '>') ++p;
    if (name[p] == ',') ++p;
  }
  return out;
}

// every non-skipped entry of the launch: the brick itself in `in` and `out`, its adjacency row, all 27 neighbours in `in`
std::vector<uint32_t> checked_list(const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, uint32_t first, uint32_t n) {
  std::vector<uint32_t> list;
  if (ids) need("brick id list", ids, (size_t)n * 4);
  for (uint32_t q = 0; q < n; ++q) {
    const uint32_t b = ids ? ids[q] : first + q;
    if (b == BRK_LIST_SKIP) continue;
    need("adjacency row", adj + (size_t)b * 27, 27 * 4);
    need("output brick", out + (size_t)b * step, 512 * 8);
    for (int s = 0; s < 27; ++s) need("input brick (neighbour)", in + (size_t)adj[(size_t)b * 27 + s] * step, 512 * 8);
    if (adj[(size_t)b * 27 + 13] != b) {
      fprintf(stderr, "hipmock: adjacency row of brick %u names %u as itself\n", b, adj[(size_t)b * 27 + 13]);
      abort();
    }
    list.push_back(b);
  }
  return list;
}

void check_faces(const uint32_t *adj, const uint32_t *faces, const std::vector<uint32_t> &list) {
  if (!faces) return;
  using brklib::adj_slot;
  const int slots[6] = {adj_slot(-1, 0, 0), adj_slot(1, 0, 0), adj_slot(0, -1, 0), adj_slot(0, 1, 0), adj_slot(0, 0, -1), adj_slot(0, 0, 1)};
  for (uint32_t b : list) {
    need("compact face row", faces + (size_t)b * 8, 32);
    for (int f = 0; f < 6; ++f)
      if (faces[(size_t)b * 8 + f] != adj[(size_t)b * 27 + slots[f]]) {
        fprintf(stderr, "hipmock: face table of brick %u disagrees with its adjacency row\n", b);
        abort();
      }
  }
}

// brk_star7.hip: struct Box { uint32_t base, ni_tot, nj_tot, lo[3], cnt[3]; }
struct BoxMirror {
  uint32_t base, ni_tot, nj_tot, lo[3], cnt[3];
};

void run_stencil(int id, const uint32_t *adj, const double *in, double *out, size_t step, const std::vector<uint32_t> &list, const brklib::Coeffs &cf) {
  if (list.empty()) return;
  if (id == BRK_MPI7_X2 || id == BRK_7PT_X2) { // the fused double sweep: 25 taps, weights in cf.c (make_coeffs)
    int off[75];
    for (int t = 0; t < 25; ++t) off[3 * t] = brklib::kDiamondTaps[t].di, off[3 * t + 1] = brklib::kDiamondTaps[t].dj, off[3 * t + 2] = brklib::kDiamondTaps[t].dk;
    orc_brick_taps(adj, in, out, step, list.data(), (long)list.size(), 25, off, cf.c);
    return;
  }
  orc_brick_stencil(id, adj, in, out, step, list.data(), (long)list.size(), 0, cf.c);
}

int rel_of(int R, int di, int dj, int dk) {
  return R == 1 ? brklib::tap_rel<1>(di, dj, dk) : R == 2 ? brklib::tap_rel<2>(di, dj, dk) : R == 3 ? brklib::tap_rel<3>(di, dj, dk) : brklib::tap_rel<4>(di, dj, dk);
}

void launch(const std::string &name, dim3 grid, dim3 block, void **args) {
  (void)block;
  if (name.find("k_link_copy") != std::string::npos) {
    const brk_link *links = arg<const brk_link *>(args, 0);
    need("link table", links, (size_t)grid.y * sizeof(brk_link));
    for (unsigned q = 0; q < grid.y; ++q) {
      if ((uint64_t)grid.x * 65536 < links[q].bytes) {
        fprintf(stderr, "hipmock: link %u of %llu bytes is longer than the launch covers\n", q, (unsigned long long)links[q].bytes);
        abort();
      }
      need("link source", links[q].from, links[q].bytes);
      need("link destination", links[q].to, links[q].bytes);
      std::memcpy(links[q].to, links[q].from, links[q].bytes);
    }
    return;
  }
  if (name.find("k_expr<") != std::string::npos) { // (adj, in, out, step, ids, first, n, ExprProg): every tap gathered by the oracle, the ops interpreted per cell
    const uint32_t *adj = arg<const uint32_t *>(args, 0);
    const double *in = arg<const double *>(args, 1);
    double *out = arg<double *>(args, 2);
    const size_t step = arg<size_t>(args, 3);
    const brklib::ExprProg pg = arg<brklib::ExprProg>(args, 7);
    const int R = (int)targs(name, "k_expr").at(0);
    if (pg.radius != R || pg.n < 1 || pg.n > brklib::kMaxExprOps) abort();
    const std::vector<uint32_t> list = checked_list(adj, in, out, step, arg<const uint32_t *>(args, 4), arg<uint32_t>(args, 5), arg<uint32_t>(args, 6));
    if (list.empty()) return;
    uint32_t top = 0;
    for (uint32_t b : list) top = std::max(top, b);
    std::map<int, std::vector<double>> gathered; // rel -> in[p + offset] over the listed bricks (laid out like `out`)
    for (int t = 0; t < pg.n; ++t) {
      if (pg.op[t].code != brklib::EX_TAP || gathered.count(pg.op[t].rel)) continue;
      int off[3] = {9, 9, 9};
      for (int dk = -R; dk <= R; ++dk)
        for (int dj = -R; dj <= R; ++dj)
          for (int di = -R; di <= R; ++di)
            if (rel_of(R, di, dj, dk) == pg.op[t].rel) off[0] = di, off[1] = dj, off[2] = dk;
      if (off[0] == 9) abort();
      std::vector<double> &g = gathered[pg.op[t].rel];
      g.assign(((size_t)top + 1) * step, 0.0);
      const double one = 1.0;
      orc_brick_taps(adj, in, g.data(), step, list.data(), (long)list.size(), 1, off, &one);
    }
    for (uint32_t b : list)
      for (int cell = 0; cell < 512; ++cell) {
        double reg[brklib::kMaxExprRegs] = {};
        for (int t = 0; t < pg.n; ++t) {
          const brklib::ExprOp &o = pg.op[t];
          const double x = reg[o.a], y = reg[o.b], z = reg[o.c];
          double v = 0;
          switch (o.code) {
          case brklib::EX_TAP: v = gathered[o.rel][(size_t)b * step + cell]; break;
          case brklib::EX_CONST: v = o.imm; break;
          case brklib::EX_ADD: v = x + y; break;
          case brklib::EX_SUB: v = x - y; break;
          case brklib::EX_MUL: v = x * y; break;
          case brklib::EX_DIV: v = x / y; break;
          case brklib::EX_NEG: v = -x; break;
          case brklib::EX_MAX: v = x > y ? x : y; break;
          case brklib::EX_MIN: v = x < y ? x : y; break;
          case brklib::EX_ABS: v = x < 0 ? -x : x; break;
          case brklib::EX_SQRT: v = std::sqrt(x); break;
          case brklib::EX_GT: v = x > y; break;
          case brklib::EX_GE: v = x >= y; break;
          case brklib::EX_LT: v = x < y; break;
          case brklib::EX_LE: v = x <= y; break;
          case brklib::EX_EQ: v = x == y; break;
          case brklib::EX_NE: v = x != y; break;
          case brklib::EX_SEL: v = x != 0 ? y : z; break;
          default: abort();
          }
          reg[o.dst] = v;
        }
        out[(size_t)b * step + cell] = reg[pg.result];
      }
    return;
  }
  if (name.find("k_taps_far<") != std::string::npos) { // (adj, in, out, step, ids, first, n, TapsFar): table in "device" memory
    const brklib::TapsFar t = arg<brklib::TapsFar>(args, 7);
    const int R = (int)targs(name, "k_taps_far").at(0);
    need("tap offsets", t.rel, (size_t)t.n * 4);
    need("tap weights", t.w, (size_t)t.n * 8);
    std::vector<int> off;
    for (int q = 0; q < t.n; ++q) {
      bool found = false;
      for (int dk = -R; dk <= R && !found; ++dk)
        for (int dj = -R; dj <= R && !found; ++dj)
          for (int di = -R; di <= R && !found; ++di)
            if (rel_of(R, di, dj, dk) == t.rel[q]) off.insert(off.end(), {di, dj, dk}), found = true;
      if (!found) abort();
    }
    const std::vector<uint32_t> list = checked_list(arg<const uint32_t *>(args, 0), arg<const double *>(args, 1), arg<double *>(args, 2), arg<size_t>(args, 3),
                                                    arg<const uint32_t *>(args, 4), arg<uint32_t>(args, 5), arg<uint32_t>(args, 6));
    if (!list.empty())
      orc_brick_taps(arg<const uint32_t *>(args, 0), arg<const double *>(args, 1), arg<double *>(args, 2), arg<size_t>(args, 3), list.data(), (long)list.size(), t.n,
                     off.data(), t.w);
    return;
  }
  if (name.find("k_star_taps<") != std::string::npos) { // (adj, in, out, step, ids, first, n, StarTaps)
    const brklib::StarTaps t = arg<brklib::StarTaps>(args, 7);
    std::vector<int> off;
    for (int q = 0; q < t.n; ++q) off.insert(off.end(), {t.axis[q] == 0 ? t.off[q] : 0, t.axis[q] == 1 ? t.off[q] : 0, t.axis[q] == 2 ? t.off[q] : 0});
    const std::vector<uint32_t> list = checked_list(arg<const uint32_t *>(args, 0), arg<const double *>(args, 1), arg<double *>(args, 2), arg<size_t>(args, 3),
                                                    arg<const uint32_t *>(args, 4), arg<uint32_t>(args, 5), arg<uint32_t>(args, 6));
    if (!list.empty())
      orc_brick_taps(arg<const uint32_t *>(args, 0), arg<const double *>(args, 1), arg<double *>(args, 2), arg<size_t>(args, 3), list.data(), (long)list.size(), t.n,
                     off.data(), t.w);
    return;
  }
  const bool s7 = name.find("k_s7_wg<") != std::string::npos, swg = name.find("k_star_wg<") != std::string::npos,
             skp = name.find("k_star_kp<") != std::string::npos || name.find("k_star_km<") != std::string::npos, cd = name.find("k_cube_dense<") != std::string::npos || name.find("k_cube_dma<") != std::string::npos,
             tp = name.find("k_taps<") != std::string::npos;
  if (s7 || swg || skp || cd || tp) {
    const uint32_t *adj = arg<const uint32_t *>(args, 0);
    const double *in = arg<const double *>(args, 1);
    double *out = arg<double *>(args, 2);
    const size_t step = arg<size_t>(args, 3);
    const uint32_t *ids = arg<const uint32_t *>(args, 4);
    const uint32_t first = arg<uint32_t>(args, 5), n = arg<uint32_t>(args, 6);
    if (tp) {
      const brklib::Taps t = arg<brklib::Taps>(args, 7);
      const std::vector<long> ta = targs(name, "k_taps");
      const int R = (int)ta.at(0);
      std::vector<int> off;
      for (int q = 0; q < t.n; ++q) {
        bool found = false;
        for (int dk = -R; dk <= R && !found; ++dk)
          for (int dj = -R; dj <= R && !found; ++dj)
            for (int di = -R; di <= R && !found; ++di)
              if (rel_of(R, di, dj, dk) == t.rel[q]) off.insert(off.end(), {di, dj, dk}), found = true;
        if (!found) abort();
      }
      const std::vector<uint32_t> list = checked_list(adj, in, out, step, ids, first, n);
      if (!list.empty()) orc_brick_taps(adj, in, out, step, list.data(), (long)list.size(), t.n, off.data(), t.w);
      return;
    }
    if (s7) {
      const std::vector<long> ta = targs(name, "k_s7_wg"); // <kind, NW, MODE, DIAG>
      const int id = ta.at(0) == 1 ? BRK_MPI7 : (ta.at(0) == 2 ? BRK_COND7 : BRK_7PT_COEFF);
      const BoxMirror bx = arg<BoxMirror>(args, 7);
      const brklib::Coeffs cf = arg<brklib::Coeffs>(args, 8);
      const uint32_t *regmask = arg<const uint32_t *>(args, 9), *faces = arg<const uint32_t *>(args, 10);
      std::vector<uint32_t> list;
      if (ta.at(2) == 2) { // box mode: ids by arithmetic
        std::vector<uint32_t> box;
        for (uint32_t k = 0; k < bx.cnt[2]; ++k)
          for (uint32_t j = 0; j < bx.cnt[1]; ++j)
            for (uint32_t i = 0; i < bx.cnt[0]; ++i) box.push_back(bx.base + ((bx.lo[2] + k) * bx.nj_tot + bx.lo[1] + j) * bx.ni_tot + bx.lo[0] + i);
        if (box.size() != n) abort();
        void *tmp = nullptr; // registered scratch so that checked_list can treat it as a device list
        (void)hipMalloc(&tmp, box.size() * 4);
        std::memcpy(tmp, box.data(), box.size() * 4);
        list = checked_list(adj, in, out, step, (const uint32_t *)tmp, 0, n);
        (void)hipFree(tmp);
        if (regmask)
          for (uint32_t b : list) {
            need("regularity mask", regmask + (b >> 5), 4);
            const bool bit = (regmask[b >> 5] >> (b & 31)) & 1u;
            const uint32_t *row = adj + (size_t)b * 27;
            const uint32_t s1 = bx.ni_tot, s2 = bx.ni_tot * bx.nj_tot;
            using brklib::adj_slot;
            const bool regular = row[adj_slot(0, 0, -1)] == b - 1 && row[adj_slot(0, 0, 1)] == b + 1 && row[adj_slot(0, -1, 0)] == b - s1 &&
                                 row[adj_slot(0, 1, 0)] == b + s1 && row[adj_slot(-1, 0, 0)] == b - s2 && row[adj_slot(1, 0, 0)] == b + s2;
            if (bit && !regular) {
              fprintf(stderr, "hipmock: regularity bit set for irregular brick %u\n", b);
              abort();
            }
          }
      } else {
        list = checked_list(adj, in, out, step, ids, first, n);
      }
      check_faces(adj, faces, list);
      run_stencil(id, adj, in, out, step, list, cf);
      return;
    }
    const brklib::Coeffs cf = arg<brklib::Coeffs>(args, 7);
    const std::vector<uint32_t> list = checked_list(adj, in, out, step, ids, first, n);
    if (cd) {
      const bool dma = name.find("k_cube_dma<") != std::string::npos; // (..., cf, linkmask, faces, cubemask, cs1, cs2): no `diag` argument
      const std::vector<long> ta = targs(name, dma ? "k_cube_dma" : "k_cube_dense");
      const uint32_t *linkmask = arg<const uint32_t *>(args, dma ? 8 : 9);
      if (dma && (!linkmask || !arg<const uint32_t *>(args, 9))) {
        fprintf(stderr, "hipmock: k_cube_dma launched without the tables of a registered adjacency\n");
        abort();
      }
      if (linkmask)
        for (uint32_t b : list) need("link mask", linkmask + 4 * (size_t)(b >> 6), 16);
      if (dma) check_faces(adj, arg<const uint32_t *>(args, 9), list);
      // box entry point over a registered table: a set bit promises that the row the kernel computes IS the row of the table
      const uint32_t *cubemask = arg<const uint32_t *>(args, 10);
      const uint32_t cs1 = arg<uint32_t>(args, 11), cs2 = arg<uint32_t>(args, 12);
      if (cubemask)
        for (uint32_t b : list) {
          need("cube regularity mask", cubemask + (b >> 5), 4);
          if (!((cubemask[b >> 5] >> (b & 31)) & 1u)) continue;
          for (int q = 0; q < 27; ++q)
            if (adj[(size_t)b * 27 + q] != b + (uint32_t)(q / 9 - 1) * cs2 + (uint32_t)((q / 3) % 3 - 1) * cs1 + (uint32_t)(q % 3 - 1)) {
              fprintf(stderr, "hipmock: cube regularity bit set for brick %u whose entry %d is not the predicted one\n", b, q);
              abort();
            }
        }
      // the kernel cannot tell BRK_MPI7_X2 from BRK_7PT_X2 either: both arrive as 25 weights
      run_stencil(ta.at(0) == 0 ? BRK_27PT_COEFF : (ta.at(0) == 1 ? BRK_MPI125 : BRK_MPI7_X2), adj, in, out, step, list, cf);
      return;
    }
    const std::vector<long> ta = targs(name, swg ? "k_star_wg" : (name.find("k_star_km<") != std::string::npos ? "k_star_km" : "k_star_kp"));
    check_faces(adj, arg<const uint32_t *>(args, 8), list);
    run_stencil(ta.at(0) == brklib::kStarMpi13 ? BRK_MPI13 : BRK_MPI25, adj, in, out, step, list, cf);
    return;
  }
  if (name.find("k_all_regular") != std::string::npos) { // (ids, n, mask, flag): *flag &= every listed brick has its bit set
    const uint32_t *ids = arg<const uint32_t *>(args, 0);
    const uint32_t n = arg<uint32_t>(args, 1);
    const uint32_t *mask = arg<const uint32_t *>(args, 2);
    uint32_t *flag = arg<uint32_t *>(args, 3);
    need("brick id list", ids, (size_t)n * 4);
    need("flag", flag, 4);
    for (uint32_t q = 0; q < n; ++q) {
      if (ids[q] == BRK_LIST_SKIP) continue;
      need("cube regularity mask", mask + (ids[q] >> 5), 4);
      if (!((mask[ids[q] >> 5] >> (ids[q] & 31)) & 1u)) *flag = 0u;
    }
    return;
  }
  if (name.find("k_cube125_q<") != std::string::npos) { // (adj, in, out, step, ids, n, cf, cs1, cs2)
    const uint32_t *adj = arg<const uint32_t *>(args, 0);
    const double *in = arg<const double *>(args, 1);
    double *out = arg<double *>(args, 2);
    const size_t step = arg<size_t>(args, 3);
    const uint32_t *ids = arg<const uint32_t *>(args, 4);
    const uint32_t n = arg<uint32_t>(args, 5);
    const brklib::Coeffs cf = arg<brklib::Coeffs>(args, 6);
    const uint32_t cs1 = arg<uint32_t>(args, 7), cs2 = arg<uint32_t>(args, 8);
    need("brick id list", ids, (size_t)n * 4);
    // what the launcher must have proven: padding in whole groups of four, the four entries of a group are consecutive i-neighbours,
    // every brick's row is the predicted one (the kernel computes it and loads no fallback halo)
    for (uint32_t g = 0; g + 3 < n || g < n; g += 4) {
      int skips = 0;
      for (uint32_t q = g; q < g + 4 && q < n; ++q) skips += ids[q] == BRK_LIST_SKIP;
      if (skips != 0 && skips != (int)std::min<uint32_t>(4, n - g)) {
        fprintf(stderr, "hipmock: k_cube125_q over a group that is partly padding (entries %u..)\n", g);
        abort();
      }
      if (skips) continue;
      if (g + 4 > n) {
        fprintf(stderr, "hipmock: k_cube125_q over an incomplete last group\n");
        abort();
      }
      for (uint32_t q = g; q < g + 4; ++q) {
        const uint32_t b = ids[q];
        need("adjacency row", adj + (size_t)b * 27, 108);
        for (int e = 0; e < 27; ++e)
          if (adj[(size_t)b * 27 + e] != b + (uint32_t)(e / 9 - 1) * cs2 + (uint32_t)((e / 3) % 3 - 1) * cs1 + (uint32_t)(e % 3 - 1)) {
            fprintf(stderr, "hipmock: k_cube125_q over brick %u whose entry %d is not the predicted one\n", b, e);
            abort();
          }
        if (q > g && ids[q] != ids[q - 1] + 1) {
          fprintf(stderr, "hipmock: k_cube125_q over a group whose entries are not consecutive i-neighbours\n");
          abort();
        }
      }
    }
    const std::vector<uint32_t> list = checked_list(adj, in, out, step, ids, 0, n);
    run_stencil(BRK_MPI125, adj, in, out, step, list, cf);
    return;
  }
  if (name.find("k_face_table") != std::string::npos) {
    const uint32_t *adj = arg<const uint32_t *>(args, 0);
    const uint32_t nb = arg<uint32_t>(args, 1);
    uint32_t *faces = arg<uint32_t *>(args, 2);
    need("adjacency table", adj, (size_t)nb * 27 * 4);
    need("face table", faces, (size_t)nb * 32);
    using brklib::adj_slot;
    for (uint32_t b = 0; b < nb; ++b) {
      const uint32_t *row = adj + (size_t)b * 27;
      const uint32_t v[8] = {row[adj_slot(-1, 0, 0)], row[adj_slot(1, 0, 0)], row[adj_slot(0, -1, 0)], row[adj_slot(0, 1, 0)],
                             row[adj_slot(0, 0, -1)], row[adj_slot(0, 0, 1)], 0u, 0u};
      std::memcpy(faces + (size_t)b * 8, v, 32);
    }
    return;
  }
  if (name.find("k_link_mask") != std::string::npos || name.find("k_regular_mask") != std::string::npos) {
    const bool reg = name.find("k_regular_mask") != std::string::npos;
    const uint32_t *adj = arg<const uint32_t *>(args, 0);
    const uint32_t nb = arg<uint32_t>(args, 1);
    const uint32_t s1 = reg ? arg<uint32_t>(args, 2) : 0, s2 = reg ? arg<uint32_t>(args, 3) : 0;
    uint32_t *mask = arg<uint32_t *>(args, reg ? 4 : 2);
    const int cube = reg ? arg<int>(args, 5) : 0;
    need("adjacency table", adj, (size_t)nb * 27 * 4);
    using brklib::adj_slot;
    // what the launch writes: per wave of 64 threads two (regular) or four (link) words, also for the waves past nbricks
    const uint32_t threads = grid.x * block.x;
    need("mask", mask, (size_t)(threads / 64) * (reg ? 8 : 16));
    std::memset(mask, 0, (size_t)(threads / 64) * (reg ? 8 : 16));
    auto linked_pair = [&](const uint32_t *left, const uint32_t *right) {
      bool ok = true;
      for (int q = 0; q < 27; ++q)
        if (q % 3 != 2) ok = ok && right[q] == left[q + 1];
      return ok;
    };
    for (uint32_t b = 0; b < nb; ++b) {
      const uint32_t *row = adj + (size_t)b * 27;
      if (reg) {
        bool ok = row[adj_slot(0, 0, -1)] == b - 1 && row[adj_slot(0, 0, 1)] == b + 1 && row[adj_slot(0, -1, 0)] == b - s1 &&
                  row[adj_slot(0, 1, 0)] == b + s1 && row[adj_slot(-1, 0, 0)] == b - s2 && row[adj_slot(1, 0, 0)] == b + s2 && b >= s2 &&
                  b + s2 < nb;
        if (cube) {
          ok = ok && b >= s2 + s1 + 1 && (uint64_t)b + s2 + s1 + 1 < nb;
          for (int q = 0; q < 27 && ok; ++q) ok = row[q] == b + (uint32_t)(q / 9 - 1) * s2 + (uint32_t)((q / 3) % 3 - 1) * s1 + (uint32_t)(q % 3 - 1);
        }
        if (ok) mask[b >> 5] |= 1u << (b & 31);
      } else {
        const uint32_t l = row[adj_slot(0, 0, -1)], r = row[adj_slot(0, 0, 1)];
        const bool okl = l < nb && linked_pair(adj + (size_t)l * 27, row), okr = r < nb && linked_pair(row, adj + (size_t)r * 27);
        uint32_t *w = mask + 4 * (size_t)(b >> 6);
        const uint32_t bit = b & 63;
        if (okl) w[bit < 32 ? 0 : 2] |= 1u << (bit & 31);
        if (okr) w[bit < 32 ? 1 : 3] |= 1u << (bit & 31);
      }
    }
    return;
  }
  if (name.find("k_iter<") != std::string::npos) {
    // (IterGeom g, double *arr, const uint32_t *grid, double *bdat, size_t step, double tol, unsigned long long *result)
    struct Geom {
      long sA1, sA2, sB1, sB2, pad[3], lo[3], cnt[3];
    };
    const Geom g = arg<Geom>(args, 0);
    double *arr = arg<double *>(args, 1);
    const uint32_t *bgrid = arg<const uint32_t *>(args, 2);
    double *bdat = arg<double *>(args, 3);
    const size_t step = arg<size_t>(args, 4);
    const double tol = arg<double>(args, 5);
    unsigned long long *result = arg<unsigned long long *>(args, 6);
    const long mode = targs(name, "k_iter").at(0);
    unsigned long long bad = 0;
    double worst = 0.0;
    for (long tk = g.lo[2]; tk < g.lo[2] + g.cnt[2]; ++tk)
      for (long tj = g.lo[1]; tj < g.lo[1] + g.cnt[1]; ++tj)
        for (long ti = g.lo[0]; ti < g.lo[0] + g.cnt[0]; ++ti) {
          need("brick grid", bgrid + tk * g.sB2 + tj * g.sB1 + ti, 4);
          const uint32_t b = bgrid[tk * g.sB2 + tj * g.sB1 + ti];
          double *bp = bdat + (size_t)b * step;
          need("brick", bp, 4096);
          for (int k = 0; k < 8; ++k)
            for (int j = 0; j < 8; ++j) {
              double *ap = arr + (g.pad[2] + tk * 8 + k) * g.sA2 + (g.pad[1] + tj * 8 + j) * g.sA1 + (g.pad[0] + ti * 8);
              need("array row", ap, 64);
              for (int i = 0; i < 8; ++i) {
                double &x = bp[k * 64 + j * 8 + i];
                if (mode == 0) x = ap[i];
                else if (mode == 1) ap[i] = x;
                else {
                  const double d = std::fabs(x - ap[i]);
                  bad += ((d < tol) || (d < (std::fabs(x) + std::fabs(ap[i])) * tol)) ? 0 : 1;
                  worst = d > worst ? d : worst;
                }
              }
            }
        }
    if (mode == 2) {
      need("compare result", result, 16);
      result[0] += bad;
      unsigned long long w;
      std::memcpy(&w, &worst, 8);
      result[1] = w > result[1] ? w : result[1];
    }
    return;
  }
  fprintf(stderr, "hipmock: no emulation for kernel %s\n", name.c_str());
  abort();
}

} // namespace

extern "C" {

void **__hipRegisterFatBinary(const void *) {
  static void *handle = nullptr;
  return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *hostFunction, char *, const char *deviceName, unsigned int, uint3 *, uint3 *, dim3 *, dim3 *, int *) {
  int status = 0;
  char *dm = abi::__cxa_demangle(deviceName, nullptr, nullptr, &status);
  std::lock_guard<std::mutex> lock(g_mu);
  g_kernels[hostFunction] = dm ? dm : deviceName;
  free(dm);
}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t st) {
  g_cfg.push_back({grid, block, shmem, st});
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *st) {
  const Cfg c = g_cfg.back();
  g_cfg.pop_back();
  *grid = c.grid, *block = c.block, *shmem = c.shmem, *st = c.st;
  return hipSuccess;
}

hipError_t hipLaunchKernel(const void *f, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t) {
  std::string name;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_kernels.find(f);
    if (it == g_kernels.end()) {
      fprintf(stderr, "hipmock: launch of an unregistered kernel\n");
      abort();
    }
    name = it->second;
    ++g_launches;
  }
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 || grid.y > 65535 || shmem > 160 * 1024) {
    g_last = hipErrorInvalidConfiguration;
    return g_last;
  }
  launch(name, grid, block, args);
  return hipSuccess;
}

hipError_t hipMalloc(void **p, size_t bytes) {
  *p = malloc(bytes ? bytes : 1);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lock(g_mu);
  g_alloc[(uintptr_t)*p] = bytes;
  return hipSuccess;
}
hipError_t hipFree(void *p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_alloc.erase((uintptr_t)p)) {
      fprintf(stderr, "hipmock: hipFree(%p): not a live device allocation (double free?)\n", p);
      abort();
    }
  }
  free(p);
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  if (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice) need("hipMemcpy destination", dst, bytes);
  if (kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice) need("hipMemcpy source", src, bytes);
  std::memcpy(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t) { return hipMemcpy(dst, src, bytes, kind); }
hipError_t hipMemset(void *dst, int v, size_t bytes) {
  need("hipMemset", dst, bytes);
  std::memset(dst, v, bytes);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int v, size_t bytes, hipStream_t) { return hipMemset(dst, v, bytes); }
hipError_t hipGetDevice(int *d) {
  *d = 0;
  return hipSuccess;
}
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
// run-time compiled modules (brk_expr_compile): there is nothing to load a code object on
hipError_t hipModuleLoadData(hipModule_t *, const void *) { return g_last = hipErrorNotSupported; }
hipError_t hipModuleGetFunction(hipFunction_t *, hipModule_t, const char *) { return g_last = hipErrorNotSupported; }
hipError_t hipModuleUnload(hipModule_t) { return g_last = hipErrorNotSupported; }
hipError_t hipModuleLaunchKernel(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t, void **, void **) {
  return g_last = hipErrorNotSupported;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) {
  std::memset(p, 0, sizeof *p);
  std::strcpy(p->gcnArchName, "gfx950");
  return hipSuccess;
}
hipError_t hipGetDeviceCount(int *n) {
  *n = 1;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) {
  *s = (hipStream_t)malloc(8);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) {
  free(s);
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipGetLastError(void) {
  const hipError_t e = g_last;
  g_last = hipSuccess;
  return e;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "hipmock error"; }
hipError_t hipEventCreate(hipEvent_t *e) {
  *e = (hipEvent_t)malloc(16);
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  free(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  *(volatile char *)e = 1; // touches the object: a destroyed event is a use-after-free report
  return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) {
  (void)*(volatile char *)e;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { return hipEventQuery(e); }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
  // successive intervals differ, so that code which chooses by measured time (brk_storage_alloc) does not always take the first
  static int turn = 0;
  static const float pattern[5] = {1.0f, 0.9f, 1.1f, 0.8f, 1.05f};
  *ms = pattern[turn++ % 5];
  return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
  *free_b = (size_t)64 << 30, *total_b = (size_t)288 << 30;
  return hipSuccess;
}

// what a test prints at the end: device allocations still live, launches emulated
unsigned long hipmock_live_allocations(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  return (unsigned long)g_alloc.size();
}
unsigned long hipmock_launches(void) { return g_launches; }
}
