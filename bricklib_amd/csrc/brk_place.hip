// brk_place.hip -- WHERE in HBM the brick storages of a process lie (brk_storage_alloc / brk_storage_free / brk_placement_probe).
//
// Measured on MI355X (profiles/r04_g_*): for a bandwidth-bound sweep the device allocations of a process fall into two kinds, and a
// sweep whose input storage and output storage are of the same kind runs 2.5-3 % slower (7-point, 512^3: 0.368-0.372 ms against
// 0.359-0.362 ms, every pair of 33 storages of 1.1 GiB in one process) than between storages of different kinds -- presumably the
// read stream and the write stream then meet in the same DRAM banks; no counter of this profiler shows it.  Which kind a hipMalloc
// gets is the driver's business and is not visible from the virtual address: storages allocated one after the other are mostly of
// the same kind (runs of 3 to 11 consecutive 1.1-GiB allocations), sometimes not -- the "fast box / slow box" spread of the earlier
// rounds' numbers.
// The reference allocates a storage wherever the allocator puts it (include/brick.h:68-75, include/brick-gpu.h:80-99); here
// the allocator of brick storages MEASURES: candidate allocations are made (6 to 24, all held until the choice is made so that each
// lies somewhere else), a 7-point sweep from the storage allocated before into each candidate is timed, the candidate with the
// fastest sweep is kept and the others are freed.  For the sweep the storages of a process fall into two kinds (same kind: slow,
// different kinds: fast; clean to 0.3 %); the search stops once both rates have been seen among at least six candidates -- the slow
// rate of the device is measured first (a sweep between the two halves of ONE allocation) -- and moves on behind untouched spacer
// allocations when every back-to-back candidate is of the partner's kind.  50-150 ms per storage, at setup.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "brk_device.h"

namespace brklib {
namespace {

struct Placed {
  void *p;
  size_t bytes;
  int device;
};
std::mutex g_place_mutex;
std::vector<Placed> g_placed; // live storages handed out by brk_storage_alloc, in allocation order

constexpr size_t kMinPlaced = (size_t)512 << 20;     // smaller storages are not measured: the probe would run out of the 256-MiB Infinity Cache (min_placed_bytes)
constexpr size_t kCandidateBudget = (size_t)28 << 30; // memory held by back-to-back candidates while one storage is placed
constexpr int kMinCandidates = 6, kMaxCandidates = 24; // runs of up to 11 consecutive allocations of one kind are common
// ... and runs of 25 x 1.2 GB and a 45-GiB allocation of one kind have been seen: when the back-to-back candidates show one rate only,
// SPACERS (allocated, never touched, freed with the rejected candidates) move the next candidates further on
// (round 5: one box of the pool showed ONE kind through 30 candidates spread over 65 GiB -- 361.6 instead of 372 GStencil/s for that
// process, profiles/r05_k_*; the next process on the same box found the other kind at its 22nd candidate -- so the search now goes on to
// ~150 GiB behind the partner before it settles)
constexpr size_t kSpacerGiB[] = {4, 8, 16, 16, 16, 32, 32};
constexpr size_t kHeldBudget = (size_t)136 << 30; // everything held at once while one storage is placed, and never more than 45 % of what is free
constexpr uint32_t kMaxProbeEdge = 82;                 // bricks per axis of the probe grid: at most 82^3 x 4 KiB = 2.3 GB swept per launch

// The measurement is the production sweep itself: the 7-point stencil (brk_stencil_apply_box, the kernel behind the headline) from
// storage a into storage b over the largest cubic lexicographic brick grid both can hold, one brick of ghost ring around the swept
// interior -- exactly the shape of the single/ driver.  A streaming copy was tried first and is NOT a substitute: between the same
// pairs of storages a linear copy, brick copies in range order, in XCD eighths and in slab order all show several levels that only
// partly follow the two levels of the sweep (profiles/r04_g_placement_probe_candidates.txt).
struct ProbeGrid {
  uint32_t edge = 0;
  uint32_t *adj = nullptr; // device, [edge^3][27]
  ~ProbeGrid() {
    if (adj) (void)hipFree(adj);
  }
};

int make_probe_grid(size_t bytes, ProbeGrid &g) {
  const size_t nb = bytes / 4096;
  uint32_t e = 3;
  while ((size_t)(e + 1) * (e + 1) * (e + 1) <= nb && e < kMaxProbeEdge) ++e;
  if ((size_t)e * e * e > nb) return brk_fail(BRK_EINVAL, "placement probe: %zu bytes hold fewer than 3 x 3 x 3 bricks", bytes);
  std::vector<uint32_t> grid((size_t)e * e * e), adj((size_t)e * e * e * 27);
  const long dims[3] = {(long)e, (long)e, (long)e};
  if (int rc = brk_init_grid(dims, grid.data(), adj.data())) return rc;
  HIP_TRY(hipMalloc(&g.adj, adj.size() * sizeof(uint32_t)));
  HIP_TRY(hipMemcpy(g.adj, adj.data(), adj.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  g.edge = e;
  return BRK_OK;
}

int probe_locked(const ProbeGrid &g, const void *a, void *b, hipStream_t st, double *ms_out) {
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  hipError_t r = hipEventCreate(&e1);
  if (r != hipSuccess) {
    (void)hipEventDestroy(e0);
    HIP_TRY(r);
  }
  const uint32_t tot[3] = {g.edge, g.edge, g.edge}, lo[3] = {1, 1, 1}, cnt[3] = {g.edge - 2, g.edge - 2, g.edge - 2};
  constexpr int kWarm = 3, kTimed = 10;
  int rc = BRK_OK;
  auto sweep = [&]() { return brk_stencil_apply_box(BRK_MPI7, g.adj, static_cast<const double *>(a), static_cast<double *>(b), 512, 0, tot, lo, cnt, nullptr, st); };
  for (int q = 0; q < kWarm && rc == BRK_OK; ++q) rc = sweep();
  if (rc == BRK_OK) r = hipEventRecord(e0, st);
  for (int q = 0; q < kTimed && rc == BRK_OK && r == hipSuccess; ++q) rc = sweep();
  if (rc == BRK_OK && r == hipSuccess) r = hipEventRecord(e1, st);
  if (rc == BRK_OK && r == hipSuccess) r = hipEventSynchronize(e1);
  float ms = 0.f;
  if (rc == BRK_OK && r == hipSuccess) r = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc) return rc;
  HIP_TRY(r);
  *ms_out = (double)ms / kTimed;
  return BRK_OK;
}

} // namespace
} // namespace brklib

using namespace brklib;

extern "C" int brk_placement_probe(const void *a_dev, void *b_dev, size_t bytes, brk_stream s, double *ms_per_sweep) {
  if (!a_dev || !b_dev || !ms_per_sweep) return brk_fail(BRK_EINVAL, "brk_placement_probe: null argument");
  if (((uintptr_t)a_dev | (uintptr_t)b_dev) % 16) return brk_fail(BRK_EINVAL, "brk_placement_probe: storages must be 16-byte aligned");
  ProbeGrid g;
  if (int rc = make_probe_grid(bytes, g)) return rc;
  return probe_locked(g, a_dev, b_dev, static_cast<hipStream_t>(s), ms_per_sweep);
}

// Minimum size of a storage (and of every partner) for a measurement: below it the probe sweep stays inside the 256-MiB Infinity Cache
// and neither kind shows (the candidates would be picked by noise, behind up to 60 GiB of spacer allocations).  Input plus output of a
// probe must clearly exceed the cache: 512 MiB per storage.  BRK_PLACEMENT_MIN_BYTES overrides it (tests of the machinery on small
// storages; read at every call).
static size_t min_placed_bytes() {
  if (const char *v = std::getenv("BRK_PLACEMENT_MIN_BYTES")) {
    const long long n = std::atoll(v);
    if (n > 0) return (size_t)n;
  }
  return kMinPlaced;
}

extern "C" int brk_storage_alloc_ex(size_t bytes, const brk_place_partner *partners_in, uint32_t npartners, brk_stream s, void **out_dev,
                                    brk_placement *report) {
  if (!out_dev) return brk_fail(BRK_EINVAL, "brk_storage_alloc: null result pointer");
  *out_dev = nullptr;
  if (report) std::memset(report, 0, sizeof *report);
  if (bytes == 0) return brk_fail(BRK_EINVAL, "brk_storage_alloc: zero bytes");
  if (npartners && !partners_in) return brk_fail(BRK_EINVAL, "brk_storage_alloc_ex: %u partners but no table", npartners);
  if (npartners > 8) return brk_fail(BRK_EINVAL, "brk_storage_alloc_ex: at most 8 partners");
  hipStream_t st = static_cast<hipStream_t>(s);
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_place_mutex);
  const size_t min_bytes = min_placed_bytes();
  struct Partner {
    Placed p;
    int relation;
  };
  std::vector<Partner> partners;
  if (npartners == 0) {
    // default: the storage a sweep will pair this one with is the most recent live one of this device that is worth measuring, and
    // the pair is a read / write pair (OPPOSITE kinds): a chain of storages placed each against its predecessor alternates A B A B --
    // what ping-pong pairs and the three rotating storages of the strong driver (0 -> 1 -> 2 -> 1 -> 0) need
    for (size_t q = g_placed.size(); q-- > 0 && partners.empty();)
      if (g_placed[q].device == device && g_placed[q].bytes >= min_bytes) partners.push_back({g_placed[q], BRK_PLACE_OPPOSITE});
  } else {
    for (uint32_t q = 0; q < npartners; ++q) {
      if (partners_in[q].relation != BRK_PLACE_OPPOSITE && partners_in[q].relation != BRK_PLACE_SAME)
        return brk_fail(BRK_EINVAL, "brk_storage_alloc_ex: partner %u: relation %d (BRK_PLACE_OPPOSITE or BRK_PLACE_SAME)", q, partners_in[q].relation);
      const Placed *found = nullptr;
      for (const Placed &p : g_placed)
        if (p.p == partners_in[q].storage_dev) found = &p;
      if (!found) return brk_fail(BRK_EINVAL, "brk_storage_alloc_ex: partner %u (%p) did not come from brk_storage_alloc", q, partners_in[q].storage_dev);
      if (found->device != device) return brk_fail(BRK_EINVAL, "brk_storage_alloc_ex: partner %u lives on device %d, the current device is %d", q, found->device, device);
      partners.push_back({*found, partners_in[q].relation});
    }
  }
  bool measurable = bytes >= min_bytes && !partners.empty();
  for (const Partner &p : partners) measurable = measurable && p.p.bytes >= min_bytes;
  void *chosen = nullptr;
  const char *off = std::getenv("BRK_PLACEMENT"); // BRK_PLACEMENT=0: every storage is a plain hipMalloc (A/B runs)
  if (!measurable || (off && off[0] == '0' && off[1] == 0)) {
    HIP_TRY(hipMalloc(&chosen, bytes));
  } else {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int ncand = (int)std::min<size_t>(kMaxCandidates, std::max<size_t>(2, kCandidateBudget / bytes));
    ncand = (int)std::min<size_t>((size_t)ncand, std::max<size_t>(1, free_b / 2 / bytes));
    size_t probe_bytes = bytes; // one probe grid for all pairs: what the smallest storage involved can hold
    for (const Partner &p : partners) probe_bytes = std::min(probe_bytes, p.p.bytes);
    ProbeGrid grid;
    if (int rc0 = make_probe_grid(probe_bytes, grid)) return rc0;
    std::vector<void *> cands, spacers;
    std::vector<double> score, first; // score: what is minimised; first: the sweep time against partner 0 (which kind is the candidate?)
    std::vector<double> shown;        // the sweep time reported for a candidate: its slowest OPPOSITE partner (its fastest SAME one when there is none)
    int best = -1;
    int rc = BRK_OK;
    size_t held = 0;
    const size_t held_cap = std::min(kHeldBudget, free_b / 20 * 9);
    // one more candidate: allocate, time the probe sweep from every partner into it; false when there is no room for it.
    // OPPOSITE partners count with their sweep time, SAME partners against it: a read / write pair inside one kind is the SLOW one, so the
    // candidate that sweeps slowest from a SAME partner is the one of that partner's kind.
    auto try_candidate = [&]() {
      void *c = nullptr;
      if (held + bytes > std::max(held_cap, bytes) || hipMalloc(&c, bytes) != hipSuccess) {
        (void)hipGetLastError(); // out of room for further candidates: settle for what was seen
        return false;
      }
      cands.push_back(c);
      held += bytes;
      double sc = 0.0, opp = 0.0, same = 1e30, ms0 = 0.0;
      for (size_t q = 0; q < partners.size(); ++q) {
        double ms = 0.0;
        rc = probe_locked(grid, partners[q].p.p, c, st, &ms);
        if (rc) return false;
        if (q == 0) ms0 = ms;
        if (partners[q].relation == BRK_PLACE_OPPOSITE) sc += ms, opp = std::max(opp, ms);
        else sc -= ms, same = std::min(same, ms);
      }
      score.push_back(sc);
      first.push_back(ms0);
      shown.push_back(opp > 0.0 ? opp : same);
      if (best < 0 || sc < score[best]) best = (int)score.size() - 1;
      return true;
    };
    // What the SLOW rate is on this device, for this probe grid: the same sweep between the two halves of ONE allocation (an allocation is
    // of one kind; should it straddle two, the reference reads fast and the search below merely runs longer).  Measured TWICE: the
    // difference of the two is the repeatability of a probe on this device, and the gap that counts as "both kinds seen" follows from it
    // (five times the repeatability, at least 1 %; the kinds lie 2.5-3 % apart, repeats within 0.3 %) instead of a constant.
    double slow_ref = 0.0, noise = 0.003;
    const size_t half = (bytes + 4095) / 4096 * 4096; // the second half starts on a brick boundary whatever `bytes` is
    if (2 * half <= held_cap) {
      void *both = nullptr;
      if (hipMalloc(&both, 2 * half) == hipSuccess) {
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
        // the first probe of a process also pays for cold clocks and the first touch of the pages: it is run and dropped (round 5: with it
        // in the pair the "repeatability" came out near 1 %, the gap at 5 %, and a 3.5 % contrast among the candidates went unseen --
        // 34 candidates and 60 GiB of spacers for a pair that candidate 16 already was)
        rc = probe_locked(grid, both, static_cast<char *>(both) + half, st, &r0);
        if (rc == BRK_OK) rc = probe_locked(grid, both, static_cast<char *>(both) + half, st, &r1);
        if (rc == BRK_OK) rc = probe_locked(grid, both, static_cast<char *>(both) + half, st, &r2);
        (void)hipFree(both);
        if (rc == BRK_OK && r1 > 0.0 && r2 > 0.0) slow_ref = std::max(r1, r2), noise = std::max(0.002, std::abs(r1 - r2) / slow_ref);
      } else {
        (void)hipGetLastError();
      }
    }
    const double gap = std::max(0.01, 5.0 * noise);
    // both rates have been seen (against partner 0) and a few candidates compared: done
    // ONE-SIDED when the reference is there: the other kind shows as a candidate FASTER than a pair inside one allocation.  (Round 5, six
    // driver-style runs: with the two-sided test `max - min >= gap` a candidate that merely probed 1.2 % SLOWER than the rest ended two
    // searches after 12 and 14 candidates of one kind -- 361.5 GStencil/s instead of 371 -- profiles/r05_m_*.)
    auto contrast = [&]() {
      if ((int)first.size() < std::min(kMinCandidates, ncand)) return false;
      const double lo = *std::min_element(first.begin(), first.end());
      if (slow_ref > 0.0) return lo <= slow_ref * (1.0 - gap);
      const double hi = *std::max_element(first.begin(), first.end());
      return hi - lo >= gap * hi;
    };
    const bool force_spacers = std::getenv("BRK_PLACEMENT_TEST_SPACERS") != nullptr; // tests: take the spacer path whatever the candidates say
    for (int t = 0; t < ncand && rc == BRK_OK; ++t) {
      if (!try_candidate()) break;
      if (contrast() && !force_spacers) break;
    }
    // every back-to-back candidate is of one kind: move on behind untouched spacer allocations -- but only with a reference to tell the
    // kinds apart by (without it "all alike" cannot be told from "all noise")
    if (rc == BRK_OK && best >= 0 && ((slow_ref > 0.0 && !contrast()) || force_spacers)) {
      for (size_t gib : kSpacerGiB) {
        const size_t sp = gib << 30;
        void *spacer = nullptr;
        if (held + sp + 2 * bytes > held_cap || hipMalloc(&spacer, sp) != hipSuccess) {
          (void)hipGetLastError();
          break;
        }
        spacers.push_back(spacer);
        held += sp;
        bool more = try_candidate() && try_candidate();
        if (!more || rc || (contrast() && !force_spacers)) break;
      }
    }
    for (void *sp : spacers) (void)hipFree(sp);
    if (rc || best < 0) {
      for (void *c : cands) (void)hipFree(c);
      if (rc) return rc;
      return brk_fail(BRK_EHIP, "brk_storage_alloc: hipMalloc of %zu bytes failed", bytes);
    }
    chosen = cands[best];
    for (size_t q = 0; q < cands.size(); ++q)
      if ((int)q != best) (void)hipFree(cands[q]);
    if (report) {
      report->candidates = (int)score.size();
      report->spacer_gib = 0;
      for (size_t q = 0; q < spacers.size(); ++q) report->spacer_gib += (int)kSpacerGiB[q];
      report->chosen = best;
      report->probe_bricks = (int)((grid.edge - 2) * (grid.edge - 2) * (grid.edge - 2));
      report->ms_chosen = shown[best];
      report->ms_worst = *std::max_element(shown.begin(), shown.end());
      report->ms_same_allocation = slow_ref;
    }
  }
  g_placed.push_back({chosen, bytes, device});
  *out_dev = chosen;
  return BRK_OK;
}

extern "C" int brk_storage_alloc(size_t bytes, void **out_dev, brk_placement *report) {
  return brk_storage_alloc_ex(bytes, nullptr, 0, nullptr, out_dev, report);
}

extern "C" int brk_storage_free(void *p_dev) {
  if (!p_dev) return BRK_OK;
  {
    std::lock_guard<std::mutex> lock(g_place_mutex);
    size_t q = 0;
    for (; q < g_placed.size(); ++q)
      if (g_placed[q].p == p_dev) break;
    if (q == g_placed.size()) return brk_fail(BRK_EINVAL, "brk_storage_free: %p did not come from brk_storage_alloc", p_dev);
    g_placed.erase(g_placed.begin() + q);
  }
  HIP_TRY(hipFree(p_dev));
  return BRK_OK;
}
