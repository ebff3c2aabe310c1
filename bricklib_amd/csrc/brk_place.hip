// brk_place.hip -- WHERE in HBM the brick storages of a process lie (brk_storage_alloc / brk_storage_free / brk_placement_probe).
//
// Measured on MI355X (profiles/r04_g_*): the 288 GB are made of regions of roughly 8 GiB, and a sweep whose input storage and
// output storage lie in regions of the same kind runs 2.7 % slower (7-point, 512^3: 0.371-0.372 ms against 0.361-0.362 ms; a
// brick-shaped copy: 0.400 against 0.386 ms) than the same sweep between storages in regions of different kinds -- the read
// stream and the write stream then meet in the same DRAM banks.  Which region a hipMalloc lands in is the driver's business and
// is not visible from the virtual address: storages allocated one after the other are usually of the same kind (8 consecutive
// 1.1-GiB allocations were), sometimes not, which is the "fast box / slow box" spread of the earlier rounds' numbers.
// The reference allocates a storage wherever the allocator puts it (include/brick.h:68-75, include/brick-gpu.h:80-99); here
// the allocator of brick storages MEASURES: up to ten candidate allocations are made, a streaming copy from the storages allocated
// before into each candidate is timed, the candidate with the fastest copy is kept and the others are freed (all candidates are
// held until the choice is made, so each one lies somewhere else).  Pairs differ by more than two levels (copy: 0.397 ... 0.454 ms
// per GiB between 32 storages of one process, reproducible to 1 %), so every candidate is measured rather than the first fast one
// taken.  About 50 ms per storage, at setup.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "brk_device.h"

namespace brklib {
namespace {

// read stream from a, write stream into b, 16 bytes per lane, results leave with non-temporal stores (as the sweeps' do)
__global__ __launch_bounds__(256) void k_place_copy(const uint4 *__restrict__ a, uint4 *__restrict__ b, size_t n16) {
  typedef unsigned v4 __attribute__((ext_vector_type(4)));
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const v4 v = reinterpret_cast<const v4 *>(a)[i];
    __builtin_nontemporal_store(v, reinterpret_cast<v4 *>(b) + i);
  }
}

struct Placed {
  void *p;
  size_t bytes;
  int device;
};
std::mutex g_place_mutex;
std::vector<Placed> g_placed; // live storages handed out by brk_storage_alloc, in allocation order

constexpr size_t kMinPlaced = (size_t)64 << 20;      // smaller storages are not worth a measurement
constexpr size_t kCandidateBudget = (size_t)16 << 30; // address space / memory held by candidates while one storage is placed
constexpr int kMaxCandidates = 10;                   // 8 consecutive allocations of one kind were seen

int probe_locked(const void *a, void *b, size_t bytes, hipStream_t st, double *ms_out) {
  const size_t n16 = bytes / 16;
  if (n16 == 0) return brk_fail(BRK_EINVAL, "brk_placement_probe: fewer than 16 bytes");
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  hipError_t r = hipEventCreate(&e1);
  if (r != hipSuccess) {
    (void)hipEventDestroy(e0);
    HIP_TRY(r);
  }
  const dim3 grid(4096), block(256);
  constexpr int kWarm = 2, kTimed = 6;
  for (int q = 0; q < kWarm; ++q) hipLaunchKernelGGL(k_place_copy, grid, block, 0, st, (const uint4 *)a, (uint4 *)b, n16);
  r = hipEventRecord(e0, st);
  for (int q = 0; q < kTimed && r == hipSuccess; ++q) hipLaunchKernelGGL(k_place_copy, grid, block, 0, st, (const uint4 *)a, (uint4 *)b, n16);
  if (r == hipSuccess) r = hipEventRecord(e1, st);
  if (r == hipSuccess) r = hipEventSynchronize(e1);
  if (r == hipSuccess) r = hipGetLastError();
  float ms = 0.f;
  if (r == hipSuccess) r = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HIP_TRY(r);
  *ms_out = (double)ms / kTimed;
  return BRK_OK;
}

} // namespace
} // namespace brklib

using namespace brklib;

extern "C" int brk_placement_probe(const void *a_dev, void *b_dev, size_t bytes, brk_stream s, double *ms_per_copy) {
  if (!a_dev || !b_dev || !ms_per_copy) return brk_fail(BRK_EINVAL, "brk_placement_probe: null argument");
  if (((uintptr_t)a_dev | (uintptr_t)b_dev) % 16) return brk_fail(BRK_EINVAL, "brk_placement_probe: storages must be 16-byte aligned");
  return probe_locked(a_dev, b_dev, bytes, static_cast<hipStream_t>(s), ms_per_copy);
}

extern "C" int brk_storage_alloc(size_t bytes, void **out_dev, brk_placement *report) {
  if (!out_dev) return brk_fail(BRK_EINVAL, "brk_storage_alloc: null result pointer");
  *out_dev = nullptr;
  if (report) std::memset(report, 0, sizeof *report);
  if (bytes == 0) return brk_fail(BRK_EINVAL, "brk_storage_alloc: zero bytes");
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(g_place_mutex);
  // the storages a sweep will pair this one with: the (at most two) most recent live ones of this device that are worth measuring
  std::vector<Placed> partners;
  for (size_t q = g_placed.size(); q-- > 0 && partners.size() < 2;)
    if (g_placed[q].device == device && g_placed[q].bytes >= kMinPlaced) partners.push_back(g_placed[q]);
  void *chosen = nullptr;
  if (bytes < kMinPlaced || partners.empty()) {
    HIP_TRY(hipMalloc(&chosen, bytes));
  } else {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    int ncand = (int)std::min<size_t>(kMaxCandidates, std::max<size_t>(2, kCandidateBudget / bytes));
    ncand = (int)std::min<size_t>((size_t)ncand, std::max<size_t>(1, free_b / 2 / bytes));
    std::vector<void *> cands;
    std::vector<double> score;
    int best = -1;
    double worst = 0.0;
    int rc = BRK_OK;
    for (int t = 0; t < ncand; ++t) {
      void *c = nullptr;
      if (hipMalloc(&c, bytes) != hipSuccess) {
        (void)hipGetLastError(); // out of room for further candidates: settle for what was seen
        break;
      }
      cands.push_back(c);
      double sc = 0.0; // milliseconds per GiB, worst partner
      for (const Placed &p : partners) {
        const size_t nb = std::min(bytes, p.bytes);
        double ms = 0.0;
        rc = probe_locked(p.p, c, nb, nullptr, &ms);
        if (rc) break;
        sc = std::max(sc, ms * (double)((size_t)1 << 30) / (double)nb);
      }
      if (rc) break;
      score.push_back(sc);
      if (best < 0 || sc < score[best]) best = (int)score.size() - 1;
      worst = std::max(worst, sc);
    }
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
      report->chosen = best;
      report->ms_per_gib_chosen = score[best];
      report->ms_per_gib_worst = worst;
    }
  }
  g_placed.push_back({chosen, bytes, device});
  *out_dev = chosen;
  return BRK_OK;
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
