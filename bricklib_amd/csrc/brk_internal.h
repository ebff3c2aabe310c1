// brk_internal.h -- shared between the translation units of libbrk.so (not installed).
#ifndef BRK_INTERNAL_H
#define BRK_INTERNAL_H
#include "brk.h"

// records the message for brk_last_error() and returns `code`
int brk_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// records, per calling thread, the name of the stencil kernel a launcher is about to enqueue (string literal): brk_last_kernel()
#ifdef __cplusplus
extern "C"
#endif
void brki_note_kernel(const char *name);

// brk_stencil_apply_expr / brk_expr_compile: known codes, registers in range and written before they are read, taps inside the 125-point
// cube; *radius = 1 or 2.  BRK_OK or a recorded failure.
#ifdef __cplusplus
extern "C"
#endif
int brki_expr_check(const brk_expr_op *ops, uint32_t nops, uint32_t max_ops, uint32_t result_reg, int *radius);
#ifdef __cplusplus
extern "C"
#endif
int brki_face_table_lock(const uint32_t *adj_dev, const uint32_t **faces); // brk_kernels.hip: compact face rows of a REGISTERED adjacency table, else
#ifdef __cplusplus
extern "C"
#endif
void brki_adj_unlock(void); // nullptr; BRK_OK leaves the tables' mutex taken until brki_adj_unlock (after the launch is enqueued)
#ifdef __cplusplus
extern "C" {
#endif
// brk_kernels.hip: the traversal order of a box as a device list; BRK_OK leaves the list cache's mutex taken until brki_box_list_unlock(stream)
int brki_box_list_lock(uint32_t base, const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], const uint32_t **ids_dev, uint32_t *n);
int brki_box_list_unlock(brk_stream s);
#ifdef __cplusplus
}
#endif
#ifdef __cplusplus
extern "C"
#endif
void brki_jit_finalize(void); // brk_finalize: unload the modules of brk_expr_compile
enum { BRK_EXPR_MAX_OPS = 120, BRK_EXPR_MAX_REGS = 16, BRK_EXPR_COMPILE_MAX_OPS = 512 }; // interpreter: the program is a kernel argument; compiled: text (straight-line code: a
// 125-tap linear sum of 501 operations already runs slower than the tap table, 3.4 vs 2.05 ms per 512^3 sweep, and 1 371 operations spill)
// // = kMaxExprOps, kMaxExprRegs of brick-device.h (static_assert in brk_kernels.hip)

#ifdef __cplusplus
#include <array>
#include <mutex>
#include <vector>

// direction sets = the reference's BitSet (include/bitset.h:20-30): axis p in 1..3, "+p" is bit p, "-p" is bit 31+p
namespace brk_dir {
inline uint64_t dir_bit(int p) { return 1ull << (p > 0 ? p : 31 - p); }
inline bool has(uint64_t s, int p) { return (s & dir_bit(p)) != 0; }
inline uint64_t negate(uint64_t s) { return ((s & 0xFFFFFFFFull) << 31) | (s >> 31); }
inline bool contains(uint64_t big, uint64_t small) { return (big & small) == small; }
void neighbors3(std::array<uint64_t, 27> &out); // src/brick-mpi.cpp:9-23 order
} // namespace brk_dir

// plan + view that brk_exchange() built for one (storage, step, rank map, communicator)
struct brk_xcache_entry {
  void *storage, *comm;
  size_t step;
  int ranks[27];
  brk_xplan *plan;
  brk_exchange_view *view;
};

struct brk_decomp {
  uint32_t dims[3], tdims[3], depth[3], stride[3];
  uint32_t sep[3];
  uint32_t nbricks;
  std::vector<uint32_t> grid, adj;
  std::vector<uint64_t> skinlist;
  std::vector<int64_t> skin_size;
  std::vector<brk_region> ghost, skin;
  std::mutex xmu;
  std::vector<brk_xcache_entry> xcache;
};
#endif

#endif
