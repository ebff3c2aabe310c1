/*
 * brk.h -- C ABI of the MI355X-native bricked stencil engine (libbrk.so).
 *
 * This is the drop-in boundary for ONE hot path of benSepanski/bricklib: the per-brick stencil
 * apply over 8x8x8 fp64 bricks plus the ghost-zone exchange bookkeeping around it.  The reference
 * has no ABI for this path -- it is a lexical macro, `brick("7pt.py", VSVEC, (8,8,8), (VFOLD), b)`
 * (include/vecscatter.h:62-90), expanded by a code generator inside user loops/kernels
 * (stencils/3axis.hip.cpp:32-41, weak/main.cu:35-43, strong/main.cu:85-111).  Each entry point
 * below names the reference call site it replaces; INTEGRATION.md shows the binding a bricklib
 * maintainer adds.  All `file:line` citations are relative to the reference tree.
 *
 * Conventions
 *  - plain pointers and sizes only; `brk_stream` is a hipStream_t passed as void* (NULL = null stream)
 *  - *_dev pointers are device (HBM) addresses, *_host pointers are host addresses
 *  - the caller owns every buffer; the library never frees caller memory
 *  - every call is asynchronous on the given stream unless stated; thread-safe for distinct streams
 *  - return 0 on success, a negative BRK_E* code otherwise; brk_last_error() gives the message of the
 *    last failure on the calling thread.  Nothing throws across this boundary.
 *  - brick layout: 512 doubles, row-major k*64+j*8+i (Brick<Dim<8,8,8>,Dim<8,8>>, stencils/hipvfold.h:5-6);
 *    brick b of a field starts at dat[b*step_elems]; adjacency index (dk+1)*9+(dj+1)*3+(di+1),
 *    13 = self (include/brick.h:235-246)
 */
#ifndef BRK_H
#define BRK_H

#ifndef __HIPCC_RTC__ /* (run-time compilation of brick-device.h, brk_expr_compile: no standard headers there) */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef void *brk_stream;

enum {
  BRK_OK = 0,
  BRK_EINVAL = -1,   /* bad argument (null pointer, unknown stencil id, misaligned/size mismatch) */
  BRK_EHIP = -2,     /* a HIP runtime call failed; brk_last_error() carries hipGetErrorString */
  BRK_ENOMEM = -3,
  BRK_ENODEV = -4,   /* no gfx950 device / code object could not be loaded */
  BRK_ECOMM = -5     /* an RCCL call failed, or librccl could not be loaded; brk_last_error() carries ncclGetErrorString */
};

/* Stencils on the path.  Coefficient vectors (host memory, doubles):
 *  BRK_7PT_COEFF  coeff[0..6]   stencils/7pt.py:10-27, stencils/3axis.cpp:33-36,60-63
 *  BRK_27PT_COEFF coeff[0..26]  stencils/3axis.cpp:184-202,214-233, including the reference quirk that
 *                               coeff[16] multiplies in[k+1][j+1][i] (3axis.cpp:195,:225)
 *  BRK_MPI7   {ALPHA,BETA}      stencils/mpi7pt.py, stencils/fake.h:11-12,346-353   (NULL -> .4,.1)
 *  BRK_MPI13  {B0,B1,B2}        stencils/mpi13pt.py, fake.h:20-22,71-84             (NULL -> defaults)
 *  BRK_MPI25  {A0..A4}          stencils/mpi25pt.py, fake.h:14-18,41-70             (NULL -> defaults)
 *  BRK_MPI125 {C0..C9}          stencils/mpi125pt.py, fake.h:24-33,90-343           (NULL -> defaults)
 *  BRK_COND7  coeff[0..6]       stencils/cond.py, stencils/3axis.cpp:94-163 (d3cond): |sum c_t * max(in_t, 0)|
 *  BRK_MPI7_X2, BRK_7PT_X2     TWO consecutive sweeps of BRK_MPI7 / BRK_7PT_COEFF in one pass over memory (same coefficient
 *                               vectors): out = S(S(in)), evaluated as the 25-point radius-2 stencil S*S whose weights are
 *                               the products of the 7-point weights.  Not in the reference; used by the weak / strong
 *                               drivers to halve the HBM traffic of a communication-avoiding cycle (ST_ITER sweeps between
 *                               exchanges, weak/main.cpp:186-207).  Agrees with two separate sweeps to rounding (~1e-16). */
enum { BRK_7PT_COEFF = 0, BRK_27PT_COEFF = 1, BRK_MPI7 = 2, BRK_MPI13 = 3, BRK_MPI25 = 4, BRK_MPI125 = 5, BRK_COND7 = 6,
       BRK_MPI7_X2 = 7, BRK_7PT_X2 = 8, BRK_NSTENCIL = 9 };

const char *brk_last_error(void);
/* version of this ABI: 1000 + additions (1001: brk_stencil_apply_taps, 1002: the fused double-sweep stencil ids,
 * 1003: the ghost-zone exchange: brk_xplan_*, brk_exchange_*, brk_comm_*; 1004: brk_adjacency_register / _release;
 * 1005: brk_stencil_apply_{range,list,box}_ex with BRK_SWEEP_* flags; 1006: brk_comm_library, the development knob
 * brk_set_variant is no longer part of the ABI; 1007: brk_storage_alloc / brk_storage_free / brk_placement_probe;
 * 1008: brk_finalize, brk_last_kernel; 1009: brk_storage_alloc_ex; 1010: brk_stencil_apply_expr;
 * 1011: brk_expr_compile / brk_expr_apply / brk_expr_release / brk_expr_source / brk_expr_kernel_source;
 * 1012: brk_expr_kernel_from_disk (BRK_JIT_CACHE_DIR);
 * 1013: brk_expr_apply_box) */
int brk_abi_version(void);
/* Orderly end of the library's use of the GPU before a process leaves main(): waits for everything enqueued on the current device
 * and releases, on the CALLING thread, what the library itself holds there (box id lists and their events, derived tables of
 * adjacency tables still registered).  Caller memory is not touched and the library stays usable.  The reference's drivers end
 * with hipFree + return (stencils/3axis.hip.cpp:193-200, weak/main.cu:330-340); a driver over this library adds this one call
 * in front of its return so that no device object is left to static destructors, which run concurrently with the HIP runtime's
 * own exit handlers. */
int brk_finalize(void);
/* Name of the kernel (template arguments included, e.g. "k_star_km<kStarMpi25,2,3>") that the last brk_stencil_apply_* call of the
 * calling thread enqueued; "" before the first.  Reporting only (bench.py's roofline.kernel); the string is static. */
const char *brk_last_kernel(void);
/* radius / number of coefficients / communication-avoiding sweeps per exchange (ST_ITER, fake.h:39-344) */
int brk_stencil_radius(int stencil_id);
int brk_stencil_ncoeff(int stencil_id);
int brk_stencil_iters(int stencil_id);

/* ---- the hot path -------------------------------------------------------------------------------------------
 * out[b] = stencil(in[b] and its neighbours through adj) for every brick id b in [brick_begin,brick_end).
 * Replaces the body of `brick(...)` launched over a grid of bricks: d3pt7_brick_trans
 * (stencils/3axis.hip.cpp:32-41,165-171), brick_kernel (weak/main.cu:35-43,277-282).
 * in/out may be two fields of one interleaved storage (single/: out_dev = in_dev+512, step_elems = 1024,
 * stencils/3axis.cpp:25-28) or two storages (step_elems = 512, weak/main.cpp:76-77).  Both fields must be 16-byte aligned
 * and step_elems even (BrickStorage allocations are 2048-byte aligned, include/brick.h:68-75 of the reference): the kernels
 * move halo rows in 16-byte lanes; anything else is refused with BRK_EINVAL. */
int brk_stencil_apply_range(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                            size_t step_elems, uint32_t brick_begin, uint32_t brick_end, const double *coeff_host,
                            brk_stream s);
/* Same over an explicit list of brick ids, e.g. the reference's `grid` array or any slice of it
 * (stencils/3axis.hip.cpp:36-39 `b = grid[tk][tj][ti]`, strong/main.cu:85-99).  Entries equal to BRK_LIST_SKIP are
 * ignored; hosts use them to separate rows of a traversal order: the kernels share i-halos between consecutive entries that
 * are i-neighbours (4 or 8 entries per workgroup) and handle two consecutive non-neighbours correctly but more slowly than
 * two runs with a skipped entry between them (bricklib_amd/order.py). */
#define BRK_LIST_SKIP 0xFFFFFFFFu
int brk_stencil_apply_list(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                           size_t step_elems, const uint32_t *brick_ids_dev, uint32_t n, const double *coeff_host,
                           brk_stream s);
/* The same for a list the caller has put in the K-PAIRED order: groups of 8 entries = a run of 4 i-neighbours followed by the run of
 * their +k neighbours (BRK_LIST_SKIP where a run is shorter or a level has no upper partner); bricklib_amd/order.py:tiled(...,
 * kpair=True) builds it.  The radius-4 star (BRK_MPI25) then runs two bricks per wave and loads the planes between the two bricks
 * of a pair once (6.8 % faster at 512^3); every other stencil, and any list that is not paired, is computed exactly as by
 * brk_stencil_apply_list.  Results do not depend on the order. */
int brk_stencil_apply_list_kpaired(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                                   size_t step_elems, const uint32_t *brick_ids_dev, uint32_t n, const double *coeff_host,
                                   brk_stream s);
/* Same over a structured box of a lexicographically numbered brick grid: id = base + (k*tot[1] + j)*tot[0] + i for
 * (i,j,k) in [lo, lo+cnt) (contiguous axis first).  This is exactly the reference launch `dim3 block(NB,NB,NB)` with
 * `b = grid[GB+z][GB+y][GB+x]` over an init_grid numbering (stencils/3axis.hip.cpp:36-39,168-170: base 0, tot = STRIDEB^3,
 * lo = GB, cnt = NB) and the inner region [1,sep0) of a BrickDecomp (base 1, tot = cnt = inner extent, lo = 0).
 * The 7-point kernels derive the ids arithmetically in their preferred traversal order (no list is read); the other
 * stencils use an id list in that order which the library builds once per box geometry and keeps on the device (a cache of
 * 16 geometries per process; the first call for a geometry is synchronous).  BRK_MPI125 over a REGISTERED adjacency table
 * (brk_adjacency_register) whose box has rows of a multiple of 8 bricks and only regular bricks (all 26 neighbours are the ids the
 * numbering predicts -- checked once per box and registration, synchronously) runs the four-waves-per-SIMD kernel k_cube125_q
 * (6 % faster at 512^3; same results to rounding); BRK_CUBE125_Q=0 in the environment keeps the general kernel (A/B runs). */
int brk_stencil_apply_box(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                          size_t step_elems, uint32_t base, const uint32_t tot_bricks[3], const uint32_t lo[3],
                          const uint32_t cnt[3], const double *coeff_host, brk_stream s);

/* The three forms above with sweep flags:
 *   BRK_SWEEP_KPAIRED  (list form) the list is in the k-paired order, see brk_stencil_apply_list_kpaired.
 *   BRK_SWEEP_REVERSE  walk the same bricks from the other end: every XCD takes its share of the range / list / box back to front
 *                      (workgroups keep their 4 or 8 consecutive entries).  Results do not depend on it.  An iterative solver calls
 *                      the sweep again and again over the same bricks (the reference's timing loops, stencils/3axis.hip.cpp:168-190,
 *                      weak/main.cu:246-287): when sweep t+1 starts where sweep t stopped, the part of the field touched last
 *                      -- up to the 256 MB of the Infinity Cache -- is read from that cache instead of HBM.  Alternate the flag
 *                      from one sweep to the next: 7-point 512^3 sweep 0.376 -> 0.363 ms, 13-point 0.395 -> 0.379, fused double
 *                      7-point pass 0.443 -> 0.425 (profiles/r03_l_zigzag_sweeps.txt).
 * Unknown flag bits, or BRK_SWEEP_KPAIRED on a range or box: BRK_EINVAL. */
#define BRK_SWEEP_KPAIRED 1u
#define BRK_SWEEP_REVERSE 2u
int brk_stencil_apply_range_ex(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                               size_t step_elems, uint32_t brick_begin, uint32_t brick_end, const double *coeff_host,
                               uint32_t flags, brk_stream s);
int brk_stencil_apply_list_ex(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                              size_t step_elems, const uint32_t *brick_ids_dev, uint32_t n, const double *coeff_host,
                              uint32_t flags, brk_stream s);
int brk_stencil_apply_box_ex(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                             size_t step_elems, uint32_t base, const uint32_t tot_bricks[3], const uint32_t lo[3],
                             const uint32_t cnt[3], const double *coeff_host, uint32_t flags, brk_stream s);

/* OPTIONAL promise about an adjacency table: from brk_adjacency_register(adj_dev, nbricks) until brk_adjacency_release(adj_dev)
 * the nbricks x 27 entries at adj_dev (current device) do not change.  The library may then keep derived tables next to it; so
 * far: per box numbering, one bit per brick saying that its six face neighbours are the ones the lexicographic numbering
 * predicts, which lets the 7-point box kernels skip those bricks' adjacency rows (108-byte rows straddling 64-byte lines: 60 MB
 * of reads and 3.4 % of a 512^3 sweep).  Results are identical with and without registration.  The reference's BrickInfo is
 * written once by init_grid / BrickDecomp::getBrickInfo and never again (include/bricksetup.h:30-90, include/brick-mpi.h:275-284),
 * so its drivers can register every table; an unregistered table is read row by row as before.  Release BEFORE freeing or
 * rewriting the table.  Errors: BRK_EINVAL for a null / duplicate / unknown table. */
int brk_adjacency_register(const uint32_t *adj_dev, uint32_t nbricks);
int brk_adjacency_release(const uint32_t *adj_dev);

/* A linear stencil given as data -- the capability of the reference's code generator for scripts that are not one of the
 * built-in ones (codegen/vecscatter evaluating stencils/<name>.py; bricklib_amd/spec.py reads such scripts into this form):
 *   out[p] = sum_t weights[t] * in[p + (offsets[3t], offsets[3t+1], offsets[3t+2])]   (i, j, k offsets; taps added in list
 *   order).  Every offset in [-4,4].  Three shapes, chosen by the library: a table inside the 125-point cube (offsets in [-2,2],
 *   ntaps <= 125: the table travels as a kernel argument); a STAR-shaped table -- every tap moves along one axis only -- up to
 *   radius 4 with ntaps <= 64 (a user-written radius-3 or radius-4 finite-difference script; star staging); any other table of
 *   radius 3 or 4, up to 9^3 = 729 taps (round 5: (8+2R)^3 tiles, the table kept in device memory per content -- the first call
 *   with a new table uploads it synchronously, brk_finalize frees them).  The reference's generator takes any script
 *   (codegen/st/codegen/base.py:678; stencils/mpi25pt.py is radius 4).  Bricks: brick_ids_dev[0..n) when non-NULL (BRK_LIST_SKIP allowed), else
 *   ids first..first+n-1.  General path: one LDS read + FMA per tap and point, not tuned like the built-in stencils. */
int brk_stencil_apply_taps(const uint32_t *adj_dev, const double *in_dev, double *out_dev, size_t step_elems,
                           const uint32_t *brick_ids_dev, uint32_t first, uint32_t n, uint32_t ntaps,
                           const int32_t *offsets_host, const double *weights_host, brk_stream s);

/* NON-LINEAR stencils given as data (ABI 1010): the reference's generator takes any expression over grid references, constants,
 * functions and If (codegen/st/codegen/base.py:678; stencils/cond.py: max, If).  A script arrives as a register program
 * (bricklib_amd/spec.py builds it from the script; at most 120 ops, 16 registers, tap offsets in [-4,4]): out[p] =
 * value of register `result_reg` after the ops have run in order for cell p.  Registers must be written before they are read.
 * The general path -- one LDS read per tap, a few instructions per op; the built-in stencils keep their tuned kernels. */
enum { BRK_EX_TAP = 0, BRK_EX_CONST, BRK_EX_ADD, BRK_EX_SUB, BRK_EX_MUL, BRK_EX_DIV, BRK_EX_NEG, BRK_EX_MAX, BRK_EX_MIN, BRK_EX_ABS,
       BRK_EX_SQRT, BRK_EX_GT, BRK_EX_GE, BRK_EX_LT, BRK_EX_LE, BRK_EX_EQ, BRK_EX_NE, BRK_EX_SEL /* dst = a != 0 ? b : c */ };
typedef struct {
  uint32_t code, dst, a, b, c; /* BRK_EX_*; destination and source registers (0..15) */
  int32_t offset[3];           /* BRK_EX_TAP: (di, dj, dk), contiguous axis first */
  double imm;                  /* BRK_EX_CONST */
} brk_expr_op;
int brk_stencil_apply_expr(const uint32_t *adj_dev, const double *in_dev, double *out_dev, size_t step_elems,
                           const uint32_t *brick_ids_dev /* NULL: id range */, uint32_t first, uint32_t n,
                           const brk_expr_op *ops_host, uint32_t nops, uint32_t result_reg, brk_stream s);

/* The same program COMPILED at run time (ABI 1011).  The reference turns a script into C++ before the compiler sees the driver
 * (codegen/vecscatter); a host that holds the script as data gets that here: brk_expr_compile writes the operations out as
 * straight-line HIP over the staged tile (the text tools/brkscatter emits for a brick(...) line; brk_expr_source returns it without
 * compiling) and compiles it with hiprtc for the current device -- about a second, once per distinct program and device (a second
 * compile of the same program returns the same kernel; up to 512 operations -- linear sums of many taps belong to
 * brk_stencil_apply_taps: 125 taps as straight-line code run slower than the tap table).  brk_expr_apply then runs at the speed of the generic tap path instead of the
 * interpreter's.  Same operations, one fp64 rounding each (-ffp-contract=off): the same numbers as brk_stencil_apply_expr.
 * BRK_ENODEV: libhiprtc could not be loaded, or there is no device.  brk_finalize unloads the modules; a kernel used after that is
 * refused until its program is compiled again.  brk_expr_release when done (NULL is a no-op). */
typedef struct brk_expr_kernel brk_expr_kernel;
int brk_expr_compile(const brk_expr_op *ops_host, uint32_t nops, uint32_t result_reg, brk_expr_kernel **out);
int brk_expr_apply(const brk_expr_kernel *k, const uint32_t *adj_dev, const double *in_dev, double *out_dev, size_t step_elems,
                   const uint32_t *brick_ids_dev /* or NULL */, uint32_t first, uint32_t n, brk_stream s);
/* ... over the interior box of a lexicographic numbering (ABI 1013; geometry as for brk_stencil_apply_box: ids base + (k * tot[1] + j) *
 * tot[0] + i for (i, j, k) in [lo, lo + cnt)), walked in the built-in stencils' order -- no id list to build */
int brk_expr_apply_box(const brk_expr_kernel *k, const uint32_t *adj_dev, const double *in_dev, double *out_dev, size_t step_elems,
                       uint32_t base, const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], brk_stream s);
int brk_expr_release(brk_expr_kernel *k);
/* the generated translation unit: of a program (buf may be NULL; *need = bytes incl. the terminator) / of a compiled kernel */
int brk_expr_source(const brk_expr_op *ops_host, uint32_t nops, uint32_t result_reg, char *buf, size_t cap, size_t *need);
const char *brk_expr_kernel_source(const brk_expr_kernel *k);
/* BRK_JIT_CACHE_DIR=<directory> (ABI 1012): code objects are kept there across processes, named after everything they depend on
 * (generated text, the embedded device headers, target); brk_expr_compile then loads instead of compiling.  1 = this kernel was. */
int brk_expr_kernel_from_disk(const brk_expr_kernel *k);
/* Development builds of the library (make -C bricklib_amd/csrc DEV=1 or VARIANTS=1) additionally export the experiment knobs
 * used by tools/kbench (kernel variant selection, LDS padding, diagnostic modes), looked up with dlsym.  They are not declared
 * here and the default libbrk.so does not contain them: not part of this ABI. */

/* ---- region copies: pack / unpack / same-GPU halo links --------------------------------------------------------
 * Replaces RegionLink + cudaCopy (strong/main.cu:60-83, launched at :440-443,:459-462,:479-482) and the
 * self-neighbour MPI messages of BrickDecomp::exchange (include/brick-mpi.h:476-484) when both ends live on
 * this GPU.  Addresses and sizes must be multiples of 16 bytes (bricks are 4096 B). */
typedef struct {
  const void *from;
  void *to;
  uint64_t bytes;
} brk_link;
int brk_link_copy(const brk_link *links_dev, uint32_t n, uint64_t max_bytes, brk_stream s);

/* ---- WHERE a brick storage lies in HBM -------------------------------------------------------------------------------
 * The reference allocates a storage wherever the allocator puts it (include/brick.h:68-75, include/brick-gpu.h:80-99).  On MI355X
 * that leaves 2.7 % of a bandwidth-bound sweep to chance: the device allocations of a process fall into two kinds, and a sweep
 * whose input and output storages are of the SAME kind runs slower (7-point, 512^3: 0.371 ms against 0.361 ms) than between
 * storages of different kinds, while two READ streams prefer one kind (profiles/r04_g_*).  brk_storage_alloc[_ex] is hipMalloc +
 * a measurement:
 *   - partners: the live storages (from brk_storage_alloc) a sweep will pair the new one with, each with the wanted relation:
 *     BRK_PLACE_OPPOSITE for a read / write pair (input and output of a sweep), BRK_PLACE_SAME for two fields a sweep READS
 *     together.  brk_storage_alloc (no table) takes ONE partner, the most recent live storage of the device, as OPPOSITE: a chain
 *     of allocations then alternates A B A B, what ping-pong pairs and three rotating storages need.  A three-field sweep
 *     out = f(a, b): allocate a; b with {a, SAME}; out with {a, OPPOSITE}, {b, OPPOSITE}.
 *   - 6 to 24 candidate allocations (at most 28 GiB of them) are made; the production 7-point sweep (brk_placement_probe) is timed
 *     from every partner into every candidate on stream s; the candidate with the smallest (sum over OPPOSITE partners - sum over
 *     SAME partners) of sweep times is kept, the others are freed before the call returns.  What counts as "both kinds seen" is
 *     derived per call: the same sweep between the two halves of ONE allocation is the slow rate of the device, its repeat gives
 *     the repeatability, the gap must be five times that (at least 1 %).  When all back-to-back candidates are of one kind, up
 *     to 60 GiB of untouched spacer allocations move further candidates on (never more than a third of the free memory is held).
 *   - COST: synchronous, 50-150 ms and tens of GiB of TRANSIENT allocations per storage; meant for set-up, not for a time loop.
 *     Storages (or partners) below 512 MiB are plain hipMallocs (candidates = 0): a probe over less runs out of the 256-MiB
 *     Infinity Cache and shows neither kind.  BRK_PLACEMENT=0 in the environment makes every storage a plain hipMalloc (A/B runs;
 *     several ranks sharing one GPU should set it: the transient allocations of one rank can starve another).
 * Memory from brk_storage_alloc[_ex] is released with brk_storage_free.
 * brk_placement_probe is the measurement itself: milliseconds per BRK_MPI7 sweep (brk_stencil_apply_box) from a into b over
 * the largest cubic brick grid (at most 82^3) that `bytes` hold, on stream s; b is OVERWRITTEN.                         */
typedef struct {
  int candidates;   /* allocations measured (0: the storage was too small or had no partner, plain hipMalloc) */
  int chosen;       /* index of the one kept */
  int probe_bricks; /* bricks swept per probe launch */
  int spacer_gib;   /* GiB of spacer allocations it took to reach candidates of the other kind (0: back-to-back candidates sufficed) */
  double ms_chosen; /* probe sweep from the slowest OPPOSITE partner (without one: the fastest SAME partner) into the chosen candidate */
  double ms_worst;  /* the same figure of the candidate where it is largest */
  double ms_same_allocation; /* the same sweep between the two halves of one allocation: the slow rate of this device (0: not measured) */
} brk_placement;
enum { BRK_PLACE_OPPOSITE = 0, BRK_PLACE_SAME = 1 };
typedef struct {
  const void *storage_dev; /* a live storage from brk_storage_alloc[_ex] on the current device */
  int relation;            /* BRK_PLACE_OPPOSITE | BRK_PLACE_SAME */
} brk_place_partner;
int brk_storage_alloc(size_t bytes, void **out_dev, brk_placement *report /* may be NULL */);
int brk_storage_alloc_ex(size_t bytes, const brk_place_partner *partners, uint32_t npartners, brk_stream s, void **out_dev,
                         brk_placement *report /* may be NULL */);
int brk_storage_free(void *p_dev);
int brk_placement_probe(const void *a_dev, void *b_dev, size_t bytes, brk_stream s, double *ms_per_sweep);

/* ---- array <-> brick (device) ------------------------------------------------------------------------------------
 * copyToBrick / copyFromBrick (include/bricksetup.h:172-221, iteration space :92-159): dimlist = iterated
 * extent in cells (contiguous axis first), padding = array-only halo, ghost = halo in both (skipped). */
int brk_array_to_brick(const long dimlist[3], const long padding[3], const long ghost[3], const double *arr_dev,
                       const uint32_t *grid_dev, double *bdat_dev, size_t step_elems, brk_stream s);
int brk_brick_to_array(const long dimlist[3], const long padding[3], const long ghost[3], double *arr_dev,
                       const uint32_t *grid_dev, const double *bdat_dev, size_t step_elems, brk_stream s);
/* compareBrick (include/brickcompare.h:30-57) on device.  result_dev[0] = number of mismatching elements
 * under the reference rule |a-b|<tol || |a-b|<(|a|+|b|)*tol, result_dev[1] = bit pattern of max |a-b|
 * (as uint64).  result_dev is zeroed by the call. */
int brk_compare_brick(const long dimlist[3], const long padding[3], const long ghost[3], const double *arr_dev,
                      const uint32_t *grid_dev, const double *bdat_dev, size_t step_elems, double tol,
                      uint64_t *result_dev, brk_stream s);

/* ---- host-side layout builders (synchronous, host memory) --------------------------------------------------------
 * init_grid<3> (include/bricksetup.h:30-90): grid[dims0*dims1*dims2], adj[size][27]. */
int brk_init_grid(const long dims_bricks[3], uint32_t *grid_host, uint32_t *adj_host);

/* BrickDecomp<3,8,8,8>(dims, ghost).initialize(skin3d_good) (include/brick-mpi.h:178-460,
 * src/brick-mpi.cpp:9-52): region-ordered brick numbering, ghost/skin run tables, adjacency. */
typedef struct brk_decomp brk_decomp;
typedef struct {
  uint64_t neighbor;  /* BitSet.set of the neighbour direction (include/bitset.h:20-30) */
  uint32_t skin_st, skin_ed, pos, len, first_pad, last_pad; /* g_region, include/brick-mpi.h:185-193 */
} brk_region;
int brk_decomp_build(const uint32_t dims_cells[3], uint32_t ghost_cells, brk_decomp **out);
/* The same for any order of the 26 surface sections (the argument of BrickDecomp::initialize, include/brick-mpi.h:322): `order` lists
 * BitSet values (brk_allneighbors3 encoding), each section exactly once, empty sets (0) allowed anywhere as run separators.
 * brk_skin3d_order fills the three orders the reference ships (src/brick-mpi.cpp:25-136; out needs BRK_MAX_SKIN entries). */
#define BRK_SKIN3D_GOOD 0
#define BRK_SKIN3D_NORMAL 1
#define BRK_SKIN3D_BAD 2
#define BRK_MAX_SKIN 64
int brk_decomp_build_order(const uint32_t dims_cells[3], uint32_t ghost_cells, const uint64_t *order, uint32_t norder, brk_decomp **out);
int brk_skin3d_order(int which, uint64_t *out, uint32_t *n);
void brk_decomp_free(brk_decomp *d);
uint32_t brk_decomp_nbricks(const brk_decomp *d);           /* incl. dummy brick 0 */
uint32_t brk_decomp_grid_size(const brk_decomp *d);
void brk_decomp_tdims(const brk_decomp *d, uint32_t out[3]); /* bricks per axis incl. ghost */
void brk_decomp_sep_pos(const brk_decomp *d, uint32_t out[3]);
const uint32_t *brk_decomp_grid(const brk_decomp *d);       /* [t2][t1][t0] logical -> brick id (operator[]) */
const uint32_t *brk_decomp_adj(const brk_decomp *d);        /* [nbricks][27]; row 0 is all zero */
uint32_t brk_decomp_nruns(const brk_decomp *d);
const brk_region *brk_decomp_ghost(const brk_decomp *d);    /* [nruns] */
const brk_region *brk_decomp_skin(const brk_decomp *d);     /* [nruns] */
uint32_t brk_decomp_nskin(const brk_decomp *d);
const uint64_t *brk_decomp_skinlist(const brk_decomp *d);   /* [nskin] BitSet.set */
const int64_t *brk_decomp_skin_size(const brk_decomp *d);   /* [nskin] bricks per surface section */

/* allneighbors(0,1,3) order (src/brick-mpi.cpp:9-23); out[13] == 0 (self) */
void brk_allneighbors3(uint64_t out[27]);
/* populate(): neighbour -> rank on a periodic cart communicator, row-major rank order as MPI_Cart_rank
 * (include/brick-mpi.h:730-753); out indexed like brk_allneighbors3 */
int brk_populate_rank_map(const int cart_dims[3], const int coo[3], int out[27]);
/* MPI_Dims_create(size,3) as used by weak/args.cpp:105 for sizes whose prime factors are 2,3,5,7 */
int brk_dims_create(int size, int out[3]);

/* ---- ghost-zone exchange: GPU-resident pack / unpack feeding RCCL send/recv ------------------------------------------
 * Replaces BrickDecomp::exchange (include/brick-mpi.h:466-495), ExchangeView::exchange (:82-123),
 * MultiStageExchangeView::exchange (:131-165, built at :587-681) and the strong driver's pack -> MPI -> unpack
 * (strong/main.cu:367-482), with the MPI messages replaced by ncclSend/ncclRecv in one group on the caller's stream.
 *
 * A PLAN is host-side bookkeeping only (offsets, sizes, peers): which byte ranges are copied where before the transport
 * ("begin": same-GPU links and packing), which messages travel, and which copies follow ("end": unpacking), per stage
 * (one stage, or three for the shift exchange).  A VIEW binds a plan to one storage on the current device (the reference's
 * ExchangeView): it owns the per-peer send/recv buffers and the device-resident brk_link tables.
 * Three "spaces" name where a byte range lives: the brick storage, or the send / receive buffer kept for one peer. */
enum { BRK_SP_STORAGE = 0, BRK_SP_SEND = 1, BRK_SP_RECV = 2 };
typedef struct {
  uint32_t stage;
  uint32_t phase;              /* 0 = begin (before the transport), 1 = end (after it) */
  uint32_t src_space, dst_space;
  int32_t src_peer, dst_peer;  /* rank whose send/recv buffer is meant; -1 for BRK_SP_STORAGE */
  uint64_t src_off, dst_off, bytes; /* byte offsets inside the space; multiples of 16 */
} brk_xcopy;
typedef struct {
  uint32_t stage;
  uint32_t is_send;            /* 1 = ncclSend to `peer`, 0 = ncclRecv from `peer` */
  int32_t peer;
  uint32_t space;              /* BRK_SP_STORAGE (zero-copy run, the reference's form) or BRK_SP_SEND / BRK_SP_RECV (packed) */
  uint64_t off, bytes;
} brk_xmsg;
typedef struct brk_xplan brk_xplan;
typedef struct brk_exchange_view brk_exchange_view;

/* mode: how the ghost runs of a BrickDecomp that leave this rank travel.
 *  BRK_XCHG_PACKED  everything bound for one peer is packed into one buffer: ONE send + ONE recv per peer (<= 7 peers on a
 *                   2x2x2 grid), unpacked by one launch -- few, large xGMI transfers (default of the drivers)
 *  BRK_XCHG_DIRECT  one message per run straight from / into the storage, <= 42 per exchange (brick-mpi.h:476-484)
 *  BRK_XCHG_SHIFT   three stages, one per axis, each trading two boundary layers that span the ghosts received in the earlier
 *                   stages: <= 2 messages per stage (MultiStageExchangeView, brick-mpi.h:587-681)
 * flags: BRK_XCHG_SELF_VIA_COMM sends runs whose peer is this rank through the communicator too (ncclSend to self) instead of
 *        as device links -- for exercising the transport on a single GPU. */
enum { BRK_XCHG_PACKED = 0, BRK_XCHG_DIRECT = 1, BRK_XCHG_SHIFT = 2 };
enum { BRK_XCHG_SELF_VIA_COMM = 1u };
/* rank_of_neighbor[27] is indexed like brk_allneighbors3 (what brk_populate_rank_map writes; the reference's rank_map). */
int brk_xplan_build_decomp(const brk_decomp *d, const int rank_of_neighbor[27], int my_rank, int mode, unsigned flags,
                           size_t step_elems, brk_xplan **out);
/* a plan from explicit tables (the strong driver builds its own: several subdomains per rank, strong/main.cu:188-395);
 * copies and messages are taken in the given order, which both ends must agree on */
int brk_xplan_build_tables(int my_rank, uint32_t ncopies, const brk_xcopy *copies, uint32_t nmsgs, const brk_xmsg *msgs,
                           brk_xplan **out);
void brk_xplan_free(brk_xplan *p);
int brk_xplan_my_rank(const brk_xplan *p);
uint32_t brk_xplan_nstages(const brk_xplan *p);
uint32_t brk_xplan_ncopies(const brk_xplan *p);
const brk_xcopy *brk_xplan_copies(const brk_xplan *p);
uint32_t brk_xplan_nmsgs(const brk_xplan *p);
const brk_xmsg *brk_xplan_msgs(const brk_xplan *p);
/* bytes per exchange that leave the rank (sum of the sends) / that are copied inside it by the begin+end launches */
uint64_t brk_xplan_bytes_remote(const brk_xplan *p);
uint64_t brk_xplan_bytes_local(const brk_xplan *p);
/* size of the send (which = BRK_SP_SEND) or receive buffer kept for `peer` in `stage`; 0 if there is none */
uint64_t brk_xplan_buffer_bytes(const brk_xplan *p, uint32_t stage, int peer, uint32_t which);

/* bind to a storage on the current device: allocates the buffers and uploads the link tables (synchronous) */
int brk_exchange_view_create(const brk_xplan *p, void *storage_dev, brk_exchange_view **out);
void brk_exchange_view_free(brk_exchange_view *v);
/* the three steps of one stage, each asynchronous on `s`:  begin = links + pack (one brk_link_copy launch),
 * transport = ncclGroupStart; ncclRecv/ncclSend per message; ncclGroupEnd  on `nccl_comm` (an ncclComm_t),
 * end = unpack (one launch).  brk_exchange_run does all stages in order. */
int brk_exchange_begin(brk_exchange_view *v, uint32_t stage, brk_stream s);
int brk_exchange_transport(brk_exchange_view *v, uint32_t stage, void *nccl_comm, brk_stream s);
int brk_exchange_end(brk_exchange_view *v, uint32_t stage, brk_stream s);
int brk_exchange_run(brk_exchange_view *v, void *nccl_comm, brk_stream s);
/* transport between views that live on ONE GPU (several virtual ranks per GPU): every message of `stage` sent by
 * views[a] to the rank of views[b] is copied into the matching receive of views[b] (device copies on `s`, message order as
 * in the plans).  All views must come from plans of the same decomposition and mode. */
int brk_exchange_loopback(brk_exchange_view *const *views, uint32_t nviews, uint32_t stage, brk_stream s);
/* device address of a view's send / receive buffer for `peer` (NULL if none): for transports outside this library */
void *brk_exchange_view_buffer(const brk_exchange_view *v, uint32_t stage, int peer, uint32_t which);
/* The reference signature: exchange the ghosts of `storage_dev` over `nccl_comm`; this rank is
 * ncclCommUserRank(comm).  Packed mode; the plan and view are built on first use and kept inside the brk_decomp. */
int brk_exchange(brk_decomp *d, double *storage_dev, size_t step_elems, void *nccl_comm, const int rank_of_neighbor[27],
                 brk_stream s);

/* RCCL communicators without MPI or torch in the loop (librccl is loaded on first use).  id128 = ncclUniqueId (128 bytes):
 * rank 0 calls brk_comm_unique_id and hands the bytes to the others (MPI_Bcast, a file, torch.distributed, ...). */
int brk_comm_unique_id(void *id128);
int brk_comm_init_rank(void **nccl_comm, int nranks, const void *id128, int rank);
int brk_comm_destroy(void *nccl_comm);
int brk_comm_rank(void *nccl_comm, int *rank, int *nranks);
/* Which RCCL the transport is bound to: the file name of the library that answers ncclCommInitRank (an RCCL the process has
 * mapped already -- e.g. the one PyTorch ships -- is preferred over loading a second instance), or NULL when none can be bound
 * (brk_last_error says why).  Replaces nothing in the reference (MPI is linked there); it exists so that a run can record it. */
const char *brk_comm_library(void);

/* Strong scaling bookkeeping (include/zmort.h:18-105, strong/args.cpp:36-56,104-113) */
uint64_t brk_zmort_get(uint64_t id, uint64_t d);
uint64_t brk_zmort_set(uint64_t id, uint64_t d, uint64_t p);
void brk_strong_partition(uint64_t allsubs, int size, int rank, uint64_t out_lr[2]);
uint64_t brk_strong_neighbor(uint64_t neighbor_set, uint64_t zid, uint64_t subdim, uint64_t allsubs, int size,
                             int *owner_rank, int *owner_slot);

#ifdef __cplusplus
}
#endif
#endif /* BRK_H */
