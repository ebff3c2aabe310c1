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

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *brk_stream;

enum {
  BRK_OK = 0,
  BRK_EINVAL = -1,   /* bad argument (null pointer, unknown stencil id, misaligned/size mismatch) */
  BRK_EHIP = -2,     /* a HIP runtime call failed; brk_last_error() carries hipGetErrorString */
  BRK_ENOMEM = -3,
  BRK_ENODEV = -4    /* no gfx950 device / code object could not be loaded */
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
/* version of this ABI: 1000 + additions (1001: brk_stencil_apply_taps, 1002: the fused double-sweep stencil ids) */
int brk_abi_version(void);
/* radius / number of coefficients / communication-avoiding sweeps per exchange (ST_ITER, fake.h:39-344) */
int brk_stencil_radius(int stencil_id);
int brk_stencil_ncoeff(int stencil_id);
int brk_stencil_iters(int stencil_id);

/* ---- the hot path -------------------------------------------------------------------------------------------
 * out[b] = stencil(in[b] and its neighbours through adj) for every brick id b in [brick_begin,brick_end).
 * Replaces the body of `brick(...)` launched over a grid of bricks: d3pt7_brick_trans
 * (stencils/3axis.hip.cpp:32-41,165-171), brick_kernel (weak/main.cu:35-43,277-282).
 * in/out may be two fields of one interleaved storage (single/: out_dev = in_dev+512, step_elems = 1024,
 * stencils/3axis.cpp:25-28) or two storages (step_elems = 512, weak/main.cpp:76-77). */
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
/* Same over a structured box of a lexicographically numbered brick grid: id = base + (k*tot[1] + j)*tot[0] + i for
 * (i,j,k) in [lo, lo+cnt) (contiguous axis first).  This is exactly the reference launch `dim3 block(NB,NB,NB)` with
 * `b = grid[GB+z][GB+y][GB+x]` over an init_grid numbering (stencils/3axis.hip.cpp:36-39,168-170: base 0, tot = STRIDEB^3,
 * lo = GB, cnt = NB) and the inner region [1,sep0) of a BrickDecomp (base 1, tot = cnt = inner extent, lo = 0).
 * The 7-point kernels derive the ids arithmetically in their preferred traversal order (no list is read); the other
 * stencils use an id list in that order which the library builds once per box geometry and keeps on the device (a cache of
 * 8 geometries per process; the first call for a geometry is synchronous). */
int brk_stencil_apply_box(int stencil_id, const uint32_t *adj_dev, const double *in_dev, double *out_dev,
                          size_t step_elems, uint32_t base, const uint32_t tot_bricks[3], const uint32_t lo[3],
                          const uint32_t cnt[3], const double *coeff_host, brk_stream s);
/* A linear stencil given as data -- the capability of the reference's code generator for scripts that are not one of the
 * built-in ones (codegen/vecscatter evaluating stencils/<name>.py; bricklib_amd/spec.py reads such scripts into this form):
 *   out[p] = sum_t weights[t] * in[p + (offsets[3t], offsets[3t+1], offsets[3t+2])]   (i, j, k offsets, each in [-2,2],
 *   1 <= ntaps <= 125, taps added in list order).  Bricks: brick_ids_dev[0..n) when non-NULL (BRK_LIST_SKIP allowed), else
 *   ids first..first+n-1.  General path: one LDS read + FMA per tap and point, not tuned like the built-in stencils. */
int brk_stencil_apply_taps(const uint32_t *adj_dev, const double *in_dev, double *out_dev, size_t step_elems,
                           const uint32_t *brick_ids_dev, uint32_t first, uint32_t n, uint32_t ntaps,
                           const int32_t *offsets_host, const double *weights_host, brk_stream s);
/* Kernel variant selection for experiments/benchmarks: 0 = default. */
int brk_set_variant(int stencil_id, int variant);

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
