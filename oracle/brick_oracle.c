/*
 * brick_oracle.c -- CPU restatement of the bricklib hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker* for the MI355X kernels in bricklib_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product path
 * (libbrk.so) never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py)
 * against fixtures in tests/golden/ that were produced by oracle/_ref/libbrickref.so, i.e. by
 * the reference's own headers (include/brick.h, bricksetup.h, brickcompare.h, brick-mpi.h,
 * bitset.h, zmort.h, stencils/fake.h) compiled in place from /root/reference
 * (see oracle/Makefile, oracle/ref_harness.cpp, oracle/gen_golden.py).
 *
 * All "ref:" citations are file:line under /root/reference.
 * Axis convention follows the reference: vectors are "contiguous first" (index 0 = i, the
 * unit-stride axis; index 2 = k).  Bricks are 8x8x8 doubles stored row-major k*64+j*8+i,
 * which is what Brick<Dim<8,8,8>,Dim<8,8>> (ref: stencils/hipvfold.h:5-6, include/brick.h:235-246)
 * and Dim<8> (AVX512) reduce to.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BS 8            /* brick edge */
#define BELEMS 512      /* 8*8*8 */

/* stencil ids -- identical to include/brk.h */
enum { ORC_7PT_COEFF = 0, ORC_27PT_COEFF = 1, ORC_MPI7 = 2, ORC_MPI13 = 3, ORC_MPI25 = 4, ORC_MPI125 = 5, ORC_COND7 = 6 };

/* ------------------------------------------------------------------------------------------
 * Deterministic inputs.  The reference seeds from random_device (ref: src/multiarray.cpp:13-23,
 * single/cpu.cpp:11-17), so it has no reproducible inputs; we use a counter hash of the global
 * coordinate so that every decomposition of one global field sees identical values.
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

double orc_hash_u01(uint64_t seed, uint64_t idx) {
  uint64_t h = orc_splitmix64(seed * 0xD1342543DE82EF95ull + idx);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0); /* 53 bits -> [0,1) */
}

/* arr has extents ext[0..2] (i first); element (i,j,k) holds the value of global periodic
 * coordinate ((origin+idx) mod gdims). */
void orc_fill_array_global(double *arr, const long ext[3], const long origin[3], const long gdims[3],
                           uint64_t seed) {
#pragma omp parallel for
  for (long k = 0; k < ext[2]; ++k) {
    long gk = ((origin[2] + k) % gdims[2] + gdims[2]) % gdims[2];
    for (long j = 0; j < ext[1]; ++j) {
      long gj = ((origin[1] + j) % gdims[1] + gdims[1]) % gdims[1];
      for (long i = 0; i < ext[0]; ++i) {
        long gi = ((origin[0] + i) % gdims[0] + gdims[0]) % gdims[0];
        uint64_t lin = ((uint64_t)gk * (uint64_t)gdims[1] + (uint64_t)gj) * (uint64_t)gdims[0] + (uint64_t)gi;
        arr[(k * ext[1] + j) * ext[0] + i] = orc_hash_u01(seed, lin);
      }
    }
  }
}

/* coeff[i] ~ U[0,1) (ref: single/cpu.cpp:11-17 draws 129 of them) */
void orc_coeffs(double *coeff, int n, uint64_t seed) {
  for (int i = 0; i < n; ++i)
    coeff[i] = orc_hash_u01(seed ^ 0xC0EFFull, (uint64_t)i);
}

/* ------------------------------------------------------------------------------------------
 * init_grid<3>  (ref: include/bricksetup.h:30-90)
 * grid[pos] = pos, i fastest.  adj[b][(dk+1)*9+(dj+1)*3+(di+1)] = grid value at the flat
 * position pos + dk*s2 + dj*s1 + di*s0 when that flat position lies inside [0,size), else 0.
 * The bounds test is on the flat pointer only (bricksetup.h:44), so bricks on the outer ring get
 * wrapped neighbours; that quirk is reproduced.
 * ---------------------------------------------------------------------------------------- */
void orc_init_grid(const long dims[3], uint32_t *grid, uint32_t *adj) {
  long s[3] = {1, dims[0], dims[0] * dims[1]};
  long size = dims[0] * dims[1] * dims[2];
  for (long p = 0; p < size; ++p)
    grid[p] = (uint32_t)p;
#pragma omp parallel for
  for (long p = 0; p < size; ++p) {
    uint32_t *a = adj + (size_t)grid[p] * 27;
    for (int dk = -1; dk <= 1; ++dk)
      for (int dj = -1; dj <= 1; ++dj)
        for (int di = -1; di <= 1; ++di) {
          long q = p + dk * s[2] + dj * s[1] + di * s[0];
          a[(dk + 1) * 9 + (dj + 1) * 3 + (di + 1)] = (q >= 0 && q < size) ? grid[q] : 0u;
        }
  }
}

/* ------------------------------------------------------------------------------------------
 * _BrickAccessor semantics (ref: include/brick.h:214-327): b[brick][k][j][i] with each index in
 * [-8,15]; index < 0 selects neighbour -1 and in-brick coordinate idx+8, index >= 8 selects
 * neighbour +1 and idx-8.
 * ---------------------------------------------------------------------------------------- */
double orc_brick_at(const uint32_t *adj, const double *dat, size_t step, uint32_t b, int k, int j, int i) {
  int dk = (k + BS) / BS, dj = (j + BS) / BS, di = (i + BS) / BS; /* 0,1,2 */
  int lk = (k + BS) % BS, lj = (j + BS) % BS, li = (i + BS) % BS;
  uint32_t nb = adj[(size_t)b * 27 + dk * 9 + dj * 3 + di];
  return dat[(size_t)nb * step + lk * 64 + lj * 8 + li];
}

/* ------------------------------------------------------------------------------------------
 * iter_grid / copyToBrick / copyFromBrick / compareBrick
 * (ref: include/bricksetup.h:92-221, include/brickcompare.h:30-57)
 * dimlist = iterated extent (cells), padding = array-only halo, ghost = halo present in both
 * and skipped.  Array extent per axis = dimlist+2*(padding+ghost); brick-grid extent =
 * (dimlist+2*ghost)/8.  Brick-grid index s runs over [ghost/8,(dimlist+ghost)/8) and maps to
 * array coordinate padding + s*8 (bricksetup.h:110-121).
 * mode: 0 = array -> brick, 1 = brick -> array, 2 = compare (returns 1 if all match).
 * ---------------------------------------------------------------------------------------- */
static int iter_grid3(const long dimlist[3], const long padding[3], const long ghost[3], double *arr,
                      const uint32_t *grid, double *bdat, size_t step, int mode, double tol) {
  long sA[3], sB[3], a = 1, b = 1;
  for (int d = 0; d < 3; ++d) {
    sA[d] = a;
    sB[d] = b;
    a *= dimlist[d] + 2 * (padding[d] + ghost[d]);
    b *= (dimlist[d] + 2 * ghost[d]) / BS;
  }
  long lo[3], hi[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = ghost[d] / BS;
    hi[d] = (dimlist[d] + ghost[d]) / BS;
  }
  int ok = 1;
#pragma omp parallel for collapse(2) reduction(&& : ok)
  for (long tk = lo[2]; tk < hi[2]; ++tk)
    for (long tj = lo[1]; tj < hi[1]; ++tj)
      for (long ti = lo[0]; ti < hi[0]; ++ti) {
        uint32_t bid = grid[tk * sB[2] + tj * sB[1] + ti * sB[0]];
        double *bp = bdat + (size_t)bid * step;
        double *ap = arr + (padding[2] + tk * BS) * sA[2] + (padding[1] + tj * BS) * sA[1] +
                     (padding[0] + ti * BS) * sA[0];
        for (int k = 0; k < BS; ++k)
          for (int j = 0; j < BS; ++j)
            for (int i = 0; i < BS; ++i) {
              double *ae = ap + k * sA[2] + j * sA[1] + i;
              double *be = bp + k * 64 + j * 8 + i;
              if (mode == 0)
                *be = *ae;
              else if (mode == 1)
                *ae = *be;
              else {
                double diff = fabs(*be - *ae);
                int r = (diff < tol) || (diff < (fabs(*be) + fabs(*ae)) * tol);
                ok = ok && r;
              }
            }
      }
  return ok;
}

void orc_copy_to_brick(const long dimlist[3], const long padding[3], const long ghost[3], const double *arr,
                       const uint32_t *grid, double *bdat, size_t step) {
  iter_grid3(dimlist, padding, ghost, (double *)arr, grid, bdat, step, 0, 0.0);
}

void orc_copy_from_brick(const long dimlist[3], const long padding[3], const long ghost[3], double *arr,
                         const uint32_t *grid, const double *bdat, size_t step) {
  iter_grid3(dimlist, padding, ghost, arr, grid, (double *)bdat, step, 1, 0.0);
}

/* ref: include/brickcompare.h:35-39, include/cmpconst.h:9 (tol = 1e-6 there) */
int orc_compare_brick(const long dimlist[3], const long padding[3], const long ghost[3], const double *arr,
                      const uint32_t *grid, const double *bdat, size_t step, double tol) {
  return iter_grid3(dimlist, padding, ghost, (double *)arr, grid, (double *)bdat, step, 2, tol);
}

/* max |brick-array| over the same iteration space (not in the reference; reporting only) */
double orc_maxdiff_brick(const long dimlist[3], const long padding[3], const long ghost[3], const double *arr,
                         const uint32_t *grid, const double *bdat, size_t step) {
  long sA[3], sB[3], a = 1, b = 1;
  for (int d = 0; d < 3; ++d) {
    sA[d] = a;
    sB[d] = b;
    a *= dimlist[d] + 2 * (padding[d] + ghost[d]);
    b *= (dimlist[d] + 2 * ghost[d]) / BS;
  }
  double m = 0.0;
#pragma omp parallel for reduction(max : m)
  for (long tk = ghost[2] / BS; tk < (dimlist[2] + ghost[2]) / BS; ++tk)
    for (long tj = ghost[1] / BS; tj < (dimlist[1] + ghost[1]) / BS; ++tj)
      for (long ti = ghost[0] / BS; ti < (dimlist[0] + ghost[0]) / BS; ++ti) {
        uint32_t bid = grid[tk * sB[2] + tj * sB[1] + ti * sB[0]];
        const double *bp = bdat + (size_t)bid * step;
        const double *ap = arr + (padding[2] + tk * BS) * sA[2] + (padding[1] + tj * BS) * sA[1] +
                           (padding[0] + ti * BS) * sA[0];
        for (int k = 0; k < BS; ++k)
          for (int j = 0; j < BS; ++j)
            for (int i = 0; i < BS; ++i) {
              double d = fabs(bp[k * 64 + j * 8 + i] - ap[k * sA[2] + j * sA[1] + i]);
              if (d > m)
                m = d;
            }
      }
  return m;
}

/* ------------------------------------------------------------------------------------------
 * Stencil definitions.
 *  ORC_7PT_COEFF : ref stencils/3axis.cpp:33-36,60-63 (term order kept), coeff[0..6]
 *  ORC_27PT_COEFF: ref stencils/3axis.cpp:184-202,214-233 incl. the quirk at :195/:225 --
 *                  coeff[16] multiplies in[k+1][j+1][i] (not in[k][j+1][i])
 *  ORC_MPI7      : ref stencils/fake.h:346-349  (sum of 6)*BETA + c*ALPHA
 *  ORC_MPI13     : ref stencils/fake.h:71-77
 *  ORC_MPI25     : ref stencils/fake.h:41-53
 *  ORC_COND7     : ref stencils/3axis.cpp:104-111,137-146 (d3cond): |sum c_t * max(in_t, 0)|, 7-pt term order
 *  ORC_MPI125    : ref stencils/fake.h:90-215  (per-class sums times C0..C9; class = sorted
 *                  |offset| triple, stencils/mpi125pt.py:12-33)
 * The point function works on any array with strides (1, sj, sk).
 * ---------------------------------------------------------------------------------------- */
static const double dflt_mpi7[2] = {0.4, 0.1};                                          /* ALPHA, BETA  fake.h:11-12 */
static const double dflt_mpi13[3] = {0.4, 0.07, 0.03};                                  /* B0..B2       fake.h:20-22 */
static const double dflt_mpi25[5] = {0.1, 0.06, 0.045, 0.03, 0.015};                    /* A0..A4       fake.h:14-18 */
static const double dflt_mpi125[10] = {0.1, 0.04, 0.03, 0.01, 0.006, 0.004, 0.005, 0.002, 0.003, 0.001}; /* fake.h:24-33 */

int orc_stencil_radius(int id) {
  switch (id) {
  case ORC_7PT_COEFF: return 1;
  case ORC_27PT_COEFF: return 1;
  case ORC_MPI7: return 1;
  case ORC_MPI13: return 2;
  case ORC_MPI25: return 4;
  case ORC_MPI125: return 2;
  case ORC_COND7: return 1;
  }
  return -1;
}

int orc_stencil_is_star(int id) { return id == ORC_7PT_COEFF || id == ORC_MPI7 || id == ORC_MPI13 || id == ORC_MPI25 || id == ORC_COND7; }

int orc_stencil_ncoeff(int id) {
  switch (id) {
  case ORC_7PT_COEFF: return 7;
  case ORC_27PT_COEFF: return 27;
  case ORC_MPI7: return 2;
  case ORC_MPI13: return 3;
  case ORC_MPI25: return 5;
  case ORC_MPI125: return 10;
  case ORC_COND7: return 7;
  }
  return -1;
}

const double *orc_stencil_default_coeff(int id) {
  switch (id) {
  case ORC_MPI7: return dflt_mpi7;
  case ORC_MPI13: return dflt_mpi13;
  case ORC_MPI25: return dflt_mpi25;
  case ORC_MPI125: return dflt_mpi125;
  }
  return NULL;
}

/* class index of a 125-pt offset: sorted(|di|,|dj|,|dk|) -> 0..9 (mpi125pt.py:12-33) */
static int cls125(int a, int b, int c) {
  a = abs(a); b = abs(b); c = abs(c);
  int t;
  if (a > b) { t = a; a = b; b = t; }
  if (b > c) { t = b; b = c; c = t; }
  if (a > b) { t = a; a = b; b = t; }
  /* (a<=b<=c) in {0,1,2}^3 */
  static const int tab[3][3][3] = {
      /* a=0 */ {{0, 1, 2}, {-1, 3, 4}, {-1, -1, 5}},
      /* a=1 */ {{-1, -1, -1}, {-1, 6, 7}, {-1, -1, 8}},
      /* a=2 */ {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, 9}}};
  return tab[a][b][c];
}

#define AT(dk, dj, di) p[(long)(dk) * sk + (long)(dj) * sj + (di)]

static inline double stencil_point(int id, const double *p, long sj, long sk, const double *c) {
  switch (id) {
  case ORC_7PT_COEFF:
    return c[5] * AT(1, 0, 0) + c[6] * AT(-1, 0, 0) + c[3] * AT(0, 1, 0) + c[4] * AT(0, -1, 0) +
           c[1] * AT(0, 0, 1) + c[2] * AT(0, 0, -1) + c[0] * AT(0, 0, 0);
  case ORC_COND7: {
#define POS(x) ((x) > 0.0 ? (x) : 0.0)
    double partial = c[5] * POS(AT(1, 0, 0)) + c[6] * POS(AT(-1, 0, 0)) + c[3] * POS(AT(0, 1, 0)) + c[4] * POS(AT(0, -1, 0)) +
                     c[1] * POS(AT(0, 0, 1)) + c[2] * POS(AT(0, 0, -1)) + c[0] * POS(AT(0, 0, 0));
#undef POS
    return partial > 0 ? partial : -partial;
  }
  case ORC_27PT_COEFF:
    return c[0] * AT(-1, -1, -1) + c[1] * AT(-1, -1, 0) + c[2] * AT(-1, -1, 1) +
           c[3] * AT(-1, 0, -1) + c[4] * AT(-1, 0, 0) + c[5] * AT(-1, 0, 1) +
           c[6] * AT(-1, 1, -1) + c[7] * AT(-1, 1, 0) + c[8] * AT(-1, 1, 1) +
           c[9] * AT(0, -1, -1) + c[10] * AT(0, -1, 0) + c[11] * AT(0, -1, 1) +
           c[12] * AT(0, 0, -1) + c[13] * AT(0, 0, 0) + c[14] * AT(0, 0, 1) +
           c[15] * AT(0, 1, -1) + c[16] * AT(1, 1, 0) /* quirk: 3axis.cpp:195 */ + c[17] * AT(0, 1, 1) +
           c[18] * AT(1, -1, -1) + c[19] * AT(1, -1, 0) + c[20] * AT(1, -1, 1) +
           c[21] * AT(1, 0, -1) + c[22] * AT(1, 0, 0) + c[23] * AT(1, 0, 1) +
           c[24] * AT(1, 1, -1) + c[25] * AT(1, 1, 0) + c[26] * AT(1, 1, 1);
  case ORC_MPI7:
    return (AT(1, 0, 0) + AT(-1, 0, 0) + AT(0, 1, 0) + AT(0, -1, 0) + AT(0, 0, 1) + AT(0, 0, -1)) * c[1] +
           AT(0, 0, 0) * c[0];
  case ORC_MPI13:
    return (AT(2, 0, 0) + AT(-2, 0, 0) + AT(0, 2, 0) + AT(0, -2, 0) + AT(0, 0, 2) + AT(0, 0, -2)) * c[2] +
           (AT(1, 0, 0) + AT(-1, 0, 0) + AT(0, 1, 0) + AT(0, -1, 0) + AT(0, 0, 1) + AT(0, 0, -1)) * c[1] +
           AT(0, 0, 0) * c[0];
  case ORC_MPI25: {
    double r = 0.0;
    /* fake.h:41-53: shells 4,3,2,1 then centre, each (sum of 6)*A_s added left to right */
    for (int s = 4; s >= 1; --s) {
      double sh = AT(s, 0, 0) + AT(-s, 0, 0) + AT(0, s, 0) + AT(0, -s, 0) + AT(0, 0, s) + AT(0, 0, -s);
      r = (s == 4) ? sh * c[4] : r + sh * c[s];
    }
    return r + AT(0, 0, 0) * c[0];
  }
  case ORC_MPI125: {
    double sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int dk = -2; dk <= 2; ++dk)
      for (int dj = -2; dj <= 2; ++dj)
        for (int di = -2; di <= 2; ++di)
          sum[cls125(di, dj, dk)] += AT(dk, dj, di);
    double r = c[0] * sum[0];
    for (int q = 1; q < 10; ++q)
      r += c[q] * sum[q];
    return r;
  }
  }
  return 0.0;
}

/* Array sweep over [lo,hi) (array coordinates, i first); ext = array extents.
 * ref: stencils/3axis.cpp:32-37 (_TILEFOR arr_func), weak/main.cpp:116-131 (array_stencil). */
void orc_array_stencil(int id, const double *in, double *out, const long ext[3], const long lo[3],
                       const long hi[3], const double *coeff) {
  if (!coeff)
    coeff = orc_stencil_default_coeff(id);
  long sj = ext[0], sk = ext[0] * ext[1];
#pragma omp parallel for collapse(2)
  for (long k = lo[2]; k < hi[2]; ++k)
    for (long j = lo[1]; j < hi[1]; ++j)
      for (long i = lo[0]; i < hi[0]; ++i) {
        const double *p = in + k * sk + j * sj + i;
        out[k * sk + j * sj + i] = stencil_point(id, p, sj, sk, coeff);
      }
}

/* 7-point sweeps without the tile gather: rows of 8 are processed straight from the brick and its six face neighbours
 * (same accessor semantics, same term order as stencil_point, so results are bit-identical to the generic path; this
 * is only the faster formulation used for the cpu_baseline timing: contiguous 8-wide inner loop the compiler vectorises). */
static void brick_stencil_star7(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                                const uint32_t *ids, long n, uint32_t first, const double *c) {
#pragma omp parallel for schedule(static)
  for (long q = 0; q < n; ++q) {
    uint32_t b = ids ? ids[q] : (uint32_t)(first + q);
    const uint32_t *a = adj + (size_t)b * 27;
    const double *src = in + (size_t)b * step;
    const double *km = in + (size_t)a[4] * step, *kp = in + (size_t)a[22] * step;
    const double *jm = in + (size_t)a[10] * step, *jp = in + (size_t)a[16] * step;
    const double *im = in + (size_t)a[12] * step, *ip = in + (size_t)a[14] * step;
    double *dst = out + (size_t)b * step;
    for (int k = 0; k < BS; ++k)
      for (int j = 0; j < BS; ++j) {
        const double *cc = src + k * 64 + j * 8;
        const double *up = k < 7 ? cc + 64 : kp + j * 8;
        const double *dn = k > 0 ? cc - 64 : km + 7 * 64 + j * 8;
        const double *jn = j < 7 ? cc + 8 : jp + k * 64;
        const double *js = j > 0 ? cc - 8 : jm + k * 64 + 56;
        double row[10];
        row[0] = im[k * 64 + j * 8 + 7];
        for (int i = 0; i < 8; ++i) row[i + 1] = cc[i];
        row[9] = ip[k * 64 + j * 8];
        double *o = dst + k * 64 + j * 8;
        if (id == ORC_7PT_COEFF) {
#pragma omp simd
          for (int i = 0; i < 8; ++i)
            o[i] = c[5] * up[i] + c[6] * dn[i] + c[3] * jn[i] + c[4] * js[i] + c[1] * row[i + 2] + c[2] * row[i] + c[0] * row[i + 1];
        } else {
#pragma omp simd
          for (int i = 0; i < 8; ++i)
            o[i] = (up[i] + dn[i] + jn[i] + js[i] + row[i + 2] + row[i]) * c[1] + row[i + 1] * c[0];
        }
      }
  }
}

/* Brick sweep over a list of brick ids (or the range [0,n) + first when ids == NULL).
 * Semantics = naive accessor form (ref: stencils/3axis.cpp:51-66,205-235, weak/main.cpp:26-36):
 * every tap is bIn[b][k+dk][j+dj][i+di] through the adjacency list.  Implemented by gathering
 * the (8+2r)^3 neighbourhood of the brick (star stencils: faces only) and applying the point
 * function on that tile. */
void orc_brick_stencil(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                       const uint32_t *ids, long n, uint32_t first, const double *coeff) {
  if (!coeff)
    coeff = orc_stencil_default_coeff(id);
  if (id == ORC_7PT_COEFF || id == ORC_MPI7) {
    brick_stencil_star7(id, adj, in, out, step, ids, n, first, coeff);
    return;
  }
  const int R = orc_stencil_radius(id);
  const int star = orc_stencil_is_star(id);
  const int T = BS + 2 * R;
#pragma omp parallel
  {
    double *tile = (double *)malloc(sizeof(double) * T * T * T);
    memset(tile, 0, sizeof(double) * T * T * T);
#pragma omp for schedule(static)
    for (long q = 0; q < n; ++q) {
      uint32_t b = ids ? ids[q] : (uint32_t)(first + q);
      const uint32_t *a = adj + (size_t)b * 27;
      for (int dk = -1; dk <= 1; ++dk)
        for (int dj = -1; dj <= 1; ++dj)
          for (int di = -1; di <= 1; ++di) {
            if (star && (abs(dk) + abs(dj) + abs(di) > 1))
              continue;
            const double *src = in + (size_t)a[(dk + 1) * 9 + (dj + 1) * 3 + (di + 1)] * step;
            int k0 = dk < 0 ? BS - R : 0, k1 = dk > 0 ? R : BS, tk0 = dk < 0 ? 0 : (dk == 0 ? R : R + BS);
            int j0 = dj < 0 ? BS - R : 0, j1 = dj > 0 ? R : BS, tj0 = dj < 0 ? 0 : (dj == 0 ? R : R + BS);
            int i0 = di < 0 ? BS - R : 0, i1 = di > 0 ? R : BS, ti0 = di < 0 ? 0 : (di == 0 ? R : R + BS);
            for (int k = k0; k < k1; ++k)
              for (int j = j0; j < j1; ++j)
                for (int i = i0; i < i1; ++i)
                  tile[((tk0 + k - k0) * T + (tj0 + j - j0)) * T + (ti0 + i - i0)] = src[k * 64 + j * 8 + i];
          }
      double *dst = out + (size_t)b * step;
      for (int k = 0; k < BS; ++k)
        for (int j = 0; j < BS; ++j)
          for (int i = 0; i < BS; ++i)
            dst[k * 64 + j * 8 + i] =
                stencil_point(id, tile + ((k + R) * T + (j + R)) * T + (i + R), T, (long)T * T, coeff);
    }
    free(tile);
  }
}

/* Same sweep, literally through orc_brick_at for every tap (slow; cross-checks the tile gather). */
void orc_brick_stencil_naive(int id, const uint32_t *adj, const double *in, double *out, size_t step,
                             const uint32_t *ids, long n, const double *coeff) {
  if (!coeff)
    coeff = orc_stencil_default_coeff(id);
  const int R = orc_stencil_radius(id);
  const int T = 2 * R + 1;
  for (long q = 0; q < n; ++q) {
    uint32_t b = ids[q];
    for (int k = 0; k < BS; ++k)
      for (int j = 0; j < BS; ++j)
        for (int i = 0; i < BS; ++i) {
          double loc[9 * 9 * 9];
          for (int dk = -R; dk <= R; ++dk)
            for (int dj = -R; dj <= R; ++dj)
              for (int di = -R; di <= R; ++di)
                loc[((dk + R) * T + (dj + R)) * T + (di + R)] = orc_brick_at(adj, in, step, b, k + dk, j + dj, i + di);
          out[(size_t)b * step + k * 64 + j * 8 + i] =
              stencil_point(id, loc + (R * T + R) * T + R, T, (long)T * T, coeff);
        }
  }
}

/* A linear stencil given as tap data, through the accessor semantics (what the reference's generated code computes for
 * `calc = c0 * in(i+a, j+b, k+c) + ...`, stencils/7pt.py:19-27, codegen evaluating the expression tap by tap): taps are
 * added in list order starting from the first product. */
void orc_brick_taps(const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, long n,
                    int ntaps, const int *offsets /* [ntaps][3] = (di,dj,dk) */, const double *weights) {
  for (long q = 0; q < n; ++q) {
    uint32_t b = ids[q];
    for (int k = 0; k < BS; ++k)
      for (int j = 0; j < BS; ++j)
        for (int i = 0; i < BS; ++i) {
          double acc = 0.0;
          for (int t = 0; t < ntaps; ++t)
            acc += weights[t] * orc_brick_at(adj, in, step, b, k + offsets[3 * t + 2], j + offsets[3 * t + 1], i + offsets[3 * t]);
          out[(size_t)b * step + k * 64 + j * 8 + i] = acc;
        }
  }
}

/* ------------------------------------------------------------------------------------------
 * BitSet (ref: include/bitset.h:20-126): +p -> bit p, -p -> bit 31+p; operator! swaps signs.
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_bitset_of(const int *elems, int n) {
  uint64_t s = 0;
  for (int q = 0; q < n; ++q) {
    long pos = elems[q];
    if (pos < 0)
      pos = 31 - pos;
    s ^= 1ull << (uint64_t)pos;
  }
  return s;
}

static int bs_get(uint64_t s, long pos) {
  if (pos < 0)
    pos = 31 - pos;
  return (s >> (uint64_t)pos) & 1ull;
}

uint64_t orc_bitset_neg(uint64_t s) {
  uint64_t mask = (1ull << 32) - 1ull;
  return ((s & mask) << 31) | (s >> 31);
}

/* allneighbors(0,1,3,out) (ref: src/brick-mpi.cpp:9-23): per dimension try '+', none, '-'
 * with dimension 1 outermost -> 27 sets, index 13 is the empty set. */
static void allneigh_rec(uint64_t cur, long idx, long dim, uint64_t *out, int *n) {
  if (idx > dim) {
    out[(*n)++] = cur;
    return;
  }
  allneigh_rec(cur ^ (1ull << idx), idx + 1, dim, out, n);
  allneigh_rec(cur, idx + 1, dim, out, n);
  allneigh_rec(cur ^ (1ull << (31 + idx)), idx + 1, dim, out, n);
}

void orc_allneighbors3(uint64_t out[27]) {
  int n = 0;
  allneigh_rec(0, 1, 3, out, &n);
}

/* skin3d_good (ref: src/brick-mpi.cpp:25-52) */
static const int skin3d_good_tab[26][3] = {
    {1, 0, 0},   {1, -3, 0},  {1, 2, -3},  {1, 2, 0},    {1, 2, 3},   {2, 3, 0},   {2, 0, 0},
    {2, -3, 0},  {-1, 2, -3}, {-1, 2, 0},  {-1, 2, 3},   {-1, 3, 0},  {-1, 0, 0},  {-3, 0, 0},
    {-1, -3, 0}, {-1, -2, -3}, {-1, -2, 0}, {-1, -2, 3}, {-2, 3, 0},  {-2, 0, 0},  {-2, -3, 0},
    {1, -2, -3}, {1, -2, 0},  {1, -2, 3},  {1, 3, 0},    {3, 0, 0}};

void orc_skin3d_good(uint64_t out[26]) {
  for (int q = 0; q < 26; ++q) {
    int e[3], n = 0;
    for (int t = 0; t < 3; ++t)
      if (skin3d_good_tab[q][t])
        e[n++] = skin3d_good_tab[q][t];
    out[q] = orc_bitset_of(e, n);
  }
}

/* ------------------------------------------------------------------------------------------
 * BrickDecomp<3,8,8,8>(dims, depth).initialize(skin3d_good)
 * (ref: include/brick-mpi.h:178-460).  Restated for brick size == page size (factor 1, all pads
 * 0: brick-mpi.h:342-368 is then a no-op) which is the only case the 8^3 fp64 path produces.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  uint64_t neighbor;
  uint32_t skin_st, skin_ed, pos, len, first_pad, last_pad;
} orc_region;

typedef struct {
  uint32_t dims[3], t_dims[3], g[3], stride[3];
  uint32_t sep_pos[3];
  uint32_t nbricks, grid_size;
  uint32_t *grid, *adj;
  int nskin;
  uint64_t skinlist[64];
  long skin_size[64];
  int nruns;
  orc_region ghost[128], skin[128];
} orc_decomp;

/* brick-mpi.h:218-239 : number the bricks of `region` of the domain displaced by `ref` */
static void dec_populate_rec(orc_decomp *D, uint64_t region, long ref, int d, uint32_t *pos) {
  if (d == -1) {
    D->grid[ref] = (*pos)++;
    return;
  }
  long sec = 0, st = 0;
  if (bs_get(region, d + 1)) {
    sec = 1;
    st = D->dims[d];
  }
  if (bs_get(region, -d - 1)) {
    sec = -1;
    st = D->g[d];
  }
  if (sec != 0) {
    for (uint32_t i = 0; i < D->g[d]; ++i)
      dec_populate_rec(D, region, ref + (st + i) * (long)D->stride[d], d - 1, pos);
  } else {
    for (uint32_t i = 2 * D->g[d]; i < D->dims[d]; ++i)
      dec_populate_rec(D, region, ref + i * (long)D->stride[d], d - 1, pos);
  }
}

/* brick-mpi.h:241-254 */
static void dec_populate(orc_decomp *D, uint64_t owner, uint64_t region, uint32_t *pos) {
  long ref = 0;
  for (int d = 0; d < 3; ++d) {
    if (bs_get(owner, d + 1))
      ref += (long)D->dims[d] * D->stride[d];
    if (bs_get(owner, -d - 1))
      ref -= (long)D->dims[d] * D->stride[d];
  }
  dec_populate_rec(D, region, ref, 2, pos);
}

static int subset(uint64_t a, uint64_t b) { return (a & b) == a; }

/* skin3d_normal / skin3d_bad (ref: src/brick-mpi.cpp:54-136): lexicographic order of (axis 3, axis 2, axis 1) in {-, none, +}
 * with the empty set in the middle / the same 26 sections with an empty set between any two.  which: 0 good, 1 normal, 2 bad */
int orc_skin3d(int which, uint64_t out[64]) {
  if (which == 0) {
    orc_skin3d_good(out);
    return 26;
  }
  int n = 0;
  const int pick[3] = {-1, 0, 1};
  for (int c = 0; c < 3; ++c)
    for (int b = 0; b < 3; ++b)
      for (int a = 0; a < 3; ++a) {
        int e[3], m = 0;
        if (pick[a]) e[m++] = pick[a] * 1;
        if (pick[b]) e[m++] = pick[b] * 2;
        if (pick[c]) e[m++] = pick[c] * 3;
        const uint64_t set = orc_bitset_of(e, m);
        if (which == 1) {
          out[n++] = set;
        } else if (set) {
          if (n) out[n++] = 0;
          out[n++] = set;
        }
      }
  return n;
}

orc_decomp *orc_decomp_build_order(const uint32_t dims_cells[3], uint32_t ghost_cells, const uint64_t *order, int norder);
orc_decomp *orc_decomp_build(const uint32_t dims_cells[3], uint32_t ghost_cells) {
  uint64_t good[64];
  orc_skin3d_good(good);
  return orc_decomp_build_order(dims_cells, ghost_cells, good, 26);
}

orc_decomp *orc_decomp_build_order(const uint32_t dims_cells[3], uint32_t ghost_cells, const uint64_t *order, int norder) {
  orc_decomp *D = (orc_decomp *)calloc(1, sizeof(orc_decomp));
  uint32_t gs = 1;
  for (int d = 0; d < 3; ++d) {
    D->dims[d] = dims_cells[d] / BS;
    D->g[d] = ghost_cells / BS;
    D->stride[d] = gs;
    D->t_dims[d] = D->dims[d] + 2 * D->g[d];
    gs *= D->t_dims[d];
  }
  D->grid_size = gs;
  D->grid = (uint32_t *)malloc(sizeof(uint32_t) * gs);
  for (int l = 0; l < norder; ++l) D->skinlist[l] = order[l];
  D->nskin = norder;

  uint32_t pos = 1; /* factor == 1: id 0 is the dummy brick (brick-mpi.h:353) */
  uint32_t st_pos[72];
  int nst = 0;

  dec_populate(D, 0, 0, &pos); /* inner region */
  st_pos[nst++] = pos;
  D->sep_pos[0] = pos;
  for (int l = 0; l < D->nskin; ++l) { /* surface regions, brick-mpi.h:389-398 */
    uint32_t ppos = pos;
    if (D->skinlist[l] != 0)
      dec_populate(D, 0, D->skinlist[l], &pos);
    st_pos[nst++] = pos;
    D->skin_size[l] = (long)pos - (long)ppos;
  }
  D->sep_pos[1] = pos;

  /* ghost regions, brick-mpi.h:404-452 */
  uint64_t nb[27];
  orc_allneighbors3(nb);
  D->nruns = 0;
  for (int q = 0; q < 27; ++q) {
    uint64_t n = nb[q];
    if (n == 0)
      continue;
    uint64_t in = orc_bitset_neg(n);
    orc_region g, s;
    memset(&g, 0, sizeof g);
    memset(&s, 0, sizeof s);
    g.neighbor = n;
    s.neighbor = in;
    int last = -1;
    for (int l = 0; l < D->nskin; ++l) {
      if (subset(in, D->skinlist[l])) {
        if (last < 0) {
          last = l;
          s.skin_st = g.skin_st = (uint32_t)l;
          g.pos = pos;
          s.pos = st_pos[l];
        }
        dec_populate(D, n, D->skinlist[l], &pos);
      } else if (last >= 0) {
        last = l;
        g.skin_ed = s.skin_ed = (uint32_t)last;
        g.len = pos - g.pos;
        s.len = st_pos[last] - s.pos;
        D->ghost[D->nruns] = g;
        D->skin[D->nruns] = s;
        D->nruns++;
        last = -1;
      }
    }
    if (last >= 0) {
      last = D->nskin;
      g.skin_ed = s.skin_ed = (uint32_t)last;
      g.len = pos - g.pos;
      s.len = st_pos[D->nskin] - s.pos;
      D->ghost[D->nruns] = g;
      D->skin[D->nruns] = s;
      D->nruns++;
    }
  }
  D->sep_pos[2] = pos;
  D->nbricks = pos;

  /* adjacency from the grid, brick-mpi.h:266-291,455-459; row 0 (dummy) is all zero here
   * (uninitialised malloc in the reference) */
  D->adj = (uint32_t *)calloc((size_t)pos * 27, sizeof(uint32_t));
  for (uint32_t ck = 0; ck < D->t_dims[2]; ++ck)
    for (uint32_t cj = 0; cj < D->t_dims[1]; ++cj)
      for (uint32_t ci = 0; ci < D->t_dims[0]; ++ci) {
        long ref = (long)ck * D->stride[2] + (long)cj * D->stride[1] + ci;
        uint32_t *a = D->adj + (size_t)D->grid[ref] * 27;
        for (int dk = -1; dk <= 1; ++dk)
          for (int dj = -1; dj <= 1; ++dj)
            for (int di = -1; di <= 1; ++di) {
              long nk = (long)ck + dk, nj = (long)cj + dj, ni = (long)ci + di;
              int inb = nk >= 0 && nk < D->t_dims[2] && nj >= 0 && nj < D->t_dims[1] && ni >= 0 && ni < D->t_dims[0];
              a[(dk + 1) * 9 + (dj + 1) * 3 + (di + 1)] =
                  inb ? D->grid[nk * D->stride[2] + nj * D->stride[1] + ni] : 0u;
            }
      }
  return D;
}

void orc_decomp_free(orc_decomp *D) {
  if (!D)
    return;
  free(D->grid);
  free(D->adj);
  free(D);
}

/* info[0..2]=sep_pos, [3]=nbricks, [4]=grid_size, [5]=nruns, [6]=nskin, [7..9]=t_dims */
void orc_decomp_info(const orc_decomp *D, uint32_t info[10]) {
  info[0] = D->sep_pos[0]; info[1] = D->sep_pos[1]; info[2] = D->sep_pos[2];
  info[3] = D->nbricks; info[4] = D->grid_size; info[5] = (uint32_t)D->nruns; info[6] = (uint32_t)D->nskin;
  info[7] = D->t_dims[0]; info[8] = D->t_dims[1]; info[9] = D->t_dims[2];
}
void orc_decomp_grid(const orc_decomp *D, uint32_t *out) { memcpy(out, D->grid, sizeof(uint32_t) * D->grid_size); }
void orc_decomp_adj(const orc_decomp *D, uint32_t *out) { memcpy(out, D->adj, sizeof(uint32_t) * 27 * (size_t)D->nbricks); }
void orc_decomp_skin_size(const orc_decomp *D, int64_t *out) {
  for (int l = 0; l < D->nskin; ++l)
    out[l] = D->skin_size[l];
}
/* which: 0 = ghost, 1 = skin.  Row layout: neighbor, skin_st, skin_ed, pos, len, first_pad, last_pad */
void orc_decomp_runs(const orc_decomp *D, int which, uint64_t *out) {
  const orc_region *r = which ? D->skin : D->ghost;
  for (int q = 0; q < D->nruns; ++q) {
    out[q * 7 + 0] = r[q].neighbor; out[q * 7 + 1] = r[q].skin_st; out[q * 7 + 2] = r[q].skin_ed;
    out[q * 7 + 3] = r[q].pos; out[q * 7 + 4] = r[q].len; out[q * 7 + 5] = r[q].first_pad; out[q * 7 + 6] = r[q].last_pad;
  }
}

/* ------------------------------------------------------------------------------------------
 * populate(): neighbour -> rank on a periodic Cartesian communicator
 * (ref: include/brick-mpi.h:730-753; MPI_Cart_rank is row-major over (coo[0],coo[1],coo[2]) with
 * periodic wrap).  Orientation quirk kept: direction +d moves coo[3-d] by -1.
 * out[q] is the rank for allneighbors()[q] (q = 13 is self).
 * ---------------------------------------------------------------------------------------- */
void orc_populate_rank_map(const int cart_dims[3], const int coo_in[3], int out[27]) {
  uint64_t nb[27];
  orc_allneighbors3(nb);
  for (int q = 0; q < 27; ++q) {
    int coo[3] = {coo_in[0], coo_in[1], coo_in[2]};
    for (int d = 1; d <= 3; ++d) {
      int dd = 3 - d;
      if (bs_get(nb[q], d))
        coo[dd] -= 1;
      if (bs_get(nb[q], -d))
        coo[dd] += 1;
    }
    for (int a = 0; a < 3; ++a)
      coo[a] = ((coo[a] % cart_dims[a]) + cart_dims[a]) % cart_dims[a];
    out[q] = (coo[0] * cart_dims[1] + coo[1]) * cart_dims[2] + coo[2];
  }
}

/* ------------------------------------------------------------------------------------------
 * BrickDecomp::exchange for R ranks living in one address space
 * (ref: include/brick-mpi.h:466-495): ghost run i of rank r  <-  skin run i of rank
 * rank_map[r][ghost[i].neighbor].  rank_maps is [R][27] in allneighbors order.
 * All receives read the pre-exchange state (skin regions are never written by the exchange).
 * ---------------------------------------------------------------------------------------- */
void orc_exchange(const orc_decomp *D, double **storages, int nranks, const int *rank_maps, size_t step) {
  uint64_t nb[27];
  orc_allneighbors3(nb);
  for (int r = 0; r < nranks; ++r)
    for (int q = 0; q < D->nruns; ++q) {
      int ni = -1;
      for (int t = 0; t < 27; ++t)
        if (nb[t] == D->ghost[q].neighbor)
          ni = t;
      int src = rank_maps[r * 27 + ni];
      memcpy(storages[r] + (size_t)D->ghost[q].pos * step, storages[src] + (size_t)D->skin[q].pos * step,
             (size_t)D->ghost[q].len * step * sizeof(double));
    }
}

/* ------------------------------------------------------------------------------------------
 * ZMORT (ref: include/zmort.h:18-105) and strong-scaling ownership
 * (ref: strong/args.cpp:36-56,104-113).
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_zmort_get(uint64_t id, uint64_t dim, uint64_t d) {
  uint64_t oid = id >> d, pos = 0, i = 0;
  while (oid) {
    pos |= (oid & 1ull) << i;
    oid >>= dim;
    ++i;
  }
  return pos;
}

uint64_t orc_zmort_set(uint64_t id, uint64_t dim, uint64_t d, uint64_t p) {
  uint64_t oid = id >> d;
  uint64_t omask = (1ull << dim) - 2ull;
  uint64_t i = d;
  uint64_t z = id & ((1ull << d) - 1);
  while (oid || p) {
    z |= ((p & 1ull) + (oid & omask)) << i;
    oid >>= dim;
    p >>= 1ull;
    i += dim;
  }
  return z;
}

/* out = {mysec_l, mysec_r, sec_split, sec_len(floor), sec_shift} for `rank` of `size` */
void orc_strong_partition(uint64_t allsubs, int size, int rank, uint64_t out[5]) {
  uint64_t sec_shift = allsubs % (uint64_t)size;
  uint64_t sec_len = allsubs / (uint64_t)size + (sec_shift > (uint64_t)rank ? 1 : 0);
  uint64_t l = (uint64_t)rank * sec_len + (sec_shift > (uint64_t)rank ? 0 : sec_shift);
  out[0] = l;
  out[1] = l + sec_len;
  out[2] = sec_shift * ((allsubs + size - 1) / size);
  out[3] = allsubs / (uint64_t)size;
  out[4] = sec_shift;
}

/* neighbour subdomain of z in direction n (bitset), periodic over subdim per axis; returns the
 * neighbour's zmort id and writes owner rank / slot (strong/args.cpp:36-56). */
uint64_t orc_strong_getrank(uint64_t n, uint64_t zid, uint64_t subdim, uint64_t sec_split, uint64_t sec_len,
                            uint64_t sec_shift, int *dst, int *sub) {
  for (int d = 0; d < 3; ++d) {
    long off = 0;
    if (bs_get(n, d + 1))
      off = 1;
    else if (bs_get(n, -1 - d))
      off = -1;
    if (off) {
      long idx = (long)orc_zmort_get(zid, 3, (uint64_t)d);
      idx = (idx + (long)subdim + off) % (long)subdim;
      zid = orc_zmort_set(zid, 3, (uint64_t)d, (uint64_t)idx);
    }
  }
  if (zid < sec_split) {
    *dst = (int)(zid / (sec_len + 1));
    *sub = (int)(zid % (sec_len + 1));
  } else {
    *dst = (int)((zid - sec_shift) / sec_len);
    *sub = (int)((zid - sec_shift) % sec_len);
  }
  return zid;
}

/* thread control for the cpu_baseline leg of bench.py */
#ifdef _OPENMP
#include <omp.h>
void orc_set_threads(int n) { omp_set_num_threads(n); }
int orc_max_threads(void) { return omp_get_max_threads(); }
#else
void orc_set_threads(int n) { (void)n; }
int orc_max_threads(void) { return 1; }
#endif
