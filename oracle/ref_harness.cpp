// ref_harness.cpp -- drives the REFERENCE's own headers so that the oracle can be pinned.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile (target `ref`) into oracle/_ref/libbrickref.so
// from the sources where they lie under /root/reference; nothing from the reference is copied into
// this repository.  This file contains only call sites: every data-structure operation below is
// executed by reference code (include/brick.h, bricksetup.h, brickcompare.h, brick-mpi.h,
// bitset.h, zmort.h, src/brick-mpi.cpp, src/brickcompare.cpp).
//
// The code-generated "Trans" path (brick(...) macro -> codegen/vecscatter) is NOT built: it needs
// generated code and the reference's CMake machinery, so it is treated as unbuildable here.  The
// naive accessor path exercised below is what the reference itself validates "Trans" against
// (stencils/3axis.cpp:79-86).
#include <cstdint>
#include <cstring>
#include <memory>
#include <algorithm>
#include <vector>

#include "brick.h"
#include "bricksetup.h"
#include "brickcompare.h"
#include "bitset.h"
#include "zmort.h"
#include "brick-mpi.h"

typedef Brick<Dim<8, 8, 8>, Dim<8, 8>> Brick3D; // stencils/hipvfold.h:5-6 fold

static BrickStorage wrap_storage(double *dat, long chunks, size_t step) {
  BrickStorage s;
  s.chunks = chunks;
  s.step = step;
  s.dat = std::shared_ptr<bElem>(dat, [](bElem *) {});
  return s;
}

static std::vector<long> v3(const long *p) { return std::vector<long>{p[0], p[1], p[2]}; }

extern "C" {

// init_grid<3> (bricksetup.h:73-90)
void ref_init_grid(const long dims[3], uint32_t *grid_out, uint32_t *adj_out) {
  unsigned *grid_ptr;
  BrickInfo<3> bInfo = init_grid<3>(grid_ptr, v3(dims));
  long size = dims[0] * dims[1] * dims[2];
  memcpy(grid_out, grid_ptr, sizeof(unsigned) * size);
  memcpy(adj_out, bInfo.adj, sizeof(unsigned) * 27 * size);
  free(grid_ptr);
  free(bInfo.adj);
}

// copyToBrick<3> (bricksetup.h:172-200); boff = element offset of the field inside a chunk
void ref_copy_to_brick(const long dimlist[3], const long padding[3], const long ghost[3], double *arr,
                       uint32_t *grid, uint32_t *adj, uint32_t nbricks, double *bdat, size_t step, unsigned boff) {
  BrickInfo<3> bInfo(0);
  free(bInfo.adj);
  bInfo.adj = (BrickInfo<3>::adjlist)adj;
  bInfo.nbricks = nbricks;
  BrickStorage st = wrap_storage(bdat, nbricks, step);
  Brick3D b(&bInfo, st, boff);
  copyToBrick<3>(v3(dimlist), v3(padding), v3(ghost), arr, grid, b);
}

// copyFromBrick<3> (bricksetup.h:213-221)
void ref_copy_from_brick(const long dimlist[3], const long padding[3], const long ghost[3], double *arr,
                         uint32_t *grid, uint32_t *adj, uint32_t nbricks, double *bdat, size_t step, unsigned boff) {
  BrickInfo<3> bInfo(0);
  free(bInfo.adj);
  bInfo.adj = (BrickInfo<3>::adjlist)adj;
  bInfo.nbricks = nbricks;
  BrickStorage st = wrap_storage(bdat, nbricks, step);
  Brick3D b(&bInfo, st, boff);
  copyFromBrick<3>(v3(dimlist), v3(padding), v3(ghost), arr, grid, b);
}

// compareBrick<3> (brickcompare.h:30-57), tolerance = BRICK_TOLERANCE (cmpconst.h:9)
int ref_compare_brick(const long dimlist[3], const long padding[3], const long ghost[3], double *arr,
                      uint32_t *grid, uint32_t *adj, uint32_t nbricks, double *bdat, size_t step, unsigned boff) {
  BrickInfo<3> bInfo(0);
  free(bInfo.adj);
  bInfo.adj = (BrickInfo<3>::adjlist)adj;
  bInfo.nbricks = nbricks;
  BrickStorage st = wrap_storage(bdat, nbricks, step);
  Brick3D b(&bInfo, st, boff);
  return compareBrick<3>(v3(dimlist), v3(padding), v3(ghost), arr, grid, b) ? 1 : 0;
}

// _BrickAccessor (brick.h:214-327)
double ref_brick_at(uint32_t *adj, uint32_t nbricks, double *bdat, size_t step, unsigned boff, uint32_t b, int k,
                    int j, int i) {
  BrickInfo<3> bInfo(0);
  free(bInfo.adj);
  bInfo.adj = (BrickInfo<3>::adjlist)adj;
  bInfo.nbricks = nbricks;
  BrickStorage st = wrap_storage(bdat, nbricks, step);
  Brick3D br(&bInfo, st, boff);
  return br[b][k][j][i];
}

// Naive brick sweep through the reference accessor with a tap table {dk,dj,di,coeff index},
// accumulated left to right exactly like the textual sums of stencils/3axis.cpp:60-63 (7-pt)
// and :214-233 (27-pt, including the coeff[16] quirk at :225).
void ref_brick_taps(uint32_t *adj, uint32_t nbricks, double *bin, double *bout, size_t step, const uint32_t *ids,
                    long n, const int *taps /*[ntaps][4]*/, int ntaps, const double *coeff) {
  BrickInfo<3> bInfo(0);
  free(bInfo.adj);
  bInfo.adj = (BrickInfo<3>::adjlist)adj;
  bInfo.nbricks = nbricks;
  BrickStorage sti = wrap_storage(bin, nbricks, step);
  BrickStorage sto = wrap_storage(bout, nbricks, step);
  Brick3D bIn(&bInfo, sti, 0), bOut(&bInfo, sto, 0);
  for (long q = 0; q < n; ++q) {
    unsigned b = ids[q];
    for (long k = 0; k < 8; ++k)
      for (long j = 0; j < 8; ++j)
        for (long i = 0; i < 8; ++i) {
          double acc = coeff[taps[3]] * bIn[b][k + taps[0]][j + taps[1]][i + taps[2]];
          for (int t = 1; t < ntaps; ++t)
            acc = acc + coeff[taps[t * 4 + 3]] * bIn[b][k + taps[t * 4]][j + taps[t * 4 + 1]][i + taps[t * 4 + 2]];
          bOut[b][k][j][i] = acc;
        }
  }
}

// d3cond brick form (stencils/3axis.cpp:137-146) through the reference accessor: taps clipped at zero with std::max,
// 7-pt term order, absolute value of the sum.
void ref_brick_cond(uint32_t *adj, uint32_t nbricks, double *bin, double *bout, size_t step, const uint32_t *ids, long n,
                    const double *coeff) {
  using std::max;
  BrickInfo<3> bInfo(0);
  free(bInfo.adj);
  bInfo.adj = (BrickInfo<3>::adjlist)adj;
  bInfo.nbricks = nbricks;
  BrickStorage sti = wrap_storage(bin, nbricks, step);
  BrickStorage sto = wrap_storage(bout, nbricks, step);
  Brick3D bIn(&bInfo, sti, 0), bOut(&bInfo, sto, 0);
  for (long q = 0; q < n; ++q) {
    unsigned b = ids[q];
    for (long k = 0; k < 8; ++k)
      for (long j = 0; j < 8; ++j)
        for (long i = 0; i < 8; ++i) {
          bElem partial = coeff[5] * max(bIn[b][k + 1][j][i], 0.0) + coeff[6] * max(bIn[b][k - 1][j][i], 0.0) +
                          coeff[3] * max(bIn[b][k][j + 1][i], 0.0) + coeff[4] * max(bIn[b][k][j - 1][i], 0.0) +
                          coeff[1] * max(bIn[b][k][j][i + 1], 0.0) + coeff[2] * max(bIn[b][k][j][i - 1], 0.0) +
                          coeff[0] * max(bIn[b][k][j][i], 0.0);
          bOut[b][k][j][i] = partial > 0 ? partial : -partial;
        }
  }
}

// ---- BitSet / neighbours / skin order -------------------------------------------------------
uint64_t ref_bitset_of(const int *e, int n) {
  BitSet s;
  for (int q = 0; q < n; ++q)
    s.flip(e[q]);
  return s.set;
}
uint64_t ref_bitset_neg(uint64_t s) { return (!BitSet(s)).set; }
int ref_bitset_le(uint64_t a, uint64_t b) { return BitSet(a) <= BitSet(b) ? 1 : 0; }
long ref_bitset_size(uint64_t a) { return BitSet(a).size(); }
void ref_allneighbors3(uint64_t out[27]) {
  std::vector<BitSet> nb;
  allneighbors(0, 1, 3, nb);
  for (int q = 0; q < 27; ++q)
    out[q] = nb[q].set;
}
int ref_skin3d_good(uint64_t *out) {
  for (size_t q = 0; q < skin3d_good.size(); ++q)
    out[q] = skin3d_good[q].set;
  return (int)skin3d_good.size();
}

// ---- BrickDecomp<3,8,8,8> (brick-mpi.h:178-460) ----------------------------------------------
typedef BrickDecomp<3, 8, 8, 8> Decomp;
struct RefDecomp {
  Decomp *d;
  std::vector<unsigned> tdims;
};

// the reference's three section orders (src/brick-mpi.cpp:25-136): which = 0 good, 1 normal, 2 bad
int ref_skin3d(int which, uint64_t *out) {
  const std::vector<BitSet> &v = which == 1 ? skin3d_normal : (which == 2 ? skin3d_bad : skin3d_good);
  for (size_t q = 0; q < v.size(); ++q) out[q] = v[q].set;
  return (int)v.size();
}
void *ref_decomp_build_order(const uint32_t dims_cells[3], uint32_t ghost_cells, int which);
void *ref_decomp_build(const uint32_t dims_cells[3], uint32_t ghost_cells) { return ref_decomp_build_order(dims_cells, ghost_cells, 0); }
void *ref_decomp_build_order(const uint32_t dims_cells[3], uint32_t ghost_cells, int which) {
  auto r = new RefDecomp;
  r->d = new Decomp({dims_cells[0], dims_cells[1], dims_cells[2]}, ghost_cells);
  r->d->initialize(which == 1 ? skin3d_normal : (which == 2 ? skin3d_bad : skin3d_good));
  for (int a = 0; a < 3; ++a)
    r->tdims.push_back((dims_cells[a] + 2 * ghost_cells) / 8);
  return r;
}
void ref_decomp_free(void *h) {
  auto r = (RefDecomp *)h;
  delete r->d;
  delete r;
}
void ref_decomp_info(void *h, uint32_t info[10]) {
  auto r = (RefDecomp *)h;
  info[0] = r->d->sep_pos[0]; info[1] = r->d->sep_pos[1]; info[2] = r->d->sep_pos[2];
  info[3] = r->d->getBrickInfo().nbricks;
  info[4] = r->tdims[0] * r->tdims[1] * r->tdims[2];
  info[5] = (uint32_t)r->d->ghost.size();
  info[6] = (uint32_t)r->d->skinlist.size();
  info[7] = r->tdims[0]; info[8] = r->tdims[1]; info[9] = r->tdims[2];
}
void ref_decomp_grid(void *h, uint32_t *out) {
  auto r = (RefDecomp *)h;
  for (unsigned k = 0; k < r->tdims[2]; ++k)
    for (unsigned j = 0; j < r->tdims[1]; ++j)
      for (unsigned i = 0; i < r->tdims[0]; ++i)
        out[(k * r->tdims[1] + j) * r->tdims[0] + i] = (*r->d)[k][j][i];
}
void ref_decomp_adj(void *h, uint32_t *out) {
  auto r = (RefDecomp *)h;
  BrickInfo<3> bi = r->d->getBrickInfo();
  memcpy(out, bi.adj, sizeof(unsigned) * 27 * (size_t)bi.nbricks);
}
void ref_decomp_skin_size(void *h, int64_t *out) {
  auto r = (RefDecomp *)h;
  for (size_t l = 0; l < r->d->skin_size.size(); ++l)
    out[l] = r->d->skin_size[l];
}
void ref_decomp_runs(void *h, int which, uint64_t *out) {
  auto r = (RefDecomp *)h;
  auto &v = which ? r->d->skin : r->d->ghost;
  for (size_t q = 0; q < v.size(); ++q) {
    out[q * 7 + 0] = v[q].neighbor.set; out[q * 7 + 1] = v[q].skin_st; out[q * 7 + 2] = v[q].skin_ed;
    out[q * 7 + 3] = v[q].pos; out[q * 7 + 4] = v[q].len; out[q * 7 + 5] = v[q].first_pad; out[q * 7 + 6] = v[q].last_pad;
  }
}

// ---- ZMORT (zmort.h:18-105) --------------------------------------------------------------------
uint64_t ref_zmort_get(uint64_t id, uint64_t dim, uint64_t d) {
  ZMORT z(id, dim);
  return z(d);
}
uint64_t ref_zmort_set(uint64_t id, uint64_t dim, uint64_t d, uint64_t p) {
  ZMORT z(id, dim);
  return z.set(d, p).id;
}

} // extern "C"
