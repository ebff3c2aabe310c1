// ref_cpu_sweep.cpp -- the REFERENCE's CPU array sweep as a timed leg: its own loop nest macro `_TILEFOR`
// (stencils/stencils.h:30-38: OpenMP parallel for collapse(2) over 8^3 tiles, omp simd on i) around its own arithmetic
// macro `ST_CPU` (stencils/fake.h:345-349, 7-point: MPI_ALPHA / MPI_BETA), both included where they lie under
// /root/reference.  TEST / BASELINE INFRASTRUCTURE ONLY (bench.py's cpu_baseline_reference leg); built by `make -C oracle ref`
// into oracle/_ref/libbrickref.so.  The only thing that is ours is the problem size: the reference fixes N by editing
// stencils.h (#define N 256); here N is redefined after the include (REF_N, default 512 = BASELINE config 2).
#include "stencils/stencils.h"
#include "stencils/fake.h"

#ifndef REF_N
#define REF_N 512
#endif
#undef N
#define N REF_N

extern "C" long ref_cpu_sweep_stride() { return STRIDE; }    // array extent per axis: N + 2 (GZ + PADDING)
extern "C" long ref_cpu_sweep_points() { return (long)STRIDEG * STRIDEG * STRIDEG; }   // points one call updates
extern "C" int ref_cpu_sweep_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}

// one array sweep, the shape of d3pt7's arr_func (stencils/3axis.cpp:31-37) with the weak driver's arithmetic
extern "C" void ref_cpu_sweep_mpi7(const bElem *in_ptr, bElem *out_ptr) {
  auto arrIn = (bElem(*)[STRIDE][STRIDE])in_ptr;
  auto arrOut = (bElem(*)[STRIDE][STRIDE])out_ptr;
  _TILEFOR ST_CPU;
}

// the same loop nest around the arithmetic of d3pt7's arr_func itself (stencils/3axis.cpp:31-37: seven coefficients, its term
// order) -- the stencil of the GPU headline (7pt.py).  The expression sits inside a lambda of d3pt7() there (that function also
// needs generated code, so it cannot be compiled from where it lies); it is restated here under the reference's own `_TILEFOR`.
extern "C" void ref_cpu_sweep_7pt(const bElem *in_ptr, bElem *out_ptr, const bElem *coeff) {
  auto arr_in = (bElem(*)[STRIDE][STRIDE])in_ptr;
  auto arr_out = (bElem(*)[STRIDE][STRIDE])out_ptr;
  _TILEFOR arr_out[k][j][i] = coeff[5] * arr_in[k + 1][j][i] + coeff[6] * arr_in[k - 1][j][i] + coeff[3] * arr_in[k][j + 1][i] +
                              coeff[4] * arr_in[k][j - 1][i] + coeff[1] * arr_in[k][j][i + 1] + coeff[2] * arr_in[k][j][i - 1] +
                              coeff[0] * arr_in[k][j][i];
}
