// ref_stencil.cpp -- array sweep whose arithmetic is the REFERENCE's own ST_CPU macro
// (stencils/fake.h:37-353).  TEST INFRASTRUCTURE ONLY; compiled once per stencil by oracle/Makefile
// with -DMPI_13PT / -DMPI_25PT / -DMPI_125PT (default: 7-pt) and -DREF_FN=<symbol>.
// Loop nest shape follows weak/main.cpp:116-131 (array_stencil) without the tiling.
#include "stencils/fake.h"

extern "C" int REF_ITER_FN() { return ST_ITER; }

extern "C" void REF_FN(const double *in, double *out, const long ext[3], const long lo[3], const long hi[3]) {
  auto arrIn = (bElem(*)[ext[1]][ext[0]])in;
  auto arrOut = (bElem(*)[ext[1]][ext[0]])out;
  for (long k = lo[2]; k < hi[2]; ++k)
    for (long j = lo[1]; j < hi[1]; ++j)
      for (long i = lo[0]; i < hi[0]; ++i)
        ST_CPU;
}
