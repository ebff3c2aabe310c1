// ref_mpi_weak.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, never shipped).
// One `arr_func` of the REFERENCE's weak driver (weak/main.cpp:134-168): exchangeArr<3> (include/array-mpi.h:146-, src/array-mpi.cpp)
// on the periodic cart communicator that populate() mapped, then ST_ITER sweeps of the ST_CPU expression of stencils/fake.h with
// the reference's loop bounds (weak/main.cpp:116-131; the last sweep skips the outer TILE ring) -- compiled from the
// reference's headers where they lie, run under mpiexec.  The input is the oracle's counter hash of the GLOBAL periodic
// coordinate (so a test can rebuild it), the output is each rank's interior after the cycle.  One binary per stencil
// (-DMPI_13PT / -DMPI_25PT / -DMPI_125PT select the expression, as in the reference's build).
// Usage: mpiexec -np N ref_mpi_weak_<st> <cells per rank and axis> <seed> <outfile prefix>
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "stencils/fake.h"     // ST_CPU, ST_ITER, MPI_ALPHA ... (reference)
#include "stencils/stencils.h" // TILE, GZ, PADDING
#undef N
#include <array-mpi.h>
#include <brick-mpi.h>

extern "C" void orc_fill_array_global(double *arr, const long ext[3], const long origin[3], const long gdims[3], unsigned long long seed);
int MPI_ITER = 1;

int main(int argc, char **argv) {
  int provided;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided);
  int size, wrank;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &wrank);
  const long N = argc > 1 ? atol(argv[1]) : 16;
  const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 42;
  const std::string prefix = argc > 3 ? argv[3] : "weak";
  int dims[3] = {0, 0, 0}, prd[3] = {1, 1, 1}, coo[3];
  MPI_Dims_create(size, 3, dims); // weak/args.cpp:105
  MPI_Comm cart;
  MPI_Cart_create(MPI_COMM_WORLD, 3, dims, prd, 0, &cart); // weak/args.cpp:106-108
  int rank;
  MPI_Comm_rank(cart, &rank);
  MPI_Cart_coords(cart, rank, 3, coo);
  std::vector<long> stride(3), strideg(3), dom_size = {N, N, N};
  for (int i = 0; i < 3; ++i) {
    stride[i] = dom_size[i] + 2 * TILE + 2 * GZ;
    strideg[i] = dom_size[i] + 2 * TILE;
  }
  BrickDecomp<3, BDIM> bDecomp({(unsigned)N, (unsigned)N, (unsigned)N}, GZ);
  bDecomp.comm = cart;
  populate(cart, bDecomp, 0, 1, coo); // weak/main.cpp:67
  // global position of this rank's subdomain under populate()'s orientation: axis d lives in coo[2-d], "+d" neighbour at coo-1
  long origin[3], gdims[3], ext[3];
  for (int a = 0; a < 3; ++a) {
    gdims[a] = (long)dims[2 - a] * N;
    origin[a] = (long)((dims[2 - a] - coo[2 - a]) % dims[2 - a]) * N - (PADDING + GZ);
    ext[a] = stride[a];
  }
  long total = stride[0] * stride[1] * stride[2];
  bElem *in_ptr = (bElem *)aligned_alloc(4096, sizeof(bElem) * total);
  bElem *out_ptr = (bElem *)aligned_alloc(4096, sizeof(bElem) * total);
  orc_fill_array_global(in_ptr, ext, origin, gdims, seed);
  for (long l = 0; l < total; ++l) out_ptr[l] = 0.0;

  auto array_stencil = [&](bElem *arrOut_ptr, bElem *arrIn_ptr, long skip) -> void { // weak/main.cpp:116-131
    auto arrIn = (bElem(*)[stride[1]][stride[0]])arrIn_ptr;
    auto arrOut = (bElem(*)[stride[1]][stride[0]])arrOut_ptr;
    long skip2 = (skip / TILE) * TILE;
    long s20 = PADDING + skip, s21 = PADDING + strideg[2];
    long s10 = PADDING + skip, s11 = PADDING + strideg[1];
    for (long tk = s20; tk < s21; tk += TILE)
      for (long tj = s10; tj < s11; tj += TILE)
        for (long ti = PADDING + skip2; ti < PADDING + strideg[0] - skip2; ti += TILE)
          for (long k = tk; k < tk + TILE; ++k)
            for (long j = tj; j < tj + TILE; ++j)
              for (long i = ti; i < ti + TILE; ++i) ST_CPU;
  };
  // arr_func, weak/main.cpp:138-168 (non-49pt branch)
  exchangeArr<3>(in_ptr, cart, bDecomp.rank_map, {dom_size[0], dom_size[1], dom_size[2]}, {PADDING, PADDING, PADDING}, {GZ, GZ, GZ});
  array_stencil(out_ptr, in_ptr, 0);
  for (int i = 0; i < ST_ITER / 2 - 1; ++i) {
    array_stencil(in_ptr, out_ptr, 0);
    array_stencil(out_ptr, in_ptr, 0);
  }
  array_stencil(in_ptr, out_ptr, TILE);

  // interior of in_ptr (where the cycle leaves its result), k slowest
  std::string fn = prefix + "_np" + std::to_string(size) + "_r" + std::to_string(rank) + ".bin";
  FILE *f = fopen(fn.c_str(), "wb");
  long hdr[8] = {size, rank, dims[0], dims[1], dims[2], coo[0], coo[1], coo[2]};
  fwrite(hdr, sizeof(long), 8, f);
  auto arr = (bElem(*)[stride[1]][stride[0]])in_ptr;
  for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
    for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j) fwrite(&arr[k][j][PADDING + GZ], sizeof(bElem), N, f);
  fclose(f);
  MPI_Finalize();
  return 0;
}
