// Host-side probe of include/brick-mpi.h under a real MPI launcher (tests/test_cpp_headers.py, mpiexec -np R): periodic cart grid,
// populate(cart, ...), BrickDecomp in one of the reference's section orders, and the two host exchanges of the reference --
// bDecomp.exchange(bStorage) (Isend/Irecv per run, include/brick-mpi.h:466-495) and bDecomp.exchange(bStorage, win) (MPI_Get between
// fences, :688-712).  Every element carries (rank, brick, element); after an exchange each ghost run must hold exactly the
// same-index skin run of the rank behind it.
#include <brick-mpi.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  int size = 1, wrank = 0;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &wrank);
  const char *order = argc > 1 ? argv[1] : "good";
  int dims[3] = {0, 0, 0}, periods[3] = {1, 1, 1}, coo[3];
  MPI_Dims_create(size, 3, dims);
  MPI_Comm cart;
  MPI_Cart_create(MPI_COMM_WORLD, 3, dims, periods, 1, &cart);
  int rank = 0;
  MPI_Comm_rank(cart, &rank);
  MPI_Cart_coords(cart, rank, 3, coo);
  BrickDecomp<3, 8, 8, 8> d({32, 16, 24}, 8);
  d.initialize(!strcmp(order, "bad") ? skin3d_bad : (!strcmp(order, "normal") ? skin3d_normal : skin3d_good));
  populate(cart, d, 0, 1, coo);
  BrickInfo<3> info = d.getBrickInfo();
  BrickStorage st = info.allocate(512);
  const size_t total = (size_t)st.chunks * st.step;
  auto fill = [&]() {
    for (size_t e = 0; e < total; ++e) st.dat.get()[e] = (double)rank * 1e9 + (double)e;
  };
  auto check = [&](const char *what) {
    size_t bad = 0, n = 0;
    for (size_t q = 0; q < d.ghost.size(); ++q) {
      const int src = d.rank_map[d.ghost[q].neighbor.set];
      for (size_t e = 0; e < (size_t)d.ghost[q].len * 512; ++e, ++n)
        bad += st.dat.get()[(size_t)d.ghost[q].pos * 512 + e] != (double)src * 1e9 + (double)((size_t)d.skin[q].pos * 512 + e);
    }
    size_t gbad = 0;
    MPI_Allreduce(&bad, &gbad, 1, MPI_UNSIGNED_LONG, MPI_SUM, cart);
    if (rank == 0) printf("%s: %s (%zu ghost elements per rank, %zu runs)\n", what, gbad ? "MISMATCH" : "ghosts match", n, d.ghost.size());
    return gbad == 0;
  };
  bool ok = true;
  fill();
  d.exchange(st);
  ok &= check("exchange(bStorage)");
  fill();
  MPI_Win win;
  MPI_Win_create(st.dat.get(), (MPI_Aint)(total * sizeof(bElem)), 1, MPI_INFO_NULL, cart, &win);
  d.exchange(st, win);
  ok &= check("exchange(bStorage, win)");
  MPI_Win_free(&win);
  free(info.adj);
  MPI_Comm_free(&cart);
  MPI_Finalize();
  return ok ? 0 : 1;
}
