// ref_mpi_harness.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, never shipped).
// Runs the REFERENCE's own MPI-side integer bookkeeping under mpirun and prints its tables, so that
// tests/golden/populate_*.npz and strong_getrank_*.npz are reference outputs, not restatements:
//   populate()            /root/reference/include/brick-mpi.h:730-753  (compiled from the header, in place)
//   getrank(), partition  /root/reference/strong/args.cpp:36-56,104-113 (the file itself is compiled and linked)
//   MPI_Dims_create       as called by weak/args.cpp:105
// Build: make -C oracle refmpi   (conda MPICH's mpicxx; the reference sources are read where they lie).
// Usage: mpirun -np N ref_mpi_tables populate             -> one line per rank
//        mpirun -np N ref_mpi_tables strong -d D -s S     -> one line per (rank, subdomain, neighbour)
#include <mpi.h>

#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "brick-mpi.h"
#include "args.h" // strong/args.h

static void print_in_rank_order(const std::string &mine, int rank, int size) {
  // gather variable-length text on rank 0 and print it in rank order
  int len = (int)mine.size();
  std::vector<int> lens(size), offs(size);
  MPI_Gather(&len, 1, MPI_INT, lens.data(), 1, MPI_INT, 0, MPI_COMM_WORLD);
  int total = 0;
  if (rank == 0)
    for (int r = 0; r < size; ++r) offs[r] = total, total += lens[r];
  std::vector<char> all(rank == 0 ? total : 1);
  MPI_Gatherv(mine.data(), len, MPI_CHAR, all.data(), lens.data(), offs.data(), MPI_CHAR, 0, MPI_COMM_WORLD);
  if (rank == 0) fwrite(all.data(), 1, total, stdout);
}

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  int rank, size;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  const std::string what = argc > 1 ? argv[1] : "";
  std::ostringstream os;
  if (what == "populate") {
    // exactly the set-up of weak/main.cpp:121-141: MPI_Dims_create, periodic cart communicator, populate(cart, bDecomp, 0, 1, coo)
    int dims[3] = {0, 0, 0}, periods[3] = {1, 1, 1}, coo[3];
    MPI_Dims_create(size, 3, dims);
    MPI_Comm cart;
    MPI_Cart_create(MPI_COMM_WORLD, 3, dims, periods, 0, &cart);
    int crank;
    MPI_Comm_rank(cart, &crank);
    MPI_Cart_coords(cart, crank, 3, coo);
    BrickDecomp<3, 8, 8, 8> bDecomp({16, 16, 16}, 8);
    bDecomp.comm = cart;
    populate(cart, bDecomp, 0, 1, coo);
    std::vector<BitSet> nb;
    allneighbors(0, 1, 3, nb);
    os << "P " << size << " " << crank << " dims " << dims[0] << " " << dims[1] << " " << dims[2] << " coo " << coo[0] << " " << coo[1]
       << " " << coo[2] << " map";
    for (auto &n : nb) os << " " << bDecomp.rank_map[n.set];
    os << "\n";
    print_in_rank_order(os.str(), rank, size);
  } else if (what == "strong") {
    parseArgs(argc - 1, argv + 1, "ref_mpi_tables"); // sets mysec_l / mysec_r and the file-static split used by getrank
    std::vector<BitSet> nb;
    allneighbors(0, 1, 3, nb);
    os << "S " << size << " " << rank << " d " << dom_size << " s " << sdom_size << " range " << mysec_l << " " << mysec_r << "\n";
    for (unsigned long z = mysec_l; z < mysec_r; ++z)
      for (auto &n : nb) {
        ZMORT zm(z, 3);
        int dst = -1, sub = -1;
        getrank(n, zm, dst, sub);
        os << "G " << rank << " " << z << " " << n.set << " " << zm.id << " " << dst << " " << sub << "\n";
      }
    print_in_rank_order(os.str(), rank, size);
  } else if (what == "dims") {
    if (rank == 0)
      for (int n = 1; n <= 64; ++n) {
        int dims[3] = {0, 0, 0};
        MPI_Dims_create(n, 3, dims);
        printf("D %d %d %d %d\n", n, dims[0], dims[1], dims[2]);
      }
  }
  MPI_Finalize();
  return 0;
}
