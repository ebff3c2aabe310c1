// Host-only probe of include/bitset.h, zmort.h, brick-mpi.h: prints the tables a BrickDecomp<3,8,8,8> holds, for
// tests/test_cpp_headers.py to compare with the Python mirror (which is pinned against fixtures made by the reference).
#define BRK_NO_MPI
#include <brick-mpi.h>
#include <zmort.h>

#include <cassert>
#include <iostream>
#include <sstream>

int main(int argc, char **argv) {
  unsigned dx = argc > 1 ? atoi(argv[1]) : 32, dy = argc > 2 ? atoi(argv[2]) : 24, dz = argc > 3 ? atoi(argv[3]) : 16;
  // BitSet semantics
  BitSet a = {1, -2}, b = {-1, 2, 3}, na = {-1, 2}, one = {1}, m2 = {-2}, p13 = {1, -3};
  assert((!a).set == na.set && a.size() == 2 && a.get(-2) && !a.get(2));
  assert(((a | b).size() == 5) && (a & b).set == 0 && (one <= a) && (a >= m2) && ((a ^ a).set == 0));
  assert(BitSet::to_set(3) == 8 && BitSet::to_set(-1) == (1ull << 32));
  std::ostringstream os;
  os << p13;
  assert(os.str() == "{1+3-}");
  // ZMORT: build from coordinates, read back, replace
  ZMORT z;
  z = z[5][3][6];                       // coordinates (6, 3, 5) for d = 0, 1, 2
  assert(z.dim == 3 && z(0) == 6 && z(1) == 3 && z(2) == 5);
  assert((unsigned long)z == brk_zmort_set(brk_zmort_set(brk_zmort_set(0, 0, 6), 1, 3), 2, 5));
  ZMORT y = z.set(1, 9);
  assert(y(0) == 6 && y(1) == 9 && y(2) == 5 && brk_zmort_get(y, 1) == 9);
  // BrickDecomp tables
  BrickDecomp<3, 8, 8, 8> d({dx, dy, dz}, 8);
  d.initialize(skin3d_good);
  std::cout << "nbricks " << d.nbricks() << "\nsep " << d.sep_pos[0] << " " << d.sep_pos[1] << " " << d.sep_pos[2] << "\ntdims " << d.t_dims[0]
            << " " << d.t_dims[1] << " " << d.t_dims[2] << "\n";
  for (size_t q = 0; q < d.ghost.size(); ++q)
    std::cout << "run " << q << " " << d.ghost[q].neighbor.set << " " << d.ghost[q].pos << " " << d.ghost[q].len << " " << d.ghost[q].skin_st
              << " " << d.ghost[q].skin_ed << " " << d.skin[q].neighbor.set << " " << d.skin[q].pos << " " << d.skin[q].len << "\n";
  unsigned long sum = 0;
  for (unsigned k = 0; k < d.t_dims[2]; ++k)
    for (unsigned j = 0; j < d.t_dims[1]; ++j)
      for (unsigned i = 0; i < d.t_dims[0]; ++i) sum = sum * 31 + d[k][j][i];
  std::cout << "gridhash " << sum << "\n";
  BrickInfo<3> info = d.getBrickInfo();
  unsigned long ah = 0;
  for (unsigned bq = 0; bq < info.nbricks; ++bq)
    for (int s = 0; s < 27; ++s) ah = ah * 31 + info.adj[bq][s];
  std::cout << "adjhash " << ah << "\n";
  const int cart[3] = {2, 2, 2}, coo[3] = {1, 0, 1};
  populate_cart(d, cart, coo);
  std::vector<BitSet> nb;
  allneighbors(0, 1, 3, nb);
  std::cout << "ranks";
  for (auto n : nb) std::cout << " " << d.rank_map[n.set];
  std::cout << "\n";
  free(info.adj);
  return 0;
}
