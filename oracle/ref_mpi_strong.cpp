// ref_mpi_strong.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, never shipped).
// One `arr_func` of the REFERENCE's strong driver as its `-v` validation runs it (strong/main.cpp:402-446): every subdomain this
// rank owns (strong/args.cpp partition, compiled as it is) is a padded array; exchangeArrAll<3> (include/array-mpi.h:334-) moves
// the ghost zones between subdomains through the rank / slot maps that getrank() (strong/args.cpp:36-56) gives; then ST_ITER
// sweeps of ST_CPU with the reference's bounds.  Input: the oracle's counter hash of the GLOBAL periodic coordinate; output: the
// interior of every subdomain.  Built with -DMPI_13PT (the strong configuration of BASELINE.json) and plain (7-point).
// Usage: mpiexec -np N ref_mpi_strong_<st> -d D -s S      (writes strong_np<N>_z<zmort id>.bin into $REF_OUT_PREFIX)
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

#include "stencils/fake.h"
#include "stencils/stencils.h"
#undef N
#include <array-mpi.h>
#include <brick-mpi.h>
#include "args.h" // strong/args.h: getrank, parseArgs, dom_size, sdom_size, mysec_l, mysec_r

extern "C" void orc_fill_array_global(double *arr, const long ext[3], const long origin[3], const long gdims[3], unsigned long long seed);

int main(int argc, char **argv) {
  int provided;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided);
  int size, rank;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  parseArgs(argc, argv, "ref_mpi_strong");
  const char *pre = getenv("REF_OUT_PREFIX");
  const std::string prefix = pre ? pre : "strong";
  const unsigned long long seed = 21;
  const long S = sdom_size, STRIDE_ = S + 2 * (GZ + PADDING), STRIDEG_ = S + 2 * GZ;
  std::vector<BitSet> neighbors;
  allneighbors(0, 1, 3, neighbors);
  struct Sub {
    ZMORT zmort;
    bElem *a[2];
    std::unordered_map<uint64_t, int> rank_map, id_map;
  };
  std::vector<Sub> subs(mysec_r - mysec_l);
  const long total = STRIDE_ * STRIDE_ * STRIDE_;
  for (unsigned long sec = mysec_l; sec < mysec_r; ++sec) {
    Sub &s = subs[sec - mysec_l];
    s.zmort = ZMORT(sec, 3);
    s.a[0] = (bElem *)aligned_alloc(4096, sizeof(bElem) * total);
    s.a[1] = (bElem *)aligned_alloc(4096, sizeof(bElem) * total);
    long ext[3] = {STRIDE_, STRIDE_, STRIDE_}, origin[3], gd[3] = {(long)dom_size, (long)dom_size, (long)dom_size};
    for (int a = 0; a < 3; ++a) origin[a] = (long)s.zmort(a) * S - (GZ + PADDING); // "+d" = coordinate + 1 (strong/args.cpp:36-48)
    orc_fill_array_global(s.a[0], ext, origin, gd, seed);
    for (long l = 0; l < total; ++l) s.a[1][l] = 0.0;
    for (auto &n : neighbors) { // strong/main.cpp:155-161
      int dst, sub;
      ZMORT z = s.zmort;
      getrank(n, z, dst, sub);
      s.rank_map[n.set] = dst;
      s.id_map[n.set] = sub;
    }
  }
  auto array_stencil = [&](bElem *arrOut_ptr, bElem *arrIn_ptr, long skip) -> void { // strong/main.cpp:406-420
    auto arrIn = (bElem(*)[STRIDE_][STRIDE_])arrIn_ptr;
    auto arrOut = (bElem(*)[STRIDE_][STRIDE_])arrOut_ptr;
    long skip2 = (skip / TILE) * TILE;
    for (long tk = PADDING + skip; tk < PADDING + STRIDEG_ - skip; tk += TILE)
      for (long tj = PADDING + skip; tj < PADDING + STRIDEG_ - skip; tj += TILE)
        for (long ti = PADDING + skip2; ti < PADDING + STRIDEG_ - skip2; ti += TILE)
          for (long k = tk; k < tk + TILE; ++k)
            for (long j = tj; j < tj + TILE; ++j)
              for (long i = ti; i < ti + TILE; ++i) ST_CPU;
  };
  std::vector<ArrExPack> arrpack; // strong/main.cpp:422-430
  for (auto &s : subs) {
    ArrExPack p;
    p.arr = s.a[0];
    p.id = (int)(s.zmort.id - mysec_l);
    p.rank_map = &s.rank_map;
    p.id_map = &s.id_map;
    arrpack.emplace_back(p);
  }
  // arr_func, strong/main.cpp:431-446 (the reference calls the exchange once per subdomain; once moves the same bytes)
  exchangeArrAll<3>(arrpack, MPI_COMM_WORLD, {S, S, S}, {PADDING, PADDING, PADDING}, {GZ, GZ, GZ});
  for (auto &s : subs) array_stencil(s.a[1], s.a[0], 0);
  for (int i = 0; i < ST_ITER / 2 - 1; ++i) {
    for (auto &s : subs) array_stencil(s.a[0], s.a[1], 0);
    for (auto &s : subs) array_stencil(s.a[1], s.a[0], 0);
  }
  for (auto &s : subs) array_stencil(s.a[0], s.a[1], TILE);
  for (auto &s : subs) {
    std::string fn = prefix + "_np" + std::to_string(size) + "_z" + std::to_string(s.zmort.id) + ".bin";
    FILE *f = fopen(fn.c_str(), "wb");
    long hdr[8] = {size, rank, (long)s.zmort.id, (long)s.zmort(0), (long)s.zmort(1), (long)s.zmort(2), (long)dom_size, S};
    fwrite(hdr, sizeof(long), 8, f);
    auto arr = (bElem(*)[STRIDE_][STRIDE_])s.a[0];
    for (long k = PADDING + GZ; k < PADDING + GZ + S; ++k)
      for (long j = PADDING + GZ; j < PADDING + GZ + S; ++j) fwrite(&arr[k][j][PADDING + GZ], sizeof(bElem), S, f);
    fclose(f);
  }
  MPI_Finalize();
  return 0;
}
