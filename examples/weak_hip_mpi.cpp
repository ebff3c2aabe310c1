// examples/weak_hip_mpi.cpp -- the reference's weak GPU driver as it is started: ONE PROCESS PER RANK under mpiexec
// (weak/main.cu:46-60,100-300): MPI_Init -> periodic MPI_Cart_create over MPI_Dims_create(size) -> BrickDecomp + populate(cart, ...)
// -> one GPU per rank (hipSetDevice(local rank)) -> an RCCL communicator over the cart's ranks (brkCommFromMPI) ->
// ExchangeView on the DEVICE storage (GPU-resident pack -> ncclSend/ncclRecv -> unpack inside libbrk.so) -> brick_func =
// exchange + ST_ITER sweeps, timed with time_mpi, and the reference's statistics lines (weak/main.cpp:271-289):
//   Bri / calc / call / wait / MPI size / perf ... GStencil/s / Total of N parts,   "result mismatch!" when validation fails.
// Validation (the reference compares the bricks with its array code, weak/main.cpp:292-294): one brick_func on a field that is a
// hash of the GLOBAL cell coordinate, against a host sweep of the same bricks through the Brick accessor (naive form).
//   build: make -C examples weak_hip_mpi   (only where <mpi.h> exists)
//   run:   mpiexec -np R examples/weak_hip_mpi [-s cells per axis per rank, default 64] [-I iterations] [-c]
//          -c: runs whose peer is this rank also travel through the communicator (RCCL send/recv to self; one-GPU boxes)
#include <mpi.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

#include "brick-hip.h"
#include "brick-mpi.h"
#include "bricksetup.h"
#include "bricktime.h"
#include "brk.h"

int MPI_ITER = 100; // weak/main.cu: GPU default

typedef Brick<Dim<8, 8, 8>, Dim<8, 8>> Brick3D;

static double cell_value(long gi, long gj, long gk) { // U[0,1) from the global periodic cell coordinate
  unsigned long long x = ((unsigned long long)gk * 1000003ull + (unsigned long long)gj) * 1000003ull + (unsigned long long)gi + 42ull;
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char **argv) {
  int provided = 0;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided);
  int size = 1, wrank = 0;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &wrank);
  unsigned N = 64;
  unsigned xflags = 0;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "-s") && a + 1 < argc) N = (unsigned)atoi(argv[++a]);
    else if (!strcmp(argv[a], "-I") && a + 1 < argc) MPI_ITER = atoi(argv[++a]);
    else if (!strcmp(argv[a], "-c")) xflags = BRK_XCHG_SELF_VIA_COMM;
  }
  int rc = 0;
  try {
    // ---- the cart communicator and this rank's place in it (weak/args.cpp:100-112, weak/main.cu:100-112)
    int dims[3] = {0, 0, 0}, periods[3] = {1, 1, 1}, coo[3];
    MPI_Dims_create(size, 3, dims);
    MPI_Comm cart;
    MPI_Cart_create(MPI_COMM_WORLD, 3, dims, periods, /*reorder*/ 1, &cart);
    int rank = 0;
    MPI_Comm_rank(cart, &rank);
    MPI_Cart_coords(cart, rank, 3, coo);
    // ---- one GPU per rank: ranks of one node take consecutive devices
    MPI_Comm node;
    MPI_Comm_split_type(cart, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
    int local = 0, ndev = 0;
    MPI_Comm_rank(node, &local);
    gpuCheck(hipGetDeviceCount(&ndev));
    if (ndev == 0) throw std::runtime_error("no GPU visible");
    if (size > 1 && local >= ndev) throw std::runtime_error("more ranks on this node than GPUs: one rank per GPU is required under RCCL");
    gpuCheck(hipSetDevice(local % ndev));

    BrickDecomp<3, 8, 8, 8> bDecomp({N, N, N}, 8);
    bDecomp.initialize(skin3d_good);
    populate(cart, bDecomp, 0, 1, coo);
    bDecomp.rccl_comm = brkCommFromMPI(cart); // rank 0's id by MPI_Bcast; the communicator's ranks ARE the cart's ranks

    BrickInfo<3> bInfo = bDecomp.getBrickInfo();
    BrickInfo<3> bInfo_dev = movBrickInfo(bInfo, hipMemcpyHostToDevice);
    // written once: the library may keep compact tables next to it (brk_adjacency_register); released, then freed, when this
    // block is left -- also by an exception
    brkAdjacencyScope<3> adjacency(bInfo_dev);
    const size_t bsize = 512;
    auto bStorage = bInfo.allocate(bsize), bStorageOut = bInfo.allocate(bsize);
    const size_t total = (size_t)bStorage.chunks * bStorage.step;
    // ---- the field: hash of the GLOBAL periodic coordinate, interior only (ghosts arrive by the exchange).  populate()'s
    // orientation (include/brick-mpi.h:741-750): axis d of the domain lives in coo[2-d] and positions decrease with coo.
    long org[3], gdim[3];
    for (int d = 0; d < 3; ++d) {
      gdim[d] = (long)dims[2 - d] * N;
      org[d] = (long)((dims[2 - d] - coo[2 - d]) % dims[2 - d]) * N;
    }
    const unsigned T = N / 8 + 2;
    Brick3D hIn(&bInfo, bStorage, 0), hOut(&bInfo, bStorageOut, 0);
    for (size_t e = 0; e < total; ++e) bStorage.dat.get()[e] = -1000.0, bStorageOut.dat.get()[e] = 0.0;
    for (unsigned e = 0; e < bsize; ++e) bStorage.dat.get()[e] = 0.0; // brick 0: what out-of-domain neighbours point to
    for (unsigned tk = 1; tk + 1 < T; ++tk)
      for (unsigned tj = 1; tj + 1 < T; ++tj)
        for (unsigned ti = 1; ti + 1 < T; ++ti) {
          const unsigned b = bDecomp[tk][tj][ti];
          for (int k = 0; k < 8; ++k)
            for (int j = 0; j < 8; ++j)
              for (int i = 0; i < 8; ++i)
                hIn[b][k][j][i] = cell_value((org[0] + (ti - 1) * 8 + i) % gdim[0], (org[1] + (tj - 1) * 8 + j) % gdim[1], (org[2] + (tk - 1) * 8 + k) % gdim[2]);
        }
    BrickStorage dev = movBrickStorage(bStorage, hipMemcpyHostToDevice), dev2 = movBrickStorage(bStorageOut, hipMemcpyHostToDevice);
    ExchangeView ev = bDecomp.exchangeView(dev, BRK_XCHG_PACKED, xflags);
    const int S = brk_stencil_iters(BRK_MPI7);
    const uint32_t *adj = (const uint32_t *)bInfo_dev.adj;

    // ---- brick_func of weak/main.cu:246-287: exchange, then ST_ITER sweeps ping-ponging the two storages (the last one without
    // the ghost ring); `calltime` = the exchange (enqueue -> complete), `calctime` = the sweeps, as the reference's GPU driver splits them
    auto brick_func = [&]() {
      double t0 = brk_time::now();
      ev.exchange();
      gpuCheck(hipDeviceSynchronize());
      double t1 = brk_time::now();
      calltime += t1 - t0;
      for (int t = 0; t < S; ++t) {
        double *in = (t & 1) ? dev2.dat.get() : dev.dat.get(), *out = (t & 1) ? dev.dat.get() : dev2.dat.get();
        const uint32_t end = t == S - 1 ? bDecomp.sep_pos[1] : bDecomp.sep_pos[2];
        if (brk_stencil_apply_range(BRK_MPI7, adj, in, out, bsize, 1, end, nullptr, nullptr) != BRK_OK) throw std::runtime_error(brk_last_error());
      }
      gpuCheck(hipDeviceSynchronize());
      calctime += brk_time::now() - t1;
    };

    // ---- validation first (fresh field): one brick_func on the GPU vs the same exchange + sweeps on the host
    brick_func();
    std::vector<bElem> got(total);
    gpuCheck(hipMemcpy(got.data(), dev.dat.get(), total * sizeof(bElem), hipMemcpyDeviceToHost));
    bool same = true;
    {
      // host twin: ghosts by MPI (the reference's run-by-run exchange on the HOST storage: rccl_comm taken away for this call, it
      // routes exchange() to the device path), then S naive sweeps through the accessor
      void *keep = bDecomp.rccl_comm;
      bDecomp.rccl_comm = nullptr;
      bDecomp.exchange(bStorage);
      bDecomp.rccl_comm = keep;
      for (int t = 0; t < S; ++t) {
        Brick3D &in = (t & 1) ? hOut : hIn, &out = (t & 1) ? hIn : hOut;
        const long end = t == S - 1 ? bDecomp.sep_pos[1] : bDecomp.sep_pos[2];
#pragma omp parallel for
        for (long b = 1; b < end; ++b)
          for (int k = 0; k < 8; ++k)
            for (int j = 0; j < 8; ++j)
              for (int i = 0; i < 8; ++i) // stencils/fake.h:346-349
                out[b][k][j][i] = (in[b][k + 1][j][i] + in[b][k - 1][j][i] + in[b][k][j + 1][i] + in[b][k][j - 1][i] + in[b][k][j][i + 1] +
                                   in[b][k][j][i - 1]) * 0.1 + in[b][k][j][i] * 0.4;
      }
      double worst = 0;
      for (size_t e = bsize; e < (size_t)bDecomp.sep_pos[1] * bsize; ++e) {
        const double a = got[e], b = bStorage.dat.get()[e], diff = std::fabs(a - b);
        if (!(diff < 1e-6 || diff < (std::fabs(a) + std::fabs(b)) * 1e-6)) same = false; // include/brickcompare.h:35-39
        worst = diff > worst ? diff : worst;
      }
      double gworst = 0;
      MPI_Reduce(&worst, &gworst, 1, MPI_DOUBLE, MPI_MAX, 0, cart);
      if (rank == 0) std::cout << "validation: max |gpu - host| over the subdomains = " << gworst << std::endl;
    }
    int all_same = same ? 1 : 0, g_same = 1;
    MPI_Allreduce(&all_same, &g_same, 1, MPI_INT, MPI_MIN, cart);

    // ---- timing with the reference's protocol, and its statistics lines
    int cnt = 0;
    const double total_t = time_mpi(brick_func, cnt, bDecomp);
    size_t tsize = 0, opt_size = 0;
    for (auto &g : bDecomp.ghost) tsize += (size_t)g.len * bsize * sizeof(bElem) * 2;
    for (auto s : ev.seclen) opt_size += s * 2;
    mpi_stats calc_s = mpi_statistics(calctime / cnt, cart), call_s = mpi_statistics(calltime / cnt, cart);
    mpi_stats wait_s = mpi_statistics(waittime / cnt, cart), size_s = mpi_statistics((double)tsize * 1.0e-6, cart);
    mpi_stats opt_s = mpi_statistics((double)opt_size * 1.0e-6, cart);
    mpi_stats mspd_s = mpi_statistics(tsize / 1.0e9 / (calltime + waittime) * cnt, cart);
    if (rank == 0) {
      const double tot = calc_s.avg + wait_s.avg + call_s.avg;
      std::cout << "Bri: " << tot << std::endl;
      std::cout << "calc " << calc_s << std::endl;
      std::cout << "call " << call_s << std::endl;
      std::cout << "wait " << wait_s << std::endl;
      std::cout << "  | MPI size (MB): " << size_s << std::endl;
      std::cout << "  | Opt MPI size (MB): " << opt_s << std::endl;
      std::cout << "  | MPI speed (GB/s): " << mspd_s << std::endl;
      const double tot_elems = (double)N * N * N * size * S; // useful updates per brick_func, all ranks (weak/args.cpp:127-129)
      std::cout << "perf " << tot_elems * 1.0e-9 / tot << " GStencil/s" << std::endl;
      std::cout << "Total of " << bDecomp.ghost.size() << " parts (" << size << " ranks, cart " << dims[0] << "x" << dims[1] << "x" << dims[2] << ", "
                << ev.seclen.size() << " RCCL messages per exchange, time_mpi " << total_t * 1e3 << " ms per brick_func)" << std::endl;
      if (!g_same) std::cout << "result mismatch!" << std::endl;
    }
    rc = g_same ? 0 : 1;
    gpuCheck(hipDeviceSynchronize());
    void *c = bDecomp.rccl_comm;
    bDecomp.rccl_comm = nullptr;
    brk_comm_destroy(c);
    free(bInfo.adj);
    MPI_Comm_free(&node);
    MPI_Comm_free(&cart);
  } catch (const std::exception &e) {
    fprintf(stderr, "weak_hip_mpi (world rank %d): %s\n", wrank, e.what());
    MPI_Abort(MPI_COMM_WORLD, 2);
  }
  brkFinalize(); // OpenMP workers joined, the library's device-side caches released, the device idle (brick-gpu.h)
  MPI_Finalize();
  return rc;
}
