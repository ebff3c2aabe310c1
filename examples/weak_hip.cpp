// examples/weak_hip.cpp -- the reference's weak driver shape (weak/main.cu:100-300) on the MI355X engine, C++ only:
//   BrickDecomp<3,8,8,8> + populate -> movBrickStorage -> ExchangeView / MultiStageExchangeView / bDecomp.exchange(bStorage)
//   (GPU-resident pack -> ncclSend/ncclRecv -> unpack inside libbrk.so) -> ST_ITER sweeps through the C ABI, timed with time_mpi.
// One rank, periodic 1x1x1 grid: every neighbour is this rank.  The exchange is run (a) as device links and (b) with the same
// runs forced through a 1-rank RCCL communicator (ncclSend to self), and every ghost brick is compared with the surface brick
// it mirrors.  With MPI one adds MPI_Init, MPI_Cart_create, populate(cart, bDecomp, 0, 1, coo) and
// bDecomp.rccl_comm = brkCommFromMPI(cart) -- the rest is unchanged.
//   build: make -C examples        run: examples/weak_hip [cells per axis, default 64]
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "brick-hip.h"
#include "brick-mpi.h"
#include "bricksetup.h"
#include "bricktime.h"
#include "brk.h"
#include "crash_trace.h"

int MPI_ITER = 20;

#define BK(x)                                                     \
  do {                                                            \
    if ((x) != BRK_OK) {                                          \
      fprintf(stderr, "%s: %s\n", #x, brk_last_error());          \
      return 1;                                                   \
    }                                                             \
  } while (0)

static int run(int argc, char **argv) {
  const unsigned N = argc > 1 ? (unsigned)atoi(argv[1]) : 64;
  BrickDecomp<3, 8, 8, 8> bDecomp({N, N, N}, 8);
  bDecomp.initialize(skin3d_good);
  const int cart[3] = {1, 1, 1}, coo[3] = {0, 0, 0};
  populate_cart(bDecomp, cart, coo);
  BrickInfo<3> bInfo = bDecomp.getBrickInfo();
  BrickInfo<3> bInfo_dev = movBrickInfo(bInfo, hipMemcpyHostToDevice);
  // written once: the library may keep compact tables next to it (brk_adjacency_register); released and freed when main leaves
  brkAdjacencyScope<3> adjacency(bInfo_dev);
  const size_t bsize = 512;
  auto bStorage = bInfo.allocate(bsize), bOther = bInfo.allocate(bsize);
  const size_t total = (size_t)bStorage.chunks * bStorage.step;
  auto fill = [&]() { // every element knows which brick it came from
    for (size_t e = 0; e < total; ++e) bStorage.dat.get()[e] = (double)e;
  };
  fill();
  BrickStorage dev = movBrickStorage(bStorage, hipMemcpyHostToDevice), dev2 = movBrickStorage(bOther, hipMemcpyHostToDevice);
  auto reset = [&]() { gpuCheck(hipMemcpy(dev.dat.get(), bStorage.dat.get(), total * sizeof(bElem), hipMemcpyHostToDevice)); };
  std::vector<bElem> back(total);
  auto ghosts_ok = [&](const char *what) {
    gpuCheck(hipDeviceSynchronize());
    gpuCheck(hipMemcpy(back.data(), dev.dat.get(), total * sizeof(bElem), hipMemcpyDeviceToHost));
    size_t bad = 0, checked = 0;
    for (size_t q = 0; q < bDecomp.ghost.size(); ++q) // ghost run q mirrors skin run q (one periodic rank)
      for (size_t e = 0; e < (size_t)bDecomp.ghost[q].len * bsize; ++e, ++checked)
        bad += back[(size_t)bDecomp.ghost[q].pos * bsize + e] != (double)((size_t)bDecomp.skin[q].pos * bsize + e);
    std::cout << what << ": " << (bad ? "ghost mismatch" : "ghosts match") << " (" << checked << " elements)" << std::endl;
    reset();
    return bad == 0;
  };
  bool ok = true;
  {
    ExchangeView ev = bDecomp.exchangeView(dev);
    ev.exchange();
    ok &= ghosts_ok("ExchangeView (packed, device links)");
    ExchangeView evd = bDecomp.exchangeView(dev, BRK_XCHG_DIRECT);
    evd.exchange();
    ok &= ghosts_ok("ExchangeView (direct)");
    MultiStageExchangeView ms = bDecomp.multiStageExchangeView(dev);
    ms.exchange();
    ok &= ghosts_ok("MultiStageExchangeView (3 stages)");
    bDecomp.exchange_device(dev);
    ok &= ghosts_ok("bDecomp.exchange(bStorage)");
  }
  // the same runs through RCCL: a 1-rank communicator made through the C ABI, ncclSend/ncclRecv to self inside libbrk.so
  {
    char id[128];
    void *comm = nullptr;
    BK(brk_comm_unique_id(id));
    BK(brk_comm_init_rank(&comm, 1, id, 0));
    bDecomp.rccl_comm = comm;
    for (int mode : {BRK_XCHG_PACKED, BRK_XCHG_DIRECT, BRK_XCHG_SHIFT}) {
      ExchangeView ev = bDecomp.exchangeView(dev, mode, BRK_XCHG_SELF_VIA_COMM);
      ev.exchange();
      char what[96];
      snprintf(what, sizeof what, "ExchangeView mode %d over RCCL (%zu messages to self)", mode, ev.seclen.size());
      ok &= ghosts_ok(what);
    }
    bDecomp.rccl_comm = nullptr;
    gpuCheck(hipDeviceSynchronize());
    BK(brk_comm_destroy(comm));
  }
  // brick_func of weak/main.cu:246-287: exchange, then ST_ITER sweeps ping-ponging the two storages; time_mpi protocol
  {
    ExchangeView ev = bDecomp.exchangeView(dev);
    const int S = brk_stencil_iters(BRK_MPI7);
    const uint32_t *adj = (const uint32_t *)bInfo_dev.adj;
    auto brick_func = [&]() {
      ev.exchange();
      for (int t = 0; t < S; ++t) {
        double *in = (t & 1) ? dev2.dat.get() : dev.dat.get(), *out = (t & 1) ? dev.dat.get() : dev2.dat.get();
        const uint32_t end = t == S - 1 ? bDecomp.sep_pos[1] : bDecomp.sep_pos[2];
        if (brk_stencil_apply_range(BRK_MPI7, adj, in, out, bsize, 1, end, nullptr, nullptr) != BRK_OK) {
          fprintf(stderr, "%s\n", brk_last_error());
          exit(1);
        }
      }
      gpuCheck(hipDeviceSynchronize());
    };
    int cnt = 0;
    const double sec = time_mpi(brick_func, cnt, bDecomp);
    std::cout << "weak brick_func: " << sec * 1e3 << " ms per exchange + " << S << " sweeps, "
              << (double)N * N * N * S / sec / 1e9 << " GStencil/s (" << cnt << " iterations, time_mpi)" << std::endl;
  }
  // the mmap allocators of the reference exist by name and say why they are not provided
  try {
    BrickStorage::mmap_alloc(16, 512);
    ok = false;
  } catch (const std::runtime_error &e) {
    std::cout << "mmap_alloc: refused (" << std::string(e.what()).substr(0, 40) << "...)" << std::endl;
  }
  free(bInfo.adj);
  std::cout << (ok ? "weak_hip: all exchanges match" : "weak_hip: FAILED") << std::endl;
  return ok ? 0 : 1;
}

// every device object of run() -- storages, the registered adjacency, views -- is gone when it returns; brkFinalize (brick-gpu.h)
// then ends the library's and the process's use of the GPU on this thread, before exit() runs
int main(int argc, char **argv) {
  brk_crash::install(); // a crash prints its thread and call stack before the process dies by the signal (crash_trace.h)
  brk_crash::census("first line of main");
  const int rc = run(argc, argv);
  brk_crash::snapshot();
  brkFinalize();
  brk_crash::census("just before return");
  return rc;
}
