// examples/strong_hip_mpi.cpp -- the reference's strong-scaling GPU driver as it is started: ONE PROCESS PER RANK under mpiexec
// (strong/main.cu:113-575, strong/args.cpp): MPI_Init -> parseArgs (-d domain, -s subdomain, -I iterations, -v validate) -> the
// rank's contiguous range of Z-Morton subdomain ids (brk_strong_partition = args.cpp:104-113) -> one GPU per rank -> an RCCL
// communicator over MPI_COMM_WORLD's ranks (brkCommFromMPI) -> all subdomains of the rank in ONE storage with a block-diagonal
// adjacency table -> the exchange tables of strong/main.cu:188-395 at run granularity, THIS rank's rows only: ghost runs whose
// source subdomain lives on this rank are aliased through the adjacency table (or, with -c, copied by device links as in the
// reference, RegionLink main.cu:201-208), everything bound for one peer rank is packed into one buffer -- ONE ncclSend + ONE
// ncclRecv per peer inside one group (brk_exchange_begin / brk_exchange_transport / brk_exchange_end; the reference posts one
// MPI_Isend + MPI_Irecv per peer, main.cu:418-435) -> brick_func = exchange + ST_ITER sweeps of the 13-point stencil over every
// subdomain of the rank in one launch (main.cu:464-476) -> time_mpi and the reference's statistics lines (main.cu:536-566):
//   Bri / calc / pack / call / wait / MPI size / MPI speed / perf ... GStencil/s / part N.
// -v (the reference's CPU-only validation, strong/main.cpp:367-395, here for the GPU path): every rank sweeps the GLOBAL periodic
// array on its host cores and compares its own subdomains; MPI_Allreduce(MAX) of the error; "Validated" / "INVALID" on rank 0.
// The twin of weak_hip_mpi.cpp; examples/strong_hip.cpp plays all ranks in one process over brk_exchange_loopback instead.
//   build: make -C examples strong_hip_mpi   (only where <mpi.h> exists)
//   run:   mpiexec -np R examples/strong_hip_mpi [-d 64] [-s 16] [-I 20] [-v] [-c]
#include <mpi.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <tuple>
#include <vector>

#include "brick-hip.h"
#include "brick-mpi.h"
#include "bricktime.h"
#include "brk.h"

int MPI_ITER = 20;
typedef BrickDecomp<3, 8, 8, 8> Decomp;

#define BK(x)                                                                \
  do {                                                                       \
    if ((x) != BRK_OK) throw std::runtime_error(std::string(#x) + ": " + brk_last_error()); \
  } while (0)

// input field: a hash of the GLOBAL cell coordinate (the reference fills with random numbers and compares with its own array sweep)
static double field(long x, long y, long z, long d) {
  unsigned long long h = 42ull + (unsigned long long)((z * d + y) * d + x) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 30, h *= 0xBF58476D1CE4E5B9ull, h ^= h >> 27, h *= 0x94D049BB133111EBull, h ^= h >> 31;
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

static int run(int argc, char **argv, int rank, int size) {
  long d = 64, s = 16;
  bool validate = false, alias_local = true;
  for (int c; (c = getopt(argc, argv, "d:s:I:vch")) != -1;) {
    if (c == 'd') d = atol(optarg);
    else if (c == 's') s = atol(optarg);
    else if (c == 'I') MPI_ITER = atoi(optarg);
    else if (c == 'v') validate = true;
    else if (c == 'c') alias_local = false;
    else {
      if (rank == 0) printf("%s [-d domain cells per axis] [-s subdomain cells per axis] [-I iterations] [-v validate] [-c copy local ghosts]\n", argv[0]);
      return c == 'h' ? 0 : 2;
    }
  }
  const uint64_t subdim = (uint64_t)(d / s), allsubs = subdim * subdim * subdim;
  if (d % s || s % 8 || (subdim & (subdim - 1)) || (uint64_t)size > allsubs) {
    if (rank == 0) fprintf(stderr, "need d %% s == 0, s %% 8 == 0, a power of two of subdomains per axis and at most one rank per subdomain\n");
    return 2;
  }
  if (rank == 0) { // strong/args.cpp:91-101
    std::cout << "MPI Size " << size << std::endl;
    std::cout << "Domain size of " << d << "^3 decomposed into " << s << "^3 subdomains" << std::endl;
    std::cout << "Total of " << allsubs << " subdomains, " << (allsubs + size - 1) / size << " per rank" << std::endl;
  }
  // ---- one GPU per rank: ranks of one node take consecutive devices
  MPI_Comm node;
  MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
  int local = 0, ndev = 0;
  MPI_Comm_rank(node, &local);
  gpuCheck(hipGetDeviceCount(&ndev));
  if (ndev == 0) throw std::runtime_error("no GPU visible");
  if (size > 1 && local >= ndev) throw std::runtime_error("more ranks on this node than GPUs: one rank per GPU is required under RCCL");
  gpuCheck(hipSetDevice(local % ndev));
  void *comm = brkCommFromMPI(MPI_COMM_WORLD);

  // ONE decomposition shared by all subdomains (main.cpp:91-94)
  Decomp bd({(unsigned)s, (unsigned)s, (unsigned)s}, 8);
  bd.initialize(skin3d_good);
  const uint32_t nb = bd.nbricks();
  BrickInfo<3> one = bd.getBrickInfo();
  const size_t esz = sizeof(bElem), bbytes = 512 * esz;
  const long SB = s / 8 + 2;

  uint64_t lr[2];
  brk_strong_partition(allsubs, size, rank, lr);
  const uint64_t L = lr[0];
  const size_t nsub = (size_t)(lr[1] - lr[0]), elems = nsub * nb * 512;
  double *st[3]; // three storages of nsub * nb bricks (main.cpp:359-365 rotates them), placed in HBM against one another
  for (int w = 0; w < 3; ++w) {
    void *placed = nullptr;
    BK(brk_storage_alloc(elems * esz, &placed, nullptr));
    st[w] = static_cast<double *>(placed);
    gpuCheck(hipMemset(st[w], 0, elems * esz));
  }
  {
    std::vector<double> host(elems, 0.0);
    for (size_t m = 0; m < nsub; ++m) {
      const uint64_t z = L + m;
      const long ox = (long)brk_zmort_get(z, 0) * s, oy = (long)brk_zmort_get(z, 1) * s, oz = (long)brk_zmort_get(z, 2) * s;
      for (long kb = 1; kb < SB - 1; ++kb)
        for (long jb = 1; jb < SB - 1; ++jb)
          for (long ib = 1; ib < SB - 1; ++ib) {
            double *b = &host[(m * nb + bd[(int)kb][(int)jb][(int)ib]) * 512];
            for (long k = 0; k < 8; ++k)
              for (long j = 0; j < 8; ++j)
                for (long i = 0; i < 8; ++i)
                  b[(k * 8 + j) * 8 + i] = field(ox + (ib - 1) * 8 + i, oy + (jb - 1) * 8 + j, oz + (kb - 1) * 8 + k, d);
          }
    }
    gpuCheck(hipMemcpy(st[0], host.data(), elems * esz, hipMemcpyHostToDevice));
  }

  // ---- this rank's exchange tables (main.cu:188-395 at run granularity); both ends enumerate a peer's runs in the same canonical
  // order (receiver Z-Morton id, run index), so the buffers need no header
  auto off = [&](size_t m, uint32_t pos) { return (uint64_t)((m * nb + pos) * bbytes); };
  std::vector<brk_xcopy> copies;
  std::vector<uint32_t> alias(nsub * nb); // brick -> the brick that holds its data
  for (size_t b = 0; b < alias.size(); ++b) alias[b] = (uint32_t)b;
  typedef std::tuple<uint64_t, uint32_t, size_t> Key; // (receiver zmort, run, local subdomain)
  std::map<int, std::vector<Key>> recv, send;
  for (size_t m = 0; m < nsub; ++m) {
    const uint64_t z = L + m;
    for (uint32_t q = 0; q < (uint32_t)bd.ghost.size(); ++q) {
      int owner = 0, slot = 0;
      // receive side: ghost run q of z comes from the subdomain in direction ghost[q].neighbor (getrank, args.cpp:36-56)
      brk_strong_neighbor(bd.ghost[q].neighbor.set, z, subdim, allsubs, size, &owner, &slot);
      if (owner == rank && alias_local)
        for (uint32_t t = 0; t < bd.ghost[q].len; ++t) alias[m * nb + bd.ghost[q].pos + t] = (uint32_t)((size_t)slot * nb + bd.skin[q].pos + t);
      else if (owner == rank)
        copies.push_back(brk_xcopy{0, 0, BRK_SP_STORAGE, BRK_SP_STORAGE, -1, -1, off((size_t)slot, bd.skin[q].pos), off(m, bd.ghost[q].pos),
                                   (uint64_t)bd.ghost[q].len * bbytes});
      else
        recv[owner].emplace_back(z, q, m);
      // send side: skin run q of z goes to the subdomain in direction skin[q].neighbor
      const uint64_t zr = brk_strong_neighbor(bd.skin[q].neighbor.set, z, subdim, allsubs, size, &owner, &slot);
      if (owner != rank) send[owner].emplace_back(zr, q, m);
    }
  }
  std::vector<int> peers;
  for (auto &kv : recv) peers.push_back(kv.first);
  for (auto &kv : send) peers.push_back(kv.first);
  std::sort(peers.begin(), peers.end());
  peers.erase(std::unique(peers.begin(), peers.end()), peers.end());
  std::vector<brk_xmsg> msgs;
  size_t tsize = 0; // bytes leaving + arriving per exchange (main.cu:530-534)
  for (int p : peers) {
    uint64_t so = 0, ro = 0;
    auto &sv = send[p], &rv = recv[p];
    std::sort(sv.begin(), sv.end());
    std::sort(rv.begin(), rv.end());
    for (auto &k : sv) {
      const uint32_t q = std::get<1>(k);
      const uint64_t n = (uint64_t)bd.skin[q].len * bbytes;
      copies.push_back(brk_xcopy{0, 0, BRK_SP_STORAGE, BRK_SP_SEND, -1, p, off(std::get<2>(k), bd.skin[q].pos), so, n});
      so += n;
    }
    for (auto &k : rv) {
      const uint32_t q = std::get<1>(k);
      const uint64_t n = (uint64_t)bd.ghost[q].len * bbytes;
      copies.push_back(brk_xcopy{0, 1, BRK_SP_RECV, BRK_SP_STORAGE, p, -1, ro, off(std::get<2>(k), bd.ghost[q].pos), n});
      ro += n;
    }
    if (ro) msgs.push_back(brk_xmsg{0, 0, p, BRK_SP_RECV, 0, ro});
    if (so) msgs.push_back(brk_xmsg{0, 1, p, BRK_SP_SEND, 0, so});
    tsize += (size_t)(so + ro);
  }
  // block-diagonal adjacency: brick b of subdomain m is brick m * nb + b of the rank's storage; aliased ghost bricks are replaced
  // by their sources and left out of the sweep over subdomains + ghost rings
  uint32_t *adj_dev = nullptr, *list_all = nullptr, *list_domain = nullptr;
  uint32_t n_all = 0, n_domain = 0;
  {
    std::vector<uint32_t> adj((size_t)nsub * nb * 27), all, dom;
    for (size_t m = 0; m < nsub; ++m) {
      for (size_t q = 0; q < (size_t)nb * 27; ++q) adj[m * nb * 27 + q] = alias[(&one.adj[0][0])[q] + (uint32_t)(m * nb)];
      for (uint32_t b = 1; b < bd.sep_pos[2]; ++b)
        if (alias[m * nb + b] == (uint32_t)(m * nb) + b) all.push_back((uint32_t)(m * nb) + b); // domain + ghost bricks
      for (uint32_t b = 1; b < bd.sep_pos[1]; ++b) dom.push_back((uint32_t)(m * nb) + b);      // the subdomain itself
    }
    n_all = (uint32_t)all.size(), n_domain = (uint32_t)dom.size();
    gpuCheck(hipMalloc(&adj_dev, adj.size() * 4));
    gpuCheck(hipMemcpy(adj_dev, adj.data(), adj.size() * 4, hipMemcpyHostToDevice));
    gpuCheck(hipMalloc(&list_all, all.size() * 4));
    gpuCheck(hipMemcpy(list_all, all.data(), all.size() * 4, hipMemcpyHostToDevice));
    gpuCheck(hipMalloc(&list_domain, dom.size() * 4));
    gpuCheck(hipMemcpy(list_domain, dom.data(), dom.size() * 4, hipMemcpyHostToDevice));
  }
  brk_xplan *plan = nullptr;
  brk_exchange_view *view = nullptr;
  BK(brk_xplan_build_tables(rank, (uint32_t)copies.size(), copies.data(), (uint32_t)msgs.size(), msgs.data(), &plan));
  BK(brk_exchange_view_create(plan, st[0], &view));
  printf("rank %d: subdomains [%llu, %llu), %zu peers, %.2f MB leave the rank per exchange, %.2f MB are copied inside it, %u of %u bricks swept "
         "per sweep (%s)\n", rank, (unsigned long long)lr[0], (unsigned long long)lr[1], peers.size(), brk_xplan_bytes_remote(plan) / 1e6,
         brk_xplan_bytes_local(plan) / 1e6, n_all, (uint32_t)(nsub * (bd.sep_pos[2] - 1)), alias_local ? "local ghosts aliased" : "local ghosts copied");
  fflush(stdout);

  const int S = brk_stencil_iters(BRK_MPI13);
  hipEvent_t ev[5];
  for (auto &e : ev) gpuCheck(hipEventCreate(&e));
  // brick_func of strong/main.cu:397-510.  The phases are timed as the reference's CUDA_AWARE build times them: `pack` = links +
  // pack and the unpack launch (events), `calc` = the sweeps (events), `call` = enqueueing the transport group, `wait` = the host
  // waiting for the stream (the transport runs on it, between the pack and the unpack, so its time is in `wait`).
  auto brick_func = [&]() {
    gpuCheck(hipEventRecord(ev[0], nullptr));
    BK(brk_exchange_begin(view, 0, nullptr)); // same-rank links + pack (main.cu:440-443,479-482)
    gpuCheck(hipEventRecord(ev[1], nullptr));
    const double t_a = brk_time::now();
    if (!peers.empty()) BK(brk_exchange_transport(view, 0, comm, nullptr)); // one send + one recv per peer (main.cu:418-435)
    const double t_b = brk_time::now();
    calltime += t_b - t_a;
    gpuCheck(hipEventRecord(ev[2], nullptr));
    BK(brk_exchange_end(view, 0, nullptr)); // unpack (main.cu:459-462)
    gpuCheck(hipEventRecord(ev[3], nullptr));
    // 1 <- 0; (2 <- 1; 1 <- 2) x (S/2 - 1); 0 <- 1 over the subdomains only (main.cpp:359-365)
    BK(brk_stencil_apply_list(BRK_MPI13, adj_dev, st[0], st[1], 512, list_all, n_all, nullptr, nullptr));
    for (int t = 0; t < S / 2 - 1; ++t) {
      BK(brk_stencil_apply_list(BRK_MPI13, adj_dev, st[1], st[2], 512, list_all, n_all, nullptr, nullptr));
      BK(brk_stencil_apply_list(BRK_MPI13, adj_dev, st[2], st[1], 512, list_all, n_all, nullptr, nullptr));
    }
    BK(brk_stencil_apply_list(BRK_MPI13, adj_dev, st[1], st[0], 512, list_domain, n_domain, nullptr, nullptr));
    gpuCheck(hipEventRecord(ev[4], nullptr));
    gpuCheck(hipEventSynchronize(ev[4]));
    waittime += brk_time::now() - t_b;
    float ms = 0, ms2 = 0;
    gpuCheck(hipEventElapsedTime(&ms, ev[3], ev[4]));
    calctime += ms / 1000.0;
    gpuCheck(hipEventElapsedTime(&ms, ev[0], ev[1]));
    gpuCheck(hipEventElapsedTime(&ms2, ev[2], ev[3]));
    packtime += (ms + ms2) / 1000.0;
  };

  brick_func();
  gpuCheck(hipDeviceSynchronize());

  int rc = 0;
  if (validate) { // the global periodic array swept S times on the CPU (stencils/fake.h:73-79 weights)
    std::vector<double> ga((size_t)d * d * d), gb(ga.size());
    for (long z = 0; z < d; ++z)
      for (long y = 0; y < d; ++y)
        for (long x = 0; x < d; ++x) ga[((size_t)z * d + y) * d + x] = field(x, y, z, d);
    const double B0 = 0.4, B1 = 0.07, B2 = 0.03;
    auto at = [&](const std::vector<double> &g, long x, long y, long z) { return g[((size_t)((z + d) % d) * d + (y + d) % d) * d + (x + d) % d]; };
    for (int t = 0; t < S; ++t) {
#pragma omp parallel for
      for (long z = 0; z < d; ++z)
        for (long y = 0; y < d; ++y)
          for (long x = 0; x < d; ++x)
            gb[((size_t)z * d + y) * d + x] =
                B2 * (at(ga, x, y, z + 2) + at(ga, x, y, z - 2) + at(ga, x, y + 2, z) + at(ga, x, y - 2, z) + at(ga, x + 2, y, z) + at(ga, x - 2, y, z)) +
                B1 * (at(ga, x, y, z + 1) + at(ga, x, y, z - 1) + at(ga, x, y + 1, z) + at(ga, x, y - 1, z) + at(ga, x + 1, y, z) + at(ga, x - 1, y, z)) +
                B0 * at(ga, x, y, z);
      ga.swap(gb);
    }
    double worst = 0.0, gworst = 0.0;
    std::vector<double> host(elems);
    gpuCheck(hipMemcpy(host.data(), st[0], elems * esz, hipMemcpyDeviceToHost));
    for (size_t m = 0; m < nsub; ++m) {
      const uint64_t z = L + m;
      const long ox = (long)brk_zmort_get(z, 0) * s, oy = (long)brk_zmort_get(z, 1) * s, oz = (long)brk_zmort_get(z, 2) * s;
      for (long kb = 1; kb < SB - 1; ++kb)
        for (long jb = 1; jb < SB - 1; ++jb)
          for (long ib = 1; ib < SB - 1; ++ib) {
            const double *b = &host[(m * nb + bd[(int)kb][(int)jb][(int)ib]) * 512];
            for (long k = 0; k < 8; ++k)
              for (long j = 0; j < 8; ++j)
                for (long i = 0; i < 8; ++i)
                  worst = std::max(worst, std::fabs(b[(k * 8 + j) * 8 + i] - at(ga, ox + (ib - 1) * 8 + i, oy + (jb - 1) * 8 + j, oz + (kb - 1) * 8 + k)));
          }
    }
    MPI_Allreduce(&worst, &gworst, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
    const bool ok = gworst <= 1e-12;
    if (rank == 0)
      printf("%s: max |GPU - CPU| over all %llu subdomains of %d ranks = %.3e after %d sweeps\n", ok ? "Validated" : "INVALID", (unsigned long long)allsubs,
             size, gworst, S);
    rc = ok ? 0 : 1;
  }

  // ---- timing with the reference's protocol and its statistics lines (main.cu:527-566); data keeps evolving, as in the reference
  int cnt = 0;
  const double tot = time_mpi(brick_func, cnt, bd);
  cnt *= S;
  mpi_stats calc_s = mpi_statistics(calctime / cnt, MPI_COMM_WORLD), pack_s = mpi_statistics(packtime / cnt, MPI_COMM_WORLD);
  mpi_stats call_s = mpi_statistics(calltime / cnt, MPI_COMM_WORLD), wait_s = mpi_statistics(waittime / cnt, MPI_COMM_WORLD);
  mpi_stats size_s = mpi_statistics((double)tsize * 1.0e-6, MPI_COMM_WORLD);
  mpi_stats mspd_s = mpi_statistics(tsize / 1.0e9 / (calltime + waittime) * cnt, MPI_COMM_WORLD);
  if (rank == 0) {
    // `wait` covers the whole stream (pack, transport, unpack, sweeps): the per-sweep cost is tot / ST_ITER, as the reference's
    // second figure on its "Bri:" line
    std::cout << "Bri: " << call_s.avg + wait_s.avg << " : " << tot / S << std::endl;
    std::cout << "calc : " << calc_s << std::endl;
    std::cout << "pack : " << pack_s << std::endl;
    std::cout << "call : " << call_s << std::endl;
    std::cout << "wait : " << wait_s << std::endl;
    std::cout << "  | MPI size (MB): " << size_s << std::endl;
    std::cout << "  | MPI speed (GB/s): " << mspd_s << std::endl;
    double perf = (double)d / 1000.0;
    perf = perf * perf * perf / (tot / S);
    std::cout << "perf " << perf << " GStencil/s" << std::endl;
    std::cout << "part " << msgs.size() << std::endl;
  }
  gpuCheck(hipDeviceSynchronize());
  for (auto &e : ev) gpuCheck(hipEventDestroy(e));
  brk_exchange_view_free(view);
  brk_xplan_free(plan);
  for (int w = 0; w < 3; ++w) BK(brk_storage_free(st[w]));
  gpuCheck(hipFree(adj_dev));
  gpuCheck(hipFree(list_all));
  gpuCheck(hipFree(list_domain));
  free(one.adj);
  brk_comm_destroy(comm);
  MPI_Comm_free(&node);
  return rc;
}

int main(int argc, char **argv) {
  int provided = 0;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SERIALIZED, &provided);
  int size = 1, rank = 0;
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  int rc = 0;
  try {
    rc = run(argc, argv, rank, size);
  } catch (const std::exception &e) {
    fprintf(stderr, "strong_hip_mpi (rank %d): %s\n", rank, e.what());
    MPI_Abort(MPI_COMM_WORLD, 2);
  }
  brkFinalize(); // OpenMP workers joined, the library's device-side caches released, the device idle (brick-gpu.h)
  MPI_Finalize();
  return rc;
}
