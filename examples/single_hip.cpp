// examples/single_hip.cpp -- the reference's single/hip driver shape (single/hip.cpp, stencils/3axis.hip.cpp:81-190)
// on top of the MI355X engine, showing BOTH ways a driver reaches the kernels:
//   (1) an unchanged `brick("7pt.py", VSVEC, (BDIM), (VFOLD), b)` line inside a __global__ kernel with one 64-thread
//       block per brick (d3pt7_brick_trans of the reference), rewritten by tools/brkscatter into brklib::brick_apply;
//   (2) one call of the C ABI, brk_stencil_apply_list, over the same `grid` entries (the fast path).
// Both are checked with compareBrick against the array sweep on the CPU, like the reference does.
//   build: make -C examples        run: examples/single_hip [N]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "brick-hip.h"
#include "brick.h"
#include "brickcompare.h"
#include "bricksetup.h"
#include "brk.h"
#include "crash_trace.h"
#include "multiarray.h"

#ifndef N
#define N 128
#endif
#define TILE 8
#define GZ TILE
#define PADDING 8
#define STRIDE (N + 2 * (GZ + PADDING))
#define STRIDEG (N + 2 * GZ)
#define NB (N / TILE)
#define GB (GZ / TILE)
#define STRIDEB ((N + 2 * GZ) / TILE)
#define BDIM TILE, TILE, TILE
#define VSVEC "HIP"
#define VFOLD 8, 8
#define COEFF_SIZE 129

typedef Brick<Dim<BDIM>, Dim<VFOLD>> Brick3D;

// Brick objects travel to the kernels BY VALUE, exactly as in the reference (stencils/3axis.hip.cpp:32-41,165-171): a Brick
// built from the device BrickInfo (movBrickInfoDeep) and the device BrickStorage (movBrickStorage).

__global__ void d3pt7_brick_trans(unsigned (*grid)[STRIDEB][STRIDEB], Brick3D bIn, Brick3D bOut, bElem *coeff) {
  long tk = GB + hipBlockIdx_z;
  long tj = GB + hipBlockIdx_y;
  long ti = GB + hipBlockIdx_x;
  unsigned b = grid[tk][tj][ti];
  brick("7pt.py", VSVEC, (BDIM), (VFOLD), b);
}

// d3cond_brick_trans of the reference (stencils/3axis.hip.cpp:214-223): same shape, "cond.py"
__global__ void d3cond_brick_trans(unsigned (*grid)[STRIDEB][STRIDEB], Brick3D bIn, Brick3D bOut, bElem *coeff) {
  long tk = GB + hipBlockIdx_z;
  long tj = GB + hipBlockIdx_y;
  long ti = GB + hipBlockIdx_x;
  unsigned b = grid[tk][tj][ti];
  brick("cond.py", VSVEC, (BDIM), (VFOLD), b);
}

// A stencil the library has no hand-written kernel for: examples/lap19.py (19 points, written for this example).
// tools/brkscatter reads the script as data and emits the generic tap path.
__global__ void lap19_brick(unsigned (*grid)[STRIDEB][STRIDEB], Brick3D bIn, Brick3D bOut, bElem *coeff) {
  long tk = GB + hipBlockIdx_z;
  long tj = GB + hipBlockIdx_y;
  long ti = GB + hipBlockIdx_x;
  unsigned b = grid[tk][tj][ti];
  brick("lap19.py", VSVEC, (BDIM), (VFOLD), b);
}

// ... and one that reaches beyond the 125-point cube: examples/star37.py, a radius-3 star with up-/downwind weights (star-shaped
// scripts are provided up to radius 4)
__global__ void star37_brick(unsigned (*grid)[STRIDEB][STRIDEB], Brick3D bIn, Brick3D bOut, bElem *coeff) {
  long tk = GB + hipBlockIdx_z;
  long tj = GB + hipBlockIdx_y;
  long ti = GB + hipBlockIdx_x;
  unsigned b = grid[tk][tj][ti];
  brick("star37.py", VSVEC, (BDIM), (VFOLD), b);
}

// ... and one that is not a sum of taps at all: examples/minmod.py (If / Func / products of grid references); tools/brkscatter writes
// the script's operations out as straight-line code over the staged tile
__global__ void minmod_brick(unsigned (*grid)[STRIDEB][STRIDEB], Brick3D bIn, Brick3D bOut, bElem *coeff) {
  long tk = GB + hipBlockIdx_z;
  long tj = GB + hipBlockIdx_y;
  long ti = GB + hipBlockIdx_x;
  unsigned b = grid[tk][tj][ti];
  brick("minmod.py", VSVEC, (BDIM), (VFOLD), b);
}

// ... and one beyond the 125-point cube: examples/blur343.py, all 343 points of the radius-3 cube (offsets and weights in LDS)
__global__ void blur343_brick(unsigned (*grid)[STRIDEB][STRIDEB], Brick3D bIn, Brick3D bOut, bElem *coeff) {
  long tk = GB + hipBlockIdx_z;
  long tj = GB + hipBlockIdx_y;
  long ti = GB + hipBlockIdx_x;
  unsigned b = grid[tk][tj][ti];
  brick("blur343.py", VSVEC, (BDIM), (VFOLD), b);
}

// ... and a star-shaped one that is none of the built-in stars: examples/aniso7.py, one weight per direction (star staging)
__global__ void aniso7_brick(unsigned (*grid)[STRIDEB][STRIDEB], Brick3D bIn, Brick3D bOut, bElem *coeff) {
  long tk = GB + hipBlockIdx_z;
  long tj = GB + hipBlockIdx_y;
  long ti = GB + hipBlockIdx_x;
  unsigned b = grid[tk][tj][ti];
  brick("aniso7.py", VSVEC, (BDIM), (VFOLD), b);
}

static int run() {
  bElem *coeff = (bElem *)malloc(COEFF_SIZE * sizeof(bElem));
  {
    bElem *r = randomArray({COEFF_SIZE});
    for (int i = 0; i < COEFF_SIZE; ++i) coeff[i] = r[i];
    free(r);
  }
  unsigned *grid_ptr;
  BrickInfo<3> bInfo = init_grid<3>(grid_ptr, {STRIDEB, STRIDEB, STRIDEB});
  unsigned *grid_dev;
  gpuCheck(hipMalloc(&grid_dev, sizeof(unsigned) * STRIDEB * STRIDEB * STRIDEB));
  gpuCheck(hipMemcpy(grid_dev, grid_ptr, sizeof(unsigned) * STRIDEB * STRIDEB * STRIDEB, hipMemcpyHostToDevice));
  auto grid = (unsigned(*)[STRIDEB][STRIDEB])grid_dev;
  BrickInfo<3> *binfo_dev = movBrickInfoDeep(bInfo, hipMemcpyHostToDevice);
  BrickInfo<3> binfo_flat = movBrickInfo(bInfo, hipMemcpyHostToDevice);

  brkSeed(7);
  bElem *in_ptr = randomArray({STRIDE, STRIDE, STRIDE});
  for (long l = 0; l < (long)STRIDE * STRIDE * STRIDE; ++l) in_ptr[l] -= 0.5; // sign changes, so that d3cond's max() matters
  bElem *out_ptr = zeroArray({STRIDE, STRIDE, STRIDE});
  auto arr_in = (bElem(*)[STRIDE][STRIDE])in_ptr;
  auto arr_out = (bElem(*)[STRIDE][STRIDE])out_ptr;
  bElem *coeff_dev;
  gpuCheck(hipMalloc(&coeff_dev, COEFF_SIZE * sizeof(bElem)));
  gpuCheck(hipMemcpy(coeff_dev, coeff, COEFF_SIZE * sizeof(bElem), hipMemcpyHostToDevice));

  auto bsize = cal_size<BDIM>::value;
  auto bstorage = BrickStorage::allocate(bInfo.nbricks, bsize * 2);
  Brick3D bIn(&bInfo, bstorage, 0), bOut(&bInfo, bstorage, bsize);
  copyToBrick<3>({STRIDEG, STRIDEG, STRIDEG}, {PADDING, PADDING, PADDING}, {0, 0, 0}, in_ptr, grid_ptr, bIn);
  BrickStorage bstorage_dev = movBrickStorage(bstorage, hipMemcpyHostToDevice);

  // CPU array sweep = what compareBrick checks against (3axis.hip.cpp:130-135)
#pragma omp parallel for collapse(2)
  for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
    for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j)
      for (long i = PADDING + GZ; i < PADDING + GZ + N; ++i)
        arr_out[k][j][i] = coeff[5] * arr_in[k + 1][j][i] + coeff[6] * arr_in[k - 1][j][i] + coeff[3] * arr_in[k][j + 1][i] +
                           coeff[4] * arr_in[k][j - 1][i] + coeff[1] * arr_in[k][j][i + 1] + coeff[2] * arr_in[k][j][i - 1] +
                           coeff[0] * arr_in[k][j][i];

  auto check = [&](const char *what) {
    gpuCheck(hipMemcpy(bstorage.dat.get(), bstorage_dev.dat.get(), bstorage.chunks * bstorage.step * sizeof(bElem),
                       hipMemcpyDeviceToHost));
    if (!compareBrick<3>({N, N, N}, {PADDING, PADDING, PADDING}, {GZ, GZ, GZ}, out_ptr, grid_ptr, bOut)) {
      std::cout << what << ": result mismatch!" << std::endl;
      exit(1);
    }
    std::cout << what << ": result match" << std::endl;
    gpuCheck(hipMemset(bstorage_dev.dat.get() + 0, 0, 0)); // keep the input; the out field is overwritten by the next path
  };

  // (1) reference-shaped kernel, brick(...) line rewritten by tools/brkscatter
  {
    Brick3D dIn(binfo_dev, bstorage_dev, 0), dOut(binfo_dev, bstorage_dev, bsize);
    dim3 block(NB, NB, NB), thread(64);
    hipLaunchKernelGGL(d3pt7_brick_trans, block, thread, 0, 0, grid, dIn, dOut, coeff_dev);
    gpuCheck(hipDeviceSynchronize());
    check("brick() kernel");
  }
  // (2) C ABI over the same grid entries
  {
    std::vector<unsigned> ids;
    for (long tk = GB; tk < STRIDEB - GB; ++tk)
      for (long tj = GB; tj < STRIDEB - GB; ++tj)
        for (long ti = GB; ti < STRIDEB - GB; ++ti) ids.push_back(grid_ptr[(tk * STRIDEB + tj) * STRIDEB + ti]);
    unsigned *ids_dev;
    gpuCheck(hipMalloc(&ids_dev, ids.size() * sizeof(unsigned)));
    gpuCheck(hipMemcpy(ids_dev, ids.data(), ids.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    gpuCheck(hipMemset(bstorage_dev.dat.get(), 0, 0));
    int rc = brk_stencil_apply_list(BRK_7PT_COEFF, (const uint32_t *)binfo_flat.adj, bstorage_dev.dat.get(),
                                    bstorage_dev.dat.get() + bsize, bstorage_dev.step, ids_dev, (uint32_t)ids.size(), coeff,
                                    nullptr);
    if (rc) {
      fprintf(stderr, "brk: %s\n", brk_last_error());
      return 1;
    }
    gpuCheck(hipDeviceSynchronize());
    check("brk_stencil_apply_list");
  }
  // (3) d3cond (stencils/3axis.cpp:94-163): |sum c*max(in,0)|, reference-shaped kernel with the rewritten brick("cond.py")
  {
    using std::max;
#pragma omp parallel for collapse(2)
    for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
      for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j)
        for (long i = PADDING + GZ; i < PADDING + GZ + N; ++i) {
          bElem partial = coeff[5] * max(arr_in[k + 1][j][i], 0.0) + coeff[6] * max(arr_in[k - 1][j][i], 0.0) +
                          coeff[3] * max(arr_in[k][j + 1][i], 0.0) + coeff[4] * max(arr_in[k][j - 1][i], 0.0) +
                          coeff[1] * max(arr_in[k][j][i + 1], 0.0) + coeff[2] * max(arr_in[k][j][i - 1], 0.0) +
                          coeff[0] * max(arr_in[k][j][i], 0.0);
          arr_out[k][j][i] = partial > 0 ? partial : -partial;
        }
    Brick3D dIn(binfo_dev, bstorage_dev, 0), dOut(binfo_dev, bstorage_dev, bsize);
    dim3 block(NB, NB, NB), thread(64);
    hipLaunchKernelGGL(d3cond_brick_trans, block, thread, 0, 0, grid, dIn, dOut, coeff_dev);
    gpuCheck(hipDeviceSynchronize());
    check("brick() cond kernel");
  }
  // (4) a user-written script (lap19.py) through the generic tap path
  {
#pragma omp parallel for collapse(2)
    for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
      for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j)
        for (long i = PADDING + GZ; i < PADDING + GZ + N; ++i) {
          bElem acc = coeff[0] * arr_in[k][j][i];
          for (int s = 1; s >= -1; s -= 2) {
            acc += coeff[1] * (arr_in[k][j][i + s] + arr_in[k][j + s][i] + arr_in[k + s][j][i]);
            for (int t = 1; t >= -1; t -= 2)
              acc += coeff[2] * (arr_in[k][j + t][i + s] + arr_in[k + t][j][i + s] + arr_in[k + t][j + s][i]);
          }
          arr_out[k][j][i] = acc;
        }
    Brick3D dIn(binfo_dev, bstorage_dev, 0), dOut(binfo_dev, bstorage_dev, bsize);
    dim3 block(NB, NB, NB), thread(64);
    hipLaunchKernelGGL(lap19_brick, block, thread, 0, 0, grid, dIn, dOut, coeff_dev);
    gpuCheck(hipDeviceSynchronize());
    check("brick() lap19 kernel (script read as data)");
  }
  // (5) a user-written radius-3 star (star37.py) through the star-shaped tap path
  {
#pragma omp parallel for collapse(2)
    for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
      for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j)
        for (long i = PADDING + GZ; i < PADDING + GZ + N; ++i) {
          bElem acc = coeff[0] * arr_in[k][j][i];
          for (int d = 1; d <= 3; ++d) {
            acc += coeff[d] * (arr_in[k][j][i + d] + arr_in[k][j + d][i] + arr_in[k + d][j][i]);
            acc += coeff[3 + d] * (arr_in[k][j][i - d] + arr_in[k][j - d][i] + arr_in[k - d][j][i]);
          }
          arr_out[k][j][i] = acc;
        }
    Brick3D dIn(binfo_dev, bstorage_dev, 0), dOut(binfo_dev, bstorage_dev, bsize);
    dim3 block(NB, NB, NB), thread(64);
    hipLaunchKernelGGL(star37_brick, block, thread, 0, 0, grid, dIn, dOut, coeff_dev);
    gpuCheck(hipDeviceSynchronize());
    check("brick() star37 kernel (radius-3 star script read as data)");
  }
  // (6) a user-written non-linear script (minmod.py): limited slopes, the script's operations as generated straight-line code
  {
    auto minmod = [](bElem l, bElem r) -> bElem { return l * r > 0 ? (l > 0 ? 1.0 : -1.0) * std::min(std::abs(l), std::abs(r)) : 0.0; };
#pragma omp parallel for collapse(2)
    for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
      for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j)
        for (long i = PADDING + GZ; i < PADDING + GZ + N; ++i) {
          const bElem c = arr_in[k][j][i];
          const bElem s = minmod(c - arr_in[k][j][i - 1], arr_in[k][j][i + 1] - c) + minmod(c - arr_in[k][j - 1][i], arr_in[k][j + 1][i] - c) +
                          minmod(c - arr_in[k - 1][j][i], arr_in[k + 1][j][i] - c);
          arr_out[k][j][i] = c - coeff[0] * s;
        }
    Brick3D dIn(binfo_dev, bstorage_dev, 0), dOut(binfo_dev, bstorage_dev, bsize);
    dim3 block(NB, NB, NB), thread(64);
    hipLaunchKernelGGL(minmod_brick, block, thread, 0, 0, grid, dIn, dOut, coeff_dev);
    gpuCheck(hipDeviceSynchronize());
    check("brick() minmod kernel (non-linear script, generated straight-line code)");
  }
  // (7) a user-written radius-3 CUBE (blur343.py): 343 taps, tap table in LDS
  {
#pragma omp parallel for collapse(2)
    for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
      for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j)
        for (long i = PADDING + GZ; i < PADDING + GZ + N; ++i) {
          bElem acc = 0;
          for (int dk = -3; dk <= 3; ++dk)
            for (int dj = -3; dj <= 3; ++dj)
              for (int di = -3; di <= 3; ++di) acc += coeff[std::abs(di) + std::abs(dj) + std::abs(dk)] * arr_in[k + dk][j + dj][i + di];
          arr_out[k][j][i] = acc;
        }
    Brick3D dIn(binfo_dev, bstorage_dev, 0), dOut(binfo_dev, bstorage_dev, bsize);
    dim3 block(NB, NB, NB), thread(64);
    hipLaunchKernelGGL(blur343_brick, block, thread, 0, 0, grid, dIn, dOut, coeff_dev);
    gpuCheck(hipDeviceSynchronize());
    check("brick() blur343 kernel (radius-3 cube script, 343 taps)");
  }
  // (8) a user-written star with one weight per direction (aniso7.py): the star staging, radius 1
  {
#pragma omp parallel for collapse(2)
    for (long k = PADDING + GZ; k < PADDING + GZ + N; ++k)
      for (long j = PADDING + GZ; j < PADDING + GZ + N; ++j)
        for (long i = PADDING + GZ; i < PADDING + GZ + N; ++i)
          arr_out[k][j][i] = coeff[0] * arr_in[k][j][i] + coeff[1] * arr_in[k][j][i + 1] + 0.5 * coeff[2] * arr_in[k][j][i - 1] +
                             coeff[3] * arr_in[k][j + 1][i] - coeff[4] * arr_in[k][j - 1][i] + coeff[5] * arr_in[k + 1][j][i] +
                             0.25 * coeff[6] * arr_in[k - 1][j][i];
    Brick3D dIn(binfo_dev, bstorage_dev, 0), dOut(binfo_dev, bstorage_dev, bsize);
    dim3 block(NB, NB, NB), thread(64);
    hipLaunchKernelGGL(aniso7_brick, block, thread, 0, 0, grid, dIn, dOut, coeff_dev);
    gpuCheck(hipDeviceSynchronize());
    check("brick() aniso7 kernel (star-shaped script, star staging)");
  }
  return 0;
}

// brkFinalize (brick-gpu.h): OpenMP workers joined, the library's device-side caches released and the device idle before exit()
// runs its handlers; what run() allocated on the device and did not free goes with the process, as in the
// reference's driver
int main() {
  brk_crash::install(); // a crash prints its thread and call stack before the process dies by the signal (crash_trace.h)
  brk_crash::census("first line of main");
  const int rc = run();
  brk_crash::snapshot();
  brkFinalize();
  brk_crash::census("just before return");
  return rc;
}
