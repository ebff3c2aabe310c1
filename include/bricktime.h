// bricktime.h -- the reference's three timing protocols for drivers, same names and return values:
//   time_func     stencils/stencils.h:40-53      CPU: warm-up, then batches of 1, 2, 4, ... calls until TOT_TIME seconds have passed;
//                                                returns (ed - st) / (it - 1) = seconds per call
//   hiptime_func  stencils/stencils.hip.h:13-31  GPU: warm-up, device synchronise, HIP_ITER launches between two events on the
//                                                null stream; returns seconds per launch
//   time_mpi      stencils/fake.h:392-404        MPI: warm-up, zero the pack/call/wait/move/calc accumulators, MPI_ITER calls;
//                                                returns seconds per call and the count in `cnt`
// If the reference's own stencils/stencils.h or stencils.hip.h has been included first, its definitions are used and the
// ones here stay out of the way.  The Python twins are in bricklib_amd/timing.py.
#ifndef BRICKTIME_H
#define BRICKTIME_H

#include <chrono>

#ifndef TOT_TIME
#define TOT_TIME 5
#endif
#ifndef HIP_ITER
#define HIP_ITER 100
#endif

namespace brk_time {
inline double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} // namespace brk_time

#ifndef BRICK_STENCILS_H
template <typename T>
double time_func(T func) {
  int it = 1;
  func(); // warm up
  double st = brk_time::now(), ed = st;
  while (ed < st + TOT_TIME) {
    for (int i = 0; i < it; ++i) func();
    it <<= 1;
    ed = brk_time::now();
  }
  return (ed - st) / (it - 1);
}
#endif

#if defined(__HIP__) && !defined(BRICK_STENCILS_HIP_H)
#include <hip/hip_runtime.h>
template <typename T>
double hiptime_func(T func) {
  func();
  hipEvent_t start_event, stop_event;
  (void)hipDeviceSynchronize();
  (void)hipEventCreate(&start_event);
  (void)hipEventCreate(&stop_event);
  (void)hipEventRecord(start_event);
  for (int i = 0; i < HIP_ITER; i++) func();
  (void)hipEventRecord(stop_event);
  (void)hipEventSynchronize(stop_event);
  float elapsed = 0;
  (void)hipEventElapsedTime(&elapsed, start_event, stop_event);
  (void)hipEventDestroy(start_event);
  (void)hipEventDestroy(stop_event);
  return elapsed / HIP_ITER / 1000.0;
}
#endif

#ifndef BRICK_FAKE_H
// the drivers define MPI_ITER (weak/args.cpp: 25 on the CPU, 100 on the GPU) and the accumulators of brick-mpi.h
extern int MPI_ITER;
template <typename T, typename decomp>
double time_mpi(T func, int &cnt, decomp &) {
  int it = MPI_ITER;
  cnt = 0;
  func(); // warm up
#ifdef BRK_HAVE_TIMERS
  packtime = calltime = waittime = movetime = calctime = 0; // brick-mpi.h (included first) defines the accumulators and this macro
#endif
  double st = brk_time::now();
  for (int i = 0; i < MPI_ITER; ++i) func();
  double ed = brk_time::now();
  cnt = it;
  return (ed - st) / it;
}
#endif

#endif // BRICKTIME_H
