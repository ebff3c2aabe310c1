// mfma64probe -- development probe: what does v_mfma_f64_16x16x4_f64 cost on gfx950, and does it run beside fp64 VALU work of
// another wave on the same SIMD?  (Round 4: the review suggested moving the i-direction fold of the 125-point stencil onto the
// matrix pipe.)  Prints cycles per instruction from s_memtime around long unrolled chains, one workgroup per CU.
//   build: make -C tools mfma64probe      run: tools/mfma64probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                          \
  do {                                                                 \
    hipError_t e = (x);                                                \
    if (e != hipSuccess) {                                             \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));           \
      exit(2);                                                         \
    }                                                                  \
  } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 256;

// MODE 0: every wave issues MFMA f64; 1: every wave issues fp64 FMAs; 2: even waves MFMA, odd waves FMA (same SIMD when 8 waves)
template <int MODE>
__global__ void k_probe(double *out, unsigned long long *cyc, int nacc) {
  const int wave = threadIdx.x >> 6;
  const bool mfma = MODE == 0 || (MODE == 2 && (wave & 4) == 0); // waves 0..3 / 4..7 land on SIMD 0..3 twice
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double f[8] = {a, b, a, b, a, b, a, b};
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (mfma) {
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 8; ++q) f[q] = __builtin_fma(f[q], a, b);
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0;
  for (int q = 0; q < 4; ++q) s += acc[q].x + acc[q].y + acc[q].z + acc[q].w;
  for (int q = 0; q < 8; ++q) s += f[q];
  // `out` holds 256 x 512 doubles, `cyc` 256 x 8 stamps: larger grids wrap around (the values are not read)
  out[(blockIdx.x & 255) * 512 + threadIdx.x] = s + nacc;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x & 255) * 8 + wave] = t1 - t0;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  double *out;
  unsigned long long *cyc, host[8];
  CK(hipMalloc(&out, 256 * 512 * 8));
  CK(hipMalloc(&cyc, 256 * 8 * 8));
  const char *names[] = {"MFMA f64 16x16x4 on every wave", "fp64 FMA on every wave", "waves 0-3 MFMA, waves 4-7 fp64 FMA"};
  for (int waves : {4, 8})
    for (int mode = 0; mode < 3; ++mode) {
      if (mode == 2 && waves == 4) continue;
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, 4);
        else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, 4);
        else hipLaunchKernelGGL(k_probe<2>, dim3(256), dim3(64 * waves), 0, 0, out, cyc, 4);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(host, cyc, waves * 8, hipMemcpyDeviceToHost));
      // s_memtime counts at 100 MHz on gfx9 (constant clock); report raw ticks per instruction of wave 0 and of the last wave
      const double n_mfma = kIters * 4.0, n_fma = kIters * 32.0;
      printf("%d waves/CU (%d per SIMD), %-36s: wave 0 %.3f ticks per %s, wave %d %.3f ticks per %s\n", waves, waves / 4, names[mode],
             host[0] / (mode == 1 ? n_fma : n_mfma), mode == 1 ? "FMA" : "MFMA", waves - 1,
             host[waves - 1] / ((mode == 1 || mode == 2) ? n_fma : n_mfma), (mode == 1 || mode == 2) ? "FMA" : "MFMA");
    }
  // wall-clock version: total time of a launch of 256 workgroups, flops per second
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    const int waves = 8;
    CK(hipEventRecord(e0));
    for (int rep = 0; rep < 20; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(2560), dim3(64 * waves), 0, 0, out, cyc, 4);
      else if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(2560), dim3(64 * waves), 0, 0, out, cyc, 4);
      else hipLaunchKernelGGL(k_probe<2>, dim3(2560), dim3(64 * waves), 0, 0, out, cyc, 4);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_waves = mode == 0 ? 8 : (mode == 2 ? 4 : 0), fma_waves = 8 - mfma_waves;
    const double flops = 20.0 * 2560 * (mfma_waves * kIters * 4 * 2048.0 + fma_waves * kIters * 32 * 128.0);
    printf("wall clock, 2560 workgroups of 8 waves, %-36s: %.3f ms, %.1f TFLOP/s fp64 (MFMA counted as 2048 flops, FMA wave-instruction as 128)\n", names[mode], ms / 20,
           flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
