// brk_sbase.h -- SGPR-base addressing for the kernels (device code only; part of what brk_jit.hip carries as text)
#ifndef BRK_SBASE_H
#define BRK_SBASE_H
namespace brklib {
// Global-address-space pointers whose (wave-uniform) value is hidden from the optimiser: accesses through them are
// emitted as global_load/global_store with an SGPR base and a 32-bit per-lane offset.  Without this hipcc folds
// base+lane into 64-bit VGPR addresses (two VGPRs and two VALU ops per access).  The address space must be kept
// explicit: a generic pointer after the asm would turn the accesses into flat_* instructions, which force
// vmcnt(0)/lgkmcnt(0) waits.
typedef const __attribute__((address_space(1))) double *gcptr;
typedef __attribute__((address_space(1))) double *gptr;
__device__ __forceinline__ gcptr sbase(const double *p) {
  gcptr g = (gcptr)p;
  asm volatile("" : "+s"(g));
  return g;
}

} // namespace brklib
#endif
