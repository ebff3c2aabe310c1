// brk_jit.hip -- stencil expressions compiled at run time (brk_expr_compile / brk_expr_apply, include/brk.h).
//
// The reference is a code generator: a stencil script becomes C++ before the compiler sees the driver (codegen/vecscatter,
// codegen/st/codegen/base.py:678).  A host that holds a script as DATA at run time gets the same here: the register program
// (brk_expr_op[], what bricklib_amd/spec.py makes of a script) is written out as straight-line HIP over the staged tile
// (brklib::brick_apply_fn of include/brick-device.h -- the same text tools/brkscatter emits for a brick(...) line) and compiled by
// hiprtc for the device in use, once per distinct program.  The kernel around it is k_expr's: one wave per brick, four bricks per
// workgroup, XCD-aware slot order.  brk_stencil_apply_expr (the interpreter) runs the same operations without a compiler.
//
// Host code only.  libhiprtc is dlopen'ed on first use (like librccl in brk_exchange.hip): a process that never compiles a script
// does not map it.  The two headers the generated source includes are embedded at build time (brk_jit_headers.inc).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>

#include "brick-device.h" // brklib::Coeffs (an argument of k_s7_wg)
#include "brk_internal.h"

namespace {
#include "brk_jit_headers.inc" // kJitDeviceHeader (brick-device.h), kJitAbiHeader (brk.h), kJitSbaseHeader (brk_sbase.h), kJitStar7Header (brk_star7_kernel.h), kJitStarWgHeader (brk_starwg_kernel.h), kJitCubeHeader (brk_cubedense_kernel.h)

struct Rtc {
  void *handle = nullptr;
  char why[256] = "";
  decltype(&hiprtcCreateProgram) Create = nullptr;
  decltype(&hiprtcCompileProgram) Compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) LogSize = nullptr;
  decltype(&hiprtcGetProgramLog) Log = nullptr;
  decltype(&hiprtcGetCodeSize) CodeSize = nullptr;
  decltype(&hiprtcGetCode) Code = nullptr;
  decltype(&hiprtcDestroyProgram) Destroy = nullptr;
  decltype(&hiprtcGetErrorString) ErrorString = nullptr;
};

Rtc &rtc_state() {
  static Rtc R;
  return R;
}

Rtc *rtc() {
  Rtc &R = rtc_state();
  static std::once_flag once;
  std::call_once(once, [&R] {
    static const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
    for (const char *n : names) // one already mapped (a framework's own copy) first
      if ((R.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t q = 0; !R.handle && q < sizeof names / sizeof *names; ++q) R.handle = dlopen(names[q], RTLD_NOW | RTLD_LOCAL);
    if (!R.handle) {
      snprintf(R.why, sizeof R.why, "libhiprtc could not be loaded: %s", dlerror());
      return;
    }
    bool ok = true;
    auto sym = [&](const char *n) {
      void *p = dlsym(R.handle, n);
      if (!p && ok) ok = false, snprintf(R.why, sizeof R.why, "libhiprtc lacks %s", n);
      return p;
    };
    R.Create = reinterpret_cast<decltype(R.Create)>(sym("hiprtcCreateProgram"));
    R.Compile = reinterpret_cast<decltype(R.Compile)>(sym("hiprtcCompileProgram"));
    R.LogSize = reinterpret_cast<decltype(R.LogSize)>(sym("hiprtcGetProgramLogSize"));
    R.Log = reinterpret_cast<decltype(R.Log)>(sym("hiprtcGetProgramLog"));
    R.CodeSize = reinterpret_cast<decltype(R.CodeSize)>(sym("hiprtcGetCodeSize"));
    R.Code = reinterpret_cast<decltype(R.Code)>(sym("hiprtcGetCode"));
    R.Destroy = reinterpret_cast<decltype(R.Destroy)>(sym("hiprtcDestroyProgram"));
    R.ErrorString = reinterpret_cast<decltype(R.ErrorString)>(sym("hiprtcGetErrorString"));
    if (!ok) R.handle = nullptr;
  });
  return R.handle ? &R : nullptr;
}

int nsources(uint32_t code) {
  return code == BRK_EX_TAP || code == BRK_EX_CONST ? 0 : (code == BRK_EX_NEG || code == BRK_EX_ABS || code == BRK_EX_SQRT ? 1 : (code == BRK_EX_SEL ? 3 : 2));
}

// One statement per operation, in program order, each an fp64 operation of its own (the unit is compiled with -ffp-contract=off):
// the same numbers as the interpreter and as the checker's evaluation of the tree.
bool star_shaped(const brk_expr_op *ops, uint32_t nops) {
  for (uint32_t t = 0; t < nops; ++t)
    if (ops[t].code == BRK_EX_TAP && (ops[t].offset[0] != 0) + (ops[t].offset[1] != 0) + (ops[t].offset[2] != 0) > 1) return false;
  return true;
}

// programs of radius 1 or 2 with taps off the axes go into the cube kernel -- while they are short: its fetch side leaves the compute
// phase 168 registers at three groups per CU, and a sum of 125 taps written out (501 operations x 8 planes) spills there (8.4 ms
// against 3.4 on the plain tile; such sums belong to brk_stencil_apply_taps anyway)
bool cube_form(const brk_expr_op *ops, uint32_t nops, int R) { return !star_shaped(ops, nops) && R <= 2 && nops <= 160; }

// waves per workgroup of the generated kernel = CubeWaves<R> / StarWaves<R> of brick-device.h
uint32_t waves_per_group(int R, bool star) { return star ? (R <= 2 ? 4u : 2u) : (R <= 2 ? 4u : (R == 3 ? 2u : 1u)); }

// The kernels a star-shaped program of radius 1 instantiates: k_s7_wg<kUser, 8, kModeList / kModeRange> of brk_star7_kernel.h in namespace
// brklib::jit (Itanium names of the explicit instantiations the generated unit ends with)
const char kS7List[] = "_ZN6brklib3jit7k_s7_wgILi3ELi8ELi1ELi0ELi0EEEvPKjPKdPdmS3_jjNS0_3BoxENS_6CoeffsES3_S3_";
const char kS7Range[] = "_ZN6brklib3jit7k_s7_wgILi3ELi8ELi0ELi0ELi0EEEvPKjPKdPdmS3_jjNS0_3BoxENS_6CoeffsES3_S3_";

// ... and a star-shaped program of radius 2, 3 or 4: k_star_wg<kStarUser2, 4, 4> / <kStarUser4, 4, 3> of brk_starwg_kernel.h
// (radius 2 at four groups per CU; five, like the built-in 13-point kernel, measured the same for a 13-tap program: 0.477 vs 0.475 ms)
const char kSw2[] = "_ZN6brklib3jit9k_star_wgILi5ELi4ELi4ELi0EEEvPKjPKdPdmS3_jjNS_6CoeffsES3_";
const char kSw4[] = "_ZN6brklib3jit9k_star_wgILi6ELi4ELi3ELi0EEEvPKjPKdPdmS3_jjNS_6CoeffsES3_";

// ... and a program of radius 1 that is not star-shaped: k_cube_dense<kCubeUser1, 4, 4, 1> of brk_cubedense_kernel.h
const char kCube1[] = "_ZN6brklib3jit12k_cube_denseILi3ELi4ELi4ELi1EEEvPKjPKdPdmS3_jjNS_6CoeffsEiS3_S3_jj";
// ... radius 2, not star-shaped: k_cube_dense<kCubeUser2, 4, 3, 5> (the 125-point kernel's fetch side: everything in 16-byte lanes to LDS)
const char kCube2[] = "_ZN6brklib3jit12k_cube_denseILi4ELi4ELi3ELi5EEEvPKjPKdPdmS3_jjNS_6CoeffsEiS3_S3_jj";

void statement_rhs(const brk_expr_op &o, const char *tap, char (&rhs)[160]) {
  char a[16], b[16], c[16];
  snprintf(a, sizeof a, "r%u", o.a), snprintf(b, sizeof b, "r%u", o.b), snprintf(c, sizeof c, "r%u", o.c);
  switch (o.code) {
  case BRK_EX_TAP: snprintf(rhs, sizeof rhs, "%s", tap); break;
  case BRK_EX_CONST: snprintf(rhs, sizeof rhs, "%a", o.imm); break; // hexadecimal floating literal: the exact value
  case BRK_EX_ADD: snprintf(rhs, sizeof rhs, "%s + %s", a, b); break;
  case BRK_EX_SUB: snprintf(rhs, sizeof rhs, "%s - %s", a, b); break;
  case BRK_EX_MUL: snprintf(rhs, sizeof rhs, "%s * %s", a, b); break;
  case BRK_EX_DIV: snprintf(rhs, sizeof rhs, "%s / %s", a, b); break;
  case BRK_EX_NEG: snprintf(rhs, sizeof rhs, "-%s", a); break;
  case BRK_EX_MAX: snprintf(rhs, sizeof rhs, "%s > %s ? %s : %s", a, b, a, b); break;
  case BRK_EX_MIN: snprintf(rhs, sizeof rhs, "%s < %s ? %s : %s", a, b, a, b); break;
  case BRK_EX_ABS: snprintf(rhs, sizeof rhs, "%s < 0 ? -%s : %s", a, a, a); break;
  case BRK_EX_SQRT: snprintf(rhs, sizeof rhs, "__builtin_sqrt(%s)", a); break;
  case BRK_EX_GT: snprintf(rhs, sizeof rhs, "%s > %s ? 1.0 : 0.0", a, b); break;
  case BRK_EX_GE: snprintf(rhs, sizeof rhs, "%s >= %s ? 1.0 : 0.0", a, b); break;
  case BRK_EX_LT: snprintf(rhs, sizeof rhs, "%s < %s ? 1.0 : 0.0", a, b); break;
  case BRK_EX_LE: snprintf(rhs, sizeof rhs, "%s <= %s ? 1.0 : 0.0", a, b); break;
  case BRK_EX_EQ: snprintf(rhs, sizeof rhs, "%s == %s ? 1.0 : 0.0", a, b); break;
  case BRK_EX_NE: snprintf(rhs, sizeof rhs, "%s != %s ? 1.0 : 0.0", a, b); break;
  default: snprintf(rhs, sizeof rhs, "%s != 0.0 ? %s : %s", a, b, c); break; // BRK_EX_SEL
  }
  if (o.code == BRK_EX_CONST && (o.imm != o.imm || o.imm - o.imm != 0.0)) // NaN / infinity have no literal
    snprintf(rhs, sizeof rhs, "%s", o.imm != o.imm ? "__builtin_nan(\"\")" : (o.imm > 0 ? "__builtin_inf()" : "-__builtin_inf()"));
}

std::string declare_registers(const brk_expr_op *ops, uint32_t nops) {
  std::string s = "    double";
  bool used[BRK_EXPR_MAX_REGS] = {};
  for (uint32_t t = 0; t < nops; ++t) used[ops[t].dst] = true;
  bool first = true;
  char buf[16];
  for (int r = 0; r < BRK_EXPR_MAX_REGS; ++r)
    if (used[r]) snprintf(buf, sizeof buf, "%s r%d", first ? "" : ",", r), s += buf, first = false;
  return s + ";\n";
}

const char kTypedefs[] = // hiprtc brings the runtime's declarations but no <cstdint>
    "typedef unsigned char uint8_t; typedef signed char int8_t; typedef unsigned short uint16_t; typedef short int16_t; typedef int int32_t;\n"
    "typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t; typedef unsigned long uintptr_t;\n";

// A star-shaped program of radius 1 (the shape of the 7-point stencils and of stencils/cond.py): the HEADLINE kernel k_s7_wg -- eight
// bricks per workgroup, i-faces out of the sibling waves' tiles, k in registers -- instantiated around the program as its cell update.
std::string generate_s7(const brk_expr_op *ops, uint32_t nops, uint32_t result) {
  std::string s = "// generated by brk_expr_compile (bricklib_amd/csrc/brk_jit.hip): the program as the cell update of k_s7_wg\n";
  s += kTypedefs;
  s += "#define BRK_JIT_POINT7\n#include \"brk_star7_kernel.h\"\nnamespace brklib {\nnamespace jit {\n"
       "__device__ __forceinline__ double brk_user_point7(double kp, double km, double jp, double jm, double ip, double im, double c0) {\n";
  s += declare_registers(ops, nops);
  char buf[256], rhs[160];
  for (uint32_t t = 0; t < nops; ++t) {
    const brk_expr_op &o = ops[t];
    const char *tap = o.offset[0] ? (o.offset[0] > 0 ? "ip" : "im") : (o.offset[1] ? (o.offset[1] > 0 ? "jp" : "jm") : (o.offset[2] ? (o.offset[2] > 0 ? "kp" : "km") : "c0"));
    statement_rhs(o, tap, rhs);
    snprintf(buf, sizeof buf, "    r%u = %s;\n", o.dst, rhs);
    s += buf;
  }
  snprintf(buf, sizeof buf, "    return r%u;\n}\n", result);
  s += buf;
  s += "template __global__ void k_s7_wg<kUser, 8, kModeList>(const uint32_t *, const double *, double *, size_t, const uint32_t *, uint32_t, uint32_t, Box, Coeffs, const uint32_t *, const uint32_t *);\n"
       "template __global__ void k_s7_wg<kUser, 8, kModeRange>(const uint32_t *, const double *, double *, size_t, const uint32_t *, uint32_t, uint32_t, Box, Coeffs, const uint32_t *, const uint32_t *);\n"
       "} // namespace jit\n} // namespace brklib\n";
  return s;
}

// A star-shaped program of radius 2..4 (finite-difference scripts of higher order): the radius-2 / radius-4 star kernel k_star_wg -- four
// bricks per workgroup, i-columns out of the sibling waves' tiles, the k-column in registers -- around the program as its cell update.
std::string generate_sw(const brk_expr_op *ops, uint32_t nops, uint32_t result, int R) {
  std::string s = "// generated by brk_expr_compile (bricklib_amd/csrc/brk_jit.hip): the program as the cell update of k_star_wg\n";
  s += kTypedefs;
  s += "#define BRK_JIT_STAR\n#include \"brk_starwg_kernel.h\"\nnamespace brklib {\nnamespace jit {\n"
       "__device__ __forceinline__ double brk_user_star(const double *ck, const double *jm, const double *jp, const double *im, const double *ip) {\n";
  s += declare_registers(ops, nops);
  char buf[256], rhs[160], tap[32];
  for (uint32_t t = 0; t < nops; ++t) {
    const brk_expr_op &o = ops[t];
    const int di = o.offset[0], dj = o.offset[1], dk = o.offset[2];
    if (di) snprintf(tap, sizeof tap, "%s[%d]", di > 0 ? "ip" : "im", (di > 0 ? di : -di) - 1);
    else if (dj) snprintf(tap, sizeof tap, "%s[%d]", dj > 0 ? "jp" : "jm", (dj > 0 ? dj : -dj) - 1);
    else snprintf(tap, sizeof tap, "ck[%d]", dk);
    statement_rhs(o, tap, rhs);
    snprintf(buf, sizeof buf, "    r%u = %s;\n", o.dst, rhs);
    s += buf;
  }
  snprintf(buf, sizeof buf, "    return r%u;\n}\n", result);
  s += buf;
  s += R == 2 ? "template __global__ void k_star_wg<kStarUser2, 4, 4>(" : "template __global__ void k_star_wg<kStarUser4, 4, 3>(";
  s += "const uint32_t *, const double *, double *, size_t, const uint32_t *, uint32_t, uint32_t, Coeffs, const uint32_t *);\n"
       "} // namespace jit\n} // namespace brklib\n";
  return s;
}

// A program of radius 1 or 2 with taps off the axes (mixed derivatives, Moore-neighbourhood rules, 19- / 27-point operators with
// limiters): the 27- / 125-point kernel's machinery -- dense bank-conflict-free tiles, i-halo columns out of the sibling waves' tiles,
// exact links, 16-byte staging -- around the program as its cell update.
std::string generate_cube(const brk_expr_op *ops, uint32_t nops, uint32_t result, int R) {
  std::string s = "// generated by brk_expr_compile (bricklib_amd/csrc/brk_jit.hip): the program as the cell update of k_cube_dense\n";
  s += kTypedefs;
  s += "#define BRK_JIT_CUBE\n#include \"brk_cubedense_kernel.h\"\nnamespace brklib {\nnamespace jit {\n"
       "__device__ __forceinline__ double brk_user_cube(const double *c, const int *col) {\n";
  s += declare_registers(ops, nops);
  char buf[256], rhs[160], tap[48];
  for (uint32_t t = 0; t < nops; ++t) {
    const brk_expr_op &o = ops[t];
    snprintf(tap, sizeof tap, "c[%d + col[%d]]", o.offset[2] * (8 + 2 * R) * 8 + o.offset[1] * 8, o.offset[0]); // plane pitch (8 + 2R) rows x 8, row pitch 8
    statement_rhs(o, tap, rhs);
    snprintf(buf, sizeof buf, "    r%u = %s;\n", o.dst, rhs);
    s += buf;
  }
  snprintf(buf, sizeof buf, "    return r%u;\n}\n", result);
  s += buf;
  s += R == 1 ? "template __global__ void k_cube_dense<kCubeUser1, 4, 4, 1>(" : "template __global__ void k_cube_dense<kCubeUser2, 4, 3, 5>(";
  s += "const uint32_t *, const double *, double *, size_t, const uint32_t *, uint32_t, uint32_t, Coeffs, int, const uint32_t *, const uint32_t *, uint32_t, uint32_t);\n"
       "} // namespace jit\n} // namespace brklib\n";
  return s;
}

std::string generate(const brk_expr_op *ops, uint32_t nops, uint32_t result, int R) {
  std::string s;
  const bool star = star_shaped(ops, nops);
  if (cube_form(ops, nops, R)) return generate_cube(ops, nops, result, R);
  if (star && R == 1) return generate_s7(ops, nops, result);
  if (star) return generate_sw(ops, nops, result, R == 2 ? 2 : 4); // radius 3 runs on the radius-4 kernel
  s += "// generated by brk_expr_compile (bricklib_amd/csrc/brk_jit.hip)\n";
  s += kTypedefs;
  s += "#include \"brick-device.h\"\n"
       "#define BRK_JIT_R " + std::to_string(R) + "\n"
       "#define BRK_JIT_WAVES brklib::" + (star ? "StarWaves" : "CubeWaves") + "<BRK_JIT_R>::WAVES\n"
       "extern \"C\" __global__ __launch_bounds__(64 * BRK_JIT_WAVES) void k_expr_jit(const uint32_t *__restrict__ adj, const double *__restrict__ in, double *__restrict__ out,\n"
       "                                                           size_t step, const uint32_t *__restrict__ ids, uint32_t first, uint32_t n) {\n"
       "  const uint32_t per_xcd = gridDim.x >> 3;\n"
       "  const uint32_t wg = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);\n"
       "  const uint32_t slot = wg * BRK_JIT_WAVES + (threadIdx.x >> 6);\n"
       "  if (slot >= n) return;\n"
       "  const uint32_t b = __builtin_amdgcn_readfirstlane(ids ? ids[slot] : first + slot);\n"
       "  if (b == BRK_LIST_SKIP) return;\n";
  char buf[256], rhs[160], tap[64];
  // star-shaped programs (every tap on one axis): the star staging, no edge / corner gathers
  if (star) snprintf(buf, sizeof buf, "  brklib::brick_apply_star_fn<%d>(adj, in, out, step, b, [](const double *ti, const double *tj, const double *ck) -> double {\n", R);
  else snprintf(buf, sizeof buf, "  brklib::brick_apply_fn<%d>(adj, in, out, step, b, [](const double *c) -> double {\n", R);
  s += buf;
  s += declare_registers(ops, nops);
  for (uint32_t t = 0; t < nops; ++t) {
    const brk_expr_op &o = ops[t];
    if (!star) snprintf(tap, sizeof tap, "c[brklib::tap_rel<%d>(%d, %d, %d)]", R, o.offset[0], o.offset[1], o.offset[2]);
    else if (o.offset[0]) snprintf(tap, sizeof tap, "ti[%d]", o.offset[0]);
    else if (o.offset[1]) snprintf(tap, sizeof tap, "tj[%d]", 8 * o.offset[1]);
    else snprintf(tap, sizeof tap, "ck[%d]", o.offset[2]);
    statement_rhs(o, tap, rhs);
    snprintf(buf, sizeof buf, "    r%u = %s;\n", o.dst, rhs);
    s += buf;
  }
  snprintf(buf, sizeof buf, "    return r%u;\n  });\n}\n", result);
  s += buf;
  return s;
}

uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char *c = static_cast<const unsigned char *>(p);
  for (size_t q = 0; q < n; ++q) h = (h ^ c[q]) * 1099511628211ull;
  return h;
}
} // namespace

struct brk_expr_kernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr, fn_range = nullptr; // s7 form: the list and the range instantiation of k_s7_wg; else one kernel
  int radius = 1, device = 0;
  bool star = false, from_disk = false;
  bool s7() const { return star && radius == 1; }
  bool cube = false; // k_cube_dense around the program (cube_form)
  std::string source, code; // generated text; the code object (kept: brk_finalize unloads the module, a later compile of the same program loads it again)
};

extern "C" int brki_expr_check(const brk_expr_op *ops, uint32_t nops, uint32_t max_ops, uint32_t result_reg, int *radius) {
  if (!ops || nops == 0 || nops > max_ops) return brk_fail(BRK_EINVAL, "1..%u ops expected (got %u)", max_ops, nops);
  if (result_reg >= (uint32_t)BRK_EXPR_MAX_REGS) return brk_fail(BRK_EINVAL, "result register %u (0..%d)", result_reg, BRK_EXPR_MAX_REGS - 1);
  bool written[BRK_EXPR_MAX_REGS] = {}; // every cell starts with an undefined register file
  int maxabs = 0;
  for (uint32_t t = 0; t < nops; ++t) {
    const brk_expr_op &o = ops[t];
    if (o.code > (uint32_t)BRK_EX_SEL) return brk_fail(BRK_EINVAL, "op %u: unknown code %u", t, o.code);
    if (o.dst >= (uint32_t)BRK_EXPR_MAX_REGS) return brk_fail(BRK_EINVAL, "op %u: destination register %u", t, o.dst);
    const uint32_t src[3] = {o.a, o.b, o.c};
    for (int q = 0; q < nsources(o.code); ++q)
      if (src[q] >= (uint32_t)BRK_EXPR_MAX_REGS || !written[src[q]])
        return brk_fail(BRK_EINVAL, "op %u: source register %u is read before it is written", t, src[q]);
    for (int a = 0; a < 3 && o.code == BRK_EX_TAP; ++a) {
      const int v = o.offset[a] < 0 ? -o.offset[a] : o.offset[a];
      if (v > maxabs) maxabs = v;
    }
    written[o.dst] = true;
  }
  if (!written[result_reg]) return brk_fail(BRK_EINVAL, "result register %u is never written", result_reg);
  if (maxabs > 4) return brk_fail(BRK_EINVAL, "tap offset %d: expression stencils reach |offset| <= 4 per axis", maxabs);
  if (radius) *radius = maxabs > 1 ? maxabs : 1;
  return BRK_OK;
}

extern "C" int brk_expr_source(const brk_expr_op *ops, uint32_t nops, uint32_t result_reg, char *buf, size_t cap, size_t *need) {
  int R = 1;
  if (int rc = brki_expr_check(ops, nops, BRK_EXPR_COMPILE_MAX_OPS, result_reg, &R)) return rc;
  const std::string s = generate(ops, nops, result_reg, R);
  if (need) *need = s.size() + 1;
  if (buf && cap) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return BRK_OK;
}

namespace {
std::mutex g_jit_mutex;
struct Cached {
  std::unique_ptr<brk_expr_kernel> k;
  int refs = 0;
};
std::map<uint64_t, Cached> &cache() {
  static auto *m = new std::map<uint64_t, Cached>; // leaked on purpose: modules are unloaded by brk_expr_release / brk_finalize, not by a static destructor
  return *m;
}
} // namespace

extern "C" int brk_expr_compile(const brk_expr_op *ops, uint32_t nops, uint32_t result_reg, brk_expr_kernel **out) {
  if (!out) return brk_fail(BRK_EINVAL, "null result");
  *out = nullptr;
  int R = 1;
  if (int rc = brki_expr_check(ops, nops, BRK_EXPR_COMPILE_MAX_OPS, result_reg, &R)) return rc;
  int dev = 0;
  char arch[300] = "--offload-arch=gfx950";
  bool have_device = hipGetDevice(&dev) == hipSuccess;
  if (have_device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) snprintf(arch, sizeof arch, "--offload-arch=%s", prop.gcnArchName);
    else have_device = false;
  }
  (void)hipGetLastError();
  // the program's identity: only what the generated text depends on
  uint64_t key = fnv(&result_reg, sizeof result_reg, fnv(&dev, sizeof dev));
  for (uint32_t t = 0; t < nops; ++t) {
    const brk_expr_op &o = ops[t];
    const uint32_t f[5] = {o.code, o.dst, nsources(o.code) > 0 ? o.a : 0, nsources(o.code) > 1 ? o.b : 0, nsources(o.code) > 2 ? o.c : 0};
    key = fnv(f, sizeof f, key);
    if (o.code == BRK_EX_TAP) key = fnv(o.offset, sizeof o.offset, key);
    if (o.code == BRK_EX_CONST) key = fnv(&o.imm, sizeof o.imm, key);
  }
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  std::string src = generate(ops, nops, result_reg, R);
  auto it = cache().find(key);
  auto load = [](brk_expr_kernel *k) -> int {
    hipError_t e = hipModuleLoadData(&k->module, k->code.data());
    if (e != hipSuccess) return brk_fail(BRK_EHIP, "hipModuleLoadData: %s", hipGetErrorString(e));
    e = hipModuleGetFunction(&k->fn, k->module, k->s7() ? kS7List : (k->star ? (k->radius == 2 ? kSw2 : kSw4) : (k->cube ? (k->radius == 1 ? kCube1 : kCube2) : "k_expr_jit")));
    if (e == hipSuccess && k->s7()) e = hipModuleGetFunction(&k->fn_range, k->module, kS7Range);
    if (e != hipSuccess) {
      (void)hipModuleUnload(k->module);
      k->module = nullptr, k->fn = nullptr, k->fn_range = nullptr;
      return brk_fail(BRK_EHIP, "hipModuleGetFunction: %s", hipGetErrorString(e));
    }
    return BRK_OK;
  };
  if (it != cache().end() && it->second.k->source == src) {
    if (!it->second.k->fn) // unloaded by brk_finalize
      if (int rc = load(it->second.k.get())) return rc;
    ++it->second.refs;
    *out = it->second.k.get();
    return BRK_OK;
  }
  if (it != cache().end()) return brk_fail(BRK_EINVAL, "two different programs share the hash %016llx", (unsigned long long)key);
  // BRK_JIT_CACHE_DIR: code objects kept across processes (a service that restarts does not compile its scripts again); the file name
  // covers everything the code depends on: generated text, the two embedded headers, target, options
  std::string disk;
  if (const char *dir = getenv("BRK_JIT_CACHE_DIR"); dir && *dir && have_device) {
    static const uint64_t hdr_hash = fnv(kJitCubeHeader, sizeof kJitCubeHeader, fnv(kJitStarWgHeader, sizeof kJitStarWgHeader, fnv(kJitStar7Header, sizeof kJitStar7Header, fnv(kJitSbaseHeader, sizeof kJitSbaseHeader, fnv(kJitAbiHeader, sizeof kJitAbiHeader, fnv(kJitDeviceHeader, sizeof kJitDeviceHeader))))));
    const uint64_t h1 = fnv(arch, std::strlen(arch), fnv(src.data(), src.size(), hdr_hash)), h2 = fnv(src.data(), src.size(), 0x9e3779b97f4a7c15ull ^ hdr_hash);
    char name[96];
    snprintf(name, sizeof name, "/brk_expr_%016llx%016llx.co", (unsigned long long)h1, (unsigned long long)h2);
    disk = std::string(dir) + name;
    if (FILE *f = fopen(disk.c_str(), "rb")) {
      std::string code;
      char chunk[65536];
      for (size_t got; (got = fread(chunk, 1, sizeof chunk, f)) > 0;) code.append(chunk, got);
      fclose(f);
      if (code.size() > 64 && std::memcmp(code.data(), "\177ELF", 4) == 0) {
        std::unique_ptr<brk_expr_kernel> k(new brk_expr_kernel);
        k->radius = R, k->device = dev, k->star = star_shaped(ops, nops), k->cube = cube_form(ops, nops, R), k->from_disk = true, k->source = src, k->code = std::move(code);
        if (load(k.get()) == BRK_OK) {
          *out = k.get();
          Cached &c = cache()[key];
          c.k = std::move(k), c.refs = 1;
          return BRK_OK;
        }
        (void)hipGetLastError(); // a stale or foreign file: compile
      }
    }
  }
  Rtc *rt = rtc();
  if (!rt) return brk_fail(BRK_ENODEV, "%s (brk_stencil_apply_expr runs a program without a compiler)", rtc_state().why);
  hiprtcProgram prog = nullptr;
  const char *hdr[] = {kJitDeviceHeader, kJitAbiHeader, kJitSbaseHeader, kJitStar7Header, kJitStarWgHeader, kJitCubeHeader};
  const char *names[] = {"brick-device.h", "brk.h", "brk_sbase.h", "brk_star7_kernel.h", "brk_starwg_kernel.h", "brk_cubedense_kernel.h"};
  hiprtcResult r = rt->Create(&prog, src.c_str(), "k_expr_jit.hip", 6, hdr, names);
  if (r != HIPRTC_SUCCESS) return brk_fail(BRK_EHIP, "hiprtcCreateProgram: %s", rt->ErrorString(r));
  // -ffp-contract=off: a script's `a * b + c` stays two roundings, as in the interpreter and in the checker
  // -load-store-opt off, as in the library's own build (Makefile): LDS accesses stay ds_read_b64 / ds_write_b64 -- merged into ds_read2_b64
  // they move half the bytes per LDS cycle on gfx950
  const char *opts[] = {arch, "-O3", "-std=c++17", "-ffp-contract=off", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt"};
  r = rt->Compile(prog, 8, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    rt->LogSize(prog, &ls);
    std::string log(ls ? ls : 1, '\0');
    if (ls) rt->Log(prog, &log[0]);
    rt->Destroy(&prog);
    return brk_fail(BRK_EHIP, "hiprtcCompileProgram: %s\n%.1500s", rt->ErrorString(r), log.c_str());
  }
  size_t cs = 0;
  rt->CodeSize(prog, &cs);
  std::string code(cs, '\0');
  rt->Code(prog, &code[0]);
  rt->Destroy(&prog);
  { // the kernels are looked up by their (Itanium) names further down: a drift between those and the headers shows here, GPU or not
    const bool star = star_shaped(ops, nops);
    const char *want[2] = {star && R == 1 ? kS7List : (star ? (R == 2 ? kSw2 : kSw4) : (cube_form(ops, nops, R) ? (R == 1 ? kCube1 : kCube2) : "k_expr_jit")),
                           star && R == 1 ? kS7Range : nullptr};
    for (const char *w : want)
      if (w && code.find(w) == std::string::npos) return brk_fail(BRK_EHIP, "the compiled unit does not define %s (brk_jit.hip and the kernel headers disagree)", w);
  }
  if (!have_device) return brk_fail(BRK_ENODEV, "the program compiled for %s (%zu bytes of code) but there is no device to load it on", arch + 15, cs);
  if (!disk.empty()) { // best effort, atomically: a reader never sees half a file
    const std::string tmp = disk + "." + std::to_string((unsigned long long)getpid());
    if (FILE *f = fopen(tmp.c_str(), "wb")) {
      const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
      if (fclose(f) != 0 || !ok || rename(tmp.c_str(), disk.c_str()) != 0) (void)remove(tmp.c_str());
    }
  }
  std::unique_ptr<brk_expr_kernel> k(new brk_expr_kernel);
  k->radius = R, k->device = dev, k->star = star_shaped(ops, nops), k->cube = cube_form(ops, nops, R), k->source = std::move(src), k->code = std::move(code);
  if (int rc = load(k.get())) return rc;
  *out = k.get();
  Cached &c = cache()[key];
  c.k = std::move(k), c.refs = 1;
  return BRK_OK;
}

extern "C" const char *brk_expr_kernel_source(const brk_expr_kernel *k) { return k ? k->source.c_str() : nullptr; }
extern "C" int brk_expr_kernel_from_disk(const brk_expr_kernel *k) { return k && k->from_disk ? 1 : 0; }

extern "C" int brk_expr_apply(const brk_expr_kernel *k, const uint32_t *adj, const double *in, double *out, size_t step, const uint32_t *ids, uint32_t first,
                              uint32_t n, brk_stream s) {
  if (!k) return brk_fail(BRK_EINVAL, "null kernel");
  if (!k->fn) return brk_fail(BRK_EINVAL, "the kernel was unloaded by brk_finalize: compile the program again");
  if (!adj || !in || !out || in == out) return brk_fail(BRK_EINVAL, "bad device pointers");
  if (step < 512) return brk_fail(BRK_EINVAL, "step_elems %zu < 512", step);
  // the tuned kernels a star-shaped program is compiled into move halo rows in 16-byte lanes (as for the built-in stencils)
  if (((uintptr_t)in | (uintptr_t)out) % 16 || step % 2) return brk_fail(BRK_EINVAL, "brick fields must be 16-byte aligned (in %p, out %p, step_elems %zu)", (const void *)in, (void *)out, step);
  if (n == 0) return BRK_OK;
  hipError_t e;
  if (k->s7()) { // k_s7_wg<kUser, 8, list / range>: eight bricks per workgroup, (adj, in, out, step, ids, first, n, Box, Coeffs, regmask, faces)
    struct { uint32_t base, ni_tot, nj_tot, lo[3], cnt[3]; } box = {}; // = brklib::Box of brk_star7_kernel.h (box mode only)
    brklib::Coeffs cf;
    std::memset(&cf, 0, sizeof cf);
    const uint32_t *none = nullptr, *faces = nullptr; // a registered adjacency table: one 32-byte row of face ids per brick instead of the 108-byte row
    if (int rc = brki_face_table_lock(adj, &faces)) return rc;
    const uint32_t wgs = (((n + 7) / 8 + 7) / 8) * 8;
    void *args[] = {&adj, &in, &out, &step, &ids, &first, &n, &box, &cf, &none, &faces};
    brki_note_kernel(ids ? "k_expr_jit<1,s7,list>" : "k_expr_jit<1,s7,range>");
    e = hipModuleLaunchKernel(ids ? k->fn : k->fn_range, wgs, 1, 1, 512, 1, 1, 0, static_cast<hipStream_t>(s), args, nullptr);
    brki_adj_unlock();
  } else if (k->star) { // k_star_wg<kStarUser2 | kStarUser4, 4, OCC>: four bricks per workgroup, (adj, in, out, step, ids, first, n, Coeffs, faces)
    brklib::Coeffs cf;
    std::memset(&cf, 0, sizeof cf);
    const uint32_t *faces = nullptr;
    if (int rc = brki_face_table_lock(adj, &faces)) return rc;
    const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
    void *args[] = {&adj, &in, &out, &step, &ids, &first, &n, &cf, &faces};
    brki_note_kernel(k->radius == 2 ? "k_expr_jit<2,star_wg>" : (k->radius == 3 ? "k_expr_jit<3,star_wg>" : "k_expr_jit<4,star_wg>"));
    e = hipModuleLaunchKernel(k->fn, wgs, 1, 1, 256, 1, 1, 0, static_cast<hipStream_t>(s), args, nullptr);
    brki_adj_unlock();
  } else if (k->cube) { // k_cube_dense<kCubeUser1, 4, 4, 1> / <kCubeUser2, 4, 3, 5>: (adj, in, out, step, ids, first, n, Coeffs, diag, linkmask, cubemask, cs1, cs2)
    brklib::Coeffs cf;
    std::memset(&cf, 0, sizeof cf);
    int diag = 0;
    const uint32_t *none = nullptr;
    uint32_t zero = 0;
    const uint32_t wgs = (((n + 3) / 4 + 7) / 8) * 8;
    void *args[] = {&adj, &in, &out, &step, &ids, &first, &n, &cf, &diag, &none, &none, &zero, &zero};
    brki_note_kernel(k->radius == 1 ? "k_expr_jit<1,cube_dense>" : "k_expr_jit<2,cube_dense>");
    e = hipModuleLaunchKernel(k->fn, wgs, 1, 1, 256, 1, 1, 0, static_cast<hipStream_t>(s), args, nullptr);
  } else {
    const uint32_t wpw = waves_per_group(k->radius, k->star);
    const uint32_t wgs = (((n + wpw - 1) / wpw + 7) / 8) * 8;
    void *args[] = {&adj, &in, &out, &step, &ids, &first, &n};
    static const char *const names[] = {"k_expr_jit<1>", "k_expr_jit<2>", "k_expr_jit<3>", "k_expr_jit<4>",
                                        "k_expr_jit<1,star>", "k_expr_jit<2,star>", "k_expr_jit<3,star>", "k_expr_jit<4,star>"};
    brki_note_kernel(names[k->radius - 1 + (k->star ? 4 : 0)]);
    e = hipModuleLaunchKernel(k->fn, wgs, 1, 1, 64 * wpw, 1, 1, 0, static_cast<hipStream_t>(s), args, nullptr);
  }
  if (e != hipSuccess) return brk_fail(BRK_EHIP, "hipModuleLaunchKernel: %s", hipGetErrorString(e));
  return BRK_OK;
}

// the reference's launch shape for a lexicographic grid (stencils/3axis.hip.cpp:36-39 `b = grid[tk][tj][ti]` over an interior box): the
// box is walked in the order the built-in stencils use (one slab of j-rows per XCD, blocks of 8 rows swept over all k, i fastest)
extern "C" int brk_expr_apply_box(const brk_expr_kernel *k, const uint32_t *adj, const double *in, double *out, size_t step, uint32_t base,
                                  const uint32_t tot[3], const uint32_t lo[3], const uint32_t cnt[3], brk_stream s) {
  if (!tot || !lo || !cnt) return brk_fail(BRK_EINVAL, "null box geometry");
  for (int a = 0; a < 3; ++a)
    if (lo[a] + cnt[a] > tot[a]) return brk_fail(BRK_EINVAL, "box exceeds the grid on axis %d", a);
  if ((uint64_t)base + (uint64_t)tot[0] * tot[1] * tot[2] > 0xFFFFFFFFull) return brk_fail(BRK_EINVAL, "brick ids overflow 32 bits");
  if ((uint64_t)cnt[0] * cnt[1] * cnt[2] == 0) return k ? BRK_OK : brk_fail(BRK_EINVAL, "null kernel");
  const uint32_t *ids = nullptr;
  uint32_t n = 0;
  if (int rc = brki_box_list_lock(base, tot, lo, cnt, &ids, &n)) return rc;
  const int rc = brk_expr_apply(k, adj, in, out, step, ids, 0, n, s);
  const int rc2 = brki_box_list_unlock(s);
  return rc ? rc : rc2;
}

extern "C" int brk_expr_release(brk_expr_kernel *k) {
  if (!k) return BRK_OK;
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  for (auto it = cache().begin(); it != cache().end(); ++it)
    if (it->second.k.get() == k) {
      if (--it->second.refs > 0) return BRK_OK;
      if (k->module) {
        (void)hipDeviceSynchronize();
        (void)hipModuleUnload(k->module);
      }
      cache().erase(it);
      return BRK_OK;
    }
  return brk_fail(BRK_EINVAL, "not a kernel of brk_expr_compile (released already?)");
}

// brk_finalize: modules still loaded go before the runtime's own exit handlers run
extern "C" void brki_jit_finalize(void) {
  std::lock_guard<std::mutex> lock(g_jit_mutex);
  for (auto &kv : cache())
    if (kv.second.k && kv.second.k->module) {
      (void)hipModuleUnload(kv.second.k->module);
      kv.second.k->module = nullptr, kv.second.k->fn = nullptr, kv.second.k->fn_range = nullptr; // handles stay valid objects; brk_expr_apply refuses them until the program is compiled again
    }
}
