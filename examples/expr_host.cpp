// expr_host -- a stencil that exists only as DATA in the host program: built op by op (what bricklib_amd/spec.py makes of a script),
// run interpreted (brk_stencil_apply_expr) and compiled at run time (brk_expr_compile / brk_expr_apply, hiprtc), both compared with a
// CPU loop.  The reference's counterpart is its generate-then-compile build step (codegen/vecscatter); a C or C++ host without Python
// uses the C ABI like this.        out = | in - 0.25 * max(in(i-1), in(i+1)) | + min(in(j+2), in(k-1)) * c
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

#include "brick-hip.h"
#include "bricksetup.h"
#include "brk.h"

int main() {
  constexpr long NB = 6; // 6^3 bricks, the inner 4^3 computed
  unsigned *grid = nullptr;
  BrickInfo<3> info = init_grid<3>(grid, {NB, NB, NB});
  const size_t nb = NB * NB * NB, elems = nb * 512;
  std::vector<double> in(elems), got_i(elems), got_c(elems);
  for (size_t q = 0; q < elems; ++q) in[q] = std::sin(0.37 * (double)q) * 0.5;
  double *din, *dout;
  unsigned *dadj, *dids;
  gpuCheck(hipMalloc(&din, elems * 8));
  gpuCheck(hipMalloc(&dout, elems * 8));
  gpuCheck(hipMalloc(&dadj, nb * 27 * 4));
  gpuCheck(hipMemcpy(din, in.data(), elems * 8, hipMemcpyHostToDevice));
  gpuCheck(hipMemcpy(dadj, info.adj, nb * 27 * 4, hipMemcpyHostToDevice));
  std::vector<unsigned> ids;
  for (long k = 1; k < NB - 1; ++k)
    for (long j = 1; j < NB - 1; ++j)
      for (long i = 1; i < NB - 1; ++i) ids.push_back(grid[(k * NB + j) * NB + i]);
  gpuCheck(hipMalloc(&dids, ids.size() * 4));
  gpuCheck(hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));

  const double c = 0.75;
  auto tap = [](unsigned dst, int di, int dj, int dk) { brk_expr_op o{}; o.code = BRK_EX_TAP, o.dst = dst, o.offset[0] = di, o.offset[1] = dj, o.offset[2] = dk; return o; };
  auto imm = [](unsigned dst, double v) { brk_expr_op o{}; o.code = BRK_EX_CONST, o.dst = dst, o.imm = v; return o; };
  auto op = [](unsigned code, unsigned dst, unsigned a, unsigned b = 0) { brk_expr_op o{}; o.code = code, o.dst = dst, o.a = a, o.b = b; return o; };
  const brk_expr_op prog[] = {tap(0, -1, 0, 0), tap(1, 1, 0, 0), op(BRK_EX_MAX, 0, 0, 1), imm(1, 0.25), op(BRK_EX_MUL, 0, 1, 0), tap(1, 0, 0, 0),
                              op(BRK_EX_SUB, 0, 1, 0), op(BRK_EX_ABS, 0, 0), tap(1, 0, 2, 0), tap(2, 0, 0, -1), op(BRK_EX_MIN, 1, 1, 2), imm(2, c),
                              op(BRK_EX_MUL, 1, 1, 2), op(BRK_EX_ADD, 3, 0, 1)};
  const unsigned nops = sizeof prog / sizeof *prog, result = 3;

  if (brk_stencil_apply_expr(dadj, din, dout, 512, dids, 0, (unsigned)ids.size(), prog, nops, result, nullptr) != BRK_OK) {
    printf("interpreter: %s\n", brk_last_error());
    return 1;
  }
  gpuCheck(hipMemcpy(got_i.data(), dout, elems * 8, hipMemcpyDeviceToHost));
  gpuCheck(hipMemset(dout, 0, elems * 8));
  brk_expr_kernel *kernel = nullptr;
  if (brk_expr_compile(prog, nops, result, &kernel) != BRK_OK) {
    printf("brk_expr_compile: %s\n", brk_last_error());
    return 1;
  }
  if (brk_expr_apply(kernel, dadj, din, dout, 512, dids, 0, (unsigned)ids.size(), nullptr) != BRK_OK) {
    printf("brk_expr_apply: %s\n", brk_last_error());
    return 1;
  }
  gpuCheck(hipMemcpy(got_c.data(), dout, elems * 8, hipMemcpyDeviceToHost));
  printf("compiled kernel: %s\n", brk_last_kernel());
  // the same bricks as the reference's launch shape: the interior box of the lexicographic grid, no id list
  std::vector<double> got_b(elems);
  gpuCheck(hipMemset(dout, 0, elems * 8));
  const uint32_t tot[3] = {NB, NB, NB}, lo[3] = {1, 1, 1}, cnt[3] = {NB - 2, NB - 2, NB - 2};
  if (brk_expr_apply_box(kernel, dadj, din, dout, 512, 0, tot, lo, cnt, nullptr) != BRK_OK) {
    printf("brk_expr_apply_box: %s\n", brk_last_error());
    return 1;
  }
  gpuCheck(hipMemcpy(got_b.data(), dout, elems * 8, hipMemcpyDeviceToHost));
  printf("box form == list form: %s\n", got_b == got_c ? "yes" : "NO");

  // CPU loop over the same cells through the accessor semantics (neighbour brick = adjacency entry)
  auto at = [&](unsigned b, int k, int j, int i) {
    const int dk = k < 0 ? -1 : (k > 7 ? 1 : 0), dj = j < 0 ? -1 : (j > 7 ? 1 : 0), di = i < 0 ? -1 : (i > 7 ? 1 : 0);
    const unsigned nbk = info.adj[b][(dk + 1) * 9 + (dj + 1) * 3 + di + 1];
    return in[(size_t)nbk * 512 + ((k - 8 * dk) * 8 + (j - 8 * dj)) * 8 + (i - 8 * di)];
  };
  bool same_i = true, same_c = true;
  for (unsigned b : ids)
    for (int k = 0; k < 8; ++k)
      for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) {
          const double m = std::max(at(b, k, j, i - 1), at(b, k, j, i + 1)) * 0.25;
          const double want = std::abs(at(b, k, j, i) - m) + std::min(at(b, k, j + 2, i), at(b, k - 1, j, i)) * c;
          const size_t q = (size_t)b * 512 + (k * 8 + j) * 8 + i;
          same_i = same_i && got_i[q] == want;
          same_c = same_c && got_c[q] == want;
        }
  printf("interpreted (brk_stencil_apply_expr): %s\n", same_i ? "result match" : "result mismatch");
  printf("compiled at run time (brk_expr_compile): %s\n", same_c ? "result match" : "result mismatch");
  brk_expr_release(kernel);
  free(info.adj);
  free(grid);
  brk_finalize();
  return same_i && same_c && got_b == got_c ? 0 : 1;
}
