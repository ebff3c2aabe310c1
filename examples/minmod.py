# A user-written NON-LINEAR stencil in the reference's script syntax: a minmod-limited slope per axis (Func, If, comparisons and
# products of grid references -- what the reference's stencils/cond.py is made of).  Not a sum of taps: tools/brkscatter writes the
# script's operations out as straight-line C++ over the staged tile (brklib::brick_apply_fn), hosts run it with
# brk_stencil_apply_expr (bricklib_amd.apply_expr).
from st.expr import Index, ConstRef, If
from st.grid import Grid
from st.func import Func

i, j, k = Index(0), Index(1), Index(2)
bIn = Grid("bIn", 3)
bOut = Grid("bOut", 3)
smaller = Func("fmin", 2)
mag = Func("fabs", 1)


def minmod(left, right):
    return If(left * right > 0, If(left > 0, 1.0, -1.0) * smaller(mag(left), mag(right)), 0.0)


here = bIn(i, j, k)
slopes = None
for axis in range(3):
    up = [i, j, k]
    down = [i, j, k]
    up[axis] = up[axis] + 1
    down[axis] = down[axis] - 1
    s = minmod(here - bIn(*down), bIn(*up) - here)
    slopes = s if slopes is None else slopes + s
bOut(i, j, k).assign(here - ConstRef("coeff[0]") * slopes)

STENCIL = [bOut]
