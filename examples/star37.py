# A user-written stencil in the reference's script syntax: a radius-3 star (19 points) with different weights up- and
# downwind -- coeff[d] at distance +d, coeff[3 + d] at distance -d on every axis, coeff[0] in the centre.  Not one of the
# built-in stencils and beyond the 125-point cube: tools/brkscatter reads it as data and routes the brick("star37.py", ...) line to
# the star-shaped tap path (brklib::brick_apply_star_taps_static, radius up to 4).
from st.expr import Index, ConstRef
from st.grid import Grid

i, j, k = Index(0), Index(1), Index(2)
bIn = Grid("bIn", 3)
bOut = Grid("bOut", 3)
c = [ConstRef("coeff[%d]" % q) for q in range(7)]

acc = c[0] * bIn(i, j, k)
for d in (1, 2, 3):
    acc = acc + c[d] * (bIn(i + d, j, k) + bIn(i, j + d, k) + bIn(i, j, k + d))
    acc = acc + c[3 + d] * (bIn(i - d, j, k) + bIn(i, j - d, k) + bIn(i, j, k - d))
bOut(i, j, k).assign(acc)

STENCIL = [bOut]
