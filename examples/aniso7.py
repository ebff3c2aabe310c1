# A user-written STAR-shaped stencil in the reference's script syntax that is none of the built-in stars: a 7-point stencil with one
# weight per direction (upwind-biased advection-diffusion).  tools/brkscatter routes it to the star staging
# (brklib::brick_apply_star_taps_static<1, 7>): own brick and six faces, no edge or corner gathers.
from st.expr import Index, ConstRef
from st.grid import Grid

i, j, k = Index(0), Index(1), Index(2)
bIn = Grid("bIn", 3)
bOut = Grid("bOut", 3)
c = [ConstRef("coeff[%d]" % q) for q in range(7)]

bOut(i, j, k).assign(c[0] * bIn(i, j, k) + c[1] * bIn(i + 1, j, k) + 0.5 * c[2] * bIn(i - 1, j, k) + c[3] * bIn(i, j + 1, k) -
                     c[4] * bIn(i, j - 1, k) + c[5] * bIn(i, j, k + 1) + 0.25 * c[6] * bIn(i, j, k - 1))

STENCIL = [bOut]
