# A user-written stencil BEYOND the 125-point cube in the reference's script syntax: all 7^3 = 343 points of the radius-3 cube, one
# coefficient per L1 distance (0..9).  tools/brkscatter reads it as data; with hundreds of taps the rewritten brick(...) line keeps
# offsets and weights in LDS (brklib::brick_apply_taps_table), hosts pass the same table to brk_stencil_apply_taps.
from st.expr import Index, ConstRef
from st.grid import Grid

i, j, k = Index(0), Index(1), Index(2)
bIn = Grid("bIn", 3)
bOut = Grid("bOut", 3)

acc = None
for dk in range(-3, 4):
    for dj in range(-3, 4):
        for di in range(-3, 4):
            term = ConstRef("coeff[%d]" % (abs(di) + abs(dj) + abs(dk))) * bIn(i + di, j + dj, k + dk)
            acc = term if acc is None else acc + term
bOut(i, j, k).assign(acc)

STENCIL = [bOut]
